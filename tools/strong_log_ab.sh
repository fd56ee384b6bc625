#!/bin/bash
# the standard path with the streaming kernel's wave logs (strong_log=1) against the bit plane (0): bench line, alternating
for t in strong_log=0 strong_log=1 strong_log=0 strong_log=1; do
  python bench.py --no-cpu-baseline --no-streamed --no-cli-e2e --steps 100 --reps 5 --tune $t "$@" > /tmp/x.json 2>/tmp/x.err
  python - <<PY
import json
d=json.load(open("/tmp/x.json")); r=d["roofline"]
print("tune [$t]: value", d["value"], "ms/step", d["ms_per_step"], "steady", d["steady_ms_per_step"], "kernel", r["ms_per_launch"], "stages", d["stage_ms_last_batch"], "spots/frame", d["config"].get("spots_per_frame"))
PY
done
