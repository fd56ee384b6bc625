#!/bin/bash
# extended dispersion: whole-pipeline step time against the number of streams in flight (how much of the sparse stage overlaps)
for st in 1 2 4 6 8; do
  python bench.py --no-cpu-baseline --no-streamed --no-cli-e2e --steps 60 --reps 3 --streams $st --algorithm dispersion_extended "$@" > /tmp/x.json 2>/tmp/x.err
  python - <<PY
import json
d=json.load(open("/tmp/x.json")); r=d["roofline"]
print("streams $st: value", d["value"], "ms/step", d["ms_per_step"], "first", r["ms_per_launch"], "rest", r["exact_kernel_ms_per_launch"], "stages", d["stage_ms_last_batch"])
PY
done
