#!/bin/bash
# standard dispersion: the one-launch sparse stage over pixels (chain_runs=1: runs only for frames beyond the LDS forest) against runs for every frame (2)
for t in chain_runs=1 chain_runs=2 chain_runs=1 chain_runs=2; do
  python bench.py --no-cpu-baseline --no-streamed --no-cli-e2e --steps 100 --reps 5 --tune $t "$@" > /tmp/x.json 2>/tmp/x.err
  python - <<PY
import json
d=json.load(open("/tmp/x.json")); r=d["roofline"]
print("tune [$t]: value", d["value"], "ms/step", d["ms_per_step"], "steady", d["steady_ms_per_step"], "kernel", r["ms_per_launch"], "stages", d["stage_ms_last_batch"])
PY
done
