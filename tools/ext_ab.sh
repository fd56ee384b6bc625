#!/bin/bash
# extended dispersion: A/B of tuning sets through bench.py
for t in "" "ccl_grid=64" "ccl_grid=128" "sparse_stage=3" "ccl_grid=16"; do
  python bench.py --no-cpu-baseline --no-streamed --no-cli-e2e --steps 60 --reps 3 --algorithm dispersion_extended ${t:+--tune $t} > /tmp/x.json 2>/tmp/x.err
  python - <<PY
import json
d=json.load(open("/tmp/x.json")); r=d["roofline"]
print("tune [$t]: value", d["value"], "ms/step", d["ms_per_step"], "first", r["ms_per_launch"], "rest", r["exact_kernel_ms_per_launch"], "ccl", d["stage_ms_last_batch"]["ccl"])
PY
done
