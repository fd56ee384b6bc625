#!/bin/bash
# The driver-style line (--steps 20 --warmup 5) under tuning sets, alternating on one box:  bash tools/tune_ab2.sh "-" "sparse_bands=2" ...
for rep in 1 2 3; do for t in "$@"; do
  tune=""; [ "$t" != "-" ] && tune="--tune $t"
  python bench.py --no-cpu-baseline --no-streamed --no-cli-e2e --steps 20 --warmup 5 --reps 7 $tune > /tmp/x.json 2>/tmp/x.err || { echo "[$t] FAILED"; tail -3 /tmp/x.err; continue; }
  python -c "
import json; d=json.load(open('/tmp/x.json')); r=d['roofline']; print('[$t]:', d['value'], 'fps | ms/step', d['ms_per_step'], d['repetitions']['ms_per_step'], 'steady', d['steady_ms_per_step'], 'drain', d['drain_ms'], '| kernel', r['ms_per_launch'], '| checked', d.get('results_checked'))"
done; done
