#!/bin/bash
# tune_ab2.sh for another workload / algorithm:  bash tools/tune_ab3.sh "--workload jungfrau9m" - rows_ahead=2
extra=$1; shift
for rep in 1 2 3; do for t in "$@"; do
  tune=""; [ "$t" != "-" ] && tune="--tune $t"
  python bench.py --no-cpu-baseline --no-streamed --no-cli-e2e --steps 20 --warmup 5 --reps 7 $extra $tune > /tmp/x.json 2>/tmp/x.err || { echo "[$t] FAILED"; tail -3 /tmp/x.err; continue; }
  python -c "
import json; d=json.load(open('/tmp/x.json')); r=d['roofline']; print('[$t]:', d['value'], 'fps | ms/step', d['ms_per_step'], 'steady', d['steady_ms_per_step'], 'drain', d['drain_ms'], '| kernel', r['ms_per_launch'], '| checked', d.get('results_checked'))"
done; done
