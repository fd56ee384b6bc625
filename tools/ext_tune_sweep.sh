#!/bin/bash
# extended algorithm: one tuning key over several values, alternating, three rounds:   tools/ext_tune_sweep.sh <key> <v1> <v2> ...
key=$1; shift
for rep in 1 2 3; do for v in "$@"; do
  python bench.py --no-cpu-baseline --no-streamed --no-cli-e2e --steps 60 --warmup 5 --reps 5 --algorithm dispersion_extended --tune $key=$v > /tmp/x.json 2>/tmp/x.err || { echo "$key=$v FAILED"; tail -3 /tmp/x.err; continue; }
  python -c "
import json; d=json.load(open('/tmp/x.json')); r=d['roofline']; print('extended $key=$v:', d['value'], 'fps | ms/step', d['ms_per_step'], 'steady', d['steady_ms_per_step'], '| first pass (events)', r['ms_per_launch'], '| checked', d.get('results_checked'))"
done; done
