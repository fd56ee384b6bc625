#!/bin/bash
# extended algorithm: rows in flight in the first pass (2 / 3), alternating
for rep in 1 2 3; do for v in 2 3; do
  python bench.py --no-cpu-baseline --no-streamed --no-cli-e2e --steps 60 --warmup 5 --reps 5 --algorithm dispersion_extended --tune rows_ahead=$v > /tmp/x.json 2>/tmp/x.err || { echo "FAILED"; tail -3 /tmp/x.err; continue; }
  python -c "
import json; d=json.load(open('/tmp/x.json')); r=d['roofline']; print('extended rows_ahead=$v:', d['value'], 'fps | ms/step', d['ms_per_step'], 'steady', d['steady_ms_per_step'], '| first pass (events)', r['ms_per_launch'], '| checked', d.get('results_checked'))"
done; done
