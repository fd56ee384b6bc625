#!/bin/bash
# Batches in flight (bench.py --streams) at the driver's arguments, alternating on one box
for rep in 1 2 3; do for st in 4 5 6 8; do
  python bench.py --no-cpu-baseline --no-streamed --no-cli-e2e --steps 20 --warmup 5 --reps 7 --streams $st > /tmp/x.json 2>/tmp/x.err || { echo "[$st] FAILED"; tail -3 /tmp/x.err; continue; }
  python -c "
import json; d=json.load(open('/tmp/x.json')); r=d['roofline']; print('streams $st:', d['value'], 'fps | ms/step', d['ms_per_step'], 'steady', d['steady_ms_per_step'], 'drain', d['drain_ms'], '| kernel', r['ms_per_launch'], '| checked', d.get('results_checked'))"
done; done
