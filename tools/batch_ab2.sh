#!/bin/bash
# Frames per step at the driver's arguments, alternating on one box (fixtures hold up to 56 frames per rank).
for rep in 1 2 3; do for b in 32 48 56; do
  python bench.py --no-cpu-baseline --no-streamed --no-cli-e2e --steps 20 --warmup 5 --reps 7 --batch $b > /tmp/x.json 2>/tmp/x.err || { echo "[$b] FAILED"; tail -3 /tmp/x.err; continue; }
  python -c "
import json; d=json.load(open('/tmp/x.json')); r=d['roofline']; print('batch $b:', d['value'], 'fps | per frame us', round(d['ms_per_step']*1000/$b,3), 'steady/frame', round(d['steady_ms_per_step']*1000/$b,3), 'drain', d['drain_ms'], '| kernel/frame', round(r['ms_per_launch']*1000/$b,3), '| checked', d.get('results_checked'))"
done; done
