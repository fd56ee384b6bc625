#!/bin/bash
# issue priority of the streaming waves (bit 1) and of the band / merge waves (bit 2), alternating on one box
for rep in 1 2 3; do for p in 1 2 3 0; do
  python bench.py --no-cpu-baseline --no-streamed --no-cli-e2e --steps 100 --warmup 5 --reps 5 --tune stream_prio=$p "$@" > /tmp/x.json 2>/tmp/x.err || { echo "prio $p FAILED"; tail -3 /tmp/x.err; continue; }
  python -c "
import json; d=json.load(open('/tmp/x.json')); r=d['roofline']; print('stream_prio $p:', d['value'], 'fps | ms/step', d['ms_per_step'], 'steady', d['steady_ms_per_step'], 'drain', d['drain_ms'], '| kernel (events)', r['ms_per_launch'], '| checked', d.get('results_checked'))"
done; done
