#!/bin/bash
# Consecutive streaming kernels handed over through two dense streams (tuning dense_overlap = 1) against one dense stream (0).
[ $# -eq 0 ] && set -- 32
for rep in 1 2 3; do for b in "$@"; do for t in 1 0; do
python bench.py --no-cpu-baseline --no-streamed --no-cli-e2e --steps 20 --warmup 5 --reps 5 --batch $b --tune dense_overlap=$t > /tmp/x.json 2>/tmp/x.err || { echo "batch $b overlap $t FAILED"; tail -3 /tmp/x.err; continue; }
python -c "
import json; d=json.load(open('/tmp/x.json')); r=d['roofline']; print('batch $b overlap $t:', d['value'], 'fps | ms/step', d['ms_per_step'], 'steady', d['steady_ms_per_step'], '| drain', d['drain_ms'], '| kernel (events)', r['ms_per_launch'], 'alone', r['ms_per_launch_alone'], '| checked', d.get('results_checked'))"
done; done; done
