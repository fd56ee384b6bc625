#!/bin/bash
# The band launch's stream priority (tuning sparse_priority: 0 highest, 1 lowest, 2 normal), by frames per batch.
[ $# -eq 0 ] && set -- 32 48
for rep in 1 2; do for b in "$@"; do for t in 0 1 2; do
python bench.py --no-cpu-baseline --no-streamed --no-cli-e2e --steps 40 --warmup 5 --reps 5 --batch $b --tune sparse_priority=$t > /tmp/x.json 2>/tmp/x.err || { echo "batch $b prio $t FAILED"; tail -3 /tmp/x.err; continue; }
python -c "
import json; d=json.load(open('/tmp/x.json')); print('batch $b prio $t:', d['value'], 'fps | ms/step', d['ms_per_step'], '| per frame us', round(d['ms_per_step']*1000/$b,3), 'steady/frame', round(d['steady_ms_per_step']*1000/$b,3), 'kernel/frame', round(d['roofline']['ms_per_launch']*1000/$b,3), '| drain', d['drain_ms'], '| checked', d.get('results_checked'), d['stage_ms_last_batch'])"
done; done; done
