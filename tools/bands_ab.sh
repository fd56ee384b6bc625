#!/bin/bash
# The sparse stage in small workgroups (tuning sparse_bands = 1) against the one-workgroup launch (0), by frames per batch.
#   gpurun -- 'bash tools/run_logged.sh <tag> bash tools/bands_ab.sh [batches...]'
[ $# -eq 0 ] && set -- 32 48 56
for rep in 1 2; do for b in "$@"; do for t in 1 0; do
python bench.py --no-cpu-baseline --no-streamed --no-cli-e2e --steps 40 --warmup 5 --reps 5 --batch $b --tune sparse_bands=$t > /tmp/x.json 2>/tmp/x.err || { echo "batch $b bands $t FAILED"; tail -3 /tmp/x.err; continue; }
python -c "
import json; d=json.load(open('/tmp/x.json')); print('batch $b bands $t:', d['value'], 'fps | ms/step', d['ms_per_step'], '| per frame us', round(d['ms_per_step']*1000/$b,3), 'steady/frame', round(d['steady_ms_per_step']*1000/$b,3), 'kernel/frame', round(d['roofline']['ms_per_launch']*1000/$b,3), '| drain', d['drain_ms'], '| checked', d.get('results_checked'), d['stage_ms_last_batch'])"
done; done; done
