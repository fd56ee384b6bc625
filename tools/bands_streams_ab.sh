#!/bin/bash
# Few batches in flight: the band launches (sparse_bands = 1) against the one-workgroup launch with its head start (0), by pipeline depth.
for st in 1 2 3 4; do for t in 1 0; do
python bench.py --no-cpu-baseline --no-streamed --no-cli-e2e --steps 40 --warmup 5 --reps 5 --streams $st --tune sparse_bands=$t > /tmp/x.json 2>/tmp/x.err || { echo "streams $st bands $t FAILED"; tail -3 /tmp/x.err; continue; }
python -c "
import json; d=json.load(open('/tmp/x.json')); print('streams $st bands $t:', d['value'], 'fps | ms/step', d['ms_per_step'], 'steady', d['steady_ms_per_step'], '| drain', d['drain_ms'], '| checked', d.get('results_checked'), d['stage_ms_last_batch'])"
done; done
