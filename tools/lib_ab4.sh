#!/bin/bash
# dense_overlap 0 / 1 with the runtime's default of four hardware queues and with eight (GPU_MAX_HW_QUEUES), same box, alternating
for rep in 1 2 3; do for cfg in q4_o0 q8_o0 q8_o1 q4_o1; do
  case $cfg in q4_*) unset GPU_MAX_HW_QUEUES;; q8_*) export GPU_MAX_HW_QUEUES=8;; esac
  case $cfg in *_o0) tune="--tune dense_overlap=0";; *_o1) tune="--tune dense_overlap=1";; esac
  python bench.py --no-cpu-baseline --no-streamed --no-cli-e2e --steps 40 --warmup 5 --reps 5 $tune "$@" > /tmp/x.json 2>/tmp/x.err || { echo "$cfg FAILED"; tail -3 /tmp/x.err; continue; }
  python -c "
import json; d=json.load(open('/tmp/x.json')); r=d['roofline']; print('$cfg:', d['value'], 'fps | ms/step', d['ms_per_step'], d['repetitions']['ms_per_step'], 'steady', d['steady_ms_per_step'], '| kernel (events)', r['ms_per_launch'], 'alone', r['ms_per_launch_alone'], '| checked', d.get('results_checked'))"
done; done
