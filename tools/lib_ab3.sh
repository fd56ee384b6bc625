#!/bin/bash
# prev library | current library with dense_overlap 0 | 1, same box, alternating
for rep in 1 2 3; do for cfg in prev cur0 cur1; do
  tune=""
  if [ $cfg = prev ]; then export FFS_HIP_LIB=$GRAFT_REPO_ROOT/tools/ab/libffs_hip_prev.so; else unset FFS_HIP_LIB; fi
  [ $cfg = cur0 ] && tune="--tune dense_overlap=0"
  [ $cfg = cur1 ] && tune="--tune dense_overlap=1"
  python bench.py --no-cpu-baseline --no-streamed --no-cli-e2e --steps 40 --warmup 5 --reps 5 $tune "$@" > /tmp/x.json 2>/tmp/x.err || { echo "$cfg FAILED"; tail -3 /tmp/x.err; continue; }
  python -c "
import json; d=json.load(open('/tmp/x.json')); r=d['roofline']; print('$cfg:', d['value'], 'fps | ms/step', d['ms_per_step'], d['repetitions']['ms_per_step'], 'steady', d['steady_ms_per_step'], '| kernel (events)', r['ms_per_launch'], 'alone', r['ms_per_launch_alone'], '| checked', d.get('results_checked'))"
done; done
