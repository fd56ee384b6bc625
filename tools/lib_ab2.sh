#!/bin/bash
# The tree's library against tools/ab/libffs_hip_prev.so, same box, alternating: the driver-style line.
for rep in 1 2 3; do for lib in cur prev; do
  if [ $lib = prev ]; then export FFS_HIP_LIB=$GRAFT_REPO_ROOT/tools/ab/libffs_hip_prev.so; else unset FFS_HIP_LIB; fi
  python bench.py --no-cpu-baseline --no-streamed --no-cli-e2e --steps 20 --warmup 5 --reps 5 "$@" > /tmp/x.json 2>/tmp/x.err || { echo "$lib FAILED"; tail -3 /tmp/x.err; continue; }
  python -c "
import json; d=json.load(open('/tmp/x.json')); r=d['roofline']; print('$lib:', d['value'], 'fps | ms/step', d['ms_per_step'], 'steady', d['steady_ms_per_step'], '| drain', d['drain_ms'], '| kernel (events)', r['ms_per_launch'], 'alone', r['ms_per_launch_alone'], '| checked', d.get('results_checked'))"
done; done
