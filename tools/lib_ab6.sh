#!/bin/bash
# this build against another build of the library, alternating on one box, 100-step regions:   tools/lib_ab6.sh <other .so> [bench args]
other=$1; shift
for rep in 1 2 3; do for lib in "" $other; do
  FFS_HIP_LIB=$lib python bench.py --no-cpu-baseline --no-streamed --no-cli-e2e --steps 100 --warmup 5 --reps 5 "$@" > /tmp/x.json 2>/tmp/x.err || { echo "${lib:-this build} FAILED"; tail -3 /tmp/x.err; continue; }
  python -c "
import json; d=json.load(open('/tmp/x.json')); r=d['roofline']; print('${lib:-this build}:', d['value'], 'fps | ms/step', d['ms_per_step'], 'steady', d['steady_ms_per_step'], '| kernel (events)', r['ms_per_launch'], 'alone', r['ms_per_launch_alone'], '| checked', d.get('results_checked'))"
done; done
