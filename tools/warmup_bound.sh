#!/bin/bash
# Upper bound of what a streaming launch without warm-up rows would gain: experiments build, bit 1024 = every wave does six rows fewer
# (results are wrong), against the full kernel, alternating; band launches as in the product (their input differs slightly).
for rep in 1 2 3; do for dbg in 0 1024; do
  FFS_EXP_K1_DEBUG=$dbg FFS_HIP_LIB=$GRAFT_REPO_ROOT/fast-feedback-service_amd/libffs_hip_exp.so python bench.py --no-cpu-baseline --no-streamed --no-cli-e2e --steps 100 --warmup 5 --reps 5 "$@" > /tmp/x.json 2>/dev/null
  python -c "
import json; d=json.load(open('/tmp/x.json')); print('dbg $dbg: fps', d['value'], 'ms/step', d['ms_per_step'], 'steady', d['steady_ms_per_step'], 'kernel (events)', d['roofline']['ms_per_launch'], 'alone', d['roofline']['ms_per_launch_alone'])"
done; done
