#!/bin/bash
# The band wave's phases at their TRUE cost: with consecutive streaming kernels overlapped (dense_overlap = 1) there are no tails for the
# sparse work to hide in.  Experiments build (results wrong): 10 = band wave and merge return at once, 12 = after the entries are in LDS,
# 13 = after decisions, counts and placement, 11 = whole band wave, no merge.
for rep in 1 2; do for stop in 10 12 13 11; do
  FFS_EXP_CHAIN_STOP=$stop FFS_HIP_LIB=$GRAFT_REPO_ROOT/fast-feedback-service_amd/libffs_hip_exp.so python bench.py --no-cpu-baseline --no-streamed --no-cli-e2e --steps 60 --warmup 5 --reps 5 --tune dense_overlap=1 "$@" > /tmp/x.json 2>/dev/null
  python -c "
import json; d=json.load(open('/tmp/x.json')); print('stop $stop overlap 1: fps', d['value'], 'ms/step', d['ms_per_step'], 'steady', d['steady_ms_per_step'], 'kernel (events)', d['roofline']['ms_per_launch'])"
done; done
