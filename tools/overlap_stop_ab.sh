#!/bin/bash
# Would consecutive streaming kernels handed over through two dense streams (dense_overlap = 1) win if the sparse stage fitted the holes
# streaming waves leave?  Experiments build, sparse launches cut short (results are wrong): 10 = band wave and merge return at once,
# 11 = whole band wave, no merge, 0 = everything -- each with dense_overlap 0 and 1, alternating.
for rep in 1 2; do for stop in 10 11 0; do for t in 0 1; do
  FFS_EXP_CHAIN_STOP=$stop FFS_HIP_LIB=$GRAFT_REPO_ROOT/fast-feedback-service_amd/libffs_hip_exp.so python bench.py --no-cpu-baseline --no-streamed --no-cli-e2e --steps 60 --warmup 5 --reps 5 --tune dense_overlap=$t "$@" > /tmp/x.json 2>/dev/null   # (the self-check fails, as it must; the line is printed first)
  python -c "
import json; d=json.load(open('/tmp/x.json')); print('stop $stop overlap $t: fps', d['value'], 'ms/step', d['ms_per_step'], 'steady', d['steady_ms_per_step'], 'kernel (events)', d['roofline']['ms_per_launch'])"
done; done; done
