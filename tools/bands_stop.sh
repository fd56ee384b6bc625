#!/bin/bash
# What the band launches cost the step when they are cut short (experiments build: results are wrong): 10 = both kernels return at once,
# 12 = the band wave after its entries are in LDS, 13 = after the placement, 11 = whole band wave but no merge, 0 = everything.
for rep in 1 2; do for stop in 10 12 13 11 0; do
  FFS_EXP_CHAIN_STOP=$stop FFS_HIP_LIB=$GRAFT_REPO_ROOT/fast-feedback-service_amd/libffs_hip_exp.so python bench.py --no-cpu-baseline --no-streamed --no-cli-e2e --steps 60 --warmup 5 --reps 5 "$@" > /tmp/x.json 2>/dev/null
  python -c "
import json; d=json.load(open('/tmp/x.json')); print('stop $stop: fps', d['value'], 'ms/step', d['ms_per_step'], 'steady', d['steady_ms_per_step'], 'kernel alone', d['roofline']['ms_per_launch'], d['stage_ms_last_batch'])"
done; done
