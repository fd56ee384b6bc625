#!/bin/bash
# what the sparse launch costs the STEP (four batches in flight) when it is cut short after phase A / L1 / L2 / U / P (experiments build)
for stop in 0 4 3 5 2 1 0; do
  FFS_EXP_CHAIN_STOP=$stop FFS_HIP_LIB=$GRAFT_REPO_ROOT/fast-feedback-service_amd/libffs_hip_exp.so python bench.py --no-cpu-baseline --no-streamed --no-cli-e2e --steps 100 --reps 3 "$@" > /tmp/x.json 2>/dev/null
  python -c "
import json; d=json.load(open('/tmp/x.json')); print('stop $stop: ms/step', d['ms_per_step'], 'steady', d['steady_ms_per_step'], d['stage_ms_last_batch'])"
done
