#!/bin/bash
# the one-launch sparse stage stopped after phase A / E (LOG: L1) / L2 / U / P (experiments build, FFS_EXP_CHAIN_STOP): time alone, one batch in flight
for t in strong_log=0 strong_log=1; do
  for stop in 1 2 5 3 4 0; do
    FFS_EXP_CHAIN_STOP=$stop FFS_HIP_LIB=$GRAFT_REPO_ROOT/fast-feedback-service_amd/libffs_hip_exp.so python bench.py --no-cpu-baseline --no-streamed --no-cli-e2e --steps 30 --reps 1 --streams 1 --tune $t "$@" > /tmp/x.json 2>/dev/null
    python -c "
import json; d=json.load(open('/tmp/x.json')); print('$t stop $stop: ccl', d['stage_ms_last_batch']['ccl'], 'threshold', d['stage_ms_last_batch']['threshold'])"
  done
done
