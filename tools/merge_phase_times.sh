#!/bin/bash
# Duration of the phases of k_frame_merge (device timestamps, experiments build): header + scan | seams | finds + folds | labels | records
for st in 1 4; do
  echo "== $st batch(es) in flight"
  FFS_EXP_CHAIN_TS=1 FFS_HIP_LIB=$GRAFT_REPO_ROOT/fast-feedback-service_amd/libffs_hip_exp.so python bench.py --no-cpu-baseline --no-streamed --no-cli-e2e --steps 100 --reps 3 --streams $st "$@" 2>&1 >/dev/null | grep "ffs exp"
done
