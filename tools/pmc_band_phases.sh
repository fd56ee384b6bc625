#!/bin/bash
# Instruction counts of the band wave by phase: the experiments build cut short at 12 (entries in LDS), 13 (decisions, counts, placement)
# and 11 (the whole band wave; the merge returns at once), counters of k_band_cc from the bench's short run.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out
export FFS_HIP_LIB=$GRAFT_REPO_ROOT/fast-feedback-service_amd/libffs_hip_exp.so
for stop in 12 13 11; do
  for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"; do
    d=$out/r06a_band_stop${stop}_$(echo $c | cut -d' ' -f1)
    rm -rf $d
    FFS_EXP_CHAIN_STOP=$stop rocprofv3 --pmc $c -d $d --output-format csv -- python3 bench.py --steps 4 --warmup 1 --reps 1 --no-cpu-baseline --no-streamed --no-cli-e2e > $d.log 2>&1
  done
  python3 tools/summarize_pmc.py $out/r06a_band_stop${stop}_* > $out/r06a_band_stop${stop}.json 2>/dev/null
  python3 - <<PY
import json
d=json.load(open("$out/r06a_band_stop${stop}.json"))
for k,v in d.items():
    if isinstance(v,dict) and "k_band" in k: print("stop $stop", {a:int(b) for a,b in v.items()})
PY
done
