#!/bin/bash
# Counters of EVERY kernel of the hot path's pipeline (the band and merge launches beside the streaming kernel), from the bench's own
# short run; the profiler serialises the kernels, so durations here are "alone".  FFS_COMMIT=<hash> tools/pmc_pipeline.sh <tag> [bench args]
tag=${1:-r05zu}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_WAVES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_SCA"; do
  d=$out/${tag}_pipe_$(echo $c | tr ' ' '_' | cut -c1-60)
  rm -rf $d
  rocprofv3 --pmc $c -d $d --output-format csv -- python3 bench.py --steps 6 --warmup 2 --reps 1 --no-cpu-baseline --no-streamed --no-cli-e2e "$@" > $d.log 2>&1 || echo "pass $c failed"
done
python3 tools/summarize_pmc.py $out/${tag}_pipe_* > $out/${tag}_pmc_pipeline.json
python3 - <<PY
import json
d=json.load(open("$out/${tag}_pmc_pipeline.json"))
for k,v in d.items():
    if isinstance(v,dict) and "SQ_INSTS_VALU" in v and ("k_band" in k or "k_frame" in k or "k_stream" in k):
        print(k[:60], {a:int(b) for a,b in v.items()})
PY
