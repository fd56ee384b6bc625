#!/bin/bash
# Where the streaming kernel's time goes: the experiments build with phases switched off (results are wrong there), then the SQ counters.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=${1:-dec}
FFS_HIP_LIB=$GRAFT_REPO_ROOT/fast-feedback-service_amd/libffs_hip_exp.so python3 tools/prof_threshold.py --iters 10 --rounds 3 --exp 0,1,2,4,32,34 > gpurun_out/${tag}_decompose.txt 2>&1
grep "round" gpurun_out/${tag}_decompose.txt | sort -k3,3n -k2,2n | awk '{print $2,$3,$4,$7,$8}'
SKIP_TRACE=1 bash tools/pmc_threshold.sh $tag > gpurun_out/${tag}_pmc.log 2>&1
python3 - <<PY
import json
d=json.load(open("gpurun_out/${tag}_pmc_threshold_eiger16m_b32.json"))
for k,v in d.items():
    if "k_stream" in k: print(k[:60], json.dumps(v))
PY
