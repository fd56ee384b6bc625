#!/bin/bash
# The round's PMC summaries (threshold stage): standard, dense byte mask, extended, Jungfrau -- tools/pmc_threshold.sh four times.
#   FFS_COMMIT=<hash> gpurun -- 'bash tools/run_logged.sh <tag> bash tools/pmc_all.sh <tag>'
tag=${1:-r05p}
SKIP_TRACE=1 bash tools/pmc_threshold.sh ${tag} eiger16m dispersion > gpurun_out/${tag}_std.log 2>&1; cp gpurun_out/${tag}_pmc_threshold_eiger16m_b32.json gpurun_out/${tag}_keep_std.json
FFS_PROFILE_DENSE=1 SKIP_TRACE=1 bash tools/pmc_threshold.sh ${tag}d eiger16m dispersion > gpurun_out/${tag}_dense.log 2>&1; cp gpurun_out/${tag}d_pmc_threshold_eiger16m_b32.json gpurun_out/${tag}_pmc_threshold_eiger16m_dense_b32.json
SKIP_TRACE=1 bash tools/pmc_threshold.sh ${tag}x eiger16m dispersion_extended > gpurun_out/${tag}_ext.log 2>&1; cp gpurun_out/${tag}x_pmc_threshold_eiger16m_extended_b32.json gpurun_out/${tag}_pmc_threshold_eiger16m_extended_b32.json
SKIP_TRACE=1 bash tools/pmc_threshold.sh ${tag}j jungfrau9m dispersion > gpurun_out/${tag}_j9.log 2>&1; cp gpurun_out/${tag}j_pmc_threshold_jungfrau9m_b32.json gpurun_out/${tag}_pmc_threshold_jungfrau9m_b32.json
cp gpurun_out/${tag}_keep_std.json gpurun_out/${tag}_pmc_threshold_eiger16m_b32.json
ls -la gpurun_out/${tag}_pmc_threshold_*.json
python3 - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/${tag}_pmc_threshold_*.json")):
    d=json.load(open(f))
    for k,v in d.items():
        if isinstance(v,dict) and "hbm_bytes_per_launch" in v: print(f.split("/")[-1], k[:50], int(v["hbm_bytes_per_launch"]), v.get("valu_busy_frac"), v.get("SQ_INSTS_VALU"))
PY
