#!/bin/bash
# rocprofv3 evidence for the threshold stage at the current commit (run on the GPU box):
#   FFS_COMMIT=<short hash> tools/pmc_threshold.sh <tag> [workload] [algorithm]     e.g. r02a eiger16m dispersion
# (the box has no .git: pass the commit in FFS_COMMIT so the summary is stamped with it; SKIP_TRACE=1: counter passes only;
#  FFS_PROFILE_DENSE=1 in the environment profiles the streaming kernel with the dense byte mask written)
# writes gpurun_out/<tag>_pmc*/ (counter passes: FETCH_SIZE, WRITE_SIZE and the SQ counters each in a run
# of their own, no tracing beside them), gpurun_out/<tag>_pmc_threshold_eiger16m_b32.json (summary,
# (2 FETCH_SIZE + WRITE_SIZE) * 1024 per the gfx950 guide) and gpurun_out/<tag>_kernel_stats_*.csv.
set -e
tag=${1:-r02x}
wl=${2:-eiger16m}
alg=${3:-dispersion}
suffix=$wl; [ "$alg" = dispersion ] || suffix=${wl}_extended
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out
DENSE=; [ -n "$FFS_PROFILE_DENSE" ] && DENSE=--dense
for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_WAVES GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_IFETCH SQ_INST_LEVEL_VMEM"; do
  d=$out/${tag}_pmc_$(echo $c | tr ' ' '_')
  rm -rf $d
  rocprofv3 --pmc $c -d $d --output-format csv -- python3 tools/prof_threshold.py --iters 3 --workload $wl --algorithm $alg $DENSE > $d.log 2>&1 || echo "pass $c failed"
done
python3 tools/summarize_pmc.py $out/${tag}_pmc_* > $out/${tag}_pmc_threshold_${suffix}_b32.json
[ -n "$SKIP_TRACE" ] && { cat $out/${tag}_pmc_threshold_${suffix}_b32.json; exit 0; }
rm -rf $out/${tag}_trace1 $out/${tag}_trace4
rocprofv3 --kernel-trace --stats -d $out/${tag}_trace1 --output-format csv -- python3 bench.py --steps 20 --warmup 3 --streams 1 --no-cpu-baseline --no-streamed --no-cli-e2e --workload $wl --algorithm $alg > $out/${tag}_trace1.log 2>&1
cp $out/${tag}_trace1/*/*kernel_stats.csv $out/${tag}_kernel_stats_bench_${suffix}_b32_1stream.csv
rocprofv3 --kernel-trace --stats -d $out/${tag}_trace4 --output-format csv -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-streamed --no-cli-e2e --workload $wl --algorithm $alg > $out/${tag}_trace4.log 2>&1
cp $out/${tag}_trace4/*/*kernel_stats.csv $out/${tag}_kernel_stats_bench_${suffix}_4streams.csv
cat $out/${tag}_pmc_threshold_${suffix}_b32.json
cut -d, -f1-4 $out/${tag}_kernel_stats_bench_${suffix}_b32_1stream.csv | head -12
