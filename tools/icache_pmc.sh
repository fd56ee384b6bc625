cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 -L 2>/dev/null | grep -i -o "SQC_ICACHE[A-Z_]*\|SQC_INST[A-Z_]*\|SQ_INST_CYCLES[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQC_TC_INST[A-Z_]*" | sort -u > gpurun_out/r05zx_counters.txt
cat gpurun_out/r05zx_counters.txt
for c in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" ; do
  d=gpurun_out/r05zx_icache_alone; rm -rf $d
  rocprofv3 --pmc $c -d $d --output-format csv -- python3 tools/prof_threshold.py --iters 3 --workload eiger16m --algorithm dispersion > $d.log 2>&1 || echo "pass failed"
  d=gpurun_out/r05zx_icache_pipe; rm -rf $d
  rocprofv3 --pmc $c -d $d --output-format csv -- python3 bench.py --steps 6 --warmup 2 --reps 1 --no-cpu-baseline --no-streamed --no-cli-e2e > $d.log 2>&1 || echo "pass failed"
done
python3 tools/summarize_pmc.py gpurun_out/r05zx_icache_alone > gpurun_out/r05zx_icache_alone.json
python3 tools/summarize_pmc.py gpurun_out/r05zx_icache_pipe > gpurun_out/r05zx_icache_pipe.json
python3 - <<PY
import json
for f in ("alone","pipe"):
    d=json.load(open(f"gpurun_out/r05zx_icache_{f}.json"))
    for k,v in d.items():
        if isinstance(v,dict) and ("k_stream" in k or "k_band" in k or "k_frame" in k): print(f, k[:50], v)
PY
