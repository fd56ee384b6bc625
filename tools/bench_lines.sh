#!/bin/bash
# The four bench lines at the driver's arguments (--steps 20 --warmup 5), one JSON line each:  gpurun -- 'bash tools/bench_lines.sh <tag>'
tag=${1:-lines}
out=gpurun_out/${tag}_bench_lines_steps20.jsonl; : > $out
python bench.py --steps 20 --warmup 5 >> $out 2> gpurun_out/${tag}_bench.err || echo "eiger16m failed"
python bench.py --steps 20 --warmup 5 --algorithm dispersion_extended --no-cli-e2e >> $out 2>> gpurun_out/${tag}_bench.err || echo "extended failed"
python bench.py --steps 20 --warmup 5 --workload jungfrau9m --no-cli-e2e >> $out 2>> gpurun_out/${tag}_bench.err || echo "jungfrau failed"
python bench.py --steps 20 --warmup 5 --workload sweep16m --no-cli-e2e >> $out 2>> gpurun_out/${tag}_bench.err || echo "sweep failed"
python - <<PY
import json
for ln in open("$out"):
    d=json.loads(ln); r=d.get("roofline",{})
    print(d["metric"][:60], "|", d["value"], "fps | ms/step", d["ms_per_step"], "steady", d.get("steady_ms_per_step"), "| kernel", r.get("ms_per_launch"), "frac", r.get("frac"), "phys/read", r.get("frac_of_measured_read"), "| checked", d.get("results_checked"))
    e=d.get("cli_e2e")
    if e: print("   cli_e2e:", {k:(v.get("frames_per_s") if isinstance(v,dict) else v) for k,v in e.items() if k in ("gpu_decode","long_run","first_pass_over_fresh_files","long_run_first_pass","cpu_decode","two_contexts_one_gpu","error","long_run_shrunk")})
PY
