#!/bin/bash
# After a change to the threshold kernels: the parity tests that exercise them, then the roofline leg of the bench.
#   gpurun -- 'bash tools/kernel_check.sh <tag> [bench args]'
tag=${1:-kcheck}; shift
out=gpurun_out/$tag; mkdir -p $out
python -m pytest tests -m gpu -q -x -k "golden or parity or fuzz or edge or extended or numerics or fullsize" > $out/tests.log 2>&1; rc=$?; tail -3 $out/tests.log
[ $rc -ne 0 ] && exit $rc
python bench.py --no-cpu-baseline --no-streamed --no-cli-e2e --steps 100 "$@" > $out/bench.json 2> $out/bench.err; rc=$?
python - <<PY
import json
d=json.load(open("$out/bench.json")); r=d["roofline"]
print("value", d["value"], "ms/step", d["ms_per_step"], "steady", d["steady_ms_per_step"], "| kernel ms", r["ms_per_launch"], "rest", r["exact_kernel_ms_per_launch"], "dense", (r.get("with_dense_mask") or {}).get("ms_per_launch"), "| read ceiling", r["measured_peak"]["read_only_GBps"])
PY
exit $rc
