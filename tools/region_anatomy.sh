#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=${1:-anat}
rm -rf gpurun_out/${tag}_prof
rocprofv3 --kernel-trace -d gpurun_out/${tag}_prof --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-streamed --no-cli-e2e > gpurun_out/${tag}_bench_under_profiler.json 2> gpurun_out/${tag}_prof.err
python3 tools/region_anatomy.py gpurun_out/${tag}_prof 20 > gpurun_out/${tag}_region_anatomy.txt
head -14 gpurun_out/${tag}_region_anatomy.txt | cut -c1-330; tail -22 gpurun_out/${tag}_region_anatomy.txt
