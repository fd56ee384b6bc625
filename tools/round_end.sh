#!/bin/bash
# The round's last record in one call: full GPU suite, counters + kernel statistics, the four bench lines, and this build against another.
#   FFS_COMMIT=<hash> gpurun --timeout 1200 -- 'bash tools/round_end.sh <tag> [other .so]'
tag=${1:-rend}; other=$2
timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/${tag}_pytest_gpu.log 2>&1; tail -2 gpurun_out/${tag}_pytest_gpu.log
bash tools/pmc_threshold.sh ${tag} eiger16m dispersion > gpurun_out/${tag}_pmc.log 2>&1; tail -4 gpurun_out/${tag}_pmc.log | cut -c1-200
bash tools/bench_lines.sh ${tag}
[ -n "$other" ] && bash tools/lib_multi_ab.sh $other - $other -
