set -e
mkdir -p gpurun_out/r2f
python -m pytest tests -m gpu -q -x > gpurun_out/r2f/gputests.log 2>&1 || true; tail -5 gpurun_out/r2f/gputests.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/r2f/prof --output-format csv -- python3 bench.py --steps 20 --warmup 3 --streams 1 --no-cpu-baseline > gpurun_out/r2f/bench1.log 2>&1
cat gpurun_out/r2f/prof/*/*kernel_stats.csv | cut -d, -f1-4 | head -16
python bench.py --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/r2f/bench.json 2> gpurun_out/r2f/bench.err; cut -c1-300 gpurun_out/r2f/bench.json
