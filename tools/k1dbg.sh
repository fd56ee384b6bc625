set -e
mkdir -p gpurun_out/r2c
rm -f gpurun_out/r2c/dbg5.log
for cfg in "0 2" "1 2" "2 2" "4 2" "0 3" "8 2"; do
  set -- $cfg
  echo "== FFS_K1_DEBUG=$1 AHEAD=$2" >> gpurun_out/r2c/dbg5.log
  FFS_K1_DEBUG=$1 FFS_K1_AHEAD=$2 python3 tools/prof_threshold.py --iters 10 --variants 2 2>&1 | grep round >> gpurun_out/r2c/dbg5.log
done
cat gpurun_out/r2c/dbg5.log
python -m pytest tests -m gpu -q -x > gpurun_out/r2c/gputests.log 2>&1; tail -5 gpurun_out/r2c/gputests.log
python bench.py --steps 50 --warmup 5 > gpurun_out/r2c/bench.json 2> gpurun_out/r2c/bench.err; cat gpurun_out/r2c/bench.json
