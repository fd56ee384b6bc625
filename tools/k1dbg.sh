mkdir -p gpurun_out/r2i
python -m pytest tests -m gpu -q -x -k "stack3d or sweep or 3d or cli" > gpurun_out/r2i/t3d.log 2>&1; tail -15 gpurun_out/r2i/t3d.log
python bench.py --workload sweep16m --steps 30 --warmup 2 > gpurun_out/r2i/sweep.json 2> gpurun_out/r2i/sweep.err; cat gpurun_out/r2i/sweep.json; tail -3 gpurun_out/r2i/sweep.err
