mkdir -p gpurun_out/r2h
python -m pytest tests -m gpu -q -x > gpurun_out/r2h/gputests.log 2>&1; tail -8 gpurun_out/r2h/gputests.log
python bench.py > gpurun_out/r2h/bench.json 2> gpurun_out/r2h/bench.err; cat gpurun_out/r2h/bench.json; tail -3 gpurun_out/r2h/bench.err
