mkdir -p gpurun_out/r2j
python -m pytest tests/test_gpu_decode.py tests/test_gpu_fuzz.py -m gpu -q -x > gpurun_out/r2j/dec.log 2>&1; tail -5 gpurun_out/r2j/dec.log
python3 tools/prof_threshold.py --iters 5 --variants 2 --decode 2>&1 | grep -E "decode|round"
python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['streamed_frames_per_s'], d['streamed_compressed'])"
