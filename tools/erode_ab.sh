#!/bin/bash
# extended dispersion: the erosion kernels side by side (tuning "ext_erode": 0 = a lane per word column, 1 / 2 = strips of 32 / 16 rows)
# through bench.py (every batch checked) and through a kernel trace.   usage (GPU box): tools/erode_ab.sh [reps]
reps=${1:-2}
root=$(pwd)
for rep in $(seq $reps); do
  for e in 0 2 "2,ext_e_sparse=1"; do
    python bench.py --no-cpu-baseline --no-streamed --no-cli-e2e --steps 20 --warmup 5 --algorithm dispersion_extended --tune ext_erode=$e > /tmp/x.json 2>/tmp/x.err || { echo "bench failed (ext_erode=$e)"; tail -5 /tmp/x.err; exit 1; }
    python -c "
import json; d=json.load(open('/tmp/x.json')); r=d['roofline']
print('ext_erode=$e: value', d['value'], 'ms/step', d['ms_per_step'], 'checked', d['results_checked'], 'first', r['ms_per_launch'], 'rest', r.get('exact_kernel_ms_per_launch'))"
  done
done
cd /tmp && export TMPDIR=/tmp
for e in 0 2 "2,ext_e_sparse=1"; do
  rocprofv3 --kernel-trace --stats -d /tmp/erode_${e%%,*}${e##*=} --output-format csv -- python3 $root/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-streamed --no-cli-e2e --algorithm dispersion_extended --tune ext_erode=$e > /tmp/erode_${e%%,*}${e##*=}.log 2>&1 || { echo "trace failed"; tail -5 /tmp/erode_${e%%,*}${e##*=}.log; exit 1; }
  echo "== ext_erode=$e (4 batches in flight): kernel, calls, average us"
  python3 - /tmp/erode_${e%%,*}${e##*=} <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Name"]
        if any(k in n for k in ("k_ext_erode", "k_ext_final", "k_stream_u16", "k_frame_chain", "fillBuffer")):
            print("   %-70s %5s %9.1f" % (n[:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
