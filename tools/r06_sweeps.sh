#!/bin/bash
# rows in flight (3 / 4) on the standard path; the extended algorithm and the 32-bit workload, this build against another
other=$1
bash tools/tune_sweep.sh rows_ahead 3 4
for rep in 1 2; do for lib in "$other" ""; do
  for wl in "--algorithm dispersion_extended" "--workload jungfrau9m"; do
  FFS_HIP_LIB=$lib python bench.py --no-cpu-baseline --no-streamed --no-cli-e2e --steps 60 --warmup 5 --reps 5 $wl > /tmp/x.json 2>/tmp/x.err || { echo "FAILED $wl"; tail -3 /tmp/x.err; continue; }
  python -c "
import json; d=json.load(open('/tmp/x.json')); r=d['roofline']; print('$(basename ${lib:-this_build}) $wl:', d['value'], 'fps | ms/step', d['ms_per_step'], 'steady', d['steady_ms_per_step'], '| kernel (events)', r['ms_per_launch'], '| checked', d.get('results_checked'))"
  done
done; done
