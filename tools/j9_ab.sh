#!/bin/bash
# the 32-bit workload, this build against another, alternating
other=$1
for rep in 1 2 3; do for lib in "$other" ""; do
  FFS_HIP_LIB=$lib python bench.py --no-cpu-baseline --no-streamed --no-cli-e2e --steps 60 --warmup 5 --reps 5 --workload jungfrau9m > /tmp/x.json 2>/tmp/x.err || { echo "FAILED"; tail -3 /tmp/x.err; continue; }
  python -c "
import json; d=json.load(open('/tmp/x.json')); r=d['roofline']; print('$(basename ${lib:-this_build}) jungfrau9m:', d['value'], 'fps | ms/step', d['ms_per_step'], 'steady', d['steady_ms_per_step'], '| kernel (events)', r['ms_per_launch'], '| checked', d.get('results_checked'))"
done; done
