#!/bin/bash
# A/B of the runtime's number of hardware queues (GPU_MAX_HW_QUEUES, default 4) under the bench's four batches in flight:
# the library's streams (a dense and a sparse one per batch in flight, plus the upload stream) share them.
# usage: tools/hwq_ab.sh [reps]   (on the GPU box; prints one line per run)
reps=${1:-2}
for rep in $(seq $reps); do
  for q in default 8 16 2; do
    for alg in dispersion dispersion_extended; do
      if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
      python bench.py --steps 20 --warmup 5 --algorithm $alg --no-cli-e2e --no-streamed --no-cpu-baseline > /tmp/hwq.json 2>/dev/null || { echo "bench failed ($q $alg)"; exit 1; }
      python -c "
import json; d=json.load(open('/tmp/hwq.json')); print('queues $q $alg:', d['value'], d['ms_per_step'], d['results_checked'], d['roofline'].get('kernel_ms'))"
    done
  done
done
