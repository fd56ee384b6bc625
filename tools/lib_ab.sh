#!/bin/bash
# bench line with two builds of the library, alternating:  tools/lib_ab.sh <other .so> [bench args]
other=$1; shift
for lib in "" $other "" $other; do
  FFS_HIP_LIB=$lib python bench.py --no-cpu-baseline --no-streamed --no-cli-e2e --steps 100 --reps 5 "$@" > /tmp/x.json 2>/dev/null
  python -c "
import json; d=json.load(open('/tmp/x.json')); print('${lib:-this build}:', d['value'], d['ms_per_step'], d['steady_ms_per_step'], 'kernel', d['roofline']['ms_per_launch'], d['stage_ms_last_batch'])"
done
