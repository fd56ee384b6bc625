#!/bin/bash
# bench line under several tuning sets, alternating (same box, same build):  tools/tune_ab.sh "<set>;<set>;..." [bench args]
# a set is 'key=value,key=value' or '-' for the defaults
IFS=';' read -ra sets <<< "$1"; shift
for round in 1 2; do
  for t in "${sets[@]}"; do
    arg=(); [ "$t" != "-" ] && arg=(--tune "$t")
    python bench.py --no-cpu-baseline --no-streamed --no-cli-e2e --steps 100 --reps 5 "${arg[@]}" "$@" > /tmp/x.json 2>/tmp/x.err || { echo "$t: FAILED"; tail -3 /tmp/x.err; continue; }
    python -c "
import json; d=json.load(open('/tmp/x.json')); print('$t:', d['value'], 'ms/step', d['ms_per_step'], 'steady', d['steady_ms_per_step'], 'kernel', d['roofline']['ms_per_launch'], 'checked', d.get('results_checked'), d['stage_ms_last_batch'])"
  done
done
