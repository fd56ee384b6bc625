#!/bin/bash
# Bands of a streaming launch and how many of them a wave takes in a row (tuning stream_bands, bands_per_wave), alternating on one box:
#   tools/runs_ab.sh "0,1" "60,2" "90,3" ...      (0 = bands from target_waves: 56 of 78 rows for 32 Eiger frames)
for rep in 1 2 3; do for cfg in "$@"; do
  nb=${cfg%,*}; bpw=${cfg#*,}
  python bench.py --no-cpu-baseline --no-streamed --no-cli-e2e --steps 100 --warmup 5 --reps 5 --tune stream_bands=$nb,bands_per_wave=$bpw > /tmp/x.json 2>/tmp/x.err || { echo "bands $nb x $bpw FAILED"; tail -3 /tmp/x.err; continue; }
  python -c "
import json; d=json.load(open('/tmp/x.json')); r=d['roofline']; print('bands $nb, $bpw a wave:', d['value'], 'fps | ms/step', d['ms_per_step'], 'steady', d['steady_ms_per_step'], 'drain', d['drain_ms'], '| kernel (events)', r['ms_per_launch'], 'alone', r['ms_per_launch_alone'], '| checked', d.get('results_checked'))"
done; done
