for rep in 1 2; do for b in 32 48 56 24; do
python bench.py --no-cpu-baseline --no-streamed --no-cli-e2e --steps 40 --warmup 5 --reps 5 --batch $b > /tmp/x.json 2>/dev/null
python -c "
import json; d=json.load(open('/tmp/x.json')); print('batch $b:', d['value'], 'ms/step', d['ms_per_step'], 'per frame us', round(d['ms_per_step']*1000/$b,3), 'steady/frame', round(d['steady_ms_per_step']*1000/$b,3), 'kernel/frame', round(d['roofline']['ms_per_launch']*1000/$b,3))"
done; done
