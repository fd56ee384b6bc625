for rep in 1 2; do for st in 3 2 4 3; do
python bench.py --no-cpu-baseline --no-streamed --no-cli-e2e --steps 20 --warmup 5 --reps 7 --streams $st > /tmp/x.json 2>/dev/null
python -c "
import json; d=json.load(open('/tmp/x.json')); print('streams $st:', d['value'], 'ms/step', d['ms_per_step'], 'steady', d['steady_ms_per_step'], 'drain', d['drain_ms'], d['repetitions']['ms_per_step'])"
done; done
