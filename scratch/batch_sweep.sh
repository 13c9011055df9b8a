#!/bin/bash
# episodes per step of the eval headline: value / ms per step / roofline fraction per batch size
for b in 20 25 30 40 50; do
  python bench.py --batch $b --steps 16 --warmup 4 --cpu-episodes 0 --no-single --no-e2e --no-sides 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['config'].get('episodes_per_step'), d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('step_effective_tflops'))"
done
