#!/bin/bash
# scan of the number of chained segments per launch at one bench config
cfg=${1:-c4}
for ms in 2 3 4; do
  timeout -k 10 280 python bench.py --config $cfg --steps 3 --warmup 2 --no-cpu-baseline --max-seg $ms 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('max_seg $ms', d['config']['workload'][:14], 'markers/s %.0f' % d['value'], 'ms/iter %.1f' % d['ms_per_step'], 'launches %.0f' % d['config']['launches_per_iter'], 'kernel_us %.1f' % (d['roofline']['kernel_ms_avg'] * 1e3))
"
done
