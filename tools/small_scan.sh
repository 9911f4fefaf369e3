#!/bin/bash
# batch x max_seg scan at small N (fixed cost regime)
for n in 5000 20000 50000; do for ms in 2 4; do for b in 128 256; do
timeout -k 10 200 python bench.py --config c2 --N $n --M 100000 --steps 3 --warmup 4 --no-cpu-baseline --batch $b --max-seg $ms 2>>gpurun_out/s.err | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('N $n max_seg $ms batch $b', '%.0f' % d['value'], '%.1f ms' % d['ms_per_step'], 'launches %.0f' % d['config']['launches_per_iter'], 'kernel_us %.1f' % (d['roofline']['kernel_ms_avg'] * 1e3))
"; done; done; done
