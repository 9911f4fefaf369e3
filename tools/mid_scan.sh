#!/bin/bash
for n in 100000 200000 300000; do for ms in 2 4; do
timeout -k 10 250 python bench.py --config c2 --N $n --M 300000 --steps 3 --warmup 3 --no-cpu-baseline --batch 256 --max-seg $ms 2>>gpurun_out/s.err | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('N $n max_seg $ms batch 256', '%.0f' % d['value'], '%.1f ms' % d['ms_per_step'], 'launches %.0f' % d['config']['launches_per_iter'], 'kernel_us %.1f' % (d['roofline']['kernel_ms_avg'] * 1e3))
"; done; done
