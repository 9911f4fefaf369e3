set -e
timeout -k 10 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for cfg in "c2 32 8" "c2 64 8" "c2 128 16" "c4 32 8" "c4 64 8" "c4 64 16" "c4 128 16" "c4 256 32"; do
  set -- $cfg
  echo "== $1 batch=$2 cpg=$3"
  timeout -k 10 200 python bench.py --config $1 --batch $2 --cpg $3 --steps 2 --warmup 2 --no-cpu-baseline | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('markers/s %.0f  ms/step %.1f  launches %.0f  nnz %.0f  kern_us %.1f  frac %.3f' % (d['value'], d['ms_per_step'], d['config']['launches_per_iter'], d['config']['nnz_updates_per_iter'], d['roofline']['kernel_ms_avg']*1e3, d['roofline']['frac']))"
done
