#!/bin/bash
# Quick kernel-trace of a short bench run on the batch engine (run on the GPU box from the repo root): per-launch durations of
# k_sweep_batch into gpurun_out/kt1/stats.csv.
set -euo pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/kt1
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python3 bench.py --steps 2 --warmup 2 --M 300000 --no-cpu-baseline --no-anatomy --opt ahead=0 > $O/b.json 2> $O/kt.err
DB=$(ls $O/kt/*results.db $O/kt/*/*results.db 2>/dev/null | head -1)
L=$(python3 -c "import json; d=json.load(open('$O/b.json')); print(int(round(d['config']['launches_per_iter']*d['steps'])))")
python3 tools/rocpd_stats.py $DB $O/stats.csv k_sweep_batch $L
rm -rf $O/kt
