#!/bin/bash
# One gpurun call that produces what profiles/ holds for a round: the default bench line, the kernel trace of the
# same command (rocprofv3 --kernel-trace --stats) with the timed-region average of the sweep kernel, and the two PMC
# passes (FETCH_SIZE, WRITE_SIZE; separate runs, counters only) behind roofline.traffic.
# usage (on the GPU box, repo root): bash tools/profile_round.sh rNN
set -e
R=${1:-r01}
export TMPDIR=/tmp
O=gpurun_out/prof_$R
rm -rf $O && mkdir -p $O
python3 bench.py > $O/${R}_c4_bench.json 2> $O/bench.err
echo "bench done"; cat $O/${R}_c4_bench.json | cut -c1-300
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline > $O/${R}_c4_bench_under_rocprof.json 2> $O/kt.err
DB=$(ls $O/kt/*results.db $O/kt/*/*results.db 2>/dev/null | head -1)
L=$(python3 -c "import json; d=json.load(open('$O/${R}_c4_bench_under_rocprof.json')); print(int(round(d['config']['launches_per_iter']*d['steps'])))")
python3 tools/rocpd_stats.py $DB $O/${R}_c4_kernel_stats.csv k_sweep_batch $L > $O/${R}_c4_kernel_timed_region.txt
cat $O/${R}_c4_kernel_timed_region.txt
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C -d $O/pmc_$C -o p -- python3 bench.py --M 60000 --steps 1 --warmup 2 --no-cpu-baseline > $O/pmc_$C.json 2> $O/pmc_$C.err
  DBP=$(ls $O/pmc_$C/*results.db $O/pmc_$C/*/*results.db 2>/dev/null | head -1)
  python3 tools/rocpd_pmc.py $DBP $C k_sweep_batch 1.0 > $O/${R}_c4_pmc_$C.txt
  cat $O/${R}_c4_pmc_$C.txt
done
# the databases are large: keep the summaries only
rm -rf $O/kt $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
ls -la $O
