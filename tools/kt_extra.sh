#!/bin/bash
# Kernel-trace summaries (rocprofv3 --kernel-trace --stats) of the workloads besides config 4: BayesW w100k, config 2, config 3.
R=${1:-r02}
export TMPDIR=/tmp
O=gpurun_out/kt_$R
rm -rf $O && mkdir -p $O
run () { # tag, bench args...
  local TAG=$1; shift
  rocprofv3 --kernel-trace --stats -d $O/$TAG -o kt -- python3 bench.py "$@" --no-cpu-baseline > $O/${R}_${TAG}_bench_under_rocprof.json 2> $O/$TAG.err
  local DB=$(ls $O/$TAG/*results.db $O/$TAG/*/*results.db 2>/dev/null | head -1)
  python3 tools/rocpd_stats.py $DB $O/${R}_${TAG}_kernel_stats.csv > $O/${R}_${TAG}_kernel_stats.txt
  cat $O/${R}_${TAG}_kernel_stats.txt
  rm -rf $O/$TAG
}
run bw_w100k --config w100k --steps 3 --warmup 1
run c2 --config c2 --steps 5 --warmup 2 --no-anatomy
run c3 --config c3 --steps 3 --warmup 2 --no-anatomy
