#!/bin/bash
# One gpurun call that produces what profiles/ holds for a round, all on the HEADLINE workload (config 4, M = 1 M):
#   1. the default bench line;
#   2. the kernel trace of the same command (rocprofv3 --kernel-trace --stats), summarised over the timed iterations;
#   3. PMC passes over the same command, counters only, one group per run (FETCH_SIZE and WRITE_SIZE do not fit one pass;
#      SQ and GRBM counters in two more), each summarised over the dispatches of the TIMED iterations (the last
#      launches_per_iter * steps dispatches of the sweep kernel: k_sweep_resident, one per sweep, or k_sweep_batch), and folded into <R>_c4_pmc_traffic.json / <R>_c4_pmc_valu.json,
#      which bench.py quotes with their source.
# usage (on the GPU box, repo root): bash tools/profile_round.sh rNN [steps] [warmup] [tag [bench.py arguments of that workload ...]]
#   tag names the files (default c4: the headline workload), e.g.  ... r04 3 2 c2 --config c2   or   ... r04 3 2 c4_missing1pct --missing 0.01
set -e
R=${1:-r02}
STEPS=${2:-3}
WARM=${3:-2}
TAG=${4:-c4}
shift 4 2>/dev/null || shift $#
EXTRA="$@"
export TMPDIR=/tmp
O=gpurun_out/prof_${R}_$TAG
rm -rf $O && mkdir -p $O
python3 bench.py $EXTRA > $O/${R}_${TAG}_bench.json 2> $O/bench.err
echo "bench done"; cut -c1-400 $O/${R}_${TAG}_bench.json

rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python3 bench.py $EXTRA --steps $STEPS --warmup $WARM --no-cpu-baseline --no-anatomy > $O/${R}_${TAG}_bench_under_rocprof.json 2> $O/kt.err
DB=$(ls $O/kt/*results.db $O/kt/*/*results.db 2>/dev/null | head -1)
L=$(python3 -c "import json; d=json.load(open('$O/${R}_${TAG}_bench_under_rocprof.json')); print(int(round(d['config']['launches_per_iter']*d['steps'])))")
KERN=$(python3 -c "import json; d=json.load(open('$O/${R}_${TAG}_bench_under_rocprof.json')); print(d['roofline']['kernel'])")
python3 tools/rocpd_stats.py $DB $O/${R}_${TAG}_kernel_stats.csv $KERN $L > $O/${R}_${TAG}_kernel_timed_region.txt
cat $O/${R}_${TAG}_kernel_timed_region.txt
rm -rf $O/kt
echo "kernel trace done"

pmc_pass () { # name, counters...
  local NAME=$1; shift
  rocprofv3 --pmc "$@" -d $O/pmc_$NAME -o p -- python3 bench.py $EXTRA --steps $STEPS --warmup $WARM --no-cpu-baseline --no-anatomy > $O/pmc_$NAME.json 2> $O/pmc_$NAME.err
  local DBP=$(ls $O/pmc_$NAME/*results.db $O/pmc_$NAME/*/*results.db 2>/dev/null | head -1)
  local LP=$(python3 -c "import json; d=json.load(open('$O/pmc_$NAME.json')); print(int(round(d['config']['launches_per_iter']*d['steps'])))")
  local KP=$(python3 -c "import json; d=json.load(open('$O/pmc_$NAME.json')); print(d['roofline']['kernel'])")
  python3 tools/rocpd_pmc.py $DBP $KP $LP > $O/${R}_${TAG}_pmc_$NAME.txt
  cat $O/${R}_${TAG}_pmc_$NAME.txt | cut -c1-300
  rm -rf $O/pmc_$NAME
  echo "pmc pass $NAME done"
}
pmc_pass FETCH_SIZE FETCH_SIZE
pmc_pass WRITE_SIZE WRITE_SIZE
pmc_pass SQ1 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE
pmc_pass SQ2 SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE
python3 tools/pmc_fold.py $O $R $TAG
ls -la $O
