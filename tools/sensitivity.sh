#!/bin/bash
# Sensitivity of the headline (config 4) to the share of causal markers and to missing calls: 10 timed iterations each.
# usage (GPU box, repo root): bash tools/sensitivity.sh rNN
R=${1:-r02}
O=gpurun_out/sens_$R
rm -rf $O && mkdir -p $O
for CF in 0.001 0.01 0.05; do
  for MS in 0 0.01; do
    python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --causal-frac $CF --missing $MS > $O/c4_causal${CF}_missing${MS}.json 2> $O/c4_causal${CF}_missing${MS}.err
    echo "causal $CF missing $MS done"
  done
done
python3 tools/bsum.py $O/*.json | tee $O/summary.txt
