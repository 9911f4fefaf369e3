#!/bin/bash
# Everything profiles/ holds for a round besides tools/profile_round.sh's output (and tools/sensitivity.sh, tools/soak.py): the other configurations' bench lines, the
# BayesW workloads, and the two-rank rehearsal of the bench path on one GPU (correctness of the path, not speed).
R=${1:-r02}
O=gpurun_out/final_$R
rm -rf $O && mkdir -p $O
python3 bench.py --config c2 > $O/${R}_c2_bench.json 2> $O/c2.err; echo c2 done
python3 bench.py --config c3 > $O/${R}_c3_bench.json 2> $O/c3.err; echo c3 done
python3 bench.py --missing 0.01 --no-cpu-baseline > $O/${R}_c4_missing1pct_bench.json 2> $O/c4m.err; echo c4m done
python3 bench.py --config w100k > $O/${R}_bw_w100k_bench.json 2> $O/w100k.err; echo w100k done
python3 bench.py --config c5 > $O/${R}_bw_c5_bench.json 2> $O/c5.err; echo c5 done
HGIBBS_BENCH_BULK=gloo HGIBBS_BENCH_DEVICE=0 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 \
    bench.py --gpus 2 --steps 3 --warmup 2 > $O/${R}_c4_two_ranks_one_gpu_rehearsal.json 2> $O/gpus2.err; echo gpus2 done
bash tools/rehearse_ranks.sh 2 125000 200000 4 > $O/${R}_rehearse_2ranks.txt 2>&1; echo rehearsal done
python3 tools/bsum.py $O/${R}_c2_bench.json $O/${R}_c3_bench.json $O/${R}_c4_missing1pct_bench.json $O/${R}_c4_two_ranks_one_gpu_rehearsal.json
