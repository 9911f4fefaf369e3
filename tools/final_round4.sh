#!/bin/bash
# Round 4: everything profiles/ holds besides tools/profile_round.sh's outputs -- the driver's command on the headline workload, the other
# configurations' bench lines, stage anatomies (build with stage clocks), a multi-GPU shard's round on one GPU, the two-rank rehearsal of the
# bench path, BayesW, the micro-benchmarks behind DESIGN.md's issue-rate statements, the ARS draw on one device lane.
# usage (on the GPU box, repo root): bash tools/final_round4.sh r04
R=${1:-r04}
O=gpurun_out/final_$R
rm -rf $O && mkdir -p $O
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/${R}_c4_bench_steps20_warmup5.json 2> $O/c4.err; echo "c4 (driver's command) done"
python3 bench.py --config c2 > $O/${R}_c2_bench.json 2> $O/c2.err; echo c2 done
python3 bench.py --config c3 > $O/${R}_c3_bench.json 2> $O/c3.err; echo c3 done
python3 bench.py --missing 0.01 --no-cpu-baseline > $O/${R}_c4_missing1pct_bench.json 2> $O/c4m.err; echo c4m done
python3 bench.py --no-cpu-baseline --opt walker=1 --opt early_advance=0 > $O/${R}_c4_first_walker_bench.json 2> $O/c4w1.err; echo "c4, first walker done"
python3 bench.py --no-cpu-baseline --opt refill=1 > $O/${R}_c4_first_form_bench.json 2> $O/c4r1.err; echo "c4, first form of the streaming workgroups done"
python3 bench.py --no-cpu-baseline --opt announce=0 > $O/${R}_c4_no_announce_bench.json 2> $O/c4na.err; echo "c4, no announcements done"
python3 bench.py --no-cpu-baseline --missing 0.01 --opt refill=1 > $O/${R}_c4_missing1pct_first_form_bench.json 2> $O/c4mr1.err; echo "c4 missing, first form done"
python3 bench.py --no-cpu-baseline --opt pivots=1 > $O/${R}_c4_pivots_bench.json 2> $O/c4p.err; echo "c4, pivots done"
python3 bench.py --config c2 --no-cpu-baseline --opt refill=1 > $O/${R}_c2_first_form_bench.json 2> $O/c2r1.err; echo "c2, first form done"
python3 bench.py --config w100k > $O/${R}_bw_w100k_bench.json 2> $O/w100k.err; echo w100k done
python3 bench.py --config c5 > $O/${R}_bw_c5_bench.json 2> $O/c5.err; echo c5 done
python3 tools/res_anatomy.py 500000 1000000 4 > $O/${R}_c4_anatomy.txt 2>&1; echo "c4 anatomy done"
python3 tools/res_anatomy.py 50000 100000 6 > $O/${R}_c2_anatomy.txt 2>&1; echo "c2 anatomy done"
python3 tools/res_anatomy.py 200000 500000 4 > $O/${R}_c3_shape_anatomy.txt 2>&1; echo "c3-shape anatomy done"
python3 tools/res_anatomy.py 500000 1000000 4 missing=0.01 > $O/${R}_c4_missing1pct_anatomy.txt 2>&1; echo "c4 missing anatomy done"
python3 tools/res_anatomy.py 62500 200000 5 > $O/${R}_shard_62500_anatomy.txt 2>&1; echo "shard anatomy done"
HGIBBS_BENCH_BULK=gloo HGIBBS_BENCH_DEVICE=0 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 \
    bench.py --gpus 2 --steps 3 --warmup 2 > $O/${R}_c4_two_ranks_one_gpu_rehearsal.json 2> $O/gpus2.err; echo gpus2 done
bash tools/rehearse_ranks.sh 2 125000 200000 4 > $O/${R}_rehearse_2ranks.txt 2>&1; echo rehearsal done
for u in valu_rate refill_block refill_col mfma_limb_dot; do [ -x tools/ubench/$u ] || hipcc --offload-arch=gfx950 -O3 -o tools/ubench/$u tools/ubench/$u.hip > /dev/null 2>&1; timeout -k 5 120 ./tools/ubench/$u > $O/${R}_ubench_$u.txt 2>&1; done; echo ubench done
timeout -k 5 120 python3 tools/ars_device_probe.py 2000 > $O/${R}_bw_ars_on_one_device_lane.txt 2>&1; echo ars done
python3 tools/bsum.py $O/${R}_c4_bench_steps20_warmup5.json $O/${R}_c2_bench.json $O/${R}_c3_bench.json $O/${R}_c4_missing1pct_bench.json $O/${R}_c4_first_walker_bench.json $O/${R}_c4_first_form_bench.json $O/${R}_c4_no_announce_bench.json $O/${R}_c4_missing1pct_first_form_bench.json $O/${R}_c4_pivots_bench.json $O/${R}_c2_first_form_bench.json
