#!/bin/bash
# Per-launch anatomy of the sweep at the shard sizes of a multi-GPU run of config 4 (N = 500 K over 8 / 4 / 2 GPUs), on ONE
# GPU: (a) one process per size, no exchange; (b) two processes sharing the GPU with the in-launch peer-mailbox exchange
# active (IPC-mapped memory on the same device: the exchange's code path, not xGMI's latency; the two processes also share
# the compute units, so their launches are slower than a rank with a GPU of its own).  usage: tools/shard_anatomy.sh rNN
R=${1:-r02}
O=gpurun_out/shards_$R
rm -rf $O && mkdir -p $O
for NG in 62500 125000 250000; do
  python3 bench.py --N $NG --M 1000000 --steps 3 --warmup 2 --no-cpu-baseline > $O/single_$NG.json 2> $O/single_$NG.err
  echo "single $NG done"
done
for NG in 62500 125000 250000; do
  HGIBBS_BENCH_BULK=gloo HGIBBS_BENCH_DEVICE=0 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
      bench.py --gpus 2 --N $((2 * NG)) --M 400000 --steps 3 --warmup 2 --exchange p2p --no-cpu-baseline > $O/two_ranks_$NG.json 2> $O/two_ranks_$NG.err
  echo "two ranks $NG done"
done
python3 tools/bsum.py $O/*.json
