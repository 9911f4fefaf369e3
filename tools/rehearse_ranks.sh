#!/bin/bash
# Rehearsal of the sharded resident engine on ONE GPU (run on the GPU box from the repo root): R processes share device 0, the
# peer mailboxes are IPC-mapped memory of the same card (not xGMI), bulk reductions over gloo.  Prints the bench lines of
#   (a) one rank holding the whole N, (b) one rank holding N / R (what one shard costs alone), (c) R ranks of N / R each.
# usage: tools/rehearse_ranks.sh <R> <N> <M> <steps>
set -uo pipefail
cd "$(dirname "$0")/.."
R=$1; N=$2; M=$3; K=$4
export HSA_ENABLE_IPC_MODE_LEGACY=0
one() { python3 - "$1" <<'PY'
import json, sys
try:
    d = json.loads(sys.argv[1]); r = d["roofline"]
    print("  n_gpus %d  N %d  M %d  engine %s  exchange %s: %.3f M markers/s, %.2f ms/step, %.1f rounds/iter, %.2f us/round" % (
        d["n_gpus"], d["config"]["N"], d["config"]["M"], r.get("engine"), d["config"]["exchange"], d["value"] / 1e6, d["ms_per_step"], r.get("rounds_per_iter", 0), r.get("us_per_round", 0)))
except Exception as e:
    print("  FAILED", repr(e), sys.argv[1][-400:])
PY
}
echo "(a) one rank, N = $N"
one "$(timeout -k 10 300 python3 bench.py --N $N --M $M --steps $K --warmup 2 --no-cpu-baseline --no-anatomy 2>/dev/null | tail -1)"
echo "(b) one rank, N = $((N / R))"
one "$(timeout -k 10 300 python3 bench.py --N $((N / R)) --M $M --steps $K --warmup 2 --no-cpu-baseline --no-anatomy 2>/dev/null | tail -1)"
echo "(c) $R ranks on one GPU, N = $N"
one "$(HGIBBS_BENCH_BULK=gloo HGIBBS_BENCH_DEVICE=0 timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $R --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus $R --N $N --M $M --steps $K --warmup 2 --no-cpu-baseline --no-anatomy 2>gpurun_out/rehearse_err.log | grep '^{' | tail -1)"
