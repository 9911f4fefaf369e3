"""Multi-rank paths on ONE GPU: two processes share device 0, individuals are
sharded between them, bulk reductions go through gloo (hgibbs_comm_init_external)
and the per-batch (s1,s2) rows through the in-launch peer-mailbox exchange
(hgibbs_p2p_export/import; IPC-mapped memory, the same code path that runs over
xGMI between GPUs).  Result must equal the single-rank chain: components exact,
beta to 1e-9; both replicas bit-identical."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _shard(N, world, rank):
    per = ((N + world - 1) // world + 3) // 4 * 4
    return min(N, rank * per), min(N, (rank + 1) * per)


def _worker(rank, world, port, bed, y, N, iters, opts, q):
    import torch
    import torch.distributed as dist
    from hydra_amd import capi
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = capi.Device(0)

        def allreduce(arr):
            if arr.dtype == np.uint64:
                t = torch.from_numpy(arr.view(np.int64))
            else:
                t = torch.from_numpy(arr)
            dist.all_reduce(t)

        dev.comm_init_external(world, rank, allreduce)
        handles = [None] * world
        dist.all_gather_object(handles, dev.p2p_export())
        dev.p2p_import(handles)
        lo, hi = _shard(N, world, rank)
        dev.load_bed(bed, N, row_begin=lo, row_end=hi, n_global=N)
        for k, v in opts.items():
            dev.set_option(k, v)
        ch = capi.Chain(dev, y, seed=1222)
        engines = set()
        for _ in range(iters):
            ch.iterate()
            engines.add(dev.sweep_stats()["engine"])
        beta, comp, acum = dev.get_beta()
        st = ch.state()
        q.put((rank, beta, comp, st["sigmaE"], st["sigmaG"], dev.get_residual(), ch.last_nnz(), sorted(engines), acum, st["cass"]))
    except Exception as e:  # surface the error text in the parent
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("batch,M,N,world,miss", [(16, 300, 9000, 2, 0.01), (64, 300, 9000, 2, 0.01), (256, 700, 33000, 2, 0.0),
                                                  (256, 500, 21000, 3, 0.0)])
def test_two_ranks_one_gpu_p2p_exchange(batch, M, N, world, miss):
    """The larger cases run several slices per column group, the four-segment build and carried dots through the
    sharded path (the carry term is one of the exchanged rows); three ranks give ragged shards."""
    import torch.multiprocessing as mp
    from hydra_amd import capi, synth
    iters = 3
    geno = synth.make_genotypes(M, N, seed=61, missing_rate=miss)
    y, _ = synth.make_phenotype(geno, seed=62, causal_frac=0.05)
    bed = synth.pack_bed_columns(geno)

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, bed, y, N, iters, {"batch": batch}, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for r in res:
        assert len(r) == 10, "rank %s failed: %s" % (r[0], r[1])
        assert r[7] == [1]
    res.sort(key=lambda r: r[0])

    dev = capi.Device(0)
    dev.load_bed(bed, N)
    dev.set_option("batch", batch)
    ch = capi.Chain(dev, y, seed=1222)
    for _ in range(iters):
        ch.iterate()
    beta, comp, _ = dev.get_beta()
    st = ch.state()
    # replicas identical
    for r in res[1:]:
        assert np.array_equal(res[0][1], r[1]) and np.array_equal(res[0][2], r[2]) and res[0][3] == r[3]
    # equal to the single-rank chain (different summation tree: tolerance on beta, exact components)
    assert np.array_equal(res[0][2], comp)
    assert np.all(np.abs(res[0][1] - beta) <= 1e-9 * np.maximum(1.0, np.abs(beta)))
    assert abs(res[0][3] - st["sigmaE"]) <= 1e-9 * st["sigmaE"]
    eps = np.concatenate([r[5] for r in res])
    assert np.allclose(eps, dev.get_residual(), rtol=0, atol=1e-9)
    assert res[0][6] == ch.last_nnz()


@pytest.mark.parametrize("M,N,world,opts,iters,miss", [(400, 9000, 2, {}, 3, 0.0), (500, 21000, 3, {"window": 64}, 3, 0.0), (300, 30000, 2, {"res_cus": 9}, 3, 0.0),
                                                       (2500, 70000, 2, {}, 8, 0.0), (400, 9000, 2, {}, 4, 0.02), (500, 21000, 3, {"window": 64}, 3, 0.01),
                                                       (300, 30000, 2, {"res_cus": 9}, 3, 0.05)])
def test_ranks_one_gpu_resident_engine(oracle, M, N, world, opts, iters, miss):
    """The resident engine sharded over ranks (here: processes that share device 0; the mailboxes are IPC-mapped memory as between
    GPUs): every rank runs one resident kernel on its shard, the walkers are replicas that add the peers' integer Gram sums and
    fixed-point raw dots from their mailboxes.  Replicas bit-identical; equal to the CPU oracle's chain on the whole data
    (components, cass exact; beta, Acum, residual to 1e-9).  Ragged shards (three ranks), a small window, two tiles per workgroup; a longer chain on 35 workgroups per rank (20 000 messages exchanged)."""
    import torch.multiprocessing as mp
    import orc
    from hydra_amd import synth
    geno = synth.make_genotypes(M, N, seed=71, missing_rate=miss)  # (missing calls: the build with s2 per column and four-term Gram sums, both exchanged)
    y, _ = synth.make_phenotype(geno, seed=72, causal_frac=0.05)
    bed = synth.pack_bed_columns(geno)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, bed, y, N, iters, dict(opts, engine=2), q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for r in res:
        assert len(r) == 10, "rank %s failed: %s" % (r[0], r[1])
        assert r[7] == [2]
    res.sort(key=lambda r: r[0])
    ref = orc.Chain(oracle, bed, N, y, seed=1222)
    for _ in range(iters):
        ref.iterate()
    for r in res[1:]:
        assert np.array_equal(res[0][1], r[1]) and np.array_equal(res[0][2], r[2]) and res[0][3] == r[3] and np.array_equal(res[0][8], r[8])
    tol = lambda a, b: np.all(np.abs(np.asarray(a) - np.asarray(b)) <= 1e-9 * np.maximum(1.0, np.abs(np.asarray(b))))
    assert np.array_equal(res[0][2], ref.arr("components")) and np.array_equal(np.asarray(res[0][9]).ravel(), ref.arr("cass"))
    assert tol(res[0][1], ref.arr("beta")) and tol(res[0][8], ref.arr("acum")) and tol(res[0][3], ref.sigmaE)
    assert tol(np.concatenate([r[5] for r in res]), ref.arr("eps"))
    assert res[0][6] == oracle.orc_chain_last_nnz(ref.h)


# ---- BayesW sharded the same way: per-batch row sums and the density sums add over the ranks ----------
def test_two_ranks_whose_grids_cannot_share_the_device_finish_on_the_batch_engine(oracle):
    """Two ranks on ONE device, each with a shard of 131 tiles: 2 x 132 workgroups of 153 KB LDS cannot be resident together on 256
    compute units (the device runs the two kernels one after the other, each grid resident at once -- so a rank's own rendezvous says
    nothing).  With default options the probe launch's handshake between the walkers finds out before any sweep has started, the
    ranks agree, and every sweep runs on the batch engine: no time-out, no error, the oracle's chain."""
    import time
    import torch.multiprocessing as mp
    import orc
    from hydra_amd import synth
    M, N, iters, world = 96, 2 * 131 * 1024, 2, 2
    geno = synth.make_genotypes(M, N, seed=81, missing_rate=0.0)
    y, _ = synth.make_phenotype(geno, seed=82, causal_frac=0.05)
    bed = synth.pack_bed_columns(geno)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, bed, y, N, iters, {}, q)) for r in range(world)]
    t0 = time.time()
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for r in res:
        assert len(r) == 10, "rank %s failed: %s" % (r[0], r[1])
        assert r[7] == [1], "engines used: %s" % (r[7],)
    res.sort(key=lambda r: r[0])
    ref = orc.Chain(oracle, bed, N, y, seed=1222)
    for _ in range(iters):
        ref.iterate()
    tol = lambda a, b: np.all(np.abs(np.asarray(a) - np.asarray(b)) <= 1e-9 * np.maximum(1.0, np.abs(np.asarray(b))))
    assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])
    assert np.array_equal(res[0][2], ref.arr("components")) and tol(res[0][1], ref.arr("beta")) and tol(res[0][3], ref.sigmaE)
    assert tol(np.concatenate([r[5] for r in res]), ref.arr("eps"))


@pytest.mark.parametrize("case", range(int(os.environ.get("HG_RANDOM_RANK_CASES", "3"))))
def test_random_sharded_configurations_match_the_oracle(oracle, case):
    """Seeded random sharded runs of the resident engine (2 to 4 processes sharing device 0, ragged shards, windows, compute units per
    rank, missing calls) against the oracle on the whole data; HG_RANDOM_RANK_CASES lengthens the campaign."""
    import torch.multiprocessing as mp
    import orc
    from hydra_amd import synth
    rng = np.random.default_rng(9100 + case)
    world = int(rng.choice([2, 2, 3, 4]))
    N = int(rng.choice([9000, 21000, 30011, 52000]))
    M = int(rng.integers(120, 500))
    miss = float(rng.choice([0.0, 0.0, 0.02]))
    opts = {"engine": 2, "window": int(rng.choice([32, 64, 128, 256]))}
    if rng.random() < 0.4:
        opts["res_cus"] = int(rng.choice([5, 9, 14]))
    if rng.random() < 0.3:
        opts["refill"] = 1
    geno = synth.make_genotypes(M, N, seed=200 + case, missing_rate=miss)
    y, _ = synth.make_phenotype(geno, seed=300 + case, causal_frac=float(rng.choice([0.02, 0.1])))
    bed = synth.pack_bed_columns(geno)
    iters = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, bed, y, N, iters, opts, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    what = "case %d: world %d N %d M %d miss %g %r" % (case, world, N, M, miss, opts)
    if all(len(r) == 2 and "does not apply" in r[1] for r in res):  # (a shard too large for the compute units drawn: every rank refuses alike)
        pytest.skip("resident engine refused: " + what)
    for r in res:
        assert len(r) == 10, "rank %s failed: %s (%s)" % (r[0], r[1], what)
    res.sort(key=lambda r: r[0])
    if res[0][7] != [2]:  # (a shard that does not fit the compute units asked for: the ranks agreed on the batch engine -- not this test's subject)
        pytest.skip("resident engine refused: " + what)
    ref = orc.Chain(oracle, bed, N, y, seed=1222)
    for _ in range(iters):
        ref.iterate()
    tol = lambda a, b: np.all(np.abs(np.asarray(a) - np.asarray(b)) <= 1e-9 * np.maximum(1.0, np.abs(np.asarray(b))))
    for r in res[1:]:
        assert np.array_equal(res[0][1], r[1]) and np.array_equal(res[0][2], r[2]), what
    assert np.array_equal(res[0][2], ref.arr("components")), what
    assert tol(res[0][1], ref.arr("beta")) and tol(res[0][3], ref.sigmaE), what
    assert tol(np.concatenate([r[5] for r in res]), ref.arr("eps")), what


def _worker_bw(rank, world, port, bed, y, fail, X, N, iters, q):
    import torch
    import torch.distributed as dist
    from hydra_amd import capi
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = capi.Device(0)

        def allreduce(arr):
            t = torch.from_numpy(arr.view(np.int64) if arr.dtype == np.uint64 else arr)
            dist.all_reduce(t)

        dev.comm_init_external(world, rank, allreduce)
        lo, hi = _shard(N, world, rank)
        dev.load_bed(bed, N, row_begin=lo, row_end=hi, n_global=N)
        ch = capi.BwChain(dev, y, fail, seed=1222, quad=9)
        if X is not None:
            ch.set_covariates(X)
        for _ in range(iters):
            ch.iterate()
        beta, comp = ch.beta()
        st = ch.state()
        q.put((rank, beta, comp, st["mu"], st["alpha"], st["sigmaG"], dev.get_residual(), ch.last_nnz(), ch.gamma()[0] if X is not None else None))
    except Exception as e:
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("with_cov", [False, True])
def test_bayesw_two_ranks_one_gpu(with_cov):
    import torch.multiprocessing as mp
    from hydra_amd import capi, synth
    M, N, iters = 120, 5000, 3
    geno = synth.make_genotypes(M, N, seed=71, missing_rate=0.01)
    y, fail, _ = synth.make_survival(geno, seed=72, causal_frac=0.05)
    bed = synth.pack_bed_columns(geno)
    X = np.random.default_rng(3).normal(size=(N, 2)) * 0.2 if with_cov else None
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_bw, args=(r, 2, port, bed, y, fail, X, N, iters, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for r in res:
        assert len(r) == 9, "rank %s failed: %s" % (r[0], r[1])
    res.sort(key=lambda r: r[0])

    dev = capi.Device(0)
    dev.load_bed(bed, N)
    ch = capi.BwChain(dev, y, fail, seed=1222, quad=9)
    if with_cov:
        ch.set_covariates(X)
    for _ in range(iters):
        ch.iterate()
    beta, comp = ch.beta()
    st = ch.state()
    tol = lambda a, b: np.all(np.abs(np.asarray(a) - np.asarray(b)) <= 1e-8 * np.maximum(1.0, np.abs(np.asarray(b))))
    # replicas identical, bit for bit (same all-reduced sums, same generators)
    for r in res[1:]:
        assert np.array_equal(res[0][1], r[1]) and np.array_equal(res[0][2], r[2]) and res[0][3] == r[3] and res[0][4] == res[1][4]
    # equal to the single-rank chain up to the summation tree
    assert np.array_equal(res[0][2], comp) and tol(res[0][1], beta) and tol(res[0][3], st["mu"]) and tol(res[0][4], st["alpha"])
    assert tol(res[0][5], st["sigmaG"]) and res[0][7] == ch.last_nnz()
    assert tol(np.concatenate([res[0][6], res[1][6]]), dev.get_residual())
    if with_cov:
        assert tol(res[0][8], ch.gamma()[0])


def test_bench_two_ranks_under_the_launcher():
    """The path the driver's multi-GPU run takes: `python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 ...`,
    here with both ranks on device 0 (HGIBBS_BENCH_DEVICE) and the bulk reductions over gloo (two processes cannot form an
    RCCL communicator on one GPU).  The launcher is a child process started before this process's ranks touch the GPU; rank 0
    prints exactly one JSON line that carries the contract's keys, the whole-job value and the exchange that ran."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HGIBBS_BENCH_BULK="gloo", HGIBBS_BENCH_DEVICE="0", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--N", "40000", "--M", "4000",
           "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["metric"] == "Gibbs markers/sec/iter"
    assert d["unit"] == "markers/s" and d["higher_is_better"] is True and d["scaling"] == "strong" and d["vs_baseline"] is None
    assert abs(d["value"] - 4000 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    assert d["config"]["N"] == 40000 and d["config"]["M"] == 4000
    assert d["config"]["exchange"] in ("p2p-mailbox", "gloo-allreduce (host)") and d["config"]["bulk_reductions"] == "gloo"
    assert "roofline" in d and d["roofline"]["bound"] in ("latency", "hbm") and "cpu_baseline" in d


# ---------------------------------------------------------------------------
# a device that is not ours alone: two INDEPENDENT processes (not ranks of one job), each with a shard whose resident grid
# needs more than half of the compute units, default options
# ---------------------------------------------------------------------------
def _lone_worker(idx, bed, y, N, iters, barrier, q):
    import time
    from hydra_amd import capi
    try:
        dev = capi.Device(0)
        dev.load_bed(bed, N)
        ch = capi.Chain(dev, y, seed=1222, shuffle=1)
        engines, times, comps, betas = [], [], [], []
        for _ in range(iters):
            barrier.wait(timeout=120)  # both processes start their sweeps together
            t0 = time.perf_counter()
            ch.iterate()
            times.append(time.perf_counter() - t0)
            engines.append(dev.sweep_stats()["engine"])
            beta, comp, _ = dev.get_beta()
            comps.append(comp.copy())
            betas.append(beta.copy())
        q.put((idx, engines, times, comps, betas))
    except Exception as e:
        q.put((idx, repr(e)))


def test_two_processes_share_the_device_with_default_options(oracle):
    """The resident grid is a set of workgroups that wait for each other: all of them must be resident at once, and the host's
    occupancy check cannot see another process.  Two processes on one GPU, each with 140 001 individuals (137 streaming
    workgroups + the walker: more than half of the 256 compute units), default engine: whichever engine each sweep ends up on
    -- the kernel's start-of-kernel rendezvous finds a partly resident grid within 0.1 s, touches nothing, and the library runs
    that sweep and the following ones on the batch engine -- both walk the oracle's chain, and no sweep stalls for seconds."""
    import torch.multiprocessing as mp
    import orc
    from hydra_amd import synth
    M, N, iters = 200, 140001, 4
    geno = synth.make_genotypes(M, N, seed=71, missing_rate=0.0)
    y, _ = synth.make_phenotype(geno, seed=72, causal_frac=0.05)
    bed = synth.pack_bed_columns(geno)
    ref = orc.Chain(oracle, bed, N, y, seed=1222, shuffle=1)
    want = []
    for _ in range(iters):
        ref.iterate()
        want.append((ref.arr("components").copy(), ref.arr("beta").copy()))
    ctx = mp.get_context("spawn")
    q, barrier = ctx.Queue(), ctx.Barrier(2)
    procs = [ctx.Process(target=_lone_worker, args=(i, bed, y, N, iters, barrier, q)) for i in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for r in res:
        assert len(r) == 5, "process %s failed: %s" % (r[0], r[1])
        idx, engines, times, comps, betas = r
        assert set(engines) <= {1, 2}
        assert max(times) < 1.5, "a sweep stalled: %s (engines %s)" % (times, engines)
        for it in range(iters):
            assert np.array_equal(comps[it], want[it][0]), "process %d diverged from the oracle at iteration %d (engines %s)" % (idx, it, engines)
            assert np.all(np.abs(betas[it] - want[it][1]) <= 1e-9 * np.maximum(1.0, np.abs(want[it][1])))
        # once a grid was found partly resident the handle stays on the batch engine
        if 1 in engines:
            assert engines[engines.index(1):] == [1] * (iters - engines.index(1))
