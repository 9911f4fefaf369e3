"""The steady state of the hot loop against the oracle.

At the headline sizes a workgroup of k_sweep_batch streams several tile groups
(4096 individuals each) per launch: the loop body of src/BayesRRm.cpp:1770-1809's
replacement then runs its next-tile prefetch (column dwords, pivot and pending
columns, the second LDS-DMA of the residual tile), hands the prefetched registers
over and accumulates dots and 16-bit Gram partials across tiles.  The small parity
shapes of test_gpu_parity.py give every workgroup exactly one tile group, so these
tests force several -- few slices at N = 20 011 for every build of the kernel the
dispatcher can pick, and shapes of 70 K-130 K individuals with the default options
-- and compare each chain with the oracle.  `tiles_per_workgroup_max` of
hgibbs_last_sweep_stats is asserted > 1 so that a change of the launch geometry
cannot silently turn them single-tile again.

Bar as in test_gpu_parity.py: components, cass, order, generator exact; beta,
acum, residual and hyper-parameters to 1e-9.
"""
import os
import socket

import numpy as np
import pytest

import orc
from hydra_amd import capi, synth

pytestmark = pytest.mark.gpu

TOL = 1e-9


def close(a, b, tol=TOL):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.all(np.abs(a - b) <= tol * np.maximum(1.0, np.abs(b)))


_CASES = {}


def _case(M, N, miss_cols, seed, causal=0.04):
    """Genotypes with missing calls (2 %) in a share `miss_cols` of the columns; cached per shape."""
    key = (M, N, miss_cols, seed, causal)
    if key not in _CASES:
        geno = synth.make_genotypes(M, N, seed=seed, missing_rate=0.0)
        rng = np.random.default_rng(seed + 1)
        for c in rng.choice(M, size=int(miss_cols * M), replace=False):
            geno[c, rng.random(N) < 0.02] = 3
        y, _ = synth.make_phenotype(geno, seed=seed + 2, causal_frac=causal)
        _CASES.clear()  # one shape at a time: the big ones are 50 MB of int8
        _CASES[key] = (synth.pack_bed_columns(geno), y)
    return _CASES[key]


def _run_vs_oracle(oracle, bed, y, N, opts, iters, mS=None, groups=None, seed=31, min_tiles=2):
    ref = orc.Chain(oracle, bed, N, y, groups=groups, mS=mS, seed=seed, shuffle=1)
    dev = capi.Device(0)
    dev.load_bed(bed, N)
    for k, v in opts.items():
        dev.set_option(k, v)
    ch = capi.Chain(dev, y, mS=mS, groups=groups, seed=seed, shuffle=1)
    tiles, streamed, carried = 0, 0, 0
    for it in range(iters):
        ref.iterate()
        ch.iterate()
        beta, comp, acum = dev.get_beta()
        st = ch.state()
        what = "it %d %r" % (it, opts)
        assert np.array_equal(ch.order(), ref.arr("order")), what
        assert np.array_equal(comp, ref.arr("components")), what
        assert np.array_equal(st["cass"].ravel(), ref.arr("cass")), what
        assert close(beta, ref.arr("beta")) and close(acum, ref.arr("acum")), what
        assert close(dev.get_residual(), ref.arr("eps")), what
        assert close(st["sigmaG"], ref.arr("sigmaG")) and close(st["sigmaE"], ref.sigmaE) and close(st["mu"], ref.mu), what
        assert ch.last_nnz() == oracle.orc_chain_last_nnz(ref.h), what
        s = dev.sweep_stats()
        assert s["accepted_markers"] == bed.shape[0]
        if s["engine"] == 1:
            assert s["working_launches"] <= s["launches"]
        else:  # the resident engine: one launch per sweep, rounds of its walker
            assert s["launches"] == 1 and s["rounds"] >= 1
        tiles = max(tiles, s["tiles_per_workgroup_max"])
        streamed += s["streamed_columns"]
        carried += s["carried_columns"]
    if s["engine"] == 1:
        assert tiles >= min_tiles, "every workgroup streamed %d tile group(s): the multi-tile loop did not run" % tiles
    else:
        assert opts.get("engine", 0) != 1, "the batch engine was asked for"

    dev.close()
    return {"tiles": tiles, "streamed": streamed, "carried": carried}


# every build of k_sweep_batch<CPG, SEG, MG, NOMISS> the dispatcher of hgibbs_sweep can pick, by the options and
# the data that select it (share of columns with missing calls; 0 => the NOMISS builds at cols_per_group 4)
BUILDS = {
    "4,2,0,1": (0.0, {"cols_per_group": 4, "max_seg": 2}),
    "4,2,0,0": (0.3, {"cols_per_group": 4, "max_seg": 2, "gram_missing": 0}),
    "4,2,1,0": (0.3, {"cols_per_group": 4, "max_seg": 2, "gram_missing": 1}),
    "8,2,1,0": (1.0, {"cols_per_group": 8, "max_seg": 2, "gram_missing": 1}),
    "4,4,0,1": (0.0, {"cols_per_group": 4, "max_seg": 4}),
    "4,4,0,0": (0.3, {"cols_per_group": 4, "max_seg": 4, "gram_missing": 0}),
    "8,4,0,0": (0.1, {"cols_per_group": 8, "max_seg": 4, "gram_missing": 0}),
    "8,2,0,0": (0.1, {"cols_per_group": 8, "max_seg": 2, "gram_missing": 0}),
    "2,2,0,0": (0.1, {"cols_per_group": 2, "max_seg": 2}),
    "16,2,0,0": (0.1, {"cols_per_group": 16, "max_seg": 2}),
}


@pytest.mark.parametrize("carry", [1, 0])
@pytest.mark.parametrize("slices", [1, 2])
@pytest.mark.parametrize("build", sorted(BUILDS))
def test_every_build_streams_several_tiles_vs_oracle(oracle, build, slices, carry):
    """N = 20 011 is five tile groups: one slice gives every workgroup five passes through the loop body, two slices
    three and two (uneven: the last prefetch of the shorter slice is skipped).  Four iterations, so that effects are
    non-zero (predicted events, Gram-corrected segments, carried dots) from the second one on."""
    miss_cols, opts = BUILDS[build]
    M, N = 480, 20011
    bed, y = _case(M, N, miss_cols, seed=900 + int(100 * miss_cols))
    # with carry on, half of the cases also stream ahead of the batch (option ahead: the columns behind the batch are taken
    # from a queue while the last workgroup draws, and reach the next launch as carried columns with raw sums)
    ahead = 40 if (carry and slices == 2) else 0
    r = _run_vs_oracle(oracle, bed, y, N, dict(opts, batch=256, slices=slices, carry=carry, ahead=ahead), iters=4,
                       min_tiles=5 if slices == 1 else 3)
    if carry and build.split(",")[2] == "0":  # the four-term build (MG) carries nothing
        assert r["carried"] > 0, "no column was ever carried: the carry path did not run"


@pytest.mark.parametrize("N,miss_cols,max_seg,carry,ahead", [(70001, 0.0, 0, -1, 0), (100003, 0.0, 2, 1, 64), (100003, 0.25, 0, 1, 0),
                                                               (130001, 1.0, 2, 1, 128), (130001, 0.0, 4, -1, 0), (70001, 0.1, 2, 1, 256)])
def test_large_shard_default_geometry_vs_oracle(oracle, N, miss_cols, max_seg, carry, ahead):
    """Shapes of 70 K-130 K individuals with the library's own launch geometry (64 column groups of four columns,
    slices = co-resident workgroups / groups): 18-32 tile groups over 12 slices, two to three per workgroup, three
    iterations, with and without missing calls (the NOMISS, plain and four-term builds) -- what configs 3 and 4 run,
    at a size the oracle finishes in seconds."""
    M = 400
    bed, y = _case(M, N, miss_cols, seed=77 + N % 100, causal=0.05)
    opts = {"max_seg": max_seg} if max_seg else {}
    opts.update({"carry": carry, "ahead": ahead})  # carried dots are off by default below 400 000 individuals: forced on in some cases
    _run_vs_oracle(oracle, bed, y, N, opts, iters=3)


@pytest.mark.parametrize("engine", [1, 2])
def test_grouped_mixture_multi_tile_vs_oracle(oracle, engine):
    """Config 3's model (two groups, its mixture variances) on a multi-tile shape, by both engines (the resident one: 88
    streaming workgroups of one wave tile each)."""
    M, N = 400, 90001
    bed, y = _case(M, N, 0.0, seed=5)
    groups = (np.arange(M) % 2).astype(np.int32)
    mS = np.array([[0.0, 0.001, 0.01, 0.1]] * 2)
    _run_vs_oracle(oracle, bed, y, N, {"engine": engine}, iters=3, mS=mS, groups=groups)


@pytest.mark.parametrize("walker", [1, 2])
@pytest.mark.parametrize("miss_cols", [0.0, 0.25])
def test_resident_two_tiles_64_workgroups_vs_oracle(oracle, miss_cols, walker):
    """The geometry the headline runs -- two wave tiles per streaming workgroup, many workgroups -- against the ORACLE: 130 001
    individuals on 65 compute units (64 streaming workgroups of 2 048 individuals + the walker), clean and with 2 % missing calls
    in a quarter of the columns (the missing-call build next to clean columns), by both walkers."""
    M, N = 400, 130001
    bed, y = _case(M, N, miss_cols, seed=11)
    dev_opts = {"engine": 2, "res_cus": 65, "walker": walker}
    _run_vs_oracle(oracle, bed, y, N, dev_opts, iters=3)
    d = capi.Device(0)
    d.load_bed(bed, N)
    for k, v in dev_opts.items():
        d.set_option(k, v)
    ch = capi.Chain(d, y, seed=31, shuffle=1)
    ch.iterate()
    ss = d.sweep_stats()
    assert ss["engine"] == 2 and ss["tiles_per_workgroup_max"] == 2 and ss["walker"] == walker
    d.close()


@pytest.mark.parametrize("N,T,miss_cols", [(500000, 2, 0.0), (500000, 2, 0.5), (1000000, 4, 0.0)])
def test_headline_geometry_vs_oracle(oracle, N, T, miss_cols):
    """The headline's own geometry against the ORACLE, directly: N = 500 000 individuals on the whole device -- 245 streaming workgroups of
    two tiles and the walker, default options -- with few enough markers (256) for the oracle to follow in seconds; clean, and with 2 %
    missing calls in half of the columns (build MISS).  And a million individuals: four tiles per workgroup."""
    M = 256 if N <= 500000 else 160
    bed, y = _case(M, N, miss_cols, seed=17)
    _run_vs_oracle(oracle, bed, y, N, {}, iters=3)
    d = capi.Device(0)
    d.load_bed(bed, N)
    ch = capi.Chain(d, y, seed=31, shuffle=1)
    ch.iterate()
    ss = d.sweep_stats()
    assert ss["engine"] == 2 and ss["tiles_per_workgroup_max"] == T and ss["walker"] == 2 and ss["refill"] == 2
    d.close()


# ---------------------------------------------------------------------------
# sharded: two processes on one GPU, more than 49 152 individuals per rank (several tile groups per workgroup on
# every rank), each replica against the oracle
# ---------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _shard(N, world, rank):
    per = ((N + world - 1) // world + 3) // 4 * 4
    return min(N, rank * per), min(N, (rank + 1) * per)


def _worker(rank, world, port, bed, y, N, iters, exchange, opts, q):
    import torch
    import torch.distributed as dist
    from hydra_amd import capi as cp
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = cp.Device(0)

        def allreduce(arr):
            t = torch.from_numpy(arr.view(np.int64) if arr.dtype == np.uint64 else arr)
            dist.all_reduce(t)

        if exchange == "rccl":
            uid = [cp.Device.unique_id() if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0)
            dev.comm_init(world, rank, uid[0])
            dev.set_option("p2p", 0)
        else:
            dev.comm_init_external(world, rank, allreduce)
            if exchange == "p2p":
                handles = [None] * world
                dist.all_gather_object(handles, dev.p2p_export())
                dev.p2p_import(handles)
            else:  # "external": dots -> the caller's all-reduce -> draw, one stream round trip per batch
                dev.set_option("p2p", 0)
        lo, hi = _shard(N, world, rank)
        dev.load_bed(bed, N, row_begin=lo, row_end=hi, n_global=N)
        for k, v in opts.items():
            dev.set_option(k, v)
        ch = cp.Chain(dev, y, seed=1222)
        out = []
        for _ in range(iters):
            ch.iterate()
            beta, comp, acum = dev.get_beta()
            st = ch.state()
            s = dev.sweep_stats()
            out.append((beta, comp, acum, st["sigmaE"], st["sigmaG"], st["mu"], dev.get_residual(), ch.last_nnz(),
                        s["tiles_per_workgroup_max"], s["launches"]))
        q.put((rank, out))
    except Exception as e:  # surface the error text in the parent
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def _two_ranks_vs_oracle(oracle, exchange, M, N, iters, opts, miss=0.0, min_tiles=2):
    import torch.multiprocessing as mp
    world = 2
    geno = synth.make_genotypes(M, N, seed=61, missing_rate=miss)
    y, _ = synth.make_phenotype(geno, seed=62, causal_frac=0.05)
    bed = synth.pack_bed_columns(geno)
    del geno
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, bed, y, N, iters, exchange, opts, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for r in res:
        assert isinstance(r[1], list), "rank %s failed: %s" % (r[0], r[1])
    res.sort(key=lambda r: r[0])
    ref = orc.Chain(oracle, bed, N, y, seed=1222, shuffle=1)
    for it in range(iters):
        ref.iterate()
        a, b = res[0][1][it], res[1][1][it]
        # the replicas agree bit for bit (same summed rows, same generator)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[3] == b[3] and a[5] == b[5]
        # and walk the oracle's chain
        assert np.array_equal(a[1], ref.arr("components")), "it %d" % it
        assert close(a[0], ref.arr("beta")) and close(a[2], ref.arr("acum")), "it %d" % it
        assert close(a[3], ref.sigmaE) and close(a[4], ref.arr("sigmaG")) and close(a[5], ref.mu)
        assert close(np.concatenate([a[6], b[6]]), ref.arr("eps"))
        assert a[7] == oracle.orc_chain_last_nnz(ref.h)
        assert min(a[8], b[8]) >= min_tiles, "a rank streamed single-tile: %d / %d" % (a[8], b[8])
    return res


def test_two_ranks_mailbox_large_shards_vs_oracle(oracle):
    """In-launch peer mailboxes, 55 002 individuals per rank (14 tile groups over 12 slices: two for some workgroups),
    default options (four-segment build, carried dots: the carry term is one of the exchanged rows)."""
    _two_ranks_vs_oracle(oracle, "p2p", M=500, N=110004, iters=3, opts={"carry": 1, "ahead": 48})


def test_two_ranks_mailbox_few_slices_vs_oracle(oracle):
    """The same exchange with two slices only: seven tile groups per workgroup on every rank, missing calls."""
    _two_ranks_vs_oracle(oracle, "p2p", M=400, N=110004, iters=3, opts={"slices": 2, "max_seg": 2, "carry": 1}, miss=0.01, min_tiles=7)


def test_two_ranks_split_path_large_shards_vs_oracle(oracle):
    """dots -> all-reduce of the batch rows -> draw as three steps (k_sweep_batch with sums_out, the transport,
    k_sweep_draw) with two ranks and more than 49 152 individuals each.  Two ranks cannot share one GPU under RCCL
    (ncclCommInitRank refuses a duplicate device), so on this one-GPU box the rows travel through the caller's
    transport (gloo here, MPI_Allreduce in hydra: src/BayesRRm.cpp:2456); the kernels on either side of the
    exchange are the ones the RCCL path runs."""
    _two_ranks_vs_oracle(oracle, "external", M=300, N=110004, iters=2, opts={})


def test_two_ranks_per_marker_allreduce_is_the_parity_baseline(oracle):
    """north_star's scheme verbatim: one all-reduce of (s1, s2) per marker (batch = 1, split path), two ranks.
    Slow by construction (a stream round trip per marker); kept as the baseline every faster exchange is compared
    with.  Small M, shards still beyond 49 152 individuals."""
    _two_ranks_vs_oracle(oracle, "external", M=60, N=110004, iters=2, opts={"batch": 1, "slices": 7}, min_tiles=2)


def test_two_ranks_rccl_on_one_device_is_refused_or_matches(oracle):
    """If this RCCL build accepts two ranks on one device the per-batch RCCL all-reduce runs for real and must give
    the oracle's chain; if it refuses (duplicate GPU), the refusal must surface as an error, not as a hang."""
    import torch.multiprocessing as mp
    M, N = 200, 110004
    geno = synth.make_genotypes(M, N, seed=61, missing_rate=0.0)
    y, _ = synth.make_phenotype(geno, seed=62, causal_frac=0.05)
    bed = synth.pack_bed_columns(geno)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, bed, y, N, 2, "rccl", {}, q)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        res = [q.get(timeout=300) for _ in procs]
    finally:
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.terminate()
    if all(isinstance(r[1], list) for r in res):
        res.sort(key=lambda r: r[0])
        ref = orc.Chain(oracle, bed, N, y, seed=1222, shuffle=1)
        for it in range(2):
            ref.iterate()
            a = res[0][1][it]
            assert np.array_equal(a[1], ref.arr("components")) and close(a[0], ref.arr("beta"))
    else:
        msgs = [r[1] for r in res if not isinstance(r[1], list)]
        assert all("HgError" in m or "nccl" in m.lower() or "rccl" in m.lower() for m in msgs), msgs
        pytest.skip("RCCL refuses two ranks on one device here: %s" % msgs[0][:160])
