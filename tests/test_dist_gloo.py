"""N>1 path on CPU: individuals sharded over 2 ranks (gloo), the per-marker
exchange is ONE all-reduce of the two scalars (s1, s2), every
rank runs the identical draw on an identical replicated MT19937 state and
updates only its own residual shard.  The emulation below follows the device
algorithm step for step with the oracle's per-marker primitives and must give
(a) bit-identical replicas and (b) the unsharded oracle chain (components
exact, beta to 1e-9).  Also checks the shard arithmetic bench.py / the CLI use.
"""
import ctypes as C
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import orc
from hydra_amd import synth


def shard_rows(N, world, rank):
    per = ((N + world - 1) // world + 3) // 4 * 4
    return min(N, rank * per), min(N, (rank + 1) * per)


def test_shard_rows_cover_and_align():
    for N in (5, 64, 1001, 500000):
        for world in (1, 2, 3, 8):
            spans = [shard_rows(N, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == N
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c
            assert all(lo % 4 == 0 for lo, hi in spans if hi > lo)  # empty trailing shards are rejected by the launcher


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, bed, y, N, M, iters, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    L = orc.load()
    lo, hi = shard_rows(N, world, rank)
    nloc = hi - lo
    # local packed shard (byte-aligned slice, tail padded as missing)
    geno_loc = synth.unpack_bed_columns(bed, N)[:, lo:hi]
    bed_loc = synth.pack_bed_columns(geno_loc)
    # global stats from all-reduced counts
    counts = np.zeros((M, 3), dtype=np.int64)
    for j in range(M):
        c = [C.c_uint64() for _ in range(4)]
        L.orc_bed_counts(orc.u8ptr(np.ascontiguousarray(bed_loc[j])), nloc, *[C.byref(x) for x in c])
        counts[j] = (c[1].value, c[2].value, c[3].value)
    t = torch.from_numpy(counts)
    dist.all_reduce(t)
    mave, mstd = np.zeros(M), np.zeros(M)
    for j in range(M):
        a, s = C.c_double(), C.c_double()
        L.orc_marker_stats(int(counts[j, 0]), int(counts[j, 1]), int(counts[j, 2]), N, C.byref(a), C.byref(s))
        mave[j], mstd[j] = a.value, s.value
    # replicated host state: run the oracle's own init on the full y to get identical
    # y-scaling, sigmaE, sigmaG draw and rng state (cheap; the hot loop below is sharded)
    full = orc.Chain(L, bed, N, y, seed=1222, shuffle=1)
    K = 4
    cVa, cVaI = full.arr("cVa").copy(), full.arr("cVaI").copy()
    estPi, sigmaG = full.arr("estPi").copy(), full.arr("sigmaG").copy()
    sigmaE, mu = full.sigmaE, 0.0
    rng = L.orc_chain_rng(full.h)  # our replica of the shared generator
    eps = full.arr("y")[lo:hi].copy()
    beta, comp = np.zeros(M), np.zeros(M, dtype=np.int32)
    order = np.arange(M, dtype=np.int32)
    for it in range(iters):
        # mu step: all-reduced sum
        eps += mu
        s = torch.tensor([eps.sum()], dtype=torch.float64)
        dist.all_reduce(s)
        mu = L.orc_rng_norm(rng, float(s[0]) / N, sigmaE / N)
        eps -= mu
        L.orc_rng_shuffle(rng, orc.iptr(order), M)
        cass = np.zeros(K, dtype=np.int64)
        for j in order:
            col = np.ascontiguousarray(bed_loc[j])
            s1, s2 = C.c_double(), C.c_double()
            L.orc_dot_dense(orc.u8ptr(col), orc.dptr(eps), nloc, 0.0, 1.0, C.byref(s1), C.byref(s2))
            red = torch.tensor([s1.value, s2.value], dtype=torch.float64)
            dist.all_reduce(red)  # THE per-marker exchange: 2 x f64
            num = mstd[j] * (float(red[0]) - mave[j] * float(red[1]))
            bnew, k, acum = C.c_double(), C.c_int(), C.c_double()
            rc = L.orc_marker_draw(num, beta[j], N, K, orc.dptr(cVa), orc.dptr(cVaI), orc.dptr(estPi), sigmaE,
                                   float(sigmaG[0]), rng, C.byref(bnew), C.byref(k), C.byref(acum))
            assert rc == 0
            db = beta[j] - bnew.value
            beta[j], comp[j] = bnew.value, k.value
            cass[k.value] += 1
            if db != 0.0:
                L.orc_update(orc.u8ptr(col), orc.dptr(eps), nloc, mave[j], mstd[j], db)
        # hyper-parameters (replicated)
        m0 = M - cass[0]
        bsq = float((beta * beta).sum())
        if m0 > 0:
            sigmaG[0] = L.orc_rng_inv_scaled_chisq(rng, 0.0001 + m0, (bsq * m0 + 0.0001 * 0.0001) / (0.0001 + m0))
            alpha = cass.astype(np.float64) + 1.0
            pi = np.zeros(K)
            L.orc_rng_dirichlet(rng, orc.dptr(alpha), K, orc.dptr(pi))
            estPi[:] = pi
        q2 = torch.tensor([(eps * eps).sum()], dtype=torch.float64)
        dist.all_reduce(q2)
        sigmaE = L.orc_rng_inv_scaled_chisq(rng, 0.0001 + N, (float(q2[0]) + 0.0001 * 0.0001) / (0.0001 + N))
    q.put((rank, beta.copy(), comp.copy(), sigmaE, float(sigmaG[0]), eps.copy(), (lo, hi)))
    dist.destroy_process_group()


def test_two_rank_sharded_chain_matches_unsharded_oracle():
    M, N, iters = 60, 203, 3
    geno = synth.make_genotypes(M, N, seed=5, missing_rate=0.03)
    y, _ = synth.make_phenotype(geno, seed=6, causal_frac=0.1)
    bed = synth.pack_bed_columns(geno)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, bed, y, N, M, iters, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # (a) replicas are bit-identical
    assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])
    assert res[0][3] == res[1][3] and res[0][4] == res[1][4]
    # (b) equals the unsharded oracle (dense form), components exact
    L = orc.load()
    L.orc_set_dot_form(1)
    try:
        ref = orc.Chain(L, bed, N, y, seed=1222, shuffle=1)
        for _ in range(iters):
            ref.iterate()
        rb = ref.arr("beta")
        assert np.array_equal(res[0][2], ref.arr("components"))
        assert np.all(np.abs(res[0][1] - rb) <= 1e-9 * np.maximum(1.0, np.abs(rb)))
        assert abs(res[0][3] - ref.sigmaE) <= 1e-9 * ref.sigmaE
        eps = np.concatenate([res[0][5], res[1][5]])
        assert np.allclose(eps, ref.arr("eps"), rtol=0, atol=1e-9)
    finally:
        L.orc_set_dot_form(0)


# ---- BayesW: the per-batch row block and the failure counts add over the ranks ---------------------------
def _worker_bw(rank, world, port, geno, eps, fail, alpha, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    L = orc.load()
    orc._bind_bw(L)
    M, N = geno.shape
    lo, hi = shard_rows(N, world, rank)
    g, e, d = geno[:, lo:hi], eps[lo:hi], fail[lo:hi]
    # load time: integer counts add over the ranks (n1, n2, nmiss, failures among genotype 1 / 2)
    cnt = np.stack([(g == 1).sum(1), (g == 2).sum(1), (g == 3).sum(1), ((g == 1) * d).sum(1), ((g == 2) * d).sum(1)], axis=1).astype(np.int64)
    t = torch.from_numpy(cnt)
    dist.all_reduce(t)
    # per batch: the row block rows[2c] = sum vi over genotype 1, rows[2c+1] over genotype 2, last = sum vi
    vi = np.exp(alpha * e - 0.577215664901532)
    rows = np.zeros(2 * M + 1)
    for c in range(M):
        rows[2 * c] = vi[g[c] == 1].sum()
        rows[2 * c + 1] = vi[g[c] == 2].sum()
    rows[2 * M] = vi.sum()
    r = torch.from_numpy(rows)
    dist.all_reduce(r)
    # every rank evaluates the same marginals and picks from the same reduced rows
    n1, n2, nm, f1, f2 = [cnt[:, k].astype(np.float64) for k in range(5)]
    mave = (n1 + 2 * n2) / (N - nm)
    sd = np.sqrt(((N - n1 - n2 - nm) * mave ** 2 + n1 * (1 - mave) ** 2 + n2 * (2 - mave) ** 2) / (N - 1))
    sumfail = ((f1 + 2 * f2) - mave * fail.sum()) / sd
    pi, cva = np.array([0.9, 0.06, 0.04]), np.array([0.001, 0.01])
    mls = np.zeros((M, 3))
    for c in range(M):
        ml = np.zeros(3)
        L.orc_bw_marginals(9, 3, orc.dptr(pi), orc.dptr(cva), alpha, 0.05, float(sumfail[c]), float(rows[2 * M]), float(rows[2 * c + 1]),
                           float(rows[2 * c]), float(rows[2 * M] - rows[2 * c] - rows[2 * c + 1]), float(mave[c]), float(sd[c]), orc.dptr(ml))
        mls[c] = ml
    q.put((rank, cnt, rows, mls))
    dist.destroy_process_group()


def test_bayesw_row_block_allreduce_matches_unsharded():
    M, N, alpha = 24, 1003, 3.5
    geno = synth.make_genotypes(M, N, seed=5, missing_rate=0.02)
    y, fail, _ = synth.make_survival(geno, seed=6, causal_frac=0.1)
    eps = y - y.mean()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_bw, args=(r, 2, port, geno, eps, fail.astype(np.float64), alpha, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=30)
    # replicas: same bits
    assert all(np.array_equal(a, b) for a, b in zip(res[0][1:], res[1][1:]))
    # unsharded
    L = orc.load()
    ref = orc.BwChain(L, synth.pack_bed_columns(geno), N, y, fail)
    cnt = res[0][1]
    n1, n2, nm = [cnt[:, k].astype(np.float64) for k in range(3)]
    assert np.array_equal((n1 + 2 * n2) / (N - nm), ref.arr("mave"))  # tables from all-reduced integer counts: exact
    f = fail.astype(np.float64)
    sumfail = ((cnt[:, 3] + 2 * cnt[:, 4]) - ref.arr("mave") * f.sum()) / ref.arr("msd")
    assert np.array_equal(sumfail, ref.arr("sum_failure"))
    vi = np.exp(alpha * eps - 0.577215664901532)
    want = np.array([[vi[geno[c] == 1].sum(), vi[geno[c] == 2].sum()] for c in range(M)]).ravel()
    assert np.allclose(res[0][2][:-1], want, rtol=1e-13, atol=0) and abs(res[0][2][-1] - vi.sum()) <= 1e-12 * vi.sum()
    assert np.all(res[0][3] > 0) and np.all(np.isfinite(res[0][3]))
