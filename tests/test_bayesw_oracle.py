"""BayesW, CPU side: what pins the oracle (oracle/orc_bayesw.cpp, orc_ars.h) and the
host-only pieces of the product (hg_ars.h, the glibc rand() restatement).

Pins available here: the reference's own ARS sampler compiled from src/BayesW_arms.cpp
(oracle/_ref/libarms.so), libc's rand(), and the quadrature literals (checked against the
reference's text by tools/gen_gh_tables.py when /root/reference is present).  The rest of
the BayesW chain is an unpinned restatement (the reference needs Eigen + Boost + MPI)."""
import ctypes as C
import math
import os
import subprocess
import sys

import numpy as np
import pytest

import orc
from hydra_amd import capi, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libc = C.CDLL(None)
DENS = C.CFUNCTYPE(C.c_double, C.c_double, C.c_void_p)
dp = C.POINTER(C.c_double)
ARMS_SIG = [dp, C.c_int, dp, dp, DENS, C.c_void_p, dp, C.c_int, C.c_int, dp, dp, C.c_int, dp, dp, C.c_int, C.POINTER(C.c_int)]

CASES = {
    "normal": (lambda x: -0.5 * x * x, [-1.0, -0.2, 0.3, 1.1], -6.0, 6.0),
    "gamma": (lambda x: 2.5 * math.log(x) - 3 * x, [0.2, 0.6, 1.0, 2.0], 1e-3, 20.0),
    # beta_dens-shaped: -a x d - exp(a x m/s) (v0 + v1 exp(-a x/s) + v2 exp(-2 a x/s)) - x^2 / (2 C sigma)
    "weibull": (lambda x: -1.3 * x * 4.0 - math.exp(1.3 * x * 0.8) * (50 + 30 * math.exp(-1.3 * x / 0.6) + 8 * math.exp(-2 * 1.3 * x / 0.6))
                - x * x / (2 * 0.01 * 0.5), [-0.05, 0.0, 0.025, 0.05], -0.5, 0.5),
    "narrow": (lambda x: -0.5 * (x / 1e-3) ** 2, [-1.0, -0.2, 0.3, 1.1], -6.0, 6.0),  # many rejections, hull grows
    "not_concave": (lambda x: -abs(x) ** 0.5, [-1.0, -0.2, 0.3, 1.1], -6.0, 6.0),   # error 2000 after a few draws
    "bad_order": (lambda x: -0.5 * x * x, [-1.0, -2.0, 0.0, 1.0], -6.0, 6.0),         # 1004
    "bad_bounds": (lambda x: -0.5 * x * x, [-7.0, -2.0, 0.0, 1.0], -6.0, 6.0),        # 1003
}


def run_arms(fn, dens, xinit, xl, xr, n, seed):
    """n successive one-sample calls on the libc rand() stream seeded with `seed`."""
    libc.srand(seed)
    cb, out = DENS(lambda x, _d: dens(x)), []
    for _ in range(n):
        xi = (C.c_double * 4)(*xinit)
        xl_, xr_, cv, xp = C.c_double(xl), C.c_double(xr), C.c_double(1.0), C.c_double(0.0)
        xs, qc, xc, ne = (C.c_double * 1)(), (C.c_double * 10)(5., 30., 70., 95.), (C.c_double * 10)(), C.c_int(0)
        err = fn(xi, 4, C.byref(xl_), C.byref(xr_), cb, None, C.byref(cv), 100, 0, C.byref(xp), xs, 1, qc, xc, 4, C.byref(ne))
        out.append((err, xs[0], ne.value))
        if err:
            break
    return out


def run_product(dens, xinit, xl, xr, n, seed):
    g = capi.GrandState()
    capi.lib().hgibbs_grand_seed(C.byref(g), seed)
    out = []
    for _ in range(n):
        r = capi.ars_sample(dens, xinit, xl, xr, g)
        out.append(r)
        if r[0]:
            break
    return out


@pytest.fixture(scope="module")
def orc_arms(oracle):
    fn = oracle.orc_ars_arms_c
    fn.argtypes, fn.restype = ARMS_SIG, C.c_int
    return fn


@pytest.fixture(scope="module")
def ref_arms():
    lib = orc.ref_arms_lib()
    if lib is None:
        pytest.skip("oracle/_ref/libarms.so not built (no /root/reference at build time)")
    fn = getattr(lib, orc.REF_ARMS_SYMBOL)
    fn.argtypes, fn.restype = ARMS_SIG, C.c_int
    return fn


@pytest.mark.parametrize("case", sorted(CASES))
def test_oracle_ars_is_the_reference_ars(orc_arms, ref_arms, case):
    dens, xinit, xl, xr = CASES[case]
    a = run_arms(ref_arms, dens, xinit, xl, xr, 250, 7)
    b = run_arms(orc_arms, dens, xinit, xl, xr, 250, 7)
    assert a == b  # samples, evaluation counts and error codes, bit for bit
    if case == "not_concave":
        assert a[-1][0] == 2000
    if case == "bad_order":
        assert a == [(1004, 0.0, 0)]
    if case == "bad_bounds":
        assert a == [(1003, 0.0, 0)]


@pytest.mark.parametrize("case", sorted(CASES))
def test_product_ars_and_rand_stream_match_the_oracle(orc_arms, case):
    """hg_ars.h (sorted-array hull) + the private glibc rand() restatement == orc_ars.h on libc rand()."""
    dens, xinit, xl, xr = CASES[case]
    want = run_arms(orc_arms, dens, xinit, xl, xr, 250, 11)
    got = run_product(dens, xinit, xl, xr, 250, 11)
    if want[-1][0]:
        assert got[-1][0] == want[-1][0] and got[:-1] == want[:-1]
    else:
        assert got == want


@pytest.mark.parametrize("seed", [0, 1, 2, 1222, 123456789, 2 ** 31 - 1, 2 ** 32 - 1])
def test_grand_is_libc_rand(seed):
    L = capi.lib()
    g = capi.GrandState()
    L.hgibbs_grand_seed(C.byref(g), seed)
    libc.srand(seed)
    libc.rand.restype = C.c_int
    assert [L.hgibbs_grand_next(C.byref(g)) for _ in range(2000)] == [libc.rand() for _ in range(2000)]


def test_gh_tables_product_and_oracle_copies_agree_and_generator_checks_the_reference():
    a = open(os.path.join(ROOT, "oracle", "gh_tables.h")).read().replace("ORC_GH", "X").replace("orc_gh", "x")
    b = open(os.path.join(ROOT, "hydra_amd", "csrc", "hg_gh_tables.h")).read().replace("HG_GH", "X").replace("hg_gh", "x")
    strip = lambda t: [l for l in t.splitlines() if "copy" not in l]
    assert strip(a) == strip(b)
    if os.path.exists("/root/reference/src/BayesW.cpp"):
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import gen_gh_tables
        assert gen_gh_tables.check_against_reference(gen_gh_tables.tables()) == 201


def _case(M=260, N=900, seed=3, missing_rate=0.01):
    geno = synth.make_genotypes(M, N, seed=seed, missing_rate=missing_rate)
    y, fail, beta = synth.make_survival(geno, seed=seed + 1, causal_frac=0.05)
    return geno, synth.pack_bed_columns(geno), y, fail, beta


def _run(oracle, bed, N, y, fail, iters, ref_lib=None, **kw):
    c = orc.BwChain(oracle, bed, N, y, fail, **kw)
    if ref_lib is not None:
        c.use_reference_arms(ref_lib)
    out = []
    for it in range(iters):
        c.iterate()
        out.append((c.mu, c.alpha, c.arr("sigmaG").copy(), c.arr("beta").copy(), c.arr("components").copy(), c.csv_line(it)))
    return out


def test_oracle_chain_runs_the_same_on_the_reference_arms(oracle):
    lib = orc.ref_arms_lib()
    if lib is None:
        pytest.skip("oracle/_ref/libarms.so not built")
    _, bed, y, fail, _ = _case()
    groups = (np.arange(bed.shape[0]) % 2).astype(np.int32)
    mS = np.array([[0.0, 0.001, 0.01], [0.0, 0.0001, 0.1]])
    a = _run(oracle, bed, 900, y, fail, 6, seed=5, quad=11, groups=groups, mS=mS)
    b = _run(oracle, bed, 900, y, fail, 6, ref_lib=lib, seed=5, quad=11, groups=groups, mS=mS)
    for x, z in zip(a, b):
        assert x[0] == z[0] and x[1] == z[1] and np.array_equal(x[3], z[3]) and np.array_equal(x[4], z[4]) and x[5] == z[5]
    assert sum(int((x[4] > 0).sum()) for x in a) > 20  # slabs were actually entered


def test_oracle_chain_recovers_the_simulated_model(oracle):
    """Not a parity statement: the restated sampler behaves like a sampler of this model."""
    geno, bed, y, fail, beta = _case(M=300, N=1500, seed=9)
    out = _run(oracle, bed, 1500, y, fail, 40, seed=2, quad=9)
    post = np.mean([o[3] for o in out[20:]], axis=0)
    assert np.corrcoef(post, beta)[0, 1] > 0.9
    assert abs(np.mean([o[0] for o in out[20:]]) - 3.0) < 0.1       # mu
    assert abs(np.mean([o[1] for o in out[20:]]) - 4.0) < 0.6       # alpha
    line = out[-1][5]
    assert line.startswith("   39, ") and line.endswith("\n") and len(line.split(",")) == 8 + 1 + 3


def test_oracle_chain_sensitivity(oracle):
    """How fast the sampler amplifies a rounding-sized perturbation of the residual: this sets the
    tolerance schedule of the GPU parity tests (1e-8 over a few iterations, 1e-4 over 25)."""
    _, bed, y, fail, _ = _case(M=250, N=1000, seed=3)

    def run(pert):
        c = orc.BwChain(oracle, bed, 1000, y, fail, seed=77, quad=9)
        if pert:
            c.arr("eps")[:] *= 1 + pert * np.random.default_rng(0).standard_normal(1000)
        out = []
        for _ in range(25):
            c.iterate()
            out.append((c.arr("eps").copy(), c.arr("components").copy()))
        return out

    a, b = run(0.0), run(1e-14)
    dev = [np.abs(x[0] - z[0]).max() for x, z in zip(a, b)]
    assert all(np.array_equal(x[1], z[1]) for x, z in zip(a, b))  # the discrete path is unchanged
    assert dev[4] < 1e-9 and 1e-9 < dev[24] < 1e-2                # but the floating-point state drifts apart


def test_quadrature_orders(oracle):
    orc._bind_bw(oracle)
    assert [n for n in range(1, 30) if oracle.orc_bw_quad_supported(n)] == [3, 5, 7, 9, 11, 13, 15, 17, 25]
    # orders 3..9 converge; from 11 on the reference integrates with node 6 = -node 3 (its slip, reproduced):
    # order 11 is visibly off, the larger orders less so because that node's weight shrinks
    pi, cva, out = np.array([0.9, 0.1]), np.array([0.01]), {}
    for n in (3, 5, 7, 9, 11, 25):
        ml = np.zeros(2)
        oracle.orc_bw_marginals(n, 2, orc.dptr(pi), orc.dptr(cva), 4.0, 0.1, 20.0, 1000.0, 90.0, 420.0, 490.0, 0.6, 0.65, orc.dptr(ml))
        assert ml[0] == 0.9 * 1.77245385090552
        out[n] = ml[1]
    assert abs(out[3] / out[9] - 1) < 1e-3 and abs(out[5] / out[9] - 1) < 1e-5 and abs(out[7] / out[9] - 1) < 1e-7
    assert abs(out[11] / out[9] - 1) > 1e-2 and abs(out[25] / out[9] - 1) < 1e-8


@pytest.mark.parametrize("with_cov", [False, True])
def test_oracle_bayesw_restore_is_exact(oracle, with_cov):
    """A checkpoint (state + srand(seed + iteration), src/BayesW.cpp:2029) restored into a fresh chain
    (init_from_restart, :869-903, srand(seed + iteration) again, :877) continues the uninterrupted chain."""
    _, bed, y, fail, _ = _case(M=120, N=500, seed=21)
    X = np.random.default_rng(2).normal(size=(500, 2)) * 0.2
    seed = 9

    def fresh(s):
        c = orc.BwChain(oracle, bed, 500, y, fail, seed=s, quad=7)
        if with_cov:
            c.set_covariates(X)
        return c

    a = fresh(seed)
    for _ in range(4):
        a.iterate()
    a.reseed_ars(seed + 3)
    snap = dict(mu=a.mu, alpha=a.alpha, sigmaG=a.arr("sigmaG").copy(), pi=a.arr("pi").copy(), beta=a.arr("beta").copy(),
                components=a.arr("components").copy(), eps=a.arr("eps").copy(), order=a.arr("order").copy(), rng_words=a.rng_words(),
                ars_seed=seed + 3)
    if with_cov:
        snap.update(gamma=a.arr("gamma").copy(), xI=a.xI())
    want = []
    for it in range(4, 7):
        a.iterate()
        want.append((a.arr("beta").copy(), a.arr("components").copy(), a.arr("eps").copy(), a.mu, a.alpha, a.csv_line(it)))
    b = fresh(777)
    b.restore(**snap)
    for it in range(4, 7):
        b.iterate()
        w = want[it - 4]
        assert np.array_equal(b.arr("beta"), w[0]) and np.array_equal(b.arr("components"), w[1]) and np.array_equal(b.arr("eps"), w[2])
        assert b.mu == w[3] and b.alpha == w[4] and b.csv_line(it) == w[5]


GOLD = os.path.join(ROOT, "tests", "golden")


def test_ars_known_answers(orc_arms):
    """tests/golden/bayesw_small.npz holds 100 draws per density made on rand() after srand(7): the numbers the
    reference's arms() gives (test_oracle_ars_is_the_reference_ars proves that whenever oracle/_ref is built)."""
    gold = np.load(os.path.join(GOLD, "bayesw_small.npz"))
    for case in ("normal", "gamma"):
        dens, xinit, xl, xr = CASES[case]
        got = run_arms(orc_arms, dens, xinit, xl, xr, 100, 7)
        assert [g[1] for g in got] == list(gold[case + "_x"]) and [g[2] for g in got] == list(gold[case + "_neval"])
        assert [g[1] for g in run_product(dens, xinit, xl, xr, 100, 7)] == list(gold[case + "_x"])


def test_bayesw_chain_regression(oracle):
    gold = np.load(os.path.join(GOLD, "bayesw_small.npz"))
    ch = orc.BwChain(oracle, gold["bw_bed"], 90, gold["bw_y"], gold["bw_fail"], mS=np.array([[0.0, 0.001, 0.01]]), seed=1222, quad=7)
    for it in range(8):
        ch.iterate()
        assert np.array_equal(ch.arr("components"), gold["bw_comp"][it]) and np.array_equal(ch.arr("beta"), gold["bw_beta"][it])
        assert [ch.mu, ch.alpha, ch.arr("sigmaG")[0]] == list(gold["bw_hyper"][it]) and ch.csv_line(it) == str(gold["bw_csv"][it])
    assert np.array_equal(ch.arr("eps"), gold["bw_eps"])


def test_ars_matches_reference_on_random_densities(orc_arms, ref_arms):
    """Property: over random parameters of the four densities BayesW samples (shapes of mu_dens, alpha_dens,
    beta_dens, gamma_dens) the restated sampler and the reference's arms() agree bit for bit."""
    rng = np.random.default_rng(2024)
    for trial in range(120):
        kind = trial % 3
        if kind == 0:   # beta_dens: -a x d - exp(a x m/s)(v0 + v1 exp(-a x/s) + v2 exp(-2 a x/s)) - x^2/(2 C sG)
            a, d, m, s_ = rng.uniform(1, 12), rng.normal(0, 30), rng.uniform(0.05, 1.5), rng.uniform(0.2, 0.9)
            v0, v1, v2, cs = rng.uniform(50, 5000), rng.uniform(10, 3000), rng.uniform(0, 500), rng.choice([1e-4, 1e-3, 1e-2]) * rng.uniform(0.01, 0.5)
            dens = lambda x: -a * x * d - math.exp(a * x * m / s_) * (v0 + v1 * math.exp(-a * x / s_) + v2 * math.exp(-2 * a * x / s_)) - x * x / (2 * cs)
            b0, L_ = rng.normal(0, 0.01), 2 * math.sqrt(rng.uniform(0.01, 0.3) * cs / rng.uniform(0.01, 0.5))
            xinit, xl, xr = [b0 - L_ / 10, b0, b0 + L_ / 20, b0 + L_ / 10], b0 - L_, b0 + L_
        elif kind == 1:  # alpha_dens: (a0 + d - 1) log x + x (S - k0) - sum exp(eps x - g)
            eps = rng.normal(0, 0.3, size=40)
            dd, S = 30.0, float((eps * (rng.random(40) < 0.8)).sum())
            dens = lambda x: (0.01 + dd - 1) * math.log(x) + x * (S - 0.01) - float(np.exp(eps * x - 0.577215664901532).sum())
            a0 = rng.uniform(1, 8)
            xinit, xl, xr = [a0 * 0.5, a0, a0 * 1.05, a0 * 1.1], 0.0, a0 * 1.3
        else:            # mu_dens: -a x d - sum exp((eps - x) a - g) - x^2 / 200
            eps, a = rng.normal(3, 0.3, size=40), rng.uniform(1, 8)
            dens = lambda x: -a * x * 30.0 - float(np.exp((eps - x) * a - 0.577215664901532).sum()) - x * x / 200.0
            mu = 3.0 + rng.normal(0, 0.05)
            xinit, xl, xr = [0.95 * mu, mu, 1.005 * mu, 1.01 * mu], 0.8 * mu, 1.1 * mu
        seed = int(rng.integers(1, 2 ** 31))
        assert run_arms(ref_arms, dens, xinit, xl, xr, 5, seed) == run_arms(orc_arms, dens, xinit, xl, xr, 5, seed), (trial, kind)
