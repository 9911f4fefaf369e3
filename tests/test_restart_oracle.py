"""Checkpoint/restart, CPU side: the oracle's restatement of Boost's mt19937
stream form (src/distributions_boost.cpp:38-55) and of init_from_restart
(src/BayesRRm.cpp:842-928, :1546-1597).  Properties only -- the reference
holds no .rng/.eps fixture and Boost is not in this image."""
import ctypes as C

import numpy as np
import pytest

import orc
from hydra_amd import synth


def _words(L, g):
    w = np.zeros(624, dtype=np.uint32)
    L.orc_rng_print_words(C.byref(g), orc.p(w, C.c_uint32))
    return w


@pytest.mark.parametrize("consumed", [0, 1, 226, 227, 228, 623, 624, 625, 1247, 1248, 3001])
def test_rng_stream_form_continues_the_stream(oracle, consumed):
    g = orc.OrcMt()
    oracle.orc_rng_seed(C.byref(g), 4357)
    for _ in range(consumed):
        oracle.orc_rng_u32(C.byref(g))
    w = _words(oracle, g)
    if consumed == 0:  # a freshly seeded engine prints its seeded state (i == n: nothing to rewind)
        assert np.array_equal(w, np.array(g.x, dtype=np.uint32))
    g2 = orc.OrcMt()
    oracle.orc_rng_load_words(C.byref(g2), orc.p(w, C.c_uint32))
    assert g2.idx == 624
    assert [oracle.orc_rng_u32(C.byref(g)) for _ in range(1400)] == [oracle.orc_rng_u32(C.byref(g2)) for _ in range(1400)]
    # printing what was just read gives the same words back (print . read == id)
    g3 = orc.OrcMt()
    oracle.orc_rng_load_words(C.byref(g3), orc.p(w, C.c_uint32))
    assert np.array_equal(_words(oracle, g3), w)


def test_rng_stream_form_is_the_sliding_window(oracle):
    """The printed words at consumed = c+1 are the words at c shifted by one, for the part both hold."""
    g = orc.OrcMt()
    oracle.orc_rng_seed(C.byref(g), 99)
    for _ in range(700):
        oracle.orc_rng_u32(C.byref(g))
    prev = _words(oracle, g)
    for _ in range(5):
        oracle.orc_rng_u32(C.byref(g))
        cur = _words(oracle, g)
        # entry 0 of the window has only its top bit defined by the recurrence
        assert np.array_equal(cur[1:623], prev[2:624]) and (cur[0] >> 31) == (prev[1] >> 31)
        prev = cur


@pytest.mark.parametrize("with_cov", [False, True])
def test_oracle_chain_restore_is_exact(oracle, with_cov):
    M, N = 90, 301
    geno = synth.make_genotypes(M, N, seed=5, missing_rate=0.02)
    y, _ = synth.make_phenotype(geno, seed=6, causal_frac=0.1)
    bed = synth.pack_bed_columns(geno)
    groups = (np.arange(M) % 2).astype(np.int32)
    mS = np.array([[0.0, 0.001, 0.01], [0.0, 0.0001, 0.1]])
    X = np.random.default_rng(1).normal(size=(N, 2))
    a = orc.Chain(oracle, bed, N, y, groups=groups, mS=mS, seed=17)
    if with_cov:
        a.set_covariates(X)
    for _ in range(4):
        a.iterate()
    snap = dict(iteration=3, sigmaE=a.sigmaE, mu=a.mu, sigmaG=a.arr("sigmaG").copy(), estPi=a.arr("estPi").copy(),
                beta=a.arr("beta").copy(), components=a.arr("components").copy(), eps=a.arr("eps").copy(),
                order=a.arr("order").copy(), rng_words=a.rng_words())
    if with_cov:
        snap.update(gamma=a.gamma(), xI=a.xI())
    b = orc.Chain(oracle, bed, N, y, groups=groups, mS=mS, seed=999)  # a different seed: everything comes from the dump
    if with_cov:
        b.set_covariates(X)
    b.restore(**snap)
    for it in range(4, 7):
        a.iterate()
        b.iterate()
        for name in ("beta", "components", "eps", "sigmaG", "estPi", "order", "cass"):
            assert np.array_equal(a.arr(name), b.arr(name)), name
        assert a.sigmaE == b.sigmaE and a.mu == b.mu and a.csv_line(it) == b.csv_line(it)
        if with_cov:
            assert np.array_equal(a.gamma(), b.gamma())
