"""Oracle decode / statistics / dot / update / chain against the reference's own
tables (dotp_lut.h via oracle/_ref and the committed fixture), the BED writer
of the reference (get_bed_marker_from_sparse, src/data.cpp:838-864), and
regression fixtures; plus the size-independent properties used at full size."""
import ctypes as C
import os

import numpy as np
import pytest

import orc
from hydra_amd import synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_decode_equals_reference_lut_fixture(oracle):
    gold = np.load(os.path.join(GOLD, "dotp_lut.npz"))
    v, m = np.zeros(4), np.zeros(4)
    for byte in range(256):
        oracle.orc_decode_byte(byte, orc.dptr(v), orc.dptr(m))
        assert np.array_equal(v * m, gold["lut_a"][4 * byte:4 * byte + 4] * gold["lut_b"][4 * byte:4 * byte + 4])
        assert np.array_equal(m, gold["lut_b"][4 * byte:4 * byte + 4])
        # dotp_lut_a alone maps missing to 0.0 as well (mk_lut.cpp:28-29)
        assert np.array_equal(v, gold["lut_a"][4 * byte:4 * byte + 4])


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libdotp_lut.so")), reason="oracle/_ref not built")
def test_decode_equals_compiled_reference_lut(oracle):
    R = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libdotp_lut.so"))
    R.ref_dotp_lut_a.restype = C.POINTER(C.c_double)
    R.ref_dotp_lut_b.restype = C.POINTER(C.c_double)
    a = np.ctypeslib.as_array(R.ref_dotp_lut_a(), shape=(1024,))
    b = np.ctypeslib.as_array(R.ref_dotp_lut_b(), shape=(1024,))
    gold = np.load(os.path.join(GOLD, "dotp_lut.npz"))
    assert np.array_equal(a, gold["lut_a"]) and np.array_equal(b, gold["lut_b"])
    v, m = np.zeros(4), np.zeros(4)
    for byte in range(256):
        oracle.orc_decode_byte(byte, orc.dptr(v), orc.dptr(m))
        assert np.array_equal(v, a[4 * byte:4 * byte + 4]) and np.array_equal(m, b[4 * byte:4 * byte + 4])


def test_bed_roundtrip_with_reference_writer_rule():
    """get_bed_marker_from_sparse (src/data.cpp:838-864): start from 0xFF, XOR
    0b01 / 0b11 / 0b10 shifted by 2*(i%4) for genotype 1 / 2 / missing."""
    rng = np.random.default_rng(0)
    N = 1003
    geno = rng.integers(0, 4, size=N).astype(np.uint8)
    nb = (N + 3) // 4
    raw = np.full(nb, 0xFF, dtype=np.uint8)
    for i, g in enumerate(geno):
        x = {0: 0, 1: 0b01, 2: 0b11, 3: 0b10}[int(g)]
        raw[i // 4] ^= x << (2 * (i % 4))
    assert np.array_equal(synth.unpack_bed_columns(raw[None, :], N)[0], geno)
    packed = synth.pack_bed_columns(geno[None, :])[0]
    # identical except for the unused tail slots of the last byte
    assert np.array_equal(packed[:-1], raw[:-1])
    assert np.array_equal(synth.unpack_bed_columns(packed[None, :], N)[0], geno)


def test_dot_cases_fixture(oracle):
    gold = np.load(os.path.join(GOLD, "dot_cases.npz"))
    idx = 0
    while "c%d" % idx in gold:
        N = int(gold["c%d" % idx][0])
        col, eps = np.ascontiguousarray(gold["c%d_col" % idx]), np.ascontiguousarray(gold["c%d_eps" % idx])
        n1, n2, nm, mave, mstd, s1, s2, dense, sparse = gold["c%d_out" % idx]
        c = [C.c_uint64() for _ in range(4)]
        oracle.orc_bed_counts(orc.u8ptr(col), N, *[C.byref(x) for x in c])
        assert (c[1].value, c[2].value, c[3].value) == (n1, n2, nm)
        assert c[0].value + c[1].value + c[2].value + c[3].value == N
        a, b = C.c_double(), C.c_double()
        got = oracle.orc_dot_dense(orc.u8ptr(col), orc.dptr(eps), N, mave, mstd, C.byref(a), C.byref(b))
        # (monomorphic columns give mstd = inf/nan: the reference does not guard, src/BayesRRm.cpp:1507)
        assert np.array_equal([a.value, b.value, got], [s1, s2, dense], equal_nan=True)
        assert np.array_equal([oracle.orc_dot(orc.u8ptr(col), orc.dptr(eps), N, mave, mstd)], [sparse], equal_nan=True)
        # the two algebraic forms of the reference agree to rounding
        if np.isfinite(mstd):
            assert abs(dense - sparse) <= 1e-12 * max(1.0, np.abs(eps).sum() * abs(mstd) * 2)
        e2 = eps.copy()
        oracle.orc_update(orc.u8ptr(col), orc.dptr(e2), N, mave, mstd, 0.0625)
        assert np.array_equal(e2, gold["c%d_eps_updated" % idx], equal_nan=True)
        idx += 1
    assert idx >= 7


def test_stats_against_numpy(oracle):
    M, N = 40, 777
    geno = synth.make_genotypes(M, N, seed=3, missing_rate=0.05)
    bed = synth.pack_bed_columns(geno)
    X = synth.standardize(geno)
    for j in range(M):
        c = [C.c_uint64() for _ in range(4)]
        oracle.orc_bed_counts(orc.u8ptr(np.ascontiguousarray(bed[j])), N, *[C.byref(x) for x in c])
        assert c[1].value == (geno[j] == 1).sum() and c[2].value == (geno[j] == 2).sum() and c[3].value == (geno[j] == 3).sum()
        a, s = C.c_double(), C.c_double()
        oracle.orc_marker_stats(c[1].value, c[2].value, c[3].value, N, C.byref(a), C.byref(s))
        # x_j'x_j == N-1 by construction of mstd (src/BayesRRm.cpp:1503-1507; used as dNm1 at :1750,:1855)
        assert abs((X[:, j] ** 2).sum() - (N - 1)) < 1e-8
        eps = np.random.default_rng(j).normal(size=N)
        ref = X[:, j] @ eps
        got = oracle.orc_dot(orc.u8ptr(np.ascontiguousarray(bed[j])), orc.dptr(eps), N, a.value, s.value)
        assert abs(got - ref) < 1e-9 * max(1.0, abs(ref))
        e2 = eps.copy()
        oracle.orc_update(orc.u8ptr(np.ascontiguousarray(bed[j])), orc.dptr(e2), N, a.value, s.value, 0.3)
        assert np.allclose(e2, eps + 0.3 * X[:, j], rtol=0, atol=1e-12)  # eps - (b_new - b_old) x_j


def test_na_compaction(oracle):
    rng = np.random.default_rng(1)
    N = 101
    geno = rng.integers(0, 4, size=(1, N)).astype(np.uint8)
    keep = (rng.random(N) > 0.2).astype(np.uint8)
    col = synth.pack_bed_columns(geno)[0]
    out = np.zeros((N + 3) // 4, dtype=np.uint8)
    n = C.c_uint32()
    oracle.orc_bed_compact(orc.u8ptr(col), N, orc.u8ptr(keep), orc.u8ptr(out), C.byref(n))
    assert n.value == keep.sum()
    assert np.array_equal(synth.unpack_bed_columns(out[None, :(n.value + 3) // 4], n.value)[0], geno[0][keep == 1])


@pytest.mark.parametrize("tag,G", [("g1", 1), ("g2", 2)])
def test_chain_regression_fixture(oracle, tag, G):
    gold = np.load(os.path.join(GOLD, "chain_small.npz"))
    bed, y = gold[tag + "_bed"], gold[tag + "_y"]
    M, N = bed.shape[0], 64
    if G == 1:
        mS, groups = np.array([[0.0, 0.0001, 0.001, 0.01]]), None
    else:
        mS, groups = np.array([[0.0, 0.001, 0.01, 0.1], [0.0, 0.001, 0.01, 0.1]]), (np.arange(M) % 2).astype(np.int32)
    ch = orc.Chain(oracle, bed, N, y, groups=groups, mS=mS, seed=1222, shuffle=1)
    for it in range(10):
        ch.iterate()
        assert np.array_equal(ch.arr("order"), gold[tag + "_order"][it])
        assert np.array_equal(ch.arr("components"), gold[tag + "_comp"][it])
        assert np.array_equal(ch.arr("beta"), gold[tag + "_beta"][it])
        assert np.array_equal(ch.arr("sigmaG"), gold[tag + "_sigmaG"][it])
        assert ch.sigmaE == gold[tag + "_sigmaE"][it] and ch.mu == gold[tag + "_mu"][it]
        assert ch.csv_line(it) == str(gold[tag + "_csv"][it])
    rx, ridx = ch.rng_state()
    assert np.array_equal(rx, gold[tag + "_rng_x"]) and ridx == int(gold[tag + "_rng_idx"][0])


def test_csv_line_layout(oracle):
    """a11: "%5d, %4d" + G x ", %20.15f" + ", %20.15f, %20.15f, %7d, %4d, %2d" + G*K x ", %20.15f" + newline;
    constant length so that offset = n * strlen works (src/BayesRRm.cpp:2742-2764)."""
    gold = np.load(os.path.join(GOLD, "chain_small.npz"))
    for tag, G, K in (("g1", 1, 4), ("g2", 2, 4)):
        lines = [str(x) for x in gold[tag + "_csv"]]
        want = 5 + 2 + 4 + G * 22 + (22 + 22 + 2 + 7 + 2 + 4 + 2 + 2) + G * K * 22 + 1
        assert {len(l) for l in lines} == {want}
        f = lines[3].strip().split(",")
        assert int(f[0]) == 3 and int(f[1]) == G and int(f[-G * K - 2]) == G and int(f[-G * K - 1]) == K
        pi = np.array([float(x) for x in f[-G * K:]]).reshape(G, K)
        assert np.allclose(pi.sum(axis=1), 1.0, atol=1e-12)


def test_chain_properties_hold(oracle):
    """Size-independent invariants (also asserted on the GPU at full size):
    eps == y - mu - X beta after every iteration; cass rows sum to the group
    sizes; m0 = markers outside component 0."""
    M, N = 120, 400
    geno = synth.make_genotypes(M, N, seed=8, missing_rate=0.02)
    y, _ = synth.make_phenotype(geno, seed=9, causal_frac=0.05)
    bed = synth.pack_bed_columns(geno)
    X = synth.standardize(geno)
    ch = orc.Chain(oracle, bed, N, y, seed=77)
    ys = ch.arr("y").copy()
    for _ in range(5):
        ch.iterate()
        resid = ys - ch.mu - X @ ch.arr("beta")
        assert np.allclose(ch.arr("eps"), resid, rtol=0, atol=1e-9)
        assert ch.arr("cass").sum() == M
        assert ch.arr("m0")[0] == (ch.arr("components") != 0).sum()
        assert np.all((ch.arr("beta") != 0) == (ch.arr("components") != 0))


def test_recovers_simulated_heritability(oracle):
    """Statistical truth in the spirit of example/normal.h2: on data simulated
    with h2 = 0.5 the posterior mean of sigmaG/(sigmaG+sigmaE) lands near 0.5."""
    M, N = 600, 1500
    geno = synth.make_genotypes(M, N, seed=21)
    y, _ = synth.make_phenotype(geno, seed=22, h2=0.5, causal_frac=0.05)
    ch = orc.Chain(oracle, synth.pack_bed_columns(geno), N, y, seed=1222)
    h2 = []
    for it in range(150):
        ch.iterate()
        if it >= 50:
            sg = ch.arr("sigmaG").sum()
            h2.append(sg / (sg + ch.sigmaE))
    assert 0.35 < np.mean(h2) < 0.65
