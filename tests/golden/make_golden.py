#!/usr/bin/env python3
"""Generate the committed fixtures under tests/golden/ (data only).

Sources, in order of authority:
  * reference data read as bytes / reference pieces compiled from their own
    sources by oracle/Makefile (needs /root/reference; run in the build container):
      - boost_ziggurat_tables.npz : the four Ziggurat tables inside the
        reference's prebuilt ELF (src/hydra, read as data at the file offsets
        SURVEY.md 8c lists; never executed)
      - dotp_lut.npz              : stdout of oracle/_ref/mk_lut (built from
        src/mk_lut.cpp), i.e. the content of src/dotp_lut.h
  * the CPU oracle itself (regression pins; "parity unpinned" w.r.t. the
    reference for the Boost-dependent arithmetic):
      - rng_kat.npz     : first draws of every distribution wrapper, seed 1222
      - chain_small.npz : N=64, M=32, K=4, 10 iterations (+ grouped G=2 variant)
      - dot_cases.npz   : (bed column, eps, mave, mstd) -> (s1, s2, num) cases
  * example/ : the input files of the reference's shipped example (data, not code)

Lives under tests/ because it drives the oracle: only tests/, smoke() and bench.py's
cpu_baseline leg may.
"""
import os
import re
import struct
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
GOLD = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"


def from_reference():
    elf = os.path.join(REF, "src", "hydra")
    if os.path.exists(elf):
        d = open(elf, "rb").read()

        def tab(off, n):
            return np.array(struct.unpack("<%dd" % n, d[off:off + 8 * n]))
        np.savez(os.path.join(GOLD, "boost_ziggurat_tables.npz"), normal_x=tab(0xD7FC0, 129), normal_y=tab(0xD83E0, 129),
                 exp_x=tab(0xD6F80, 257), exp_y=tab(0xD77A0, 257))
        print("boost_ziggurat_tables.npz from", elf)
    mk = os.path.join(ROOT, "oracle", "_ref", "mk_lut")
    if os.path.exists(mk):
        out = subprocess.check_output([mk]).decode()
        a_txt, b_txt = out.split("const double dotp_lut_b")
        num = re.compile(r"^\s+([0-9.]+),", re.M)
        a = np.array([float(x) for x in num.findall(a_txt)])
        b = np.array([float(x) for x in num.findall(b_txt)])
        assert a.size == 1024 and b.size == 1024
        np.savez(os.path.join(GOLD, "dotp_lut.npz"), lut_a=a, lut_b=b)
        print("dotp_lut.npz from", mk)


EXAMPLE_FILES = ["t_M10K_N_5K.fam", "t_M10K_N_5K.bim", "t_M10K_N_5K.dim", "normal.phen", "normal.group", "normal.mS", "normal.h2",
                 "Weibull.phen", "Weibull.fail", "Weibull.h2"]


def example_inputs():
    """The input files of the reference's shipped example (data, not code): sample and variant lists,
    the two phenotype files, the failure indicator, the group / mixture files of BASELINE configs 1, 3
    and 5.  The genotype file example/t_M10K_N_5K.bed itself is not in the checkout."""
    import shutil
    src = os.path.join(REF, "example")
    if not os.path.isdir(src):
        return
    dst = os.path.join(GOLD, "example")
    os.makedirs(dst, exist_ok=True)
    for f in EXAMPLE_FILES:
        shutil.copyfile(os.path.join(src, f), os.path.join(dst, f))
    print("example inputs copied to", dst)


def from_oracle():
    import ctypes as C
    import orc
    from hydra_amd import synth
    L = orc.load()
    g = orc.OrcMt()
    kat = {}
    L.orc_rng_seed(g, 1222)
    kat["u32"] = np.array([L.orc_rng_u32(g) for _ in range(16)], dtype=np.uint32)
    L.orc_rng_seed(g, 1222)
    kat["unif"] = np.array([L.orc_rng_unif(g) for _ in range(16)])
    L.orc_rng_seed(g, 1222)
    kat["norm"] = np.array([L.orc_rng_norm(g, 0.5, 2.0) for _ in range(4000)])
    L.orc_rng_seed(g, 1222)
    kat["exp"] = np.array([L.orc_rng_exp(g, 1.5) for _ in range(2000)])
    for name, shape in (("gamma_lt1", 0.3), ("gamma_eq1", 1.0), ("gamma_gt1", 7.25)):
        L.orc_rng_seed(g, 1222)
        kat[name] = np.array([L.orc_rng_gamma(g, shape, 2.0) for _ in range(500)])
    L.orc_rng_seed(g, 1222)
    kat["beta"] = np.array([L.orc_rng_beta(g, 1.0, 1.0) for _ in range(100)])
    L.orc_rng_seed(g, 1222)
    kat["inv_scaled_chisq"] = np.array([L.orc_rng_inv_scaled_chisq(g, 10.0001, 0.37) for _ in range(200)])
    L.orc_rng_seed(g, 1222)
    alpha = np.array([5.0, 1.0, 2.0, 9.0])
    out = np.zeros(4)
    dirs = []
    for _ in range(50):
        L.orc_rng_dirichlet(g, orc.dptr(alpha), 4, orc.dptr(out))
        dirs.append(out.copy())
    kat["dirichlet"] = np.array(dirs)
    L.orc_rng_seed(g, 1222)
    v = np.arange(40, dtype=np.int32)
    L.orc_rng_shuffle(g, orc.iptr(v), 40)
    kat["shuffle40"] = v.copy()
    np.savez(os.path.join(GOLD, "rng_kat.npz"), **kat)

    # dot / update cases, ragged N
    cases = {}
    rng = np.random.default_rng(5)
    for idx, N in enumerate((1, 3, 4, 5, 37, 64, 1023)):
        geno = rng.integers(0, 4, size=(1, N)).astype(np.uint8)
        geno[0, 0] = 0
        if N > 1:
            geno[0, 1] = 2
        col = synth.pack_bed_columns(geno)[0]
        eps = rng.normal(size=N)
        c = [C.c_uint64() for _ in range(4)]
        L.orc_bed_counts(orc.u8ptr(col), N, *[C.byref(x) for x in c])
        mave, mstd = C.c_double(), C.c_double()
        L.orc_marker_stats(c[1].value, c[2].value, c[3].value, max(N, 2), C.byref(mave), C.byref(mstd))
        s1, s2 = C.c_double(), C.c_double()
        dense = L.orc_dot_dense(orc.u8ptr(col), orc.dptr(eps), N, mave.value, mstd.value, C.byref(s1), C.byref(s2))
        sparse = L.orc_dot(orc.u8ptr(col), orc.dptr(eps), N, mave.value, mstd.value)
        eps2 = eps.copy()
        L.orc_update(orc.u8ptr(col), orc.dptr(eps2), N, mave.value, mstd.value, 0.0625)
        cases["c%d" % idx] = np.array([N], dtype=np.int64)
        cases["c%d_col" % idx] = col
        cases["c%d_eps" % idx] = eps
        cases["c%d_out" % idx] = np.array([c[1].value, c[2].value, c[3].value, mave.value, mstd.value, s1.value, s2.value,
                                           dense, sparse])
        cases["c%d_eps_updated" % idx] = eps2
    np.savez(os.path.join(GOLD, "dot_cases.npz"), **cases)

    # small chains
    out = {}
    for tag, G in (("g1", 1), ("g2", 2)):
        M, N = 32, 64
        geno = synth.make_genotypes(M, N, seed=11, missing_rate=0.03)
        y, _ = synth.make_phenotype(geno, seed=12, causal_frac=0.2)
        bed = synth.pack_bed_columns(geno)
        if G == 1:
            mS, groups = np.array([[0.0, 0.0001, 0.001, 0.01]]), None
        else:
            mS, groups = np.array([[0.0, 0.001, 0.01, 0.1], [0.0, 0.001, 0.01, 0.1]]), (np.arange(M) % 2).astype(np.int32)
        ch = orc.Chain(L, bed, N, y, groups=groups, mS=mS, seed=1222, shuffle=1)
        betas, comps, sGs, sEs, mus, pis, orders, csv = [], [], [], [], [], [], [], []
        for it in range(10):
            ch.iterate()
            betas.append(ch.arr("beta").copy())
            comps.append(ch.arr("components").copy())
            sGs.append(ch.arr("sigmaG").copy())
            sEs.append(ch.sigmaE)
            mus.append(ch.mu)
            pis.append(ch.arr("estPi").copy())
            orders.append(ch.arr("order").copy())
            csv.append(ch.csv_line(it))
        rx, ridx = ch.rng_state()
        out.update({tag + "_bed": bed, tag + "_y": y, tag + "_beta": np.array(betas), tag + "_comp": np.array(comps),
                    tag + "_sigmaG": np.array(sGs), tag + "_sigmaE": np.array(sEs), tag + "_mu": np.array(mus),
                    tag + "_pi": np.array(pis), tag + "_order": np.array(orders), tag + "_rng_x": rx,
                    tag + "_rng_idx": np.array([ridx]), tag + "_csv": np.array(csv), tag + "_eps": ch.arr("eps").copy()})
    np.savez(os.path.join(GOLD, "chain_small.npz"), **out)

    # BayesW: ARS known answers (the same numbers oracle/_ref/libarms.so gives; tests/test_bayesw_oracle.py checks
    # that live when the reference build is present) and one small chain
    import ctypes
    import math
    libc = ctypes.CDLL(None)
    orc._bind_bw(L)
    DENS = ctypes.CFUNCTYPE(ctypes.c_double, ctypes.c_double, ctypes.c_void_p)
    dp = ctypes.POINTER(ctypes.c_double)
    fn = L.orc_ars_arms_c
    fn.argtypes = [dp, ctypes.c_int, dp, dp, DENS, ctypes.c_void_p, dp, ctypes.c_int, ctypes.c_int, dp, dp, ctypes.c_int, dp, dp, ctypes.c_int,
                   ctypes.POINTER(ctypes.c_int)]
    fn.restype = ctypes.c_int
    ars = {}
    for name, dens, xinit, xl, xr in (("normal", lambda x: -0.5 * x * x, [-1.0, -0.2, 0.3, 1.1], -6.0, 6.0),
                                      ("gamma", lambda x: 2.5 * math.log(x) - 3 * x, [0.2, 0.6, 1.0, 2.0], 1e-3, 20.0)):
        libc.srand(7)
        cb, xs_, ne_ = DENS(lambda x, _d, f=dens: f(x)), [], []
        for _ in range(100):
            xi = (ctypes.c_double * 4)(*xinit)
            a, b, cv, xp = ctypes.c_double(xl), ctypes.c_double(xr), ctypes.c_double(1.0), ctypes.c_double(0.0)
            xs, qc, xc, ne = (ctypes.c_double * 1)(), (ctypes.c_double * 10)(5., 30., 70., 95.), (ctypes.c_double * 10)(), ctypes.c_int(0)
            assert fn(xi, 4, ctypes.byref(a), ctypes.byref(b), cb, None, ctypes.byref(cv), 100, 0, ctypes.byref(xp), xs, 1, qc, xc, 4,
                      ctypes.byref(ne)) == 0
            xs_.append(xs[0])
            ne_.append(ne.value)
        ars[name + "_x"], ars[name + "_neval"] = np.array(xs_), np.array(ne_)
    M, N = 40, 90
    geno = synth.make_genotypes(M, N, seed=21, missing_rate=0.03)
    y, fail, _ = synth.make_survival(geno, seed=22, causal_frac=0.2)
    bed = synth.pack_bed_columns(geno)
    ch = orc.BwChain(L, bed, N, y, fail, mS=np.array([[0.0, 0.001, 0.01]]), seed=1222, quad=7)
    betas, comps, hyp, csv = [], [], [], []
    for it in range(8):
        ch.iterate()
        betas.append(ch.arr("beta").copy())
        comps.append(ch.arr("components").copy())
        hyp.append([ch.mu, ch.alpha, ch.arr("sigmaG")[0]])
        csv.append(ch.csv_line(it))
    ars.update(bw_bed=bed, bw_y=y, bw_fail=fail, bw_beta=np.array(betas), bw_comp=np.array(comps), bw_hyper=np.array(hyp), bw_csv=np.array(csv),
               bw_eps=ch.arr("eps").copy())
    np.savez(os.path.join(GOLD, "bayesw_small.npz"), **ars)
    print("oracle fixtures written")


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    from_reference()
    example_inputs()
    from_oracle()
