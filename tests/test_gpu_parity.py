"""GPU parity tests proper: the HIP path, called through the C ABI
(include/hgibbs.h), against the CPU oracle on the same seeded inputs.

Bar: bit-exact for integer work (counts, mixture-component indices, cass,
RNG state) and for the residual update (same three constants, one add);
|beta - beta_ref| <= 1e-9 * max(1, |beta_ref|) and hyper-parameters to rel
1e-9 for floating point (the device sums in a different,
fixed order than the oracle's sequential loop).
"""
import ctypes as C

import numpy as np
import pytest

import orc
from hydra_amd import capi, synth

pytestmark = pytest.mark.gpu

BETA_TOL = 1e-9


def close(a, b, tol=BETA_TOL):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.all(np.abs(a - b) <= tol * np.maximum(1.0, np.abs(b)))


def make_case(M, N, seed=7, missing_rate=0.02):
    geno = synth.make_genotypes(M, N, seed=seed, missing_rate=missing_rate)
    y, _ = synth.make_phenotype(geno, seed=seed + 1, h2=0.5, causal_frac=0.05)
    return synth.pack_bed_columns(geno), y


@pytest.mark.parametrize("N", [37, 1024, 4099, 9001])
def test_marker_stats_bit_exact(oracle, N):
    M = 25
    bed, _ = make_case(M, N, seed=N)
    dev = capi.Device(0)
    dev.load_bed(bed, N)
    mave, mstd, n1, n2, nm = dev.marker_stats()
    for j in range(M):
        c = [C.c_uint64() for _ in range(4)]
        col = np.ascontiguousarray(bed[j])
        oracle.orc_bed_counts(orc.u8ptr(col), N, *[C.byref(x) for x in c])
        assert (n1[j], n2[j], nm[j]) == (c[1].value, c[2].value, c[3].value)
        a, s = C.c_double(), C.c_double()
        oracle.orc_marker_stats(c[1].value, c[2].value, c[3].value, N, C.byref(a), C.byref(s))
        assert mave[j] == a.value and mstd[j] == s.value
    # the packed shard comes back unchanged
    assert np.array_equal(dev.get_bed(), bed)


@pytest.mark.parametrize("N", [37, 4099])
def test_residual_roundtrip_and_reductions(N):
    bed, y = make_case(5, N, seed=3)
    dev = capi.Device(0)
    dev.load_bed(bed, N)
    dev.set_residual(y)
    assert np.array_equal(dev.get_residual(), y)
    s, q = dev.reduce_eps()
    assert abs(s - y.sum()) <= 1e-12 * max(1.0, np.abs(y).sum())
    assert abs(q - (y * y).sum()) <= 1e-12 * (y * y).sum()
    dev.add_scalar(0.375)
    assert np.array_equal(dev.get_residual(), y + 0.375)
    # padding slots stay zero: the sum only moved by N * c
    s2, _ = dev.reduce_eps()
    assert abs(s2 - (y + 0.375).sum()) <= 1e-12 * max(1.0, np.abs(y + 0.375).sum())


@pytest.mark.parametrize("N", [37, 1023, 4099])
def test_dot_and_update_single_marker(oracle, N):
    M = 12
    bed, y = make_case(M, N, seed=11 + N, missing_rate=0.05)
    dev = capi.Device(0)
    dev.load_bed(bed, N)
    mave, mstd, *_ = dev.marker_stats()
    eps = y.copy()
    dev.set_residual(eps)
    rng = np.random.default_rng(5)
    for j in range(M):
        col = np.ascontiguousarray(bed[j])
        ref = oracle.orc_dot(orc.u8ptr(col), orc.dptr(eps), N, mave[j], mstd[j])
        got = dev.dot_marker(j)
        scale = np.abs(eps).sum() * mstd[j] * 2
        assert abs(got - ref) <= 1e-13 * scale
        db = float(rng.normal()) * 0.01
        oracle.orc_update(orc.u8ptr(col), orc.dptr(eps), N, mave[j], mstd[j], db)
        dev.update_marker(j, db)
        assert np.array_equal(dev.get_residual(), eps)  # bit-exact: same constants, one add


def _gpu_sweep_vs_oracle(oracle, M, N, G, mS, groups, batch, iters=3, missing_rate=0.02, seed=1222, cpg=None):
    bed, y = make_case(M, N, seed=M + N, missing_rate=missing_rate)
    ref = orc.Chain(oracle, bed, N, y, groups=groups, mS=mS, seed=seed, shuffle=1)
    dev = capi.Device(0)
    dev.load_bed(bed, N)
    dev.set_option("batch", batch)
    if cpg:
        dev.set_option("cols_per_group", cpg)
    ch = capi.Chain(dev, y, mS=mS, groups=groups, seed=seed, shuffle=1)
    for it in range(iters):
        ref.iterate()
        ch.iterate()
        beta, comp, acum = dev.get_beta()
        st = ch.state()
        assert np.array_equal(ch.order(), ref.arr("order")), "marker order diverged at it %d" % it
        assert np.array_equal(comp, ref.arr("components")), "component indices differ at it %d" % it
        assert np.array_equal(st["cass"].ravel(), ref.arr("cass"))
        assert close(beta, ref.arr("beta")), "beta beyond tolerance at it %d" % it
        assert close(acum, ref.arr("acum"))
        assert close(st["sigmaG"], ref.arr("sigmaG")) and close(st["estPi"].ravel(), ref.arr("estPi"))
        assert close(st["sigmaE"], ref.sigmaE) and close(st["mu"], ref.mu)
        rx, ridx = ref.rng_state()
        assert st["rng_idx"] % 624 == ridx % 624 and np.array_equal(st["rng_x"], rx) or \
            _same_stream(st["rng_x"], st["rng_idx"], rx, ridx)
        assert close(dev.get_residual(), ref.arr("eps"), 1e-9)
        assert ch.last_nnz() == oracle.orc_chain_last_nnz(ref.h)
    return ch, ref


def _same_stream(xa, ia, xb, ib):
    """Two MT19937 states are the same generator iff their next outputs agree
    (idx 624 with the old block == idx 0 with the twisted block)."""
    L = orc.load()
    ga, gb = orc.OrcMt(), orc.OrcMt()
    for g, x, i in ((ga, xa, ia), (gb, xb, ib)):
        for k in range(624):
            g.x[k] = int(x[k])
        g.idx = int(i)
    return all(L.orc_rng_u32(ga) == L.orc_rng_u32(gb) for _ in range(1300))


@pytest.mark.parametrize("batch", [1, 7, 32, 64, 200, 256])
def test_chain_vs_oracle_small(oracle, batch):
    _gpu_sweep_vs_oracle(oracle, M=300, N=517, G=1, mS=np.array([[0.0, 0.0001, 0.001, 0.01]]), groups=None,
                         batch=batch, iters=4)


def test_chain_vs_oracle_dense_form(oracle):
    """Same chain against the oracle run with the reference's dense LUT algebra
    (src/BayesRRm.cpp:1766-1809), the form the HIP kernel accumulates in."""
    oracle.orc_set_dot_form(1)
    try:
        _gpu_sweep_vs_oracle(oracle, M=300, N=2100, G=1, mS=np.array([[0.0, 0.0001, 0.001, 0.01]]), groups=None,
                             batch=48, iters=4, missing_rate=0.03)
    finally:
        oracle.orc_set_dot_form(0)


@pytest.mark.parametrize("cpg", [2, 4, 8, 16])
def test_chain_vs_oracle_groups(oracle, cpg):
    """Grouped mixture (BASELINE config 3's shape), once per build of the sweep kernel (columns per workgroup)."""
    M = 400
    groups = (np.arange(M) % 2).astype(np.int32)
    mS = np.array([[0.0, 0.001, 0.01, 0.1], [0.0, 0.001, 0.01, 0.1]])
    _gpu_sweep_vs_oracle(oracle, M=M, N=4099, G=2, mS=mS, groups=groups, batch=32, iters=5, cpg=cpg)


def test_chain_vs_oracle_long(oracle):
    # crosses several MT19937 block boundaries on the device and on the host
    _gpu_sweep_vs_oracle(oracle, M=1500, N=2048, G=1, mS=np.array([[0.0, 0.0001, 0.001, 0.01]]), groups=None,
                         batch=32, iters=10, missing_rate=0.0)


@pytest.mark.parametrize("missing_rate", [0.0, 0.02])
def test_batch_width_does_not_change_the_chain(missing_rate):
    """Plain speculative batching is exact: with the Gram-corrected extension off
    every batch width gives bit-identical output (each accepted marker saw the
    residual the sequential loop would have given it)."""
    bed, y = make_case(500, 3000, seed=99, missing_rate=missing_rate)
    outs = []
    for batch in (1, 16, 64, 256):
        dev = capi.Device(0)
        dev.load_bed(bed, 3000)
        dev.set_option("batch", batch)
        dev.set_option("gram", 0)
        ch = capi.Chain(dev, y, seed=5)
        for _ in range(3):
            ch.iterate()
        outs.append((dev.get_beta(), dev.get_residual(), ch.state()))
    for o in outs[1:]:
        for a, b in zip(o[0], outs[0][0]):
            assert np.array_equal(a, b)
        assert np.array_equal(o[1], outs[0][1])
        assert o[2]["sigmaE"] == outs[0][2]["sigmaE"] and np.array_equal(o[2]["rng_x"], outs[0][2]["rng_x"])


@pytest.mark.parametrize("batch", [2, 9, 64, 256])
def test_gram_extension_matches_oracle_and_plain_path(oracle, batch):
    """Past the first predicted event a launch continues with dots corrected by
    dbeta * x_j'x_pivot (integer Gram term): same chain as the plain path up to
    fp rounding -- components exact, beta / residual to 1e-9 -- and as the oracle."""
    M, N = 600, 2600
    bed, y = make_case(M, N, seed=123, missing_rate=0.0)
    ref = orc.Chain(oracle, bed, N, y, seed=4)
    devs = []
    for gram in (1, 0):
        dev = capi.Device(0)
        dev.load_bed(bed, N)
        dev.set_option("batch", batch)
        dev.set_option("gram", gram)
        devs.append((dev, capi.Chain(dev, y, seed=4)))
    fewer = False
    for it in range(6):
        ref.iterate()
        for dev, ch in devs:
            ch.iterate()
        (bg, cg, ag), (bp, cp, ap) = devs[0][0].get_beta(), devs[1][0].get_beta()
        assert np.array_equal(cg, cp) and np.array_equal(cg, ref.arr("components")), "it %d" % it
        assert close(bg, bp) and close(bg, ref.arr("beta")) and close(ag, ref.arr("acum"))
        assert close(devs[0][0].get_residual(), ref.arr("eps"))
        sg, sp = devs[0][1].state(), devs[1][1].state()
        assert close(sg["sigmaE"], ref.sigmaE) and np.array_equal(sg["rng_x"], sp["rng_x"]) and sg["rng_idx"] == sp["rng_idx"]
        fewer = fewer or devs[0][0].sweep_stats()["launches"] < devs[1][0].sweep_stats()["launches"]
    if batch >= 9:
        assert fewer, "the extension never saved a launch: it is not being exercised"


@pytest.mark.parametrize("col_frac,cpg", [(0.2, 4), (1.0, 4), (1.0, 8)])
def test_missing_calls_ride_through_the_extension(oracle, col_frac, cpg):
    """Columns with missing calls inside the Gram-corrected extension (the four-term build, option gram_missing):
    same chain as the oracle and as the build that ends the extension at such columns, in fewer launches."""
    M, N = 1200, 3000
    rng = np.random.default_rng(9)
    geno = synth.make_genotypes(M, N, seed=29, missing_rate=0.0)
    for c in rng.choice(M, size=int(col_frac * M), replace=False):
        geno[c, rng.random(N) < 0.02] = 3
    y, _ = synth.make_phenotype(geno, seed=30, causal_frac=0.03)
    bed = synth.pack_bed_columns(geno)
    ref = orc.Chain(oracle, bed, N, y, seed=8)
    launches = {}
    for mg in (0, 1):
        dev = capi.Device(0)
        dev.load_bed(bed, N)
        dev.set_option("max_seg", 2)
        dev.set_option("batch", 256)
        dev.set_option("cols_per_group", cpg)
        dev.set_option("gram_missing", mg)
        ch = capi.Chain(dev, y, seed=8)
        tot = 0
        for it in range(6):
            if mg:
                ref.iterate()
            ch.iterate()
            tot += dev.sweep_stats()["launches"]
            if mg:
                beta, comp, _ = dev.get_beta()
                assert np.array_equal(comp, ref.arr("components")) and close(beta, ref.arr("beta")) and close(dev.get_residual(), ref.arr("eps"))
        launches[mg] = tot
    assert launches[1] < launches[0]


@pytest.mark.parametrize("max_seg,cpg", [(2, 4), (4, 4), (2, 8), (4, 8)])
def test_carried_dots_match_the_streamed_ones(oracle, max_seg, cpg):
    """Columns that lay behind the event that ended a launch are not streamed again: the next launch corrects their
    dots by dbeta * x_j'x_event from an integer Gram term (option carry).  Same chain as with the option off (same
    launches, components, generator; floating point to 1e-9) and as the oracle; a column with missing calls ends the carry."""
    M, N = 1500, 9000
    geno = synth.make_genotypes(M, N, seed=41, missing_rate=0.0)
    rng = np.random.default_rng(3)
    for c in rng.choice(M, size=M // 25, replace=False):
        geno[c, rng.random(N) < 0.01] = 3
    y, _ = synth.make_phenotype(geno, seed=42, causal_frac=0.04)
    bed = synth.pack_bed_columns(geno)
    ref = orc.Chain(oracle, bed, N, y, seed=11)
    runs = []
    for carry in (1, 0):
        dev = capi.Device(0)
        dev.load_bed(bed, N)
        dev.set_option("batch", 256)
        dev.set_option("max_seg", max_seg)
        dev.set_option("cols_per_group", cpg)
        dev.set_option("gram_missing", 0)
        dev.set_option("carry", carry)
        runs.append((dev, capi.Chain(dev, y, seed=11)))
    carried = 0
    for it in range(6):
        ref.iterate()
        for dev, ch in runs:
            ch.iterate()
            beta, comp, acum = dev.get_beta()
            assert np.array_equal(comp, ref.arr("components")), "it %d" % it
            assert close(beta, ref.arr("beta")) and close(acum, ref.arr("acum")) and close(dev.get_residual(), ref.arr("eps"))
        s1, s0 = runs[0][0].sweep_stats(), runs[1][0].sweep_stats()
        assert s1["launches"] == s0["launches"] and s0["carried_columns"] == 0
        assert np.array_equal(runs[0][1].state()["rng_x"], runs[1][1].state()["rng_x"])
        carried += s1["carried_columns"]
    assert carried > M, "the carry was hardly exercised: %d columns" % carried


def test_graph_replay_is_the_same_chain():
    """The sweep's launches replayed from a captured HIP graph (option "graph") walk the very same chain."""
    M, N = 900, 3000
    bed, y = make_case(M, N, seed=23)
    out = []
    for graph in (0, 1):
        dev = capi.Device(0)
        dev.load_bed(bed, N)
        dev.set_option("graph", graph)
        ch = capi.Chain(dev, y, seed=3)
        for _ in range(4):
            ch.iterate()
        beta, comp, _ = dev.get_beta()
        out.append((beta, comp, dev.get_residual(), ch.state()["rng_x"], dev.sweep_stats()["launches"]))
    assert all(np.array_equal(a, b) for a, b in zip(out[0][:4], out[1][:4]))
    assert out[1][4] >= out[0][4]  # replay rounds the launch count up to whole graphs; the extra launches find nothing to do


@pytest.mark.parametrize("max_seg,cpg", [(3, 8), (4, 8), (4, 4)])
def test_chained_segments_match_the_oracle(oracle, max_seg, cpg):
    """Up to four segments per launch (the wider kernel tier: three Gram terms per column, four pending
    updates): same chain as the oracle, fewer launches than the two-segment default once effects are
    non-zero (predicted events to chain through)."""
    M, N = 1500, 2048
    bed, y = make_case(M, N, seed=19, missing_rate=0.0)
    ref = orc.Chain(oracle, bed, N, y, seed=5)
    launches = {}
    for ms in (2, max_seg):
        dev = capi.Device(0)
        dev.load_bed(bed, N)
        dev.set_option("batch", 256)
        dev.set_option("cols_per_group", cpg)
        dev.set_option("max_seg", ms)
        ch = capi.Chain(dev, y, seed=5)
        tot = 0
        for it in range(6):
            if ms == max_seg:
                ref.iterate()
            ch.iterate()
            tot += dev.sweep_stats()["launches"]
            if ms == max_seg:
                beta, comp, _ = dev.get_beta()
                assert np.array_equal(comp, ref.arr("components")) and close(beta, ref.arr("beta"))
                assert close(dev.get_residual(), ref.arr("eps"))
        launches[ms] = tot
    assert launches[max_seg] < launches[2]


@pytest.mark.parametrize("with_comm", [False, True])
def test_split_path_equals_fused_path(with_comm):
    """The multi-GPU structure (local sums -> all-reduce -> replicated draw, three
    launches per batch) run on one rank -- with a 1-rank RCCL communicator so that
    the ncclAllReduce calls execute -- gives the fused kernel's chain bit for bit."""
    bed, y = make_case(400, 2500, seed=17)
    outs = []
    for split in (False, True):
        dev = capi.Device(0)
        if split and with_comm:
            dev.comm_init(1, 0, capi.Device.unique_id())
        dev.load_bed(bed, 2500)
        dev.set_option("batch", 48)
        dev.set_option("force_split", 1 if split else 0)
        dev.set_option("ahead", 0)  # the split path streams nothing ahead: bit-identity holds between the same launch plans
        ch = capi.Chain(dev, y, seed=9)
        for _ in range(3):
            ch.iterate()
        outs.append((dev.get_beta(), dev.get_residual(), ch.state()))
        dev.close()
    for a, b in zip(outs[0][0], outs[1][0]):
        assert np.array_equal(a, b)
    assert np.array_equal(outs[0][1], outs[1][1])
    assert outs[0][2]["sigmaE"] == outs[1][2]["sigmaE"] and np.array_equal(outs[0][2]["rng_x"], outs[1][2]["rng_x"])


def test_na_phenotype_rows_are_dropped(oracle):
    M, N = 60, 1000
    bed, y = make_case(M, N, seed=21)
    keep = np.ones(N, dtype=np.uint8)
    keep[[0, 5, 6, 7, 500, 999]] = 0
    dev = capi.Device(0)
    dev.load_bed(bed, N, keep=keep)
    nk = int(keep.sum())
    assert dev.n_local == nk
    got = dev.get_bed()
    for j in range(M):
        out = np.zeros((nk + 3) // 4, dtype=np.uint8)
        n_out = C.c_uint32()
        oracle.orc_bed_compact(orc.u8ptr(np.ascontiguousarray(bed[j])), N, orc.u8ptr(keep), orc.u8ptr(out), C.byref(n_out))
        assert n_out.value == nk and np.array_equal(got[j], out)


def test_synth_bed_matches_host_hash():
    from hydra_amd.synth import synth_bed_reference
    N, M = 1003, 17
    dev = capi.Device(0)
    dev.synth_bed(N, M, seed=42, missing_rate=0.01)
    assert np.array_equal(dev.get_bed(), synth_bed_reference(N, M, seed=42, missing_rate=0.01))
    # sharding by rows reproduces the same matrix
    dev2 = capi.Device(0)
    dev2.synth_bed(N, M, seed=42, missing_rate=0.01, row_begin=500, row_end=1003)
    full = synth.unpack_bed_columns(dev.get_bed(), N)
    part = synth.unpack_bed_columns(dev2.get_bed(), 503)
    assert np.array_equal(part, full[:, 500:])


# ---------------------------------------------------------------------------
# the CLI: hydra's flags in, hydra's files out (src/BayesRRm.cpp:2736-2794)
# ---------------------------------------------------------------------------
import os
import struct
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "hydra_amd", "bin", "hydra_mi355x")


def _read_bet(path, M, dtype):
    raw = open(path, "rb").read()
    assert struct.unpack("<I", raw[:4])[0] == M  # first word = Mtot (postproc/beta_converter.cpp:33-46)
    item = np.dtype(dtype).itemsize
    rec = 4 + M * item
    assert (len(raw) - 4) % rec == 0
    its, vals = [], []
    for k in range((len(raw) - 4) // rec):
        off = 4 + k * rec
        its.append(struct.unpack("<I", raw[off:off + 4])[0])
        vals.append(np.frombuffer(raw[off + 4:off + rec], dtype=dtype))
    return its, np.array(vals)


@pytest.mark.parametrize("grouped", [False, True])
def test_cli_end_to_end_files_match_oracle(oracle, tmp_path, grouped):
    M, N, iters = 120, 403, 6
    geno = synth.make_genotypes(M, N, seed=31, missing_rate=0.02)
    y, _ = synth.make_phenotype(geno, seed=32, causal_frac=0.05)
    bed = synth.pack_bed_columns(geno)
    na_rows = [3, 100, 101, 402]
    prefix = str(tmp_path / "data")
    synth.write_plink(prefix, bed, N, y=y, na_rows=na_rows)
    out = str(tmp_path / "out")
    cmd = [EXE, "--mpibayes", "bayesMPI", "--bfile", prefix, "--pheno", prefix + ".phen", "--mcmc-out-dir", out,
           "--mcmc-out-name", "run", "--number-individuals", str(N), "--number-markers", str(M), "--chain-length", str(iters),
           "--thin", "1", "--save", "2", "--seed", "1222", "--shuf-mark", "1"]
    if grouped:
        groups = (np.arange(M) % 2).astype(np.int32)
        np.savetxt(prefix + ".group", groups, fmt="%d")
        open(prefix + ".mS", "w").write("0.001,0.01,0.1;0.001,0.01,0.1\n")
        cmd += ["--groupIndexFile", prefix + ".group", "--groupMixtureFile", prefix + ".mS"]
        mS = np.array([[0.0, 0.001, 0.01, 0.1], [0.0, 0.001, 0.01, 0.1]])
    else:
        cmd += ["--S", "0.0001,0.001,0.01"]
        groups, mS = None, np.array([[0.0, 0.0001, 0.001, 0.01]])
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "RESULT : it    0, rank    0: proc =" in r.stdout

    keep = np.ones(N, dtype=np.uint8)
    keep[na_rows] = 0
    geno_k = geno[:, keep == 1]
    ref = orc.Chain(oracle, synth.pack_bed_columns(geno_k), int(keep.sum()), y[keep == 1], groups=groups, mS=mS, seed=1222)
    its, betas = _read_bet(out + "/run.bet", M, np.float64)
    _, comps = _read_bet(out + "/run.cpn", M, np.int32)
    _, acus = _read_bet(out + "/run.acu", M, np.float64)
    csv = open(out + "/run.csv").read().splitlines()
    assert its == list(range(iters)) and len(csv) == iters
    mus = open(out + "/run.mus.0", "rb").read()
    assert len(mus) == iters * 12
    for it in range(iters):
        ref.iterate()
        assert np.array_equal(comps[it], ref.arr("components"))
        assert close(betas[it], ref.arr("beta")) and close(acus[it], ref.arr("acum"))
        got = [float(x) for x in csv[it].split(",")]
        want = [float(x) for x in ref.csv_line(it).split(",")]
        assert len(csv[it]) + 1 == len(ref.csv_line(it)) and close(got, want, 1e-9)
        k, mu = struct.unpack("<Id", mus[12 * it:12 * it + 12])
        assert k == it and close(mu, ref.mu)
    # --save 2: last checkpoint taken at iteration 4
    eps_raw = open(out + "/run.eps.0", "rb").read()
    it_s, n_s = struct.unpack("<II", eps_raw[:8])
    assert (it_s, n_s) == (4, int(keep.sum())) and len(eps_raw) == 8 + 8 * n_s
    xb = open(out + "/run.xbet", "rb").read()
    assert struct.unpack("<II", xb[:8]) == (M, 4) and close(np.frombuffer(xb[8:], dtype=np.float64), betas[4])


def test_cli_option_file_reads_S_as_float(oracle, tmp_path):
    """--inp-file: the mixture variances of the option file go through stof (src/options.cpp:380-386), so the chain runs with
    float32(0.0001) = 9.99999974737875e-05 and not with the double the command line's --S gives (stod, src/options.cpp:233)."""
    M, N, iters = 90, 300, 4
    geno = synth.make_genotypes(M, N, seed=41, missing_rate=0.0)
    y, _ = synth.make_phenotype(geno, seed=42, causal_frac=0.1)
    bed = synth.pack_bed_columns(geno)
    prefix = str(tmp_path / "data")
    synth.write_plink(prefix, bed, N, y=y, na_rows=[])
    (tmp_path / "out").mkdir()
    opt = tmp_path / "run.opt"
    opt.write_text("analysisType RAM\nbayesType bayesMPI\nbedFile %s\nphenotypeFile %s.phen\nmcmcOut %s/out/run\n"
                   "numberIndividuals %d\nnumberMarkers %d\nchainLength %d\nthin 1\nsave 100\nseed 1222\nshuffleMarkers 1\n"
                   "S 0.0001,0.001,0.01\n" % (prefix, prefix, tmp_path, N, M, iters))
    r = subprocess.run([EXE, "--inp-file", str(opt)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    s32 = np.float32([0.0001, 0.001, 0.01]).astype(np.float64)
    assert s32[0] != 0.0001
    _, betas = _read_bet(str(tmp_path / "out" / "run.bet"), M, np.float64)
    _, comps = _read_bet(str(tmp_path / "out" / "run.cpn"), M, np.int32)
    as_float = orc.Chain(oracle, bed, N, y, mS=np.array([[0.0, *s32]]), seed=1222)
    as_double = orc.Chain(oracle, bed, N, y, mS=np.array([[0.0, 0.0001, 0.001, 0.01]]), seed=1222)
    for it in range(iters):
        as_float.iterate()
        as_double.iterate()
        assert np.array_equal(comps[it], as_float.arr("components")) and close(betas[it], as_float.arr("beta"), 1e-12)
    # the two readings give different chains: the test would not pass with stod in the option file
    assert np.max(np.abs(as_float.arr("beta") - as_double.arr("beta"))) > 1e-9


def test_full_size_properties_c2():
    """BASELINE config 2 (N=50 000, M=100 000) through size-independent
    properties: (1) residual identity eps + X beta == y - mu after the sweep
    (round trip through the update operator), (2) counts add up, (3) neither the batch
    width nor the engine (batch / resident) changes the chain, (4) the held sum of eps drifts by roundings only."""
    N, M = 50000, 100000
    res = []
    for batch in (64, 17, 0):
        dev = capi.Device(0)
        if batch:
            dev.set_option("batch", batch)  # pins the batch engine; the last pass runs the default, the resident engine
        dev.synth_bed(N, M, seed=42)
        rng = np.random.default_rng(1)
        y = rng.normal(size=N)
        dev.set_residual(y)
        for j, b in zip(rng.choice(M, 300, replace=False), rng.normal(0, 0.04, 300)):
            dev.update_marker(int(j), -float(b))
        y = dev.get_residual()
        ch = capi.Chain(dev, y, seed=1222)
        ys = dev.get_residual()  # centred / scaled phenotype as the chain holds it
        for _ in range(2):
            ch.iterate()
            ss = dev.sweep_stats()
            assert ss["engine"] == (1 if batch else 2) and ss["eps_sum_drift"] <= 1e-9
        beta, comp, acum = dev.get_beta()
        st = ch.state()
        eps = dev.get_residual()
        res.append((beta, comp, eps, st))
        assert st["cass"].sum() == M and st["m0"][0] == (comp != 0).sum()
        assert np.all((beta != 0) == (comp != 0)) and np.all((acum >= 0) & (acum <= 1.0 + 1e-12))
        # add X beta back onto eps with the product's own update operator
        for j in np.flatnonzero(beta):
            dev.update_marker(int(j), float(beta[j]))
        back = dev.get_residual()
        assert np.max(np.abs(back - (ys - st["mu"]))) < 1e-8
        dev.close()
    # different batch widths cut the Gram-corrected extensions differently: equal up to fp rounding
    for r in res[1:]:
        assert np.array_equal(res[0][1], r[1]) and close(res[0][0], r[0]) and close(res[0][2], r[2])
        assert close(res[0][3]["sigmaE"], r[3]["sigmaE"])


@pytest.mark.parametrize("cfg", ["c3", "c4", "c4-missing", "n1m"])
def test_full_size_properties_c3_c4(cfg):
    """BASELINE configs 3 (N=200 000, M=500 000, two groups) and 4 (N=500 000, M=1 000 000: 125 GB of packed
    genotypes in HBM) at full size through the same size-independent properties: counts add up, the residual
    identity eps + X beta == y - mu holds after the sweeps (round trip through the update operator), and the three ways through
    the sweep -- the resident engine (the default at these shapes), the batch engine with Gram corrections, and the batch
    engine's plain path (gram = 0, another batch width) -- walk the same chain."""
    N, M, G = (200000, 500000, 2) if cfg == "c3" else ((1000000, 60000, 1) if cfg == "n1m" else (500000, 1000000, 1))
    # ("n1m": a shard of a million individuals on one GPU -- four tiles per workgroup, 245 streaming workgroups of the second form)
    missing = 0.01 if cfg == "c4-missing" else 0.0  # (1 % missing calls in every column: the resident MISS build at 245 workgroups against the batch engine's)
    groups = None if G == 1 else (np.arange(M) % 2).astype(np.int32)
    mS = np.array([[0.0, 0.0001, 0.001, 0.01]]) if G == 1 else np.array([[0.0, 0.001, 0.01, 0.1]] * 2)
    res = []
    for batch, gram in ((0, 1), (256, 1), (200, 0)):
        dev = capi.Device(0)
        if batch:  # naming a batch width pins the batch engine; without options the library picks the resident engine here
            dev.set_option("batch", batch)
            dev.set_option("gram", gram)
        dev.synth_bed(N, M, seed=42, missing_rate=missing)
        rng = np.random.default_rng(1)
        dev.set_residual(rng.normal(size=N))
        for j, b in zip(rng.choice(M, 300, replace=False), rng.normal(0, 0.04, 300)):
            dev.update_marker(int(j), -float(b))
        y = dev.get_residual()
        ch = capi.Chain(dev, y, mS=mS, groups=groups, seed=1222)
        ys = dev.get_residual()
        drift = 0.0
        for _ in range(2):
            ch.iterate()
            drift = max(drift, dev.sweep_stats()["eps_sum_drift"])
        beta, comp, acum = dev.get_beta()
        st, eps = ch.state(), dev.get_residual()
        res.append((beta, comp, eps, st))
        # the sum of eps is reduced once per sweep and held (a standardised column sums to zero): what a sweep's roundings move
        # it by, measured by the library at sweep end, stays far below the 1e-9 the dots are compared at
        assert drift <= 1e-8 * max(1.0, abs(float(eps.sum()))), drift
        assert dev.sweep_stats()["engine"] == (1 if batch else 2)
        if cfg == "n1m" and not batch:
            assert dev.sweep_stats()["tiles_per_workgroup_max"] == 4 and dev.sweep_stats()["refill"] == 2
        assert st["cass"].sum() == M and st["m0"].sum() == (comp != 0).sum()
        assert np.all((beta != 0) == (comp != 0)) and np.all((acum >= 0) & (acum <= 1.0 + 1e-12))
        if gram:
            for j in np.flatnonzero(beta):
                dev.update_marker(int(j), float(beta[j]))
            assert np.max(np.abs(dev.get_residual() - (ys - st["mu"]))) < 1e-8
        dev.close()
    for r in res[1:]:
        assert np.array_equal(res[0][1], r[1]) and close(res[0][0], r[0]) and close(res[0][2], r[2])
        assert close(res[0][3]["sigmaE"], r[3]["sigmaE"]) and close(res[0][3]["sigmaG"], r[3]["sigmaG"])


# ---------------------------------------------------------------------------
# edge cases of the model and of the data shapes
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("K", [2, 3, 8])
def test_mixture_sizes(oracle, K):
    mS = np.array([[0.0] + [10.0 ** (-(K - 1 - k)) for k in range(1, K)]])
    _gpu_sweep_vs_oracle(oracle, M=150, N=700, G=1, mS=mS, groups=None, batch=64, iters=4)


@pytest.mark.parametrize("case", range(32))
def test_random_configurations_match_the_oracle(oracle, case):
    """Seeded random combinations of shape (ragged N, M), mixture size, groups, share of columns with missing calls,
    causal share, and of every tuning knob of the sweep (batch width, columns per workgroup, segments, Gram / carry /
    missing-call builds): four iterations against the oracle each."""
    rng = np.random.default_rng(1000 + case)
    N = int(rng.choice([61, 700, 4097, 9000, 20011]))
    M = int(rng.integers(40, 900))
    K = int(rng.integers(2, 9))
    G = int(rng.choice([1, 1, 2, 3]))
    geno = synth.make_genotypes(M, N, seed=500 + case, missing_rate=0.0, maf_lo=0.1 if N < 1000 else 0.01)
    for c in rng.choice(M, size=int(rng.choice([0.0, 0.1, 1.0]) * M), replace=False):
        geno[c, rng.random(N) < 0.03] = 3
    for c in range(M):  # no monomorphic marker: its scale is a division by zero in the reference too (src/BayesRRm.cpp:1506)
        if len(np.unique(geno[c][geno[c] != 3])) < 2:
            geno[c, :3] = (0, 1, 2)
    y, _ = synth.make_phenotype(geno, seed=600 + case, causal_frac=float(rng.choice([0.01, 0.05, 0.2])))
    bed = synth.pack_bed_columns(geno)
    groups = None if G == 1 else rng.integers(0, G, size=M).astype(np.int32)
    mS = np.tile(np.array([[0.0] + [10.0 ** (-(K - 1 - k)) for k in range(1, K)]]), (G, 1))
    opts = {"batch": int(rng.choice([3, 32, 100, 256])), "cols_per_group": int(rng.choice([2, 4, 8, 16])),
            "max_seg": int(rng.choice([0, 1, 2, 3, 4])), "gram": int(rng.choice([0, 1, 1, 1])),
            "carry": int(rng.choice([0, 1, 1])), "ahead": int(rng.choice([0, 0, 24, 256])), "gram_missing": int(rng.choice([-1, 0, 1])),
            "ext_limit": int(rng.choice([8, 256]))}
    ref = orc.Chain(oracle, bed, N, y, groups=groups, mS=mS, seed=77 + case, shuffle=1)
    dev = capi.Device(0)
    dev.load_bed(bed, N)
    for k, v in opts.items():
        dev.set_option(k, v)
    ch = capi.Chain(dev, y, mS=mS, groups=groups, seed=77 + case, shuffle=1)
    for it in range(4):
        ref.iterate()
        ch.iterate()
        beta, comp, acum = dev.get_beta()
        what = "case %d it %d: N=%d M=%d K=%d G=%d %r" % (case, it, N, M, K, G, opts)
        assert np.array_equal(comp, ref.arr("components")), what
        assert close(beta, ref.arr("beta")) and close(acum, ref.arr("acum")) and close(dev.get_residual(), ref.arr("eps")), what
        assert close(ch.state()["sigmaE"], ref.sigmaE) and ch.last_nnz() == oracle.orc_chain_last_nnz(ref.h), what


def test_too_many_components_is_an_error():
    bed, y = make_case(10, 50, seed=1)
    dev = capi.Device(0)
    dev.load_bed(bed, 50)
    with pytest.raises(capi.HgError):
        capi.Chain(dev, y, mS=np.array([[0.0] + [0.1] * 9]))


def test_many_groups_uses_global_tables(oracle):
    """G*K > 64: hyper tables are read from global memory instead of LDS."""
    M, G = 360, 20
    groups = (np.arange(M) % G).astype(np.int32)
    mS = np.tile(np.array([[0.0, 0.001, 0.01, 0.1]]), (G, 1))
    _gpu_sweep_vs_oracle(oracle, M=M, N=900, G=G, mS=mS, groups=groups, batch=64, iters=4)


def test_group_frozen_out_and_empty_group(oracle):
    """A group whose markers all fall into component 0 is frozen (sigmaG = 0,
    adaV = 0, src/BayesRRm.cpp:2534-2542): its markers then take no RNG draw
    (:1923-1926).  Group 2 has no markers at all (:1239-1240, :2528)."""
    M, N = 90, 600
    rng = np.random.default_rng(4)
    geno = synth.make_genotypes(M, N, seed=40, missing_rate=0.01)
    X = synth.standardize(geno)
    y = X[:, :30] @ rng.normal(0, 0.3, 30) + rng.normal(size=N) * 0.5  # only group 0 carries signal
    groups = np.where(np.arange(M) < M - 4, 0, 1).astype(np.int32)  # group 1: four markers without signal
    mS = np.tile(np.array([[0.0, 1e-5, 1e-4, 1e-3]]), (3, 1))
    bed = synth.pack_bed_columns(geno)
    ref = orc.Chain(oracle, bed, N, y, groups=groups, mS=mS, seed=2, shuffle=1)
    dev = capi.Device(0)
    dev.load_bed(bed, N)
    ch = capi.Chain(dev, y, mS=mS, groups=groups, seed=2, shuffle=1)
    frozen = False
    for it in range(25):
        ref.iterate()
        ch.iterate()
        beta, comp, acum = dev.get_beta()
        st = ch.state()
        assert np.array_equal(comp, ref.arr("components")) and close(beta, ref.arr("beta")) and close(acum, ref.arr("acum"))
        assert close(st["sigmaG"], ref.arr("sigmaG")) and st["sigmaG"][2] == 0.0
        frozen = frozen or (ref.arr("sigmaG")[1] == 0.0)
    assert frozen, "test data did not freeze group 1; pick another seed"
    assert np.all(acum[groups == 1] == 1.0) and np.all(beta[groups == 1] == 0.0)


@pytest.mark.parametrize("N,M", [(5, 3), (64, 1), (4096, 2), (4097, 9), (12289, 70)])
def test_ragged_shapes(oracle, N, M):
    rng = np.random.default_rng(N)
    geno = rng.integers(0, 3, size=(M, N)).astype(np.uint8)
    geno[:, 0], geno[:, 1] = 0, 2  # polymorphic
    if N > 8:
        geno[:, 5] = 3
    y = rng.normal(size=N)
    bed = synth.pack_bed_columns(geno)
    ref = orc.Chain(oracle, bed, N, y, seed=8)
    dev = capi.Device(0)
    dev.load_bed(bed, N)
    ch = capi.Chain(dev, y, seed=8)
    for _ in range(3):
        ref.iterate()
        ch.iterate()
        beta, comp, _ = dev.get_beta()
        assert np.array_equal(comp, ref.arr("components")) and close(beta, ref.arr("beta"))
    assert close(dev.get_residual(), ref.arr("eps"))


@pytest.mark.parametrize("missing_rate,shuffle", [(0.0, 0), (0.3, 1)])
def test_missingness_extremes_and_no_shuffle(oracle, missing_rate, shuffle):
    M, N = 200, 1300
    bed, y = make_case(M, N, seed=50, missing_rate=missing_rate)
    ref = orc.Chain(oracle, bed, N, y, seed=2, shuffle=shuffle)
    dev = capi.Device(0)
    dev.load_bed(bed, N)
    ch = capi.Chain(dev, y, seed=2, shuffle=shuffle)
    for _ in range(4):
        ref.iterate()
        ch.iterate()
    if not shuffle:
        assert np.array_equal(ch.order(), np.arange(M))
    beta, comp, _ = dev.get_beta()
    assert np.array_equal(comp, ref.arr("components")) and close(beta, ref.arr("beta"))


def test_argument_errors_are_reported():
    dev = capi.Device(0)
    with pytest.raises(capi.HgError):
        dev.set_residual(np.zeros(0))  # no data loaded
    bed, y = make_case(8, 40, seed=2)
    with pytest.raises(capi.HgError):
        dev.load_bed(bed, 40, row_begin=2, row_end=40)  # shard must start on a byte boundary
    dev.load_bed(bed, 40)
    with pytest.raises(capi.HgError):
        dev.load_bed(bed, 40)  # already loaded
    with pytest.raises(capi.HgError):
        dev.dot_marker(8)
    with pytest.raises(capi.HgError):
        dev.set_option("batch", 100000)
    ch = capi.Chain(dev, y)
    st = ch.state()
    rng = capi.RngState()
    with pytest.raises(capi.HgError):  # order outside [0, M)
        dev.sweep(np.array([0, 1, 2, 3, 4, 5, 6, 99], dtype=np.int32), st["sigmaE"], st["sigmaG"], st["estPi"],
                  np.ones(8, dtype=np.uint8), rng)


REF_TOOLS = os.path.join(ROOT, "oracle", "_ref")


@pytest.mark.skipif(not os.path.exists(os.path.join(REF_TOOLS, "pp_beta_converter")), reason="oracle/_ref tools not built")
def test_reference_postproc_tools_read_our_files(oracle, tmp_path):
    """the reference's own converters (postproc/*.cpp, compiled
    unchanged into oracle/_ref) parse the CLI's output files and print the
    oracle's numbers."""
    M, N, iters = 40, 300, 4
    geno = synth.make_genotypes(M, N, seed=71, missing_rate=0.0)
    y, _ = synth.make_phenotype(geno, seed=72, causal_frac=0.1)
    prefix, out = str(tmp_path / "d"), str(tmp_path / "o")
    synth.write_plink(prefix, synth.pack_bed_columns(geno), N, y=y)
    r = subprocess.run([EXE, "--mpibayes", "bayesMPI", "--bfile", prefix, "--pheno", prefix + ".phen", "--mcmc-out-dir", out,
                        "--mcmc-out-name", "r", "--number-individuals", str(N), "--number-markers", str(M), "--chain-length",
                        str(iters), "--thin", "1", "--save", "1", "--seed", "7", "--S", "0.001,0.01,0.1"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    ref = orc.Chain(oracle, synth.pack_bed_columns(geno), N, y, mS=np.array([[0.0, 0.001, 0.01, 0.1]]), seed=7)
    betas = []
    for _ in range(iters):
        ref.iterate()
        betas.append(ref.arr("beta").copy())
    txt = subprocess.run([os.path.join(REF_TOOLS, "pp_beta_converter"), out + "/r.bet", str(iters - 1)], capture_output=True,
                         text=True).stdout
    assert "%d markers were processed." % M in txt
    got = {}
    for line in txt.splitlines():
        if "/" in line and "=" in line and not line.startswith("read"):
            left, val = line.split("=")
            it, mk = left.split("/")
            got[(int(it), int(mk))] = float(val)
    assert len(got) == iters * M
    for it in range(iters):
        for mk in range(M):
            assert abs(got[(it, mk)] - betas[it][mk]) < 1e-9
    nz = subprocess.run([os.path.join(REF_TOOLS, "pp_extract_non_zero_betaAll"), out + "/r.bet", "0", str(iters - 1)],
                        capture_output=True, text=True).stdout
    rows = [l.split() for l in nz.splitlines() if l.strip()]
    assert len(rows) == sum(int((np.abs(b) > 1e-17).sum()) for b in betas)
    eps_txt = subprocess.run([os.path.join(REF_TOOLS, "pp_epsilon_converter"), out + "/r.eps.0"], capture_output=True, text=True).stdout
    assert "iteration %d was last logged into epsilon file." % (iters - 1) in eps_txt and "%d individuals were processed." % N in eps_txt
    vals = [float(l.split("=")[1]) for l in eps_txt.splitlines() if "/" in l and "=" in l]
    assert np.allclose(vals, ref.arr("eps"), rtol=0, atol=1e-9)


# ---------------------------------------------------------------------------
# fixed-effect covariates (src/BayesRRm.cpp:2646-2681)
# ---------------------------------------------------------------------------
def _cov_case(M, N, C, seed):
    geno = synth.make_genotypes(M, N, seed=seed, missing_rate=0.01)
    y, _ = synth.make_phenotype(geno, seed=seed + 1, causal_frac=0.05)
    rng = np.random.default_rng(seed + 2)
    X = rng.normal(size=(N, C))
    X = (X - X.mean(0)) / X.std(0, ddof=0) * np.sqrt((N - 1) / N)  # x'x = N-1, as hydra expects
    y = y + X @ rng.normal(0, 0.3, C)
    return geno, y, X


def test_covariates_chain_vs_oracle(oracle):
    M, N, C = 200, 1800, 3
    geno, y, X = _cov_case(M, N, C, seed=81)
    bed = synth.pack_bed_columns(geno)
    ref = orc.Chain(oracle, bed, N, y, seed=11)
    ref.set_covariates(X)
    dev = capi.Device(0)
    dev.load_bed(bed, N)
    ch = capi.Chain(dev, y, seed=11)
    ch.set_covariates(X)
    for it in range(6):
        ref.iterate()
        ch.iterate()
        beta, comp, _ = dev.get_beta()
        g, xi = ch.gamma()
        assert np.array_equal(comp, ref.arr("components")) and close(beta, ref.arr("beta"))
        assert close(g, ref.gamma()) and close(ch.state()["sigmaE"], ref.sigmaE)
        assert close(dev.get_residual(), ref.arr("eps"))
    assert np.all(np.abs(g) > 1e-3)  # the fixed effects were actually sampled


def test_cli_covariates_with_na(oracle, tmp_path):
    M, N, C, iters = 80, 500, 2, 5
    geno, y, X = _cov_case(M, N, C, seed=91)
    prefix, out = str(tmp_path / "d"), str(tmp_path / "o")
    synth.write_plink(prefix, synth.pack_bed_columns(geno), N, y=y, na_rows=[7])
    with open(prefix + ".cov", "w") as f:
        for i in range(N):
            vals = ["NA" if (i == 20 and c == 1) else repr(float(X[i, c])) for c in range(C)]
            f.write("fam%d ind%d %s\n" % (i, i, " ".join(vals)))
    r = subprocess.run([EXE, "--mpibayes", "bayesMPI", "--bfile", prefix, "--pheno", prefix + ".phen", "--covariates",
                        prefix + ".cov", "--mcmc-out-dir", out, "--mcmc-out-name", "r", "--number-individuals", str(N),
                        "--number-markers", str(M), "--chain-length", str(iters), "--thin", "1", "--save", "2", "--seed", "5"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    keep = np.ones(N, dtype=bool)
    keep[[7, 20]] = False
    ref = orc.Chain(oracle, synth.pack_bed_columns(geno[:, keep]), int(keep.sum()), y[keep], seed=5,
                    mS=np.array([[0.0, 0.01, 0.001, 0.0001]]))  # hydra's default --S order (options.hpp:108-111)
    ref.set_covariates(X[keep])
    _, betas = _read_bet(out + "/r.bet", M, np.float64)
    gam_it4 = None
    for it in range(iters):
        ref.iterate()
        assert close(betas[it], ref.arr("beta"))
        if it == 4:
            gam_it4 = ref.gamma()
    raw = open(out + "/r.gam.0", "rb").read()
    it_s, glen = struct.unpack("<II", raw[:8])
    assert (it_s, glen) == (4, C) and close(np.frombuffer(raw[8:8 + 8 * C], dtype=np.float64), gam_it4)


# ---- checkpoint / restart (src/BayesRRm.cpp:842-928, :2802-2838) --------------
@pytest.mark.parametrize("with_cov", [False, True])
def test_chain_restore_continues_the_chain(oracle, with_cov):
    """Dump at iteration 3 with full precision, restore into a fresh device + chain: the
    continuation is the uninterrupted chain, bit for bit, and matches the oracle's mirror."""
    M, N, C_ = 150, 2300, 2
    geno, y, X = _cov_case(M, N, C_, seed=61)
    bed = synth.pack_bed_columns(geno)

    def fresh(seed):
        dev = capi.Device(0)
        dev.load_bed(bed, N)
        dev.set_option("gram", 0)
        ch = capi.Chain(dev, y, seed=seed)
        if with_cov:
            ch.set_covariates(X)
        return dev, ch

    dev_a, a = fresh(23)
    ref = orc.Chain(oracle, bed, N, y, seed=23)
    if with_cov:
        ref.set_covariates(X)
    for _ in range(4):
        a.iterate()
        ref.iterate()
    st = a.state()
    beta, comp, _ = dev_a.get_beta()
    snap = dict(iteration=3, sigmaE=st["sigmaE"], mu=st["mu"], sigmaG=st["sigmaG"], estPi=st["estPi"], beta=beta,
                components=comp, eps=dev_a.get_residual(), order=a.order(), rng_words=a.rng_words())
    if with_cov:
        g, xi = a.gamma()
        snap.update(gamma=g, xI=xi)
    assert np.array_equal(snap["rng_words"], ref.rng_words())  # same stream position, same 624 printed words
    dev_b, b = fresh(4242)
    b.restore(**snap)
    ref_b = orc.Chain(oracle, bed, N, y, seed=1)
    if with_cov:
        ref_b.set_covariates(X)
    ref_b.restore(**snap)
    for it in range(4, 8):
        a.iterate()
        b.iterate()
        ref_b.iterate()
        ba, ca, _ = dev_a.get_beta()
        bb, cb, _ = dev_b.get_beta()
        assert np.array_equal(ba, bb) and np.array_equal(ca, cb)
        assert np.array_equal(dev_a.get_residual(), dev_b.get_residual())
        sa, sb = a.state(), b.state()
        assert sa["sigmaE"] == sb["sigmaE"] and sa["mu"] == sb["mu"] and np.array_equal(sa["sigmaG"], sb["sigmaG"])
        assert a.csv_line(it) == b.csv_line(it)
        assert np.array_equal(cb, ref_b.arr("components")) and close(bb, ref_b.arr("beta")) and close(sb["sigmaE"], ref_b.sigmaE)


def _dump(path, dtype):
    raw = open(path, "rb").read()
    it, n = struct.unpack("<II", raw[:8])
    return it, np.frombuffer(raw[8:8 + n * np.dtype(dtype).itemsize], dtype=dtype)


def test_cli_restart_from_dump_files(oracle, tmp_path):
    """6 iterations with --save 3 (one dump, at iteration 3), then --restart up to 9: <name>_rs.* carry
    iterations 4..8 and equal the oracle restored from the very same files (hyper-parameters at the
    .csv's printed precision, as init_from_restart takes them)."""
    M, N, C_ = 100, 450, 2
    geno, y, X = _cov_case(M, N, C_, seed=71)
    prefix, out = str(tmp_path / "d"), str(tmp_path / "o")
    synth.write_plink(prefix, synth.pack_bed_columns(geno), N, y=y)
    with open(prefix + ".cov", "w") as f:
        for i in range(N):
            f.write("fam%d ind%d %r %r\n" % (i, i, float(X[i, 0]), float(X[i, 1])))
    base = [EXE, "--mpibayes", "bayesMPI", "--bfile", prefix, "--pheno", prefix + ".phen", "--covariates", prefix + ".cov",
            "--mcmc-out-dir", out, "--mcmc-out-name", "r", "--number-individuals", str(N), "--number-markers", str(M),
            "--thin", "1", "--save", "3", "--seed", "8", "--S", "0.0001,0.001,0.01"]
    r = subprocess.run(base + ["--chain-length", "6"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    words = np.array(open(out + "/r.rng.0").read().split(), dtype=np.uint64)
    assert words.shape == (624,) and words.max() < 2 ** 32
    first = {p: open(out + "/" + p, "rb").read() for p in ("r.csv", "r.bet", "r.xbet", "r.eps.0")}

    r = subprocess.run(base + ["--chain-length", "9", "--restart"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "RESTART: iteration_to_restart_from = 3" in r.stdout and "will restart from iteration: 4" in r.stdout
    for p, raw in first.items():  # the failed job's files stay untouched (:1206)
        assert open(out + "/" + p, "rb").read() == raw

    # the oracle, restored from the same dump files the way init_from_restart reads them
    csv3 = [float(x) for x in open(out + "/r.csv").read().splitlines()[3].split(",")]
    assert csv3[0] == 3 and csv3[1] == 1
    sigmaG, sigmaE, pi = csv3[2:3], csv3[3], csv3[8:12]
    it_e, eps = _dump(out + "/r.eps.0", np.float64)
    it_m, mrk = _dump(out + "/r.mrk.0", np.int32)
    it_g, gam = _dump(out + "/r.gam.0", np.float64)
    it_x, xiv = _dump(out + "/r.xiv.0", np.int32)
    assert (it_e, it_m, it_g, it_x) == (3, 3, 3, 3) and eps.shape == (N,) and sorted(mrk) == list(range(M))
    xb, xc = open(out + "/r.xbet", "rb").read(), open(out + "/r.xcpn", "rb").read()
    assert struct.unpack("<II", xb[:8]) == (M, 3) and struct.unpack("<II", xc[:8]) == (M, 3)
    mus = open(out + "/r.mus.0", "rb").read()
    k, mu = struct.unpack("<Id", mus[36:48])
    assert k == 3
    ref = orc.Chain(oracle, synth.pack_bed_columns(geno), N, y, seed=12345)
    ref.set_covariates(X)
    ref.restore(3, sigmaE, mu, sigmaG, pi, np.frombuffer(xb[8:], dtype=np.float64), np.frombuffer(xc[8:], dtype=np.int32), eps, mrk,
                words.astype(np.uint32), gamma=gam, xI=xiv)
    its, betas = _read_bet(out + "/r_rs.bet", M, np.float64)
    _, comps = _read_bet(out + "/r_rs.cpn", M, np.int32)
    csv = open(out + "/r_rs.csv").read().splitlines()
    assert its == [4, 5, 6, 7, 8] and len(csv) == 5
    for k, it in enumerate(its):
        ref.iterate()
        assert np.array_equal(comps[k], ref.arr("components")) and close(betas[k], ref.arr("beta"))
        got, want = [float(x) for x in csv[k].split(",")], [float(x) for x in ref.csv_line(it).split(",")]
        assert close(got, want, 1e-9)
    it_e2, eps2 = _dump(out + "/r_rs.eps.0", np.float64)
    assert it_e2 == 6 and os.path.exists(out + "/r_rs.rng.0")
    # --ignore-xfiles reads the same state out of the .bet/.cpn history
    r = subprocess.run(base + ["--chain-length", "6", "--restart", "--ignore-xfiles"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    its2, betas2 = _read_bet(out + "/r_rs.bet", M, np.float64)
    assert its2 == [4, 5] and np.array_equal(betas2, betas[:2])


def test_cli_restart_refuses_bad_inputs(tmp_path):
    M, N = 20, 64
    geno = synth.make_genotypes(M, N, seed=1)
    y, _ = synth.make_phenotype(geno, seed=2)
    prefix, out = str(tmp_path / "d"), str(tmp_path / "o")
    synth.write_plink(prefix, synth.pack_bed_columns(geno), N, y=y)
    base = [EXE, "--mpibayes", "bayesMPI", "--bfile", prefix, "--pheno", prefix + ".phen", "--mcmc-out-dir", out, "--mcmc-out-name", "r",
            "--number-individuals", str(N), "--number-markers", str(M), "--thin", "1", "--seed", "8"]
    r = subprocess.run(base + ["--chain-length", "3", "--save", "5"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0
    r = subprocess.run(base + ["--chain-length", "9", "--save", "5", "--restart"], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "restarting a chain from iteration 0" in r.stderr  # only it 0 is a multiple of --save
    r = subprocess.run(base + ["--chain-length", "4", "--save", "2"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0
    with open(out + "/r.mrk.0", "r+b") as f:  # a dump from another iteration is refused (data.cpp:53-56)
        f.write(struct.pack("<I", 1))
    r = subprocess.run(base + ["--chain-length", "9", "--save", "2", "--restart"], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "Mismatch between expected and read" in r.stderr


def test_sampler_recovers_the_simulated_model():
    """Not a parity statement: 60 iterations at the size of the shipped example (N = 5 000, M = 10 000, 1 % causal,
    h2 = 0.5) -- the posterior mean of the effects lines up with the simulated ones and the variance components
    land near the truth."""
    N, M = 5000, 10000
    geno = synth.make_genotypes(M, N, seed=101)
    y, beta_true = synth.make_phenotype(geno, seed=102, h2=0.5, causal_frac=0.01)
    dev = capi.Device(0)
    dev.load_bed(synth.pack_bed_columns(geno), N)
    ch = capi.Chain(dev, y, seed=1222)
    post, h2 = np.zeros(M), []
    for it in range(60):
        ch.iterate()
        if it >= 30:
            post += dev.get_beta()[0] / 30.0
            st = ch.state()
            h2.append(st["sigmaG"].sum() / (st["sigmaG"].sum() + st["sigmaE"]))
    scale = np.sqrt(((y - y.mean()) ** 2).sum() / (N - 1))  # the chain works on the standardised phenotype
    assert np.corrcoef(post, beta_true / scale)[0, 1] > 0.85
    assert 0.35 < np.mean(h2) < 0.65


def test_three_hundred_iterations_stay_on_the_oracle(oracle):
    """A long grouped chain with missing calls: 300 iterations, components identical in every one of them, the
    generator in the same state at the end, effects within the tolerance throughout (the BayesR chain does not
    amplify rounding differences the way the ARS-driven BayesW chain does)."""
    M, N, iters = 3000, 2500, 300
    geno = synth.make_genotypes(M, N, seed=77, missing_rate=0.01)
    y, _ = synth.make_phenotype(geno, seed=78, causal_frac=0.03)
    bed = synth.pack_bed_columns(geno)
    groups = (np.arange(M) % 3 == 0).astype(np.int32)
    mS = np.array([[0.0, 0.0001, 0.001, 0.01], [0.0, 0.001, 0.01, 0.1]])
    ref = orc.Chain(oracle, bed, N, y, groups=groups, mS=mS, seed=99)
    dev = capi.Device(0)
    dev.load_bed(bed, N)
    ch = capi.Chain(dev, y, mS=mS, groups=groups, seed=99)
    for it in range(iters):
        ref.iterate()
        ch.iterate()
        beta, comp, _ = dev.get_beta()
        assert np.array_equal(comp, ref.arr("components")), it
        assert close(beta, ref.arr("beta")), it
    st = ch.state()
    rx, ridx = ref.rng_state()
    assert _same_stream(st["rng_x"], st["rng_idx"], rx, ridx) and close(st["sigmaE"], ref.sigmaE) and close(st["sigmaG"], ref.arr("sigmaG"))
