"""The resident sweep engine (hydra_amd/csrc/hg_resident.hip.h: one launch per sweep, individuals sharded over the
compute units, eps in registers, the column window in LDS, one walker workgroup) against the CPU oracle, through the C ABI.

Same bar as tests/test_gpu_parity.py: marker order, mixture-component indices, cass, the generator state and the number
of updates exact; beta, acum, the hyper-parameters and the residual within 1e-9 (the dots are summed in a different,
fixed order -- here even as integers -- than the oracle's sequential loop).
"""
import os

import numpy as np
import pytest

import orc
from hydra_amd import capi, synth
from test_gpu_parity import close, _same_stream

pytestmark = pytest.mark.gpu


def make_case(M, N, seed=7, causal_frac=0.05, missing_rate=0.0, missing_cols=1.0):
    geno = synth.make_genotypes(M, N, seed=seed, missing_rate=0.0)
    if missing_rate > 0.0:  # missing calls in a share of the columns (the others take the plain path inside the same build)
        rng = np.random.default_rng(seed + 17)
        for c in np.flatnonzero(rng.random(M) < missing_cols):
            geno[c, rng.random(N) < missing_rate] = 3
            if len(np.unique(geno[c][geno[c] != 3])) < 2:  # (no monomorphic marker: division by zero in the reference too)
                geno[c, :3] = (0, 1, 2)
    y, _ = synth.make_phenotype(geno, seed=seed + 1, h2=0.5, causal_frac=causal_frac)
    return synth.pack_bed_columns(geno), y


def run_vs_oracle(oracle, M, N, iters=3, groups=None, mS=None, opts=None, seed=1222, causal_frac=0.05, expect_T=None, missing_rate=0.0, missing_cols=1.0, expect_walker=None,
                  expect_refill=None):
    bed, y = make_case(M, N, seed=M + N, causal_frac=causal_frac, missing_rate=missing_rate, missing_cols=missing_cols)
    ref = orc.Chain(oracle, bed, N, y, groups=groups, mS=mS, seed=seed, shuffle=1)
    dev = capi.Device(0)
    dev.load_bed(bed, N)
    dev.set_option("engine", 2)
    for k, v in (opts or {}).items():
        dev.set_option(k, v)
    ch = capi.Chain(dev, y, mS=mS, groups=groups, seed=seed, shuffle=1)
    for it in range(iters):
        ref.iterate()
        ch.iterate()
        beta, comp, acum = dev.get_beta()
        st = ch.state()
        ss = dev.sweep_stats()
        assert ss["engine"] == 2 and ss["launches"] == 1 and ss["accepted_markers"] == M
        if expect_walker:
            assert ss["walker"] == expect_walker
        if expect_refill:
            assert ss["refill"] == expect_refill
        if expect_T:
            assert ss["tiles_per_workgroup_max"] == expect_T
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
        assert ss["eps_sum_drift"] <= 1e-10 * max(1.0, abs(float(np.sum(ref.arr("eps")))))
    return ch, ref, dev


@pytest.mark.parametrize("walker", [1, 2])
@pytest.mark.parametrize("N", [37, 1024, 4099, 9001])
def test_small_shapes(oracle, N, walker):
    run_vs_oracle(oracle, 300, N, opts={"walker": walker}, expect_walker=walker)


@pytest.mark.parametrize("window", [8, 32, 128, 256])
def test_window_sizes(oracle, window):
    # a window of 8 columns forces rounds that only advance, refills inside a walk and events at the window's edge
    run_vs_oracle(oracle, 500, 3000, opts={"window": window})


@pytest.mark.parametrize("refill", [1, 2])
@pytest.mark.parametrize("cus,T", [(9, 1), (7, 2), (5, 2)])
def test_tiles_per_workgroup(oracle, cus, T, refill):
    # N = 8000 -> 8 wave tiles: res_cus decides how many tiles one workgroup holds in registers.  Both forms of the streaming workgroups:
    # 1 = every wave whole columns, fused multiply-adds per individual; 2 (the default) = every wave a slice of the individuals, integer
    # matrix products over eps's signed base-256 digits (hg_streamer2.hip.h)
    run_vs_oracle(oracle, 400, 8000, opts={"res_cus": cus, "refill": refill}, expect_T=T, expect_refill=refill)


@pytest.mark.parametrize("N,cus,miss", [(8000, 3, 0.0), (8000, 3, 0.03), (20011, 6, 0.0), (14001, 5, 0.02)])
def test_four_tiles_per_workgroup(oracle, N, cus, miss):
    """More than 2048 individuals per compute unit: the second form of the streaming workgroups holds four tiles (32 bytes of a column per
    lane and group of sixteen, the digit image made in two halves, a 128-column window) -- shards of up to 255 x 4096 individuals stay on
    the resident engine.  8 tiles on 2 streaming workgroups, 20 on 5, 14 on 4 (the last one ragged); clean and with missing calls."""
    ch, ref, dev = run_vs_oracle(oracle, 400, N, iters=3, opts={"res_cus": cus}, expect_T=4, expect_refill=2, missing_rate=miss)


def test_many_workgroups_multishard(oracle):
    # 20 wave tiles -> 20 streaming workgroups over 8 Gram shards and 4 raw-dot shards
    run_vs_oracle(oracle, 400, 20011, iters=3)


@pytest.mark.parametrize("K,mS", [(2, [0.0, 0.01]), (3, [0.0, 0.001, 0.01]), (5, [0.0, 1e-4, 1e-3, 1e-2, 1e-1]),
                                  (8, [0.0, 1e-5, 1e-4, 5e-4, 1e-3, 5e-3, 1e-2, 1e-1])])
def test_mixture_sizes(oracle, K, mS):
    run_vs_oracle(oracle, 300, 2500, mS=np.array([mS]))


def test_groups(oracle):
    M = 400
    groups = (np.arange(M) % 3).astype(np.int32)
    mS = np.array([[0.0, 0.001, 0.01, 0.1], [0.0, 0.0001, 0.001, 0.01], [0.0, 0.01, 0.1, 1.0]])
    run_vs_oracle(oracle, M, 3000, groups=groups, mS=mS)


def test_dense_model_many_events(oracle):
    # half of the markers causal: most rounds end on an event a few positions on; predicted and new events interleave
    run_vs_oracle(oracle, 300, 2000, iters=4, causal_frac=0.5)


def test_longer_chain_crosses_generator_blocks(oracle):
    # 8 iterations of 700 markers: the walker regenerates and switches MT19937 blocks several times per sweep
    run_vs_oracle(oracle, 700, 1500, iters=8)


def test_engines_agree(oracle):
    """The batch engine and the resident engine run the same chain: components identical, floating point within tolerance."""
    M, N = 600, 6000
    bed, y = make_case(M, N, seed=5)
    out = []
    for engine in (1, 2):
        dev = capi.Device(0)
        dev.load_bed(bed, N)
        dev.set_option("engine", engine)
        ch = capi.Chain(dev, y, seed=77, shuffle=1)
        for _ in range(4):
            ch.iterate()
        beta, comp, acum = dev.get_beta()
        assert dev.sweep_stats()["engine"] == engine
        out.append((beta, comp, acum, dev.get_residual(), ch.state()))
    assert np.array_equal(out[0][1], out[1][1])
    assert close(out[0][0], out[1][0]) and close(out[0][2], out[1][2]) and close(out[0][3], out[1][3])
    assert np.array_equal(out[0][4]["rng_x"], out[1][4]["rng_x"]) and out[0][4]["rng_idx"] == out[1][4]["rng_idx"]


def test_geometry_does_not_change_the_chain(oracle):
    """Sums over workgroups are integers (Gram terms; raw dots as fixed point), so the chain does not depend on the order
    in which workgroups arrive; the number of workgroups, tiles per workgroup and the window only move the roundings of the
    dots (a workgroup's part is rounded to the fixed-point grid): components identical, floating point far inside the tolerance."""
    M, N = 500, 8000
    bed, y = make_case(M, N, seed=9)
    outs = []
    for opts in ({"res_cus": 9}, {"res_cus": 5}, {"res_cus": 6, "window": 64}, {"window": 16}):
        dev = capi.Device(0)
        dev.load_bed(bed, N)
        dev.set_option("engine", 2)
        for k, v in opts.items():
            dev.set_option(k, v)
        ch = capi.Chain(dev, y, seed=31, shuffle=1)
        for _ in range(3):
            ch.iterate()
        beta, comp, acum = dev.get_beta()
        outs.append((beta, comp, acum))
    for o in outs[1:]:
        assert np.array_equal(o[1], outs[0][1])
        assert close(o[0], outs[0][0], 1e-10) and close(o[2], outs[0][2], 1e-10)


@pytest.mark.parametrize("window", [16, 64, 256])
def test_predicted_pivots(oracle, window):
    """Option pivots: the Gram terms of the markers whose effect is non-zero at sweep start come with the streamed columns, and their
    events need no round trip (messages RS_PIVOT, corrections from the stored terms, also for columns whose dot arrives later)."""
    ch, ref, dev = run_vs_oracle(oracle, 500, 5000, iters=5, opts={"window": window, "pivots": 1}, causal_frac=0.2)
    assert dev.sweep_stats()["pivots"] > 0


def test_predicted_pivots_many_workgroups(oracle):
    ch, ref, dev = run_vs_oracle(oracle, 600, 20011, iters=4, opts={"pivots": 1}, causal_frac=0.1)
    assert dev.sweep_stats()["pivots"] > 0


@pytest.mark.parametrize("refill", [1, 2])
@pytest.mark.parametrize("N,rate,cols", [(1024, 0.02, 1.0), (4099, 0.01, 1.0), (9001, 0.05, 0.3), (20011, 0.01, 1.0)])
def test_missing_calls(oracle, N, rate, cols, refill):
    """Columns with missing calls (the build that keeps s2 = sum of eps over a column's calls per column and takes the four-term
    Gram sums A, B, C, D of src/BayesRRm.cpp:1785-1790's algebra): in every column, or in a share of them next to clean ones.
    Both forms of the streaming workgroups (the second takes R = sum of eps over the missing calls as a second matrix product)."""
    run_vs_oracle(oracle, 400, N, iters=4, missing_rate=rate, missing_cols=cols, opts={"refill": refill}, expect_refill=refill)


def test_a_residual_beyond_the_digits_range(oracle):
    """The second form of the streaming workgroups holds eps as round(eps 2^44) in seven signed base-256 digits: |eps| < 64 is required
    (standardised phenotypes stay within a few units).  Left to itself the library sees the outlier in its pre-sweep reduction (the sum of
    eps^8 bounds max |eps|) and takes the first form, with the oracle's chain; asked for the second form by option, the kernel refuses
    the sweep with error 5 -- never a wrapped sum."""
    M, N = 200, 9001
    bed, y = make_case(M, N, seed=5)
    y = y.copy()
    y[7] += 1e6 * np.std(y)  # one outlier: ~ sqrt(N - 1) = 95 standard deviations after the phenotype is standardised
    ref = orc.Chain(oracle, bed, N, y, seed=1222, shuffle=1)
    dev = capi.Device(0)
    dev.load_bed(bed, N)
    dev.set_option("engine", 2)
    ch = capi.Chain(dev, y, seed=1222, shuffle=1)
    assert np.abs(dev.get_residual()).max() > 64.0
    for _ in range(2):
        ref.iterate()
        ch.iterate()
        assert dev.sweep_stats()["refill"] == 1
        beta, comp, _ = dev.get_beta()
        assert np.array_equal(comp, ref.arr("components")) and close(beta, ref.arr("beta"))
    dev2 = capi.Device(0)
    dev2.load_bed(bed, N)
    dev2.set_option("engine", 2)
    dev2.set_option("refill", 2)
    ch2 = capi.Chain(dev2, y, seed=1222, shuffle=1)
    with pytest.raises(capi.HgError, match="abort code 5"):
        ch2.iterate()


def test_missing_calls_small_window_and_two_tiles(oracle):
    run_vs_oracle(oracle, 400, 8000, iters=3, missing_rate=0.03, opts={"window": 16, "res_cus": 5}, expect_T=2)


def test_missing_calls_heavy(oracle):
    # a fifth of all calls missing: the sparse terms P, Q, X are as large as they get against A
    run_vs_oracle(oracle, 300, 3000, iters=3, missing_rate=0.2, causal_frac=0.2)


def test_refused_where_it_does_not_apply(oracle):
    geno = synth.make_genotypes(50, 6000, seed=1, missing_rate=0.0)
    y, _ = synth.make_phenotype(geno, seed=2)
    dev = capi.Device(0)
    dev.load_bed(synth.pack_bed_columns(geno), 6000)
    dev.set_option("engine", 2)
    dev.set_option("res_cus", 2)  # one streaming workgroup: six wave tiles are more than it holds in registers
    ch = capi.Chain(dev, y, seed=1)
    with pytest.raises(capi.HgError, match="resident engine does not apply"):
        ch.iterate()


@pytest.mark.parametrize("case", range(int(os.environ.get("HG_RANDOM_CASES", "24"))))
def test_random_configurations_match_the_oracle(oracle, case):
    """Seeded random combinations of shape (ragged N, M), mixture size, groups, share of columns with missing calls and their rate,
    causal share, window, compute units in use (one or two tiles per workgroup) and predicted pivots: four iterations against the
    oracle each, on the resident engine."""
    rng = np.random.default_rng(7000 + case)
    N = int(rng.choice([61, 700, 4097, 9000, 20011]))
    M = int(rng.integers(40, 900))
    K = int(rng.integers(2, 9))
    G = int(rng.choice([1, 1, 2, 3]))
    mS = np.sort(rng.uniform(1e-5, 0.5, size=(G, K - 1)), axis=1)
    mS = np.concatenate([np.zeros((G, 1)), mS], axis=1)
    groups = None if G == 1 else rng.integers(0, G, size=M).astype(np.int32)
    if groups is not None:
        groups[:G] = np.arange(G)  # every group has a marker
    missing_cols = float(rng.choice([0.0, 0.0, 0.2, 1.0]))
    opts = {"window": int(rng.choice([8, 32, 128, 256]))}
    tiles = (N + 4095) // 4096 * 4  # wave tiles of the padded shard
    if tiles > 1 and rng.random() < 0.5:
        opts["res_cus"] = (tiles + 1) // 2 + 1  # two tiles per workgroup
    if missing_cols == 0.0 and rng.random() < 0.3:
        opts["pivots"] = 1
    if case >= 24:  # (HG_RANDOM_CASES > 24: a longer campaign over the knobs of round 4 as well -- forms of the streaming workgroups, walkers,
        # announcements, where the window ends, early advances, four tiles per workgroup; the first 24 cases stay what they were)
        if rng.random() < 0.3:
            opts["refill"] = 1
        if rng.random() < 0.25:
            opts["walker"] = 1
        if rng.random() < 0.3:
            opts["announce"] = 0
        if rng.random() < 0.3:
            opts["window_end16"] = 0
        opts["early_advance"] = int(rng.choice([0, 8, 24, 64]))
        if tiles >= 8 and opts.get("refill", 2) == 2 and "pivots" not in opts and rng.random() < 0.4:
            opts["res_cus"] = (tiles + 3) // 4 + 1  # four tiles per workgroup
    run_vs_oracle(oracle, M, N, iters=4, groups=groups, mS=mS, opts=opts, seed=int(rng.integers(1, 1 << 30)), causal_frac=float(rng.choice([0.01, 0.05, 0.3])),
                  missing_rate=0.03 if missing_cols > 0.0 else 0.0, missing_cols=missing_cols)


def test_auto_engine_falls_back_to_the_batch_engine(oracle):
    """engine = 0 (the default): where the resident engine does not apply -- here one streaming workgroup for six wave tiles --
    the sweep runs on the batch engine, silently and with the same chain."""
    M, N = 200, 6000
    bed, y = make_case(M, N, seed=3)
    ref = orc.Chain(oracle, bed, N, y, seed=5, shuffle=1)
    dev = capi.Device(0)
    dev.load_bed(bed, N)
    dev.set_option("res_cus", 2)
    ch = capi.Chain(dev, y, seed=5, shuffle=1)
    for _ in range(3):
        ref.iterate()
        ch.iterate()
        assert dev.sweep_stats()["engine"] == 1
        beta, comp, _ = dev.get_beta()
        assert np.array_equal(comp, ref.arr("components")) and close(beta, ref.arr("beta"))
