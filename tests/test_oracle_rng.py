"""The oracle's RNG restatement against every pin available without Boost:
MT19937 known answer (C++ standard), the Ziggurat tables inside the reference's
ELF (golden fixture + live bytes when /root/reference is present), distribution
laws (KS tests vs scipy), and regression vectors."""
import ctypes as C
import os
import struct

import numpy as np
import pytest
from scipy import stats

import orc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_mt19937_known_answer(oracle):
    g = orc.OrcMt()
    oracle.orc_rng_seed(g, 5489)
    for _ in range(9999):
        oracle.orc_rng_u32(g)
    assert oracle.orc_rng_u32(g) == 4123659995  # [rand.predef] 10000th invocation


def _oracle_tables(oracle):
    out = []
    for which in range(4):
        n = C.c_int()
        p = oracle.orc_zig_table(which, C.byref(n))
        out.append(np.ctypeslib.as_array(p, shape=(n.value,)).copy())
    return out


def test_ziggurat_tables_match_reference_elf_fixture(oracle):
    gold = np.load(os.path.join(GOLD, "boost_ziggurat_tables.npz"))
    for got, key in zip(_oracle_tables(oracle), ("normal_x", "normal_y", "exp_x", "exp_y")):
        assert got.tobytes() == gold[key].tobytes(), key  # bit for bit


@pytest.mark.skipif(not os.path.exists("/root/reference/src/hydra"), reason="reference checkout not present")
def test_ziggurat_tables_match_live_elf_bytes(oracle):
    d = open("/root/reference/src/hydra", "rb").read()
    for got, (off, n) in zip(_oracle_tables(oracle), ((0xD7FC0, 129), (0xD83E0, 129), (0xD6F80, 257), (0xD77A0, 257))):
        assert got.tobytes() == d[off:off + 8 * n]


def test_product_tables_equal_oracle_tables(oracle):
    """The product carries its own copy of the generated header."""
    root = os.path.dirname(GOLD[:-len("/golden")])
    a = open(os.path.join(root, "oracle", "zig_tables.h")).read().replace("ORC_ZIG_", "X_")
    b = open(os.path.join(root, "hydra_amd", "csrc", "hg_zig_tables.h")).read().replace("HG_ZIG_", "X_")
    assert a == b


def test_distribution_laws(oracle):
    g = orc.OrcMt()
    oracle.orc_rng_seed(g, 4242)
    n = 60000
    z = np.array([oracle.orc_rng_norm(g, 1.5, 4.0) for _ in range(n)])
    assert stats.kstest(z, "norm", args=(1.5, 2.0)).pvalue > 1e-3
    e = np.array([oracle.orc_rng_exp(g, 2.5) for _ in range(n)])
    assert stats.kstest(e, "expon", args=(0, 1 / 2.5)).pvalue > 1e-3
    for a in (0.3, 1.0, 2.5, 40.0):
        x = np.array([oracle.orc_rng_gamma(g, a, 1.7) for _ in range(n)])
        assert stats.kstest(x, "gamma", args=(a, 0, 1.7)).pvalue > 1e-3, a
    b = np.array([oracle.orc_rng_beta(g, 2.0, 5.0) for _ in range(n)])
    assert stats.kstest(b, "beta", args=(2.0, 5.0)).pvalue > 1e-3
    # inv_scaled_chisq(nu, s2) = 1/Gamma(nu/2, scale 2/(nu s2))  (src/distributions_boost.cpp:89-107)
    nu, s2 = 9.0, 0.4
    v = np.array([oracle.orc_rng_inv_scaled_chisq(g, nu, s2) for _ in range(n)])
    assert stats.kstest(1.0 / v, "gamma", args=(nu / 2, 0, 2.0 / (nu * s2))).pvalue > 1e-3
    u = np.array([oracle.orc_rng_unif(g) for _ in range(n)])
    assert u.min() >= 0 and u.max() < 1 and stats.kstest(u, "uniform").pvalue > 1e-3


def test_normal_tail_and_wedge_paths_are_exercised(oracle):
    g = orc.OrcMt()
    oracle.orc_rng_seed(g, 7)
    z = np.array([oracle.orc_rng_norm(g, 0.0, 1.0) for _ in range(400000)])
    tail = np.abs(z) > 3.4426198558966523
    assert 0.5 < tail.sum() / (400000 * 2 * stats.norm.sf(3.4426198558966523)) < 1.5


def test_int_float_pair_layout(oracle):
    """u1 low 8 bits = bucket, high 24 bits + 29 bits of u2 = 53-bit uniform
    (the layout found in the ELF's generate_int_float_pair): the
    first normal draw can be recomputed by hand from the first two outputs."""
    g = orc.OrcMt()
    oracle.orc_rng_seed(g, 99)
    u1, u2 = oracle.orc_rng_u32(g), oracle.orc_rng_u32(g)
    gold = np.load(os.path.join(GOLD, "boost_ziggurat_tables.npz"))
    r = ((u1 >> 8) * 2.0 ** -24 + (u2 & 0x1FFFFFFF)) * 2.0 ** -29
    bucket = u1 & 0xFF
    i, sign = bucket >> 1, (bucket & 1) * 2 - 1
    x = r * gold["normal_x"][i]
    oracle.orc_rng_seed(g, 99)
    z = oracle.orc_rng_norm(g, 0.0, 1.0)
    if x < gold["normal_x"][i + 1]:
        assert z == x * sign
    else:
        pytest.skip("first draw left the fast path for this seed")


def test_regression_vectors(oracle):
    gold = np.load(os.path.join(GOLD, "rng_kat.npz"))
    g = orc.OrcMt()
    oracle.orc_rng_seed(g, 1222)
    assert np.array_equal(gold["u32"], np.array([oracle.orc_rng_u32(g) for _ in range(16)], dtype=np.uint32))
    oracle.orc_rng_seed(g, 1222)
    assert np.array_equal(gold["unif"], np.array([oracle.orc_rng_unif(g) for _ in range(16)]))
    assert np.array_equal(gold["unif"], gold["u32"] / 4294967296.0)
    oracle.orc_rng_seed(g, 1222)
    assert np.array_equal(gold["norm"], np.array([oracle.orc_rng_norm(g, 0.5, 2.0) for _ in range(4000)]))
    oracle.orc_rng_seed(g, 1222)
    assert np.array_equal(gold["exp"], np.array([oracle.orc_rng_exp(g, 1.5) for _ in range(2000)]))
    for name, shape in (("gamma_lt1", 0.3), ("gamma_eq1", 1.0), ("gamma_gt1", 7.25)):
        oracle.orc_rng_seed(g, 1222)
        assert np.array_equal(gold[name], np.array([oracle.orc_rng_gamma(g, shape, 2.0) for _ in range(500)]))
    oracle.orc_rng_seed(g, 1222)
    assert np.array_equal(gold["beta"], np.array([oracle.orc_rng_beta(g, 1.0, 1.0) for _ in range(100)]))
    oracle.orc_rng_seed(g, 1222)
    v = np.arange(40, dtype=np.int32)
    oracle.orc_rng_shuffle(g, orc.iptr(v), 40)
    assert sorted(v) == list(range(40))
    assert np.array_equal(v, gold["shuffle40"])  # libstdc++ 6.5 algorithm, restated (the reference binary's toolchain)


def test_gamma_restatement_has_the_call_structure_of_the_reference_elf():
    """Structural pin of the Boost 1.67 gamma_distribution restatement (orc_rng.h: orc_rgamma; hg_rng.h: rgamma)
    against the reference's prebuilt binary, READ AS DATA (disassembled, never executed): the instantiated
    gamma_distribution<double>::operator()(mt19937&) calls exactly one tan, one log, one pow, five exp and has two
    call sites of the exponential's generate_int_float_pair -- what the restated algorithm needs: alpha > 1: tan,
    log, exp (Cauchy rejection); alpha < 1: exponential draw, exp(-y/alpha), exp(-x) or pow(x, alpha - 1);
    alpha == 1: exponential draw; and one exp in the wedge test of each of the two inlined exponential Ziggurats.
    Skipped where the reference checkout or binutils are absent (the GPU box)."""
    import re
    import shutil
    import subprocess
    elf = "/root/reference/src/hydra"
    if not os.path.exists(elf) or not shutil.which("nm") or not shutil.which("objdump"):
        pytest.skip("reference ELF or binutils not available")
    syms = subprocess.check_output(["nm", "-n", elf]).decode().splitlines()
    addr = None
    for i, l in enumerate(syms):
        if "gamma_distributionIdEclINS0_23mersenne_twister_engine" in l:
            addr = int(l.split()[0], 16)
            nxt = next(int(m.split()[0], 16) for m in syms[i + 1:] if m.split()[0] != l.split()[0] and len(m.split()) == 3)
            break
    assert addr is not None, "gamma_distribution<double>::operator() not found in the ELF's symbol table"
    dis = subprocess.check_output(["objdump", "-d", "--no-show-raw-insn", "--start-address=0x%x" % addr, "--stop-address=0x%x" % nxt, elf]).decode()
    calls = re.findall(r"call\s+[0-9a-f]+ <([^>]+)>", dis)
    census = {name: sum(1 for c in calls if c == name) for name in ("tan", "log", "pow", "exp")}
    assert census == {"tan": 1, "log": 1, "pow": 1, "exp": 5}
    assert sum(1 for c in calls if "generate_int_float_pair" in c) == 2
    assert not [c for c in calls if c.split("@")[0] not in ("tan", "log", "pow", "exp") and "generate_int_float_pair" not in c]
    # its scalar double constants: 1, 2^-32 (uniform_01 on a 32-bit engine), the exponential Ziggurat's tail shift
    # table_x[1], pi (tan(pi * u)) and 2 (sqrt(2 alpha - 1))
    blob = open(elf, "rb").read()
    consts = sorted({struct.unpack("<d", blob[int(a, 16) - 0x400000:int(a, 16) - 0x400000 + 8])[0]
                     for a in re.findall(r"sd\s+0x[0-9a-f]+\(%rip\),%xmm\d+\s+# ([0-9a-f]+)", dis)})
    import orc as _orc
    L = _orc.load()
    n = C.c_int()
    ex = np.ctypeslib.as_array(L.orc_zig_table(2, C.byref(n)), shape=(257,))
    assert consts == sorted([1.0, 2.0 ** -32, float(ex[1]), 3.14159265358979323846, 2.0])


def test_shuffle_is_the_libstdcxx6_algorithm(oracle):
    """The marker order of a given seed must not depend on the host's libstdc++: oracle and product (the GPU parity
    tests compare their marker orders) restate the algorithm of the reference binary's toolchain, gcc 6.5 -- one
    uniform_int_distribution(0, i) draw per element with the classic down-scaling.  Checked against an
    independent Python statement of that algorithm on the MT19937 stream."""
    g = orc.OrcMt()
    oracle.orc_rng_seed(C.byref(g), 4321)
    v = np.arange(1000, dtype=np.int32)
    oracle.orc_rng_shuffle(C.byref(g), orc.iptr(v), 1000)
    g2 = orc.OrcMt()
    oracle.orc_rng_seed(C.byref(g2), 4321)
    w = list(range(1000))
    for i in range(1, 1000):
        scaling = 0xffffffff // (i + 1)
        past = (i + 1) * scaling
        while True:
            r = oracle.orc_rng_u32(C.byref(g2))
            if r < past:
                break
        j = r // scaling
        w[i], w[j] = w[j], w[i]
    assert list(v) == w and sorted(w) == list(range(1000))
    assert oracle.orc_rng_u32(C.byref(g)) == oracle.orc_rng_u32(C.byref(g2))  # same number of engine calls


def test_reference_elf_shuffle_is_the_one_draw_per_element_form():
    """Structural pin (ELF read as data): std::shuffle<vector<int>::iterator, mt19937&> in the reference binary
    divides twice and multiplies once around its inlined generator (scaling = range / n, past = n * scaling,
    index = draw / scaling) and the toolchain strings name gcc 6.5.0 -- the libstdc++ that predates the
    two-indices-per-draw shuffle (gcc 7) and the Lemire-style uniform_int_distribution (gcc 11)."""
    import re
    import shutil
    import subprocess
    elf = "/root/reference/src/hydra"
    if not os.path.exists(elf) or not shutil.which("nm") or not shutil.which("objdump") or not shutil.which("readelf"):
        pytest.skip("reference ELF or binutils not available")
    assert "GCC: (GNU) 6.5.0" in subprocess.check_output(["readelf", "-p", ".comment", elf]).decode()
    syms = subprocess.check_output(["nm", "-n", elf]).decode().splitlines()
    idx = next(i for i, l in enumerate(syms) if "_ZSt7shuffleIN9__gnu_cxx17__normal_iteratorIPiSt6vectorIiSaIiEEEE" in l)
    addr = int(syms[idx].split()[0], 16)
    dis = subprocess.check_output(["objdump", "-d", "--no-show-raw-insn", "--start-address=0x%x" % addr, "--stop-address=0x%x" % (addr + 0x5a0), elf]).decode()
    ops = [l.split("\t")[-1].split()[0] for l in dis.splitlines() if "\t" in l]
    first_call = next(i for i, l in enumerate(dis.splitlines()) if "uniform_int_distribution" in l)
    head = [l.split("\t")[-1] for l in dis.splitlines()[:first_call] if "\t" in l]
    divs = [h for h in head if h.startswith("div")]
    assert len(divs) == 2 and any(h.startswith("imul") and "$0x" not in h for h in head)
    assert ops.count("div") == 2


def test_reference_elf_beta_is_two_gammas_and_a_ratio():
    """Structural pin (ELF read as data) of beta_rng = x / (x + y), x ~ gamma(a, 1), y ~ gamma(b, 1)
    (orc_beta_rng; used once at start-up for sigmaG, src/BayesRRm.cpp:1233): two calls of the gamma sampler,
    three divisions (one e / (alpha + e) per gamma parameter set, then the ratio)."""
    import re
    import shutil
    import subprocess
    elf = "/root/reference/src/hydra"
    if not os.path.exists(elf) or not shutil.which("nm") or not shutil.which("objdump"):
        pytest.skip("reference ELF or binutils not available")
    syms = subprocess.check_output(["nm", "-n", elf]).decode().splitlines()
    addr = next(int(l.split()[0], 16) for l in syms if l.endswith("_ZN19Distributions_boost8beta_rngEdd"))
    dis = subprocess.check_output(["objdump", "-d", "--no-show-raw-insn", "--start-address=0x%x" % addr, "--stop-address=0x%x" % (addr + 0xe0), elf]).decode()
    body = dis[:dis.index("ret")]
    assert len(re.findall(r"call\s+[0-9a-f]+ <_ZN5boost6random18gamma_distributionIdEclI", body)) == 2
    assert len(re.findall(r"\bvdivsd\b", body)) == 3
