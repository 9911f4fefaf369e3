"""BASELINE configs 1, 3 and 5 at their own size on the reference's shipped example inputs
(tests/golden/example: t_M10K_N_5K.fam/.bim, normal.phen, normal.group/.mS, Weibull.phen/.fail,
copied by tests/golden/make_golden.py).  The genotype file of the example is not in the reference
checkout, so a seeded synthetic 5000 x 10000 .bed stands in; everything else is read from the
reference's own files through the command line, and compared with the oracle."""
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

import orc
from hydra_amd import synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "hydra_amd", "bin", "hydra_mi355x")
EX = os.path.join(ROOT, "tests", "golden", "example")
N, M = 5000, 10000


def close(a, b, tol):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.all(np.abs(a - b) <= tol * np.maximum(1.0, np.abs(b)))


def _hist(path, dtype):
    raw = open(path, "rb").read()
    assert struct.unpack("<I", raw[:4])[0] == M
    rec = 4 + M * np.dtype(dtype).itemsize
    return np.array([np.frombuffer(raw[8 + k * rec:4 + (k + 1) * rec], dtype=dtype) for k in range((len(raw) - 4) // rec)])


@pytest.fixture(scope="module")
def example(tmp_path_factory):
    assert open(os.path.join(EX, "t_M10K_N_5K.dim")).read().split() == [str(N), str(M)]
    d = tmp_path_factory.mktemp("example")
    geno = synth.make_genotypes(M, N, seed=2020, missing_rate=0.001)
    bed = synth.pack_bed_columns(geno)
    with open(d / "t_M10K_N_5K.bed", "wb") as f:
        f.write(bytes([0x6c, 0x1b, 0x01]))
        f.write(bed.tobytes())
    for f in ("t_M10K_N_5K.fam", "t_M10K_N_5K.bim"):
        shutil.copyfile(os.path.join(EX, f), d / f)
    return str(d / "t_M10K_N_5K"), bed, str(d)


def _phen(name):
    return np.array([float(l.split()[2]) for l in open(os.path.join(EX, name))])


@pytest.mark.parametrize("grouped", [False, True])
def test_config1_and_3_bayesr_on_the_shipped_phenotype(oracle, example, grouped):
    prefix, bed, d = example
    out, iters = os.path.join(d, "out_r%d" % grouped), 4
    cmd = [EXE, "--mpibayes", "bayesMPI", "--bfile", prefix, "--pheno", os.path.join(EX, "normal.phen"), "--mcmc-out-dir", out,
           "--mcmc-out-name", "normal", "--number-individuals", str(N), "--number-markers", str(M), "--chain-length", str(iters),
           "--thin", "1", "--save", "2", "--seed", "1222", "--shuf-mark", "1"]
    if grouped:  # srun_groups.sh of the reference: normal.group + normal.mS
        cmd += ["--groupIndexFile", os.path.join(EX, "normal.group"), "--groupMixtureFile", os.path.join(EX, "normal.mS")]
        groups = np.loadtxt(os.path.join(EX, "normal.group"), dtype=np.int32)
        mS = np.array([[0.0, 0.001, 0.01, 0.1], [0.0, 0.001, 0.01, 0.1]])
    else:
        cmd += ["--S", "0.0001,0.001,0.01"]
        groups, mS = None, np.array([[0.0, 0.0001, 0.001, 0.01]])
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    ref = orc.Chain(oracle, bed, N, _phen("normal.phen"), groups=groups, mS=mS, seed=1222)
    betas, comps = _hist(out + "/normal.bet", np.float64), _hist(out + "/normal.cpn", np.int32)
    csv = open(out + "/normal.csv").read().splitlines()
    assert len(betas) == iters and len(csv) == iters
    for it in range(iters):
        ref.iterate()
        assert np.array_equal(comps[it], ref.arr("components")) and close(betas[it], ref.arr("beta"), 1e-9)
        assert close([float(x) for x in csv[it].split(",")], [float(x) for x in ref.csv_line(it).split(",")], 1e-9)


def test_config5_bayesw_on_the_shipped_weibull_files(oracle, example):
    prefix, bed, d = example
    out, iters = os.path.join(d, "out_w"), 3
    fail = np.loadtxt(os.path.join(EX, "Weibull.fail"), dtype=np.int32)
    assert set(np.unique(fail)) <= {0, 1} and fail.shape == (N,)
    cmd = [EXE, "--mpibayes", "bayesWMPI", "--bfile", prefix, "--pheno", os.path.join(EX, "Weibull.phen"), "--failure",
           os.path.join(EX, "Weibull.fail"), "--quad_points", "25", "--mcmc-out-dir", out, "--mcmc-out-name", "weibull",
           "--number-individuals", str(N), "--number-markers", str(M), "--chain-length", str(iters), "--thin", "1", "--save", "2",
           "--seed", "1222", "--S", "0.001,0.01"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    ref = orc.BwChain(oracle, bed, N, _phen("Weibull.phen"), fail, mS=np.array([[0.0, 0.001, 0.01]]), seed=1222, quad=25)
    betas, comps = _hist(out + "/weibull.bet", np.float64), _hist(out + "/weibull.cpn", np.int32)
    csv = open(out + "/weibull.csv").read().splitlines()
    assert len(betas) == iters
    for it in range(iters):
        ref.iterate()
        assert np.array_equal(comps[it], ref.arr("components")) and close(betas[it], ref.arr("beta"), 1e-8)
        assert close([float(x) for x in csv[it].split(",")], [float(x) for x in ref.csv_line(it).split(",")], 1e-8)
        if it == 2:
            ref.reseed_ars(1222 + 2)
    mu = float(csv[-1].split(",")[1])
    assert 3.9 < mu < 4.3  # Weibull.h2 says mu = 4.1; the intercept is identified whatever the genotypes are
