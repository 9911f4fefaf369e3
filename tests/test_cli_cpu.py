"""Command-line front end, the part that runs before any device is touched: hydra's option names,
its mandatory options and its messages (src/options.cpp:7-326, src/main.cpp:17-195).  No GPU needed."""
import os
import subprocess

import numpy as np
import pytest

from hydra_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "hydra_amd", "bin", "hydra_mi355x")


def run(*args):
    return subprocess.run([EXE] + list(args), capture_output=True, text=True, timeout=60)


@pytest.fixture(scope="module")
def plink(tmp_path_factory):
    d = tmp_path_factory.mktemp("cli")
    geno = synth.make_genotypes(12, 30, seed=1)
    y, _ = synth.make_phenotype(geno, seed=2)
    synth.write_plink(str(d / "x"), synth.pack_bed_columns(geno), 30, y=y)
    return str(d / "x"), str(d)


def test_no_arguments():
    r = run()
    assert r.returncode == 1 and "Did you forget to give the input parameters?" in r.stderr


def test_invalid_option_is_named():
    r = run("--mpibayes", "bayesMPI", "--no-such-flag")
    assert r.returncode != 0 and 'invalid option "--no-such-flag"' in r.stderr


def test_out_of_scope_options_say_so():
    r = run("--mpibayes", "bayesMPI", "--sparse-dir", "x")
    assert r.returncode != 0 and "does not reproduce" in r.stderr


def test_mandatory_output_options():
    r = run("--mpibayes", "bayesMPI", "--bfile", "x", "--pheno", "y")
    assert r.returncode != 0 and "--mcmc-out-dir is mandatory" in r.stderr
    r = run("--mpibayes", "bayesMPI", "--bfile", "x", "--pheno", "y", "--mcmc-out-dir", "o")
    assert r.returncode != 0 and "--mcmc-out-name is mandatory" in r.stderr


def test_wrong_analysis_returns_zero_like_the_reference():
    r = run("--mpibayes", "bayesFHMPI", "--mcmc-out-dir", "o", "--mcmc-out-name", "n")
    assert r.returncode == 0 and "Wrong analysis requested" in r.stderr  # src/main.cpp:179-189 catches and returns 0


def test_group_files_come_in_pairs(plink):
    prefix, d = plink
    r = run("--mpibayes", "bayesMPI", "--bfile", prefix, "--pheno", prefix + ".phen", "--mcmc-out-dir", d, "--mcmc-out-name", "n",
            "--groupIndexFile", prefix + ".group")
    assert r.returncode != 0 and "both --groupIndexFile and --groupMixtureFile" in r.stderr


def test_input_checks_before_the_device(plink):
    prefix, d = plink
    base = ["--mpibayes", "bayesMPI", "--pheno", prefix + ".phen", "--mcmc-out-dir", d, "--mcmc-out-name", "n"]
    r = run(*base, "--bfile", prefix + "_missing", "--number-individuals", "30", "--number-markers", "12")
    assert r.returncode != 0 and "can not open the file" in r.stderr
    r = run(*base, "--bfile", prefix, "--number-markers", "12")
    assert r.returncode != 0 and "opt.numberIndividuals is zero" in r.stderr
    r = run(*base, "--bfile", prefix, "--number-individuals", "30")
    assert r.returncode != 0 and "opt.numberMarkers is zero" in r.stderr
    r = run(*base, "--bfile", prefix, "--number-individuals", "31", "--number-markers", "12")
    assert r.returncode != 0 and "does not match the .fam file" in r.stderr
    r = run(*base, "--bfile", prefix, "--number-individuals", "30", "--number-markers", "13")
    assert r.returncode != 0 and "exceeds the .bim file" in r.stderr
    r = run(*base, "--bfile", prefix, "--number-individuals", "30", "--number-markers", "12", "--S", "0.1,-1")
    assert r.returncode != 0 and "strictly positive" in r.stderr


def test_bayesw_needs_failure_and_quadrature(plink):
    prefix, d = plink
    base = ["--mpibayes", "bayesWMPI", "--bfile", prefix, "--pheno", prefix + ".phen", "--mcmc-out-dir", d, "--mcmc-out-name", "w",
            "--number-individuals", "30", "--number-markers", "12"]
    r = run(*base, "--quad_points", "9")
    assert r.returncode != 0 and "--failure is mandatory" in r.stderr
    np.savetxt(prefix + ".fail", np.ones(30, dtype=int), fmt="%d")
    r = run(*base, "--failure", prefix + ".fail")
    assert r.returncode != 0 and "Possible number of quad_points" in r.stderr


def test_without_a_gpu_the_run_fails_loudly(plink):
    """No CPU fallback: with valid inputs the first device call must refuse (on a GPU box this test is moot)."""
    import ctypes
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        n = ctypes.c_int(0)
        has_gpu = hip.hipGetDeviceCount(ctypes.byref(n)) == 0 and n.value > 0
    except OSError:
        has_gpu = False
    if has_gpu:
        pytest.skip("a GPU is present")
    prefix, d = plink
    r = run("--mpibayes", "bayesMPI", "--bfile", prefix, "--pheno", prefix + ".phen", "--mcmc-out-dir", d, "--mcmc-out-name", "n",
            "--number-individuals", "30", "--number-markers", "12", "--chain-length", "1", "--seed", "1")
    assert r.returncode != 0 and "hgibbs_create" in r.stderr


def test_option_file_replaces_the_command_line(plink, tmp_path):
    """--inp-file (src/options.cpp:8-11, 335-397): "key value" pairs, keys that start with # or // skip their value token,
    an unknown key is named; the parsed options reach the same input checks as the flags do."""
    prefix, d = plink
    f = tmp_path / "run.opt"
    f.write_text("bedFile %s\nphenotypeFile %s.phen\nanalysisType RAM\nbayesType bayesMPI\nmcmcOut %s/n\n"
                 "# comment\nnumberIndividuals 31\nnumberMarkers 12\nchainLength 1\nseed 1\nS 0.1,0.01\n" % (prefix, prefix, d))
    r = run("--inp-file", str(f))
    assert r.returncode != 0 and "does not match the .fam file" in r.stderr  # the file's numberIndividuals (31) was read and checked
    f.write_text("bedFile %s\nnoSuchKey 3\n" % prefix)
    r = run("--inp-file", str(f))
    assert r.returncode != 0 and "invalid option noSuchKey 3" in r.stderr
    r = run("--inp-file", str(tmp_path / "absent.opt"))
    assert r.returncode != 0 and "can not open the file" in r.stderr
