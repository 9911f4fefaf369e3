"""The C-ABI library loads without a GPU and exports every symbol
include/hgibbs.h declares; compute entry points fail loudly (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "hgibbs.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b((?:hgibbs|hydraw?)_[a-z0-9_]+)\s*\(", txt)))


def test_header_declares_what_the_binding_lists():
    from hydra_amd import capi
    assert sorted(capi.ABI_SYMBOLS) == header_symbols()


def test_library_exports_every_declared_symbol(gpu_lib):
    for name in header_symbols():
        assert hasattr(gpu_lib, name), name
    out = subprocess.check_output(["nm", "-D", "--defined-only", os.path.join(ROOT, "hydra_amd", "libhgibbs.so")]).decode()
    exported = set(re.findall(r" T ((?:hgibbs|hydraw?)_\w+)", out))
    assert set(header_symbols()) <= exported


def test_library_contains_gfx950_code_object():
    blob = open(os.path.join(ROOT, "hydra_amd", "libhgibbs.so"), "rb").read()
    assert b"gfx950" in blob and b"k_sweep_batch" in blob


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_fails_loudly_without_a_gpu(gpu_lib):
    from hydra_amd import capi
    with pytest.raises(capi.HgError) as e:
        capi.Device(0)
    assert "no CPU fallback" in str(e.value) or "hipGetDeviceCount" in str(e.value)


def test_product_never_touches_the_oracle():
    """only tests/, smoke() and bench.py's cpu_baseline leg may use oracle/"""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "hydra_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", "Makefile")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle/" not in src and "liboracle" not in src and "orc_" not in src, os.path.join(dirpath, f)


def test_cli_rejects_what_hydra_rejects(tmp_path):
    exe = os.path.join(ROOT, "hydra_amd", "bin", "hydra_mi355x")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 1 and "Did you forget to give the input parameters?" in r.stderr
    r = subprocess.run([exe, "--frobnicate"], capture_output=True, text=True)
    assert r.returncode == 1 and 'invalid option "--frobnicate"' in r.stderr
    r = subprocess.run([exe, "--mpibayes", "bayesMPI", "--bfile", "x", "--pheno", "y"], capture_output=True, text=True)
    assert r.returncode == 1 and "--mcmc-out-dir is mandatory" in r.stderr
    # missing --number-individuals is fatal (src/BayesRRm.cpp:3125-3128)
    from hydra_amd import synth
    geno = synth.make_genotypes(6, 12, seed=1)
    synth.write_plink(str(tmp_path / "d"), synth.pack_bed_columns(geno), 12, y=np.arange(12.0))
    r = subprocess.run([exe, "--mpibayes", "bayesMPI", "--bfile", str(tmp_path / "d"), "--pheno", str(tmp_path / "d.phen"),
                        "--mcmc-out-dir", str(tmp_path / "o"), "--mcmc-out-name", "t", "--number-markers", "6"],
                       capture_output=True, text=True)
    assert r.returncode == 1 and "opt.numberIndividuals is zero" in r.stderr


def test_inline_assembly_lds_reads_are_not_touched_before_their_wait(tmp_path):
    """The streaming loops read LDS from inline assembly (the compiler would otherwise drain every load in flight before an
    LDS read it emits itself) and place the lgkmcnt wait by hand: in the device assembly of the library no instruction may
    mention the destination registers of such a read before that wait (tools/asm_check.py)."""
    import shutil
    import subprocess
    import sys
    hipcc = shutil.which("hipcc")
    if not hipcc:
        pytest.skip("hipcc not on PATH")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tmp_path / "hgibbs.s"
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-S", "--cuda-device-only",
                           "-o", str(out), os.path.join(root, "hydra_amd", "csrc", "hgibbs.hip")], stderr=subprocess.DEVNULL)
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "asm_check.py"), str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
    assert "violations: 0" in r.stdout and "checked: 0," not in r.stdout
    # (same script: every LDS-DMA load of a load-issue block comes before the block's plain loads, which is what the loops' hand-placed
    # s_waitcnt vmcnt(N > 0) rely on)
    assert "plain loads checked: 0" not in r.stdout
    # the resident sweep kernels land their refill columns in hand-named registers (v216 .. v255; the second form of the streaming
    # workgroups v200 .. v255) by loads the compiler does not see: nothing it emits itself may name them, and the kernel body calls nothing
    # but the walkers (tools/asm_check_loads.py; hg_resident.hip.h, hg_streamer2.hip.h)
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "asm_check_loads.py"), str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
    assert "violations: 0" in r.stdout and "kernels checked: 20" in r.stdout  # two forms x T = 1, 2 (+ T = 4 of the second) x stage clocks on/off x missing-call build on/off
