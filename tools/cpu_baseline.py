#!/usr/bin/env python3
"""cpu_baseline leg of bench.py, run as a CHILD process (a crash here -- e.g. an
illegal instruction from a -march=native build made on another CPU -- must not
take the bench down).  Reads a .npz with the sample (bed columns, y, model),
times one Gibbs iteration of the oracle's "restated hydra AVX2 path" after one
warm-up iteration, prints one JSON line."""
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main_bayesw(sample, lib_name):
    import orc
    L = orc.load(lib_name)
    d = np.load(sample)
    ch = orc.BwChain(L, d["bed"], int(d["N"]), d["y"], d["fail"], mS=d["mS"], seed=1222, shuffle=1, quad=int(d["quad"]))
    ch.iterate()
    t0 = time.perf_counter()
    ch.iterate()
    dt = time.perf_counter() - t0
    print(json.dumps({"markers_per_s": d["bed"].shape[0] / dt, "seconds": dt, "lib": lib_name, "threads": 1}))


def main():
    sample, lib_name, threads = sys.argv[1], sys.argv[2], int(sys.argv[3])
    if len(sys.argv) > 4 and sys.argv[4] == "bayesw":
        return main_bayesw(sample, lib_name)
    if lib_name == "liboracle_omp.so":  # build for THIS host's CPU
        subprocess.check_call(["make", "-B", "-C", os.path.join(ROOT, "oracle"), "--quiet",
                               os.path.join(ROOT, "oracle", "liboracle_omp.so")])
    import orc
    L = orc.load(lib_name)
    d = np.load(sample)
    N = int(d["N"])
    groups = d["groups"] if d["groups"].size else None
    L.orc_set_threads(threads)
    L.orc_set_dot_form(2)  # the reference's dense LUT/AVX2 loop structure incl. its bookkeeping passes
    ch = orc.Chain(L, d["bed"], N, d["y"], groups=groups, mS=d["mS"], seed=1222, shuffle=1)
    ch.iterate()
    reps = 3  # a few timed iterations: the box's host cores are shared, single iterations scatter by +-20 %
    t0 = time.perf_counter()
    for _ in range(reps):
        ch.iterate()
    dt = (time.perf_counter() - t0) / reps
    print(json.dumps({"markers_per_s": d["bed"].shape[0] / dt, "seconds": dt, "lib": lib_name, "threads": threads, "iterations": reps}))


if __name__ == "__main__":
    main()
