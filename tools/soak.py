#!/usr/bin/env python3
"""Soak: the same chain at a headline-sized shard under different launch geometries, many iterations.
Every variant must walk the same chain (components exact, beta to 1e-9 of the plain path's): the variants differ in what
runs concurrently inside a launch (carried columns / Gram-only groups, the ahead queue, slices, segments), so a rare
ordering bug in the hand-offs would show as a divergence at some iteration.
usage: soak.py [N] [M] [iterations] [missing rate]   (SOAK_VARIANTS=plain,resident keeps the variants whose name contains one of the words)"""
import sys
import os
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hydra_amd import capi
import bench

N = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
M = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
ITERS = int(sys.argv[3]) if len(sys.argv) > 3 else 40
MISSING = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
VARIANTS = [("plain: gram=0, batch 200", {"gram": 0, "batch": 200}),
            ("batch engine, defaults", {"engine": 1}),
            ("resident engine", {"engine": 2}),
            ("resident engine, first form of the streaming workgroups", {"engine": 2, "refill": 1}),
            ("resident engine, first walker, no announcements", {"engine": 2, "walker": 1}),
            ("resident engine, window 64, predicted pivots", {"engine": 2, "window": 64, "pivots": 1}),
            ("resident engine, window 128, 250 compute units", {"engine": 2, "window": 128, "res_cus": 250}),
            ("carry on, ahead 64", {"carry": 1, "ahead": 64}),
            ("four segments, carry on, ahead 128", {"max_seg": 4, "carry": 1, "ahead": 128}),
            ("8 slices, cols_per_group 8", {"slices": 8, "cols_per_group": 8})]
if os.environ.get("SOAK_VARIANTS"):  # comma-separated substrings of the variant names to keep (the first one kept is the reference)
    keep = [w.strip() for w in os.environ["SOAK_VARIANTS"].split(",") if w.strip()]
    VARIANTS = [v for v in VARIANTS if any(w in v[0] for w in keep)]
devs = []
for name, opts in VARIANTS:
    dev = capi.Device(0)
    for k, v in opts.items():
        dev.set_option(k, v)
    dev.synth_bed(N, M, seed=42, missing_rate=MISSING)
    y = bench.make_phenotype_on_device(dev, N, M, (0, N), seed=43)
    ch = capi.Chain(dev, y, mS=np.array([[0.0, 0.0001, 0.001, 0.01]]), seed=1222)
    devs.append((name, dev, ch))
t0 = time.time()
worst = 0.0
drift = 0.0  # |sum of eps at sweep end - the sum reduced at sweep start and held|, largest over variants and iterations
for it in range(ITERS):
    ref = None
    for name, dev, ch in devs:
        ch.iterate()
        drift = max(drift, dev.sweep_stats()["eps_sum_drift"])
        beta, comp, _ = dev.get_beta()
        st = ch.state()
        if ref is None:
            ref = (beta, comp, st["sigmaE"], ch.last_nnz())
            continue
        assert np.array_equal(comp, ref[1]), "iteration %d: components of '%s' differ from the plain path" % (it, name)
        err = float(np.max(np.abs(beta - ref[0]) / np.maximum(1.0, np.abs(ref[0]))))
        worst = max(worst, err)
        assert err <= 1e-9, "iteration %d: beta of '%s' off by %.3g" % (it, name, err)
        assert ch.last_nnz() == ref[3] and abs(st["sigmaE"] - ref[2]) <= 1e-9 * ref[2]
    if it % 5 == 4:
        print("iteration %d: %d variants agree (nnz %d, worst relative beta difference so far %.2e, eps-sum drift <= %.2e, %.0f s)" % (it + 1, len(devs), ref[3], worst, drift, time.time() - t0), flush=True)
print("SOAK OK: N=%d M=%d missing %g, %d iterations, %d variants: %s; eps-sum drift <= %.3e" % (N, M, MISSING, ITERS, len(devs), "; ".join(n for n, _, _ in devs), drift))
