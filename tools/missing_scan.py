#!/usr/bin/env python3
"""Sweep rate against the fraction of columns that carry missing calls, with and without the
four-term Gram build (option gram_missing).  usage: missing_scan.py N M"""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from hydra_amd import capi, synth  # noqa: E402

N, M = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(1)
p = rng.uniform(0.05, 0.5, size=M)
geno = rng.binomial(2, p[:, None], size=(M, N)).astype(np.uint8)
y, _ = synth.make_phenotype(geno, seed=2, causal_frac=0.01)
for frac in (0.0, 0.1, 0.5, 1.0):
    g = geno.copy()
    cols = rng.choice(M, size=int(frac * M), replace=False)
    for c in cols:
        g[c, rng.random(N) < 0.01] = 3
    bed = synth.pack_bed_columns(g)
    for mg in (0, 1):
        dev = capi.Device(0)
        dev.load_bed(bed, N)
        dev.set_option("gram_missing", mg)
        dev.set_option("max_seg", 2)
        ch = capi.Chain(dev, y, seed=1222)
        for _ in range(3):
            ch.iterate()
        t = time.time()
        L = 0
        for _ in range(4):
            ch.iterate()
            L += dev.sweep_stats()["launches"]
        dt = (time.time() - t) / 4
        print("missing columns %.0f%%  gram_missing %d: %.0f markers/s  %.1f ms/iter  launches %d" % (100 * frac, mg, M / dt, dt * 1e3, L // 4), flush=True)
        dev.close()
