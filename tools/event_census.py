#!/usr/bin/env python3
"""Per iteration: effects that are non-zero at sweep start (predicted events: certain to change), effects that
changed (nnz), and the difference = markers that went from zero to non-zero (surprises: they end a launch's chain).
usage: event_census.py CONFIG ITERS"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from hydra_amd import capi  # noqa: E402

cfg, iters = sys.argv[1], int(sys.argv[2])
N, M, G, mS = bench.CONFIGS[cfg]
dev = capi.Device(0)
dev.synth_bed(N, M, seed=42)
y = bench.make_phenotype_on_device(dev, N, M, (0, N), seed=43)
groups = None if G == 1 else (np.arange(M) % G).astype(np.int32)
ch = capi.Chain(dev, y, mS=np.array(mS), groups=groups)
for it in range(iters):
    nz0 = int(np.count_nonzero(dev.get_beta()[0]))
    ch.iterate()
    st = dev.sweep_stats()
    nnz = ch.last_nnz()
    print("it %d: non-zero at start %d, changed %d, surprises %d, launches %d, %.1f ms" % (it, nz0, nnz, nnz - nz0, st["launches"], st["device_ms"]), flush=True)
