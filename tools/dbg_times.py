"""Diagnostic: stage timestamps of the sweep kernel's last launch (see hgibbs_debug_times)."""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hydra_amd import capi
import bench

cfg, batch, cpg = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
N, M, G, mS = bench.CONFIGS[cfg]
if len(sys.argv) > 4: M = int(sys.argv[4])
if os.environ.get("DBG_N"): N = int(os.environ["DBG_N"])
slices = int(sys.argv[5]) if len(sys.argv) > 5 else 0
ext = int(sys.argv[6]) if len(sys.argv) > 6 else -1
max_seg = int(sys.argv[7]) if len(sys.argv) > 7 else 0
iters = int(sys.argv[8]) if len(sys.argv) > 8 else 3
dev = capi.Device(0)
dev.set_option("batch", batch); dev.set_option("cols_per_group", cpg); dev.set_option("debug_timing", 1)
if slices: dev.set_option("slices", slices)
if ext >= 0: dev.set_option("ext_limit", ext)
if max_seg: dev.set_option("max_seg", max_seg)
for kv in filter(None, os.environ.get("DBG_OPTS", "").split(",")):  # library options, "name=value,..."
    dev.set_option(kv.split("=")[0], int(kv.split("=")[1]))
dev.synth_bed(N, M, seed=42, missing_rate=float(os.environ.get("DBG_MISSING", "0")))
y = bench.make_phenotype_on_device(dev, N, M, (0, N), seed=43)
ch = capi.Chain(dev, y, mS=np.array(mS))
L = capi.lib(); L.hgibbs_debug_times.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
for it in range(iters):
    ch.iterate()
    t = (C.c_uint64 * 48)(); L.hgibbs_debug_times(dev.h, t)
    t = [x for x in t]
    st = dev.sweep_stats()
    print(cfg, "it", it, "kern_us %.1f" % (st["kernel_ms_avg"] * 1e3), "launches", st["launches"],
          "work_launches", t[15], "accepted/launch %.1f" % (t[12] / max(1, t[15])),
          "avg stages_us: main+ticket %.2f reduce %.2f posterior %.2f walk %.2f" % tuple(t[8 + i] / 100.0 / max(1, t[15]) for i in range(4)),
          "| last arriver: skew %.2f loop %.2f reduce+drain %.2f ticket %.2f" % tuple(t[i] / 100.0 / max(1, t[15]) for i in (17, 13, 14, 16)),
          "sweep_ms %.1f" % st["device_ms"])
    n, n2 = max(1, t[15]), max(1, t[44])
    print("   draw phase us: stage %.2f | seg0 posterior %.2f walk %.2f | seg1 (%.0f%% of launches) posterior %.2f walk %.2f | plan+desc %.2f" % (
        t[36] / 100.0 / n, t[38] / 100.0 / n, t[39] / 100.0 / n, 100.0 * t[44] / n, t[41] / 100.0 / n2,
        t[42] / 100.0 / n2, t[43] / 100.0 / n))
