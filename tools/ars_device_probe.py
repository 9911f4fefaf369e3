"""The adaptive-rejection draw of a BayesW effect on ONE device lane against the same draw on the host (hg_ars.h compiles for both):
what continuing an event on the device would cost.  usage: ars_device_probe.py [ndraws]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from hydra_amd import capi  # noqa: E402

nd = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
dev = capi.Device(0)
L = capi.lib()
# a typical event of the w100k workload: alpha 10, one slab variance, sums of vi of the order of the number of failures
dens = np.array([10.0, 0.01, 3.0, 0.6, 0.8, 0.01, 60000.0, 30000.0, 8000.0], dtype=np.float64)
beta_old, safe = 0.0, 2.0 * np.sqrt(0.01 * 0.01)
us, ev, last = C.c_double(), C.c_double(), C.c_double()
capi.check(L.hgibbs_w_ars_device_probe(dev.h, dens.ctypes.data_as(C.POINTER(C.c_double)), beta_old, safe, 1234, nd, C.byref(us), C.byref(ev), C.byref(last)))
print("device, one lane: %.1f us per draw, %.1f density evaluations per draw (last draw %.6g)" % (us.value, ev.value, last.value))


def dens_py(x, _):
    a, sG, sf, sd, r, mv, v0, v1, v2 = dens
    return -a * x * sf - np.exp(a * x * r) * (v0 + v1 * np.exp(-a * x / sd) + v2 * np.exp(-2 * a * x / sd)) - x * x / (2 * mv * sG)


g = capi.GRand(1234) if hasattr(capi, "GRand") else None
if g is not None:
    t0 = time.perf_counter()
    n = 200
    for _ in range(n):
        capi.ars_sample(lambda x: dens_py(x, None), [beta_old - safe / 10, beta_old, beta_old + safe / 20, beta_old + safe / 10], beta_old - safe, beta_old + safe, g)
    print("host through ctypes callbacks (python density; an upper bound only): %.1f us per draw" % ((time.perf_counter() - t0) / n * 1e6))
