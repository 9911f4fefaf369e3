"""Where a round of the resident sweep engine spends its time (option debug_timing: 100 MHz stage clocks of the walker
and of streaming workgroup 0, accumulated over a sweep).  usage: res_anatomy.py [N M [iters [name=value ... | missing=rate | timing=0/1]]]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from hydra_amd import capi  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
M = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 4
dev = capi.Device(0)
dev.set_option("engine", 2)
timing = 1
missing = 0.0
for kv in sys.argv[4:]:
    k, _, v = kv.partition("=")
    if k == "timing":
        timing = int(v)
    elif k == "missing":
        missing = float(v)
    else:
        dev.set_option(k, int(v))
dev.synth_bed(N, M, seed=42, missing_rate=missing, row_begin=0, row_end=N)
y = bench.make_phenotype_on_device(dev, N, M, (0, N), seed=43, causal_frac=0.01)
ch = capi.Chain(dev, y, seed=1222, shuffle=1)
for it in range(iters):
    dev.set_option("debug_timing", timing if it == iters - 1 else 0)
    t0 = time.perf_counter()
    ch.iterate()
    dt = time.perf_counter() - t0
    s = dev.sweep_stats()
    print("[%.0f MHz] it %d: %.1f ms wall, %.1f ms device, %d rounds (%d events of which %d pivots, %d advances), %.2f us/round, %d chunks, %d refolds, %.3f M markers/s, drift %.2e"
          % (s["shader_mhz"], it, dt * 1e3, s["device_ms"], s["rounds"], s["events"], s["pivots"], s["advances"], s["device_ms"] * 1e3 / max(1, s["rounds"]),
             s["chunks"], s["refolds"], M / dt / 1e6, s["eps_sum_drift"]))
t = s["ticks"]
n = max(1, s["rounds"])
us = lambda x: x / 100.0 / n
print("walker per round (us): fold %.2f  collect %.2f  evaluate [terms %.2f  barrier %.2f  sums %.2f]  scan+draw %.2f  message+results+prefetch %.2f  | sum %.2f"
      % (us(t[0]), us(t[1]), us(t[5]), us(t[6]), us(t[2]), us(t[3]), us(t[4]), us(sum(t[0:7]))))
print("streaming workgroup 0 per round (us): wait %.2f  update %.2f  gram %.2f  refill dots %.2f  barrier %.2f  raw atomics + drain %.2f  barrier + count %.2f  prefetch issue %.2f | sum %.2f"
      % (us(t[8]), us(t[9]), us(t[10]), us(t[11]), us(t[12]), us(t[13]), us(t[14]), us(t[15]), us(sum(t[8:16]))))

if timing:
    tr = dev.resident_trace().astype(np.int64)
    nmsg = int(s["rounds"])
    idx = np.arange(max(2, nmsg - 4000), nmsg - 1) % 4096  # message numbers of the last rounds (seq = 1 .. rounds)
    W0, W1, W2, NC, S0, S1, S2, S3 = (tr[i][idx] for i in range(8))
    ev = S1 > W0  # messages that carried an update (stamps of this sweep)
    f = lambda x: "%.2f" % (np.mean(x) / 100.0)
    print("per message with an update (us, n=%d): message in flight %s | update %s | gram %s | stream %s | Gram atomics -> walker has them %s | walker: collect done -> next message %s | consumed %.1f"
          % (ev.sum(), f((S0 - W0)[ev]), f((S1 - S0)[ev]), f((S2 - S1)[ev]), f((S3 - S2)[ev]), f((W1 - S2)[ev]), f((W2 - W1)[ev]), np.mean(NC[ev])))
    # how long a round lasts, by what its message carried (stamps of consecutive messages)
    idn = (idx + 1) % 4096
    dur = (tr[0][idn] - W0) / 100.0
    okd = (dur > 0) & (dur < 1000)
    print("round = message to next message (us): rounds with an update %.2f (n=%d, %.1f positions), rounds that only advance %.2f (n=%d, %.1f positions)"
          % (np.mean(dur[ev & okd]), (ev & okd).sum(), np.mean(NC[ev & okd]), np.mean(dur[~ev & okd]) if (~ev & okd).any() else 0.0, (~ev & okd).sum(), np.mean(NC[~ev & okd]) if (~ev & okd).any() else 0.0))
    print("percentiles of message in flight (us):", np.percentile((S0 - W0)[ev] / 100.0, [5, 50, 95]).round(2), " of atomics -> walker:", np.percentile((W1 - S2)[ev] / 100.0, [5, 50, 95]).round(2))
if timing and s.get("walker") == 2:
    print("clock read: %.3f us each (64 reads back to back)" % (t[7] / 100.0 / 64.0))
    print("walker 2, per round (us): absorb %.2f  wait Gram / answers %.2f  wait dots %.2f  exact decision %.2f  what the round is (early advance, counts) %.2f  flow control %.2f  message+bookkeeping %.2f"
          % (us(t[5]), us(t[1]), us(t[0]), us(t[3]), us(t[6]), us(t[2]), us(t[4])))
