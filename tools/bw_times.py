#!/usr/bin/env python3
"""Time BayesW iterations on synthetic data generated on the device.
usage: bw_times.py N M [iters] [batch] [quad]"""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from hydra_amd import capi  # noqa: E402


def make_survival_on_device(dev, N, M, seed=44, h2=0.5, causal_frac=0.01, mu=3.0, alpha=4.0, censor_rate=0.3):
    rng = np.random.default_rng(seed)
    m_causal = max(1, int(round(M * causal_frac)))
    causal = rng.choice(M, size=m_causal, replace=False)
    var_e = np.pi ** 2 / (6.0 * alpha ** 2)
    var_g = var_e * h2 / (1.0 - h2)
    beta = rng.normal(0.0, np.sqrt(var_g / m_causal), size=m_causal)
    dev.set_residual(np.zeros(N))
    for j, b in zip(causal, beta):
        dev.update_marker(int(j), -float(b))  # eps += b * x_j (standardised column)
    gval = dev.get_residual()
    w = np.log(rng.exponential(1.0, size=N)) + 0.577215664901532
    y = mu + gval + w / alpha
    fail = (rng.random(N) >= censor_rate).astype(np.int32)
    y = np.where(fail == 1, y, y - rng.exponential(0.2, size=N))
    return y, fail


def main():
    N, M = int(sys.argv[1]), int(sys.argv[2])
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    batch = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    quad = int(sys.argv[5]) if len(sys.argv) > 5 else 9
    dev = capi.Device(0)
    t = time.time()
    dev.synth_bed(N, M, seed=42)
    y, fail = make_survival_on_device(dev, N, M)
    print("data ready in %.1f s" % (time.time() - t), flush=True)
    if batch:
        dev.set_option("batch", batch)
    ch = capi.BwChain(dev, y, fail, mS=np.array([[0.0, 0.0001, 0.001, 0.01]]), seed=1222, quad=quad)
    for it in range(iters):
        t = time.time()
        ch.iterate()
        dt = time.time() - t
        st, ss = ch.state(), ch.sweep_stats()
        print("it %d: %.3f s  sweep %.1f ms  launches %d  nnz %d  ars %d/%d  m0 %d  mu %.4f alpha %.4f sigmaG %.5f  -> %.0f markers/s" % (
            it, dt, ss["device_ms"], ss["launches"], ss["nnz_updates"], ss["ars_draws"], ss["ars_evals"], int(st["m0"].sum()), st["mu"], st["alpha"],
            st["sigmaG"].sum(), M / dt), flush=True)


if __name__ == "__main__":
    main()
