#!/usr/bin/env python3
"""bench.py -- Gibbs markers/sec/iteration of the BayesRR hot path on MI355X.

A "step" is one full Gibbs iteration of BayesRRm::runMpiGibbs (mu, shuffle, the
sweep over all M markers, sigmaG/pi, sigmaE) on synthetic genotypes generated
directly in HBM.  The headline workload is the one
BASELINE.json's metric is quoted on: N = 500 000 individuals, M = 1 000 000
markers (125 GB of packed .bed resident in HBM); at --gpus G the individuals
are sharded G ways ("strong" scaling: total work fixed).

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nproc-per-node 8 ... bench.py --gpus 8 ...

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline`
and `cpu_baseline` objects.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (N, M, groups, mS)
    "c1": (5000, 10000, 1, [[0.0, 0.0001, 0.001, 0.01]]),
    "c2": (50000, 100000, 1, [[0.0, 0.0001, 0.001, 0.01]]),
    "c3": (200000, 500000, 2, [[0.0, 0.001, 0.01, 0.1], [0.0, 0.001, 0.01, 0.1]]),
    "c4": (500000, 1000000, 1, [[0.0, 0.0001, 0.001, 0.01]]),
}
# BayesW (Weibull survival, SURVEY.md 8f-1 / BASELINE config 5): name: (N, M, mS, quad_points)
BW_CONFIGS = {
    "c5": (5000, 10000, [[0.0, 0.001, 0.01]], 25),       # the shape of example/t_M10K_N_5K + Weibull.phen/.fail
    "w100k": (100000, 100000, [[0.0, 0.0001, 0.001, 0.01]], 9),
}
HBM_PEAK_GBPS = 8000.0  # MI355X nominal (MI355X_MICROARCH.md); ~6300 measured copy ceiling


def algorithmic_bytes(n_local, M, nnz):
    """BASELINE.md section 3 / SURVEY.md 8(d): per marker ceil(N_g/4) + 8 N_g,
    plus 16 N_g for every marker whose effect changed; per iteration 16 N_g + 20 M."""
    col = (n_local + 3) // 4
    return M * (col + 8 * n_local) + nnz * 16 * n_local + 16 * n_local + 20 * M


def algorithmic_bytes_bw(n, M, nnz, nshift):
    """BayesW, per marker: the packed column + vi (ceil(N/4) + 8N); a marker whose effect was not
    zero reads eps instead of vi (same count); every changed effect rewrites eps and vi
    (8N read + 16N write)."""
    col = (n + 3) // 4
    return M * (col + 8 * n) + nnz * 24 * n


def valu_floor_ms(n, M, refill=2, clock_hz=2.4e9, simds=1024):
    """The least time one sweep's decoding arithmetic can take on this design, one vector instruction per SIMD per four clocks
    (tools/ubench/valu_rate.hip measures 4.2-4.4).  First form of the streaming workgroups (refill = 1): three vector instructions per
    genotype-lane (extract, convert, multiply-add).  Second form (refill = 2, hg_streamer2.hip.h): a dword of sixteen genotypes costs a
    lane seven instructions to expand into an MFMA operand and two for the window's code form -- 9/16 of an instruction per genotype-lane
    (the matrix product itself runs beside them: 4 x 16 x 16 x 64 byte products per instruction)."""
    per_lane = 3.0 if refill == 1 else 9.0 / 16.0
    return per_lane * n * M / 64.0 * 4.0 / simds / clock_hz * 1e3


def committed_profile(kind, N, M, world, kernel="k_sweep_batch", missing=0.0):
    """A PMC summary committed under profiles/ (tools/profile_round.sh), newest round first, IF it was taken on this very
    workload (same N and M, one GPU); (dict, "file: command") or (None, None).  bench.py never presents such a figure
    without its source: the counters need rocprofv3 passes of their own and cannot be read inside this run."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_%s.json" % kind)), reverse=True):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("N") == N and d.get("M") == M and world == 1 and d.get("kernel", "k_sweep_batch") == kernel and float(d.get("missing_rate", 0.0)) == float(missing):
            return d, "%s: %s" % (os.path.relpath(f, ROOT), d.get("command", ""))
    return None, None


def launch_anatomy(dev, chain, dist=None):
    """One more (untimed) iteration on the build of the sweep kernel that carries stage timestamps (option
    debug_timing): where a working launch (batch engine) or a round of the walker (resident engine) spends its time, in
    microseconds, averaged over that iteration.  With several ranks the iteration is collective: the ranks first agree
    (a MIN over a flag) that every one of them switched builds, and none enters the iteration unless all did."""
    ok = 1
    try:
        dev.set_option("debug_timing", 1)
        dev.debug_times()  # clear
    except Exception as e:
        print("launch anatomy: this rank could not switch builds: %r" % (e,), file=sys.stderr)
        ok = 0
    if dist is not None:
        import torch
        flag = torch.tensor([ok], dtype=torch.int64)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        ok = int(flag[0])
    if not ok:
        dev.set_option("debug_timing", 0)
        raise RuntimeError("a rank could not switch to the build with stage clocks; anatomy skipped on every rank")
    try:
        chain.iterate()
        t = dev.debug_times()
        st = dev.sweep_stats()
    finally:
        dev.set_option("debug_timing", 0)
    if st["engine"] == 2:
        k = st["ticks"]
        n = max(1, int(st["rounds"]))
        us = lambda x: float(x) / 100.0 / n  # 100 MHz ticks
        return {"engine": "resident", "period_us": st["device_ms"] * 1e3 / n, "rounds": int(st["rounds"]), "events": int(st["events"]),
                "advances": int(st["advances"]), "walks_that_waited_for_dots": int(st["refolds"]), "shader_mhz": st["shader_mhz"],
                "walker_us": {"wait_for_raw_dots": us(k[0]), "wait_for_gram_terms": us(k[1]), "bound_test": us(k[2]),
                              "exact_event_and_draw": us(k[3]), "message_results_prefetch_fold": us(k[4])},
                "streaming_workgroup0_us": {"wait_for_message": us(k[8]), "update": us(k[9]), "gram_terms": us(k[10]), "refill_dots": us(k[11]),
                                            "barrier": us(k[12]), "raw_atomics_and_drain": us(k[13]), "count": us(k[14])},
                "note": "build with stage clocks (a few % slower than the timed one); walker and streaming workgroup 0"}
    n = max(1, int(t[15]))
    us = lambda x: float(x) / 100.0 / n  # 100 MHz ticks
    inside = us(t[8] + t[9] + t[10] + t[11])
    period = st["kernel_ms_avg"] * 1e3
    return {"period_us": period, "entry_and_streaming_loop_us": us(t[13]), "wave_reduction_and_drain_us": us(t[14]),
            "hand_off_tickets_us": us(t[16]) + us(t[9]), "draw_phase_us": us(t[10] + t[11]),
            "draw_detail_us": {"staging": us(t[36]), "seg0_posterior": us(t[38]), "seg0_walk": us(t[39]),
                               "later_segments": us(t[41] + t[42]), "plan_and_descriptor": us(t[43])},
            "launch_gap_and_skew_us": max(0.0, period - inside), "accepted_per_launch": float(t[12]) / n,
            "note": "build with stage timestamps (a few % slower than the timed one); last-arriving workgroup's view"}


def make_phenotype_on_device(dev, n_global, M, rank_rows, seed, h2=0.5, causal_frac=0.01):
    """y = X_std beta + e built with the product's own residual-update operator:
    eps <- e, then eps += beta_j x_j for the causal markers.  Returns y (all
    rows on every rank: the update is replayed on the local shard and gathered
    by the caller when sharded)."""
    rng = np.random.default_rng(seed)
    m_causal = max(1, int(round(M * causal_frac)))
    causal = rng.choice(M, size=m_causal, replace=False)
    beta = rng.normal(0.0, np.sqrt(h2 / m_causal), size=m_causal)
    e = rng.normal(0.0, np.sqrt(1.0 - h2), size=n_global)
    lo, hi = rank_rows
    dev.set_residual(e[lo:hi])
    for j, b in zip(causal, beta):
        dev.update_marker(int(j), -float(b))  # eps += b * x_j
    return dev.get_residual()


BULK = None  # transport of the bulk reductions, agreed on by all ranks at the first handle: "rccl" or "gloo"


def make_multirank_device(capi, dist, world, rank, local_rank, want_p2p):
    """One handle per rank: a communicator for the rare bulk reductions (RCCL; or
    gloo through the external hook when HGIBBS_BENCH_BULK=gloo, which is how two
    ranks can share one GPU in rehearsals) and, if asked for and available on
    every rank, the in-launch peer-mailbox exchange for the per-batch scalars.
    Every collective below is executed by every rank whatever fails locally.
    Returns (device, p2p_enabled)."""
    import torch

    def allreduce(arr):
        t = torch.from_numpy(arr.view(np.int64) if arr.dtype == np.uint64 else arr)
        dist.all_reduce(t)

    global BULK
    dev = capi.Device(local_rank)
    if BULK is None:
        BULK = os.environ.get("HGIBBS_BENCH_BULK", "rccl")
    if BULK == "rccl":
        # RCCL for the bulk reductions; if ANY rank cannot join the communicator, every rank falls back to gloo together
        # (and stays there for the rest of the run: the ranks must not disagree on the transport)
        ok, uid = 1, [None]
        try:
            if rank == 0:
                uid = [capi.Device.unique_id()]
        except Exception as e:
            print("rank 0: RCCL unique id failed: %r" % (e,), file=sys.stderr)
        dist.broadcast_object_list(uid, src=0)
        try:
            if uid[0] is None:
                raise RuntimeError("no RCCL unique id")
            dev.comm_init(world, rank, uid[0])
        except Exception as e:
            print("rank %d: hgibbs_comm_init (RCCL) failed: %r" % (rank, e), file=sys.stderr)
            ok = 0
        t = torch.tensor([ok], dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        if not int(t[0]):
            if rank == 0:
                print("RCCL communicator not available on every rank: bulk reductions over gloo (host) instead", file=sys.stderr)
            BULK = "gloo"
            dev.close()
            dev = capi.Device(local_rank)
    if BULK == "gloo":
        dev.comm_init_external(world, rank, allreduce)
    handle = None
    if want_p2p:
        try:
            handle = dev.p2p_export()
        except Exception as e:
            print("rank %d: p2p export failed: %r" % (rank, e), file=sys.stderr)
    handles = [None] * world
    dist.all_gather_object(handles, handle)
    ok = 0
    if want_p2p and all(h is not None for h in handles):
        try:
            dev.p2p_import(handles)
            ok = 1
        except Exception as e:
            print("rank %d: p2p import failed: %r" % (rank, e), file=sys.stderr)
    t = torch.tensor([ok], dtype=torch.int32)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    p2p = bool(int(t[0]))
    dev.set_option("p2p", 1 if p2p else 0)
    return dev, p2p


def exchange_self_check(capi, dist, world, rank, local_rank):
    """Run the same tiny sharded chain three ways -- the resident engine over the peer mailboxes, the batch engine over the peer
    mailboxes, the batch engine over the RCCL split path -- and decide what the timed run may use: the mailboxes only if the
    batch engine reproduces the RCCL path with them, the resident engine only if it then reproduces the batch engine
    (components exact, beta to 1e-9, on every rank).  Returns (use_p2p, use_resident)."""
    import torch
    Ns, Ms = 16384, 1500
    per = ((Ns + world - 1) // world + 3) // 4 * 4
    lo, hi = min(Ns, rank * per), min(Ns, (rank + 1) * per)
    y = np.random.default_rng(7).normal(size=Ns)
    out = {}
    for name, want_p2p, opts in (("batch-p2p", True, {"batch": 64}), ("resident-p2p", True, {"engine": 2}), ("batch-split", False, {"batch": 64})):
        if not want_p2p and BULK == "gloo":  # (decided by the first pass: no RCCL split path to compare with)
            break
        ok = 1
        res = None
        try:
            d, p2p = make_multirank_device(capi, dist, world, rank, local_rank, want_p2p)
            if want_p2p and not p2p:
                raise RuntimeError("p2p import failed on some rank")
            for k, v in opts.items():
                d.set_option(k, v)
            d.synth_bed(Ns, Ms, seed=5, row_begin=lo, row_end=hi)
            ch = capi.Chain(d, y, seed=99)
            for _ in range(2):
                ch.iterate()
            res = d.get_beta()[:2]
            d.close()
        except Exception as e:
            print("rank %d: exchange self-check (%s) failed: %r" % (rank, name, e), file=sys.stderr)
            ok = 0
        t = torch.tensor([ok], dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        out[name] = res if int(t[0]) else None

    def same(a, b):
        eq = a is not None and b is not None and np.array_equal(a[1], b[1]) and bool(np.all(np.abs(a[0] - b[0]) <= 1e-9 * np.maximum(1.0, np.abs(b[0]))))
        t = torch.tensor([1 if eq else 0], dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(int(t[0]))

    if "batch-split" in out and out["batch-split"] is not None:
        use_p2p = same(out["batch-p2p"], out["batch-split"])
    else:
        use_p2p = out["batch-p2p"] is not None  # the mailboxes are the transport that works
    use_resident = use_p2p and same(out["resident-p2p"], out["batch-p2p"])
    return use_p2p, use_resident


def cpu_model():
    try:
        lines = open("/proc/cpuinfo").read().splitlines()
        name = next(l.split(":", 1)[1].strip() for l in lines if l.startswith("model name"))
        sockets = len({l.split(":", 1)[1].strip() for l in lines if l.startswith("physical id")}) or 1
        return "%s, %d socket(s), %d logical CPUs visible" % (name, sockets, len(os.sched_getaffinity(0)))
    except Exception:
        return "unknown"


def reference_probe():
    """hydra itself is the preferred CPU baseline if Eigen 3.3.x and Boost >= 1.67 headers
    are on the box; probe, do not assume."""
    roots = ["/usr/include", "/usr/local/include", "/opt/conda/include", "/usr/include/eigen3", "/opt/rocm/include"]
    eigen = [r for r in roots if os.path.exists(os.path.join(r, "Eigen", "Eigen"))]
    boost = [r for r in roots if os.path.exists(os.path.join(r, "boost", "random.hpp"))]
    return "Eigen headers: %s; Boost.Random headers: %s => hydra itself %s be built here" % (
        eigen or "absent", boost or "absent", "could" if eigen and boost else "cannot")


def cpu_baseline(dev, y, N, M, mS, groups, sample_markers, threads):
    """The oracle (CPU restatement of hydra's path: LUT + AVX2 dot, OpenMP over
    individuals as the reference's loops are, its bookkeeping passes) timed on a
    bounded sample: the first `sample_markers` columns of the same genotype
    matrix, one Gibbs iteration after a warm-up one.  Runs in a child process
    (tools/cpu_baseline.py); falls back to the portable single-thread build."""
    import subprocess
    import tempfile
    ms = min(sample_markers, M)
    bed = dev.get_bed(0, ms)
    g = np.zeros(0, dtype=np.int32) if groups is None else np.ascontiguousarray(groups[:ms])
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "sample.npz")
        np.savez(path, bed=bed, y=y, N=np.array(N), groups=g, mS=np.array(mS))
        for lib_name, thr in (("liboracle_omp.so", threads), ("liboracle.so", 1)):
            try:
                out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "cpu_baseline.py"), path, lib_name, str(thr)],
                                     capture_output=True, text=True, timeout=900)
                if out.returncode == 0:
                    r = json.loads(out.stdout.strip().splitlines()[-1])
                    flags = "-O3 -march=native -fopenmp" if lib_name == "liboracle_omp.so" else "-O2"
                    return {"value": r["markers_per_s"], "unit": "markers/s", "cores": thr, "kind": "port", "cpu": cpu_model(),
                            "reference_probe": reference_probe(),
                            "sample": "restated hydra AVX2 path (LUT + _mm256 dot, OpenMP over individuals, reference's update "
                                      "bookkeeping passes), first %d of %d markers, N=%d, mean of 3 Gibbs iterations after 1 warm-up, %s, %s"
                                      % (ms, M, N, lib_name, flags)}
                print("cpu_baseline with %s failed (rc %d): %s" % (lib_name, out.returncode, out.stderr[-400:]), file=sys.stderr)
            except Exception as e:
                print("cpu_baseline with %s failed: %r" % (lib_name, e), file=sys.stderr)
    return None


def make_survival_on_device(dev, N, M, seed=44, h2=0.5, causal_frac=0.01, mu=4.1, alpha=10.0, censor_rate=0.1):
    """Weibull log-times shaped like example/Weibull.phen + .fail (mu 4.1, alpha 10, h2 0.5, 10% censored):
    genetic values from the product's own residual-update operator on the resident genotypes."""
    rng = np.random.default_rng(seed)
    m_causal = max(1, int(round(M * causal_frac)))
    causal = rng.choice(M, size=m_causal, replace=False)
    var_e = np.pi ** 2 / (6.0 * alpha ** 2)
    beta = rng.normal(0.0, np.sqrt(var_e * h2 / (1.0 - h2) / m_causal), size=m_causal)
    dev.set_residual(np.zeros(N))
    for j, b in zip(causal, beta):
        dev.update_marker(int(j), -float(b))
    gval = dev.get_residual()
    y = mu + gval + (np.log(rng.exponential(1.0, size=N)) + 0.577215664901532) / alpha
    fail = (rng.random(N) >= censor_rate).astype(np.int32)
    return np.where(fail == 1, y, y - rng.exponential(0.05, size=N)), fail


def cpu_baseline_bw(dev, y, fail, N, M, mS, quad, sample_markers):
    """BayesW: the oracle's restatement of hydra's sampler (single thread, -O2) on the first
    `sample_markers` columns, one iteration after a warm-up one, in a child process."""
    import subprocess
    import tempfile
    ms = min(sample_markers, M)
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "sample.npz")
        np.savez(path, bed=dev.get_bed(0, ms), y=y, fail=fail, N=np.array(N), mS=np.array(mS), quad=np.array(quad))
        try:
            out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "cpu_baseline.py"), path, "liboracle.so", "1", "bayesw"],
                                 capture_output=True, text=True, timeout=900)
            if out.returncode == 0:
                r = json.loads(out.stdout.strip().splitlines()[-1])
                return {"value": r["markers_per_s"], "unit": "markers/s", "cores": 1, "kind": "port", "cpu": cpu_model(),
                        "sample": "restated hydra BayesW sampler (sequential sums, libm exp, ARS), first %d of %d markers, N=%d, "
                                  "1 Gibbs iteration after 1 warm-up, liboracle.so, -O2" % (ms, M, N)}
            print("cpu_baseline (bayesw) failed (rc %d): %s" % (out.returncode, out.stderr[-400:]), file=sys.stderr)
        except Exception as e:
            print("cpu_baseline (bayesw) failed: %r" % (e,), file=sys.stderr)
    return None


def main_bayesw(args):
    """One GPU; a step = one full BayesW iteration (mu, alpha by ARS, vi, shuffle, sweep, sigmaG, pi)."""
    if int(os.environ.get("WORLD_SIZE", "1")) != 1 or args.gpus != 1:
        raise SystemExit("the BayesW workloads run on one GPU (SURVEY.md 8f-1)")
    from hydra_amd import capi
    N, M, mS, quad = BW_CONFIGS[args.config]
    N, M = args.N or N, args.M or M
    dev = capi.Device(int(os.environ.get("HGIBBS_BENCH_DEVICE", os.environ.get("LOCAL_RANK", "0"))))
    t_setup = time.perf_counter()
    dev.synth_bed(N, M, seed=42, missing_rate=args.missing)
    y, fail = make_survival_on_device(dev, N, M)
    if args.batch:
        dev.set_option("batch", args.batch)
    dev.set_option("w_kernel_timing", 1)
    chain = capi.BwChain(dev, y, fail, mS=np.array(mS), seed=1222, shuffle=1, quad=quad)
    t_setup = time.perf_counter() - t_setup
    for _ in range(args.warmup):
        chain.iterate()
    stats = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        chain.iterate()
        stats.append(chain.sweep_stats())
    dt = time.perf_counter() - t0
    K = args.steps
    launches = sum(s["launches"] for s in stats)
    nnz = sum(s["nnz_updates"] for s in stats)
    kernel_ms_avg = sum(s["sums_kernel_ms"] for s in stats) / max(1, launches)
    bytes_alg = sum(algorithmic_bytes_bw(N, M, s["nnz_updates"], 0) for s in stats)
    # the dominant kernel's share: the per-marker streaming reads (updates run in k_bw_refresh)
    bytes_sums = K * M * ((N + 3) // 4 + 8 * N)
    achieved = (bytes_sums / max(1, launches)) / (kernel_ms_avg * 1e-3) / 1e9 if kernel_ms_avg > 0 else 0.0
    out = {
        "metric": "Gibbs markers/sec/iter", "value": M * K / dt, "unit": "markers/s", "n_gpus": 1, "steps": K, "warmup": args.warmup,
        "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "BayesW %s: N=%d individuals x M=%d markers, K=%d mixture, %d quadrature points, .bed resident in HBM, 1 GPU"
                               % (args.config, N, M, len(mS[0]), quad),
                   "N": N, "M": M, "batch": args.batch or "auto", "nnz_updates_per_iter": nnz / K, "launches_per_iter": launches / K,
                   "ars_draws_per_iter": sum(s["ars_draws"] for s in stats) / K, "setup_s": t_setup},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                     "traffic": None, "kernel": "k_bw_sums", "kernel_ms_avg": kernel_ms_avg,
                     "algorithmic_bytes_per_launch": bytes_sums / max(1, launches),
                     "algorithmic_bytes_per_iter_all_kernels": bytes_alg / K,
                     "sweep_ms_per_iter": sum(s["device_ms"] for s in stats) / K},
    }
    if not args.no_cpu_baseline:
        sample = args.cpu_sample or max(64, min(M, int(1.0e9 / max(1, N))))
        out["cpu_baseline"] = cpu_baseline_bw(dev, y, fail, N, M, mS, quad, sample)
    else:
        out["cpu_baseline"] = None
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10, help="timed iterations (BASELINE.md section 4: iterations 2-11 after 2 warm-up ones)")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="c4", choices=sorted(CONFIGS) + sorted(BW_CONFIGS))
    ap.add_argument("--N", type=int, default=0, help="override individuals")
    ap.add_argument("--M", type=int, default=0, help="override markers")
    ap.add_argument("--batch", type=int, default=0, help="speculative batch width (0 = library default)")
    ap.add_argument("--cpg", type=int, default=0, help="columns per workgroup column-group")
    ap.add_argument("--graph", type=int, default=-1, help="replay the sweep's launches from a HIP graph (1) or launch them one by one (0)")
    ap.add_argument("--max-seg", type=int, default=0, help="segments (predicted events) one launch may chain through (0 = library default)")
    ap.add_argument("--missing", type=float, default=0.0)
    ap.add_argument("--causal-frac", type=float, default=0.01, help="share of markers with a simulated effect (SURVEY.md 8d: 1 %%)")
    ap.add_argument("--no-anatomy", action="store_true", help="skip the extra untimed iteration with stage timestamps")
    ap.add_argument("--exchange", default="auto", choices=["auto", "p2p", "rccl"],
                    help="per-batch cross-GPU exchange: in-launch peer mailboxes, RCCL all-reduce, or self-checked choice")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE", help="library option (hgibbs_set_option), repeatable: option scans")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=0, help="markers in the CPU baseline sample (0 = auto)")
    args = ap.parse_args()
    if args.config in BW_CONFIGS:
        return main_bayesw(args)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("HGIBBS_BENCH_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))

    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # control plane only.  The gloo library announces its connections on stdout: keep stdout for the one JSON line
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
            dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)

    from hydra_amd import capi

    N, M, G, mS = CONFIGS[args.config]
    if args.N:
        N = args.N
    if args.M:
        M = args.M
    groups = None if G == 1 else (np.arange(M) % G).astype(np.int32)

    # individuals sharded in multiples of 4 (byte-aligned column slices)
    per = ((N + world - 1) // world + 3) // 4 * 4
    lo, hi = min(N, rank * per), min(N, (rank + 1) * per)

    exchange = "none"
    want_p2p, resident_ok = False, True
    if world > 1:
        want_p2p = args.exchange in ("auto", "p2p") and os.environ.get("HGIBBS_DISABLE_P2P", "0") != "1"
        if want_p2p and args.exchange == "auto":
            want_p2p, resident_ok = exchange_self_check(capi, dist, world, rank, local_rank)

    def sync():
        if dist is not None:
            dist.barrier()

    def build_problem(force_batch_engine):
        """Handle, data and chain of the timed run (every collective in here is executed by every rank)."""
        nonlocal exchange
        if world > 1:
            d, p2p = make_multirank_device(capi, dist, world, rank, local_rank, want_p2p)
            exchange = "p2p-mailbox" if p2p else ("rccl-allreduce" if BULK == "rccl" else "gloo-allreduce (host)")
        else:
            d = capi.Device(local_rank)
        if force_batch_engine:
            d.set_option("engine", 1)
        if args.batch:
            d.set_option("batch", args.batch)
        if args.cpg:
            d.set_option("cols_per_group", args.cpg)
        if args.max_seg:
            d.set_option("max_seg", args.max_seg)
        if args.graph >= 0:
            d.set_option("graph", args.graph)
        for kv in args.opt:
            name, _, val = kv.partition("=")
            d.set_option(name, int(val))
        d.synth_bed(N, M, seed=42, missing_rate=args.missing, row_begin=lo, row_end=hi)
        y_loc = make_phenotype_on_device(d, N, M, (lo, hi), seed=43, causal_frac=args.causal_frac)
        if world > 1:
            parts = [None] * world
            dist.all_gather_object(parts, y_loc)
            yy = np.concatenate(parts)
        else:
            yy = y_loc
        return d, yy, capi.Chain(d, yy, mS=np.array(mS), groups=groups, seed=1222, shuffle=1)

    t_setup = time.perf_counter()
    dev, y, chain = build_problem(not resident_ok)  # (not resident_ok: the resident engine's cross-rank sums did not reproduce the batch engine's in the self-check)
    t_setup = time.perf_counter() - t_setup

    # warm-up.  Several ranks: if the sweep fails on ANY rank at the full size (the resident engine's exchange has been rehearsed on one
    # GPU only; its waits are bounded, a rank that loses its peers comes back with an error), every rank rebuilds the problem on the
    # batch engine and the run goes on there -- said on stderr and in config.exchange.
    for w in range(args.warmup):
        ok = 1
        try:
            chain.iterate()
        except Exception as e:
            if world == 1:
                raise
            print("rank %d: warm-up iteration %d failed: %r" % (rank, w, e), file=sys.stderr)
            ok = 0
        if world > 1:
            import torch
            flag = torch.tensor([ok], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if not int(flag[0]):
                if rank == 0:
                    print("a rank failed in the warm-up: the timed run uses the batch engine", file=sys.stderr)
                try:
                    dev.close()
                except Exception:
                    pass
                dev, y, chain = build_problem(True)
                exchange += " (batch engine after a failed warm-up)"
                for _ in range(args.warmup):
                    chain.iterate()
                break

    sync()
    stats, step_s = [], []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ts = time.perf_counter()
        chain.iterate()  # returns only after the sweep's stream has drained
        step_s.append(time.perf_counter() - ts)
        stats.append((dev.sweep_stats(), chain.last_nnz()))
    sync()
    dt = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0])

    # one more (untimed) iteration on the build with stage timestamps: every rank runs it (the chain is collective)
    anatomy_all = None
    if not args.no_anatomy:
        try:
            anatomy_all = launch_anatomy(dev, chain, dist)
        except Exception as e:
            print("launch anatomy not measured: %r" % (e,), file=sys.stderr)

    if rank == 0:
        K = args.steps
        ms_per_step = dt / K * 1e3
        value = M * K / dt
        n_local = hi - lo
        sweep_ms = sum(s["device_ms"] for s, _ in stats)
        enqueued = sum(s["launches"] for s, _ in stats)
        launches = sum(s["working_launches"] for s, _ in stats)  # launches that accepted markers or applied an update
        accepted = sum(s["accepted_markers"] for s, _ in stats)
        streamed = sum(s["streamed_columns"] for s, _ in stats)
        carried = sum(s["carried_columns"] for s, _ in stats)
        nnz = sum(n for _, n in stats)
        bytes_alg = sum(algorithmic_bytes(n_local, M, n) for _, n in stats)
        # HIP events on the sweep's stream around each sweep / working launches: the average period of a working launch
        # (back-to-back launches: duration + the gap between dependent launches; rocprofv3's average duration agrees, profiles/)
        engine = int(stats[-1][0].get("engine", 1))
        resident = engine == 2
        if resident:
            # one launch per sweep: the kernel's launch duration IS the sweep (HIP events around it on its stream); its rounds
            # (one per message of the walker) take the place of the batch engine's working launches in the per-launch figures
            rounds = launches
            launches = enqueued
        kernel_ms_avg = sweep_ms / max(1, launches)
        achieved = (bytes_alg / max(1, launches)) / (kernel_ms_avg * 1e-3) / 1e9
        refill = int(stats[-1][0].get("refill", 0))
        tiles = int(stats[-1][0].get("tiles_per_workgroup_max", 0))
        kname = (("k_sweep_limb4" if tiles == 4 else "k_sweep_limb") if refill == 2 else "k_sweep_resident") if resident else "k_sweep_batch"
        traffic, traffic_src = committed_profile("traffic", N, M, world, kname, args.missing)
        valu, valu_src = committed_profile("valu", N, M, world, kname, args.missing)
        anatomy = anatomy_all  # measured by every rank together (below the timed region), reported by rank 0
        col_bytes = (n_local + 3) // 4
        # bytes that MUST cross the HBM boundary per sweep: every column once; the batch engine also reads and writes eps once per update
        # (the resident engine holds eps in registers for the whole sweep: one read at its start, one write at its end)
        compulsory = M * K * col_bytes + ((2 * K * 8 * n_local) if resident else (nnz * 16 * n_local))
        roof = {"bound": "latency", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                "traffic": traffic["traffic_bytes_per_launch"] if traffic else None, "traffic_source": traffic_src,
                "kernel": kname, "engine": "resident" if resident else "batch", "kernel_ms_avg": kernel_ms_avg,
                "algorithmic_bytes_per_launch": bytes_alg / max(1, launches),
                "sweep_ms_per_iter": sweep_ms / K,
                # the section-8(d) byte model counts eps once per MARKER from HBM; the batch engine reads it once per column group from
                # L2 / Infinity Cache, the resident engine keeps it in registers: `frac` exceeds 1 and bounds nothing.  What does, each with its distance:
                "compulsory_bytes_per_launch": compulsory / max(1, launches),
                "hbm_frac_compulsory": compulsory / (sweep_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                "hbm_frac_measured": (traffic["traffic_bytes_per_launch"] / (kernel_ms_avg * 1e-3) / 1e9 / HBM_PEAK_GBPS) if traffic else None,
                "valu_issue_frac_whole_launch": valu["valu_issue_frac"] if valu else None,
                "valu_source": valu_src,
                # what the decoding arithmetic costs at the very least (valu_floor_ms above): a SIMD issues ONE vector instruction per four
                # clocks however many waves it holds.  The first form of the refill (3 instructions per genotype-lane) was bound by it; the
                # second (integer matrix products, 9/16 of an instruction) is not -- the chain of rounds is
                "valu_floor_ms": valu_floor_ms(n_local, M, refill),
                "frac_of_valu_floor": valu_floor_ms(n_local, M, refill) / (sweep_ms / K) if sweep_ms > 0 else None,
                "refill_form": refill,
                "working_launches_per_iter": launches / K, "enqueued_launches_per_iter": enqueued / K,
                "accepted_per_launch": accepted / max(1, launches),
                "columns_streamed_per_accepted": streamed / max(1, accepted),
                "columns_carried_per_accepted": carried / max(1, accepted)}
        if resident:
            roof["rounds_per_iter"] = rounds / K
            roof["us_per_round"] = sweep_ms * 1e3 / max(1, rounds)
            roof["markers_per_round"] = accepted / max(1, rounds)
            roof["events_per_iter"] = sum(s["events"] for s, _ in stats) / K
            roof["advances_per_iter"] = sum(s["advances"] for s, _ in stats) / K
            # census of why rounds end: an event at a marker whose effect was non-zero at sweep start (certain to change), an event
            # that came unannounced, or the window ran out (advance); rounds whose walk had to wait for dots still on their way
            roof["census_per_iter"] = {"predicted_events": sum(s["predicted"] for s, _ in stats) / K,
                                       "unannounced_events": sum(s["events"] - s["predicted"] for s, _ in stats) / K,
                                       "window_ran_out": sum(s["advances"] for s, _ in stats) / K,
                                       "walks_that_waited_for_dots": sum(s["refolds"] for s, _ in stats) / K}
            roof["eps_sum_drift_max"] = max(s["eps_sum_drift"] for s, _ in stats)
            if anatomy:
                roof["anatomy_us"] = anatomy
            roof["bound_statement"] = ("latency of the exact sequential chain: one round per event (message -> eps update + Gram terms on every compute unit -> "
                                       "memory-side atomic adds -> walker), %.1f us per round, %.0f markers per round; HBM carries %.1f %% of its peak "
                                       "(compulsory column bytes), the refill's arithmetic overlaps the walker" % (
                                           roof["us_per_round"], roof["markers_per_round"], 100 * roof["hbm_frac_compulsory"]))
        elif anatomy:
            loop = anatomy["entry_and_streaming_loop_us"]
            roof["anatomy_us"] = anatomy
            roof["fixed_us_per_launch"] = anatomy["period_us"] - loop
            roof["streaming_share_of_launch"] = loop / anatomy["period_us"] if anatomy["period_us"] else None
            if valu and loop > 0:
                # the VALU instructions of a launch are issued almost entirely inside the streaming loop (the draw phase is one
                # workgroup): issue-slot share while streaming = whole-launch share x period / loop time
                roof["valu_frac_streaming"] = valu["valu_issue_frac"] * anatomy["period_us"] / loop
        if not resident:
          roof["bound_statement"] = ("latency + VALU issue: HBM carries %s of its peak; the streaming loop (%s of a launch) runs at %s of the VALU issue rate; "
                                   "the rest of a launch is the serial hand-off and draw of one workgroup" % (
                                       ("%.1f %%" % (100 * roof["hbm_frac_measured"])) if roof["hbm_frac_measured"] else "%.1f %% (compulsory bytes)" % (100 * roof["hbm_frac_compulsory"]),
                                       ("%.0f %%" % (100 * roof["streaming_share_of_launch"])) if anatomy else "n/a",
                                       ("%.0f %%" % (100 * roof["valu_frac_streaming"])) if roof.get("valu_frac_streaming") else "n/a"))
        out = {
            "metric": "Gibbs markers/sec/iter",
            "value": value,
            "unit": "markers/s",
            "n_gpus": world,
            "steps": K,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "ms_per_step_best": min(step_s) * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "BayesRR %s: N=%d individuals x M=%d markers, K=%d mixture, G=%d groups, "
                                   ".bed resident in HBM, individuals sharded over %d GPU(s)"
                                   % (args.config, N, M, len(mS[0]), G, world),
                       "N": N, "M": M, "batch": args.batch or "auto", "exchange": exchange, "bulk_reductions": (BULK or "none"), "nnz_updates_per_iter": nnz / K,
                       "launches_per_iter": enqueued / K, "working_launches_per_iter": launches / K,
                       "accepted_per_launch": accepted / max(1, launches), "columns_streamed_per_accepted": streamed / max(1, accepted),
                       "carried_columns_per_iter": carried / K, "causal_frac": args.causal_frac,
                       "tiles_per_workgroup": [min(s["tiles_per_workgroup_min"] for s, _ in stats), max(s["tiles_per_workgroup_max"] for s, _ in stats)],
                       "setup_s": t_setup,
                       **({"multi_gpu_note": "no scaling curve has been measured on hardware (the pool gives one GPU); DESIGN.md section 5: the chain's "
                                             "round is latency, not streaming -- sharding the individuals buys capacity (shards of 522 K individuals per GPU), "
                                             "predicted 0.7-0.8x of one GPU at 2-8 GPUs (one GPU: 6.7 us a round; a shard: 5.9 us + the exchange)"} if world > 1 else {}),
                       **({"missing_rate": args.missing} if args.missing else {}), **({"options": args.opt} if args.opt else {})},
            "roofline": roof,
        }
        try:  # measured streaming ceiling beside the nominal peak: 2 GiB device copy, far beyond the Infinity Cache
            ceiling = dev.stream_ceiling(2 << 30, 10)
            out["roofline"]["measured_stream_GBps"] = ceiling
            out["roofline"]["frac_of_measured_stream"] = achieved / ceiling
        except Exception as e:
            print("stream ceiling not measured: %r" % (e,), file=sys.stderr)
        if not args.no_cpu_baseline and world == 1:
            threads = max(1, min(len(os.sched_getaffinity(0)), 16))  # the GPU box's CPU share for one GPU
            sample = args.cpu_sample or max(64, min(M, int(4.0e9 / max(1, N))))
            out["cpu_baseline"] = cpu_baseline(dev, y, N, M, mS, groups, sample, threads)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
