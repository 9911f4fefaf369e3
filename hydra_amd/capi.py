"""ctypes binding of the C ABI in include/hgibbs.h (libhgibbs.so, HIP/gfx950).

This is plumbing only: every compute entry point lives in the shared library.
There is no CPU fallback -- if the library is missing or no gfx950 device is
visible the calls raise.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libhgibbs.so")

# every symbol include/hgibbs.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = [
    "hgibbs_last_error", "hgibbs_version", "hgibbs_create", "hgibbs_destroy", "hgibbs_comm_unique_id",
    "hgibbs_comm_init", "hgibbs_comm_init_external", "hgibbs_p2p_export", "hgibbs_p2p_import", "hgibbs_load_bed", "hgibbs_synth_bed", "hgibbs_dims", "hgibbs_get_bed",
    "hgibbs_marker_stats", "hgibbs_set_residual", "hgibbs_get_residual", "hgibbs_reduce_eps", "hgibbs_add_scalar",
    "hgibbs_update_marker", "hgibbs_dot_marker", "hgibbs_set_covariates", "hgibbs_cov_dot", "hgibbs_cov_update",
    "hydra_chain_set_covariates", "hydra_chain_gamma", "hgibbs_set_components", "hydra_chain_restore",
    "hydra_rng_to_boost_words", "hydra_rng_from_boost_words", "hgibbs_set_model", "hgibbs_set_beta", "hgibbs_get_beta",
    "hgibbs_beta_sqnorm", "hgibbs_sweep", "hgibbs_set_option", "hgibbs_last_sweep_stats", "hgibbs_stream_ceiling", "hgibbs_debug_times", "hgibbs_resident_trace", "hydra_chain_create",
    "hydra_chain_destroy", "hydra_chain_iterate", "hydra_chain_state", "hydra_chain_csv_line", "hydra_chain_order",
    "hydra_chain_last_nnz",
    # BayesW
    "hgibbs_grand_seed", "hgibbs_grand_next", "hgibbs_ars_sample", "hgibbs_w_init", "hgibbs_w_marker_stats", "hgibbs_w_set_model",
    "hgibbs_w_reduce", "hgibbs_w_refresh_vi", "hgibbs_w_get_vi", "hgibbs_w_marker_sums", "hgibbs_w_sweep", "hgibbs_w_last_sweep_stats", "hgibbs_w_ars_device_probe",
    "hgibbs_w_get_beta", "hgibbs_w_set_beta", "hydraw_chain_create", "hydraw_chain_destroy", "hydraw_chain_set_covariates",
    "hydraw_chain_reseed_ars", "hydraw_chain_iterate", "hydraw_chain_state", "hydraw_chain_gamma", "hydraw_chain_order",
    "hydraw_chain_last_nnz", "hydraw_chain_csv_line", "hydraw_chain_restore",
]


class RngState(C.Structure):
    _fields_ = [("x", C.c_uint32 * 624), ("idx", C.c_uint32)]


class SweepStats(C.Structure):
    _fields_ = [("launches", C.c_uint64), ("nnz_updates", C.c_uint64), ("device_ms", C.c_double),
                ("kernel_ms_avg", C.c_double), ("carried_columns", C.c_uint64), ("working_launches", C.c_uint64),
                ("accepted_markers", C.c_uint64), ("streamed_columns", C.c_uint64), ("tiles_per_workgroup_min", C.c_uint32),
                ("tiles_per_workgroup_max", C.c_uint32), ("engine", C.c_uint32), ("walker", C.c_uint32), ("eps_sum_drift", C.c_double),
                ("rounds", C.c_uint64), ("events", C.c_uint64), ("advances", C.c_uint64), ("chunks", C.c_uint64), ("refolds", C.c_uint64), ("pivots", C.c_uint64), ("predicted", C.c_uint64), ("shader_mhz", C.c_double),
                ("ticks", C.c_uint64 * 16), ("refill", C.c_uint32), ("reserved_", C.c_uint32)]


class RestartState(C.Structure):
    _fields_ = [("iteration", C.c_uint32), ("sigmaE", C.c_double), ("mu", C.c_double),
                ("sigmaG", C.POINTER(C.c_double)), ("estPi", C.POINTER(C.c_double)), ("beta", C.POINTER(C.c_double)),
                ("components", C.POINTER(C.c_int32)), ("eps", C.POINTER(C.c_double)), ("order", C.POINTER(C.c_int32)),
                ("gamma", C.POINTER(C.c_double)), ("xI", C.POINTER(C.c_int32)), ("rng", RngState)]


class GrandState(C.Structure):
    _fields_ = [("r", C.c_int32 * 31), ("f", C.c_int32), ("b", C.c_int32)]


class WSweepStats(C.Structure):
    _fields_ = [("launches", C.c_uint64), ("nnz_updates", C.c_uint64), ("ars_draws", C.c_uint64), ("ars_evals", C.c_uint64),
                ("device_ms", C.c_double), ("sums_kernel_ms", C.c_double)]


class WModelDesc(C.Structure):
    _fields_ = [("seed", C.c_uint32), ("shuffle", C.c_int32), ("G", C.c_int32), ("K", C.c_int32),
                ("groups", C.POINTER(C.c_int32)), ("mS", C.POINTER(C.c_double)), ("quad_points", C.c_int32)]


class WRestartState(C.Structure):
    _fields_ = [("iteration", C.c_uint32), ("mu", C.c_double), ("alpha", C.c_double), ("sigmaG", C.POINTER(C.c_double)),
                ("pi", C.POINTER(C.c_double)), ("beta", C.POINTER(C.c_double)), ("components", C.POINTER(C.c_int32)),
                ("eps", C.POINTER(C.c_double)), ("order", C.POINTER(C.c_int32)), ("gamma", C.POINTER(C.c_double)),
                ("xI", C.POINTER(C.c_int32)), ("rng", RngState), ("ars_seed", C.c_uint32)]


LOGDENS_FN = C.CFUNCTYPE(C.c_double, C.c_double, C.c_void_p)


class ModelDesc(C.Structure):
    _fields_ = [("seed", C.c_uint32), ("shuffle", C.c_int32), ("G", C.c_int32), ("K", C.c_int32),
                ("groups", C.POINTER(C.c_int32)), ("mS", C.POINTER(C.c_double))]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int)


class HgError(RuntimeError):
    pass


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("HGIBBS_LIB", LIB_PATH)  # A/B runs of two builds of the same ABI
    if not os.path.exists(path):
        raise HgError("%s not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                      "(there is no CPU fallback)" % path)
    L = C.CDLL(path)
    vp, dp, ip = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int32)
    u8p, u64p, u32p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)
    L.hgibbs_last_error.restype = C.c_char_p
    L.hgibbs_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.hgibbs_destroy.argtypes = [vp]
    L.hgibbs_comm_unique_id.argtypes = [C.c_void_p]
    L.hgibbs_comm_init.argtypes = [vp, C.c_int, C.c_int, C.c_void_p]
    L.hgibbs_comm_init_external.argtypes = [vp, C.c_int, C.c_int, ALLREDUCE_FN, C.c_void_p]
    L.hgibbs_p2p_export.argtypes = [vp, C.c_void_p]
    L.hgibbs_p2p_import.argtypes = [vp, C.c_void_p]
    L.hgibbs_load_bed.argtypes = [vp, u8p, C.c_uint64, C.c_uint32, C.c_uint32, u8p, C.c_uint32, C.c_uint32, C.c_uint32]
    L.hgibbs_synth_bed.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_double]
    L.hgibbs_dims.argtypes = [vp, u32p, u32p, u32p, u32p]
    L.hgibbs_get_bed.argtypes = [vp, C.c_uint32, C.c_uint32, u8p, C.c_uint64]
    L.hgibbs_marker_stats.argtypes = [vp, dp, dp, u64p, u64p, u64p]
    L.hgibbs_set_residual.argtypes = [vp, dp]
    L.hgibbs_get_residual.argtypes = [vp, dp]
    L.hgibbs_reduce_eps.argtypes = [vp, dp, dp]
    L.hgibbs_add_scalar.argtypes = [vp, C.c_double]
    L.hgibbs_update_marker.argtypes = [vp, C.c_uint32, C.c_double]
    L.hgibbs_dot_marker.argtypes = [vp, C.c_uint32, dp]
    L.hgibbs_set_covariates.argtypes = [vp, dp, C.c_int]
    L.hgibbs_cov_dot.argtypes = [vp, C.c_int, C.c_double, dp]
    L.hgibbs_cov_update.argtypes = [vp, C.c_int, C.c_double]
    L.hydra_chain_set_covariates.argtypes = [vp, dp, C.c_int]
    L.hydra_chain_gamma.argtypes = [vp, dp, ip]
    L.hgibbs_set_components.argtypes = [vp, ip]
    L.hydra_chain_restore.argtypes = [vp, C.POINTER(RestartState)]
    L.hydra_rng_to_boost_words.argtypes = [C.POINTER(RngState), u32p]
    L.hydra_rng_from_boost_words.argtypes = [u32p, C.POINTER(RngState)]
    L.hgibbs_set_model.argtypes = [vp, C.c_int, C.c_int, ip, dp, dp]
    L.hgibbs_set_beta.argtypes = [vp, dp]
    L.hgibbs_get_beta.argtypes = [vp, dp, ip, dp]
    L.hgibbs_beta_sqnorm.argtypes = [vp, dp]
    L.hgibbs_sweep.argtypes = [vp, ip, C.c_double, dp, dp, u8p, C.POINTER(RngState), ip, u64p]
    L.hgibbs_set_option.argtypes = [vp, C.c_char_p, C.c_int64]
    L.hgibbs_last_sweep_stats.argtypes = [vp, C.POINTER(SweepStats)]
    L.hgibbs_stream_ceiling.argtypes = [vp, C.c_uint64, C.c_int, dp]
    L.hgibbs_debug_times.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.hgibbs_resident_trace.argtypes = [vp, C.POINTER(C.c_uint64), C.c_uint64]
    L.hydra_chain_create.argtypes = [vp, C.POINTER(ModelDesc), dp, C.POINTER(vp)]
    L.hydra_chain_destroy.argtypes = [vp]
    L.hydra_chain_iterate.argtypes = [vp]
    L.hydra_chain_state.argtypes = [vp, dp, dp, dp, dp, ip, ip, C.POINTER(RngState)]
    L.hydra_chain_csv_line.argtypes = [vp, C.c_uint32, C.c_char_p, C.c_size_t]
    L.hydra_chain_order.argtypes = [vp]
    L.hydra_chain_order.restype = ip
    L.hydra_chain_last_nnz.argtypes = [vp]
    gp = C.POINTER(GrandState)
    L.hgibbs_grand_seed.argtypes = [gp, C.c_uint32]
    L.hgibbs_grand_seed.restype = None
    L.hgibbs_grand_next.argtypes = [gp]
    L.hgibbs_grand_next.restype = C.c_int32
    L.hgibbs_ars_sample.argtypes = [dp, C.c_double, C.c_double, LOGDENS_FN, C.c_void_p, gp, dp, ip]
    L.hgibbs_w_init.argtypes = [vp, ip]
    L.hgibbs_w_marker_stats.argtypes = [vp, dp, dp, dp]
    L.hgibbs_w_set_model.argtypes = [vp, C.c_int, C.c_int, ip, dp, C.c_int]
    L.hgibbs_w_reduce.argtypes = [vp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, dp]
    L.hgibbs_w_refresh_vi.argtypes = [vp, C.c_double]
    L.hgibbs_w_get_vi.argtypes = [vp, dp, dp]
    L.hgibbs_w_marker_sums.argtypes = [vp, C.c_uint32, C.c_double, C.c_double, dp, dp, dp]
    L.hgibbs_w_sweep.argtypes = [vp, ip, C.c_double, dp, dp, C.c_double, C.POINTER(RngState), gp, ip, dp, u64p]
    L.hgibbs_w_last_sweep_stats.argtypes = [vp, C.POINTER(WSweepStats)]
    L.hgibbs_w_ars_device_probe.argtypes = [vp, dp, C.c_double, C.c_double, C.c_uint32, C.c_uint32, dp, dp, dp]
    L.hgibbs_w_get_beta.argtypes = [vp, dp, ip]
    L.hgibbs_w_set_beta.argtypes = [vp, dp, ip]
    L.hydraw_chain_create.argtypes = [vp, C.POINTER(WModelDesc), dp, ip, C.POINTER(vp)]
    L.hydraw_chain_destroy.argtypes = [vp]
    L.hydraw_chain_set_covariates.argtypes = [vp, dp, C.c_int]
    L.hydraw_chain_reseed_ars.argtypes = [vp, C.c_uint32]
    L.hydraw_chain_iterate.argtypes = [vp]
    L.hydraw_chain_state.argtypes = [vp, dp, dp, dp, dp, ip, ip, C.POINTER(RngState), gp]
    L.hydraw_chain_gamma.argtypes = [vp, dp, ip]
    L.hydraw_chain_order.argtypes = [vp]
    L.hydraw_chain_order.restype = C.POINTER(C.c_int32)
    L.hydraw_chain_last_nnz.argtypes = [vp]
    L.hydraw_chain_last_nnz.restype = C.c_uint64
    L.hydraw_chain_csv_line.argtypes = [vp, C.c_uint32, C.c_char_p, C.c_size_t]
    L.hydraw_chain_restore.argtypes = [vp, C.POINTER(WRestartState)]
    L.hydra_chain_last_nnz.restype = C.c_uint64
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise HgError(lib().hgibbs_last_error().decode(errors="replace"))


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _u8(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def _u64(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint64))


class Device:
    """One GPU's share of the problem (hgibbs_t)."""

    def __init__(self, device_id=0):
        self.L = lib()
        self.h = C.c_void_p()
        check(self.L.hgibbs_create(device_id, C.byref(self.h)))
        self.M = self.n_local = self.n_global = self.row_begin = 0
        self.G = self.K = 0

    def close(self):
        if self.h:
            self.L.hgibbs_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- distributed --
    @staticmethod
    def unique_id():
        buf = (C.c_uint8 * 128)()
        check(lib().hgibbs_comm_unique_id(buf))
        return bytes(buf)

    def comm_init(self, nranks, rank, uid):
        buf = (C.c_uint8 * 128).from_buffer_copy(uid) if uid else None
        check(self.L.hgibbs_comm_init(self.h, nranks, rank, buf))

    def comm_init_external(self, nranks, rank, allreduce):
        """allreduce(numpy_array) must sum the array in place over all ranks."""
        def cb(user, buf, count, dtype):
            try:
                ty = C.c_double if dtype == 0 else C.c_uint64
                arr = np.ctypeslib.as_array(C.cast(buf, C.POINTER(ty)), shape=(count,))
                allreduce(arr)
                return 0
            except Exception:  # pragma: no cover - reported through the return code
                import traceback
                traceback.print_exc()
                return 1
        self._cb = ALLREDUCE_FN(cb)  # keep alive
        check(self.L.hgibbs_comm_init_external(self.h, nranks, rank, self._cb, None))

    def p2p_export(self):
        buf = (C.c_uint8 * 64)()
        check(self.L.hgibbs_p2p_export(self.h, buf))
        return bytes(buf)

    def p2p_import(self, handles):
        blob = b"".join(handles)
        buf = (C.c_uint8 * len(blob)).from_buffer_copy(blob)
        check(self.L.hgibbs_p2p_import(self.h, buf))

    # -- data --
    def _dims(self):
        a, b, c, d = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
        check(self.L.hgibbs_dims(self.h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        self.n_global, self.n_local, self.M, self.row_begin = a.value, b.value, c.value, d.value

    def load_bed(self, bed, n_total, keep=None, row_begin=0, row_end=None, n_global=None):
        bed = np.ascontiguousarray(bed, dtype=np.uint8)
        M, stride = bed.shape
        kept = int(np.count_nonzero(keep)) if keep is not None else n_total
        if row_end is None:
            row_end = kept
        if n_global is None:
            n_global = kept
        kp = _u8(np.ascontiguousarray(keep, dtype=np.uint8)) if keep is not None else None
        check(self.L.hgibbs_load_bed(self.h, _u8(bed), stride, n_total, M, kp, row_begin, row_end, n_global))
        self._dims()

    def synth_bed(self, n_global, M, seed=42, missing_rate=0.0, row_begin=0, row_end=None):
        if row_end is None:
            row_end = n_global
        check(self.L.hgibbs_synth_bed(self.h, n_global, M, row_begin, row_end, seed, missing_rate))
        self._dims()

    def get_bed(self, m0=0, mcount=None):
        if mcount is None:
            mcount = self.M - m0
        out = np.zeros((mcount, (self.n_local + 3) // 4), dtype=np.uint8)
        check(self.L.hgibbs_get_bed(self.h, m0, mcount, _u8(out), out.shape[1]))
        return out

    def marker_stats(self):
        M = self.M
        mave, mstd = np.zeros(M), np.zeros(M)
        n1, n2, nm = (np.zeros(M, dtype=np.uint64) for _ in range(3))
        check(self.L.hgibbs_marker_stats(self.h, _dp(mave), _dp(mstd), _u64(n1), _u64(n2), _u64(nm)))
        return mave, mstd, n1, n2, nm

    def set_residual(self, eps):
        eps = np.ascontiguousarray(eps, dtype=np.float64)
        assert eps.shape[0] == self.n_local
        check(self.L.hgibbs_set_residual(self.h, _dp(eps)))

    def get_residual(self):
        out = np.zeros(self.n_local)
        check(self.L.hgibbs_get_residual(self.h, _dp(out)))
        return out

    def reduce_eps(self):
        s, q = C.c_double(), C.c_double()
        check(self.L.hgibbs_reduce_eps(self.h, C.byref(s), C.byref(q)))
        return s.value, q.value

    def add_scalar(self, c):
        check(self.L.hgibbs_add_scalar(self.h, c))

    def set_covariates(self, X):
        X = np.ascontiguousarray(X, dtype=np.float64)
        assert X.shape[0] == self.n_local
        check(self.L.hgibbs_set_covariates(self.h, _dp(X), X.shape[1]))

    def update_marker(self, marker, dbeta):
        check(self.L.hgibbs_update_marker(self.h, marker, dbeta))

    def dot_marker(self, marker):
        v = C.c_double()
        check(self.L.hgibbs_dot_marker(self.h, marker, C.byref(v)))
        return v.value

    def set_model(self, groups, cVa, cVaI):
        cVa = np.ascontiguousarray(cVa, dtype=np.float64)
        cVaI = np.ascontiguousarray(cVaI, dtype=np.float64)
        G, K = cVa.shape
        g = _ip(np.ascontiguousarray(groups, dtype=np.int32)) if groups is not None else None
        check(self.L.hgibbs_set_model(self.h, G, K, g, _dp(cVa), _dp(cVaI)))
        self.G, self.K = G, K

    def set_beta(self, beta):
        beta = np.ascontiguousarray(beta, dtype=np.float64)
        check(self.L.hgibbs_set_beta(self.h, _dp(beta)))

    def get_beta(self):
        beta, comp, acum = np.zeros(self.M), np.zeros(self.M, dtype=np.int32), np.zeros(self.M)
        check(self.L.hgibbs_get_beta(self.h, _dp(beta), _ip(comp), _dp(acum)))
        return beta, comp, acum

    def beta_sqnorm(self):
        out = np.zeros(self.G)
        check(self.L.hgibbs_beta_sqnorm(self.h, _dp(out)))
        return out

    def stream_ceiling(self, nbytes=2 << 30, reps=10):
        out = C.c_double()
        check(self.L.hgibbs_stream_ceiling(self.h, nbytes, reps, C.byref(out)))
        return out.value

    def set_option(self, name, value):
        check(self.L.hgibbs_set_option(self.h, name.encode(), int(value)))

    def debug_times(self):
        """The 48 stage-timestamp words of the sweep kernel's debug build (option debug_timing; 100 MHz ticks, accumulated
        over the launches since the last call, which clears them); layout in hg_sweep.hip.h (sweep_draw_phase)."""
        t = (C.c_uint64 * 48)()
        check(self.L.hgibbs_debug_times(self.h, t))
        return [int(x) for x in t]

    def sweep(self, order, sigmaE, sigmaG, estPi, adaV, rng):
        """rng: RngState, updated in place.  Returns (cass[G,K], nnz_updates)."""
        order = np.ascontiguousarray(order, dtype=np.int32)
        sigmaG = np.ascontiguousarray(sigmaG, dtype=np.float64)
        estPi = np.ascontiguousarray(estPi, dtype=np.float64)
        adaV = np.ascontiguousarray(adaV, dtype=np.uint8)
        cass = np.zeros((self.G, self.K), dtype=np.int32)
        nnz = C.c_uint64()
        check(self.L.hgibbs_sweep(self.h, _ip(order), float(sigmaE), _dp(sigmaG), _dp(estPi), _u8(adaV), C.byref(rng),
                                  _ip(cass), C.byref(nnz)))
        return cass, nnz.value

    def resident_trace(self):
        out = np.zeros((10, 4096), dtype=np.uint64)
        check(self.L.hgibbs_resident_trace(self.h, out.ctypes.data_as(C.POINTER(C.c_uint64)), out.size))
        return out

    def sweep_stats(self):
        s = SweepStats()
        check(self.L.hgibbs_last_sweep_stats(self.h, C.byref(s)))
        return {"launches": s.launches, "nnz_updates": s.nnz_updates, "device_ms": s.device_ms,
                "kernel_ms_avg": s.kernel_ms_avg, "carried_columns": s.carried_columns, "working_launches": s.working_launches,
                "accepted_markers": s.accepted_markers, "streamed_columns": s.streamed_columns,
                "tiles_per_workgroup_min": s.tiles_per_workgroup_min, "tiles_per_workgroup_max": s.tiles_per_workgroup_max,
                "engine": s.engine, "walker": s.walker, "refill": s.refill, "eps_sum_drift": s.eps_sum_drift, "rounds": s.rounds, "events": s.events,
                "advances": s.advances, "chunks": s.chunks, "refolds": s.refolds, "pivots": s.pivots, "predicted": s.predicted, "shader_mhz": s.shader_mhz, "ticks": list(s.ticks)}


class Chain:
    """hydra_chain_t: the runMpiGibbs body on top of a loaded Device."""

    def __init__(self, dev, y, mS=None, groups=None, seed=1222, shuffle=1):
        self.dev = dev
        self.L = dev.L
        if mS is None:
            mS = np.array([[0.0, 0.0001, 0.001, 0.01]])
        self.mS = np.ascontiguousarray(mS, dtype=np.float64)
        self.G, self.K = self.mS.shape
        self.groups = None if groups is None else np.ascontiguousarray(groups, dtype=np.int32)
        y = np.ascontiguousarray(y, dtype=np.float64)
        assert y.shape[0] == dev.n_global
        d = ModelDesc(seed, shuffle, self.G, self.K, _ip(self.groups) if self.groups is not None else None, _dp(self.mS))
        self.h = C.c_void_p()
        check(self.L.hydra_chain_create(dev.h, C.byref(d), _dp(y), C.byref(self.h)))
        dev.G, dev.K = self.G, self.K

    def __del__(self):
        try:
            if self.h:
                self.L.hydra_chain_destroy(self.h)
        except Exception:
            pass

    def set_covariates(self, X):
        X = np.ascontiguousarray(X, dtype=np.float64)
        assert X.shape[0] == self.dev.n_global
        self.C = X.shape[1]
        check(self.L.hydra_chain_set_covariates(self.h, _dp(X), self.C))

    def gamma(self):
        g, xi = np.zeros(self.C), np.zeros(self.C, dtype=np.int32)
        check(self.L.hydra_chain_gamma(self.h, _dp(g), _ip(xi)))
        return g, xi

    def iterate(self):
        check(self.L.hydra_chain_iterate(self.h))

    def state(self):
        G, K = self.G, self.K
        sE, mu = C.c_double(), C.c_double()
        sG, pi = np.zeros(G), np.zeros((G, K))
        m0, cass = np.zeros(G, dtype=np.int32), np.zeros((G, K), dtype=np.int32)
        rng = RngState()
        check(self.L.hydra_chain_state(self.h, C.byref(sE), C.byref(mu), _dp(sG), _dp(pi), _ip(m0), _ip(cass), C.byref(rng)))
        return {"sigmaE": sE.value, "mu": mu.value, "sigmaG": sG, "estPi": pi, "m0": m0, "cass": cass,
                "rng_x": np.array(rng.x, dtype=np.uint32), "rng_idx": int(rng.idx)}

    def order(self):
        return np.ctypeslib.as_array(self.L.hydra_chain_order(self.h), shape=(self.dev.M,)).copy()

    def rng_words(self):
        """dist.rng in Boost's stream form (what hydra's .rng.<rank> holds)."""
        rng = RngState()
        check(self.L.hydra_chain_state(self.h, None, None, None, None, None, None, C.byref(rng)))
        out = np.zeros(624, dtype=np.uint32)
        check(self.L.hydra_rng_to_boost_words(C.byref(rng), out.ctypes.data_as(C.POINTER(C.c_uint32))))
        return out

    def restore(self, iteration, sigmaE, mu, sigmaG, estPi, beta, components, eps, order, rng_words, gamma=None, xI=None):
        """init_from_restart: eps = this rank's rows; the chain continues at iteration + 1."""
        f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
        keep = [f64(sigmaG), f64(estPi), f64(beta), i32(components), f64(eps), i32(order)]
        st = RestartState()
        st.iteration, st.sigmaE, st.mu = iteration, sigmaE, mu
        st.sigmaG, st.estPi, st.beta = _dp(keep[0]), _dp(keep[1]), _dp(keep[2])
        st.components, st.eps, st.order = _ip(keep[3]), _dp(keep[4]), _ip(keep[5])
        if gamma is not None:
            keep += [f64(gamma), i32(xI)]
            st.gamma, st.xI = _dp(keep[6]), _ip(keep[7])
        w = np.ascontiguousarray(rng_words, dtype=np.uint32)
        check(self.L.hydra_rng_from_boost_words(w.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(st.rng)))
        check(self.L.hydra_chain_restore(self.h, C.byref(st)))

    def last_nnz(self):
        return int(self.L.hydra_chain_last_nnz(self.h))

    def csv_line(self, it):
        buf = C.create_string_buffer(50000)
        n = self.L.hydra_chain_csv_line(self.h, it, buf, 50000)
        return buf.raw[:n].decode()


def ars_sample(logdens, xinit, xl, xr, grand):
    """One ARS draw from exp(logdens) on [xl, xr] (host code; no GPU involved). Returns (err, x, neval)."""
    L = lib()
    cb = LOGDENS_FN(lambda x, _d: logdens(x))
    xi = np.ascontiguousarray(xinit, dtype=np.float64)
    out, ne = C.c_double(0.0), C.c_int(0)
    err = L.hgibbs_ars_sample(_dp(xi), xl, xr, cb, None, C.byref(grand), C.byref(out), C.byref(ne))
    return err, out.value, ne.value


class BwOps:
    """BayesW operators of a loaded Device (hgibbs_w_*)."""

    def __init__(self, dev, failure):
        self.dev, self.L = dev, dev.L
        self.fail = np.ascontiguousarray(failure, dtype=np.int32)
        assert self.fail.shape[0] == dev.n_global
        check(self.L.hgibbs_w_init(dev.h, _ip(self.fail)))

    def marker_stats(self):
        M = self.dev.M
        a, b, c = np.zeros(M), np.zeros(M), np.zeros(M)
        check(self.L.hgibbs_w_marker_stats(self.dev.h, _dp(a), _dp(b), _dp(c)))
        return a, b, c

    def set_model(self, mS, groups=None, quad=9):
        mS = np.ascontiguousarray(mS, dtype=np.float64)
        self.G, self.K = mS.shape
        g = None if groups is None else np.ascontiguousarray(groups, dtype=np.int32)
        check(self.L.hgibbs_w_set_model(self.dev.h, self.G, self.K, _ip(g) if g is not None else None, _dp(mS), quad))

    def reduce(self, kind, col=0, p0=0.0, p1=0.0, p2=0.0):
        out = C.c_double()
        check(self.L.hgibbs_w_reduce(self.dev.h, kind, col, p0, p1, p2, C.byref(out)))
        return out.value

    def refresh_vi(self, alpha):
        check(self.L.hgibbs_w_refresh_vi(self.dev.h, alpha))

    def get_vi(self):
        v, s = np.zeros(self.dev.n_local), C.c_double()
        check(self.L.hgibbs_w_get_vi(self.dev.h, _dp(v), C.byref(s)))
        return v, s.value

    def marker_sums(self, marker, beta_old, alpha):
        a, b, c = C.c_double(), C.c_double(), C.c_double()
        check(self.L.hgibbs_w_marker_sums(self.dev.h, marker, beta_old, alpha, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def get_beta(self):
        b, c = np.zeros(self.dev.M), np.zeros(self.dev.M, dtype=np.int32)
        check(self.L.hgibbs_w_get_beta(self.dev.h, _dp(b), _ip(c)))
        return b, c

    def sweep_stats(self):
        st = WSweepStats()
        check(self.L.hgibbs_w_last_sweep_stats(self.dev.h, C.byref(st)))
        return {f: getattr(st, f) for f, _ in st._fields_}


class BwChain:
    """hydraw_chain_t: the runMpiGibbs_bW body on top of a loaded Device."""

    def __init__(self, dev, y, failure, mS=None, groups=None, seed=1222, shuffle=1, quad=9):
        self.dev, self.L = dev, dev.L
        if mS is None:
            mS = np.array([[0.0, 0.001, 0.01]])
        self.mS = np.ascontiguousarray(mS, dtype=np.float64)
        self.G, self.K = self.mS.shape
        self.groups = None if groups is None else np.ascontiguousarray(groups, dtype=np.int32)
        y = np.ascontiguousarray(y, dtype=np.float64)
        self.fail = np.ascontiguousarray(failure, dtype=np.int32)
        assert y.shape[0] == dev.n_global and self.fail.shape[0] == dev.n_global
        d = WModelDesc(seed, shuffle, self.G, self.K, _ip(self.groups) if self.groups is not None else None, _dp(self.mS), quad)
        self.h = C.c_void_p()
        self.C = 0
        check(self.L.hydraw_chain_create(dev.h, C.byref(d), _dp(y), _ip(self.fail), C.byref(self.h)))

    def __del__(self):
        try:
            if self.h:
                self.L.hydraw_chain_destroy(self.h)
        except Exception:
            pass

    def set_covariates(self, X):
        X = np.ascontiguousarray(X, dtype=np.float64)
        self.C = X.shape[1]
        check(self.L.hydraw_chain_set_covariates(self.h, _dp(X), self.C))

    def reseed_ars(self, seed):
        check(self.L.hydraw_chain_reseed_ars(self.h, seed))

    def rng_words(self):
        st = self.state()
        rng = RngState()
        rng.x[:] = list(st["rng_x"])
        rng.idx = st["rng_idx"]
        out = np.zeros(624, dtype=np.uint32)
        check(self.L.hydra_rng_to_boost_words(C.byref(rng), out.ctypes.data_as(C.POINTER(C.c_uint32))))
        return out

    def restore(self, iteration, mu, alpha, sigmaG, pi, beta, components, eps, order, rng_words, ars_seed, gamma=None, xI=None):
        f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
        keep = [f64(sigmaG), f64(pi), f64(beta), i32(components), f64(eps), i32(order)]
        st = WRestartState()
        st.iteration, st.mu, st.alpha, st.ars_seed = iteration, mu, alpha, ars_seed
        st.sigmaG, st.pi, st.beta = _dp(keep[0]), _dp(keep[1]), _dp(keep[2])
        st.components, st.eps, st.order = _ip(keep[3]), _dp(keep[4]), _ip(keep[5])
        if gamma is not None:
            keep += [f64(gamma), i32(xI)]
            st.gamma, st.xI = _dp(keep[6]), _ip(keep[7])
        w = np.ascontiguousarray(rng_words, dtype=np.uint32)
        check(self.L.hydra_rng_from_boost_words(w.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(st.rng)))
        check(self.L.hydraw_chain_restore(self.h, C.byref(st)))

    def iterate(self):
        check(self.L.hydraw_chain_iterate(self.h))

    def state(self):
        G, K = self.G, self.K
        mu, alpha = C.c_double(), C.c_double()
        sG, pi = np.zeros(G), np.zeros((G, K))
        m0, cass = np.zeros(G, dtype=np.int32), np.zeros((G, K), dtype=np.int32)
        rng, gr = RngState(), GrandState()
        check(self.L.hydraw_chain_state(self.h, C.byref(mu), C.byref(alpha), _dp(sG), _dp(pi), _ip(m0), _ip(cass), C.byref(rng), C.byref(gr)))
        return {"mu": mu.value, "alpha": alpha.value, "sigmaG": sG, "pi": pi, "m0": m0, "cass": cass,
                "rng_x": np.array(rng.x, dtype=np.uint32), "rng_idx": int(rng.idx), "grand": gr}

    def gamma(self):
        g, xi = np.zeros(self.C), np.zeros(self.C, dtype=np.int32)
        check(self.L.hydraw_chain_gamma(self.h, _dp(g), _ip(xi)))
        return g, xi

    def beta(self):
        b, c = np.zeros(self.dev.M), np.zeros(self.dev.M, dtype=np.int32)
        check(self.L.hgibbs_w_get_beta(self.dev.h, _dp(b), _ip(c)))
        return b, c

    def order(self):
        return np.ctypeslib.as_array(self.L.hydraw_chain_order(self.h), shape=(self.dev.M,)).copy()

    def last_nnz(self):
        return int(self.L.hydraw_chain_last_nnz(self.h))

    def sweep_stats(self):
        st = WSweepStats()
        check(self.L.hgibbs_w_last_sweep_stats(self.dev.h, C.byref(st)))
        return {f: getattr(st, f) for f, _ in st._fields_}

    def csv_line(self, it):
        buf = C.create_string_buffer(50000)
        n = self.L.hydraw_chain_csv_line(self.h, it, buf, 50000)
        return buf.raw[:n].decode()
