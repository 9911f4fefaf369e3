"""ctypes loader for the CPU oracle (oracle/liboracle.so) -- test infrastructure only."""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")


class OrcMt(C.Structure):
    _fields_ = [("x", C.c_uint32 * 624), ("idx", C.c_uint32)]


def _build():
    lib = os.path.join(ORACLE_DIR, "liboracle.so")
    src = [os.path.join(ORACLE_DIR, f) for f in ("orc_gibbs.cpp", "orc_bayesw.cpp", "orc_rng.h", "orc_ars.h", "zig_tables.h", "gh_tables.h")]
    if (not os.path.exists(lib)) or any(os.path.getmtime(s) > os.path.getmtime(lib) for s in src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "--quiet", os.path.join(ORACLE_DIR, "liboracle.so")])
    return lib


def load(name="liboracle.so"):
    path = _build() if name == "liboracle.so" else os.path.join(ORACLE_DIR, name)
    L = C.CDLL(path)
    dp, ip, u8p = C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_uint8)
    u64p = C.POINTER(C.c_uint64)
    mtp = C.POINTER(OrcMt)
    L.orc_set_threads.argtypes = [C.c_int]
    L.orc_set_dot_form.argtypes = [C.c_int]
    L.orc_decode_byte.argtypes = [C.c_uint8, dp, dp]
    L.orc_bed_counts.argtypes = [u8p, C.c_uint32, u64p, u64p, u64p, u64p]
    L.orc_marker_stats.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, dp, dp]
    L.orc_dot.argtypes = [u8p, dp, C.c_uint32, C.c_double, C.c_double]
    L.orc_dot.restype = C.c_double
    L.orc_dot_dense.argtypes = [u8p, dp, C.c_uint32, C.c_double, C.c_double, dp, dp]
    L.orc_dot_dense.restype = C.c_double
    L.orc_update.argtypes = [u8p, dp, C.c_uint32, C.c_double, C.c_double, C.c_double]
    L.orc_center_and_scale.argtypes = [dp, C.c_uint32]
    L.orc_bed_compact.argtypes = [u8p, C.c_uint32, u8p, u8p, C.POINTER(C.c_uint32)]
    L.orc_sweep.argtypes = [u8p, C.c_uint64, C.c_uint32, C.c_uint32, dp, dp, C.c_int, C.c_int, ip, dp, dp,
                            ip, C.c_double, dp, dp, u8p, dp, dp, ip, dp, ip, mtp]
    L.orc_sweep.restype = C.c_long
    L.orc_marker_draw.argtypes = [C.c_double, C.c_double, C.c_uint32, C.c_int, dp, dp, dp, C.c_double, C.c_double, mtp, dp, ip, dp]
    L.orc_chain_create.argtypes = [u8p, C.c_uint64, C.c_uint32, C.c_uint32, dp, C.c_int, C.c_int, ip, dp,
                                   C.c_uint32, C.c_int]
    L.orc_chain_create.restype = C.c_void_p
    for f in ("destroy", "iter_begin", "iter_end", "iterate"):
        getattr(L, "orc_chain_" + f).argtypes = [C.c_void_p]
        getattr(L, "orc_chain_" + f).restype = None
    L.orc_chain_set_covariates.argtypes = [C.c_void_p, dp, C.c_int]
    L.orc_chain_gamma.argtypes = [C.c_void_p]
    L.orc_chain_gamma.restype = dp
    L.orc_chain_xI.argtypes = [C.c_void_p]
    L.orc_chain_xI.restype = C.POINTER(C.c_uint32)
    u32p = C.POINTER(C.c_uint32)
    L.orc_chain_restore.argtypes = [C.c_void_p, C.c_uint32, C.c_double, C.c_double, dp, dp, dp, ip, dp, ip, dp, ip, u32p]
    L.orc_chain_restore.restype = None
    L.orc_rng_print_words.argtypes = [mtp, u32p]
    L.orc_rng_load_words.argtypes = [mtp, u32p]
    L.orc_chain_sweep.argtypes = [C.c_void_p]
    L.orc_chain_sweep.restype = C.c_long
    for f in ("beta", "acum", "eps", "y", "sigmaG", "estPi", "mave", "mstd", "cVa", "cVaI"):
        getattr(L, "orc_chain_" + f).argtypes = [C.c_void_p]
        getattr(L, "orc_chain_" + f).restype = dp
    for f in ("components", "order", "cass", "m0"):
        getattr(L, "orc_chain_" + f).argtypes = [C.c_void_p]
        getattr(L, "orc_chain_" + f).restype = ip
    L.orc_chain_adaV.argtypes = [C.c_void_p]
    L.orc_chain_adaV.restype = u8p
    L.orc_chain_sigmaE.argtypes = [C.c_void_p]
    L.orc_chain_sigmaE.restype = C.c_double
    L.orc_chain_mu.argtypes = [C.c_void_p]
    L.orc_chain_mu.restype = C.c_double
    L.orc_chain_last_nnz.argtypes = [C.c_void_p]
    L.orc_chain_last_nnz.restype = C.c_long
    L.orc_chain_rng.argtypes = [C.c_void_p]
    L.orc_chain_rng.restype = mtp
    L.orc_chain_csv_line.argtypes = [C.c_void_p, C.c_uint32, C.c_char_p, C.c_size_t]
    L.orc_rng_seed.argtypes = [mtp, C.c_uint32]
    L.orc_rng_u32.argtypes = [mtp]
    L.orc_rng_u32.restype = C.c_uint32
    for f, extra in (("unif", []), ("norm", [C.c_double] * 2), ("exp", [C.c_double]),
                     ("gamma", [C.c_double] * 2), ("beta", [C.c_double] * 2),
                     ("inv_scaled_chisq", [C.c_double] * 2)):
        getattr(L, "orc_rng_" + f).argtypes = [mtp] + extra
        getattr(L, "orc_rng_" + f).restype = C.c_double
    L.orc_rng_dirichlet.argtypes = [mtp, dp, C.c_int, dp]
    L.orc_rng_shuffle.argtypes = [mtp, ip, C.c_int]
    L.orc_zig_table.argtypes = [C.c_int, ip]
    L.orc_zig_table.restype = dp
    return L


def p(a, ty):
    return a.ctypes.data_as(C.POINTER(ty))


def dptr(a):
    assert a.dtype == np.float64 and a.flags.c_contiguous
    return p(a, C.c_double)


def iptr(a):
    assert a.dtype == np.int32 and a.flags.c_contiguous
    return p(a, C.c_int)


def u8ptr(a):
    assert a.dtype == np.uint8 and a.flags.c_contiguous
    return p(a, C.c_uint8)


def view(ptr, n, dtype):
    return np.ctypeslib.as_array(ptr, shape=(n,)).view(dtype) if n else np.zeros(0, dtype)


class Chain:
    """Thin Python face of the oracle's whole-chain driver."""

    def __init__(self, L, bed, N, y, groups=None, mS=None, seed=1222, shuffle=1):
        self.L = L
        self.bed = np.ascontiguousarray(bed, dtype=np.uint8)
        self.M, self.stride = self.bed.shape
        self.N = N
        if mS is None:
            mS = np.array([[0.0, 0.0001, 0.001, 0.01]])
        self.mS = np.ascontiguousarray(mS, dtype=np.float64)
        self.G, self.K = self.mS.shape
        if groups is None:
            groups = np.zeros(self.M, dtype=np.int32)
        self.groups = np.ascontiguousarray(groups, dtype=np.int32)
        self.y_raw = np.ascontiguousarray(y, dtype=np.float64)
        self.h = L.orc_chain_create(u8ptr(self.bed), self.stride, N, self.M, dptr(self.y_raw), self.G, self.K,
                                    iptr(self.groups), dptr(self.mS), seed, shuffle)

    def __del__(self):
        try:
            self.L.orc_chain_destroy(self.h)
        except Exception:
            pass

    def arr(self, name):
        L, h = self.L, self.h
        n = {"beta": self.M, "acum": self.M, "eps": self.N, "y": self.N, "sigmaG": self.G,
             "estPi": self.G * self.K, "mave": self.M, "mstd": self.M, "cVa": self.G * self.K,
             "cVaI": self.G * self.K, "components": self.M, "order": self.M, "cass": self.G * self.K,
             "m0": self.G, "adaV": self.M}[name]
        ptr = getattr(L, "orc_chain_" + name)(h)
        return np.ctypeslib.as_array(ptr, shape=(n,))

    @property
    def sigmaE(self):
        return self.L.orc_chain_sigmaE(self.h)

    @property
    def mu(self):
        return self.L.orc_chain_mu(self.h)

    def rng_state(self):
        r = self.L.orc_chain_rng(self.h).contents
        return np.array(r.x, dtype=np.uint32), int(r.idx)

    def set_covariates(self, X):
        self.X = np.ascontiguousarray(X, dtype=np.float64)
        self.C = self.X.shape[1]
        self.L.orc_chain_set_covariates(self.h, dptr(self.X), self.C)

    def gamma(self):
        return np.ctypeslib.as_array(self.L.orc_chain_gamma(self.h), shape=(self.C,)).copy()

    def xI(self):
        return np.ctypeslib.as_array(self.L.orc_chain_xI(self.h), shape=(self.C,)).astype(np.int32)

    def rng_words(self):
        """dist.rng as `file << rng` prints it (624 words)."""
        out = np.zeros(624, dtype=np.uint32)
        self.L.orc_rng_print_words(self.L.orc_chain_rng(self.h), p(out, C.c_uint32))
        return out

    def restore(self, iteration, sigmaE, mu, sigmaG, estPi, beta, components, eps, order, rng_words, gamma=None, xI=None):
        f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
        keep = [f64(sigmaG), f64(estPi), f64(beta), i32(components), f64(eps), i32(order),
                f64(gamma if gamma is not None else np.zeros(1)), i32(xI if xI is not None else np.zeros(1)),
                np.ascontiguousarray(rng_words, dtype=np.uint32)]
        self.L.orc_chain_restore(self.h, iteration, sigmaE, mu, dptr(keep[0]), dptr(keep[1]), dptr(keep[2]), iptr(keep[3]),
                                 dptr(keep[4]), iptr(keep[5]), dptr(keep[6]), iptr(keep[7]), p(keep[8], C.c_uint32))

    def iterate(self):
        self.L.orc_chain_iterate(self.h)

    def csv_line(self, it):
        buf = C.create_string_buffer(50000)
        n = self.L.orc_chain_csv_line(self.h, it, buf, 50000)
        return buf.raw[:n].decode()


def _bind_bw(L):
    if getattr(L, "_bw_bound", False):
        return
    dp, ip, u8p = C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_uint8)
    vp = C.c_void_p
    L.orc_bw_create.argtypes = [u8p, C.c_uint64, C.c_uint32, C.c_uint32, dp, dp, C.c_int, C.c_int, ip, dp, C.c_uint32, C.c_int, C.c_int]
    L.orc_bw_create.restype = vp
    L.orc_bw_destroy.argtypes = [vp]
    L.orc_bw_set_arms.argtypes = [vp, vp]
    L.orc_bw_set_covariates.argtypes = [vp, dp, C.c_int]
    L.orc_bw_reseed_ars.argtypes = [vp, C.c_uint32]
    L.orc_bw_restore.argtypes = [vp, C.c_double, C.c_double, dp, dp, dp, ip, dp, ip, dp, ip, C.POINTER(C.c_uint32), C.c_uint32]
    L.orc_bw_restore.restype = None
    L.orc_rng_print_words.argtypes = [C.POINTER(OrcMt), C.POINTER(C.c_uint32)]
    L.orc_bw_iterate.argtypes = [vp]
    L.orc_bw_quad_supported.argtypes = [C.c_int]
    for f in ("beta", "eps", "vi", "sigmaG", "pi", "mave", "msd", "sum_failure", "gamma"):
        getattr(L, "orc_bw_" + f).argtypes = [vp]
        getattr(L, "orc_bw_" + f).restype = dp
    for f in ("components", "order", "cass", "m0"):
        getattr(L, "orc_bw_" + f).argtypes = [vp]
        getattr(L, "orc_bw_" + f).restype = ip
    L.orc_bw_xI.argtypes = [vp]
    L.orc_bw_xI.restype = C.POINTER(C.c_uint32)
    for f in ("mu", "alpha"):
        getattr(L, "orc_bw_" + f).argtypes = [vp]
        getattr(L, "orc_bw_" + f).restype = C.c_double
    for f in ("last_nnz", "ars_evals"):
        getattr(L, "orc_bw_" + f).argtypes = [vp]
        getattr(L, "orc_bw_" + f).restype = C.c_long
    L.orc_bw_rng.argtypes = [vp]
    L.orc_bw_rng.restype = C.POINTER(OrcMt)
    L.orc_bw_csv_line.argtypes = [vp, C.c_uint32, C.c_char_p, C.c_size_t]
    L.orc_bw_beta_dens.argtypes = [C.c_double] * 10
    L.orc_bw_beta_dens.restype = C.c_double
    L.orc_bw_marginals.argtypes = [C.c_int, C.c_int, dp, dp] + [C.c_double] * 9 + [dp]
    L._bw_bound = True


REF_ARMS_SYMBOL = "_Z4armsPdiS_S_PFddPvES0_S_iiS_S_iS_S_iPi"  # arms(...) of src/BayesW_arms.cpp, compiled as C++


def ref_arms_lib():
    """oracle/_ref/libarms.so (the reference's own ARS sampler) or None."""
    path = os.path.join(ORACLE_DIR, "_ref", "libarms.so")
    return C.CDLL(path) if os.path.exists(path) else None


class BwChain:
    """Python face of the oracle's BayesW driver (libc rand() is process-global: run one chain at a time)."""

    def __init__(self, L, bed, N, y, fail, groups=None, mS=None, seed=1222, shuffle=1, quad=9):
        _bind_bw(L)
        self.L = L
        self.bed = np.ascontiguousarray(bed, dtype=np.uint8)
        self.M, self.stride = self.bed.shape
        self.N = N
        if mS is None:
            mS = np.array([[0.0, 0.001, 0.01]])
        self.mS = np.ascontiguousarray(mS, dtype=np.float64)
        self.G, self.K = self.mS.shape
        self.groups = np.ascontiguousarray(groups if groups is not None else np.zeros(self.M), dtype=np.int32)
        self.y = np.ascontiguousarray(y, dtype=np.float64)
        self.fail = np.ascontiguousarray(fail, dtype=np.float64)
        self.C = 0
        self.h = L.orc_bw_create(u8ptr(self.bed), self.stride, N, self.M, dptr(self.y), dptr(self.fail), self.G, self.K,
                                 iptr(self.groups), dptr(self.mS), seed, shuffle, quad)

    def __del__(self):
        try:
            self.L.orc_bw_destroy(self.h)
        except Exception:
            pass

    def use_reference_arms(self, lib):
        fn = getattr(lib, REF_ARMS_SYMBOL)
        self.L.orc_bw_set_arms(self.h, C.cast(fn, C.c_void_p))

    def set_covariates(self, X):
        self.X = np.ascontiguousarray(X, dtype=np.float64)
        self.C = self.X.shape[1]
        self.L.orc_bw_set_covariates(self.h, dptr(self.X), self.C)

    def reseed_ars(self, seed):
        self.L.orc_bw_reseed_ars(self.h, seed)

    def rng_words(self):
        out = np.zeros(624, dtype=np.uint32)
        self.L.orc_rng_print_words(self.L.orc_bw_rng(self.h), p(out, C.c_uint32))
        return out

    def xI(self):
        return np.ctypeslib.as_array(self.L.orc_bw_xI(self.h), shape=(self.C,)).astype(np.int32)

    def restore(self, mu, alpha, sigmaG, pi, beta, components, eps, order, rng_words, ars_seed, gamma=None, xI=None):
        f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
        k = [f64(sigmaG), f64(pi), f64(beta), i32(components), f64(eps), i32(order),
             f64(gamma if gamma is not None else np.zeros(1)), i32(xI if xI is not None else np.zeros(1)),
             np.ascontiguousarray(rng_words, dtype=np.uint32)]
        self.L.orc_bw_restore(self.h, mu, alpha, dptr(k[0]), dptr(k[1]), dptr(k[2]), iptr(k[3]), dptr(k[4]), iptr(k[5]), dptr(k[6]), iptr(k[7]),
                              p(k[8], C.c_uint32), ars_seed)

    def iterate(self):
        err = self.L.orc_bw_iterate(self.h)
        if err:
            raise RuntimeError("ARS error code %d" % err)

    def arr(self, name):
        n = {"beta": self.M, "eps": self.N, "vi": self.N, "sigmaG": self.G, "pi": self.G * self.K, "mave": self.M, "msd": self.M,
             "sum_failure": self.M, "gamma": self.C, "components": self.M, "order": self.M, "cass": self.G * self.K, "m0": self.G}[name]
        return np.ctypeslib.as_array(getattr(self.L, "orc_bw_" + name)(self.h), shape=(n,))

    @property
    def mu(self):
        return self.L.orc_bw_mu(self.h)

    @property
    def alpha(self):
        return self.L.orc_bw_alpha(self.h)

    def last_nnz(self):
        return self.L.orc_bw_last_nnz(self.h)

    def ars_evals(self):
        return self.L.orc_bw_ars_evals(self.h)

    def csv_line(self, it):
        buf = C.create_string_buffer(50000)
        n = self.L.orc_bw_csv_line(self.h, it, buf, 50000)
        return buf.raw[:n].decode()
