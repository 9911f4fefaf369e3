"""Seeded synthetic genotype / phenotype generator and PLINK .bed packing.

Shapes and distributions follow BASELINE.md section 4: g_ij ~ Binomial(2, p_j),
p_j ~ U(0.01, 0.5), optional missing calls, 1 % causal markers with
beta ~ N(0, h2/m_causal), y = X_std beta + e.

2-bit packing is the inverse of the reference's decode (src/data.cpp:1189-1200,
src/dotp_lut.h): code 00 -> 2, 10 -> 1, 11 -> 0, 01 -> missing; individual i sits
in byte i//4, bits 2*(i%4).  SNP-major, column stride ceil(N/4), file offset 3
after the magic bytes 6c 1b 01 (src/data.cpp:685,700).
"""
import os
import numpy as np

BED_MAGIC = bytes([0x6C, 0x1B, 0x01])
# genotype value (0,1,2, 3=missing) -> 2-bit PLINK code
_CODE_OF_GENO = np.array([0b11, 0b10, 0b00, 0b01], dtype=np.uint8)


def pack_bed_columns(geno):
    """geno: (M, N) uint8 with values 0,1,2 or 3 (missing) -> (M, ceil(N/4)) uint8.

    Tail slots of the last byte are filled with the missing code (they lie
    beyond N and are never decoded by the reference, which bounds by N)."""
    geno = np.ascontiguousarray(geno, dtype=np.uint8)
    M, N = geno.shape
    nb = (N + 3) // 4
    codes = _CODE_OF_GENO[geno]
    if nb * 4 != N:
        pad = np.full((M, nb * 4 - N), 0b01, dtype=np.uint8)
        codes = np.concatenate([codes, pad], axis=1)
    codes = codes.reshape(M, nb, 4)
    out = codes[:, :, 0] | (codes[:, :, 1] << 2) | (codes[:, :, 2] << 4) | (codes[:, :, 3] << 6)
    return np.ascontiguousarray(out, dtype=np.uint8)


def unpack_bed_columns(bed, N):
    """(M, ceil(N/4)) uint8 -> (M, N) uint8 genotypes with 3 = missing."""
    bed = np.asarray(bed, dtype=np.uint8)
    M = bed.shape[0]
    codes = np.stack([(bed >> (2 * s)) & 3 for s in range(4)], axis=2).reshape(M, -1)[:, :N]
    geno_of_code = np.array([2, 3, 1, 0], dtype=np.uint8)  # 00->2, 01->miss, 10->1, 11->0
    return geno_of_code[codes]


def make_genotypes(M, N, seed=42, missing_rate=0.0, maf_lo=0.01, maf_hi=0.5):
    rng = np.random.default_rng(seed)
    p = rng.uniform(maf_lo, maf_hi, size=M)
    geno = rng.binomial(2, p[:, None], size=(M, N)).astype(np.uint8)
    if missing_rate > 0:
        miss = rng.random((M, N)) < missing_rate
        geno[miss] = 3
    # keep every marker polymorphic among its non-missing calls (the reference
    # divides by zero on monomorphic markers, src/BayesRRm.cpp:1507)
    for j in range(M):
        nm = geno[j] != 3
        vals = geno[j][nm]
        if vals.size == 0 or vals.min() == vals.max():
            idx = np.flatnonzero(nm)
            if idx.size < 2:
                geno[j, :2] = (0, 1)
            else:
                geno[j, idx[0]] = 0 if vals[0] != 0 else 1
    return geno


def standardize(geno):
    """Column-standardised X (N x M) as the reference defines it (mave/mstd from
    counts, missing -> 0 after centring), src/BayesRRm.cpp:1502-1507."""
    M, N = geno.shape
    X = np.zeros((N, M))
    for j in range(M):
        g = geno[j].astype(np.float64)
        nm = geno[j] != 3
        n1 = np.sum(geno[j] == 1)
        n2 = np.sum(geno[j] == 2)
        mave = (n1 + 2.0 * n2) / nm.sum()
        ss = np.sum((g[nm] - mave) ** 2)
        mstd = np.sqrt((N - 1) / ss)
        X[nm, j] = (g[nm] - mave) * mstd
    return X


def make_phenotype(geno, seed=43, h2=0.5, causal_frac=0.01):
    M, N = geno.shape
    rng = np.random.default_rng(seed)
    m_causal = max(1, int(round(M * causal_frac)))
    causal = rng.choice(M, size=m_causal, replace=False)
    beta = np.zeros(M)
    beta[causal] = rng.normal(0.0, np.sqrt(h2 / m_causal), size=m_causal)
    X = standardize(geno[causal])
    gval = X @ beta[causal]
    e = rng.normal(0.0, np.sqrt(1.0 - h2), size=N)
    return gval + e, beta


def make_survival(geno, seed=44, h2=0.5, causal_frac=0.01, mu=3.0, alpha=4.0, censor_rate=0.3):
    """Weibull log-time phenotype for BayesW: y = mu + X beta + w / alpha with w the log of a
    unit exponential shifted by Euler's constant (mean 0), so exp(alpha * (y - mu - X beta) - gamma_E)
    is unit exponential as the model assumes (src/BayesW.cpp:42,1457-1459).  Returns (y, failure, beta);
    failure = 1 for an observed event, 0 for a censored one (censored times are shortened)."""
    M, N = geno.shape
    rng = np.random.default_rng(seed)
    m_causal = max(1, int(round(M * causal_frac)))
    causal = rng.choice(M, size=m_causal, replace=False)
    var_e = np.pi ** 2 / (6.0 * alpha ** 2)
    var_g = var_e * h2 / (1.0 - h2)
    beta = np.zeros(M)
    beta[causal] = rng.normal(0.0, np.sqrt(var_g / m_causal), size=m_causal)
    gval = standardize(geno[causal]) @ beta[causal]
    w = np.log(rng.exponential(1.0, size=N)) + 0.577215664901532
    y = mu + gval + w / alpha
    fail = (rng.random(N) >= censor_rate).astype(np.int32)
    y = np.where(fail == 1, y, y - rng.exponential(0.2, size=N))
    return y, fail, beta


def write_plink(prefix, bed, N, y=None, na_rows=()):
    """Write <prefix>.bed/.fam/.bim (+ .phen) for a packed (M, ceil(N/4)) array."""
    M = bed.shape[0]
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    with open(prefix + ".bed", "wb") as f:
        f.write(BED_MAGIC)
        f.write(np.ascontiguousarray(bed, dtype=np.uint8).tobytes())
    with open(prefix + ".fam", "w") as f:
        for i in range(N):
            f.write("fam%d ind%d 0 0 0 -9\n" % (i, i))
    with open(prefix + ".bim", "w") as f:
        for j in range(M):
            f.write("1 snp%d 0 %d A C\n" % (j, j + 1))
    if y is not None:
        na = set(int(r) for r in na_rows)
        with open(prefix + ".phen", "w") as f:
            for i in range(N):
                f.write("fam%d ind%d %s\n" % (i, i, "NA" if i in na else repr(float(y[i]))))


def read_bed(path, N, M):
    raw = np.fromfile(path, dtype=np.uint8)
    if bytes(raw[:3]) != BED_MAGIC:
        raise ValueError("not a SNP-major PLINK .bed: %s" % path)
    nb = (N + 3) // 4
    return raw[3:3 + M * nb].reshape(M, nb)


# ---- host twin of the device-side synthetic generator (hgibbs_synth_bed) ----
_M64 = (1 << 64) - 1


def _mix64(z):
    z = (z + 0x9E3779B97F4A7C15) & _M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


def synth_bed_reference(n_global, M, seed=42, missing_rate=0.0, row_begin=0, row_end=None):
    """Pure-Python restatement of k_synth_bed (hydra_amd/csrc/hgibbs.hip) for
    small sizes: same counter hash, same integer thresholds."""
    if row_end is None:
        row_end = n_global
    n_local = row_end - row_begin
    mr = min(max(missing_rate, 0.0), 1.0)
    miss_thr = int(min(4294967295.0, mr * 4294967296.0))
    geno = np.zeros((M, n_local), dtype=np.uint8)
    for j in range(M):
        up = _mix64(seed ^ ((0xA5A5A5A5 + j * 0x100000001B3) & _M64)) >> 32
        p = 0.01 + 0.49 * (float(up) * (1.0 / 4294967296.0))
        q0 = (1.0 - p) * (1.0 - p)
        q1 = q0 + 2.0 * p * (1.0 - p)
        t0, t1 = int(q0 * 4294967296.0), int(q1 * 4294967296.0)
        for il in range(n_local):
            ig = row_begin + il
            h = _mix64((seed + j * 0x9E3779B1 + ig * 0xD1B54A32D192ED03) & _M64)
            ug, um = h >> 32, h & 0xFFFFFFFF
            if um < miss_thr:
                geno[j, il] = 3
            else:
                geno[j, il] = 0 if ug < t0 else (1 if ug < t1 else 2)
    return pack_bed_columns(geno)
