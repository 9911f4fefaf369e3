/* oracle/orc_rng.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C) of the random number machinery the reference's
 * Gibbs sampler draws from.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may include/link this.
 *
 * The reference wraps Boost.Random (un-vendored third party; pinned to 1.67.0
 * by the prebuilt ELF's build paths) in
 *   src/distributions_boost.cpp:34-36   reset_rng        -> mt19937(seed)
 *   src/distributions_boost.cpp:57-61   rgamma           -> gamma_distribution
 *   src/distributions_boost.cpp:63-67   unif_rng         -> uniform_real_distribution(0,1)
 *   src/distributions_boost.cpp:79-87   dirichlet_rng
 *   src/distributions_boost.cpp:89-107  inv_gamma_rng / inv_scaled_chisq_rng
 *   src/distributions_boost.cpp:109-113 norm_rng         -> normal_distribution (Ziggurat)
 *   src/distributions_boost.cpp:132-136 beta_rng         -> beta_distribution
 * Boost itself is absent from /root/reference, so what follows restates the
 * published algorithms of Boost.Random 1.67:
 *   - mersenne_twister_engine (Matsumoto & Nishimura MT19937),
 *   - detail::generate_int_float_pair<double,8> on a 32-bit engine:
 *       u1 = eng(); bucket = u1 & 0xff; r = (u1 >> 8) * 2^-24;
 *       u2 = eng(); r = (r + (u2 & 0x1fffffff)) * 2^-29
 *     (confirmed against the ELF's disassembly),
 *   - unit_normal_distribution: 128-layer Ziggurat with tangent/diagonal
 *     wedge bounds (the ELF's norm_rng has exactly this shape: two bound
 *     tests against table_x[i] >= 1, then one exp call),
 *   - unit_exponential_distribution: 256-layer Ziggurat, tail by shift,
 *   - gamma_distribution: alpha==1 exponential; alpha>1 Cauchy-envelope
 *     rejection (tan method); alpha<1 Ahrens-Dieter GS,
 *   - beta_distribution: X/(X+Y) of two unit gammas,
 *   - uniform_01 / uniform_real(0,1): one 32-bit output times 2^-32.
 * PARITY PINNING: the reference holds no golden vectors for this arithmetic
 * (SURVEY.md 8c "parity unpinned").  What IS pinned: the MT19937 known answer
 * of the C++ standard (10000th output of seed 5489 = 4123659995) and the four
 * Ziggurat tables, bit for bit, against the tables inside the reference's
 * prebuilt ELF (tests/test_oracle_rng.py).
 */
#ifndef ORC_RNG_H
#define ORC_RNG_H

#include <math.h>
#include <stdint.h>
#include "zig_tables.h"

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MT_N 624
#define ORC_MT_M 397

typedef struct {
    uint32_t x[ORC_MT_N];
    uint32_t idx; /* next output position; ORC_MT_N means "twist first" */
} orc_mt;

static inline void orc_mt_seed(orc_mt* g, uint32_t seed)
{
    g->x[0] = seed;
    for (uint32_t i = 1; i < ORC_MT_N; ++i)
        g->x[i] = 1812433253u * (g->x[i - 1] ^ (g->x[i - 1] >> 30)) + i;
    g->idx = ORC_MT_N;
}

static inline void orc_mt_twist(orc_mt* g)
{
    uint32_t* x = g->x;
    for (int i = 0; i < ORC_MT_N; ++i) {
        uint32_t y = (x[i] & 0x80000000u) | (x[(i + 1) % ORC_MT_N] & 0x7fffffffu);
        x[i] = x[(i + ORC_MT_M) % ORC_MT_N] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    g->idx = 0;
}

static inline uint32_t orc_mt_next(orc_mt* g)
{
    if (g->idx >= ORC_MT_N) orc_mt_twist(g);
    uint32_t z = g->x[g->idx++];
    z ^= (z >> 11);
    z ^= (z << 7) & 0x9d2c5680u;
    z ^= (z << 15) & 0xefc60000u;
    z ^= (z >> 18);
    return z;
}

/* uniform_01<double> / uniform_real_distribution<double>(0,1) on mt19937:
 * numerator / 2^32, retried while the result is not < 1 (never for 32 bits). */
/* Stream form of boost::random::mersenne_twister_engine (Boost 1.67,
 * boost/random/mersenne_twister.hpp: print(), rewind(), rewind_find(); operator>> reads n
 * words and sets i = n).  hydra dumps/restores dist.rng this way
 * (src/distributions_boost.cpp:38-55).  The n printed words are the window ending right
 * before the next output: the i consumed words of the current block go last, the n - i words
 * before them are recovered by running the recurrence backwards. */
static inline uint32_t orc_mt_rewind_find(const orc_mt* g, const uint32_t* last, size_t size, size_t j)
{
    const size_t n = ORC_MT_N;
    const size_t index = (j + n - size + n - 1) % n;
    if (index < n - size) return g->x[index];
    return *(last - (n - 1 - index));
}

static inline void orc_mt_rewind(const orc_mt* g, uint32_t* last, size_t z)
{
    const uint32_t upper = 0x80000000u, lower = 0x7fffffffu, a = 0x9908b0dfu;
    uint32_t y0 = g->x[ORC_MT_M - 1] ^ g->x[ORC_MT_N - 1];
    y0 = (y0 & upper) ? (((y0 ^ a) << 1) | 1u) : (y0 << 1);
    for (size_t sz = 0; sz < z; ++sz) {
        uint32_t y1 = orc_mt_rewind_find(g, last, sz, ORC_MT_M - 1) ^ orc_mt_rewind_find(g, last, sz, ORC_MT_N - 1);
        y1 = (y1 & upper) ? (((y1 ^ a) << 1) | 1u) : (y1 << 1);
        *(last - sz) = (y0 & upper) | (y1 & lower);
        y0 = y1;
    }
}

static inline void orc_mt_print_words(const orc_mt* g, uint32_t* data /* n */)
{
    const size_t n = ORC_MT_N;
    size_t i = g->idx > n ? n : g->idx;
    for (size_t j = 0; j < i; ++j) data[j + n - i] = g->x[j];
    if (i != n) orc_mt_rewind(g, &data[n - i - 1], n - i);
}

static inline void orc_mt_load_words(orc_mt* g, const uint32_t* data /* n */)
{
    for (int j = 0; j < ORC_MT_N; ++j) g->x[j] = data[j];
    g->idx = ORC_MT_N;
}

static inline double orc_unif01(orc_mt* g)
{
    return (double)orc_mt_next(g) * (1.0 / 4294967296.0);
}

/* generate_int_float_pair<double, 8>: 8-bit bucket + 53-bit uniform in [0,1) */
static inline double orc_int_float_pair(orc_mt* g, int* bucket)
{
    uint32_t u1 = orc_mt_next(g);
    *bucket = (int)(u1 & 0xffu);
    double r = (double)(u1 >> 8) * (1.0 / 16777216.0);
    uint32_t u2 = orc_mt_next(g);
    r += (double)(u2 & 0x1fffffffu);
    r *= (1.0 / 536870912.0);
    return r;
}

/* unit_exponential_distribution<double> */
static inline double orc_unit_exponential(orc_mt* g)
{
    const double* tx = ORC_ZIG_EXP_X;
    const double* ty = ORC_ZIG_EXP_Y;
    double shift = 0.0;
    for (;;) {
        int i;
        double r = orc_int_float_pair(g, &i);
        double x = r * tx[i];
        if (x < tx[i + 1]) return shift + x;
        if (i == 0) {
            shift += tx[1];
        } else {
            double y01 = orc_unif01(g);
            double y = ty[i] + y01 * (ty[i + 1] - ty[i]);
            double y_above_ubound = (tx[i] - tx[i + 1]) * y01 - (tx[i] - x);
            double y_above_lbound = y - (ty[i + 1] + (tx[i + 1] - x) * ty[i + 1]);
            if (y_above_ubound < 0 && (y_above_lbound < 0 || y < exp(-x)))
                return x + shift;
        }
    }
}

/* exponential_distribution<double>(lambda) */
static inline double orc_exponential(orc_mt* g, double lambda)
{
    return orc_unit_exponential(g) / lambda;
}

/* unit_normal_distribution<double> */
static inline double orc_unit_normal(orc_mt* g)
{
    const double* tx = ORC_ZIG_NORMAL_X;
    const double* ty = ORC_ZIG_NORMAL_Y;
    for (;;) {
        int i;
        double r = orc_int_float_pair(g, &i);
        int sign = (i & 1) * 2 - 1;
        i >>= 1;
        double x = r * tx[i];
        if (x < tx[i + 1]) return x * sign;
        if (i == 0) {
            /* tail: rejection from exponential(tail_start) shifted by it */
            const double tail_start = tx[1];
            for (;;) {
                double xt = orc_exponential(g, tail_start);
                double yt = orc_exponential(g, 1.0);
                if (2 * yt > xt * xt) return (xt + tail_start) * sign;
            }
        }
        double y01 = orc_unif01(g);
        double y = ty[i] + y01 * (ty[i + 1] - ty[i]);
        double y_above_ubound, y_above_lbound;
        if (tx[i] >= 1) { /* convex region (incl. the inflection layer) */
            y_above_ubound = (tx[i] - tx[i + 1]) * y01 - (tx[i] - x);
            y_above_lbound = y - (ty[i] + (tx[i] - x) * ty[i] * tx[i]);
        } else { /* concave */
            y_above_lbound = (tx[i] - tx[i + 1]) * y01 - (tx[i] - x);
            y_above_ubound = y - (ty[i] + (tx[i] - x) * ty[i] * tx[i]);
        }
        if (y_above_ubound < 0 && (y_above_lbound < 0 || y < exp(-(x * x / 2))))
            return x * sign;
    }
}

/* Distributions_boost::norm_rng(mean, sigma2): normal_distribution(mean, sqrt(sigma2)) */
static inline double orc_norm_rng(orc_mt* g, double mean, double sigma2)
{
    double sigma = sqrt(sigma2);
    return orc_unit_normal(g) * sigma + mean;
}

/* std::shuffle(first, last, dist.rng) as the reference binary runs it (src/BayesRRm.cpp:1692, :2653;
 * src/BayesW.cpp:1368, :1462): libstdc++ 6.5's loop -- for i = 1..n-1 swap(v[i], v[d(0, i)]) -- with
 * uniform_int_distribution<unsigned long>'s classic down-scaling on a 32-bit engine.  PINNED structurally:
 * the ELF's std::shuffle<vector<int>::iterator, mt19937&> has this div / imul / div around its inlined
 * generator and the toolchain strings say gcc-6.5.0.  (src/BayesRRm.cpp:1688-1692 warns that the result
 * depends on the toolchain; the restatement removes that dependence.) */
static inline void orc_shuffle_u32(orc_mt* g, uint32_t* v, size_t n)
{
    const uint64_t urngrange = 0xffffffffull;
    for (size_t i = 1; i < n; ++i) {
        const uint64_t uerange = (uint64_t)i + 1;
        const uint64_t scaling = urngrange / uerange, past = uerange * scaling;
        uint64_t ret;
        do {
            ret = orc_mt_next(g);
        } while (ret >= past);
        ret /= scaling;
        {
            const uint32_t t = v[i];
            v[i] = v[ret];
            v[ret] = t;
        }
    }
}

/* Distributions_boost::unif_rng() */
static inline double orc_unif_rng(orc_mt* g) { return orc_unif01(g); }

/* Distributions_boost::rgamma(shape, scale): gamma_distribution<double>(shape, scale) */
static inline double orc_rgamma(orc_mt* g, double alpha, double beta)
{
    if (alpha == 1.0) {
        return orc_exponential(g, 1.0) * beta;
    } else if (alpha > 1.0) {
        const double pi = 3.14159265358979323846;
        for (;;) {
            double y = tan(pi * orc_unif01(g));
            double x = sqrt(2.0 * alpha - 1.0) * y + alpha - 1.0;
            if (x <= 0.0) continue;
            if (orc_unif01(g) >
                (1.0 + y * y) * exp((alpha - 1.0) * log(x / (alpha - 1.0)) - sqrt(2.0 * alpha - 1.0) * y))
                continue;
            return x * beta;
        }
    } else {
        const double p = exp(1.0) / (alpha + exp(1.0));
        for (;;) {
            double u = orc_unif01(g);
            double y = orc_exponential(g, 1.0);
            double x, q;
            if (u < p) {
                x = exp(-y / alpha);
                q = p * exp(-x);
            } else {
                x = 1.0 + y;
                q = p + (1.0 - p) * pow(x, alpha - 1.0);
            }
            if (u >= q) continue;
            return x * beta;
        }
    }
}

/* distributions_boost.cpp:89-91 */
static inline double orc_inv_gamma_rng(orc_mt* g, double shape, double scale)
{
    return 1.0 / orc_rgamma(g, shape, 1.0 / scale);
}

/* distributions_boost.cpp:105-107 */
static inline double orc_inv_scaled_chisq_rng(orc_mt* g, double dof, double scale)
{
    return orc_inv_gamma_rng(g, 0.5 * dof, 0.5 * dof * scale);
}

/* distributions_boost.cpp:132-136: beta_distribution = a/(a+b), two unit gammas */
static inline double orc_beta_rng(orc_mt* g, double a, double b)
{
    double x = orc_rgamma(g, a, 1.0);
    double y = orc_rgamma(g, b, 1.0);
    return x / (x + y);
}

/* distributions_boost.cpp:79-87: K gamma(alpha_k,1) draws normalised by their
 * sum (Eigen's VectorXd::sum() of K<=8 elements; restated as index order). */
static inline void orc_dirichlet_rng(orc_mt* g, const double* alpha, int len, double* out)
{
    double s = 0.0;
    for (int i = 0; i < len; ++i) out[i] = orc_rgamma(g, alpha[i], 1.0);
    for (int i = 0; i < len; ++i) s += out[i];
    for (int i = 0; i < len; ++i) out[i] /= s;
}

#ifdef __cplusplus
}
#endif
#endif /* ORC_RNG_H */
