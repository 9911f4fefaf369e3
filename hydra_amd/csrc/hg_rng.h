// hg_rng.h -- MT19937 + the Boost.Random 1.67 distribution algorithms the
// reference draws through (src/distributions_boost.cpp:34-36,57-113,132-136),
// written once for host and device (gfx950).  Boost is an un-vendored
// dependency of the reference; these are its published algorithms:
// generate_int_float_pair<double,8>, 128-layer Ziggurat normal with
// tangent/diagonal wedge bounds, 256-layer Ziggurat exponential, gamma by
// Cauchy-envelope rejection (alpha>1) / Ahrens-Dieter GS (alpha<1), beta as a
// ratio of gammas.  Generator state is shared with the device sweep kernel
// through hgibbs_rng_state, so host and device consume ONE stream in the
// reference's order.
#pragma once

#include <math.h>
#include <stdint.h>
#if !defined(__HIP_DEVICE_COMPILE__)
#include <vector>
#endif

#include "hg_zig_tables.h"

#if defined(__HIP__)
#include <hip/hip_runtime.h>
#define HG_HD __host__ __device__ __forceinline__
#else
#define HG_HD inline
#endif

namespace hg {

constexpr int MT_N = 624;
constexpr int MT_M = 397;

HG_HD uint32_t mt_temper(uint32_t z)
{
    z ^= (z >> 11);
    z ^= (z << 7) & 0x9d2c5680u;
    z ^= (z << 15) & 0xefc60000u;
    z ^= (z >> 18);
    return z;
}

HG_HD uint32_t mt_mix(uint32_t cur, uint32_t nxt, uint32_t far)
{
    uint32_t y = (cur & 0x80000000u) | (nxt & 0x7fffffffu);
    return far ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

// Plain sequential generator over a 624-word state (host side, and one device
// lane working on a private/LDS copy).
struct Mt {
    uint32_t* x;
    uint32_t idx;
    HG_HD void seed(uint32_t s)
    {
        x[0] = s;
        for (uint32_t i = 1; i < MT_N; ++i) x[i] = 1812433253u * (x[i - 1] ^ (x[i - 1] >> 30)) + i;
        idx = MT_N;
    }
    HG_HD void twist()
    {
        for (int i = 0; i < MT_N; ++i) x[i] = mt_mix(x[i], x[(i + 1) % MT_N], x[(i + MT_M) % MT_N]);
        idx = 0;
    }
    HG_HD uint32_t next()
    {
        if (idx >= MT_N) twist();
        return mt_temper(x[idx++]);
    }
};

struct ZigTables {
    const double* nx;
    const double* ny;
    const double* ex;
    const double* ey;
};

template <class Gen>
HG_HD double unif01(Gen& g)
{
    return (double)g.next() * (1.0 / 4294967296.0);
}

template <class Gen>
HG_HD double int_float_pair(Gen& g, int& bucket)
{
    uint32_t u1 = g.next();
    bucket = (int)(u1 & 0xffu);
    double r = (double)(u1 >> 8) * (1.0 / 16777216.0);
    uint32_t u2 = g.next();
    r += (double)(u2 & 0x1fffffffu);
    r *= (1.0 / 536870912.0);
    return r;
}

template <class Gen, class Tab>
HG_HD double unit_exponential(Gen& g, const Tab& t)
{
    double shift = 0.0;
    for (;;) {
        int i;
        double r = int_float_pair(g, i);
        double x = r * t.ex[i];
        if (x < t.ex[i + 1]) return shift + x;
        if (i == 0) {
            shift += t.ex[1];
        } else {
            double y01 = unif01(g);
            double y = t.ey[i] + y01 * (t.ey[i + 1] - t.ey[i]);
            double y_above_ubound = (t.ex[i] - t.ex[i + 1]) * y01 - (t.ex[i] - x);
            double y_above_lbound = y - (t.ey[i + 1] + (t.ex[i + 1] - x) * t.ey[i + 1]);
            if (y_above_ubound < 0 && (y_above_lbound < 0 || y < exp(-x))) return x + shift;
        }
    }
}

template <class Gen, class Tab>
HG_HD double unit_normal(Gen& g, const Tab& t)
{
    for (;;) {
        int i;
        double r = int_float_pair(g, i);
        int sign = (i & 1) * 2 - 1;
        i >>= 1;
        double x = r * t.nx[i];
        if (x < t.nx[i + 1]) return x * sign;
        if (i == 0) {
            const double tail_start = t.nx[1];
            for (;;) {
                double xt = unit_exponential(g, t) / tail_start;
                double yt = unit_exponential(g, t) / 1.0;
                if (2 * yt > xt * xt) return (xt + tail_start) * sign;
            }
        }
        double y01 = unif01(g);
        double y = t.ny[i] + y01 * (t.ny[i + 1] - t.ny[i]);
        double y_above_ubound, y_above_lbound;
        if (t.nx[i] >= 1) {
            y_above_ubound = (t.nx[i] - t.nx[i + 1]) * y01 - (t.nx[i] - x);
            y_above_lbound = y - (t.ny[i] + (t.nx[i] - x) * t.ny[i] * t.nx[i]);
        } else {
            y_above_lbound = (t.nx[i] - t.nx[i + 1]) * y01 - (t.nx[i] - x);
            y_above_ubound = y - (t.ny[i] + (t.nx[i] - x) * t.ny[i] * t.nx[i]);
        }
        if (y_above_ubound < 0 && (y_above_lbound < 0 || y < exp(-(x * x / 2)))) return x * sign;
    }
}

// Distributions_boost::norm_rng(mean, sigma2), with sigma = sqrt(sigma2) given
template <class Gen, class Tab>
HG_HD double norm_rng_sd(Gen& g, const Tab& t, double mean, double sigma)
{
    return unit_normal(g, t) * sigma + mean;
}

// ---- host-only distributions (hyper-parameter draws, once per iteration) ----
inline ZigTables host_tables() { return ZigTables{HG_ZIG_NORMAL_X, HG_ZIG_NORMAL_Y, HG_ZIG_EXP_X, HG_ZIG_EXP_Y}; }

inline double norm_rng(Mt& g, double mean, double sigma2)
{
    ZigTables t = host_tables();
    return norm_rng_sd(g, t, mean, sqrt(sigma2));
}

inline double rgamma(Mt& g, double alpha, double beta)
{
    ZigTables t = host_tables();
    if (alpha == 1.0) {
        return unit_exponential(g, t) / 1.0 * beta;
    } else if (alpha > 1.0) {
        const double pi = 3.14159265358979323846;
        for (;;) {
            double y = tan(pi * unif01(g));
            double x = sqrt(2.0 * alpha - 1.0) * y + alpha - 1.0;
            if (x <= 0.0) continue;
            if (unif01(g) > (1.0 + y * y) * exp((alpha - 1.0) * log(x / (alpha - 1.0)) - sqrt(2.0 * alpha - 1.0) * y))
                continue;
            return x * beta;
        }
    } else {
        const double p = exp(1.0) / (alpha + exp(1.0));
        for (;;) {
            double u = unif01(g);
            double y = unit_exponential(g, t) / 1.0;
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

inline double inv_gamma_rng(Mt& g, double shape, double scale) { return 1.0 / rgamma(g, shape, 1.0 / scale); }
inline double inv_scaled_chisq_rng(Mt& g, double dof, double scale) { return inv_gamma_rng(g, 0.5 * dof, 0.5 * dof * scale); }
inline double beta_rng(Mt& g, double a, double b)
{
    double x = rgamma(g, a, 1.0);
    double y = rgamma(g, b, 1.0);
    return x / (x + y);
}
inline void dirichlet_rng(Mt& g, const double* alpha, int len, double* out)
{
    double s = 0.0;
    for (int i = 0; i < len; ++i) out[i] = rgamma(g, alpha[i], 1.0);
    for (int i = 0; i < len; ++i) s += out[i];
    for (int i = 0; i < len; ++i) out[i] /= s;
}

// Boost.Random's stream form of mersenne_twister_engine (what `file << rng` writes at
// src/distributions_boost.cpp:38-44): the n words of the window that ENDS just before the
// next output, so that a reader can load them and set i = n (operator>> does exactly that,
// :46-55).  The last idx words are the consumed part of the current block; the first
// n - idx words belong to the previous block and are recovered by running the recurrence
// backwards: each step yields the top bit of word k and the low 31 bits of word k+1.
inline void mt_to_boost_words(const uint32_t* x, uint32_t idx, uint32_t* out /* 624 */)
{
    const uint32_t UP = 0x80000000u, LO = 0x7fffffffu, A = 0x9908b0dfu;
    if (idx > (uint32_t)MT_N) idx = MT_N;
    for (uint32_t j = 0; j < idx; ++j) out[j + MT_N - idx] = x[j];
    if (idx == (uint32_t)MT_N) return;
    // word w(k), k in [idx, n): previous block's entry k; known: current block x[0..n)
    // x[k] = far(k) ^ G(w(k), w(k+1)),  far(k) = w(k+m) for k+m < n else x[k+m-n];  w(n) := x[0]
    auto unmix = [&](uint32_t y) { // y = G(a,b) -> (top bit of a) | (low 31 bits of b)
        if (y & UP) return ((y ^ A) << 1) | 1u;
        return y << 1;
    };
    std::vector<uint32_t> w(MT_N + 1, 0u);
    w[MT_N] = x[0];
    // step k gives top bit of w(k) and low bits of w(k+1); go from k = n-1 down to idx-1
    uint32_t have_low_of = MT_N; // w[MT_N] fully known
    (void)have_low_of;
    for (int k = MT_N - 1; k >= (int)idx - 1 && k >= 0; --k) {
        const uint32_t far = (k + MT_M < MT_N) ? w[k + MT_M] : x[k + MT_M - MT_N];
        const uint32_t y = unmix(x[k] ^ far);
        // low 31 bits of w(k+1) (consistency only for k+1 == n) and top bit of w(k)
        if (k + 1 < MT_N) w[k + 1] = (w[k + 1] & UP) | (y & LO);
        w[k] = (w[k] & LO) | (y & UP);
    }
    // the top bit of w(idx) came from step idx, its low bits from step idx-1 (when idx > 0);
    // for idx == 0 the low bits of w(0) are never used by the forward recurrence: leave them 0
    for (uint32_t k = idx; k < (uint32_t)MT_N; ++k) out[k - idx] = w[k];
}

// std::shuffle(first, last, dist.rng) (src/BayesRRm.cpp:1692, :2653; src/BayesW.cpp:1368, :1462) as the
// reference binary runs it: libstdc++ 6.5 (the ELF's .comment and rpath name gcc-6.5.0), i.e. for
// i = 1 .. n-1 one uniform_int_distribution<unsigned long>(0, i) draw and a swap, with the distribution's
// classic down-scaling on a 32-bit engine: scaling = (2^32 - 1) / (i + 1), reject draws >= (i + 1) * scaling,
// index = draw / scaling.  (The ELF's std::shuffle instantiation shows exactly this div / imul / div around
// its inlined generator.)  Later libstdc++ releases draw two indices per engine call and use Lemire's
// method, so std::shuffle itself would tie the chain to the host's toolchain.
template <class T>
inline void shuffle_libstdcxx6(T* first, size_t n, Mt& g)
{
    const uint64_t urngrange = 0xffffffffull;
    for (size_t i = 1; i < n; ++i) {
        const uint64_t uerange = (uint64_t)i + 1; // n <= 2^32 - 1 markers: always the down-scaling branch
        const uint64_t scaling = urngrange / uerange, past = uerange * scaling;
        uint64_t ret;
        do ret = g.next();
        while (ret >= past);
        ret /= scaling;
        const T t = first[i];
        first[i] = first[ret];
        first[ret] = t;
    }
}

} // namespace hg
