// hg_bayesw_math.h -- the scalar half of BayesW's per-marker step, for host and device:
// adaptive Gauss-Hermite marginal likelihoods (src/BayesW.cpp:161-726), the categorical
// walk over mixture components (:1536-1598) and the log density of one effect (:145-156).
// Everything here works on the three masked sums of vi = exp(alpha*eps - EuMasc) that the
// streaming kernel produces per marker.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define HG_BW_HD __host__ __device__
#else
#define HG_BW_HD
#endif

namespace hg {
namespace bw {

// src/BayesW.cpp:38-42
constexpr double PI_BW = 3.14159265359;
constexpr double PI_SQUARED = 9.86960440109;
constexpr double SQRT_PI = 1.77245385090552;
constexpr double EULER = 0.577215664901532;
// src/BayesW.hpp:85-89
constexpr double ALPHA_0 = 0.01, KAPPA_0 = 0.01, SIGMA_MU = 100, ALPHA_SIGMA = 1, BETA_SIGMA = 0.0001;

constexpr int MAX_K = 8;

struct MarkerSums {
    double vi_sum, vi_1, vi_2; // all individuals / genotype 1 / genotype 2 (vi_0 = the rest, missing calls included)
};

// one quadrature node: exp of the log integrand, src/BayesW.cpp:161-169
HG_BW_HD inline double gh_integrand(double s, double alpha, double dj, double sqrt_2Ck_sigmaG, double vi_sum, double vi_2, double vi_1,
                                    double vi_0, double sd, double mean_sd_ratio)
{
    const double temp = -alpha * s * dj * sqrt_2Ck_sigmaG + vi_sum -
                        exp(alpha * mean_sd_ratio * s * sqrt_2Ck_sigmaG) *
                            (vi_0 + vi_1 * exp(-alpha * s * sqrt_2Ck_sigmaG / sd) + vi_2 * exp(-2 * alpha * s * sqrt_2Ck_sigmaG / sd)) -
                        s * s;
    return exp(temp);
}

// sigma * (w_1 f(sigma x_1) + ... + w_{n-1} f(sigma x_{n-1}) + w_n), src/BayesW.cpp:174-709
HG_BW_HD inline double gh_integral(int n, const double* X, const double* W, double C_k, double sigma, double alpha, double sigmaG,
                                   double sum_failure, double vi_sum, double vi_2, double vi_1, double vi_0, double sd, double mean_sd_ratio)
{
    const double sqrt_2ck_sigma = sqrt(2 * C_k * sigmaG);
    double temp = 0.0;
    for (int q = 0; q < n - 1; ++q) {
        const double term = W[q] * gh_integrand(sigma * X[q], alpha, sum_failure, sqrt_2ck_sigma, vi_sum, vi_2, vi_1, vi_0, sd, mean_sd_ratio);
        temp = (q == 0) ? term : temp + term;
    }
    temp = temp + W[n - 1];
    return sigma * temp;
}

// ml[0..K-1], ml[0] = pi_0 * sqrt(pi); src/BayesW.cpp:713-726, :1476-1478
HG_BW_HD inline void marginals(int n, const double* X, const double* W, int K, const double* pi_row, const double* cva_row, double alpha,
                               double sigmaG, double sum_failure, const MarkerSums& s, double mean, double sd, double* ml)
{
    const double vi_0 = s.vi_sum - s.vi_1 - s.vi_2;
    ml[0] = pi_row[0] * SQRT_PI;
    const double exp_sum = (s.vi_1 * (1 - 2 * mean) + 4 * (1 - mean) * s.vi_2 + s.vi_sum * mean * mean) / (sd * sd);
    for (int i = 0; i < K - 1; ++i) {
        const double sigma = 1.0 / sqrt(1 + alpha * alpha * sigmaG * cva_row[i] * exp_sum);
        ml[i + 1] = pi_row[i + 1] * gh_integral(n, X, W, cva_row[i], sigma, alpha, sigmaG, sum_failure, s.vi_sum, s.vi_2, s.vi_1, vi_0, sd, mean / sd);
    }
}

// the categorical walk, as written at src/BayesW.cpp:1536-1598 (the cumulative jumps to 1 one step early)
HG_BW_HD inline int pick_component(int K, const double* ml, double p)
{
    const int km1 = K - 1;
    double sum = 0.0;
    for (int k = 0; k < K; ++k) sum += ml[k];
    double acum = ml[0] / sum;
    for (int k = 0; k < K; ++k) {
        if (p <= acum) return k;
        if ((k + 1) == km1) acum = 1;
        else if (k + 1 < K) acum += ml[k + 1] / sum;
    }
    return -1; // p is NaN or the likelihoods are: the reference would leave the marker untouched
}

// log density of one effect given the marker's masked sums, src/BayesW.cpp:145-156
struct BetaLogDensity {
    double alpha, sigmaG, sum_failure, sd, mean_sd_ratio, mixture_value, vi_0, vi_1, vi_2;
    HG_BW_HD double operator()(double x) const
    {
        return -alpha * x * sum_failure - exp(alpha * x * mean_sd_ratio) * (vi_0 + vi_1 * exp(-alpha * x / sd) + vi_2 * exp(-2 * alpha * x / sd)) -
               x * x / (2 * mixture_value * sigmaG);
    }
};

// sparse_scaadd's three values (src/BayesRRm.cpp:250-281) with sig_inv = 1/sd, as BayesW calls it
// (src/BayesW.cpp:1502-1506, :1611-1616): what eps gains at genotype 0 / 1 / 2 (missing gains 0)
HG_BW_HD inline void delta_values(double dMULT, double mu, double sd, double (&out)[3])
{
    const double sig_inv = 1 / sd;
    out[0] = -(mu * sig_inv * dMULT);
    out[1] = dMULT * (1.0 - mu) * sig_inv;
    out[2] = dMULT * (2.0 - mu) * sig_inv;
}

} // namespace bw
} // namespace hg
