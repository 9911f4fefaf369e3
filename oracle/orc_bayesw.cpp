/* oracle/orc_bayesw.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Single-threaded fp64 CPU restatement of hydra's BayesW sampler (Weibull
 * survival model with a spike-and-slab mixture on marker effects),
 * `BayesW::runMpiGibbs_bW` (src/BayesW.cpp:905-2176) for world_size == 1,
 * --sync-rate 1, delta updates (opt.deltaUpdate, options.hpp:82), data from a
 * dense 2-bit PLINK .bed.  Helpers restated: the log densities
 * (src/BayesW.cpp:77-156), the adaptive Gauss-Hermite marginal likelihoods
 * (:161-726), init (:729-866), the marker loop (:1471-1622), the update and vi
 * refresh (:1812-1834), sigmaG / pi (:1889-1903) and the .csv line (:1942-1963).
 *
 * PARITY: the reference cannot be built here (Eigen + Boost + MPI absent), so
 * the chain as a whole is PARITY UNPINNED.  Pinned pieces: the ARS sampler
 * (orc_ars.h) against the reference's own arms() compiled from
 * src/BayesW_arms.cpp (oracle/_ref/libarms.so) -- and this file can run the
 * whole chain on that very arms() (orc_bw_set_arms); the quadrature constants
 * literal by literal (tools/gen_gh_tables.py); the Boost distributions as in
 * orc_rng.h.  Known restatement choices: Eigen's .sum()/.mean() reductions are
 * restated as sequential sums in index order (Eigen's vectorised tree order is
 * a build-time property of the reference); `exp` is libm's.
 *
 * Only tests/ and __graft_entry__.smoke() may link or call this.
 */
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "gh_tables.h"
#include "orc_ars.h"
#include "orc_rng.h"

namespace {

/* src/BayesW.cpp:38-42 */
const double BW_PI = 3.14159265359;
const double BW_PI_SQUARED = 9.86960440109;
const double BW_SQRT_PI = 1.77245385090552;
const double BW_EULER = 0.577215664901532;
/* src/BayesW.hpp:85-89 */
const double BW_ALPHA_0 = 0.01, BW_KAPPA_0 = 0.01, BW_SIGMA_MU = 100, BW_ALPHA_SIGMA = 1, BW_BETA_SIGMA = 0.0001;

inline int geno_at(const uint8_t* col, uint32_t i)
{
    const unsigned v = (col[i >> 2] >> (2 * (i & 3u))) & 3u; /* src/data.cpp:1189-1200 */
    if (v == 1u) return -1;
    return 2 - (int)((v & 1u) + ((v >> 1) & 1u));
}

typedef int (*arms_fn)(double*, int, double*, double*, double (*)(double, void*), void*, double*, int, int, double*, double*, int, double*,
                       double*, int, int*);

struct MuPars { /* struct pars, src/BayesW.hpp:20-45, as mu_dens / gamma_dens read it */
    const double* eps; /* used_data.epsilon */
    const double* Xj;  /* covariate column (stride C) or NULL */
    int stride;
    uint32_t n;
    double alpha, d, sum_failure, sigma_mu;
};
struct AlphaPars { /* struct pars_alpha, :70-79 */
    const double* eps;
    const double* fail;
    uint32_t n;
    double alpha_0, kappa_0, d;
};
struct BetaPars { /* struct pars_beta_sparse, :47-68 */
    double alpha, sigmaG, sum_failure, mean, sd, mean_sd_ratio, mixture_value, vi_0, vi_1, vi_2;
};

/* src/BayesW.cpp:77-88 */
double mu_dens(double x, void* v)
{
    const MuPars& p = *static_cast<MuPars*>(v);
    double s = 0.0;
    for (uint32_t i = 0; i < p.n; ++i) s += std::exp((p.eps[i] - x) * p.alpha - BW_EULER);
    return -p.alpha * x * p.d - s - x * x / (2 * p.sigma_mu);
}

/* src/BayesW.cpp:118-129 */
double gamma_dens(double x, void* v)
{
    const MuPars& p = *static_cast<MuPars*>(v);
    double s = 0.0;
    for (uint32_t i = 0; i < p.n; ++i) s += std::exp(((p.eps[i] - p.Xj[(size_t)i * p.stride] * x) * p.alpha) - BW_EULER);
    return -p.alpha * x * p.sum_failure - s - x * x / (2 * p.sigma_mu);
}

/* src/BayesW.cpp:132-142 */
double alpha_dens(double x, void* v)
{
    const AlphaPars& p = *static_cast<AlphaPars*>(v);
    double ef = 0.0, s = 0.0;
    for (uint32_t i = 0; i < p.n; ++i) ef += p.eps[i] * p.fail[i];
    for (uint32_t i = 0; i < p.n; ++i) s += std::exp((p.eps[i] * x) - BW_EULER);
    return (p.alpha_0 + p.d - 1) * std::log(x) + x * (ef - p.kappa_0) - s;
}

/* src/BayesW.cpp:145-156 */
double beta_dens(double x, void* v)
{
    const BetaPars& p = *static_cast<BetaPars*>(v);
    return -p.alpha * x * p.sum_failure -
           std::exp(p.alpha * x * p.mean_sd_ratio) * (p.vi_0 + p.vi_1 * std::exp(-p.alpha * x / p.sd) + p.vi_2 * std::exp(-2 * p.alpha * x / p.sd)) -
           x * x / (2 * p.mixture_value * p.sigmaG);
}

/* src/BayesW.cpp:161-169 */
inline double gh_integrand(double s, double alpha, double dj, double sqrt_2Ck_sigmaG, double vi_sum, double vi_2, double vi_1, double vi_0,
                           double sd, double mean_sd_ratio)
{
    const double temp = -alpha * s * dj * sqrt_2Ck_sigmaG + vi_sum -
                        std::exp(alpha * mean_sd_ratio * s * sqrt_2Ck_sigmaG) *
                            (vi_0 + vi_1 * std::exp(-alpha * s * sqrt_2Ck_sigmaG / sd) + vi_2 * std::exp(-2 * alpha * s * sqrt_2Ck_sigmaG / sd)) -
                        std::pow(s, 2);
    return std::exp(temp);
}

/* src/BayesW.cpp:174-709: w1 f(sigma x1) + ... + w_{n-1} f(sigma x_{n-1}) + w_n, times sigma */
double gh_integral(int n, double C_k, double sigma, const BetaPars& b, double vi_sum, double vi_2, double vi_1, double vi_0, double sd,
                   double mean_sd_ratio)
{
    const double* X;
    const double* W;
    orc_gh_lookup(n, &X, &W);
    const double sqrt_2ck_sigma = std::sqrt(2 * C_k * b.sigmaG);
    double temp = 0.0;
    for (int q = 0; q < n - 1; ++q) {
        const double xq = sigma * X[q];
        const double term = W[q] * gh_integrand(xq, b.alpha, b.sum_failure, sqrt_2ck_sigma, vi_sum, vi_2, vi_1, vi_0, sd, mean_sd_ratio);
        temp = (q == 0) ? term : temp + term;
    }
    temp = temp + W[n - 1];
    return sigma * temp;
}

} // namespace

struct orc_bw {
    uint32_t N, M;
    uint64_t stride;
    const uint8_t* bed;
    int G, K, quad;
    std::vector<int> groups, MtotGrp, order, components, cass, m0;
    std::vector<double> cVa; /* G x (K-1) */
    std::vector<double> pi;  /* G x K */
    std::vector<double> sigmaG, mave, msd, sum_failure, y, fail, eps, vi, beta;
    std::vector<double> work, work2;
    double mu, alpha, d, sumSigmaG;
    int shuffle;
    long last_nnz;
    long ars_evals;
    orc_mt rng;
    arms_fn arms;
    int C;
    std::vector<double> X, gamma, sum_failure_fix;
    std::vector<unsigned int> xI;
    int last_err;
};

extern "C" {

int orc_bw_quad_supported(int n)
{
    const double *x, *w;
    return orc_gh_lookup(n, &x, &w);
}

/* the restated sampler with the reference's signature, for the pin test */
int orc_ars_arms_c(double* xinit, int ninit, double* xl, double* xr, double (*f)(double, void*), void* data, double* convex, int npoint,
                   int dometrop, double* xprev, double* xsamp, int nsamp, double* qcent, double* xcent, int ncent, int* neval)
{
    return orc_ars_arms(xinit, ninit, xl, xr, f, data, convex, npoint, dometrop, xprev, xsamp, nsamp, qcent, xcent, ncent, neval);
}

/* beta_dens with plain arguments, for cross-checks of the product's copy */
double orc_bw_beta_dens(double x, double alpha, double sigmaG, double sum_failure, double mean, double sd, double mixture_value, double vi_0,
                        double vi_1, double vi_2)
{
    BetaPars p{alpha, sigmaG, sum_failure, mean, sd, mean / sd, mixture_value, vi_0, vi_1, vi_2};
    return beta_dens(x, &p);
}

/* marginal likelihoods of one marker (src/BayesW.cpp:713-726): out[0..K-1], out[0] = pi_0 * sqrt(pi) */
void orc_bw_marginals(int quad, int K, const double* pi_row, const double* cVa_row, double alpha, double sigmaG, double sum_failure,
                      double vi_sum, double vi_2, double vi_1, double vi_0, double mean, double sd, double* out)
{
    BetaPars b{};
    b.alpha = alpha;
    b.sigmaG = sigmaG;
    b.sum_failure = sum_failure;
    out[0] = pi_row[0] * BW_SQRT_PI;
    const double exp_sum = (vi_1 * (1 - 2 * mean) + 4 * (1 - mean) * vi_2 + vi_sum * mean * mean) / (sd * sd);
    for (int i = 0; i < K - 1; ++i) {
        const double sigma = 1.0 / std::sqrt(1 + alpha * alpha * sigmaG * cVa_row[i] * exp_sum);
        out[i + 1] = pi_row[i + 1] * gh_integral(quad, cVa_row[i], sigma, b, vi_sum, vi_2, vi_1, vi_0, sd, mean / sd);
    }
}

/* Init = src/BayesW.cpp:729-866 (model, priors, mu, alpha, sigmaG), :1012-1013 (srand(seed)),
 * :934-937 (dist seed), :1201-1232 (marker statistics, sum_failure).  y: log-time of the kept
 * individuals (not centred, :1280-1296), fail: 0/1 per kept individual; mS is G x K with column 0 == 0. */
orc_bw* orc_bw_create(const uint8_t* bed, uint64_t stride, uint32_t N, uint32_t M, const double* y, const double* fail, int G, int K,
                      const int* groups, const double* mS, uint32_t seed, int shuffle, int quad)
{
    orc_bw* c = new orc_bw();
    c->N = N;
    c->M = M;
    c->stride = stride;
    c->bed = bed;
    c->G = G;
    c->K = K;
    c->quad = quad;
    c->shuffle = shuffle;
    c->arms = orc_ars_arms;
    c->C = 0;
    c->last_err = 0;
    c->ars_evals = 0;
    c->groups.assign(groups, groups + M);
    c->MtotGrp.assign(G, 0);
    for (uint32_t i = 0; i < M; ++i) c->MtotGrp[groups[i]] += 1;
    c->cVa.resize((size_t)G * (K - 1));
    for (int g = 0; g < G; ++g)
        for (int k = 1; k < K; ++k) c->cVa[(size_t)g * (K - 1) + (k - 1)] = mS[(size_t)g * K + k];
    /* :797-799: 1/Mtot everywhere, 0.99 in column 0, column 1 = 1 - 0.99 - (km1-1)/Mtot with an unsigned integer division */
    c->pi.assign((size_t)G * K, 1.0 / M);
    const unsigned int_div = (unsigned)(K - 1 - 1) / (unsigned)M;
    for (int g = 0; g < G; ++g) {
        c->pi[(size_t)g * K + 0] = 0.99;
        c->pi[(size_t)g * K + 1] = 1 - c->pi[(size_t)g * K + 0] - int_div;
    }
    c->y.assign(y, y + N);
    c->fail.assign(fail, fail + N);
    c->eps.resize(N);
    c->vi.resize(N);
    c->work.resize(N);
    c->work2.resize(N);
    c->beta.assign(M, 0.0);
    c->components.assign(M, 0);
    c->cass.assign((size_t)G * K, 0);
    c->m0.assign(G, 0);
    c->order.resize(M);
    for (uint32_t i = 0; i < M; ++i) c->order[i] = (int)i;
    double s = 0.0;
    for (uint32_t i = 0; i < N; ++i) s += y[i];
    c->mu = s / (double)N; /* :811 */
    double ss = 0.0;
    for (uint32_t i = 0; i < N; ++i) ss += (y[i] - c->mu) * (y[i] - c->mu);
    const double denominator = (6 * ss / (double)(N - 1)); /* :817 */
    c->alpha = BW_PI / std::sqrt(denominator);
    for (uint32_t i = 0; i < N; ++i) c->eps[i] = y[i] - c->mu;
    c->sigmaG.assign(G, BW_PI_SQUARED / (6 * std::pow(c->alpha, 2)) / G); /* :828 */
    c->sumSigmaG = 0.0;
    for (int g = 0; g < G; ++g) c->sumSigmaG += c->sigmaG[g];
    c->d = 0.0;
    for (uint32_t i = 0; i < N; ++i) c->d += fail[i];

    /* :1201-1232 */
    c->mave.resize(M);
    c->msd.resize(M);
    c->sum_failure.resize(M);
    const double dN = (double)N;
    for (uint32_t j = 0; j < M; ++j) {
        const uint8_t* col = bed + (size_t)j * stride;
        uint64_t n1 = 0, n2 = 0, nm = 0;
        int fsum = 0;
        for (uint32_t i = 0; i < N; ++i) {
            const int g = geno_at(col, i);
            if (g == 1) {
                ++n1;
                fsum += (int)fail[i];
            } else if (g == 2) {
                ++n2;
                fsum += 2 * (int)fail[i];
            } else if (g < 0) {
                ++nm;
            }
        }
        const double mave = ((double)n1 + 2.0 * (double)n2) / (dN - (double)nm);
        const double tmp1 = (double)n1 * (1.0 - mave) * (1.0 - mave);
        const double tmp2 = (double)n2 * (2.0 - mave) * (2.0 - mave);
        const double tmp0 = (double)(N - n1 - n2 - nm) * (0.0 - mave) * (0.0 - mave);
        c->mave[j] = mave;
        c->msd[j] = std::sqrt((tmp0 + tmp1 + tmp2) / (double)(N - 1));
        c->sum_failure[j] = ((double)fsum - mave * c->d) / c->msd[j];
    }
    srand(seed);             /* :1012 */
    orc_mt_seed(&c->rng, seed); /* :937, rank 0 */
    return c;
}

void orc_bw_destroy(orc_bw* c) { delete c; }

/* run the chain on another arms() with the same signature (the reference's own, from oracle/_ref/libarms.so) */
void orc_bw_set_arms(orc_bw* c, void* fn) { c->arms = fn ? (arms_fn)fn : (arms_fn)orc_ars_arms; }

void orc_bw_set_covariates(orc_bw* c, const double* X, int C)
{
    c->C = C;
    c->X.assign(X, X + (size_t)c->N * C);
    c->gamma.assign(C, 0.0);
    c->xI.resize(C);
    for (int i = 0; i < C; ++i) c->xI[i] = (unsigned)i;
    c->sum_failure_fix.assign(C, 0.0);
    for (int k = 0; k < C; ++k) { /* :1236-1240 */
        double s = 0.0;
        for (uint32_t i = 0; i < c->N; ++i) s += X[(size_t)i * C + k] * c->fail[i];
        c->sum_failure_fix[k] = s;
    }
}

/* sparse_scaadd (src/BayesRRm.cpp:250-281) as BayesW calls it: sig_inv = 1/sd */
static inline void delta_values(double dMULT, double mu, double sd, double out[3])
{
    const double sig_inv = 1 / sd;
    out[0] = -(mu * sig_inv * dMULT);
    out[1] = dMULT * (1.0 - mu) * sig_inv;
    out[2] = dMULT * (2.0 - mu) * sig_inv;
}

/* srand at a checkpoint (src/BayesW.cpp:2029) and after a restart (:877) */
void orc_bw_reseed_ars(orc_bw* c, uint32_t seed)
{
    (void)c;
    srand(seed);
}

/* one full iteration, src/BayesW.cpp:1326-1907; returns 0 or the ARS error code (the reference exits, :67-72) */
int orc_bw_iterate(orc_bw* c)
{
    const uint32_t N = c->N, M = c->M;
    const int G = c->G, K = c->K, km1 = K - 1;
    int err, ninit = 4, npoint = 100, nsamp = 1, ncent = 4, neval = 0, dometrop = 0;
    double xsamp[1], xcent[10], qcent[10] = {5., 30., 70., 95.};
    double convex = 1.0, xprev = 0.0;
    double* eps = c->eps.data();
    double* vi = c->vi.data();
    double* used = c->work.data();

    /* 1. intercept, :1334-1363 */
    {
        const double mu = c->mu;
        double xinit[4] = {0.95 * mu, mu, 1.005 * mu, 1.01 * mu};
        double xl = 0.8 * mu, xr = 1.1 * mu;
        for (uint32_t i = 0; i < N; ++i) used[i] = eps[i] + mu;
        MuPars p{used, nullptr, 0, N, c->alpha, c->d, 0.0, BW_SIGMA_MU};
        err = c->arms(xinit, ninit, &xl, &xr, mu_dens, &p, &convex, npoint, dometrop, &xprev, xsamp, nsamp, qcent, xcent, ncent, &neval);
        c->ars_evals += neval;
        if (err) return c->last_err = err;
        c->mu = xsamp[0];
        for (uint32_t i = 0; i < N; ++i) eps[i] = used[i] - c->mu;
    }
    /* 1a. fixed effects, :1365-1415 */
    if (c->C > 0) {
        orc_shuffle_u32(&c->rng, c->xI.data(), c->xI.size());
        for (int fix_i = 0; fix_i < c->C; ++fix_i) {
            const unsigned col = c->xI[fix_i];
            const double gamma_old = c->gamma[col];
            neval = 0;
            xsamp[0] = 0;
            convex = 1.0;
            xprev = 0.0;
            double xinit[4] = {gamma_old - 0.075 / 30, gamma_old, gamma_old + 0.075 / 60, gamma_old + 0.075 / 30};
            double xl = gamma_old - 0.075, xr = gamma_old + 0.075;
            const double* Xj = c->X.data() + col;
            for (uint32_t k = 0; k < N; ++k) used[k] = eps[k] + Xj[(size_t)k * c->C] * gamma_old;
            MuPars p{used, Xj, c->C, N, c->alpha, c->d, c->sum_failure_fix[col], BW_SIGMA_MU};
            err = c->arms(xinit, ninit, &xl, &xr, gamma_dens, &p, &convex, npoint, dometrop, &xprev, xsamp, nsamp, qcent, xcent, ncent, &neval);
            c->ars_evals += neval;
            if (err) return c->last_err = err;
            c->gamma[col] = xsamp[0];
            for (uint32_t k = 0; k < N; ++k) eps[k] = used[k] - Xj[(size_t)k * c->C] * c->gamma[col];
        }
    }
    /* 2. alpha, :1423-1453 */
    {
        neval = 0;
        xsamp[0] = 0;
        convex = 1.0;
        xprev = 0.0;
        double xinit[4] = {c->alpha * 0.5, c->alpha, c->alpha * 1.05, c->alpha * 1.10};
        double xl = 0.0, xr = c->alpha * 1.30;
        AlphaPars p{eps, c->fail.data(), N, BW_ALPHA_0, BW_KAPPA_0, c->d};
        err = c->arms(xinit, ninit, &xl, &xr, alpha_dens, &p, &convex, npoint, dometrop, &xprev, xsamp, nsamp, qcent, xcent, ncent, &neval);
        c->ars_evals += neval;
        if (err) return c->last_err = err;
        c->alpha = xsamp[0];
    }
    for (uint32_t i = 0; i < N; ++i) vi[i] = std::exp(c->alpha * eps[i] - BW_EULER); /* :1457-1459 */

    if (c->shuffle) { /* :1461-1463 */
        orc_shuffle_u32(&c->rng, reinterpret_cast<uint32_t*>(c->order.data()), c->order.size());
    }
    std::fill(c->m0.begin(), c->m0.end(), 0);
    std::fill(c->cass.begin(), c->cass.end(), 0);
    std::vector<double> ml0(G), ml(K), bsq(G, 0.0);
    for (int g = 0; g < G; ++g) ml0[g] = c->pi[(size_t)g * K] * BW_SQRT_PI; /* :1476-1478 */
    long nnz = 0;

    for (uint32_t j = 0; j < M; ++j) { /* :1484-1622 */
        const int marker = c->order[j];
        const int grp = c->groups[marker];
        const uint8_t* col = c->bed + (size_t)marker * c->stride;
        const double mave = c->mave[marker], sd = c->msd[marker];
        const double beta_old = c->beta[marker];
        BetaPars b{};
        b.alpha = c->alpha;
        b.sigmaG = c->sigmaG[grp];
        double vi_sum = 0.0, vi_1 = 0.0, vi_2 = 0.0;
        const double* v = vi;
        if (beta_old != 0) { /* :1499-1516: the residual without this marker's effect */
            double dv[3];
            delta_values(beta_old, mave, sd, dv);
            double* tv = c->work2.data();
            for (uint32_t i = 0; i < N; ++i) {
                const int g = geno_at(col, i);
                const double de = (g < 0) ? 0.0 : dv[g];
                tv[i] = std::exp(c->alpha * (eps[i] + de) - BW_EULER);
            }
            v = tv;
        }
        for (uint32_t i = 0; i < N; ++i) vi_sum += v[i];
        for (uint32_t i = 0; i < N; ++i)
            if (geno_at(col, i) == 2) vi_2 += v[i];
        for (uint32_t i = 0; i < N; ++i)
            if (geno_at(col, i) == 1) vi_1 += v[i];
        const double vi_0 = vi_sum - vi_1 - vi_2;

        const double p = orc_unif_rng(&c->rng); /* :1528 */
        b.sum_failure = c->sum_failure[marker];
        ml[0] = ml0[grp];
        {
            const double exp_sum = (vi_1 * (1 - 2 * mave) + 4 * (1 - mave) * vi_2 + vi_sum * mave * mave) / (sd * sd);
            for (int i = 0; i < km1; ++i) {
                const double cva = c->cVa[(size_t)grp * km1 + i];
                const double sigma = 1.0 / std::sqrt(1 + b.alpha * b.alpha * b.sigmaG * cva * exp_sum);
                ml[i + 1] = c->pi[(size_t)grp * K + i + 1] * gh_integral(c->quad, cva, sigma, b, vi_sum, vi_2, vi_1, vi_0, sd, mave / sd);
            }
        }
        auto mlsum = [&]() {
            double s = 0.0;
            for (int k = 0; k < K; ++k) s += ml[k];
            return s;
        };
        double acum = ml[0] / mlsum(); /* :1536 */
        for (int k = 0; k < K; ++k) {
            if (p <= acum) {
                if (k == 0) {
                    c->beta[marker] = 0;
                    c->cass[(size_t)grp * K + 0] += 1;
                    c->components[marker] = k;
                } else {
                    b.mean = mave;
                    b.sd = sd;
                    b.mean_sd_ratio = mave / sd;
                    b.mixture_value = c->cVa[(size_t)grp * km1 + (k - 1)];
                    b.vi_0 = vi_0;
                    b.vi_1 = vi_1;
                    b.vi_2 = vi_2;
                    const double safe_limit = 2 * std::sqrt(c->sumSigmaG * b.mixture_value);
                    neval = 0;
                    xsamp[0] = 0;
                    convex = 1.0;
                    xprev = 0.0;
                    double xinit[4] = {beta_old - safe_limit / 10, beta_old, beta_old + safe_limit / 20, beta_old + safe_limit / 10};
                    double xl = beta_old - safe_limit, xr = beta_old + safe_limit;
                    err = c->arms(xinit, ninit, &xl, &xr, beta_dens, &b, &convex, npoint, dometrop, &xprev, xsamp, nsamp, qcent, xcent, ncent,
                                  &neval);
                    c->ars_evals += neval;
                    if (err) return c->last_err = err;
                    c->beta[marker] = xsamp[0];
                    c->cass[(size_t)grp * K + k] += 1;
                    c->components[marker] = k;
                    bsq[grp] += c->beta[marker] * c->beta[marker];
                }
                break;
            } else {
                if ((k + 1) == km1) acum = 1; /* :1592-1596, as written in the reference */
                else acum += ml[k + 1] / mlsum();
            }
        }
        const double deltaBeta = beta_old - c->beta[marker];
        if (deltaBeta != 0.0) { /* :1606-1622, :1812, :1832-1834 */
            double dv[3];
            delta_values(deltaBeta, mave, sd, dv);
            for (uint32_t i = 0; i < N; ++i) {
                const int g = geno_at(col, i);
                const double de = (g < 0) ? 0.0 : dv[g];
                eps[i] = eps[i] + (0.0 + de);
            }
            for (uint32_t i = 0; i < N; ++i) vi[i] = std::exp(c->alpha * eps[i] - BW_EULER);
            ++nnz;
        }
    }
    c->last_nnz = nnz;

    for (int g = 0; g < G; ++g) c->m0[g] = c->MtotGrp[g] - c->cass[(size_t)g * K]; /* :1879-1881 */
    for (int g = 0; g < G; ++g)                                                   /* :1886-1888 */
        c->sigmaG[g] = orc_inv_gamma_rng(&c->rng, (double)(BW_ALPHA_SIGMA + 0.5 * c->m0[g]), (double)(BW_BETA_SIGMA + 0.5 * (double)c->m0[g] * bsq[g]));
    std::vector<double> dirin(K), out(K);
    for (int g = 0; g < G; ++g) { /* :1893-1898 */
        for (int k = 0; k < K; ++k) dirin[k] = (double)(c->cass[(size_t)g * K + k] + 1);
        orc_dirichlet_rng(&c->rng, dirin.data(), K, out.data());
        for (int k = 0; k < K; ++k) c->pi[(size_t)g * K + k] = out[k];
    }
    c->sumSigmaG = 0.0;
    for (int g = 0; g < G; ++g) c->sumSigmaG += c->sigmaG[g];
    return 0;
}

/* BayesW::init_from_restart (src/BayesW.cpp:869-903) and the restart branch of runMpiGibbs_bW
 * (:1300-1311, :1018): dumped state over the regular init, srand(ars_seed) */
void orc_bw_restore(orc_bw* c, double mu, double alpha, const double* sigmaG, const double* pi, const double* beta, const int* components,
                    const double* eps, const int* order, const double* gamma, const int* xI, const uint32_t* rng_words, uint32_t ars_seed)
{
    c->mu = mu;
    c->alpha = alpha;
    for (int g = 0; g < c->G; ++g) c->sigmaG[g] = sigmaG[g];
    for (int i = 0; i < c->G * c->K; ++i) c->pi[i] = pi[i];
    for (uint32_t i = 0; i < c->M; ++i) {
        c->beta[i] = beta[i];
        c->components[i] = components[i];
        c->order[i] = order[i];
    }
    for (uint32_t i = 0; i < c->N; ++i) c->eps[i] = eps[i];
    for (int i = 0; i < c->C; ++i) {
        c->gamma[i] = gamma[i];
        c->xI[i] = (unsigned)xI[i];
    }
    c->sumSigmaG = 0.0;
    for (int g = 0; g < c->G; ++g) c->sumSigmaG += c->sigmaG[g];
    orc_mt_load_words(&c->rng, rng_words);
    srand(ars_seed);
}

double* orc_bw_beta(orc_bw* c) { return c->beta.data(); }
int* orc_bw_components(orc_bw* c) { return c->components.data(); }
double* orc_bw_eps(orc_bw* c) { return c->eps.data(); }
double* orc_bw_vi(orc_bw* c) { return c->vi.data(); }
double* orc_bw_sigmaG(orc_bw* c) { return c->sigmaG.data(); }
double* orc_bw_pi(orc_bw* c) { return c->pi.data(); }
double* orc_bw_mave(orc_bw* c) { return c->mave.data(); }
double* orc_bw_msd(orc_bw* c) { return c->msd.data(); }
double* orc_bw_sum_failure(orc_bw* c) { return c->sum_failure.data(); }
double* orc_bw_gamma(orc_bw* c) { return c->gamma.data(); }
unsigned int* orc_bw_xI(orc_bw* c) { return c->xI.data(); }
int* orc_bw_order(orc_bw* c) { return c->order.data(); }
int* orc_bw_cass(orc_bw* c) { return c->cass.data(); }
int* orc_bw_m0(orc_bw* c) { return c->m0.data(); }
double orc_bw_mu(orc_bw* c) { return c->mu; }
double orc_bw_alpha(orc_bw* c) { return c->alpha; }
long orc_bw_last_nnz(orc_bw* c) { return c->last_nnz; }
long orc_bw_ars_evals(orc_bw* c) { return c->ars_evals; }
orc_mt* orc_bw_rng(orc_bw* c) { return &c->rng; }

/* .csv line, src/BayesW.cpp:1942-1963 */
int orc_bw_csv_line(orc_bw* c, uint32_t iteration, char* buf, size_t len)
{
    const int G = c->G, K = c->K;
    double sg = 0.0;
    int m0s = 0;
    for (int g = 0; g < G; ++g) {
        sg += c->sigmaG[g];
        m0s += c->m0[g];
    }
    size_t n = (size_t)snprintf(buf, len, "%5d, %20.15f, %20.15f, %20.15f, %20.15f, %7d, %7d, %2d", iteration, c->mu, sg, c->alpha,
                                sg / (sg + BW_PI_SQUARED / (6 * c->alpha * c->alpha)), m0s, G, K);
    for (int g = 0; g < G && n < len; ++g) n += (size_t)snprintf(buf + n, len - n, ", %20.15f", c->sigmaG[g]);
    for (int i = 0; i < G * K && n < len; ++i) n += (size_t)snprintf(buf + n, len - n, ", %20.15f", c->pi[i]);
    if (n < len) n += (size_t)snprintf(buf + n, len - n, "\n");
    return (int)n;
}

} /* extern "C" */
