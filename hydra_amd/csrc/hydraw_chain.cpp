// hydraw_chain.cpp -- layer 2 of the C ABI for BayesW: the body of
// `BayesW::runMpiGibbs_bW` (src/BayesW.cpp:905-2176) between data load and
// output, written on top of the hgibbs_w_* device operators.  Host code only:
// the scalar conditionals (mu, covariate effects, alpha) are drawn here by ARS
// (hg_ars.h) with every N-length sum inside their log densities computed on the
// device; the marker sweep is hgibbs_w_sweep.
//
// Order of one iteration (file:line in src/BayesW.cpp):
//   mu by ARS (:1334-1363) -> covariate effects by ARS in shuffled order (:1365-1415)
//   -> alpha by ARS (:1423-1453) -> vi (:1457-1459) -> shuffle (:1461-1463) -> marker
//   sweep (:1484-1622) -> m0, sigmaG ~ invGamma (:1879-1888) -> pi ~ Dirichlet (:1893-1898).
// Two generators, as in the reference: boost::mt19937 `dist.rng` for the shuffles,
// the per-marker uniform, sigmaG and pi; libc rand() for every ARS draw.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/hgibbs.h"
#include "hg_ars.h"
#include "hg_bayesw_math.h"
#include "hg_rng.h"

extern "C" void hgibbs_set_error_(const char* msg);

using namespace hg;

struct hydraw_chain {
    hgibbs_t dev = nullptr;
    uint32_t N = 0, M = 0;
    int G = 1, K = 0, quad = 0, shuffle = 1;
    std::vector<int32_t> groups, MtotGrp, order, cass, m0, fail;
    std::vector<double> pi, sigmaG;
    double mu = 0.0, alpha = 0.0, d = 0.0, sumSigmaG = 0.0;
    hgibbs_rng_state rng{};
    hgibbs_grand_state grand{};
    uint64_t last_nnz = 0;
    int C = 0;
    std::vector<double> gamma, sum_failure_fix;
    std::vector<unsigned int> xI;
    uint32_t iteration = 0;
    uint32_t row_begin = 0;
};

static int wfail(const std::string& m)
{
    hgibbs_set_error_(m.c_str());
    return 1;
}

namespace {

struct GrandUniform {
    GlibcRand* g;
    double operator()() { return g->uniform(); }
};

// a log density whose N-length sum lives on the device; a failed device call is remembered and
// turns the value into NaN so that the sampler stops at its next comparison
struct DeviceDensity {
    hgibbs_t dev;
    int kind, col;
    double alpha, lin, prior_var, ef;
    bool failed = false;
    double operator()(double x)
    {
        double s = 0.0;
        int rc;
        if (kind == 1) { // alpha_dens, :132-142: (alpha_0 + d - 1) log x + x (sum eps*fail - kappa_0) - sum exp(eps x - EuMasc)
            rc = hgibbs_w_reduce(dev, 1, 0, x, 0.0, 0.0, &s);
            if (rc) failed = true;
            return (bw::ALPHA_0 + lin - 1) * std::log(x) + x * (ef - bw::KAPPA_0) - s;
        }
        // mu_dens :77-88 / gamma_dens :118-129: -alpha x d - sum exp((used - [x_i] x) alpha - EuMasc) - x^2 / (2 sigma_mu)
        rc = hgibbs_w_reduce(dev, kind, col, 0.0, x, alpha, &s);
        if (rc) failed = true;
        return -alpha * x * lin - s - x * x / (2 * prior_var);
    }
};

template <class F>
int draw(F& f, const double (&xinit)[4], double xl, double xr, hgibbs_grand_state* grand, double& out, const char* what)
{
    GrandUniform u{reinterpret_cast<GlibcRand*>(grand)};
    ars::Hull hull;
    int neval = 0;
    const int err = ars::sample(xinit, xl, xr, f, u, hull, out, neval);
    if (f.failed) return 1; // hgibbs_last_error() already says why
    if (err) return wfail(std::string("Error code = ") + std::to_string(err) + " (ARS on " + what + ")"); // errorCheck, :64-69
    return 0;
}

} // namespace

extern "C" {

/* The device handle must hold the genotypes (hgibbs_load_bed / hgibbs_synth_bed).  y: log-time of the
 * n_global kept individuals (not centred or scaled, src/BayesW.cpp:1280-1296); failure: 1 = event
 * observed, 0 = censored.  Init = :729-866, :934-937, :1012, :1201-1232. */
int hydraw_chain_create(hgibbs_t dev, const hydraw_model_desc* model, const double* y_host, const int32_t* failure_host, hydraw_chain_t* out)
{
    if (!dev || !model || !y_host || !failure_host || !out) return wfail("hydraw_chain_create: null argument");
    uint32_t n_global = 0, n_local = 0, M = 0, row_begin = 0;
    if (hgibbs_dims(dev, &n_global, &n_local, &M, &row_begin)) return 1;
    if (n_global < 2 || M == 0) return wfail("hydraw_chain_create: load genotypes first (hgibbs_load_bed / hgibbs_synth_bed)");
    if (model->K < 2) return wfail("hydraw_chain_create: K must be >= 2 (zero component + at least one mixture)");
    if (hgibbs_w_init(dev, failure_host)) return 1;
    if (hgibbs_w_set_model(dev, model->G, model->K, model->groups, model->mS, model->quad_points)) return 1;
    if (hgibbs_w_marker_stats(dev, nullptr, nullptr, nullptr)) return 1;
    hydraw_chain* c = new hydraw_chain();
    c->dev = dev;
    c->N = n_global;
    c->M = M;
    c->row_begin = row_begin;
    c->G = model->G;
    c->K = model->K;
    c->quad = model->quad_points;
    c->shuffle = model->shuffle;
    const int G = c->G, K = c->K;
    c->groups.assign(M, 0);
    if (model->groups) c->groups.assign(model->groups, model->groups + M);
    c->MtotGrp.assign(G, 0);
    for (uint32_t i = 0; i < M; ++i) c->MtotGrp[c->groups[i]] += 1;
    // :797-799 -- 1/Mtot everywhere, 0.99 in column 0, column 1 = 1 - 0.99 - (km1 - 1)/Mtot (unsigned integer division)
    c->pi.assign((size_t)G * K, 1.0 / M);
    const unsigned int_div = (unsigned)(K - 1 - 1) / (unsigned)M;
    for (int g = 0; g < G; ++g) {
        c->pi[(size_t)g * K + 0] = 0.99;
        c->pi[(size_t)g * K + 1] = 1 - c->pi[(size_t)g * K + 0] - int_div;
    }
    c->fail.assign(failure_host, failure_host + n_global);
    c->cass.assign((size_t)G * K, 0);
    c->m0.assign(G, 0);
    c->order.resize(M);
    for (uint32_t i = 0; i < M; ++i) c->order[i] = (int32_t)i;

    double s = 0.0;
    for (uint32_t i = 0; i < n_global; ++i) s += y_host[i];
    c->mu = s / (double)n_global; // :811
    double ss = 0.0;
    for (uint32_t i = 0; i < n_global; ++i) ss += (y_host[i] - c->mu) * (y_host[i] - c->mu);
    const double denominator = (6 * ss / (double)(n_global - 1)); // :817
    c->alpha = bw::PI_BW / std::sqrt(denominator);
    std::vector<double> eps(n_global);
    for (uint32_t i = 0; i < n_global; ++i) eps[i] = y_host[i] - c->mu; // :823-826
    if (hgibbs_set_residual(dev, eps.data() + row_begin)) { // this rank's rows
        delete c;
        return 1;
    }
    c->sigmaG.assign(G, bw::PI_SQUARED / (6 * std::pow(c->alpha, 2)) / G); // :828
    c->sumSigmaG = 0.0;
    for (int g = 0; g < G; ++g) c->sumSigmaG += c->sigmaG[g];
    c->d = 0.0;
    for (uint32_t i = 0; i < n_global; ++i) c->d += (double)failure_host[i];

    hgibbs_grand_seed(&c->grand, model->seed); // srand(opt.seed), :1012
    Mt gen{c->rng.x, 0};
    gen.seed(model->seed); // dist.reset_rng(seed + rank*1000), rank 0, :937
    c->rng.idx = gen.idx;
    *out = c;
    return 0;
}

int hydraw_chain_destroy(hydraw_chain_t c)
{
    delete c;
    return 0;
}

int hydraw_chain_set_covariates(hydraw_chain_t c, const double* X_host, int C)
{
    if (!c) return wfail("hydraw_chain_set_covariates: null chain");
    if (C < 0 || (C > 0 && !X_host)) return wfail("hydraw_chain_set_covariates: bad argument");
    if (c->iteration != 0) return wfail("hydraw_chain_set_covariates: call before the first iteration");
    c->C = C;
    c->gamma.assign(C, 0.0);
    c->xI.resize(C);
    for (int i = 0; i < C; ++i) c->xI[i] = (unsigned)i;
    c->sum_failure_fix.assign(C, 0.0);
    for (int k = 0; k < C; ++k) { // :1236-1240
        double s = 0.0;
        for (uint32_t i = 0; i < c->N; ++i) s += X_host[(size_t)i * C + k] * (double)c->fail[i];
        c->sum_failure_fix[k] = s;
    }
    return hgibbs_set_covariates(c->dev, C ? X_host + (size_t)c->row_begin * C : nullptr, C);
}

/* srand(seed) between iterations: the reference reseeds at every checkpoint (:2029) and after a restart (:877) */
int hydraw_chain_reseed_ars(hydraw_chain_t c, uint32_t seed)
{
    if (!c) return wfail("hydraw_chain_reseed_ars: null chain");
    hgibbs_grand_seed(&c->grand, seed);
    return 0;
}

int hydraw_chain_restore(hydraw_chain_t c, const hydraw_restart_state* st)
{
    if (!c || !st || !st->sigmaG || !st->pi || !st->beta || !st->components || !st->eps || !st->order)
        return wfail("hydraw_chain_restore: null argument");
    if (c->C > 0 && (!st->gamma || !st->xI)) return wfail("hydraw_chain_restore: covariates set but no gamma/xI given");
    const int G = c->G, K = c->K;
    for (uint32_t i = 0; i < c->M; ++i)
        if (st->order[i] < 0 || (uint32_t)st->order[i] >= c->M) return wfail("hydraw_chain_restore: marker index out of range");
    c->mu = st->mu;
    c->alpha = st->alpha;
    std::copy(st->sigmaG, st->sigmaG + G, c->sigmaG.begin());
    std::copy(st->pi, st->pi + (size_t)G * K, c->pi.begin());
    std::copy(st->order, st->order + c->M, c->order.begin());
    for (int i = 0; i < c->C; ++i) {
        c->gamma[i] = st->gamma[i];
        c->xI[i] = (unsigned)st->xI[i];
    }
    c->sumSigmaG = 0.0; // :1018
    for (int g = 0; g < G; ++g) c->sumSigmaG += c->sigmaG[g];
    c->rng = st->rng;
    hgibbs_grand_seed(&c->grand, st->ars_seed);
    if (hgibbs_w_set_beta(c->dev, st->beta, st->components)) return 1;
    if (hgibbs_set_residual(c->dev, st->eps)) return 1; /* this rank's rows */
    c->iteration = st->iteration + 1;
    return 0;
}

int hydraw_chain_iterate(hydraw_chain_t c)
{
    if (!c) return wfail("hydraw_chain_iterate: null chain");
    const int G = c->G, K = c->K;
    Mt gen{c->rng.x, c->rng.idx};

    // 1. intercept, :1334-1363
    {
        const double mu = c->mu;
        const double xinit[4] = {0.95 * mu, mu, 1.005 * mu, 1.01 * mu};
        if (hgibbs_add_scalar(c->dev, mu)) return 1; // used_data.epsilon = epsilon + mu
        DeviceDensity f{c->dev, 0, 0, c->alpha, c->d, bw::SIGMA_MU, 0.0};
        double x = 0.0;
        if (draw(f, xinit, 0.8 * mu, 1.1 * mu, &c->grand, x, "mu")) return 1;
        c->mu = x;
        if (hgibbs_add_scalar(c->dev, -c->mu)) return 1;
    }
    // 1a. fixed effects, :1365-1415
    if (c->C > 0) {
        hg::shuffle_libstdcxx6(c->xI.data(), c->xI.size(), gen);
        for (int i = 0; i < c->C; ++i) {
            const int col = (int)c->xI[i];
            const double gamma_old = c->gamma[col];
            const double xinit[4] = {gamma_old - 0.075 / 30, gamma_old, gamma_old + 0.075 / 60, gamma_old + 0.075 / 30};
            if (hgibbs_cov_update(c->dev, col, gamma_old)) return 1; // used = eps + X_j * gamma_old
            DeviceDensity f{c->dev, 2, col, c->alpha, c->sum_failure_fix[col], bw::SIGMA_MU, 0.0};
            double x = 0.0;
            if (draw(f, xinit, gamma_old - 0.075, gamma_old + 0.075, &c->grand, x, "a covariate effect")) return 1;
            c->gamma[col] = x;
            if (hgibbs_cov_update(c->dev, col, -c->gamma[col])) return 1;
        }
    }
    // 2. alpha, :1423-1453
    {
        const double a = c->alpha;
        const double xinit[4] = {a * 0.5, a, a * 1.05, a * 1.10};
        double ef = 0.0;
        if (hgibbs_w_reduce(c->dev, 3, 0, 0.0, 0.0, 0.0, &ef)) return 1;
        DeviceDensity f{c->dev, 1, 0, 0.0, c->d, 0.0, ef};
        double x = 0.0;
        if (draw(f, xinit, 0.0, a * 1.30, &c->grand, x, "alpha")) return 1;
        c->alpha = x;
    }
    if (hgibbs_w_refresh_vi(c->dev, c->alpha)) return 1; // :1457-1459

    if (c->shuffle) { // :1461-1463
        hg::shuffle_libstdcxx6(c->order.data(), c->order.size(), gen);
    }
    std::fill(c->m0.begin(), c->m0.end(), 0);
    c->rng.idx = gen.idx;
    std::vector<double> bsq(G, 0.0);
    if (hgibbs_w_sweep(c->dev, c->order.data(), c->alpha, c->sigmaG.data(), c->pi.data(), c->sumSigmaG, &c->rng, &c->grand, c->cass.data(),
                       bsq.data(), &c->last_nnz))
        return 1;
    gen.idx = c->rng.idx;

    for (int g = 0; g < G; ++g) c->m0[g] = c->MtotGrp[g] - c->cass[(size_t)g * K]; // :1879-1881
    for (int g = 0; g < G; ++g)                                                    // :1886-1888
        c->sigmaG[g] = inv_gamma_rng(gen, (double)(bw::ALPHA_SIGMA + 0.5 * c->m0[g]), (double)(bw::BETA_SIGMA + 0.5 * (double)c->m0[g] * bsq[g]));
    std::vector<double> dirin(K), out(K);
    for (int g = 0; g < G; ++g) { // :1893-1898
        for (int k = 0; k < K; ++k) dirin[k] = (double)(c->cass[(size_t)g * K + k] + 1);
        dirichlet_rng(gen, dirin.data(), K, out.data());
        for (int k = 0; k < K; ++k) c->pi[(size_t)g * K + k] = out[k];
    }
    c->sumSigmaG = 0.0;
    for (int g = 0; g < G; ++g) c->sumSigmaG += c->sigmaG[g];
    c->rng.idx = gen.idx;
    c->iteration += 1;
    return 0;
}

int hydraw_chain_state(hydraw_chain_t c, double* mu, double* alpha, double* sigmaG, double* pi, int32_t* m0, int32_t* cass,
                       hgibbs_rng_state* rng, hgibbs_grand_state* ars_rng)
{
    if (!c) return wfail("hydraw_chain_state: null chain");
    if (mu) *mu = c->mu;
    if (alpha) *alpha = c->alpha;
    if (sigmaG) std::copy(c->sigmaG.begin(), c->sigmaG.end(), sigmaG);
    if (pi) std::copy(c->pi.begin(), c->pi.end(), pi);
    if (m0) std::copy(c->m0.begin(), c->m0.end(), m0);
    if (cass) std::copy(c->cass.begin(), c->cass.end(), cass);
    if (rng) *rng = c->rng;
    if (ars_rng) *ars_rng = c->grand;
    return 0;
}

int hydraw_chain_gamma(hydraw_chain_t c, double* gamma_out, int32_t* xI_out)
{
    if (!c) return wfail("hydraw_chain_gamma: null chain");
    for (int i = 0; i < c->C; ++i) {
        if (gamma_out) gamma_out[i] = c->gamma[i];
        if (xI_out) xI_out[i] = (int32_t)c->xI[i];
    }
    return 0;
}

const int32_t* hydraw_chain_order(hydraw_chain_t c) { return c ? c->order.data() : nullptr; }
uint64_t hydraw_chain_last_nnz(hydraw_chain_t c) { return c ? c->last_nnz : 0; }

/* .csv line, src/BayesW.cpp:1942-1963: it, mu, sum sigmaG, alpha, h2, sum m0, G, K, sigmaG per group, pi */
int hydraw_chain_csv_line(hydraw_chain_t c, uint32_t iteration, char* buf, size_t len)
{
    if (!c || !buf) return -1;
    const int G = c->G, K = c->K;
    double sg = 0.0;
    int m0s = 0;
    for (int g = 0; g < G; ++g) {
        sg += c->sigmaG[g];
        m0s += c->m0[g];
    }
    size_t n = (size_t)std::snprintf(buf, len, "%5d, %20.15f, %20.15f, %20.15f, %20.15f, %7d, %7d, %2d", iteration, c->mu, sg, c->alpha,
                                     sg / (sg + bw::PI_SQUARED / (6 * c->alpha * c->alpha)), m0s, G, K);
    for (int g = 0; g < G && n < len; ++g) n += (size_t)std::snprintf(buf + n, len - n, ", %20.15f", c->sigmaG[g]);
    for (int i = 0; i < G * K && n < len; ++i) n += (size_t)std::snprintf(buf + n, len - n, ", %20.15f", c->pi[i]);
    if (n < len) n += (size_t)std::snprintf(buf + n, len - n, "\n");
    return (int)n;
}

} // extern "C"
