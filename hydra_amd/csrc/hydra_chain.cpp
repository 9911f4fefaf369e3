// hydra_chain.cpp -- layer 2 of the C ABI: the body of BayesRRm::runMpiGibbs
// (src/BayesRRm.cpp:933-2939) for --mpibayes bayesMPI, restated on top of the
// hgibbs_* device operators.  Everything N- or M-sized stays on the GPU; this
// file only handles the K/G-sized hyper-parameters, the marker permutation and
// the host half of the shared MT19937 stream, in the reference's call order:
//
//   init      :1037-1110 model tables, :1228-1240 seed + sigmaG ~ beta(1,1),
//             :1502-1508 marker stats, :1564-1597 y -> eps, sigmaE, adaV
//   iteration :1675-1686 mu, :1691-1694 shuffle, :1696-1697 counters,
//             :1709-2490 marker sweep (device), :2495-2578 sigmaG + pi per
//             group, :2685-2690 sigmaE
//
// Every rank of a multi-GPU run executes this identically (replicated
// determinism): the only rank-dependent state is the individuals shard on the
// device, and every quantity derived from it arrives all-reduced.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../include/hgibbs.h"
#include "hg_rng.h"

extern "C" void hgibbs_set_error_(const char* msg);

namespace {
const double V0E = 0.0001, S02E = 0.0001, V0G = 0.0001, S02G = 0.0001; // src/BayesRRm.h:30-33
}

struct hydra_chain {
    hgibbs_t dev = nullptr;
    uint32_t N = 0, M = 0;
    int G = 1, K = 0;
    int shuffle = 1;
    std::vector<int32_t> groups, MtotGrp, order, cass, m0;
    std::vector<double> cVa, cVaI, priorPi, estPi, sigmaG;
    std::vector<uint8_t> adaV;
    double sigmaE = 0.0, mu = 0.0;
    hgibbs_rng_state rng{};
    uint64_t last_nnz = 0;
    // fixed effects (:1113-1114, :1552-1553)
    int C = 0;
    std::vector<double> gamma;
    std::vector<unsigned int> xI;
    uint32_t row_begin = 0, n_local = 0;
    uint32_t iteration = 0;
};

static int cfail(const std::string& m)
{
    hgibbs_set_error_(m.c_str());
    return 1;
}

extern "C" {

/* The device handle must already hold this rank's genotype shard
 * (hgibbs_load_bed / hgibbs_synth_bed).  y_host: n_global phenotypes (all
 * ranks pass the same vector); row range of this rank = what was loaded. */
int hydra_chain_create(hgibbs_t dev, const hydra_model_desc* model, const double* y_host, hydra_chain_t* out)
{
    if (!dev || !model || !y_host || !out) return cfail("hydra_chain_create: null argument");
    uint32_t n_global = 0, n_local = 0, M = 0, row_begin = 0;
    if (hgibbs_dims(dev, &n_global, &n_local, &M, &row_begin)) return 1;
    if (n_global < 2 || M == 0) return cfail("hydra_chain_create: load genotypes first (hgibbs_load_bed / hgibbs_synth_bed)");
    if (model->K < 2) return cfail("hydra_chain_create: K must be >= 2 (zero component + at least one mixture)");
    hydra_chain* c = new hydra_chain();
    c->dev = dev;
    c->N = n_global;
    c->M = M;
    c->row_begin = row_begin;
    c->n_local = n_local;
    c->G = model->G;
    c->K = model->K;
    c->shuffle = model->shuffle;
    const int G = c->G, K = c->K, km1 = K - 1;

    c->groups.assign(M, 0);
    if (model->groups) c->groups.assign(model->groups, model->groups + M);
    for (uint32_t i = 0; i < M; ++i)
        if (c->groups[i] < 0 || c->groups[i] >= G) {
            delete c;
            return cfail("hydra_chain_create: group index out of range");
        }

    // :1086-1110
    c->cVa.assign((size_t)G * K, 0.0);
    c->cVaI.assign((size_t)G * K, 0.0);
    c->priorPi.assign((size_t)G * K, 0.0);
    for (int g = 0; g < G; ++g) {
        c->priorPi[g * K] = 0.5;
        double s = 0.0;
        for (int k = 1; k <= km1; ++k) {
            const double v = model->mS[g * K + k];
            if (!(v > 0.0)) {
                delete c;
                return cfail("FATAL  : mixture value can only be strictly positive"); // :992-993
            }
            c->cVa[g * K + k] = v;
            c->cVaI[g * K + k] = 1.0 / v;
            s += v;
        }
        for (int k = 1; k <= km1; ++k) c->priorPi[g * K + k] = c->priorPi[g * K] * c->cVa[g * K + k] / s;
    }
    c->estPi = c->priorPi;
    c->sigmaG.assign(G, 0.0);
    c->cass.assign((size_t)G * K, 0);
    c->m0.assign(G, 0);
    c->MtotGrp.assign(G, 0);
    for (uint32_t i = 0; i < M; ++i) c->MtotGrp[c->groups[i]] += 1;

    if (hgibbs_set_model(dev, G, K, c->groups.data(), c->cVa.data(), c->cVaI.data())) {
        delete c;
        return 1;
    }

    // :1228-1240
    hg::Mt gen{c->rng.x, 0};
    gen.seed(model->seed);
    for (int g = 0; g < G; ++g) c->sigmaG[g] = hg::beta_rng(gen, 1.0, 1.0);
    for (int g = 0; g < G; ++g)
        if (c->MtotGrp[g] == 0) c->sigmaG[g] = 0.0;
    c->rng.idx = gen.idx;

    // :1502-1508 (device, from all-reduced counts)
    if (hgibbs_marker_stats(dev, nullptr, nullptr, nullptr, nullptr, nullptr)) {
        delete c;
        return 1;
    }

    // :1520-1521
    c->order.resize(M);
    for (uint32_t i = 0; i < M; ++i) c->order[i] = (int32_t)i;

    // :1564-1579 center_and_scale (:371-388) on the full phenotype vector, sequential order
    std::vector<double> y(y_host, y_host + n_global);
    {
        double mean = 0.0;
        for (uint32_t i = 0; i < n_global; ++i) mean += y[i];
        mean /= n_global;
        for (uint32_t i = 0; i < n_global; ++i) y[i] -= mean;
        double sqn = 0.0;
        for (uint32_t i = 0; i < n_global; ++i) sqn += y[i] * y[i];
        sqn = sqrt((double)(n_global - 1) / sqn);
        for (uint32_t i = 0; i < n_global; ++i) y[i] *= sqn;
    }
    double se = 0.0;
    for (uint32_t i = 0; i < n_global; ++i) se += y[i] * y[i];
    c->sigmaE = se / (double)n_global * 0.5;
    if (hgibbs_set_residual(dev, y.data() + row_begin)) {
        delete c;
        return 1;
    }

    // :1592-1597
    c->adaV.assign(M, 1);
    for (uint32_t i = 0; i < M; ++i)
        if (c->sigmaG[c->groups[i]] == 0.0) c->adaV[i] = 0;

    std::vector<double> zeros(M, 0.0);
    if (hgibbs_set_beta(dev, zeros.data())) {
        delete c;
        return 1;
    }
    *out = c;
    return 0;
}

int hydra_chain_destroy(hydra_chain_t c)
{
    delete c;
    return 0;
}

int hydra_chain_iterate(hydra_chain_t c)
{
    if (!c) return cfail("hydra_chain_iterate: null chain");
    const int G = c->G, K = c->K;
    const double dN = (double)c->N;
    hg::Mt gen{c->rng.x, c->rng.idx};
    // HGIBBS_TIMING=1: where the host side of an iteration goes (stderr, milliseconds)
    static const bool timing = std::getenv("HGIBBS_TIMING") != nullptr;
    auto tnow = [] { return std::chrono::steady_clock::now(); };
    auto tms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const auto t_0 = tnow();

    // :1675-1686
    if (hgibbs_add_scalar(c->dev, c->mu)) return 1;
    double epssum = 0.0;
    if (hgibbs_reduce_eps(c->dev, &epssum, nullptr)) return 1;
    c->mu = hg::norm_rng(gen, epssum / dN, c->sigmaE / dN);
    if (hgibbs_add_scalar(c->dev, -c->mu)) return 1;

    const auto t_mu = tnow();
    // :1691-1694
    if (c->shuffle) {
        hg::shuffle_libstdcxx6(c->order.data(), c->order.size(), gen);
    }
    std::fill(c->m0.begin(), c->m0.end(), 0);
    const auto t_shuffle = tnow();

    // :1709-2490 on the device; the generator travels with the call
    c->rng.idx = gen.idx;
    if (hgibbs_sweep(c->dev, c->order.data(), c->sigmaE, c->sigmaG.data(), c->estPi.data(), c->adaV.data(), &c->rng,
                     c->cass.data(), &c->last_nnz))
        return 1;
    gen.idx = c->rng.idx;
    const auto t_sweep = tnow();

    // :2495-2578
    std::vector<double> bsq(G, 0.0);
    if (hgibbs_beta_sqnorm(c->dev, bsq.data())) return 1;
    const auto t_bsq = tnow();
    std::vector<double> dirin(K), pi(K);
    for (int g = 0; g < G; ++g) {
        if (c->MtotGrp[g] == 0) continue;
        c->m0[g] = c->MtotGrp[g] - c->cass[g * K];
        int rowsum = 0;
        for (int k = 0; k < K; ++k) rowsum += c->cass[g * K + k];
        if (c->m0[g] == 0 || rowsum == 0) {
            for (uint32_t i = 0; i < c->M; ++i)
                if (c->groups[i] == g) c->adaV[i] = 0;
            c->sigmaG[g] = 0.0;
            continue;
        }
        const double dm0 = (double)c->m0[g];
        c->sigmaG[g] = hg::inv_scaled_chisq_rng(gen, V0G + dm0, (bsq[g] * dm0 + V0G * S02G) / (V0G + dm0));
        for (int k = 0; k < K; ++k) dirin[k] = (double)c->cass[g * K + k] + 1.0;
        hg::dirichlet_rng(gen, dirin.data(), K, pi.data());
        for (int k = 0; k < K; ++k) c->estPi[g * K + k] = pi[k];
    }

    // :2646-2681 fixed effects: one conditional normal per covariate, shuffled order
    if (c->C > 0) {
        hg::shuffle_libstdcxx6(c->xI.data(), c->xI.size(), gen);
        const double sigmaF = 1.0; // s02F, src/BayesRRm.h:34 (sigmaF = s02F, :2680)
        const double sigE_sigF = c->sigmaE / sigmaF;
        const double dNm1 = (double)(c->N - 1);
        for (int i = 0; i < c->C; ++i) {
            const int col = (int)c->xI[i];
            const double gamma_old = c->gamma[col];
            double num_f = 0.0;
            if (hgibbs_cov_dot(c->dev, col, gamma_old, &num_f)) return 1;
            const double denom_f = dNm1 + sigE_sigF;
            c->gamma[col] = hg::norm_rng(gen, num_f / denom_f, c->sigmaE / denom_f);
            if (hgibbs_cov_update(c->dev, col, gamma_old - c->gamma[col])) return 1;
        }
    }

    // :2685-2690
    double e_sqn = 0.0;
    if (hgibbs_reduce_eps(c->dev, nullptr, &e_sqn)) return 1;
    c->sigmaE = hg::inv_scaled_chisq_rng(gen, V0E + dN, (e_sqn + V0E * S02E) / (V0E + dN));
    c->rng.idx = gen.idx;
    c->iteration += 1;
    if (timing)
        std::fprintf(stderr, "[hydra_chain] iteration %u: mu %.3f  shuffle %.3f  sweep call %.3f  beta sqnorm %.3f  hyper-parameters + sigmaE %.3f  | %.3f ms\n", c->iteration - 1,
                     tms(t_0, t_mu), tms(t_mu, t_shuffle), tms(t_shuffle, t_sweep), tms(t_sweep, t_bsq), tms(t_bsq, tnow()), tms(t_0, tnow()));
    return 0;
}

int hydra_chain_state(hydra_chain_t c, double* sigmaE, double* mu, double* sigmaG, double* estPi, int32_t* m0, int32_t* cass,
                      hgibbs_rng_state* rng)
{
    if (!c) return cfail("hydra_chain_state: null chain");
    if (sigmaE) *sigmaE = c->sigmaE;
    if (mu) *mu = c->mu;
    if (sigmaG) std::copy(c->sigmaG.begin(), c->sigmaG.end(), sigmaG);
    if (estPi) std::copy(c->estPi.begin(), c->estPi.end(), estPi);
    if (m0) std::copy(c->m0.begin(), c->m0.end(), m0);
    if (cass) std::copy(c->cass.begin(), c->cass.end(), cass);
    if (rng) *rng = c->rng;
    return 0;
}

int hydra_chain_set_covariates(hydra_chain_t c, const double* X_host, int C)
{
    if (!c) return cfail("hydra_chain_set_covariates: null chain");
    if (C < 0 || (C > 0 && !X_host)) return cfail("hydra_chain_set_covariates: bad argument");
    if (c->iteration != 0) return cfail("hydra_chain_set_covariates: call before the first iteration");
    c->C = C;
    c->gamma.assign(C, 0.0);
    c->xI.resize(C);
    for (int i = 0; i < C; ++i) c->xI[i] = (unsigned)i;
    return hgibbs_set_covariates(c->dev, C ? X_host + (size_t)c->row_begin * C : nullptr, C);
}

int hydra_chain_restore(hydra_chain_t c, const hydra_restart_state* st)
{
    if (!c || !st || !st->sigmaG || !st->estPi || !st->beta || !st->components || !st->eps || !st->order)
        return cfail("hydra_chain_restore: null argument");
    if (c->C > 0 && (!st->gamma || !st->xI)) return cfail("hydra_chain_restore: covariates set but no gamma/xI given");
    const int G = c->G, K = c->K;
    c->sigmaE = st->sigmaE;
    c->mu = st->mu;
    std::copy(st->sigmaG, st->sigmaG + G, c->sigmaG.begin());
    std::copy(st->estPi, st->estPi + (size_t)G * K, c->estPi.begin());
    for (uint32_t i = 0; i < c->M; ++i) {
        if (st->order[i] < 0 || (uint32_t)st->order[i] >= c->M) return cfail("hydra_chain_restore: marker index out of range");
        c->order[i] = st->order[i];
    }
    for (int i = 0; i < c->C; ++i) {
        c->gamma[i] = st->gamma[i];
        c->xI[i] = (unsigned)st->xI[i];
    }
    c->rng = st->rng;
    if (hgibbs_set_beta(c->dev, st->beta)) return 1;
    if (hgibbs_set_components(c->dev, st->components)) return 1;
    if (hgibbs_set_residual(c->dev, st->eps)) return 1;
    // adaV is rebuilt from sigmaG as after init_from_restart (src/BayesRRm.cpp:1592-1597)
    for (uint32_t i = 0; i < c->M; ++i) c->adaV[i] = (c->sigmaG[c->groups[i]] == 0.0) ? 0 : 1;
    c->iteration = st->iteration + 1;
    return 0;
}

int hydra_rng_to_boost_words(const hgibbs_rng_state* st, uint32_t* words624)
{
    if (!st || !words624) return cfail("hydra_rng_to_boost_words: null argument");
    hg::mt_to_boost_words(st->x, st->idx, words624);
    return 0;
}

int hydra_rng_from_boost_words(const uint32_t* words624, hgibbs_rng_state* st)
{
    if (!st || !words624) return cfail("hydra_rng_from_boost_words: null argument");
    std::memcpy(st->x, words624, 624 * sizeof(uint32_t));
    st->idx = 624; // mt.i = mt.state_size
    return 0;
}

int hydra_chain_gamma(hydra_chain_t c, double* gamma_out, int32_t* xI_out)
{
    if (!c) return cfail("hydra_chain_gamma: null chain");
    for (int i = 0; i < c->C; ++i) {
        if (gamma_out) gamma_out[i] = c->gamma[i];
        if (xI_out) xI_out[i] = (int32_t)c->xI[i];
    }
    return 0;
}

uint64_t hydra_chain_last_nnz(hydra_chain_t c) { return c ? c->last_nnz : 0; }

/* a11: src/BayesRRm.cpp:2742-2760 */
int hydra_chain_csv_line(hydra_chain_t c, uint32_t iteration, char* buf, size_t len)
{
    if (!c || !buf) return -1;
    const int G = c->G, K = c->K;
    size_t o = 0;
    o += snprintf(buf + o, len - o, "%5d, %4d", (int)iteration, G);
    double sg = 0.0;
    for (int g = 0; g < G; ++g) {
        o += snprintf(buf + o, len - o, ", %20.15f", c->sigmaG[g]);
        sg += c->sigmaG[g];
    }
    int m0sum = 0;
    for (int g = 0; g < G; ++g) m0sum += c->m0[g];
    o += snprintf(buf + o, len - o, ", %20.15f, %20.15f, %7d, %4d, %2d", c->sigmaE, sg / (c->sigmaE + sg), m0sum, G, K);
    for (int g = 0; g < G; ++g)
        for (int k = 0; k < K; ++k) o += snprintf(buf + o, len - o, ", %20.15f", c->estPi[g * K + k]);
    o += snprintf(buf + o, len - o, "\n");
    return (int)o;
}

const int32_t* hydra_chain_order(hydra_chain_t c) { return c ? c->order.data() : nullptr; }

} // extern "C"
