/* oracle/orc_gibbs.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Single-threaded fp64 CPU restatement of the reference's BayesRR single-site
 * Gibbs sampler, `BayesRRm::runMpiGibbs` (src/BayesRRm.cpp:933-2939), for
 * world_size == 1, --sync-rate <= 1, --mpibayes bayesMPI (the bayesFH branches
 * at :1125-1163, :1727-1738, :1942-1957, :2557-2568 are dead code on this path),
 * no covariates, data read from a dense 2-bit PLINK .bed.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * link or call this.  The product (hydra_amd/) never does.
 *
 * The reference cannot be compiled here (Eigen + Boost absent);
 * PARITY UNPINNED for the Boost-dependent arithmetic -- see orc_rng.h for what
 * is pinned (MT19937 KAT, Ziggurat tables vs the ELF, LUT decode vs
 * oracle/_ref built from src/mk_lut.cpp).
 *
 * Floating-point form (SURVEY.md 3.1 "traps"): plain --bfile runs the SPARSE
 * formulas, so that is what is restated:
 *   dot    src/BayesRRm.cpp:316-342   num = mstd*((S1*1.0 + S2*2.0) - mave*(Sall - SM))
 *   update src/BayesRRm.cpp:250-281   eps_i += {-(mave*mstd*D), D*(1-mave)*mstd, D*(2-mave)*mstd, 0}[g_i]
 *          (+ :2022 dEpsSum += deltaEps, :2471 eps = tmpEps + dEpsSum == eps_old + v)
 * Summation order (documented, fixed): every sum runs sequentially in
 * increasing individual index with one accumulator each (S1, S2, SM, Sall),
 * exactly what the reference's loops do with OpenMP disabled.
 *
 * std::shuffle is toolchain dependent (src/BayesRRm.cpp:1688-1692 says so); the
 * reference binary was built against libstdc++ 6.5, whose algorithm is restated
 * (orc_rng.h: orc_shuffle_u32), as the product's host driver does.
 */
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "orc_rng.h"

#if defined(__AVX2__)
#include <immintrin.h>
#endif

#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

/* a1: src/data.cpp:1189-1200 -- code v=(byte>>2i)&3; v==1 missing, else 2-(b0+b1) */
inline int decode_code(unsigned v)
{
    if (v == 1u) return -1;
    return 2 - (int)((v & 1u) + ((v >> 1) & 1u));
}

inline int genotype_at(const uint8_t* col, uint32_t i)
{
    return decode_code((col[i >> 2] >> (2 * (i & 3u))) & 3u);
}

int g_threads = 1; /* >1 only for the cpu_baseline leg of bench.py */
int g_dot_form = 0; /* 0 = sparse form (plain --bfile, :316-342), 1 = dense LUT form (:1766-1809),
                       2 = dense LUT form with the reference's loop structure (LUT gathers, AVX2, OpenMP
                           reduction, and its bookkeeping passes): the "restated hydra AVX path" timed
                           by bench.py's cpu_baseline leg (fallback 2) */
double g_lut_a[1024], g_lut_b[1024];
bool g_lut_ready = false;
std::vector<double> g_tmpEps, g_deltaEps, g_dEpsSum;

} // namespace

extern "C" {

void orc_set_threads(int n) { g_threads = n < 1 ? 1 : n; }
/* which of the reference's two algebraically equal dot forms the sweep uses */
void orc_set_dot_form(int form) { g_dot_form = (form < 0 || form > 2) ? 0 : form; }
double orc_dot_dense(const uint8_t* col, const double* eps, uint32_t N, double mave, double mstd, double* s1_out, double* s2_out);

/* (byte, slot) -> (genotype value, non-missing mask): the two quantities
 * dotp_lut_a / dotp_lut_b tabulate (src/dotp_lut.h:3,1033; src/mk_lut.cpp:24-35,54-65) */
void orc_decode_byte(uint8_t byte, double* val4, double* mask4)
{
    for (int s = 0; s < 4; ++s) {
        int g = decode_code((byte >> (2 * s)) & 3u);
        val4[s] = g < 0 ? 0.0 : (double)g;
        mask4[s] = g < 0 ? 0.0 : 1.0;
    }
}

/* src/data.cpp:1284-1286 counts per marker */
void orc_bed_counts(const uint8_t* col, uint32_t N, uint64_t* n0, uint64_t* n1, uint64_t* n2, uint64_t* nm)
{
    uint64_t c[4] = {0, 0, 0, 0};
    for (uint32_t i = 0; i < N; ++i) {
        int g = genotype_at(col, i);
        c[g < 0 ? 3 : g] += 1;
    }
    *n0 = c[0]; *n1 = c[1]; *n2 = c[2]; *nm = c[3];
}

/* a2: src/BayesRRm.cpp:1502-1507 */
void orc_marker_stats(uint64_t n1, uint64_t n2, uint64_t nm, uint32_t N, double* mave, double* mstd)
{
    double dN = (double)N;
    double av = ((double)n1 + 2.0 * (double)n2) / (dN - (double)nm);
    double tmp1 = (double)n1 * (1.0 - av) * (1.0 - av);
    double tmp2 = (double)n2 * (2.0 - av) * (2.0 - av);
    double tmp0 = (double)(N - n1 - n2 - nm) * (0.0 - av) * (0.0 - av);
    *mave = av;
    *mstd = sqrt((double)(N - 1) / (tmp0 + tmp1 + tmp2));
}

/* a4 (sparse form): src/BayesRRm.cpp:316-342.  Returns num BEFORE `+= beta*(N-1)`. */
double orc_dot(const uint8_t* col, const double* eps, uint32_t N, double mave, double mstd)
{
    double s1 = 0.0, s2 = 0.0, sm = 0.0, sall = 0.0;
    if (g_threads <= 1) {
        for (uint32_t i = 0; i < N; ++i) {
            int g = genotype_at(col, i);
            double e = eps[i];
            sall += e;
            if (g == 1) s1 += e;
            else if (g == 2) s2 += e;
            else if (g < 0) sm += e;
        }
    } else {
#ifdef _OPENMP
#pragma omp parallel for reduction(+ : s1, s2, sm, sall) num_threads(g_threads) schedule(static)
#endif
        for (int64_t b = 0; b < (int64_t)((N + 3) / 4); ++b) {
            unsigned byte = col[b];
            for (int s = 0; s < 4; ++s) {
                uint32_t i = (uint32_t)b * 4u + (uint32_t)s;
                if (i >= N) break;
                int g = decode_code((byte >> (2 * s)) & 3u);
                double e = eps[i];
                sall += e;
                if (g == 1) s1 += e;
                else if (g == 2) s2 += e;
                else if (g < 0) sm += e;
            }
        }
    }
    double dp = 0.0;
    dp += s1 * 1.0;
    dp += s2 * 2.0;
    double syt = sall;
    syt -= sm;
    dp -= (mave * syt);
    dp *= mstd;
    return dp;
}

/* dense-LUT form of the same number (src/BayesRRm.cpp:1766-1809): s1 = sum c1*(c2*eps),
 * s2 = sum c2*eps, num = mstd*(s1 - mave*s2).  Kept as a cross-check. */
double orc_dot_dense(const uint8_t* col, const double* eps, uint32_t N, double mave, double mstd,
                     double* s1_out, double* s2_out)
{
    double s1 = 0.0, s2 = 0.0;
    for (uint32_t i = 0; i < N; ++i) {
        int g = genotype_at(col, i);
        double c1 = g < 0 ? 0.0 : (double)g;
        double c2 = g < 0 ? 0.0 : 1.0;
        s1 += c1 * (c2 * eps[i]);
        s2 += (c2 * eps[i]);
    }
    if (s1_out) *s1_out = s1;
    if (s2_out) *s2_out = s2;
    return mstd * (s1 - mave * s2);
}

/* a8: src/BayesRRm.cpp:250-281 + :2022 + :2471.  dbeta = beta_old - beta_new. */
void orc_update(const uint8_t* col, double* eps, uint32_t N, double mave, double mstd, double dbeta)
{
    if (dbeta == 0.0) return;
    double v0 = -(mave * mstd * dbeta);
    double v1 = dbeta * (1.0 - mave) * mstd;
    double v2 = dbeta * (2.0 - mave) * mstd;
#ifdef _OPENMP
#pragma omp parallel for num_threads(g_threads) schedule(static) if (g_threads > 1)
#endif
    for (int64_t b = 0; b < (int64_t)((N + 3) / 4); ++b) {
        unsigned byte = col[b];
        for (int s = 0; s < 4; ++s) {
            uint32_t i = (uint32_t)b * 4u + (uint32_t)s;
            if (i >= N) break;
            int g = decode_code((byte >> (2 * s)) & 3u);
            double v = g < 0 ? 0.0 : (g == 0 ? v0 : (g == 1 ? v1 : v2));
            eps[i] = eps[i] + (0.0 + v);
        }
    }
}


/* ---- "restated hydra AVX path" (cpu_baseline only) -------------------------
 * dot: src/BayesRRm.cpp:1770-1809 -- per byte two 4-wide LUT gathers, eps load,
 * mul, add, mul, add, OpenMP reduction over bytes, scalar tail for N%4.
 * update: :1976-2010 (deltaEps from the LUT), :2022 (dEpsSum += deltaEps),
 * :2471 (eps = tmpEps + dEpsSum), :2478 (tmpEps = eps), :2481 (dEpsSum = 0). */
static void lut_init()
{
    if (g_lut_ready) return;
    for (int byte = 0; byte < 256; ++byte)
        for (int s = 0; s < 4; ++s) {
            int g = decode_code((byte >> (2 * s)) & 3u);
            g_lut_a[4 * byte + s] = g < 0 ? 0.0 : (double)g;
            g_lut_b[4 * byte + s] = g < 0 ? 0.0 : 1.0;
        }
    g_lut_ready = true;
}

double orc_dot_hydra_avx(const uint8_t* col, const double* eps, uint32_t N, double mave, double mstd)
{
    lut_init();
    const int fullb = (int)(N / 4);
    double s1 = 0.0, s2 = 0.0;
#if defined(__AVX2__)
    double r1[4] = {0, 0, 0, 0}, r2[4] = {0, 0, 0, 0};
#pragma omp parallel num_threads(g_threads)
    {
        __m256d vsum1 = _mm256_setzero_pd(), vsum2 = _mm256_setzero_pd();
#pragma omp for schedule(static) nowait
        for (int ii = 0; ii < fullb; ++ii) {
            __m256d p4c1 = _mm256_loadu_pd(&g_lut_a[col[ii] * 4]);
            __m256d p4c2 = _mm256_loadu_pd(&g_lut_b[col[ii] * 4]);
            __m256d p4eps = _mm256_loadu_pd(&eps[ii * 4]);
            __m256d p4sum = _mm256_mul_pd(p4c2, p4eps);
            vsum2 = _mm256_add_pd(vsum2, p4sum);
            p4sum = _mm256_mul_pd(p4sum, p4c1);
            vsum1 = _mm256_add_pd(vsum1, p4sum);
        }
        double t1[4], t2[4];
        _mm256_storeu_pd(t1, vsum1);
        _mm256_storeu_pd(t2, vsum2);
#pragma omp critical
        for (int k = 0; k < 4; ++k) {
            r1[k] += t1[k];
            r2[k] += t2[k];
        }
    }
    s1 = r1[0] + r1[1] + r1[2] + r1[3];
    s2 = r2[0] + r2[1] + r2[2] + r2[3];
#else
#pragma omp parallel for reduction(+ : s1, s2) num_threads(g_threads) schedule(static)
    for (int ii = 0; ii < fullb; ++ii)
        for (int k = 0; k < 4; ++k) {
            double c1 = g_lut_a[col[ii] * 4 + k], c2 = g_lut_b[col[ii] * 4 + k];
            s1 += c1 * (c2 * eps[ii * 4 + k]);
            s2 += c2 * eps[ii * 4 + k];
        }
#endif
    for (uint32_t i = (uint32_t)fullb * 4u; i < N; ++i) {
        int idx = col[fullb] * 4 + (int)(i - (uint32_t)fullb * 4u);
        s1 += g_lut_a[idx] * (g_lut_b[idx] * eps[i]);
        s2 += g_lut_b[idx] * eps[i];
    }
    return mstd * (s1 - mave * s2);
}

void orc_update_hydra(const uint8_t* col, double* eps, uint32_t N, double mave, double mstd, double dbeta)
{
    lut_init();
    if (g_tmpEps.size() != N) {
        g_tmpEps.assign(eps, eps + N);
        g_deltaEps.assign(N, 0.0);
        g_dEpsSum.assign(N, 0.0);
    }
    double* tmpEps = g_tmpEps.data();
    double* deltaEps = g_deltaEps.data();
    double* dEpsSum = g_dEpsSum.data();
    const double sigdb = mstd * dbeta;
    const int nb = (int)((N + 3) / 4);
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int ii = 0; ii < nb; ++ii)
        for (int k = 0; k < 4; ++k) {
            uint32_t i = (uint32_t)ii * 4u + (uint32_t)k;
            if (i < N) deltaEps[i] = (g_lut_a[col[ii] * 4 + k] - mave) * g_lut_b[col[ii] * 4 + k] * sigdb;
        }
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int64_t i = 0; i < (int64_t)N; ++i) dEpsSum[i] += deltaEps[i];
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int64_t i = 0; i < (int64_t)N; ++i) eps[i] = tmpEps[i] + dEpsSum[i];
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int64_t i = 0; i < (int64_t)N; ++i) tmpEps[i] = eps[i];
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int64_t i = 0; i < (int64_t)N; ++i) dEpsSum[i] = 0.0;
}

/* a3: src/BayesRRm.cpp:371-388 */
void orc_center_and_scale(double* vec, uint32_t N)
{
    double mean = 0.0;
    for (uint32_t i = 0; i < N; ++i) mean += vec[i];
    mean /= N;
    for (uint32_t i = 0; i < N; ++i) vec[i] -= mean;
    double sqn = 0.0;
    for (uint32_t i = 0; i < N; ++i) sqn += vec[i] * vec[i];
    sqn = sqrt((double)(N - 1) / sqn);
    for (uint32_t i = 0; i < N; ++i) vec[i] *= sqn;
}

/* NA-phenotype row removal (src/data.cpp:1112-1158 does it on the sparse index
 * lists; on a dense column it is a repack of the kept 2-bit fields). */
void orc_bed_compact(const uint8_t* col_in, uint32_t N_in, const uint8_t* keep, uint8_t* col_out, uint32_t* N_out)
{
    uint32_t o = 0;
    for (uint32_t i = 0; i < N_in; ++i) {
        if (!keep[i]) continue;
        unsigned v = (col_in[i >> 2] >> (2 * (i & 3u))) & 3u;
        if ((o & 3u) == 0) col_out[o >> 2] = 0;
        col_out[o >> 2] |= (uint8_t)(v << (2 * (o & 3u)));
        ++o;
    }
    /* pad the tail of the last byte with the "missing" code so padding never contributes */
    for (uint32_t p = o; (p & 3u) != 0; ++p) col_out[p >> 2] |= (uint8_t)(1u << (2 * (p & 3u)));
    *N_out = o;
}

/* ------------------------------------------------------------------ */
/* a5-a8: one full sweep over `order` (src/BayesRRm.cpp:1709-2025 + :2468-2487) */
/* ------------------------------------------------------------------ */
struct orc_model {
    int G, K;
    const int* groups;      /* M */
    const double* cVa;      /* G*K, col 0 = 0 */
    const double* cVaI;     /* G*K, col 0 = 0 */
};

/* a5-a7 for ONE marker given its dot product `num` (x_j'eps, before the
 * `+= beta*(N-1)` of :1855): src/BayesRRm.cpp:1721-1723,1744-1753,1855-1921.
 * Returns 0, or -1 on the reference's "logL overflow" abort (:1910-1913). */
int orc_marker_draw(double num, double beta_old, uint32_t N, int K, const double* cVa_g, const double* cVaI_g,
                    const double* estPi_g, double sigmaE, double sigmaG_g, orc_mt* rng,
                    double* beta_new, int* component, double* acum_out)
{
    const double dNm1 = (double)(N - 1);
    const int km1 = K - 1;
    double denom[16], muk[16], logL[16];
    const double sigE_G = sigmaE / sigmaG_g;
    const double sigG_E = sigmaG_g / sigmaE;
    const double i_2sigE = 1.0 / (2.0 * sigmaE);
    muk[0] = 0.0;
    for (int i = 1; i <= km1; ++i) denom[i - 1] = dNm1 + sigE_G * cVaI_g[i];
    num += beta_old * (double)(N - 1);
    for (int i = 1; i <= km1; ++i) muk[i] = num / denom[i - 1];
    for (int i = 0; i < K; ++i) logL[i] = log(estPi_g[i]);
    for (int i = 1; i < 1 + km1; ++i)
        logL[i] = logL[i] - 0.5 * log(sigG_E * dNm1 * cVa_g[i] + 1.0) + muk[i] * num * i_2sigE;

    double prob = orc_unif_rng(rng);

    double acum = 0.0;
    bool big = false;
    for (int i = 1; i < K; ++i)
        if (fabs(logL[i] - logL[0]) > 700.0) big = true;
    if (big) {
        acum = 0.0;
    } else {
        double s = 0.0;
        for (int i = 0; i < K; ++i) s += exp(logL[i] - logL[0]);
        acum = 1.0 / s;
    }
    *acum_out = acum;

    for (int k = 0; k < K; ++k) {
        if (prob <= acum || k == km1) {
            if (k == 0) *beta_new = 0.0;
            else *beta_new = orc_norm_rng(rng, muk[k], sigmaE / denom[k - 1]);
            *component = k;
            return 0;
        } else {
            if (k + 1 >= K) return -1;
            bool big2 = false;
            for (int l = k + 1; l < K; ++l)
                if (fabs(logL[l] - logL[k + 1]) > 700.0) big2 = true;
            if (big2) {
                acum += 0.0;
            } else {
                double s = 0.0;
                for (int l = 0; l < K; ++l) s += exp(logL[l] - logL[k + 1]);
                acum += 1.0 / s;
            }
        }
    }
    return -1;
}

/* Returns the number of markers with deltaBeta != 0.  cass is G*K, zeroed by the caller
 * (src/BayesRRm.cpp:1697).  Returns -1 on the reference's "logL overflow" abort (:1910-1913). */
long orc_sweep(const uint8_t* bed, uint64_t stride, uint32_t N, uint32_t M,
               const double* mave, const double* mstd,
               int G, int K, const int* groups, const double* cVa, const double* cVaI,
               const int* order, double sigmaE, const double* sigmaG, const double* estPi,
               const uint8_t* adaV,
               double* eps, double* beta, int* components, double* acum_out, int* cass,
               orc_mt* rng)
{
    long nnz = 0;
    if (g_dot_form == 2) { // :1700 tmpEps = epsilon, dEpsSum = 0 (:1544)
        g_tmpEps.assign(eps, eps + N);
        g_deltaEps.assign(N, 0.0);
        g_dEpsSum.assign(N, 0.0);
    }

    for (uint32_t j = 0; j < M; ++j) {
        const int marker = order[j];
        const int grp = groups[marker];
        const uint8_t* col = bed + (uint64_t)marker * stride;
        double b = beta[marker];

        if (adaV[marker]) {
            double num = g_dot_form == 2   ? orc_dot_hydra_avx(col, eps, N, mave[marker], mstd[marker])
                         : g_dot_form == 1 ? orc_dot_dense(col, eps, N, mave[marker], mstd[marker], nullptr, nullptr)
                                           : orc_dot(col, eps, N, mave[marker], mstd[marker]);
            int k = 0;
            if (orc_marker_draw(num, b, N, K, cVa + (size_t)grp * K, cVaI + (size_t)grp * K, estPi + (size_t)grp * K, sigmaE,
                                sigmaG[grp], rng, &beta[marker], &k, &acum_out[marker]))
                return -1;
            cass[grp * K + k] += 1;
            components[marker] = k;
        } else {
            beta[marker] = 0.0;
            acum_out[marker] = 1.0;
        }

        const double betaOld = b;
        b = beta[marker];
        const double deltaBeta = betaOld - b;
        if (deltaBeta != 0.0) {
            if (g_dot_form == 2) orc_update_hydra(col, eps, N, mave[marker], mstd[marker], deltaBeta);
            else orc_update(col, eps, N, mave[marker], mstd[marker], deltaBeta);
            ++nnz;
        }
    }
    return nnz;
}

/* ------------------------------------------------------------------ */
/* whole-chain driver                                                  */
/* ------------------------------------------------------------------ */
struct orc_chain {
    uint32_t N, M;
    uint64_t stride;
    const uint8_t* bed;
    int G, K;
    std::vector<int> groups, MtotGrp, order, components, cass, m0;
    std::vector<double> cVa, cVaI, priorPi, estPi, sigmaG;
    std::vector<double> mave, mstd, y, eps, beta, acum;
    std::vector<uint8_t> adaV;
    double sigmaE, mu;
    int shuffle;
    long last_nnz;
    orc_mt rng;
    uint32_t iteration;
    /* fixed effects (src/BayesRRm.cpp:1113-1114,1552-1553,2646-2681) */
    int C;
    std::vector<double> X; /* N x C row-major */
    std::vector<double> gamma;
    std::vector<unsigned int> xI;
};

static const double V0E = 0.0001, S02E = 0.0001, V0G = 0.0001, S02G = 0.0001; /* src/BayesRRm.h:30-33 */

/* Init = src/BayesRRm.cpp:967-1110 (model), :1228-1240 (seed, sigmaG), :1492-1508 (stats),
 * :1564-1597 (y, eps, sigmaE, adaV), :1520-1521 (markerI).  y_raw has NA rows already removed;
 * mS is G x K with column 0 == 0.0. */
orc_chain* orc_chain_create(const uint8_t* bed, uint64_t stride, uint32_t N, uint32_t M,
                            const double* y_raw, int G, int K, const int* groups, const double* mS,
                            uint32_t seed, int shuffle)
{
    orc_chain* c = new orc_chain();
    c->N = N; c->M = M; c->stride = stride; c->bed = bed; c->G = G; c->K = K;
    c->shuffle = shuffle; c->iteration = 0; c->last_nnz = 0; c->C = 0;
    c->groups.assign(groups, groups + M);
    const int km1 = K - 1;

    c->cVa.assign((size_t)G * K, 0.0);
    c->cVaI.assign((size_t)G * K, 0.0);
    c->priorPi.assign((size_t)G * K, 0.0);
    for (int g = 0; g < G; ++g) {
        c->priorPi[g * K + 0] = 0.5;
        double s = 0.0;
        for (int k = 1; k <= km1; ++k) {
            c->cVa[g * K + k] = mS[g * K + k];
            c->cVaI[g * K + k] = 1.0 / c->cVa[g * K + k];
            s += c->cVa[g * K + k];
        }
        for (int k = 1; k <= km1; ++k)
            c->priorPi[g * K + k] = c->priorPi[g * K + 0] * c->cVa[g * K + k] / s;
    }
    c->estPi = c->priorPi;
    c->mu = 0.0;
    c->beta.assign(M, 0.0);
    c->components.assign(M, 0);
    c->acum.assign(M, 0.0);
    c->sigmaG.assign(G, 0.0);
    c->sigmaE = 0.0;
    c->cass.assign((size_t)G * K, 0);
    c->m0.assign(G, 0);

    c->MtotGrp.assign(G, 0);
    for (uint32_t i = 0; i < M; ++i) c->MtotGrp[groups[i]] += 1;

    orc_mt_seed(&c->rng, seed);
    for (int g = 0; g < G; ++g) c->sigmaG[g] = orc_beta_rng(&c->rng, 1.0, 1.0);
    for (int g = 0; g < G; ++g)
        if (c->MtotGrp[g] == 0) c->sigmaG[g] = 0.0;

    c->mave.resize(M); c->mstd.resize(M);
    for (uint32_t i = 0; i < M; ++i) {
        uint64_t n0, n1, n2, nm;
        orc_bed_counts(bed + (uint64_t)i * stride, N, &n0, &n1, &n2, &nm);
        orc_marker_stats(n1, n2, nm, N, &c->mave[i], &c->mstd[i]);
    }

    c->order.resize(M);
    for (uint32_t i = 0; i < M; ++i) c->order[i] = (int)i;

    c->y.assign(y_raw, y_raw + N);
    orc_center_and_scale(c->y.data(), N);
    c->eps = c->y;
    double se = 0.0;
    for (uint32_t i = 0; i < N; ++i) se += c->eps[i] * c->eps[i];
    c->sigmaE = se / (double)N * 0.5;

    c->adaV.assign(M, 1);
    for (uint32_t i = 0; i < M; ++i)
        if (c->sigmaG[groups[i]] == 0.0) c->adaV[i] = 0;
    return c;
}

void orc_chain_destroy(orc_chain* c) { delete c; }

/* src/BayesRRm.cpp:1675-1697: mu update, shuffle, reset counters */
void orc_chain_iter_begin(orc_chain* c)
{
    const uint32_t N = c->N;
    const double dN = (double)N;
    for (uint32_t i = 0; i < N; ++i) c->eps[i] += c->mu;
    double epssum = 0.0;
    for (uint32_t i = 0; i < N; ++i) epssum += c->eps[i];
    c->mu = orc_norm_rng(&c->rng, epssum / dN, c->sigmaE / dN);
    for (uint32_t i = 0; i < N; ++i) c->eps[i] -= c->mu;
    if (c->shuffle) {
        orc_shuffle_u32(&c->rng, reinterpret_cast<uint32_t*>(c->order.data()), c->order.size());
    }
    std::fill(c->m0.begin(), c->m0.end(), 0);
    std::fill(c->cass.begin(), c->cass.end(), 0);
}

long orc_chain_sweep(orc_chain* c)
{
    c->last_nnz = orc_sweep(c->bed, c->stride, c->N, c->M, c->mave.data(), c->mstd.data(), c->G, c->K,
                            c->groups.data(), c->cVa.data(), c->cVaI.data(), c->order.data(), c->sigmaE,
                            c->sigmaG.data(), c->estPi.data(), c->adaV.data(), c->eps.data(), c->beta.data(),
                            c->components.data(), c->acum.data(), c->cass.data(), &c->rng);
    return c->last_nnz;
}

/* src/BayesRRm.cpp:2495-2578 (groups), :2685-2690 (sigmaE) */
void orc_chain_iter_end(orc_chain* c)
{
    const int G = c->G, K = c->K;
    const uint32_t N = c->N, M = c->M;
    const double dN = (double)N;
    std::vector<double> bsq(G, 0.0);
    for (uint32_t i = 0; i < M; ++i) bsq[c->groups[i]] += c->beta[i] * c->beta[i];

    std::vector<double> dirin(K), pi(K);
    for (int g = 0; g < G; ++g) {
        if (c->MtotGrp[g] == 0) continue;
        c->m0[g] = c->MtotGrp[g] - c->cass[g * K + 0];
        int rowsum = 0;
        for (int k = 0; k < K; ++k) rowsum += c->cass[g * K + k];
        if (c->m0[g] == 0 || rowsum == 0) {
            for (uint32_t i = 0; i < M; ++i)
                if (c->groups[i] == g) c->adaV[i] = 0;
            c->sigmaG[g] = 0.0;
            continue;
        }
        double dm0 = (double)c->m0[g];
        c->sigmaG[g] = orc_inv_scaled_chisq_rng(&c->rng, V0G + dm0, (bsq[g] * dm0 + V0G * S02G) / (V0G + dm0));
        for (int k = 0; k < K; ++k) dirin[k] = (double)c->cass[g * K + k] + 1.0;
        orc_dirichlet_rng(&c->rng, dirin.data(), K, pi.data());
        for (int k = 0; k < K; ++k) c->estPi[g * K + k] = pi[k];
    }

    if (c->C > 0) { /* :2646-2681 */
        orc_shuffle_u32(&c->rng, c->xI.data(), c->xI.size());
        const double sigmaF = 1.0;
        const double sigE_sigF = c->sigmaE / sigmaF;
        const double dNm1 = (double)(N - 1);
        for (int i = 0; i < c->C; ++i) {
            const unsigned col = c->xI[i];
            const double gamma_old = c->gamma[col];
            double num_f = 0.0;
            for (uint32_t k = 0; k < N; ++k) num_f += c->X[(size_t)k * c->C + col] * (c->eps[k] + gamma_old * c->X[(size_t)k * c->C + col]);
            const double denom_f = dNm1 + sigE_sigF;
            c->gamma[col] = orc_norm_rng(&c->rng, num_f / denom_f, c->sigmaE / denom_f);
            for (uint32_t k = 0; k < N; ++k) c->eps[k] = c->eps[k] + (gamma_old - c->gamma[col]) * c->X[(size_t)k * c->C + col];
        }
    }

    double e_sqn = 0.0;
    for (uint32_t i = 0; i < N; ++i) e_sqn += c->eps[i] * c->eps[i];
    c->sigmaE = orc_inv_scaled_chisq_rng(&c->rng, V0E + dN, (e_sqn + V0E * S02E) / (V0E + dN));
    c->iteration += 1;
}

void orc_chain_set_covariates(orc_chain* c, const double* X, int C)
{
    c->C = C;
    c->X.assign(X, X + (size_t)c->N * C);
    c->gamma.assign(C, 0.0);
    c->xI.resize(C);
    for (int i = 0; i < C; ++i) c->xI[i] = (unsigned)i;
}
double* orc_chain_gamma(orc_chain* c) { return c->gamma.data(); }
unsigned int* orc_chain_xI(orc_chain* c) { return c->xI.data(); }

/* Restart: what runMpiGibbs does with the *_restart members after init_from_restart
 * (src/BayesRRm.cpp:842-928 reads them; :1546-1573 installs gamma/xI/epsilon/markerI/mu;
 * :1592-1597 rebuilds adaV; :924 iteration_start = it + 1).  Beta and components were read
 * straight into the live vectors (:881-896). */
void orc_chain_restore(orc_chain* c, uint32_t iteration, double sigmaE, double mu, const double* sigmaG,
                       const double* estPi, const double* beta, const int* components, const double* eps,
                       const int* order, const double* gamma, const int* xI, const uint32_t* rng_words)
{
    c->sigmaE = sigmaE;
    c->mu = mu;
    for (int g = 0; g < c->G; ++g) c->sigmaG[g] = sigmaG[g];
    for (int i = 0; i < c->G * c->K; ++i) c->estPi[i] = estPi[i];
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
    for (uint32_t i = 0; i < c->M; ++i) c->adaV[i] = (c->sigmaG[c->groups[i]] == 0.0) ? 0 : 1;
    orc_mt_load_words(&c->rng, rng_words);
    c->iteration = iteration + 1;
}

void orc_chain_iterate(orc_chain* c)
{
    orc_chain_iter_begin(c);
    orc_chain_sweep(c);
    orc_chain_iter_end(c);
}

/* accessors (pointers stay valid for the life of the chain) */
double* orc_chain_beta(orc_chain* c) { return c->beta.data(); }
int* orc_chain_components(orc_chain* c) { return c->components.data(); }
double* orc_chain_acum(orc_chain* c) { return c->acum.data(); }
double* orc_chain_eps(orc_chain* c) { return c->eps.data(); }
double* orc_chain_y(orc_chain* c) { return c->y.data(); }
double* orc_chain_sigmaG(orc_chain* c) { return c->sigmaG.data(); }
double* orc_chain_estPi(orc_chain* c) { return c->estPi.data(); }
double* orc_chain_mave(orc_chain* c) { return c->mave.data(); }
double* orc_chain_mstd(orc_chain* c) { return c->mstd.data(); }
double* orc_chain_cVa(orc_chain* c) { return c->cVa.data(); }
double* orc_chain_cVaI(orc_chain* c) { return c->cVaI.data(); }
int* orc_chain_order(orc_chain* c) { return c->order.data(); }
int* orc_chain_cass(orc_chain* c) { return c->cass.data(); }
int* orc_chain_m0(orc_chain* c) { return c->m0.data(); }
uint8_t* orc_chain_adaV(orc_chain* c) { return c->adaV.data(); }
double orc_chain_sigmaE(orc_chain* c) { return c->sigmaE; }
double orc_chain_mu(orc_chain* c) { return c->mu; }
long orc_chain_last_nnz(orc_chain* c) { return c->last_nnz; }
orc_mt* orc_chain_rng(orc_chain* c) { return &c->rng; }

/* a11: the .csv line of src/BayesRRm.cpp:2742-2760 */
int orc_chain_csv_line(orc_chain* c, uint32_t iteration, char* buf, size_t len)
{
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

/* ---- RNG known-answer entry points for tests ---- */
void orc_rng_seed(orc_mt* g, uint32_t seed) { orc_mt_seed(g, seed); }
void orc_rng_print_words(const orc_mt* g, uint32_t* out624) { orc_mt_print_words(g, out624); }
void orc_rng_load_words(orc_mt* g, const uint32_t* in624) { orc_mt_load_words(g, in624); }
uint32_t orc_rng_u32(orc_mt* g) { return orc_mt_next(g); }
double orc_rng_unif(orc_mt* g) { return orc_unif_rng(g); }
double orc_rng_norm(orc_mt* g, double mean, double var) { return orc_norm_rng(g, mean, var); }
double orc_rng_exp(orc_mt* g, double lambda) { return orc_exponential(g, lambda); }
double orc_rng_gamma(orc_mt* g, double shape, double scale) { return orc_rgamma(g, shape, scale); }
double orc_rng_beta(orc_mt* g, double a, double b) { return orc_beta_rng(g, a, b); }
double orc_rng_inv_scaled_chisq(orc_mt* g, double dof, double scale) { return orc_inv_scaled_chisq_rng(g, dof, scale); }
void orc_rng_dirichlet(orc_mt* g, const double* alpha, int len, double* out) { orc_dirichlet_rng(g, alpha, len, out); }
void orc_rng_shuffle(orc_mt* g, int* v, int n)
{
    orc_shuffle_u32(g, reinterpret_cast<uint32_t*>(v), (size_t)n);
}
const double* orc_zig_table(int which, int* len)
{
    switch (which) {
    case 0: *len = 129; return ORC_ZIG_NORMAL_X;
    case 1: *len = 129; return ORC_ZIG_NORMAL_Y;
    case 2: *len = 257; return ORC_ZIG_EXP_X;
    default: *len = 257; return ORC_ZIG_EXP_Y;
    }
}

} /* extern "C" */
