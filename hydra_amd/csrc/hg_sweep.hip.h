// hg_sweep.hip.h -- the hot kernel (see hg_kernels.h for the design summary).
#pragma once

#include "hg_kernels.h"

namespace hg {

#define HG_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

// Word source over the two MT19937 blocks staged in LDS (untempered words).
struct LdsGen {
    const uint32_t* w;
    uint32_t pos;
    uint32_t limit; // words staged: MT_N, or MT_BUF when the next block was generated
    uint32_t err;
    __device__ __forceinline__ uint32_t next()
    {
        if (pos >= limit) {
            err = 2u;
            return 0u;
        }
        return mt_temper(w[pos++]);
    }
};

struct SweepShared {
    uint32_t mt[MT_BUF];
    double wpart[BLOCK_WAVES][3 * MAX_BATCH + 1];
    double tot[3 * MAX_BATCH + 1];
    double thr[MAX_BATCH][MAX_K];
    double muk[MAX_BATCH][MAX_K];
    uint32_t flag_last;
    uint32_t new_idx;
};

// Next 624 untempered words from the current block, 256 threads, into mt[624..1247].
__device__ __forceinline__ void mt_next_block(uint32_t* mt, int tid)
{
    for (int i = tid; i < 227; i += BLOCK) mt[MT_N + i] = mt_mix(mt[i], mt[i + 1], mt[i + MT_M]);
    __syncthreads();
    for (int i = 227 + tid; i < 454; i += BLOCK) mt[MT_N + i] = mt_mix(mt[i], mt[i + 1], mt[MT_N + i - 227]);
    __syncthreads();
    for (int i = 454 + tid; i < 623; i += BLOCK) mt[MT_N + i] = mt_mix(mt[i], mt[i + 1], mt[MT_N + i - 227]);
    __syncthreads();
    if (tid == 0) mt[MT_N + 623] = mt_mix(mt[623], mt[MT_N], mt[MT_N + 396]);
    __syncthreads();
}

// Posterior + draw + bookkeeping for the nb markers of this batch, given the
// reduced sums in sh.tot: rows [3j..3j+2] = (S1,S2,SM) of batch column j, row
// 3*MAX_BATCH = sum of eps.  Runs in ONE workgroup of 256 threads.
// a5-a7: src/BayesRRm.cpp:1721-1723,1744-1753,1855-1921; sparse dot algebra :325-341.
__device__ __forceinline__ void sweep_draw_phase(const SweepParams& p, const SweepDesc& d, uint32_t nb, SweepShared& sh)
{
    const int tid = threadIdx.x;
    const int K = p.K;

    // stage the generator
    for (int i = tid; i < MT_N; i += BLOCK) sh.mt[i] = p.mt[i];
    __syncthreads();
    const uint32_t idx0 = d.rng_idx;
    const bool need_next = idx0 + 2 * MAX_BATCH + 64 > (uint32_t)MT_N; // uniform
    if (need_next) mt_next_block(sh.mt, tid);

    // ---- per-marker posterior, one thread per batch column -----------------
    int marker = -1, grp = 0;
    bool ada = false;
    double bold = 0.0, mave = 0.0, mstd = 0.0, thr0 = 1.0;
    if ((uint32_t)tid < nb) {
        marker = p.order[d.cursor + tid];
        grp = p.groups[marker];
        ada = p.adaV[marker] != 0;
        bold = p.beta[marker];
        mave = p.mave[marker];
        mstd = p.mstd[marker];
        if (ada) {
            const double S1 = sh.tot[3 * tid], S2 = sh.tot[3 * tid + 1], SM = sh.tot[3 * tid + 2];
            const double Sall = sh.tot[3 * MAX_BATCH];
            double dp = 0.0;
            dp += S1 * 1.0;
            dp += S2 * 2.0;
            double syt = Sall;
            syt -= SM;
            dp -= (mave * syt);
            dp *= mstd;
            double num = dp;
            num += bold * p.n_minus_1;

            double logL[MAX_K];
            const double* den = p.denom + (size_t)grp * K;
            const double* lpi = p.logpi + (size_t)grp * K;
            const double* hlg = p.hlog + (size_t)grp * K;
            logL[0] = lpi[0];
            sh.muk[tid][0] = 0.0;
            for (int k = 1; k < K; ++k) {
                double mk = num / den[k];
                sh.muk[tid][k] = mk;
                logL[k] = lpi[k] - hlg[k] + mk * num * p.i_2sigE;
            }
            // cumulative thresholds of the component walk (:1883-1921)
            double acum;
            bool big = false;
            for (int k = 1; k < K; ++k)
                if (fabs(logL[k] - logL[0]) > 700.0) big = true;
            if (big) {
                acum = 0.0;
            } else {
                double s = 0.0;
                for (int k = 0; k < K; ++k) s += exp(logL[k] - logL[0]);
                acum = 1.0 / s;
            }
            thr0 = acum;
            sh.thr[tid][0] = acum;
            for (int k = 0; k + 2 < K; ++k) {
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
                sh.thr[tid][k + 1] = acum;
            }
        }
    }
    __syncthreads();

    // ---- the walk: wave 0 consumes the stream in marker order ---------------
    if (tid < WAVE) {
        const int lane = tid;
        const bool valid = (uint32_t)lane < nb;
        const unsigned long long am = __ballot(valid && ada);
        const uint32_t jeff = (uint32_t)__popcll(am & ((1ull << lane) - 1ull));
        int k = 0;
        if (valid && ada) {
            const uint32_t u = mt_temper(sh.mt[idx0 + jeff]);
            const double prob = (double)u * (1.0 / 4294967296.0);
            k = K - 1;
            for (int kk = K - 2; kk >= 0; --kk)
                if (prob <= sh.thr[lane][kk]) k = kk; // ends at the FIRST kk that accepts
        }
        const bool event = valid && (ada ? (k != 0 || bold != 0.0) : (bold != 0.0));
        const unsigned long long em = __ballot(event);
        const uint32_t f = em ? (uint32_t)(__ffsll((long long)em) - 1) : nb; // first event
        const uint32_t naccept = (f < nb) ? f + 1 : nb;

        double bnew = 0.0;
        uint32_t consumed = 0, gerr = 0;
        if ((uint32_t)lane == f && ada && k > 0) {
            LdsGen g{sh.mt, idx0 + jeff + 1u, need_next ? (uint32_t)MT_BUF : (uint32_t)MT_N, 0u};
            bnew = norm_rng_sd(g, p.zig, sh.muk[lane][k], p.sdk[(size_t)grp * K + k]);
            consumed = g.pos - (idx0 + jeff + 1u);
            gerr = g.err;
        }
        const double dbeta = bold - bnew;

        // results of accepted markers (:1892,:1899-1905,:1924-1925)
        if ((uint32_t)lane < naccept) {
            if (ada) {
                const int kk = ((uint32_t)lane == f) ? k : 0;
                p.beta[marker] = ((uint32_t)lane == f) ? bnew : 0.0;
                p.comp[marker] = kk;
                p.acum[marker] = thr0;
                atomicAdd(&p.cass[grp * K + kk], 1);
            } else {
                p.beta[marker] = 0.0;
                p.acum[marker] = 1.0;
            }
        }

        // hand the state to the next launch
        const uint32_t used = (uint32_t)__popcll(am & ((naccept >= 64u) ? ~0ull : ((1ull << naccept) - 1ull)));
        const int src = (f < nb) ? (int)f : 0;
        const double f_dbeta = __shfl(dbeta, src, 64);
        const double f_mave = __shfl(mave, src, 64);
        const double f_mstd = __shfl(mstd, src, 64);
        const int f_marker = __shfl(marker, src, 64);
        const uint32_t f_consumed = (uint32_t)__shfl((int)consumed, src, 64);
        const uint32_t f_err = (uint32_t)__shfl((int)gerr, src, 64);
        if (lane == 0) {
            SweepDesc n = d;
            n.cursor = d.cursor + naccept;
            if (d.pend_marker >= 0) n.cur = d.cur ^ 1u;
            n.pend_marker = -1;
            if (f < nb && f_dbeta != 0.0) {
                n.pend_marker = f_marker;
                n.pv[0] = -(f_mave * f_mstd * f_dbeta);
                n.pv[1] = f_dbeta * (1.0 - f_mave) * f_mstd;
                n.pv[2] = f_dbeta * (2.0 - f_mave) * f_mstd;
                n.nnz = d.nnz + 1;
            }
            uint32_t nidx = idx0 + used + ((f < nb) ? f_consumed : 0u);
            sh.new_idx = nidx;
            if (nidx >= (uint32_t)MT_N) nidx -= (uint32_t)MT_N;
            n.rng_idx = nidx;
            n.launches = d.launches + 1;
            n.accepted_sum = d.accepted_sum + naccept;
            if (f < nb && f_err) n.error = f_err;
            *p.desc = n;
        }
    }
    __syncthreads();
    // generator crossed into the next block: make it the current one
    if (sh.new_idx >= (uint32_t)MT_N)
        for (int i = tid; i < MT_N; i += BLOCK) p.mt[i] = sh.mt[MT_N + i];
}

// One launch of the sweep (grid = (n_pad/4096, ceil(MAX_BATCH/cols_per_group))).
__global__ __launch_bounds__(BLOCK) void k_sweep_batch(SweepParams p)
{
    __shared__ SweepShared sh;
    const SweepDesc d = *p.desc;
    const bool pend = d.pend_marker >= 0;
    const uint32_t remaining = (d.cursor < p.M) ? p.M - d.cursor : 0u;
    const uint32_t nb = d.batch < remaining ? d.batch : remaining;
    if ((nb == 0 && !pend) || d.error) return; // whole grid agrees: nothing left to do

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t tile = blockIdx.x * BLOCK_WAVES + wave;
    const uint32_t cpg = p.cols_per_group;
    const uint32_t c0 = blockIdx.y * cpg;
    const uint32_t c1 = (c0 + cpg < nb) ? c0 + cpg : nb;
    const bool first_group = blockIdx.y == 0;
    const double* eps_in = d.cur ? p.eps1 : p.eps0;
    double* eps_out = d.cur ? p.eps0 : p.eps1;

    if (c0 < nb || first_group) {
        double e[IPT];
        load_eps16(eps_in, tile, lane, e);
        if (pend) {
            const uint32_t w =
                *reinterpret_cast<const uint32_t*>(p.bed + (size_t)d.pend_marker * p.stride + ((size_t)tile << 8) + (lane << 2));
            apply_update16(w, d.pv[0], d.pv[1], d.pv[2], e);
            if (first_group) store_eps16(eps_out, tile, lane, e);
        }
        if (first_group) {
            double s = 0.0;
#pragma unroll
            for (int i = 0; i < IPT; ++i) s += e[i];
            s = wave_sum(s);
            if (lane == 0) sh.wpart[wave][3 * MAX_BATCH] = s;
        }
        for (uint32_t j = c0; j < c1; ++j) {
            const int marker = p.order[d.cursor + j];
            const uint32_t w =
                *reinterpret_cast<const uint32_t*>(p.bed + (size_t)marker * p.stride + ((size_t)tile << 8) + (lane << 2));
            uint32_t m1, m2, mm;
            code_masks(w, m1, m2, mm);
            double s1 = 0.0, s2 = 0.0, sm = 0.0;
#pragma unroll
            for (int s = 0; s < IPT; ++s) {
                s1 += mask_f64(e[s], ((int)(m1 << (31 - 2 * s))) >> 31);
                s2 += mask_f64(e[s], ((int)(m2 << (31 - 2 * s))) >> 31);
                sm += mask_f64(e[s], ((int)(mm << (31 - 2 * s))) >> 31);
            }
            s1 = wave_sum(s1);
            s2 = wave_sum(s2);
            sm = wave_sum(sm);
            if (lane == 0) {
                sh.wpart[wave][3 * j] = s1;
                sh.wpart[wave][3 * j + 1] = s2;
                sh.wpart[wave][3 * j + 2] = sm;
            }
        }
    }
    __syncthreads();

    // block partial = waves 0..3 in order, published write-through (sc1)
    {
        const uint32_t nrow = (c1 > c0) ? 3 * (c1 - c0) : 0u;
        if ((uint32_t)tid < nrow) {
            const uint32_t r = 3 * c0 + tid;
            double v = sh.wpart[0][r];
            v += sh.wpart[1][r];
            v += sh.wpart[2][r];
            v += sh.wpart[3][r];
            __hip_atomic_store(p.partials + (size_t)r * p.nblk_x + blockIdx.x, v, HG_RLX_AGENT);
        }
        if (first_group && tid == BLOCK - 1) {
            const uint32_t r = 3 * MAX_BATCH;
            double v = sh.wpart[0][r];
            v += sh.wpart[1][r];
            v += sh.wpart[2][r];
            v += sh.wpart[3][r];
            __hip_atomic_store(p.partials + (size_t)r * p.nblk_x + blockIdx.x, v, HG_RLX_AGENT);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every storing wave drains (eps + partials)
    __syncthreads();
    if (tid == 0) {
        const uint32_t t = __hip_atomic_fetch_add(p.ticket, 1u, HG_RLX_AGENT);
        sh.flag_last = (t == gridDim.x * gridDim.y - 1u) ? 1u : 0u;
    }
    __syncthreads();
    if (!sh.flag_last) return;

    // ---- last-arriving workgroup: fixed-order reduction over blocks ---------
    {
        const uint32_t nrows = 3 * nb + 1;
        for (uint32_t rr = wave; rr < nrows; rr += BLOCK_WAVES) {
            const uint32_t r = (rr == 3 * nb) ? 3 * MAX_BATCH : rr;
            const double* row = p.partials + (size_t)r * p.nblk_x;
            double v = 0.0;
            for (uint32_t b = lane; b < p.nblk_x; b += WAVE) v += __hip_atomic_load(row + b, HG_RLX_AGENT);
            v = wave_sum(v);
            if (lane == 0) sh.tot[r] = v;
        }
    }
    if (tid == 0) __hip_atomic_store(p.ticket, 0u, HG_RLX_AGENT);
    __syncthreads();

    if (p.sums_out) { // multi-GPU: hand the local sums to the all-reduce
        for (int r = tid; r < 3 * MAX_BATCH + 1; r += BLOCK) p.sums_out[r] = (r < 3 * (int)nb || r == 3 * MAX_BATCH) ? sh.tot[r] : 0.0;
        return;
    }
    sweep_draw_phase(p, d, nb, sh);
}

// Multi-GPU second half: sums_out has been all-reduced over ranks.
__global__ __launch_bounds__(BLOCK) void k_sweep_draw(SweepParams p)
{
    __shared__ SweepShared sh;
    const SweepDesc d = *p.desc;
    const bool pend = d.pend_marker >= 0;
    const uint32_t remaining = (d.cursor < p.M) ? p.M - d.cursor : 0u;
    const uint32_t nb = d.batch < remaining ? d.batch : remaining;
    if ((nb == 0 && !pend) || d.error) return;
    for (int r = threadIdx.x; r < 3 * MAX_BATCH + 1; r += BLOCK) sh.tot[r] = p.sums_out[r];
    __syncthreads();
    sweep_draw_phase(p, d, nb, sh);
}

} // namespace hg
