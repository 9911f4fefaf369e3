// hg_sweep.hip.h -- the hot kernel (see hg_kernels.h for the design summary).
#pragma once

#include <utility>

#include "hg_kernels.h"

namespace hg {

#define HG_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
#define HG_RLX_SYSTEM __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM

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

// LDS carve-up (dynamic, sized by the host from batch capacity, K and cols_per_group
// so that the streaming workgroups keep their occupancy):
struct SweepShared {
    uint32_t* mt;      // MT_BUF words: current + next MT19937 block
    double* zig_nx;    // 129 + 129: normal Ziggurat layers staged for the draw
    double* zig_ny;
    double* wpart;     // [BLOCK_WAVES][wstride] per-wave partials of this block's columns (+ sum of eps)
    double* tot;       // NROW*bcap + 1 reduced sums
    double* thr;       // [bcap][K-1]
    double* muk;       // [bcap][K]
    double* logl;      // [bcap][K]
    double* bold;      // [bcap]
    double* mave;
    double* mstd;
    double* dp;        // [bcap] x_j'eps as reduced from the dots (before corrections)
    int32_t* marker;
    int32_t* grp;
    uint32_t* flags;   // see F_* below
    uint32_t* red_u;   // 4 words: block minimum scratch
    uint8_t* ada;
    unsigned char* estage; // BLOCK_WAVES x 8 KiB: eps tile of each wave, filled by LDS-DMA (overlays the tail arrays)
    double* htab;      // 4 x HT_LDS staged hyper tables (denom, logpi, hlog, sdk) when G*K <= HT_LDS
    double* red;       // 128 doubles: exchange buffer of the tail reduction
    double* ev;        // 8 doubles: event hand-off between the walk and the rest of the workgroup
    uint32_t wstride;  // NROW*cpg + 1
    uint32_t bcap;     // batch capacity of this launch
};
enum { F_LAST = 0, F_POS = 1, F_P2PTMO = 2, F_NACC = 3, F_STOP = 4, F_FPOS = 5, F_FMARK = 6, F_ERR = 7 };

constexpr size_t EPS_STAGE_BYTES = (size_t)BLOCK_WAVES * TILE * sizeof(double); // one wave tile of eps per wave (LDS-DMA target)

__host__ __device__ inline size_t sweep_lds_tail_bytes(uint32_t bcap, int K)
{
    size_t n = 0;
    n += (size_t)(NROW * bcap + 1) * 8;                                                       // tot
    n += (size_t)bcap * (K - 1) * 8 + (size_t)2 * bcap * K * 8 + (size_t)4 * bcap * 8;        // thr, muk, logl, bold/mave/mstd/dp
    n += (size_t)2 * bcap * 4 + ((bcap + 15) & ~15u);                                         // marker, grp, ada
    return (n + 15) & ~(size_t)15;
}

__host__ __device__ inline size_t sweep_lds_fixed_bytes(uint32_t cpg)
{
    size_t n = 0;
    n += MT_BUF * 4 + 2 * 130 * 8 + (size_t)4 * HT_LDS * 8 + 128 * 8 + 8 * 8 + 32 + 16;
    n += (size_t)BLOCK_WAVES * (NROW * cpg + 1) * 8;
    return (n + 15) & ~(size_t)15;
}

// fixed region (staged generator / tables / small scratch, live for the whole launch), then a
// UNION: the streaming loop's eps staging tiles and the last arriver's per-batch arrays
__host__ __device__ inline size_t sweep_lds_bytes(uint32_t bcap, uint32_t cpg, int K)
{
    const size_t tail = sweep_lds_tail_bytes(bcap, K);
    return sweep_lds_fixed_bytes(cpg) + (tail > EPS_STAGE_BYTES ? tail : EPS_STAGE_BYTES);
}

__device__ __forceinline__ SweepShared sweep_lds_carve(unsigned char* base, uint32_t bcap, uint32_t cpg, int K)
{
    SweepShared sh;
    unsigned char* q = base;
    sh.mt = reinterpret_cast<uint32_t*>(q); q += MT_BUF * 4;
    sh.zig_nx = reinterpret_cast<double*>(q); q += 130 * 8;
    sh.zig_ny = reinterpret_cast<double*>(q); q += 130 * 8;
    sh.htab = reinterpret_cast<double*>(q); q += (size_t)4 * HT_LDS * 8;
    sh.red = reinterpret_cast<double*>(q); q += 128 * 8;
    sh.ev = reinterpret_cast<double*>(q); q += 8 * 8;
    sh.flags = reinterpret_cast<uint32_t*>(q); q += 32;
    sh.red_u = reinterpret_cast<uint32_t*>(q); q += 16;
    sh.wstride = NROW * cpg + 1;
    sh.wpart = reinterpret_cast<double*>(q);
    q = base + sweep_lds_fixed_bytes(cpg);
    sh.estage = q; // union starts here
    sh.tot = reinterpret_cast<double*>(q); q += (size_t)(NROW * bcap + 1) * 8;
    sh.thr = reinterpret_cast<double*>(q); q += (size_t)bcap * (K - 1) * 8;
    sh.muk = reinterpret_cast<double*>(q); q += (size_t)bcap * K * 8;
    sh.logl = reinterpret_cast<double*>(q); q += (size_t)bcap * K * 8;
    sh.bold = reinterpret_cast<double*>(q); q += (size_t)bcap * 8;
    sh.mave = reinterpret_cast<double*>(q); q += (size_t)bcap * 8;
    sh.mstd = reinterpret_cast<double*>(q); q += (size_t)bcap * 8;
    sh.dp = reinterpret_cast<double*>(q); q += (size_t)bcap * 8;
    sh.marker = reinterpret_cast<int32_t*>(q); q += (size_t)bcap * 4;
    sh.grp = reinterpret_cast<int32_t*>(q); q += (size_t)bcap * 4;
    sh.ada = q;
    sh.bcap = bcap;
    return sh;
}

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

// Per-thread marker metadata for the draw phase; loaded EARLY (before the
// streaming loop) from the sweep-ordered side arrays: no dependent gather.
struct MarkerMeta {
    int marker, grp;
    bool ada, miss;
    double bold, mave, mstd;
};

__device__ __forceinline__ MarkerMeta load_marker_meta(const SweepParams& p, const SweepDesc& d, uint32_t nb, int tid)
{
    MarkerMeta m{-1, 0, false, false, 0.0, 0.0, 0.0};
    if ((uint32_t)tid < nb) {
        const uint32_t j = d.cursor + tid;
        m.marker = p.order[j];
        const int ga = p.s_ga[j];
        m.grp = ga & 0x0fffffff;
        m.ada = (ga & 0x40000000) != 0;
        m.miss = (ga & 0x20000000) != 0;
        m.bold = p.s_bold[j];
        m.mave = p.s_mave[j];
        m.mstd = p.s_mstd[j];
    }
    return m;
}

// Positions (relative to the cursor) t and t + BLOCK of the sweep order: bit 0 = the
// marker's effect is non-zero at sweep start (it WILL change: a predicted event),
// bit 1 = its column has missing calls.  Loaded before the streaming loop.
struct PivotScan {
    uint32_t f0, f1;
};

__device__ __forceinline__ PivotScan load_pivot_scan(const SweepParams& p, const SweepDesc& d, int tid)
{
    PivotScan s{0u, 0u};
    const uint32_t j0 = d.cursor + (uint32_t)tid, j1 = j0 + BLOCK;
    if (j0 < p.M) s.f0 = (p.s_bold[j0] != 0.0 ? 1u : 0u) | ((p.s_ga[j0] & 0x20000000) ? 2u : 0u);
    if (j1 < p.M) s.f1 = (p.s_bold[j1] != 0.0 ? 1u : 0u) | ((p.s_ga[j1] & 0x20000000) ? 2u : 0u);
    return s;
}

// stage generator + normal tables + hyper tables in LDS (issued early as well)
__device__ __forceinline__ void stage_rng(const SweepParams& p, const SweepShared& sh, int tid)
{
    for (int i = tid; i < MT_N; i += BLOCK) sh.mt[i] = p.mt[i];
    for (int i = tid; i < 129; i += BLOCK) {
        sh.zig_nx[i] = p.zig.nx[i];
        sh.zig_ny[i] = p.zig.ny[i];
    }
    if (p.GK <= HT_LDS)
        for (int i = tid; i < 4 * p.GK; i += BLOCK) sh.htab[(i / p.GK) * HT_LDS + (i % p.GK)] = p.denom[i]; // 4 tables are contiguous
}

__device__ __forceinline__ uint32_t block_min_u32(const SweepShared& sh, uint32_t v, int tid)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const uint32_t o = (uint32_t)__shfl_xor((int)v, off, 64);
        v = o < v ? o : v;
    }
    __syncthreads();
    if ((tid & 63) == 0) sh.red_u[tid >> 6] = v;
    __syncthreads();
    uint32_t best = sh.red_u[0];
    for (int w = 1; w < BLOCK_WAVES; ++w) best = sh.red_u[w] < best ? sh.red_u[w] : best;
    return best;
}

// Posterior + draw + bookkeeping for the markers of this batch, given the reduced
// sums in sh.tot: rows [3j, 3j+1, 3j+2] = (s1, s2, A) of batch column j, last row
// = sum of eps.  A_j = sum_i gw_j gw_pivot (integer) for the columns of the
// EXTENSION [nb1, nb2): those dots were taken before the pivot's (position
// nb1-1, a predicted event) update and are corrected here once its new effect is
// known:  x_j'eps_new = x_j'eps_old + dbeta_p * x_j'x_p,
//         x_j'x_p = mstd_j mstd_p (A_j - N mave_j mave_p)   (columns without missing calls).
// Runs in ONE workgroup of 256 threads.
// a5-a7: src/BayesRRm.cpp:1721-1723,1744-1753,1855-1921; dense dot algebra :1785-1790,1809.
__device__ __forceinline__ void sweep_draw_phase(const SweepParams& p, const SweepDesc& d, uint32_t nb1, uint32_t nb2,
                                                 const SweepShared& sh, const MarkerMeta& mm, const PivotScan& scan)
{
    const int tid = threadIdx.x;
    const int K = p.K;
    const uint32_t idx0 = d.rng_idx;
    const bool need_next = idx0 + MAX_BATCH + 64 > (uint32_t)MT_N; // uniform
    if (need_next) mt_next_block(sh.mt, tid);

    // per-column state -> LDS, dot products from the reduced rows
    if ((uint32_t)tid < nb2) {
        sh.marker[tid] = mm.marker;
        sh.grp[tid] = mm.grp;
        sh.ada[tid] = mm.ada ? 1 : 0;
        sh.bold[tid] = mm.bold;
        sh.mave[tid] = mm.mave;
        sh.mstd[tid] = mm.mstd;
        // dense BED form of the reference (src/BayesRRm.cpp:1785-1790,1809):
        // s1 = sum c1*(c2*eps), s2 = sum c2*eps, num = mstd*(s1 - mave*s2).
        // A column without missing calls has s2 == sum of eps, bit for bit
        // (same lanes, same order), so it is not accumulated per column.
        const double s1 = sh.tot[NROW * tid];
        const double s2 = mm.miss ? sh.tot[NROW * tid + 1] : sh.tot[NROW * sh.bcap];
        sh.dp[tid] = mm.mstd * (s1 - mm.mave * s2);
    }
    if (tid == 0) {
        sh.flags[F_POS] = idx0;
        sh.flags[F_NACC] = 0;
        sh.flags[F_ERR] = 0;
    }
    __syncthreads();

    // pending updates handed to the next launch
    int pend_marker[2] = {-1, -1};
    double pend_pv[2][3] = {{0, 0, 0}, {0, 0, 0}};
    int npend = 0;
    unsigned long long nnz_add = 0;

    for (int seg = 0; seg < 2; ++seg) {
        const uint32_t lo = seg ? nb1 : 0u, hi = seg ? nb2 : nb1;
        if (lo >= hi) break; // uniform

        // ---- posterior of [lo, hi), one thread per batch column -----------------
        if ((uint32_t)tid >= lo && (uint32_t)tid < hi && mm.ada) {
            double num = sh.dp[tid];
            if (seg) { // Gram correction for the pivot's update
                const double A = sh.tot[NROW * tid + 2];
                const double xx = mm.mstd * sh.ev[2] * (A - p.n_total * (mm.mave * sh.ev[1]));
                num += sh.ev[0] * xx;
            }
            num += mm.bold * p.n_minus_1;
            double den[MAX_K], lpi[MAX_K], hlg[MAX_K];
            if (p.GK <= HT_LDS) {
                for (int k = 0; k < K; ++k) {
                    den[k] = sh.htab[mm.grp * K + k];
                    lpi[k] = sh.htab[HT_LDS + mm.grp * K + k];
                    hlg[k] = sh.htab[2 * HT_LDS + mm.grp * K + k];
                }
            } else {
                for (int k = 0; k < K; ++k) {
                    den[k] = p.denom[(size_t)mm.grp * K + k];
                    lpi[k] = p.logpi[(size_t)mm.grp * K + k];
                    hlg[k] = p.hlog[(size_t)mm.grp * K + k];
                }
            }
            sh.logl[tid * K] = lpi[0];
            sh.muk[tid * K] = 0.0;
            for (int k = 1; k < K; ++k) {
                double mk = num / den[k];
                sh.muk[tid * K + k] = mk;
                sh.logl[tid * K + k] = lpi[k] - hlg[k] + mk * num * p.i_2sigE;
            }
        }
        __syncthreads();
        // increments of the component walk (:1883-1921), one thread per (marker, component):
        // q[j][kk] = 0 if any |logL_l - logL_kk| > 700 (l >= max(kk,1)) else 1 / sum_l exp(logL_l - logL_kk)
        for (uint32_t it = tid; it < (hi - lo) * (uint32_t)(K - 1); it += BLOCK) {
            const uint32_t j = lo + it / (uint32_t)(K - 1), kk = it % (uint32_t)(K - 1);
            if (!sh.ada[j]) continue;
            const double* L = sh.logl + j * K;
            const double base = L[kk];
            bool big = false;
            for (int l = (kk ? (int)kk : 1); l < K; ++l)
                if (fabs(L[l] - base) > 700.0) big = true;
            double q = 0.0;
            if (!big) {
                double sum = 0.0;
                for (int l = 0; l < K; ++l) sum += exp(L[l] - base);
                q = 1.0 / sum;
            }
            sh.thr[j * (K - 1) + kk] = q;
        }
        __syncthreads();
        if (p.dbg && tid == 0 && seg == 0) p.dbg[3] = wall_clock64();

        // ---- the walk: wave 0 consumes the stream in marker order, 64 at a time --
        if (tid < WAVE) {
            const int lane = tid;
            uint32_t pos = sh.flags[F_POS]; // stream position
            uint32_t naccept = 0;
            bool stopped = false;
            for (uint32_t base = lo; base < hi && !stopped; base += WAVE) {
                const uint32_t j = base + lane;
                const bool valid = j < hi;
                const bool ada = valid && sh.ada[valid ? j : 0] != 0;
                const double bold = valid ? sh.bold[j] : 0.0;
                const int grp = valid ? sh.grp[j] : 0;
                const int marker = valid ? sh.marker[j] : -1;
                const unsigned long long am = __ballot(ada);
                const uint32_t jeff = (uint32_t)__popcll(am & ((1ull << lane) - 1ull));
                int k = 0;
                if (ada) {
                    const uint32_t u = mt_temper(sh.mt[pos + jeff]);
                    const double prob = (double)u * (1.0 / 4294967296.0);
                    k = K - 1;
                    double acum = 0.0; // acum_k = q_0 + ... + q_k, added in the reference's order
                    bool found = false;
                    for (int kk = 0; kk + 1 < K; ++kk) {
                        acum = kk ? acum + sh.thr[j * (K - 1) + kk] : sh.thr[j * (K - 1)];
                        if (!found && prob <= acum) {
                            k = kk;
                            found = true;
                        }
                    }
                }
                const bool event = valid && (ada ? (k != 0 || bold != 0.0) : (bold != 0.0));
                const unsigned long long em = __ballot(event);
                const uint32_t nvalid = (hi - base < (uint32_t)WAVE) ? hi - base : (uint32_t)WAVE;
                const uint32_t f = em ? (uint32_t)(__ffsll((long long)em) - 1) : nvalid; // first event in this chunk
                const uint32_t nacc = (f < nvalid) ? f + 1 : nvalid;

                double bnew = 0.0;
                uint32_t consumed = 0, gerr = 0;
                if ((uint32_t)lane == f && ada && k > 0) {
                    LdsGen g{sh.mt, pos + jeff + 1u, need_next ? (uint32_t)MT_BUF : (uint32_t)MT_N, 0u};
                    ZigTables zt{sh.zig_nx, sh.zig_ny, p.zig.ex, p.zig.ey};
                    const double sd = (p.GK <= HT_LDS) ? sh.htab[3 * HT_LDS + grp * K + k] : p.sdk[(size_t)grp * K + k];
                    bnew = norm_rng_sd(g, zt, sh.muk[j * K + k], sd);
                    consumed = g.pos - (pos + jeff + 1u);
                    gerr = g.err;
                }
                const double dbeta = bold - bnew;

                // results of accepted markers (:1892,:1899-1905,:1924-1925)
                if ((uint32_t)lane < nacc) {
                    if (ada) {
                        const int kk = ((uint32_t)lane == f) ? k : 0;
                        p.beta[marker] = ((uint32_t)lane == f) ? bnew : 0.0;
                        p.comp[marker] = kk;
                        p.acum[marker] = sh.thr[j * (K - 1)];
                        atomicAdd(&p.cass[grp * K + kk], 1);
                    } else {
                        p.beta[marker] = 0.0;
                        p.acum[marker] = 1.0;
                    }
                }
                const uint32_t used = (uint32_t)__popcll(am & ((nacc >= 64u) ? ~0ull : ((1ull << nacc) - 1ull)));
                naccept += nacc;
                pos += used;
                if (f < nvalid) { // an event ends the segment (later dots are stale)
                    stopped = true;
                    const int src = (int)f;
                    const double f_dbeta = __shfl(dbeta, src, 64);
                    const int f_marker = __shfl(marker, src, 64);
                    pos += (uint32_t)__shfl((int)consumed, src, 64);
                    const uint32_t f_err = (uint32_t)__shfl((int)gerr, src, 64);
                    if (lane == 0) {
                        sh.ev[0] = f_dbeta;
                        sh.ev[1] = sh.mave[base + f];
                        sh.ev[2] = sh.mstd[base + f];
                        sh.flags[F_FPOS] = base + f;
                        sh.flags[F_FMARK] = (uint32_t)f_marker;
                        if (f_err) sh.flags[F_ERR] = f_err;
                    }
                }
            }
            if (lane == 0) {
                sh.flags[F_POS] = pos;
                sh.flags[F_NACC] += naccept;
                sh.flags[F_STOP] = stopped ? 1u : 0u;
            }
        }
        __syncthreads();

        // an event: its update is pending for the next launch
        const bool stopped = sh.flags[F_STOP] != 0;
        if (stopped) {
            const double db = sh.ev[0], av = sh.ev[1], sd = sh.ev[2];
            if (db != 0.0) {
                pend_marker[npend] = (int)sh.flags[F_FMARK];
                pend_pv[npend][0] = -(av * sd * db);
                pend_pv[npend][1] = db * (1.0 - av) * sd;
                pend_pv[npend][2] = db * (2.0 - av) * sd;
                ++npend;
                ++nnz_add;
            }
        }
        // go on into the extension only if segment 0 ran to its end and ended ON the pivot
        if (!(seg == 0 && stopped && sh.flags[F_FPOS] == nb1 - 1u && nb2 > nb1)) break;
        __syncthreads(); // sh.ev stays valid for the corrections; flags are rewritten by the next walk
    }

    // ---- hand the state to the next launch ---------------------------------------
    const uint32_t naccept = sh.flags[F_NACC];
    const uint32_t pos = sh.flags[F_POS];
    if (tid == 0) {
        SweepDesc n = d;
        n.cursor = d.cursor + naccept;
        if (d.pend_marker[0] >= 0) n.cur = d.cur ^ 1u;
        for (int q = 0; q < 2; ++q) {
            n.pend_marker[q] = pend_marker[q];
            for (int c = 0; c < 3; ++c) n.pv[q][c] = pend_pv[q][c];
        }
        n.nnz = d.nnz + nnz_add;
        n.rng_idx = (pos >= (uint32_t)MT_N) ? pos - (uint32_t)MT_N : pos;
        n.launches = d.launches + 1;
        n.seq = d.seq + 1;
        n.accepted_sum = d.accepted_sum + naccept;
        if (sh.flags[F_ERR]) n.error = sh.flags[F_ERR];
        *p.desc = n;
        if (p.dbg) { // accumulate stage durations over all launches: [8+i] += t[i+1]-t[i], [15] = count
            p.dbg[4] = wall_clock64();
            for (int i = 0; i < 4; ++i) p.dbg[8 + i] += p.dbg[i + 1] - p.dbg[i];
            p.dbg[13] += p.dbg[6] - p.dbg[5]; // last arriver: entry -> loop done
            p.dbg[14] += p.dbg[7] - p.dbg[6]; // last arriver: loop done -> drained
            p.dbg[16] += p.dbg[1] - p.dbg[7]; // last arriver: drained -> past ticket
            p.dbg[17] += p.dbg[5] - p.dbg[0]; // first block entry -> last arriver entry
            p.dbg[12] += naccept;
            p.dbg[15] += 1;
        }
    }

    // ---- plan of the next launch ---------------------------------------------------
    // batch  = up to and including the first predicted event after the new cursor (the pivot);
    // batch2 = further up to and including the NEXT predicted event, as long as the columns
    //          involved have no missing calls (their dots get the Gram correction above).
    {
        const uint32_t cap = p.batch_limit;
        const uint32_t q0 = (uint32_t)tid, q1 = (uint32_t)tid + BLOCK; // positions relative to the old cursor
        uint32_t cand = 0xffffffffu;
        if ((scan.f0 & 1u) && q0 >= naccept) cand = q0 - naccept;
        else if ((scan.f1 & 1u) && q1 >= naccept) cand = q1 - naccept;
        const uint32_t p1 = block_min_u32(sh, cand, tid); // distance of the pivot from the new cursor
        uint32_t want1 = cap, want2 = cap;
        if (p1 != 0xffffffffu && p1 + 1u <= cap) {
            want1 = p1 + 1u;
            // is the pivot's own column free of missing calls?
            uint32_t pm = 0xffffffffu;
            if (q0 == naccept + p1) pm = (scan.f0 >> 1) & 1u;
            if (q1 == naccept + p1) pm = (scan.f1 >> 1) & 1u;
            const uint32_t pivot_miss = block_min_u32(sh, pm, tid);
            // first later position that ends the extension: a predicted event or a column with missing calls
            uint32_t c2 = 0xffffffffu, c2m = 0u;
            if (q0 > naccept + p1 && scan.f0) {
                c2 = q0 - naccept;
                c2m = (scan.f0 >> 1) & 1u;
            } else if (q1 > naccept + p1 && scan.f1) {
                c2 = q1 - naccept;
                c2m = (scan.f1 >> 1) & 1u;
            }
            const uint32_t e = block_min_u32(sh, c2, tid);
            uint32_t em = 0xffffffffu;
            if (c2 == e && e != 0xffffffffu) em = c2m;
            const uint32_t e_miss = block_min_u32(sh, em, tid);
            if (!p.gram || pivot_miss != 0u) want2 = want1;
            else if (e == 0xffffffffu) want2 = cap;
            else want2 = e_miss ? e : e + 1u; // a column with missing calls stays out of the extension
            if (want2 > want1 + p.ext_limit) want2 = want1 + p.ext_limit;
            if (want2 > cap) want2 = cap;
            if (want2 < want1) want2 = want1;
        }
        if (tid == 0) {
            p.desc->batch = want1;
            p.desc->batch2 = want2;
        }
    }
    __syncthreads();
    // generator crossed into the next block: make it the current one
    if (pos >= (uint32_t)MT_N)
        for (int i = tid; i < MT_N; i += BLOCK) p.mt[i] = sh.mt[MT_N + i];
}

extern __shared__ __attribute__((aligned(16))) unsigned char hg_smem[];

// Inner step of the dot product for one slot S of four columns: field extract,
// int -> f64, fused multiply-add, issued as three groups of four so that the
// in-order SIMD always has three independent instructions between a producer and
// its consumer.  a += double((g >> 2S) & 3) * e  -- the product is exact, one rounding per add.
template <int S>
__device__ __forceinline__ void fma_slot4(uint32_t g0, uint32_t g1, uint32_t g2, uint32_t g3, double e, double& a0, double& a1,
                                          double& a2, double& a3)
{
    uint32_t t0, t1, t2, t3;
    double w0, w1, w2, w3;
    asm("v_bfe_u32 %[t0], %[g0], %[sh], 2\n\t"
        "v_bfe_u32 %[t1], %[g1], %[sh], 2\n\t"
        "v_bfe_u32 %[t2], %[g2], %[sh], 2\n\t"
        "v_bfe_u32 %[t3], %[g3], %[sh], 2\n\t"
        "v_cvt_f64_u32 %[w0], %[t0]\n\t"
        "v_cvt_f64_u32 %[w1], %[t1]\n\t"
        "v_cvt_f64_u32 %[w2], %[t2]\n\t"
        "v_cvt_f64_u32 %[w3], %[t3]\n\t"
        "v_fmac_f64 %[a0], %[w0], %[e]\n\t"
        "v_fmac_f64 %[a1], %[w1], %[e]\n\t"
        "v_fmac_f64 %[a2], %[w2], %[e]\n\t"
        "v_fmac_f64 %[a3], %[w3], %[e]"
        : [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2),
          [t3] "=&v"(t3), [w0] "=&v"(w0), [w1] "=&v"(w1), [w2] "=&v"(w2), [w3] "=&v"(w3)
        : [g0] "v"(g0), [g1] "v"(g1), [g2] "v"(g2), [g3] "v"(g3), [e] "v"(e), [sh] "i"(2 * S));
}

template <int... S>
__device__ __forceinline__ void fma_slots4(uint32_t g0, uint32_t g1, uint32_t g2, uint32_t g3, const double (&e)[IPT], double& a0,
                                           double& a1, double& a2, double& a3, std::integer_sequence<int, S...>)
{
    (fma_slot4<S>(g0, g1, g2, g3, e[S], a0, a1, a2, a3), ...);
}

// sum over the 16 slots of gw_a * gw_b (2-bit fields with values 0,1,2): integer Gram term
__device__ __forceinline__ uint32_t gram16(uint32_t ga, uint32_t gb)
{
    const uint32_t la = ga & 0x55555555u, ha = (ga >> 1) & 0x55555555u;
    const uint32_t lb = gb & 0x55555555u, hb = (gb >> 1) & 0x55555555u;
    return (uint32_t)__popc(la & lb) + 2u * (uint32_t)(__popc(la & hb) + __popc(ha & lb)) + 4u * (uint32_t)__popc(ha & hb);
}

// Cross-GPU sum of the batch rows held in sh.tot, inside the launch: push my
// rows into every rank's mailbox (system-scope stores over xGMI), publish one
// flag per destination, wait for the nranks flags in my own mailbox, then add
// the contributions in RANK ORDER so that every GPU gets the same bits.  Two
// parities: a peer can be at most one batch ahead (it needs my rows of batch
// b+1 before it can finish b+1).  Bounded spin: returns false on timeout.
__device__ __forceinline__ bool p2p_exchange(const SweepParams& p, const SweepDesc& d, uint32_t nb, const SweepShared& sh)
{
    const int tid = threadIdx.x;
    const int nr = p.p2p.nranks, me = p.p2p.rank;
    const uint32_t nrows = NROW * nb + 1;
    const uint32_t parity = (uint32_t)(d.seq & 1ull);
    const unsigned long long epoch = d.seq + 1ull;
    const size_t slot = (size_t)(parity * MAX_RANKS + (uint32_t)me) * ROWS_CAP;
    for (uint32_t it = tid; it < nrows * (uint32_t)nr; it += BLOCK) {
        const uint32_t dst = it / nrows, rr = it % nrows;
        const double v = sh.tot[(rr == NROW * nb) ? NROW * sh.bcap : rr];
        __hip_atomic_store(p.p2p.data[dst] + slot + rr, v, HG_RLX_SYSTEM);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid < nr) __hip_atomic_store(p.p2p.flags[tid] + parity * MAX_RANKS + me, epoch, HG_RLX_SYSTEM);
    bool ok = true;
    if (tid < nr) {
        const unsigned long long* f = p.p2p.flags[me] + parity * MAX_RANKS + tid;
        const unsigned long long t0 = wall_clock64();
        while (__hip_atomic_load(f, HG_RLX_SYSTEM) != epoch) {
            if (wall_clock64() - t0 > 300000000ull) { // 3 s at 100 MHz: a peer is gone
                ok = false;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    if (!ok) sh.flags[F_P2PTMO] = 1u;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    __syncthreads();
    if (sh.flags[F_P2PTMO]) return false;
    for (uint32_t rr = tid; rr < nrows; rr += BLOCK) {
        double acc = 0.0;
        for (int r = 0; r < nr; ++r)
            acc += __hip_atomic_load(p.p2p.data[me] + (size_t)(parity * MAX_RANKS + (uint32_t)r) * ROWS_CAP + rr, HG_RLX_SYSTEM);
        sh.tot[(rr == NROW * nb) ? NROW * sh.bcap : rr] = acc;
    }
    __syncthreads();
    return true;
}

// One launch of the sweep.  grid = (S, ceil(batch_limit/CPG)): blockIdx.y owns a
// group of up to CPG batch columns, blockIdx.x a strided set of tile groups
// (4 wave tiles = 4096 individuals each).  Every lane keeps the sums of its
// columns in registers across all its tiles; one wave/block reduction per
// launch, then per-slice partial rows for the last arriver.
template <int CPG>
__global__ __launch_bounds__(BLOCK, (CPG <= 8 ? 3 : 2)) void k_sweep_batch(SweepParams p)
{
    const SweepShared sh = sweep_lds_carve(hg_smem, p.batch_cap, p.cols_per_group, p.K);
    const SweepDesc d = *p.desc;
    const bool pend = d.pend_marker[0] >= 0;
    const uint32_t remaining = (d.cursor < p.M) ? p.M - d.cursor : 0u;
    const uint32_t nb2 = d.batch2 < remaining ? d.batch2 : remaining; // all columns of this launch
    const uint32_t nb1 = d.batch < nb2 ? d.batch : nb2;               // ... of which [nb1, nb2) are the extension
    const uint32_t nb = nb2;
    if ((nb == 0 && !pend) || d.error) return; // whole grid agrees: nothing left to do

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // 1-D grid of slices_max * groups_max workgroups; the ACTIVE ones are the first S * nactive
    // in dispatch order (idle ones behind them leave at once and delay nobody)
    const uint32_t nactive = (nb + CPG - 1) / CPG > 0 ? (nb + CPG - 1) / CPG : 1u;
    // slices actually used: keep the active workgroups co-resident (3 per CU at this register
    // budget; 2 at CPG = 16) -- a second wave of workgroups would double the streaming phase
    constexpr uint32_t RES = (CPG >= 16) ? 512u : 768u; // co-resident workgroups at this instantiation's register budget
    const uint32_t S = (RES / nactive) < p.slices_max ? ((RES / nactive) ? RES / nactive : 1u) : p.slices_max;
    if (blockIdx.x >= S * nactive) return;
    const uint32_t slice = blockIdx.x % S, group = blockIdx.x / S;
    const uint32_t c0 = group * CPG;
    const uint32_t c1 = (c0 + CPG < nb) ? c0 + CPG : nb;
    const uint32_t ncol = (c1 > c0) ? c1 - c0 : 0u;
    const bool first_group = group == 0;
    const double* eps_in = d.cur ? p.eps1 : p.eps0;
    double* eps_out = d.cur ? p.eps0 : p.eps1;
    const uint32_t ntg = p.n_pad / BLOCK_IND; // tile groups

    // latency-bound loads of the draw phase, issued by EVERY workgroup before the
    // streaming loop (any of them may turn out to be the last arriver)
    const MarkerMeta meta = p.sums_out ? MarkerMeta{-1, 0, false, false, 0.0, 0.0, 0.0} : load_marker_meta(p, d, nb, tid);
    if (!p.sums_out) stage_rng(p, sh, tid);
    const PivotScan scan = p.sums_out ? PivotScan{0u, 0u} : load_pivot_scan(p, d, tid);
    if (p.dbg && blockIdx.x == 0 && tid == 0) p.dbg[0] = wall_clock64();
    const unsigned long long t_entry = p.dbg ? wall_clock64() : 0ull;

    unsigned long long t_loop = 0ull;
    double a1[CPG], a2[CPG], sall = 0.0;
    uint32_t ag[CPG]; // integer Gram partial with the pivot column (extension columns only)
#pragma unroll
    for (int c = 0; c < CPG; ++c) {
        a1[c] = a2[c] = 0.0;
        ag[c] = 0u;
    }
    const uint8_t* colp[CPG];
    bool cmiss[CPG]; // wave-uniform: column has missing calls -> needs its own s2
    bool cgram[CPG]; // wave-uniform: column lies in the extension -> needs its Gram term with the pivot
    bool any_gram = false;
#pragma unroll
    for (int c = 0; c < CPG; ++c) {
        const uint32_t j = (c0 + c < nb) ? c0 + c : (nb ? nb - 1 : 0);
        const int marker = nb ? p.order[d.cursor + j] : 0;
        cmiss[c] = nb ? ((p.s_ga[d.cursor + j] & 0x20000000) != 0) : false;
        cgram[c] = (c0 + c < nb) && (c0 + c >= nb1) && nb1 > 0;
        any_gram = any_gram || cgram[c];
        colp[c] = p.bed + (size_t)marker * p.stride + (lane << 2);
    }
    const uint8_t* pivp = p.bed + (size_t)(any_gram ? p.order[d.cursor + nb1 - 1] : 0) * p.stride + (lane << 2);
    const uint8_t* pendp0 = p.bed + (size_t)(pend ? d.pend_marker[0] : 0) * p.stride + (lane << 2);
    const bool pend1 = d.pend_marker[1] >= 0;
    const uint8_t* pendp1 = p.bed + (size_t)(pend1 ? d.pend_marker[1] : 0) * p.stride + (lane << 2);

    {
        // eps tiles arrive by LDS-DMA (global_load_lds_dwordx4: no VGPRs, lane-linear 1 KiB pieces --
        // exactly the permuted eps layout) one tile ahead of the arithmetic; column dwords of the
        // next tile are prefetched into registers.
        unsigned char* const est = sh.estage + (size_t)wave * (TILE * sizeof(double));
        auto dma_eps = [&](uint32_t tile) {
            const double* g = eps_in + ((size_t)tile << 10) + (lane << 1);
#pragma unroll
            for (int k = 0; k < 8; ++k)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + k * 128),
                                                 (__attribute__((address_space(3))) void*)(est + k * 1024), 16, 0, 0);
        };
        uint32_t w[CPG], wn[CPG], wpiv = 0, wpivn = 0, wp0 = 0, wp0n = 0, wp1 = 0, wp1n = 0;
        uint32_t tg = slice;
        if (tg < ntg) {
            const uint32_t tile = tg * BLOCK_WAVES + wave;
            dma_eps(tile);
#pragma unroll
            for (int c = 0; c < CPG; ++c) w[c] = *reinterpret_cast<const uint32_t*>(colp[c] + ((size_t)tile << 8));
            if (any_gram) wpiv = *reinterpret_cast<const uint32_t*>(pivp + ((size_t)tile << 8));
            if (pend) wp0 = *reinterpret_cast<const uint32_t*>(pendp0 + ((size_t)tile << 8));
            if (pend1) wp1 = *reinterpret_cast<const uint32_t*>(pendp1 + ((size_t)tile << 8));
        }
        for (; tg < ntg; tg += S) {
            const uint32_t tile = tg * BLOCK_WAVES + wave;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this tile's DMA (and column dwords) have landed
            double e[IPT];
            {
                const double2* lp = reinterpret_cast<const double2*>(est) + lane;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const double2 v = lp[k * 64];
                    e[2 * k] = v.x;
                    e[2 * k + 1] = v.y;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // eps is in registers: the staging tile may be overwritten
            const uint32_t tgn = tg + S;
            if (tgn < ntg) {
                const uint32_t tilen = tgn * BLOCK_WAVES + wave;
                dma_eps(tilen);
#pragma unroll
                for (int c = 0; c < CPG; ++c) wn[c] = *reinterpret_cast<const uint32_t*>(colp[c] + ((size_t)tilen << 8));
                if (any_gram) wpivn = *reinterpret_cast<const uint32_t*>(pivp + ((size_t)tilen << 8));
                if (pend) wp0n = *reinterpret_cast<const uint32_t*>(pendp0 + ((size_t)tilen << 8));
                if (pend1) wp1n = *reinterpret_cast<const uint32_t*>(pendp1 + ((size_t)tilen << 8));
            }
            uint32_t gwp = 0;
            if (any_gram) {
                uint32_t nmp;
                code_weights(wpiv, gwp, nmp);
            }
            if (pend) { // the previous launch's event(s), in order
                apply_update16(wp0, d.pv[0][0], d.pv[0][1], d.pv[0][2], e);
                if (pend1) apply_update16(wp1, d.pv[1][0], d.pv[1][1], d.pv[1][2], e);
                if (first_group) store_eps16(eps_out, tile, lane, e);
            }
            if (first_group) {
#pragma unroll
                for (int i = 0; i < IPT; ++i) sall += e[i];
            }
            uint32_t gw[CPG], nm[CPG];
#pragma unroll
            for (int c = 0; c < CPG; ++c) code_weights(w[c], gw[c], nm[c]);
            // s1 += (g*nm) * eps: weight 0/1/2 is exact, one rounding per add; each column adds
            // its slots in increasing order
            if constexpr (CPG % 4 == 0) {
#pragma unroll
                for (int c0g = 0; c0g < CPG; c0g += 4)
                    fma_slots4(gw[c0g], gw[c0g + 1], gw[c0g + 2], gw[c0g + 3], e, a1[c0g], a1[c0g + 1], a1[c0g + 2], a1[c0g + 3],
                               std::make_integer_sequence<int, IPT>{});
            } else {
#pragma unroll
                for (int s = 0; s < IPT; ++s) {
#pragma unroll
                    for (int c = 0; c < CPG; ++c) a1[c] = __builtin_fma((double)((gw[c] >> (2 * s)) & 3u), e[s], a1[c]);
                }
            }
#pragma unroll
            for (int c = 0; c < CPG; ++c) {
                if (cmiss[c]) {
#pragma unroll
                    for (int s = 0; s < IPT; ++s) a2[c] = __builtin_fma((double)((nm[c] >> (2 * s)) & 1u), e[s], a2[c]);
                }
                if (cgram[c]) ag[c] += gram16(gw[c], gwp);
            }
            if (tgn < ntg) {
#pragma unroll
                for (int c = 0; c < CPG; ++c) w[c] = wn[c];
                wpiv = wpivn;
                wp0 = wp0n;
                wp1 = wp1n;
            }
        }
        __syncthreads(); // every wave is done with its staging tile: the union region may be reused
        t_loop = p.dbg ? wall_clock64() : 0ull;
        // one cross-lane reduction per launch
#pragma unroll
        for (int c = 0; c < CPG; ++c) {
            const double t1 = wave_sum(a1[c]), t2 = wave_sum(a2[c]);
            const double tg = any_gram ? wave_sum((double)ag[c]) : 0.0; // exact: integers far below 2^53
            if (lane == 0) {
                sh.wpart[wave * sh.wstride + NROW * c] = t1;
                sh.wpart[wave * sh.wstride + NROW * c + 1] = t2;
                sh.wpart[wave * sh.wstride + NROW * c + 2] = tg;
            }
        }
        if (first_group) {
            const double t = wave_sum(sall);
            if (lane == 0) sh.wpart[wave * sh.wstride + NROW * CPG] = t;
        }
    }
    __syncthreads();

    // block partial = waves 0..3 in order, published write-through (sc1)
    {
        const uint32_t nrow = NROW * ncol;
        for (uint32_t t = tid; t < nrow; t += BLOCK) {
            double v = sh.wpart[t];
            v += sh.wpart[sh.wstride + t];
            v += sh.wpart[2 * sh.wstride + t];
            v += sh.wpart[3 * sh.wstride + t];
            __hip_atomic_store(p.partials + (size_t)slice * ROWS_CAP + (NROW * c0 + t), v, HG_RLX_AGENT);
        }
        if (first_group && tid == BLOCK - 1) {
            const uint32_t t = NROW * CPG;
            double v = sh.wpart[t];
            v += sh.wpart[sh.wstride + t];
            v += sh.wpart[2 * sh.wstride + t];
            v += sh.wpart[3 * sh.wstride + t];
            __hip_atomic_store(p.partials + (size_t)slice * ROWS_CAP + NROW * MAX_BATCH, v, HG_RLX_AGENT);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every storing wave drains (eps + partials)
    __syncthreads();
    const unsigned long long t_drain = p.dbg ? wall_clock64() : 0ull;
    if (tid == 0) {
        const uint32_t t = __hip_atomic_fetch_add(p.ticket, 1u, HG_RLX_AGENT);
        sh.flags[F_LAST] = (t == S * nactive - 1u) ? 1u : 0u;
    }
    __syncthreads();
    if (!sh.flags[F_LAST]) return;

    // ---- last-arriving workgroup ---------------------------------------------
    if (p.dbg && tid == 0) {
        p.dbg[1] = wall_clock64();
        p.dbg[5] = t_entry;
        p.dbg[6] = t_loop;
        p.dbg[7] = t_drain;
    }

    // fixed-order reduction over the S slices.  partials is [slice][row], so a
    // wave's load of one slice covers 64 consecutive rows (coalesced).  Threads
    // 0..127 sum slices 0..31 of row t, threads 128..255 slices 32..63 (all 32
    // loads in flight); row total = (slices 0..31) + (slices 32..63).
    {
        const uint32_t nrows = NROW * nb + 1;
        const uint32_t half = tid >> 7, rl = tid & 127u;
        for (uint32_t rr0 = 0; rr0 < nrows; rr0 += 128) {
            const uint32_t rr = rr0 + rl;
            const bool live = rr < nrows;
            const uint32_t r = (rr == NROW * nb) ? NROW * MAX_BATCH : rr;
            const double* col = p.partials + (size_t)(half * 32u) * ROWS_CAP + (live ? r : 0);
            double v[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) v[u] = (live && half * 32u + u < S) ? __hip_atomic_load(col + (size_t)u * ROWS_CAP, HG_RLX_AGENT) : 0.0;
            double acc = 0.0;
#pragma unroll
            for (int u = 0; u < 32; ++u) acc += v[u];
            if (rr0) __syncthreads(); // previous round's exchange buffer is free again
            if (half == 1) sh.red[rl] = acc;
            __syncthreads();
            if (live && half == 0) sh.tot[(rr == NROW * nb) ? NROW * sh.bcap : rr] = acc + sh.red[rl];
        }
    }
    if (tid == 0) {
        __hip_atomic_store(p.ticket, 0u, HG_RLX_AGENT);
        sh.flags[F_P2PTMO] = 0u;
    }
    __syncthreads();
    if (p.dbg && tid == 0) p.dbg[2] = wall_clock64();

    if (p.p2p.nranks > 1 && !p.sums_out) { // multi-GPU, in-launch exchange
        if (!p2p_exchange(p, d, nb, sh)) {
            if (tid == 0) {
                SweepDesc n = d;
                n.error = 3u;
                *p.desc = n;
            }
            return;
        }
    }
    if (p.sums_out) { // multi-GPU: hand the local sums to the all-reduce
        for (int r = tid; r < NROW * MAX_BATCH + 1; r += BLOCK) {
            double v = 0.0;
            if (r < NROW * (int)nb) v = sh.tot[r];
            if (r == NROW * MAX_BATCH) v = sh.tot[NROW * sh.bcap];
            p.sums_out[r] = v;
        }
        return;
    }
    sweep_draw_phase(p, d, nb1, nb2, sh, meta, scan);
}

// Multi-GPU second half: sums_out has been all-reduced over ranks.
__global__ __launch_bounds__(BLOCK) void k_sweep_draw(SweepParams p)
{
    const SweepShared sh = sweep_lds_carve(hg_smem, p.batch_cap, p.cols_per_group, p.K);
    const SweepDesc d = *p.desc;
    const bool pend = d.pend_marker[0] >= 0;
    const uint32_t remaining = (d.cursor < p.M) ? p.M - d.cursor : 0u;
    const uint32_t nb2 = d.batch2 < remaining ? d.batch2 : remaining;
    const uint32_t nb1 = d.batch < nb2 ? d.batch : nb2;
    if ((nb2 == 0 && !pend) || d.error) return;
    const MarkerMeta meta = load_marker_meta(p, d, nb2, threadIdx.x);
    const PivotScan scan = load_pivot_scan(p, d, threadIdx.x);
    stage_rng(p, sh, threadIdx.x);
    for (int r = threadIdx.x; r < NROW * (int)nb2; r += BLOCK) sh.tot[r] = p.sums_out[r];
    if (threadIdx.x == 0) sh.tot[NROW * sh.bcap] = p.sums_out[NROW * MAX_BATCH];
    __syncthreads();
    sweep_draw_phase(p, d, nb1, nb2, sh, meta, scan);
}

} // namespace hg
