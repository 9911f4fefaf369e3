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

// The few descriptor fields the streaming workgroups need (the whole SweepDesc in scalar
// registers would crowd out the loop's pointers).
struct DescHead {
    uint32_t cursor, cur, rng_idx, error, carry_n, carry_left;
    unsigned long long seq;
    int32_t pend_marker[MAX_SEG];
    uint32_t seg_end[MAX_SEG];
};

__device__ __forceinline__ DescHead load_desc_head(const SweepDesc* g)
{
    DescHead d;
    d.cursor = g->cursor;
    d.cur = g->cur;
    d.rng_idx = g->rng_idx;
    d.error = g->error;
    d.seq = g->seq;
    d.carry_n = g->carry_n;
    d.carry_left = g->carry_left;
#pragma unroll
    for (int q = 0; q < MAX_SEG; ++q) {
        d.pend_marker[q] = g->pend_marker[q];
        d.seg_end[q] = g->seg_end[q];
    }
    return d;
}

// LDS carve-up (dynamic, sized by the host from batch capacity, K and cols_per_group
// so that the streaming workgroups keep their occupancy):
struct SweepShared {
    uint32_t* mt;      // MT_BUF words: current + next MT19937 block
    double* zig_nx;    // 129 + 129: normal Ziggurat layers staged for the draw
    double* zig_ny;
    double* wpart;     // [BLOCK_WAVES][wstride] per-wave partials of this block's columns
    double* tot;       // NROW*bcap + 1 reduced sums
    double* thr;       // [bcap][K-1]
    double* numf;      // [bcap] full numerator of a column once its posterior is evaluated (dot, Gram corrections, old effect's term)
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
    double* ev;        // 3 per segment: (dbeta, mave, mstd) of the event that ended it
    double* pvl;       // 3*MAX_SEG: update constants of the events this launch hands on (draw phase)
    double* pel;       // 3*MAX_SEG: (dbeta, mave, mstd) of the same events
    double* pev;       // 3*MAX_SEG: (dbeta, mave, mstd) of the PENDING updates of this launch (staged from the descriptor)
    double2* pvt;      // [MAX_SEG][16]: the pending updates as a table over a PAIR of 2-bit codes (see apply_update16_lds)
    int32_t* pmk;      // MAX_SEG: markers of the events this launch hands on
    uint8_t* scanf;    // 2*BLOCK flag bytes of the sweep positions after the cursor (bit 0 predicted event, bit 1 missing calls)
    uint32_t wstride;  // NROW*cpg + 1
    uint32_t bcap;     // batch capacity of this launch
};
enum { F_LAST = 0, F_POS = 1, F_P2PTMO = 2, F_NACC = 3, F_STOP = 4, F_FPOS = 5, F_FMARK = 6, F_ERR = 7 };

constexpr size_t EPS_STAGE_BYTES = (size_t)BLOCK_WAVES * TILE * sizeof(double); // one wave tile of eps per wave (LDS-DMA target)

__host__ __device__ inline size_t sweep_lds_tail_bytes(uint32_t bcap, int K, int nr)
{
    size_t n = 0;
    n += (size_t)(nr * bcap + 1) * 8;                                                       // tot
    n += (size_t)bcap * (K - 1) * 8 + (size_t)2 * bcap * 8;                                    // thr, dp, numf
    return (n + 15) & ~(size_t)15;
}

// per-column metadata of the batch, staged by every workgroup BEFORE the streaming loop (any of them may
// turn out to be the last arriver): lives in the fixed region so that it is neither held in registers
// across the loop nor overlaid by the eps staging tiles
__host__ __device__ inline size_t sweep_lds_meta_bytes(uint32_t bcap)
{
    return (size_t)3 * bcap * 8 + (size_t)2 * bcap * 4 + ((bcap + 15) & ~15u); // bold, mave, mstd, marker, grp, ada
}

__host__ __device__ inline size_t sweep_lds_fixed_bytes(uint32_t cpg, int nr, uint32_t bcap)
{
    size_t n = sweep_lds_meta_bytes(bcap);
    n += MT_BUF * 4 + 2 * 130 * 8 + (size_t)4 * HT_LDS * 8 + MAX_SEG * 16 * 16 + 16 * 8 + 32 + 16 + 2 * BLOCK + 3 * 3 * MAX_SEG * 8 + 16;
    n += (size_t)BLOCK_WAVES * (nr * cpg + 1) * 8;
    return (n + 15) & ~(size_t)15;
}

// fixed region (staged generator / tables / small scratch, live for the whole launch), then a
// UNION: the streaming loop's eps staging tiles and the last arriver's per-batch arrays
__host__ __device__ inline size_t sweep_lds_bytes(uint32_t bcap, uint32_t cpg, int K, int nr)
{
    const size_t tail = sweep_lds_tail_bytes(bcap, K, nr);
    return sweep_lds_fixed_bytes(cpg, nr, bcap) + (tail > EPS_STAGE_BYTES ? tail : EPS_STAGE_BYTES);
}

__device__ __forceinline__ SweepShared sweep_lds_carve(unsigned char* base, uint32_t bcap, uint32_t cpg, int K, int nr)
{
    SweepShared sh;
    unsigned char* q = base;
    // first, on a 256-byte boundary each (the streaming loop ORs the index into the address): the pending updates as tables over
    // a pair of codes; live for the whole launch (the ahead phase streams with them after the hand-off)
    sh.pvt = reinterpret_cast<double2*>(q); q += MAX_SEG * 16 * 16;
    sh.mt = reinterpret_cast<uint32_t*>(q); q += MT_BUF * 4;
    sh.zig_nx = reinterpret_cast<double*>(q); q += 130 * 8;
    sh.zig_ny = reinterpret_cast<double*>(q); q += 130 * 8;
    sh.htab = reinterpret_cast<double*>(q); q += (size_t)4 * HT_LDS * 8;
    // the exchange buffer of the group-stage reductions (1 KiB) lies on the SECOND generator block: that one is written only by
    // the draw phase (mt_next_block), after the drawing workgroup's own group stage.  The carve-up must stay within a third of
    // a compute unit's LDS in the hardware's allocation granules (3 x 42 x 1280 B): 32 bytes more once cost a second round
    // of workgroups, +10 us per launch
    sh.red = reinterpret_cast<double*>(sh.mt + MT_N);
    static_assert((MT_BUF - MT_N) * 4 >= 128 * 8 && (MT_N * 4) % 8 == 0, "the exchange buffer fits the second generator block");
    sh.ev = reinterpret_cast<double*>(q); q += 16 * 8;
    sh.flags = reinterpret_cast<uint32_t*>(q); q += 32;
    sh.red_u = reinterpret_cast<uint32_t*>(q); q += 16;
    sh.scanf = q; q += 2 * BLOCK;
    sh.pvl = reinterpret_cast<double*>(q); q += 3 * MAX_SEG * 8;
    sh.pel = reinterpret_cast<double*>(q); q += 3 * MAX_SEG * 8;
    sh.pev = reinterpret_cast<double*>(q); q += 3 * MAX_SEG * 8;
    sh.pmk = reinterpret_cast<int32_t*>(q); q += 16;
    sh.wstride = nr * cpg + 1;
    sh.wpart = reinterpret_cast<double*>(q); q += (size_t)BLOCK_WAVES * sh.wstride * 8;
    sh.bold = reinterpret_cast<double*>(q); q += (size_t)bcap * 8;
    sh.mave = reinterpret_cast<double*>(q); q += (size_t)bcap * 8;
    sh.mstd = reinterpret_cast<double*>(q); q += (size_t)bcap * 8;
    sh.marker = reinterpret_cast<int32_t*>(q); q += (size_t)bcap * 4;
    sh.grp = reinterpret_cast<int32_t*>(q); q += (size_t)bcap * 4;
    sh.ada = q;
    q = base + sweep_lds_fixed_bytes(cpg, nr, bcap);
    sh.estage = q; // union starts here
    sh.tot = reinterpret_cast<double*>(q); q += (size_t)(nr * bcap + 1) * 8;
    sh.thr = reinterpret_cast<double*>(q); q += (size_t)bcap * (K - 1) * 8;
    sh.numf = reinterpret_cast<double*>(q); q += (size_t)bcap * 8;
    sh.dp = reinterpret_cast<double*>(q); q += (size_t)bcap * 8;
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

// Per-column metadata for the draw phase, staged EARLY (before the streaming loop) from the
// sweep-ordered side arrays into LDS: no dependent gather left for the last arriver.
// sh.ada bit 0 = adaV, bit 1 = the column has missing calls.
__device__ __forceinline__ void stage_marker_meta(const SweepParams& p, const DescHead& d, uint32_t nb, int tid, const SweepShared& sh)
{
    if ((uint32_t)tid < nb) {
        const uint32_t j = d.cursor + tid;
        const int ga = p.s_ga[j];
        sh.marker[tid] = p.order[j];
        sh.grp[tid] = ga & 0x0fffffff;
        sh.ada[tid] = (uint8_t)(((ga & 0x40000000) ? 1 : 0) | ((ga & 0x20000000) ? 2 : 0));
        sh.bold[tid] = p.s_bold[j];
        sh.mave[tid] = p.s_mave[j];
        sh.mstd[tid] = p.s_mstd[j];
    }
    // positions t and t + BLOCK after the cursor: bit 0 = the marker's effect is non-zero at sweep start (it WILL
    // change: a predicted event), bit 1 = its column has missing calls
    const uint32_t j0 = d.cursor + (uint32_t)tid, j1 = j0 + BLOCK;
    uint32_t f0 = 0u, f1 = 0u;
    if (j0 < p.M) f0 = (p.s_bold[j0] != 0.0 ? 1u : 0u) | ((p.s_ga[j0] & 0x20000000) ? 2u : 0u);
    if (j1 < p.M) f1 = (p.s_bold[j1] != 0.0 ? 1u : 0u) | ((p.s_ga[j1] & 0x20000000) ? 2u : 0u);
    sh.scanf[tid] = (uint8_t)f0;
    sh.scanf[tid + BLOCK] = (uint8_t)f1;
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

// Slices (workgroups per group) of a launch with nactive groups: as many as keep every active workgroup
// co-resident, at most slices_max; a workgroup then streams ceil(ntg / S) tile groups.
__device__ __forceinline__ uint32_t sweep_slices(const SweepParams& p, uint32_t nactive)
{
    const uint32_t fit = p.resident / (nactive ? nactive : 1u);
    return fit < p.slices_max ? (fit ? fit : 1u) : p.slices_max;
}

// carried prefix of a batch of nb columns (their dots were handed on by the previous launch: SweepDesc::carry_n)
__device__ __forceinline__ uint32_t sweep_carried(const SweepParams& p, const DescHead& d, uint32_t nb)
{
    return (p.carry_on && p.gram && d.pend_marker[0] >= 0) ? (d.carry_n < nb ? d.carry_n : nb) : 0u;
}

// Columns a launch streams AHEAD: while the last-arriving workgroup draws, the others take the columns that follow the batch
// from a queue (dots against this launch's residual, exactly like its fresh columns); the next launch finds them as carried
// columns behind the ones this launch left over.  In the builds that carry on the one-term Gram identity a column with
// missing calls ends the range.  flags: the staged scan flags (bit 1: missing calls) of the positions after the cursor.
template <int MG, int NOMISS>
__device__ __forceinline__ uint32_t sweep_ahead_cols(const SweepParams& p, const DescHead& d, uint32_t nb, const SweepShared& sh, int tid)
{
    const uint32_t remaining = (d.cursor < p.M) ? p.M - d.cursor : 0u;
    uint32_t n = (p.ahead_cols && p.carry_on && p.gram && !p.sums_out && remaining > nb) ? remaining - nb : 0u;
    n = n < p.ahead_cols ? n : p.ahead_cols;
    n = nb + n > 2u * BLOCK ? 2u * BLOCK - nb : n; // the flags are staged for 2 * BLOCK positions
    if (MG || NOMISS || n == 0u) return n;
    uint32_t bad = n;
    for (uint32_t a = (uint32_t)tid; a < n; a += BLOCK)
        if ((sh.scanf[nb + a] & 2u) && a < bad) bad = a;
    return block_min_u32(sh, bad, tid);
}

// groups of a launch: the update group (when updates are pending), Gram-only groups of the carried columns, fresh groups
__device__ __forceinline__ uint32_t sweep_groups(const SweepParams& p, const DescHead& d, uint32_t nb, int seg, int mg)
{
    const uint32_t ncar = sweep_carried(p, d, nb), ccg = (uint32_t)carried_cpg(seg, mg);
    return (d.pend_marker[0] >= 0 ? 1u : 0u) + (ncar + ccg - 1) / ccg + (nb - ncar + p.cols_per_group - 1) / p.cols_per_group;
}

// Posterior of the batch columns [lo, hi) of segment `seg` (a5: src/BayesRRm.cpp:1721-1723,1744-1753,1855-1921), one
// thread per (column j, step kk of the component walk): numerator of the column (dot, Gram corrections for the earlier
// pivots' updates in order, old effect's term), log-likelihoods of all components in registers, then
//     thr[j][kk] = 0 if any |logL_l - logL_kk| > 700 (l >= max(kk,1)) else 1 / sum_l exp(logL_l - logL_kk)   (:1883-1921).
// The K - 1 threads of a column repeat its numerator and divisions (bit-identical): no LDS round trip and no barrier
// between log-likelihoods and thresholds, and the divisions / exponentials of one thread are independent instruction
// streams (KT is the unrolled component count; lanes beyond K compute on neutral operands).
template <int KT, int MG, int NR>
__device__ __forceinline__ void posterior_items(const SweepParams& p, const SweepShared& sh, uint32_t lo, uint32_t hi, int seg)
{
    const int K = p.K;
    const bool lds_tab = p.GK <= HT_LDS;
    for (uint32_t it = threadIdx.x; it < (hi - lo) * (uint32_t)(K - 1); it += BLOCK) {
        const uint32_t j = lo + it / (uint32_t)(K - 1);
        const int kk = (int)(it % (uint32_t)(K - 1));
        if (!(sh.ada[j] & 1)) continue;
        const int grp = sh.grp[j];
        const double mave = sh.mave[j], mstd = sh.mstd[j];
        double num = sh.dp[j];
        for (int q = 0; q < seg; ++q) {
            double xx;
            if constexpr (MG) {
                // missing calls in either column: x_j'x_p = mstd_j mstd_p (A - m_p B - m_j C + m_j m_p D) with the integer sums
                // A = sum gw_j gw_p, B = sum gw_j nm_p, C = sum nm_j gw_p, D = sum nm_j nm_p (gw = genotype * non-missing)
                const double* r = sh.tot + NR * j + NSUM + 4 * q;
                const double mp = sh.ev[3 * q + 1];
                xx = mstd * sh.ev[3 * q + 2] * (((r[0] - mp * r[1]) - mave * r[2]) + (mave * mp) * r[3]);
            } else {
                const double A = sh.tot[NR * j + NSUM + q];
                xx = mstd * sh.ev[3 * q + 2] * (A - p.n_total * (mave * sh.ev[3 * q + 1]));
            }
            num += sh.ev[3 * q] * xx;
        }
        num += sh.bold[j] * p.n_minus_1;
        if (kk == 0) sh.numf[j] = num; // mu_k = numf / denom_k is recomputed by the drawing lane
        double L[KT];
#pragma unroll
        for (int k = 0; k < KT; ++k) {
            const bool on = k < K;
            const int t = grp * K + (on ? k : 0);
            const double lpi = lds_tab ? sh.htab[HT_LDS + t] : p.logpi[t];
            if (k == 0) {
                L[k] = lpi;
            } else {
                const double den = lds_tab ? sh.htab[t] : p.denom[t];
                const double hlg = lds_tab ? sh.htab[2 * HT_LDS + t] : p.hlog[t];
                const double mk = num / (on ? den : 1.0);
                L[k] = lpi - hlg + mk * num * p.i_2sigE;
            }
        }
        double base = L[0];
#pragma unroll
        for (int k = 1; k < KT - 1; ++k) base = (kk == k) ? L[k] : base;
        bool big = false;
        double e[KT];
#pragma unroll
        for (int l = 0; l < KT; ++l) {
            const double d = L[l] - base;
            if (l < K && l >= (kk ? kk : 1) && fabs(d) > 700.0) big = true;
            e[l] = exp(l < K ? d : 0.0);
        }
        double sum = 0.0;
#pragma unroll
        for (int l = 0; l < KT; ++l)
            if (l < K) sum += e[l];
        sh.thr[j * (K - 1) + kk] = big ? 0.0 : 1.0 / sum;
    }
}

// Posterior + draw + bookkeeping for the markers of this batch, given the reduced
// sums in sh.tot: rows [NROW*j ...] = (s1, s2, A_0, A_1, A_2) of batch column j, last
// row = sum of eps.  The batch is a chain of up to MAX_SEG segments [nbs[s-1], nbs[s]);
// segment s < last ends ON a predicted event (its pivot, position nbs[s]-1).  The dots
// of a later segment were taken before the earlier pivots' updates and are corrected
// here once those new effects are known:
//         x_j'eps_new = x_j'eps_old + sum_{q<s} dbeta_q * x_j'x_q,
//         x_j'x_q = mstd_j mstd_q (A_jq - N mave_j mave_q)   (columns without missing calls),
// with A_jq = sum_i gw_j gw_q the integer Gram term accumulated by the streaming loop.
// Runs in ONE workgroup of 256 threads.
// a5-a7: src/BayesRRm.cpp:1721-1723,1744-1753,1855-1921; dense dot algebra :1785-1790,1809.
template <int SEG, int MG, int NOMISS, int DBG>
__device__ __forceinline__ void sweep_draw_phase(const SweepParams& p, const DescHead& d, const uint32_t (&nbs)[SEG],
                                                 const SweepShared& sh)
{
    unsigned long long* const dbgp = DBG ? p.dbg : nullptr; // stage timestamps: compiled out of the production builds
    const int tid = threadIdx.x;
    constexpr int NR = sweep_rows(SEG, MG); // rows per batch column at this tier
    const int K = p.K;
    const uint32_t nb = nbs[SEG - 1];
    const uint32_t idx0 = d.rng_idx;
    const bool need_next = idx0 + MAX_BATCH + 64 > (uint32_t)MT_N; // uniform
    if (need_next) mt_next_block(sh.mt, tid);
    const bool carry_on = p.carry_on && p.gram;
    const uint32_t ncarry = sweep_carried(p, d, nb);
    int npend_in = 0; // pending updates this launch applied (what the carried dots were not yet corrected for)
#pragma unroll
    for (int q = 0; q < MAX_SEG; ++q) npend_in += d.pend_marker[q] >= 0 ? 1 : 0;

    // per-column state -> LDS, dot products from the reduced rows
    struct {
        int grp;
        bool ada, miss;
        double bold, mave, mstd;
    } mm{0, false, false, 0.0, 0.0, 0.0};
    if ((uint32_t)tid < nb) {
        mm.grp = sh.grp[tid];
        mm.ada = (sh.ada[tid] & 1) != 0;
        mm.miss = (sh.ada[tid] & 2) != 0;
        mm.bold = sh.bold[tid];
        mm.mave = sh.mave[tid];
        mm.mstd = sh.mstd[tid];
        // dense BED form of the reference (src/BayesRRm.cpp:1785-1790,1809):
        // s1 = sum c1*(c2*eps), s2 = sum c2*eps, num = mstd*(s1 - mave*s2).
        // A column without missing calls has s2 == sum of eps: it is not accumulated per column.  p.eps_sum is reduced once at
        // sweep start and held: an update adds dbeta times a standardised column, whose entries sum to zero, so the sum of eps
        // moves by rounding only.  It is the running value within that rounding, not bit for bit; the host reduces eps again at
        // sweep end and reports the gap (stats eps_sum_drift, asserted in tests; num moves by mstd*mave*drift at most).
        const double s1 = sh.tot[NR * tid];
        const double s2 = mm.miss ? sh.tot[NR * tid + 1] : p.eps_sum;
        sh.dp[tid] = mm.mstd * (s1 - mm.mave * s2);
        // a carried column: the dot the previous launch handed on (against ITS residual) plus sum_q dbeta_q x_j'x_q over the
        // updates applied since.  Row 0 holds G_j = sum_q dbeta_q mstd_j mstd_q A_jq (integer Gram terms, summed over ranks);
        // without missing calls x_j'x_q = mstd_j mstd_q (A_jq - N mave_j mave_q): the second part is added here
        // (a column the previous launch streamed ahead has no finished dot: its raw sums went into row 0 as well, and without
        // missing calls its s2 is the sum of eps, added here once for all ranks)
        if ((uint32_t)tid < ncarry) {
            double v = ((uint32_t)tid < d.carry_left ? p.carry[tid] : (mm.miss ? 0.0 : -(mm.mstd * (mm.mave * p.eps_sum)))) + s1;
            if constexpr (!MG) {
                for (int q = 0; q < npend_in; ++q)
                    v -= sh.pev[3 * q] * (mm.mstd * sh.pev[3 * q + 2] * (p.n_total * (mm.mave * sh.pev[3 * q + 1])));
            }
            sh.dp[tid] = v;
        }
    }
    if (tid == 0) {
        sh.flags[F_POS] = idx0;
        sh.flags[F_NACC] = 0;
        sh.flags[F_ERR] = 0;
    }
    __syncthreads();
    if (dbgp && tid == 0) dbgp[24] = wall_clock64(); // generator block + dots staged

    // pending updates handed to the next launch: kept in LDS by thread 0 (sh.pvl is free again: the
    // streaming loop is over), markers in sh.pmk
    if (tid < MAX_SEG) sh.pmk[tid] = -1;
    if (tid < 3 * MAX_SEG) sh.pvl[tid] = sh.pel[tid] = 0.0;
    bool ev_miss = false; // an update handed on comes from a column with missing calls (uniform)
    int npend = 0;
    unsigned long long nnz_add = 0;
    int stop_seg = -1; // segment whose walk ended on an event with a non-zero update (the last pending one)

    for (int seg = 0; seg < SEG; ++seg) {
        const uint32_t lo = seg ? nbs[seg - 1] : 0u, hi = nbs[seg];
        if (lo >= hi) break; // uniform

        // ---- posterior of [lo, hi): one thread per (column, walk step kk) ------------
        if (K <= 4) posterior_items<4, MG, NR>(p, sh, lo, hi, seg);
        else posterior_items<MAX_K, MG, NR>(p, sh, lo, hi, seg);
        __syncthreads();
        if (dbgp && tid == 0 && seg == 0) dbgp[3] = wall_clock64();
        if (dbgp && tid == 0 && seg < 2) dbgp[26 + 4 * seg] = wall_clock64(); // thresholds

        // ---- the walk: wave 0 consumes the stream in marker order, 64 at a time --
        if (tid < WAVE) {
            const int lane = tid;
            uint32_t pos = sh.flags[F_POS]; // stream position
            uint32_t naccept = 0;
            bool stopped = false;
            for (uint32_t base = lo; base < hi && !stopped; base += WAVE) {
                const uint32_t j = base + lane;
                const bool valid = j < hi;
                const bool ada = valid && (sh.ada[valid ? j : 0] & 1) != 0;
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
                    const double den = (p.GK <= HT_LDS) ? sh.htab[grp * K + k] : p.denom[(size_t)grp * K + k];
                    bnew = norm_rng_sd(g, zt, sh.numf[j] / den, sd); // the same division the posterior made
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
                        sh.ev[3 * seg] = f_dbeta;
                        sh.ev[3 * seg + 1] = sh.mave[base + f];
                        sh.ev[3 * seg + 2] = sh.mstd[base + f];
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
        if (dbgp && tid == 0 && seg < 2) dbgp[27 + 4 * seg] = wall_clock64(); // walk

        // an event: its update is pending for the next launch
        const bool stopped = sh.flags[F_STOP] != 0;
        if (stopped) {
            const double db = sh.ev[3 * seg], av = sh.ev[3 * seg + 1], sd = sh.ev[3 * seg + 2];
            if (db != 0.0) {
                if (tid == 0) {
                    sh.pmk[npend] = (int)sh.flags[F_FMARK];
                    sh.pvl[3 * npend] = -(av * sd * db);
                    sh.pvl[3 * npend + 1] = db * (1.0 - av) * sd;
                    sh.pvl[3 * npend + 2] = db * (2.0 - av) * sd;
                    sh.pel[3 * npend] = db;
                    sh.pel[3 * npend + 1] = av;
                    sh.pel[3 * npend + 2] = sd;
                }
                ev_miss = ev_miss || (sh.ada[sh.flags[F_FPOS]] & 2) != 0;
                ++npend;
                ++nnz_add;
                stop_seg = seg;
            } else {
                stop_seg = -1;
            }
        }
        // go on into the next segment only if this one ran to its end and ended ON its last column
        // (the column the later segments' Gram terms were taken with)
        if (!(stopped && sh.flags[F_FPOS] == hi - 1u && seg + 1 < SEG && nbs[seg + 1] > hi)) break;
        __syncthreads(); // sh.ev stays valid for the corrections; flags are rewritten by the next walk
    }

    // ---- hand the state to the next launch ---------------------------------------
    const uint32_t naccept = sh.flags[F_NACC];
    const uint32_t pos = sh.flags[F_POS];
    // Carried dots.  The walk ended on an event at batch position fpos with columns left behind it: their dots (sh.dp:
    // against the residual this launch streamed, i.e. with its pending updates applied) are stale only by the updates of
    // THIS launch's events, all of which are handed on as pending.  They go to the next launch as they are; it takes the
    // integer Gram terms with its pending columns instead of streaming them again.  The one-term Gram identity needs both
    // columns free of missing calls: in the builds without the four-term sums such a column, or such an event, ends the carry.
    // Behind them follow the columns this launch streams AHEAD while this phase runs (sweep_ahead_cols): usable by the next
    // launch when the carried run reaches the end of the batch (or nothing is left over).
    const uint32_t n_ahead = sweep_ahead_cols<MG, NOMISS>(p, d, nb, sh, tid);
    uint32_t carry_next = 0u, carry_left = 0u;
    {
        const uint32_t fpos = sh.flags[F_FPOS];
        const bool usable = carry_on && (MG || !ev_miss);                                 // uniform
        const bool left = sh.flags[F_STOP] != 0 && fpos + 1u < nb;                          // columns lie behind the event that ended the walk
        bool reach_end = true;
        if (usable && left) {
            const uint32_t bad = (!MG && (uint32_t)tid > fpos && (uint32_t)tid < nb && mm.miss) ? (uint32_t)tid : nb;
            const uint32_t first_bad = block_min_u32(sh, bad, tid);
            carry_left = first_bad - (fpos + 1u);
            reach_end = first_bad == nb;
            if ((uint32_t)tid > fpos && (uint32_t)tid < first_bad) p.carry[(uint32_t)tid - (fpos + 1u)] = sh.dp[tid];
        }
        carry_next = usable ? carry_left + (reach_end ? n_ahead : 0u) : 0u;
        if (!usable) carry_left = 0u;
    }
    if (dbgp && tid == 0) dbgp[23] = wall_clock64(); // segments done
    // ---- plan of the next launch (positions relative to the NEW cursor), by wave 0 ------------
    // segment 0 = up to and including the first predicted event (its pivot); every further
    // segment = up to and including the next predicted event, as long as the previous pivot's
    // column and the columns involved have no missing calls (their dots get the Gram
    // corrections above); a column with missing calls stays out and ends the chain.
    uint32_t want[MAX_SEG];
#pragma unroll
    for (int q = 0; q < MAX_SEG; ++q) want[q] = p.batch_limit;
    if (tid < WAVE) {
        const uint32_t cap = p.batch_limit;
        const uint32_t wnd = 2u * BLOCK - naccept; // flags are known for this many positions
        unsigned long long mev[MAX_BATCH / WAVE], many[MAX_BATCH / WAVE];
#pragma unroll
        for (int c = 0; c < MAX_BATCH / WAVE; ++c) {
            const uint32_t r = (uint32_t)c * WAVE + (uint32_t)tid;
            const uint32_t f = (r < wnd && r < cap) ? sh.scanf[naccept + r] : 0u;
            mev[c] = __ballot((f & 1u) != 0u);
            many[c] = __ballot(f != 0u);
        }
        auto first_set = [&](const unsigned long long (&m)[MAX_BATCH / WAVE], uint32_t from, uint32_t lim) -> uint32_t {
            for (uint32_t c = from / WAVE; c < (uint32_t)(MAX_BATCH / WAVE); ++c) {
                unsigned long long w = m[c];
                if (c == from / WAVE) w &= ~0ull << (from % WAVE);
                if (w) {
                    const uint32_t r = c * WAVE + (uint32_t)(__ffsll((long long)w) - 1);
                    return r < lim ? r : 0xffffffffu;
                }
            }
            return 0xffffffffu;
        };
        auto is_event = [&](uint32_t r) -> bool { return (mev[r / WAVE] >> (r % WAVE)) & 1ull; };
        const uint32_t e0 = first_set(mev, 0u, cap);
        if (e0 != 0xffffffffu) {
            want[0] = e0 + 1u;
            // a pivot whose own column has missing calls cannot be corrected for: (any & ~ev) cannot tell, so read its flag
            bool chain = p.gram && (MG || !(sh.scanf[naccept + e0] & 2u));
            const uint32_t lim = (want[0] + p.ext_limit < cap) ? want[0] + p.ext_limit : cap;
            for (int q = 1; q < MAX_SEG; ++q) {
                want[q] = want[q - 1];
                if (!chain || q >= SEG || (uint32_t)q >= p.max_seg || want[q - 1] >= lim) {
                    chain = false;
                    continue;
                }
                const uint32_t e = MG ? first_set(mev, want[q - 1], lim) : first_set(many, want[q - 1], lim);
                if (e == 0xffffffffu) { // no further event in reach: run to the limit, nothing can follow
                    want[q] = lim;
                    chain = false;
                } else if (!MG && (sh.scanf[naccept + e] & 2u)) { // a column with missing calls stays out of the extension
                    want[q] = e;
                    chain = false;
                } else {
                    (void)is_event;
                    want[q] = e + 1u;
                }
            }
        }
    }
    if (tid == 0) {
        SweepDesc n; // every field from registers / LDS: no load in front of the hand-over (the counters live behind it, see SweepCounters)
        n.cursor = d.cursor + naccept;
        n.cur = (d.pend_marker[0] >= 0) ? d.cur ^ 1u : d.cur;
        for (int q = 0; q < MAX_SEG; ++q) {
            n.pend_marker[q] = sh.pmk[q];
            for (int c = 0; c < 3; ++c) {
                n.pv[q][c] = sh.pvl[3 * q + c];
                n.pend_ev[q][c] = sh.pel[3 * q + c];
            }
        }
        n.rng_idx = (pos >= (uint32_t)MT_N) ? pos - (uint32_t)MT_N : pos;
        n.seq = d.seq + 1;
        n.error = sh.flags[F_ERR] ? sh.flags[F_ERR] : d.error;
        for (int q = 0; q < MAX_SEG; ++q) n.seg_end[q] = want[q];
        n.carry_n = carry_next;
        n.carry_left = carry_left;
        if (p.entered) { // every workgroup of THIS launch has read its descriptor (a late one needs only a slot: bounded wait all the same)
            const unsigned long long t0 = wall_clock64();
            for (;;) {
                uint32_t e = 0u;
                for (int x = 0; x < 8; ++x) e += __hip_atomic_load(p.entered + 32 * x, HG_RLX_AGENT);
                if (e >= gridDim.x) break;
                if (wall_clock64() - t0 > 300000000ull) {
                    n.error = 3u;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
            for (int x = 0; x < 8; ++x) __hip_atomic_store(p.entered + 32 * x, 0u, HG_RLX_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        *p.desc = n;
        { // the sweep's counters: a read-modify-write of their own, off the hand-over's path
            SweepCounters c = *p.counters;
            c.nnz += nnz_add;
            c.launches += 1;
            c.accepted_sum += naccept;
            c.carried_sum += carry_next;
            c.streamed_sum += nb - ncarry + n_ahead;
            if (nb > ncarry) {
                const uint32_t ntg = p.n_pad / BLOCK_IND, S = sweep_slices(p, sweep_groups(p, d, nb, SEG, MG));
                const uint32_t tiles = (ntg + S - 1) / S;
                c.tiles_min = tiles < c.tiles_min ? tiles : c.tiles_min;
                c.tiles_max = tiles > c.tiles_max ? tiles : c.tiles_max;
            }
            *p.counters = c;
        }
        if (dbgp) { // accumulate stage durations over all launches: [8+i] += t[i+1]-t[i], [15] = count
            dbgp[4] = wall_clock64();
            for (int i = 0; i < 4; ++i) dbgp[8 + i] += dbgp[i + 1] - dbgp[i];
            dbgp[13] += dbgp[6] - dbgp[5]; // last arriver: entry -> loop done
            dbgp[14] += dbgp[7] - dbgp[6]; // last arriver: loop done -> drained
            dbgp[16] += dbgp[1] - dbgp[7]; // last arriver: drained -> past ticket
            dbgp[17] += dbgp[5] - dbgp[0]; // first block entry -> last arriver entry
            dbgp[12] += naccept;
            dbgp[15] += 1;
            // finer stages of the draw phase: [36] staging, [38] seg 0 posterior (numerators + thresholds), [39] seg 0 walk,
            // [41], [42] the same for segment 1 (when it ran), [43] plan + descriptor, [44] launches with a second segment
            dbgp[36] += dbgp[24] - dbgp[2];
            dbgp[38] += dbgp[26] - dbgp[24];
            dbgp[39] += dbgp[27] - dbgp[26];
            if (dbgp[30] > dbgp[27]) {
                dbgp[41] += dbgp[30] - dbgp[27];
                dbgp[42] += dbgp[31] - dbgp[30];
                dbgp[44] += 1;
            }
            dbgp[43] += dbgp[4] - dbgp[23];
            dbgp[30] = 0;
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

// Integer Gram terms on the device codes.  A weight dword holds 16 genotypes g in {0,1,2} as 2-bit fields (b1 b0).  Its
// "x form" keeps b1 and puts u = [g >= 1] = b0 | b1 in the even bit: g = u + v with v = [g == 2] = b1, so
//     sum_i g_a g_p = popc(x_a & UU_p) + popc(x_a & VV_p),
// UU_p / VV_p being u_p / v_p replicated into both bits of every field: two ANDs and two accumulating popcounts per
// (column, pivot) and dword, one more AND-OR + shift per column for x_a, a handful per pivot and tile for UU, VV.
__device__ __forceinline__ uint32_t gram_xform(uint32_t gw) { return gw | ((gw >> 1) & 0x55555555u); }
struct GramPivot {
    uint32_t uu, vv;
};
__device__ __forceinline__ GramPivot gram_pivot(uint32_t gw)
{
    const uint32_t ue = (gw | (gw >> 1)) & 0x55555555u, vo = gw & 0xAAAAAAAAu;
    return GramPivot{ue | (ue << 1), vo | (vo >> 1)};
}
__device__ __forceinline__ uint32_t gram16x(uint32_t xa, const GramPivot& p) { return (uint32_t)__popc(xa & p.uu) + (uint32_t)__popc(xa & p.vv); }
// the same from two weight dwords (tests of the identity, single-marker helpers)
__device__ __forceinline__ uint32_t gram16(uint32_t ga, uint32_t gb) { return gram16x(gram_xform(ga), gram_pivot(gb)); }

// Cross-GPU sum of the batch rows held in sh.tot, inside the launch: push my
// rows into every rank's mailbox (system-scope stores over xGMI), publish one
// flag per destination, wait for the nranks flags in my own mailbox, then add
// the contributions in RANK ORDER so that every GPU gets the same bits.  Two
// parities: a peer can be at most one batch ahead (it needs my rows of batch
// b+1 before it can finish b+1).  Bounded spin: returns false on timeout.
template <int NR>
__device__ __forceinline__ bool p2p_exchange(const SweepParams& p, const DescHead& d, uint32_t nb, const SweepShared& sh)
{
    const int tid = threadIdx.x;
    const int nr = p.p2p.nranks, me = p.p2p.rank;
    const uint32_t nrows = NR * nb;
    const uint32_t parity = (uint32_t)(d.seq & 1ull);
    const unsigned long long epoch = d.seq + 1ull;
    const size_t slot = (size_t)(parity * MAX_RANKS + (uint32_t)me) * ROWS_CAP;
    for (uint32_t it = tid; it < nrows * (uint32_t)nr; it += BLOCK) {
        const uint32_t dst = it / nrows, rr = it % nrows;
        const double v = sh.tot[rr];
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
    // my mailbox is uncached memory: every load is a full round trip, so all ranks' values of two rows are in flight at once;
    // added in rank order
    for (uint32_t rr0 = tid; rr0 < nrows; rr0 += 2 * BLOCK) {
        double v[2][MAX_RANKS];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const uint32_t rr = rr0 + (uint32_t)i * BLOCK;
#pragma unroll
            for (int r = 0; r < MAX_RANKS; ++r)
                v[i][r] = (rr < nrows && r < nr)
                              ? __hip_atomic_load(p.p2p.data[me] + (size_t)(parity * MAX_RANKS + (uint32_t)r) * ROWS_CAP + rr, HG_RLX_SYSTEM)
                              : 0.0;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const uint32_t rr = rr0 + (uint32_t)i * BLOCK;
            double acc = 0.0;
#pragma unroll
            for (int r = 0; r < MAX_RANKS; ++r)
                if (r < nr) acc += v[i][r];
            if (rr < nrows) sh.tot[rr] = acc;
        }
    }
    __syncthreads();
    return true;
}

// 64-lane integer sum on the DPP path (the Gram partials): total returned wave-uniform
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);  // quad_perm [1,0,3,2]
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false);  // quad_perm [2,3,0,1]
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, false); // row_half_mirror
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, false); // row_mirror
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false); // row_bcast15
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false); // row_bcast31
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt()
{
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field on gfx9");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// One launch of the sweep.  The batch is [carried columns | fresh columns]: the first `carry_n` columns were streamed by
// the previous launch (they lay behind the event that ended it), their dots against THAT launch's residual are in
// SweepParams::carry.  A 1-D grid of S * nactive workgroups; a group of S workgroups (slices: strided sets of tile
// groups of 4096 individuals) is one of
//   the update group   (one, when updates are pending) applies the pending updates of the previous launch's events to
//                      eps and stores the other eps buffer -- nothing else, so that no other workgroup stores;
//   Gram-only groups   CCG carried columns each: no eps, no LDS, only the integer Gram terms x_j'x_q with every pending
//                      column q (the updates their dots do not contain yet) and with this launch's earlier pivots;
//   fresh groups       CPG columns each: the pending updates on the eps tile in registers, then s1 = sum g nm eps (and
//                      s2 for a column with missing calls), plus the Gram terms with this launch's earlier pivots.
// Every lane keeps its sums in registers across all its tiles; column dwords are prefetched TWO tiles ahead (the count
// of loads per tile is a compile-time constant, so the loop waits on vmcnt(loads of one tile) instead of vmcnt(0)); one
// wave / block reduction per launch, per-slice partial rows, the last of a group's S workgroups sums them over the slices
// (for carried columns: straight into G_j = sum_q dbeta_q mstd_j mstd_q A_jq), the last group runs the draw phase.
// NOMISS: the data has no missing call in any column (known from the marker statistics): the second masked sum and its
// reductions are compiled out (s2 is the sum of eps for every column).
// DBG: stage timestamps (option debug_timing); the production builds carry none of it.
template <int CPG, int SEG, int MG, int NOMISS = 0, int DBG = 0>
__global__ __launch_bounds__(BLOCK, ((CPG <= 4 || (CPG <= 8 && SEG <= 2)) ? 3 : 2)) void k_sweep_batch(SweepParams p)
{
    unsigned long long* const dbgp = DBG ? p.dbg : nullptr;
    static_assert(!(MG && NOMISS), "the missing-call Gram build is for data with missing calls");
    static_assert(!MG || SEG == 2, "the missing-call Gram terms are carried by the two-segment build only");
    constexpr int NR = sweep_rows(SEG, MG);     // rows per batch column in `totals`: s1 (or G_j), s2, T per earlier pivot
    constexpr int T = MG ? 4 : 1;               // integer sums per Gram term
    constexpr int CCG = carried_cpg(SEG, MG);   // carried columns per Gram-only workgroup
    constexpr int NRC = carried_rows(SEG, MG);  // rows a Gram-only group publishes per column: T per pending update, T per pivot
    constexpr int RB = group_rows(CPG, SEG, MG); // row block of one group in `partials`
    const SweepShared sh = sweep_lds_carve(hg_smem, p.batch_cap, p.cols_per_group, p.K, NR);
    const DescHead d = load_desc_head(p.desc);
    // this workgroup has the launch's descriptor.  (The value added depends on loaded fields -- both terms are zero, which the
    // compiler cannot know -- so the add cannot be issued before the descriptor's loads have returned.)
    if (threadIdx.x == 0) __hip_atomic_fetch_add(p.entered + 32u * (blockIdx.x & 7u), 1u + (d.error >> 31) + (uint32_t)(d.seq >> 63), HG_RLX_AGENT);
    const bool pend = d.pend_marker[0] >= 0;
    const uint32_t remaining = (d.cursor < p.M) ? p.M - d.cursor : 0u;
    uint32_t nbs[SEG]; // cumulative ends of this launch's segments
#pragma unroll
    for (int q = 0; q < SEG; ++q) {
        const uint32_t e = d.seg_end[q] < remaining ? d.seg_end[q] : remaining;
        nbs[q] = (q && e < nbs[q - 1]) ? nbs[q - 1] : e;
    }
    const uint32_t nb = nbs[SEG - 1]; // all columns of this launch
    if ((nb == 0 && !pend) || d.error) return; // whole grid agrees: nothing left to do
    if (lds_addr(hg_smem) & 255u) { // the update tables are addressed by OR: the dynamic LDS must start on a 256-byte boundary (it starts at 0)
        if (blockIdx.x == 0 && threadIdx.x == 0) p.desc->error = 4u;
        return;
    }

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t voff = (uint32_t)lane << 2; // the lane's dword inside a 256-byte column piece; bases stay wave-uniform
    const uint32_t ncar = sweep_carried(p, d, nb);
    const uint32_t nu = pend ? 1u : 0u, ncg = (ncar + CCG - 1) / CCG, nfg = (nb - ncar + CPG - 1) / CPG;
    const uint32_t nactive = nu + ncg + nfg; // >= 1: pending updates or columns
    // slices: the active workgroups (the first S * nactive in dispatch order; idle ones behind them leave at once) must be
    // co-resident -- a second round of workgroups would double the streaming phase
    const uint32_t S = sweep_slices(p, nactive);
    if (blockIdx.x >= S * nactive) return;
    const uint32_t group = blockIdx.x / S;
    // the work item of this workgroup: first its part of the batch (slice `slice` of group `group`), then -- behind the hand-off,
    // while the last arriver draws -- items of the ahead queue (a fresh group of columns behind the batch, `ahead` set)
    uint32_t slice = blockIdx.x % S, Sw = S; // slice of Sw of the current item
    int kind = group < nu ? 0 : (group < nu + ncg ? 1 : 2); // update / Gram-only / fresh
    uint32_t c0 = kind == 1 ? (group - nu) * CCG : ncar + (kind == 2 ? (group - nu - ncg) * CPG : 0u);
    const uint32_t cend = kind == 1 ? ncar : nb;
    uint32_t c1 = kind == 0 ? c0 : ((c0 + (kind == 1 ? CCG : CPG) < cend) ? c0 + (kind == 1 ? CCG : CPG) : cend);
    uint32_t ncol = c1 - c0;
    const uint32_t rows_per_col = kind == 1 ? NRC : NR;
    const double* eps_in = d.cur ? p.eps1 : p.eps0;
    double* eps_out = d.cur ? p.eps0 : p.eps1;
    const uint32_t ntg = p.n_pad / BLOCK_IND; // tile groups
    uint32_t nt = slice < ntg ? (ntg - slice + Sw - 1) / Sw : 0u; // tile groups of the current item
    auto tile_at = [&](uint32_t k) { // the k-th tile of this wave; past the end: the last one again (loads stay in range, results unused)
        const uint32_t kk = k < nt ? k : (nt ? nt - 1 : 0u);
        return (slice + kk * Sw) * BLOCK_WAVES + wave;
    };
    uint32_t n_ahead = 0u, a_slices = 1u, a_groups = 0u; // ahead range of this launch (items of its queue: a group of columns x a slice)
    const uint32_t apar = (uint32_t)(d.seq & 1ull);                    // this launch's half of the ahead buffers and queue counters
    __builtin_amdgcn_s_setprio(2);

    if (pend && tid < 16 * SEG) { // entry (c1 << 2 | c0) of pending update q: the addends of two neighbouring individuals
        const int q = tid >> 4;
        const double* pv = p.desc->pv[q];
        auto addend = [&](uint32_t c) { return 0.0 + ((c == GC_G0) ? pv[0] : ((c == GC_G1) ? pv[1] : ((c == GC_G2) ? pv[2] : 0.0))); };
        sh.pvt[tid] = make_double2(addend((uint32_t)tid & 3u), addend(((uint32_t)tid >> 2) & 3u));
    }
    if (dbgp && blockIdx.x == 0 && tid == 0) dbgp[0] = wall_clock64();
    const unsigned long long t_entry = dbgp ? wall_clock64() : 0ull;
    unsigned long long t_loop = 0ull;

    int npend = 0;
#pragma unroll
    for (int q = 0; q < SEG; ++q) npend += d.pend_marker[q] >= 0 ? 1 : 0;
    // columns a lane's loads go to when a slot is not in use: any valid column (the loads keep the per-tile count constant)
    const uint8_t* const anyp = p.bed + (size_t)(pend ? d.pend_marker[0] : (nb ? p.order[d.cursor] : 0)) * p.stride;
    const uint8_t* pendp[SEG];
#pragma unroll
    for (int q = 0; q < SEG; ++q) pendp[q] = q < npend ? p.bed + (size_t)d.pend_marker[q] * p.stride : anyp;
    double* wbase = sh.wpart;       // per-wave partial rows of this workgroup: [BLOCK_WAVES][wstr]
    uint32_t wstr = sh.wstride;

    // lane mask of valid slots for a tile that reaches into the padding behind the shard's last individual (data without
    // missing calls holds the missing code only there: everywhere else a dword is used as loaded)
    auto pad_keep = [&](uint32_t tile, bool& pad_tile) -> uint32_t {
        pad_tile = NOMISS && (tile + 1u) * (uint32_t)TILE > p.n_local; // wave-uniform
        if (!pad_tile) return 0xffffffffu;
        const uint32_t i0 = tile * (uint32_t)TILE + ((uint32_t)lane << 4);
        const uint32_t nv = i0 >= p.n_local ? 0u : (p.n_local - i0 >= 16u ? 16u : p.n_local - i0);
        return nv >= 16u ? 0xffffffffu : ((1u << (2u * nv)) - 1u);
    };

    // draw-phase state every workgroup stages before its loop (any of them may turn out to be the last arriver)
    auto stage_all = [&]() {
        if (!p.sums_out) stage_marker_meta(p, d, nb, tid, sh);
        if (!p.sums_out) stage_rng(p, sh, tid);
        if (tid < 3 * MAX_SEG) sh.pev[tid] = pend ? p.desc->pend_ev[tid / 3][tid % 3] : 0.0;
    };

    // ---- a fresh group: a4 (src/BayesRRm.cpp:1766-1809) of CPG columns [c0, c1) (batch positions) against eps + pending updates,
    // slice `slice` of Sw; instantiated for the batch's own groups and, without the Gram terms, for the items of the ahead queue
    auto fresh_item = [&](auto ahead_tag) __attribute__((always_inline)) {
        constexpr bool AHEAD = decltype(ahead_tag)::value;
        constexpr int NPV = AHEAD ? 0 : SEG - 1;  // pivot columns loaded per tile (an ahead item takes plain dots)
        constexpr int NLF = CPG + NPV + SEG;      // loads per tile besides the eps tile's LDS-DMA
        double a1[CPG], a2[CPG];
        // integer Gram partials of a column of segment s with the pivots of segments 0..s-1, as 16-bit fields:
        // ag01 = pivot 0 | pivot 1 << 16, ag2 = pivot 2; with missing calls (MG): ag01 = A | B << 16, ag2 = C | D << 16
        uint32_t ag01[CPG], ag2[CPG];
#pragma unroll
        for (int c = 0; c < CPG; ++c) {
            a1[c] = a2[c] = 0.0;
            ag01[c] = ag2[c] = 0u;
        }
        const uint8_t* colp[CPG];
        bool cmiss[CPG]; // wave-uniform: column has missing calls -> needs its own s2
        int cseg[CPG];   // wave-uniform: segment of the column = number of earlier pivots its dot is corrected for
        int ng = 0;      // Gram terms this workgroup needs (segment of its last live column)
#pragma unroll
        for (int c = 0; c < CPG; ++c) {
            const bool live = c0 + c < c1;
            const uint32_t j = live ? c0 + c : c1 - 1; // ncol >= 1 in a fresh group
            colp[c] = p.bed + (size_t)p.order[d.cursor + j] * p.stride;
            cmiss[c] = !NOMISS ? ((p.s_ga[d.cursor + j] & 0x20000000) != 0) : false;
            int sg = 0;
#pragma unroll
            for (int q = 0; q < SEG - 1; ++q) sg += (!AHEAD && live && c0 + c >= nbs[q]) ? 1 : 0;
            cseg[c] = sg;
            ng = sg > ng ? sg : ng;
        }
        const bool any_gram = ng > 0;
        const uint8_t* pivp[SEG - 1];
#pragma unroll
        for (int q = 0; q < SEG - 1; ++q) pivp[q] = (q < ng && nbs[q] > 0) ? p.bed + (size_t)p.order[d.cursor + nbs[q] - 1] * p.stride : anyp;

        __syncthreads(); // sh.pvt is staged; an ahead item: the previous item's LDS reads are over
        // eps tiles arrive by LDS-DMA (global_load_lds_dwordx4: no VGPRs, lane-linear 1 KiB pieces -- exactly the permuted
        // eps layout) one tile ahead of the arithmetic; the column dwords two tiles ahead, in registers
        unsigned char* const est = sh.estage + (size_t)wave * (TILE * sizeof(double));
        const uint32_t est_addr = lds_addr(est), pvt_addr = lds_addr(sh.pvt); // (a multiple of 256: the dynamic LDS starts at 0, the table is first)
        auto dma_eps = [&](uint32_t tile) {
            // eight 1 KiB pieces: the instruction's immediate offset (<= 4095) moves the global AND the LDS address, so two base
            // addresses (and two M0 values) serve four pieces each
            const double* g = eps_in + ((size_t)tile << 10) + (lane << 1);
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const auto gp = (const __attribute__((address_space(1))) void*)(g + h2 * 512);
                const auto lp = (__attribute__((address_space(3))) void*)(est + h2 * 4096);
                __builtin_amdgcn_global_load_lds(gp, lp, 16, 0, 0);
                __builtin_amdgcn_global_load_lds(gp, lp, 16, 1024, 0);
                __builtin_amdgcn_global_load_lds(gp, lp, 16, 2048, 0);
                __builtin_amdgcn_global_load_lds(gp, lp, 16, 3072, 0);
            }
        };
        // three register sets of column dwords ([0, CPG) columns, then SEG - 1 pivots, then SEG pending columns) rotate through
        // the roles "tile k" / "tile k + 1, in flight" / "free: receives tile k + 2" without a register copy (a copy would wait
        // for the loads it moves): the loop body is instantiated once per role assignment
        uint32_t wA[NLF], wB[NLF], wC[NLF];
        // (plain loads: the compiler tracks them and places exact vmcnt waits in straight-line code; for the two sets that are in
        // flight across the loop's back edge it falls back to vmcnt(0) at their first use, i.e. one body in three drains the
        // loads of the tile after it -- loads the compiler does not see would avoid that, but a register copy the allocator
        // may place at the back edge would then read a register whose load has not landed)
        auto loads = [&](uint32_t tile, uint32_t (&dst)[NLF]) {
            const uint32_t off = (tile << 8) + voff; // 32 bits (a column is n_pad / 4 < 2^30 bytes): scalar base + vector offset addressing
#pragma unroll
            for (int c = 0; c < CPG; ++c) dst[c] = *reinterpret_cast<const uint32_t*>(colp[c] + off);
#pragma unroll
            for (int q = 0; q < NPV; ++q) dst[CPG + q] = *reinterpret_cast<const uint32_t*>(pivp[q] + off);
#pragma unroll
            for (int q = 0; q < SEG; ++q) dst[CPG + NPV + q] = *reinterpret_cast<const uint32_t*>(pendp[q] + off);
        };
        if (nt) { // issue order: eps tile 0, columns of tile 0, columns of tile 1
            dma_eps(tile_at(0));
            loads(tile_at(0), wA);
            loads(tile_at(1), wB);
        }
        // latency-bound loads of the draw phase, issued by EVERY workgroup before the streaming loop -- but behind the first
        // tiles' loads, so that their load -> LDS round trips overlap with the column bytes' way from HBM
        if (!AHEAD) stage_all();
        auto body = [&](uint32_t k, uint32_t (&w)[NLF], uint32_t (&wfree)[NLF]) __attribute__((always_inline)) {
            const uint32_t tile = tile_at(k);
            wait_vmcnt<NLF>(); // everything but the most recent tile's column loads has landed: eps tile k, columns of tile k
            double e[IPT];
            lds_eps16(est_addr, lane, e);
            if (k + 1 < nt) { // issue order: eps tile k + 1, columns of tile k + 2
                dma_eps(tile_at(k + 1));
                loads(tile_at(k + 2), wfree);
            }
            bool pad_tile;
            const uint32_t keep = pad_keep(tile, pad_tile);
            auto weights = [&](uint32_t wd, uint32_t& gwd, uint32_t& nmd) {
                if constexpr (NOMISS) {
                    gwd = pad_tile ? (wd & keep) : wd;
                    nmd = 0x55555555u;
                } else {
                    code_weights(wd, gwd, nmd);
                }
            };
            GramPivot gpv[SEG - 1];
            uint32_t nmp[SEG - 1], xp0 = 0u; // MG: non-missing mask of the pivot, x form of pivot 0
#pragma unroll
            for (int q = 0; q < SEG - 1; ++q) {
                gpv[q] = GramPivot{0u, 0u};
                nmp[q] = 0u;
                if (!AHEAD && q < ng) {
                    uint32_t gwq;
                    weights(w[CPG + (AHEAD ? 0 : q)], gwq, nmp[q]);
                    gpv[q] = gram_pivot(gwq);
                    if (MG && q == 0) xp0 = gram_xform(gwq);
                }
            }
            if (pend) { // the previous launch's event(s), in order
#pragma unroll
                for (int q = 0; q < SEG; ++q)
                    if (q < npend) apply_update16_asm(w[CPG + NPV + q], pvt_addr + 256u * (uint32_t)q, e);
            }
            uint32_t gw[CPG], nm[CPG];
#pragma unroll
            for (int c = 0; c < CPG; ++c) weights(w[c], gw[c], nm[c]);
            // s1 += (g*nm) * eps: weight 0/1/2 is exact, one rounding per add; each column adds its slots in increasing order
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
                if constexpr (!NOMISS) {
                    if (cmiss[c]) {
#pragma unroll
                        for (int s = 0; s < IPT; ++s) a2[c] = __builtin_fma((double)((nm[c] >> (2 * s)) & 1u), e[s], a2[c]);
                    }
                }
                if (!AHEAD && cseg[c] > 0) {
                    const uint32_t xc = gram_xform(gw[c]);
                    uint32_t g = gram16x(xc, gpv[0]);
                    if constexpr (MG) {
                        // B = sum gw_j nm_p, C = sum nm_j gw_p, D = sum nm_j nm_p: popcounts against masks replicated into both bits
                        const uint32_t nmp2 = nmp[0] | (nmp[0] << 1), nmj2 = nm[c] | (nm[c] << 1);
                        g |= (uint32_t)__popc(xc & nmp2) << 16;
                        ag2[c] += (uint32_t)__popc(nmj2 & xp0) | ((uint32_t)__popc(nm[c] & nmp[0]) << 16);
                    }
                    if constexpr (SEG > 2) {
                        if (cseg[c] > 1) g |= gram16x(xc, gpv[1]) << 16;
                        if (cseg[c] > 2) ag2[c] += gram16x(xc, gpv[SEG > 3 ? 2 : 0]);
                    }
                    ag01[c] += g;
                }
            }
        };
        for (uint32_t k = 0; k < nt; k += 3) {
            body(k, wA, wC);
            if (k + 1 < nt) body(k + 1, wB, wA);
            if (k + 2 < nt) body(k + 2, wC, wB);
        }
        wait_vmcnt<0>();
        __syncthreads(); // every wave is done with its staging tile: the union region may be reused
        t_loop = dbgp ? wall_clock64() : 0ull;
        // one cross-lane reduction per launch
#pragma unroll
        for (int c = 0; c < CPG; ++c) {
            const double t1 = wave_sum(a1[c]), t2 = NOMISS ? 0.0 : wave_sum(a2[c]);
            const uint32_t g0 = any_gram ? wave_sum_u32(ag01[c] & 0xffffu) : 0u;
            const uint32_t g1 = (MG ? any_gram : ng > 1) ? wave_sum_u32(ag01[c] >> 16) : 0u;
            const uint32_t g2 = (MG ? any_gram : ng > 2) ? wave_sum_u32(ag2[c] & 0xffffu) : 0u;
            const uint32_t g3 = (MG && any_gram) ? wave_sum_u32(ag2[c] >> 16) : 0u;
            if (lane == 0) {
                double* wp_ = wbase + wave * wstr + NR * c;
                wp_[0] = t1;
                wp_[1] = t2;
                if constexpr (SEG > 1) wp_[2] = (double)g0; // exact: integers far below 2^53
                if constexpr (MG) {
                    wp_[3] = (double)g1;
                    wp_[4] = (double)g2;
                    wp_[5] = (double)g3;
                } else {
                    if constexpr (SEG > 2) wp_[3] = (double)g1;
                    if constexpr (SEG > 3) wp_[4] = (double)g2;
                }
            }
        }
    };

    if (kind == 0) {
        // ---- the update group: eps_out = eps_in + pending updates, a8 (src/BayesRRm.cpp:1976-2010,2022,2471) ---------------
        __syncthreads(); // sh.pvt is staged
        unsigned char* const est = sh.estage + (size_t)wave * (TILE * sizeof(double));
        uint32_t wp[SEG], wpn[SEG];
        auto loads = [&](uint32_t tile, uint32_t (&dst)[SEG]) {
            const double* g = eps_in + ((size_t)tile << 10) + (lane << 1);
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) { // (immediate offsets move the global and the LDS address alike: see the fresh groups)
                const auto gp = (const __attribute__((address_space(1))) void*)(g + h2 * 512);
                const auto lp = (__attribute__((address_space(3))) void*)(est + h2 * 4096);
                __builtin_amdgcn_global_load_lds(gp, lp, 16, 0, 0);
                __builtin_amdgcn_global_load_lds(gp, lp, 16, 1024, 0);
                __builtin_amdgcn_global_load_lds(gp, lp, 16, 2048, 0);
                __builtin_amdgcn_global_load_lds(gp, lp, 16, 3072, 0);
            }
            const uint32_t off = (tile << 8) + voff;
#pragma unroll
            for (int q = 0; q < SEG; ++q) dst[q] = *reinterpret_cast<const uint32_t*>(pendp[q] + off);
        };
        if (nt) loads(tile_at(0), wp);
        stage_all();
        for (uint32_t k = 0; k < nt; ++k) {
            const uint32_t tile = tile_at(k);
            wait_vmcnt<0>();
            double e[IPT];
            {
                const double2* lp = reinterpret_cast<const double2*>(est) + lane;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const double2 v = lp[i * 64];
                    e[2 * i] = v.x;
                    e[2 * i + 1] = v.y;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // eps is in registers: the staging tile may be overwritten
            if (k + 1 < nt) loads(tile_at(k + 1), wpn);
#pragma unroll
            for (int q = 0; q < SEG; ++q)
                if (q < npend) apply_update16_lds(wp[q], sh.pvt + 16 * q, e);
            store_eps16(eps_out, tile, lane, e);
#pragma unroll
            for (int q = 0; q < SEG; ++q) wp[q] = wpn[q];
        }
        wait_vmcnt<0>();
        __syncthreads();
        t_loop = dbgp ? wall_clock64() : 0ull;
    } else if (kind == 1) {
        // ---- a Gram-only group: integer Gram terms of CCG carried columns --------------------------------------------------
        constexpr int NLG = CCG + SEG + (SEG - 1); // loads per tile
        wbase = reinterpret_cast<double*>(sh.estage); // no eps tiles here: the staging region holds this group's wave partials
        wstr = CCG * NRC;
        const uint8_t* colp[CCG];
        uint32_t segmask = 0u; // 2 bits per column: its segment = number of earlier pivots of THIS launch its dot is corrected for
        int ng = 0;
#pragma unroll
        for (int c = 0; c < CCG; ++c) {
            const bool live = c0 + c < c1;
            colp[c] = live ? p.bed + (size_t)p.order[d.cursor + c0 + c] * p.stride : anyp;
            int sg = 0;
#pragma unroll
            for (int q = 0; q < SEG - 1; ++q) sg += (live && c0 + c >= nbs[q]) ? 1 : 0;
            segmask |= (uint32_t)sg << (2 * c);
            ng = sg > ng ? sg : ng;
        }
        const uint8_t* pivp[SEG - 1];
#pragma unroll
        for (int q = 0; q < SEG - 1; ++q) pivp[q] = (q < ng && nbs[q] > 0) ? p.bed + (size_t)p.order[d.cursor + nbs[q] - 1] * p.stride : anyp;
        uint32_t wA[NLG], wB[NLG], wC[NLG]; // three rotating register sets, as in the fresh groups
        auto loads = [&](uint32_t tile, uint32_t (&dst)[NLG]) {
            const uint32_t off = (tile << 8) + voff; // 32 bits (a column is n_pad / 4 < 2^30 bytes): scalar base + vector offset addressing
#pragma unroll
            for (int c = 0; c < CCG; ++c) dst[c] = *reinterpret_cast<const uint32_t*>(colp[c] + off);
#pragma unroll
            for (int q = 0; q < SEG; ++q) dst[CCG + q] = *reinterpret_cast<const uint32_t*>(pendp[q] + off);
#pragma unroll
            for (int q = 0; q < SEG - 1; ++q) dst[CCG + SEG + q] = *reinterpret_cast<const uint32_t*>(pivp[q] + off);
        };
        // 16-bit fields (a lane adds at most 64 per tile and term; the host keeps tiles per lane below 1000)
        constexpr int NF = NRC;                  // fields per column: T per pending update, then T per pivot
        constexpr int NREG = (NF + 1) / 2;
        uint32_t acc[CCG][NREG];
#pragma unroll
        for (int c = 0; c < CCG; ++c)
#pragma unroll
            for (int r = 0; r < NREG; ++r) acc[c][r] = 0u;
        if (nt) {
            loads(tile_at(0), wA);
            loads(tile_at(1), wB);
        }
        stage_all();
        auto body = [&](uint32_t k, uint32_t (&w)[NLG], uint32_t (&wfree)[NLG]) __attribute__((always_inline)) {
            wait_vmcnt<NLG>(); // all but the most recent tile's loads have landed: w is complete
            if (k + 1 < nt) loads(tile_at(k + 2), wfree);
            bool pad_tile;
            const uint32_t keep = pad_keep(tile_at(k), pad_tile);
            auto weights = [&](uint32_t wd, uint32_t& gwd, uint32_t& nmd) {
                if constexpr (NOMISS) {
                    gwd = pad_tile ? (wd & keep) : wd;
                    nmd = 0x55555555u;
                } else {
                    code_weights(wd, gwd, nmd);
                }
            };
            // the columns the terms are taken with: pending updates 0..npend-1, then this launch's pivots
            GramPivot gq[SEG + SEG - 1];
            uint32_t xq[SEG + SEG - 1], nmq[SEG + SEG - 1];
#pragma unroll
            for (int q = 0; q < SEG + SEG - 1; ++q) {
                uint32_t gwq;
                weights(w[CCG + q], gwq, nmq[q]);
                gq[q] = gram_pivot(gwq);
                xq[q] = MG ? gram_xform(gwq) : 0u;
            }
#pragma unroll
            for (int c = 0; c < CCG; ++c) {
                uint32_t gwc, nmc;
                weights(w[c], gwc, nmc);
                const uint32_t xc = gram_xform(gwc);
                const int sg = (int)((segmask >> (2 * c)) & 3u);
                [[maybe_unused]] const uint32_t nmc2 = nmc | (nmc << 1);
#pragma unroll
                for (int q = 0; q < SEG + SEG - 1; ++q) {
                    const bool on = q < SEG ? q < npend : (q - SEG) < sg; // wave-uniform
                    if (!on) continue;
                    if constexpr (MG) {
                        // A = sum gw_j gw_q, B = sum gw_j nm_q, C = sum nm_j gw_q, D = sum nm_j nm_q
                        const uint32_t nmq2 = nmq[q] | (nmq[q] << 1);
                        acc[c][2 * q] += gram16x(xc, gq[q]) | ((uint32_t)__popc(xc & nmq2) << 16);
                        acc[c][2 * q + 1] += (uint32_t)__popc(nmc2 & xq[q]) | ((uint32_t)__popc(nmc & nmq[q]) << 16);
                    } else {
                        acc[c][q >> 1] += gram16x(xc, gq[q]) << (16 * (q & 1));
                    }
                }
            }
        };
        for (uint32_t k = 0; k < nt; k += 3) {
            body(k, wA, wC);
            if (k + 1 < nt) body(k + 1, wB, wA);
            if (k + 2 < nt) body(k + 2, wC, wB);
        }
        wait_vmcnt<0>();
        __syncthreads(); // the staging region is free (nobody uses it in this group) and every wave is past its loads
        t_loop = dbgp ? wall_clock64() : 0ull;
        // one cross-lane reduction per launch and field in use (a term is in use when its pending update / pivot exists: wave-uniform);
        // a lane holds at most 64 per tile and field, so with fewer than 16 tiles the two 16-bit fields of a register are summed
        // over the 64 lanes together (64 * 64 * 15 < 2^16), else one by one
        const bool packed = nt < 16u;
#pragma unroll
        for (int c = 0; c < CCG; ++c) {
            const int sg = (int)((segmask >> (2 * c)) & 3u);
#pragma unroll
            for (int r = 0; r < NREG; ++r) {
                // fields 2r, 2r + 1 belong to term (2r) / T and (2r + 1) / T
                const int q0 = (2 * r) / T, q1 = (2 * r + 1) / T;
                const bool on0 = q0 < SEG ? q0 < npend : (q0 - SEG) < sg, on1 = (2 * r + 1 < NF) && (q1 < SEG ? q1 < npend : (q1 - SEG) < sg);
                uint32_t lo = 0u, hi = 0u;
                if (on0 || on1) {
                    if (packed) {
                        const uint32_t v = wave_sum_u32(acc[c][r]);
                        lo = v & 0xffffu;
                        hi = v >> 16;
                    } else {
                        lo = on0 ? wave_sum_u32(acc[c][r] & 0xffffu) : 0u;
                        hi = on1 ? wave_sum_u32(acc[c][r] >> 16) : 0u;
                    }
                }
                if (lane == 0) { // exact: integers far below 2^53
                    wbase[wave * wstr + NRC * c + 2 * r] = (double)lo;
                    if (2 * r + 1 < NF) wbase[wave * wstr + NRC * c + 2 * r + 1] = (double)hi;
                }
            }
        }
    } else {
        fresh_item(std::false_type{});
    }
    __syncthreads();

    {
    // block partial = waves 0..3 in order, published write-through (sc1); this group's row block starts at group * RB
    const uint32_t rb = group * (uint32_t)RB;
    const uint32_t nrowg = rows_per_col * ncol;
    for (uint32_t t = tid; t < nrowg; t += BLOCK) {
        double v = wbase[t];
        v += wbase[wstr + t];
        v += wbase[2 * wstr + t];
        v += wbase[3 * wstr + t];
        __hip_atomic_store(p.partials + (size_t)slice * PROWS_CAP + rb + t, v, HG_RLX_AGENT);
    }
    wait_vmcnt<0>(); // every storing wave drains (eps + partials)
    __syncthreads();
    const unsigned long long t_drain = dbgp ? wall_clock64() : 0ull;
    // Two-level hand-off.  First the S workgroups of one group: the last of them to arrive sums that group's rows over the
    // slices in fixed order -- threads 0..127 take slices 0..31 of row t, threads 128..255 slices 32..63 (all 32 loads in
    // flight; partials is [slice][row], so the loads of one slice are coalesced), row total = (slices 0..31) + (slices
    // 32..63) -- and publishes the totals.  The reduction thus runs in all groups at once, behind the stragglers of the
    // streaming phase.
    if (tid == 0) {
        const uint32_t t = __hip_atomic_fetch_add(p.gticket + group, 1u, HG_RLX_AGENT);
        sh.flags[F_LAST] = (t == S - 1u) ? 1u : 0u;
    }
    __syncthreads();
    bool last_of_all = false;
    if (sh.flags[F_LAST]) { // last workgroup of its group
    if (kind != 0) {
        double* const gred = sh.tot; // rows of a Gram-only group, reduced over the slices, before they are folded per column
        const uint32_t half = tid >> 7, rl = tid & 127u;
        for (uint32_t rr0 = 0; rr0 < nrowg; rr0 += 128) {
            const uint32_t rr = rr0 + rl;
            const bool live = rr < nrowg;
            const double* col = p.partials + (size_t)(half * 32u) * PROWS_CAP + rb + (live ? rr : 0);
            double v[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) v[u] = (live && half * 32u + u < S) ? __hip_atomic_load(col + (size_t)u * PROWS_CAP, HG_RLX_AGENT) : 0.0;
            double acc = 0.0;
#pragma unroll
            for (int u = 0; u < 32; ++u) acc += v[u];
            if (rr0) __syncthreads(); // previous round's exchange buffer is free again
            if (half == 1) sh.red[rl] = acc;
            __syncthreads();
            if (live && half == 0) {
                if (kind == 2) __hip_atomic_store(p.totals + NR * c0 + rr, acc + sh.red[rl], HG_RLX_AGENT);
                else gred[rr] = acc + sh.red[rl];
            }
        }
        if (kind == 1) {
            // A carried column's dot lacks the pending updates: x_j'eps_now = carried_j + sum_q dbeta_q x_j'x_q with
            // x_j'x_q = mstd_j mstd_q (A_jq - N mave_j mave_q) (no missing calls) or mstd_j mstd_q (A - m_q B - m_j C + m_j m_q D).
            // This rank's part G_j = sum_q dbeta_q mstd_j mstd_q (integer sums over ITS individuals) goes into the column's
            // s1 row: it adds over ranks like every other row; the draw phase adds the carried dot and the N mave mave part.
            // A column the previous launch streamed AHEAD brings this rank's raw sums instead of a finished dot: they go
            // into the same row, mstd_j (s1 - mave_j s2) (s2 only where the column has missing calls; the draw phase adds
            // the common - mstd_j mave_j sum(eps) otherwise).
            __syncthreads();
            if ((uint32_t)tid < ncol) {
                const uint32_t j = c0 + (uint32_t)tid;
                const double mj = p.s_mave[d.cursor + j], sj = p.s_mstd[d.cursor + j];
                const double* r = gred + NRC * tid;
                double G = 0.0;
                if (j >= d.carry_left) {
                    const double* raw = p.ahead_raw + ((size_t)(apar ^ 1u) * AHEAD_MAX + (j - d.carry_left)) * 2;
                    const bool miss = !NOMISS && (p.s_ga[d.cursor + j] & 0x20000000) != 0;
                    G = sj * (raw[0] - (miss ? mj * raw[1] : 0.0));
                }
                for (int q = 0; q < npend; ++q) {
                    const double db = sh.pev[3 * q], mq = sh.pev[3 * q + 1], sq = sh.pev[3 * q + 2];
                    double a;
                    if constexpr (MG) a = ((r[4 * q] - mq * r[4 * q + 1]) - mj * r[4 * q + 2]) + (mj * mq) * r[4 * q + 3];
                    else a = r[q];
                    G += db * (sj * sq * a);
                }
                __hip_atomic_store(p.totals + NR * j, G, HG_RLX_AGENT);
                __hip_atomic_store(p.totals + NR * j + 1, 0.0, HG_RLX_AGENT);
                for (int f = 0; f < (SEG - 1) * T; ++f) __hip_atomic_store(p.totals + NR * j + NSUM + f, r[SEG * T + f], HG_RLX_AGENT);
            }
        }
    }
    wait_vmcnt<0>();
    __syncthreads();
    // then the groups: the last one to publish runs the draw phase
    if (tid == 0) {
        __hip_atomic_store(p.gticket + group, 0u, HG_RLX_AGENT);
        const uint32_t t = __hip_atomic_fetch_add(p.ticket, 1u, HG_RLX_AGENT);
        sh.flags[F_LAST] = (t == nactive - 1u) ? 1u : 0u;
    }
    __syncthreads();
    last_of_all = sh.flags[F_LAST] != 0;
    }

    if (last_of_all) {
    // ---- last-arriving workgroup ---------------------------------------------
    __builtin_amdgcn_s_setprio(3); // the whole chain waits for this workgroup; the others are streaming ahead at priority 0
    if (dbgp && tid == 0) {
        dbgp[1] = wall_clock64();
        dbgp[5] = t_entry;
        dbgp[6] = t_loop;
        dbgp[7] = t_drain;
    }
    {
        // all rows of this thread in flight at once
        constexpr int NRT = (NR * MAX_BATCH + BLOCK - 1) / BLOCK;
        const uint32_t nrows = NR * nb;
        double v[NRT];
#pragma unroll
        for (int i = 0; i < NRT; ++i) {
            const uint32_t rr = (uint32_t)tid + (uint32_t)i * BLOCK;
            v[i] = (rr < nrows) ? __hip_atomic_load(p.totals + rr, HG_RLX_AGENT) : 0.0;
        }
#pragma unroll
        for (int i = 0; i < NRT; ++i) {
            const uint32_t rr = (uint32_t)tid + (uint32_t)i * BLOCK;
            if (rr < nrows) sh.tot[rr] = v[i];
        }
    }
    if (tid == 0) {
        __hip_atomic_store(p.ticket, 0u, HG_RLX_AGENT);
        __hip_atomic_store(p.aqueue + (apar ^ 1u), 0u, HG_RLX_AGENT); // the next launch's ahead queue (nobody is on that half now)
        sh.flags[F_P2PTMO] = 0u;
    }
    __syncthreads();
    if (dbgp && tid == 0) dbgp[2] = wall_clock64();

    if (p.p2p.nranks > 1 && !p.sums_out) { // multi-GPU, in-launch exchange
        if (!p2p_exchange<NR>(p, d, nb, sh)) {
            if (tid == 0) {
                p.desc->error = 3u;
            }
            return;
        }
    }
    if (p.sums_out) { // multi-GPU: hand the local sums to the all-reduce
        for (int r = tid; r < NR * MAX_BATCH; r += BLOCK) p.sums_out[r] = (r < NR * (int)nb) ? sh.tot[r] : 0.0;
        return;
    }
    sweep_draw_phase<SEG, MG, NOMISS, DBG>(p, d, nbs, sh);
    __syncthreads(); // (then on to the ahead queue like everybody else: a launch of a single workgroup has nobody else to serve it)
    }

    // ---- the ahead queue ---------------------------------
    if (!last_of_all) __builtin_amdgcn_s_setprio(0); // behind the stragglers of the batch and behind the drawing workgroup on this compute unit
    n_ahead = sweep_ahead_cols<MG, NOMISS>(p, d, nb, sh, tid); // (the staged flags are in place: several barriers ago)
    a_groups = (n_ahead + CPG - 1) / CPG;
    {
        const uint32_t fit = a_groups ? p.resident / a_groups : 1u;
        a_slices = fit < p.slices_max ? (fit ? fit : 1u) : p.slices_max;
    }
    }
    if (a_groups == 0u) return;
    for (;;) { // items of the ahead queue
        if (tid == 0) sh.flags[F_POS] = __hip_atomic_fetch_add(p.aqueue + apar, 1u, HG_RLX_AGENT);
        __syncthreads();
        const uint32_t aitem = sh.flags[F_POS];
        __syncthreads();
        if (aitem >= a_groups * a_slices) return;
        const uint32_t ag = aitem / a_slices, a0 = ag * CPG;
        slice = aitem % a_slices;
        Sw = a_slices;
        nt = slice < ntg ? (ntg - slice + Sw - 1) / Sw : 0u;
        c0 = nb + a0;
        c1 = (c0 + CPG < nb + n_ahead) ? c0 + CPG : nb + n_ahead;
        ncol = c1 - c0;
        wbase = sh.wpart;
        wstr = sh.wstride;
        fresh_item(std::true_type{});
        __syncthreads();
        // publish the item's rows (s1, s2 per column); the last of the item's slices sums them over the slices in order
        for (uint32_t t = tid; t < 2u * ncol; t += BLOCK) {
            const uint32_t c = t >> 1, r = t & 1u;
            double v = wbase[NR * c + r];
            v += wbase[wstr + NR * c + r];
            v += wbase[2 * wstr + NR * c + r];
            v += wbase[3 * wstr + NR * c + r];
            __hip_atomic_store(p.apartials + (size_t)slice * (2 * AHEAD_MAX) + 2 * (a0 + c) + r, v, HG_RLX_AGENT);
        }
        wait_vmcnt<0>();
        __syncthreads();
        if (tid == 0) {
            const uint32_t t = __hip_atomic_fetch_add(p.aticket + ag, 1u, HG_RLX_AGENT);
            sh.flags[F_LAST] = (t == a_slices - 1u) ? 1u : 0u;
        }
        __syncthreads();
        if (sh.flags[F_LAST]) {
            if ((uint32_t)tid < 2u * ncol) { // slices 0..31 then 32..63, all loads of a half in flight, added in order
                const double* row = p.apartials + 2 * a0 + tid;
                double acc = 0.0;
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    double v[32];
#pragma unroll
                    for (int u = 0; u < 32; ++u) v[u] = ((uint32_t)(32 * h2 + u) < a_slices) ? __hip_atomic_load(row + (size_t)(32 * h2 + u) * (2 * AHEAD_MAX), HG_RLX_AGENT) : 0.0;
#pragma unroll
                    for (int u = 0; u < 32; ++u) acc += v[u];
                }
                p.ahead_raw[((size_t)apar * AHEAD_MAX + a0) * 2 + tid] = acc; // read by the next launch
            }
            if (tid == 0) __hip_atomic_store(p.aticket + ag, 0u, HG_RLX_AGENT);
        }
        __syncthreads();
    }
}

// Multi-GPU second half: sums_out has been all-reduced over ranks.
template <int SEG, int MG>
__global__ __launch_bounds__(BLOCK) void k_sweep_draw(SweepParams p)
{
    unsigned long long* const dbgp = nullptr;
    (void)dbgp;
    constexpr int NR = sweep_rows(SEG, MG);
    const SweepShared sh = sweep_lds_carve(hg_smem, p.batch_cap, p.cols_per_group, p.K, NR);
    const DescHead d = load_desc_head(p.desc);
    const bool pend = d.pend_marker[0] >= 0;
    const uint32_t remaining = (d.cursor < p.M) ? p.M - d.cursor : 0u;
    uint32_t nbs[SEG];
#pragma unroll
    for (int q = 0; q < SEG; ++q) {
        const uint32_t e = d.seg_end[q] < remaining ? d.seg_end[q] : remaining;
        nbs[q] = (q && e < nbs[q - 1]) ? nbs[q - 1] : e;
    }
    const uint32_t nb2 = nbs[SEG - 1];
    if ((nb2 == 0 && !pend) || d.error) return;
    stage_marker_meta(p, d, nb2, threadIdx.x, sh);
    stage_rng(p, sh, threadIdx.x);
    if (threadIdx.x < 3 * MAX_SEG) sh.pev[threadIdx.x] = pend ? p.desc->pend_ev[threadIdx.x / 3][threadIdx.x % 3] : 0.0;
    for (int r = threadIdx.x; r < NR * (int)nb2; r += BLOCK) sh.tot[r] = p.sums_out[r];
    __syncthreads();
    sweep_draw_phase<SEG, MG, 0, 0>(p, d, nbs, sh);
}

} // namespace hg
