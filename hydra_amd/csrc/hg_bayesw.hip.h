// hg_bayesw.hip.h -- BayesW (Weibull survival) operators behind include/hgibbs.h, for gfx950.
// Included at the end of hgibbs.hip: shares the handle, the packed genotype shard, the permuted
// residual layout and the covariate columns with the BayesR path.
//
// What the reference does per marker (src/BayesW.cpp:1484-1622): three sums of
// vi = exp(alpha*eps - EuMasc) masked by genotype, adaptive Gauss-Hermite marginal likelihoods
// of the K-1 slab components, a categorical walk with one Boost uniform, and -- if a slab is
// chosen -- one adaptive-rejection draw of the effect on scalars only; a non-zero change of the
// effect updates eps and refreshes vi.  Here:
//   k_bw_sums     one launch = the masked sums of up to 256 markers against the SAME vi (valid
//                 until the first effect changes), 8 columns per workgroup per pass over vi.
//                 A marker whose effect was non-zero rides along as one extra column group
//                 computing exp(alpha*(eps + effect*x) - EuMasc) on the fly.
//   k_bw_tail     one wavefront per column: fixed-order reduction of the slice partials, every
//                 (component, node) term of the quadrature on its own lane, the walk over the
//                 components with the column's uniform; the last wavefront reports the first
//                 column that is an event (slab chosen, or an effect that was non-zero) straight
//                 into pinned host memory.
//   k_bw_refresh  eps += delta(genotype) for the event's marker fused with vi = exp(alpha*eps - EuMasc)
//                 and the block partials of sum(vi).  Inside a sweep this pass rides in the NEXT batch's k_bw_sums
//                 (the pending update, as in the BayesR sweep): one extra group of workgroups stores the other eps / vi
//                 buffers and the block partials, the streaming groups take vi * exp(alpha*delta(genotype)) -- three
//                 constants, from a pair table in LDS -- on their tile in registers.  The kernel itself remains for the
//                 single-marker operators and for the last event of a sweep.
//   k_bw_reduce   the N-length sums inside the log densities of mu, alpha and the covariates.
// The uniforms are one u32 per marker in sweep order, so they are generated up front; the ARS
// draw runs on the host between launches (hg_ars.h), on scalars the launch hands back.
#pragma once

#include "hg_ars.h"
#include "hg_bayesw_math.h"
#include "hg_gh_tables.h"

namespace {

constexpr uint32_t BW_ROWS = 2 * MAX_BATCH + 4; // per slice: (vi_1, vi_2) per column + (sum, vi_1, vi_2) of the shifted column

struct BwResult {
    uint32_t event; // index in the batch of the first event, == columns in the batch if none
    int32_t k;      // component picked for the event column
    double vi_sum, vi_1, vi_2; // its masked sums
    unsigned long long seq;    // written last: the batch this result belongs to
};

struct BwBatchParams {
    const uint8_t* bed;
    uint64_t stride;
    const double* eps;
    const double* vi;
    uint32_t n_pad, n_local;
    const int32_t* markers; // device: order + cursor
    uint32_t ncols;         // columns whose effect is zero
    int32_t shifted_marker; // -1, or the marker right after them whose effect is not zero
    double dv[3];           // what eps gains per genotype when that effect is taken out
    // the previous batch's event, not applied yet (pend_marker >= 0): eps gains pend_dv[genotype], vi is multiplied by
    // pend_f[genotype] = exp(alpha * pend_dv[genotype]); the update group stores eps_out / vi_out (vi recomputed with exp)
    int32_t pend_marker;
    double pend_dv[3], pend_f[3];
    double* eps_out;
    double* vi_out;
    double* vipart_out;     // block partials of sum(vi_out) (read by this batch's tail)
    uint32_t upd_groups;    // groups of workgroups that share the update pass (its tiles cost several times a streaming tile: exp per individual)
    double alpha;
    uint32_t slices;
    double* partials; // [slice][BW_ROWS]
    uint32_t* ticket;
    // quadrature + walk
    const double* p_unif; // device: uniforms + cursor
    const double *mave, *sd, *sumfail;
    const int32_t* groups;
    const double *cva, *pi, *sigmaG;
    int K;
    const double *ghx, *ghw;
    int quad;
    const double* vipart; // block partials of sum(vi) from the last refresh
    uint32_t n_vipart;
    double* rows;         // several ranks: BW_ROWS local row sums (k_bw_rows), all-reduced before the tail reads them; else null
    BwResult* result;     // pinned host memory
    unsigned long long seq;
    int32_t* picks;       // MAX_BATCH + 1
    int32_t* col_k;       // MAX_BATCH + 1
    double* col_sums;     // (MAX_BATCH + 1) x 3
    uint32_t* first_event;
};

// eps (+ delta of one marker) -> eps, vi; per-block partial of sum(vi).  marker < 0: refresh only.
__global__ __launch_bounds__(BLOCK) void k_bw_refresh(const uint8_t* __restrict__ bed, uint64_t stride, int32_t marker, double v0, double v1,
                                                      double v2, double* eps, double* vi, double alpha, uint32_t n_local, double* partial)
{
    __shared__ double sh[BLOCK_WAVES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t tile = blockIdx.x * BLOCK_WAVES + wave;
    double e[IPT], v[IPT];
    load_eps16(eps, tile, lane, e);
    if (marker >= 0) {
        const uint32_t w = *reinterpret_cast<const uint32_t*>(bed + (size_t)marker * stride + ((size_t)tile << 8) + (lane << 2));
        apply_update16(w, v0, v1, v2, e);
        store_eps16(eps, tile, lane, e);
    }
    const uint32_t i0 = (tile << 10) + ((uint32_t)lane << 4);
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
        v[k] = (i0 + k < n_local) ? exp(alpha * e[k] - bw::EULER) : 0.0;
        s += v[k];
    }
    store_eps16(vi, tile, lane, v);
    s = wave_sum(s);
    if (lane == 0) sh[wave] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < BLOCK_WAVES; ++w) t += sh[w];
        partial[blockIdx.x] = t;
    }
}

// N-length sums inside the scalar log densities.  kind 0: mu_dens (src/BayesW.cpp:77-88) with
// used = eps + p0:  sum exp((used - p1)*p2 - EuMasc);  kind 1: alpha_dens (:132-142) sum exp(eps*p0 - EuMasc);
// kind 2: gamma_dens (:118-129) with used = eps + x*p0: sum exp((used - x*p1)*p2 - EuMasc);
// kind 3: sum eps*failure (:139).
__global__ __launch_bounds__(BLOCK) void k_bw_reduce(const double* __restrict__ eps, const double* __restrict__ x,
                                                     const uint32_t* __restrict__ failspread, int kind, double p0, double p1, double p2,
                                                     uint32_t n_local, double* partial)
{
    __shared__ double sh[BLOCK_WAVES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t tile = blockIdx.x * BLOCK_WAVES + wave;
    double e[IPT], xv[IPT];
    load_eps16(eps, tile, lane, e);
    if (kind == 2) load_eps16(x, tile, lane, xv);
    const uint32_t fs = (kind == 3) ? failspread[(tile << 6) + lane] : 0u;
    const uint32_t i0 = (tile << 10) + ((uint32_t)lane << 4);
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
        if (i0 + k >= n_local) continue;
        double t;
        if (kind == 0) t = exp(((e[k] + p0) - p1) * p2 - bw::EULER);
        else if (kind == 1) t = exp((e[k] * p0) - bw::EULER);
        else if (kind == 2) t = exp((((e[k] + xv[k] * p0) - xv[k] * p1) * p2) - bw::EULER);
        else t = ((fs >> (2 * k)) & 1u) ? e[k] : 0.0;
        s += t;
    }
    s = wave_sum(s);
    if (lane == 0) sh[wave] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < BLOCK_WAVES; ++w) t += sh[w];
        partial[blockIdx.x] = t;
    }
}

// per marker: sum of the failure indicator over genotype 1 / genotype 2 (src/BayesW.cpp:1221-1228)
__global__ __launch_bounds__(BLOCK) void k_bw_fail_counts(const uint8_t* __restrict__ bed, uint64_t stride, const uint32_t* __restrict__ failspread,
                                                          uint32_t ndw, unsigned long long* out)
{
    __shared__ unsigned long long sh[2][BLOCK_WAVES];
    const uint32_t marker = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t* col = reinterpret_cast<const uint32_t*>(bed + (size_t)marker * stride);
    unsigned long long c1 = 0, c2 = 0;
    for (uint32_t i = threadIdx.x; i < ndw; i += BLOCK) {
        uint32_t m1, m2, mm;
        code_masks(col[i], m1, m2, mm);
        const uint32_t fs = failspread[i];
        c1 += (unsigned)__popc(m1 & fs);
        c2 += (unsigned)__popc(m2 & fs);
    }
    for (int off = 32; off > 0; off >>= 1) {
        c1 += __shfl_down(c1, off);
        c2 += __shfl_down(c2, off);
    }
    if (lane == 0) {
        sh[0][wave] = c1;
        sh[1][wave] = c2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long a = 0, b = 0;
        for (int w = 0; w < BLOCK_WAVES; ++w) {
            a += sh[0][w];
            b += sh[1][w];
        }
        out[2 * marker] = a;
        out[2 * marker + 1] = b;
    }
}

// failure indicator of 16 consecutive individuals -> bits 0,2,4,...,30 (the genotype bit planes' positions)
__global__ void k_bw_failspread(const int32_t* __restrict__ fail, uint32_t n_local, uint32_t ndw, uint32_t* out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ndw) return;
    uint32_t m = 0;
    for (int s = 0; s < 16; ++s) {
        const uint32_t ind = i * 16 + s;
        if (ind < n_local && fail[ind] != 0) m |= 1u << (2 * s);
    }
    out[i] = m;
}

// the three masked sums of ONE marker, per-block partials (the single-marker operator)
__global__ __launch_bounds__(BLOCK) void k_bw_marker_sums(const uint8_t* __restrict__ bed, uint64_t stride, uint32_t marker,
                                                          const double* __restrict__ eps, const double* __restrict__ vi, bool shifted, double d0,
                                                          double d1, double d2, double alpha, uint32_t n_local, double* partial)
{
    __shared__ double sh[3][BLOCK_WAVES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t tile = blockIdx.x * BLOCK_WAVES + wave;
    double e[IPT];
    load_eps16(shifted ? eps : vi, tile, lane, e);
    const uint32_t w = *reinterpret_cast<const uint32_t*>(bed + (size_t)marker * stride + ((size_t)tile << 8) + (lane << 2));
    const uint32_t i0 = (tile << 10) + ((uint32_t)lane << 4);
    double s = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
        const uint32_t c = (w >> (2 * k)) & 3u;
        double t = e[k];
        if (shifted) {
            const double de = (c == GC_G0) ? d0 : ((c == GC_G1) ? d1 : ((c == GC_G2) ? d2 : 0.0));
            t = (i0 + k < n_local) ? exp(alpha * (e[k] + de) - bw::EULER) : 0.0;
        }
        s += t;
        s1 += (c == GC_G1) ? t : 0.0;
        s2 += (c == GC_G2) ? t : 0.0;
    }
    s = wave_sum(s);
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if (lane == 0) {
        sh[0][wave] = s;
        sh[1][wave] = s1;
        sh[2][wave] = s2;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        double t = 0.0;
        for (int w2 = 0; w2 < BLOCK_WAVES; ++w2) t += sh[threadIdx.x][w2];
        partial[3 * blockIdx.x + threadIdx.x] = t;
    }
}

template <int CPG>
__global__ __launch_bounds__(BLOCK, 3) void k_bw_sums(BwBatchParams p)
{
    __shared__ double wpart[BLOCK_WAVES][2 * CPG + 4];
    __shared__ double2 ftab[16], atab[16]; // the pending update over a PAIR of codes: factors of vi, addends of eps
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t ngn = (p.ncols + CPG - 1) / CPG;
    const uint32_t S = p.slices;
    const uint32_t slice = blockIdx.x % S, group = blockIdx.x / S;
    const uint32_t ntg = p.n_pad / BLOCK_IND;
    const bool pend = p.pend_marker >= 0;
    if (pend && tid < 16) {
        auto fac = [&](uint32_t c) { return (c == GC_G0) ? p.pend_f[0] : ((c == GC_G1) ? p.pend_f[1] : ((c == GC_G2) ? p.pend_f[2] : 1.0)); };
        auto add = [&](uint32_t c) { return 0.0 + ((c == GC_G0) ? p.pend_dv[0] : ((c == GC_G1) ? p.pend_dv[1] : ((c == GC_G2) ? p.pend_dv[2] : 0.0))); };
        ftab[tid] = make_double2(fac((uint32_t)tid & 3u), fac(((uint32_t)tid >> 2) & 3u));
        atab[tid] = make_double2(add((uint32_t)tid & 3u), add(((uint32_t)tid >> 2) & 3u));
    }
    const uint8_t* pendp = p.bed + (size_t)(pend ? p.pend_marker : 0) * p.stride + (lane << 2);
    __syncthreads();
    const uint32_t nshift = p.shifted_marker >= 0 ? 1u : 0u;

    if (group < ngn) {
        const uint32_t c0 = group * CPG;
        const uint32_t ncol = (c0 + CPG <= p.ncols) ? CPG : p.ncols - c0;
        const uint8_t* colp[CPG];
        double a1[CPG], a2[CPG];
#pragma unroll
        for (int c = 0; c < CPG; ++c) {
            const uint32_t j = (c0 + c < p.ncols) ? c0 + c : p.ncols - 1;
            colp[c] = p.bed + (size_t)p.markers[j] * p.stride + (lane << 2);
            a1[c] = a2[c] = 0.0;
        }
        for (uint32_t tg = slice; tg < ntg; tg += S) {
            const uint32_t tile = tg * BLOCK_WAVES + wave;
            double e[IPT];
            load_eps16(p.vi, tile, lane, e);
            if (pend) { // vi with the pending update: exp(alpha (eps + delta) - EuMasc) = vi * exp(alpha delta), delta one of three constants
                const uint32_t wq = *reinterpret_cast<const uint32_t*>(pendp + ((size_t)tile << 8));
#pragma unroll
                for (int s2 = 0; s2 < IPT; s2 += 2) {
                    const double2 f = ftab[(wq >> (2 * s2)) & 15u];
                    e[s2] = e[s2] * f.x;
                    e[s2 + 1] = e[s2 + 1] * f.y;
                }
            }
            uint32_t m1[CPG], m2[CPG];
#pragma unroll
            for (int c = 0; c < CPG; ++c) {
                uint32_t mm;
                code_masks(*reinterpret_cast<const uint32_t*>(colp[c] + ((size_t)tile << 8)), m1[c], m2[c], mm);
            }
            // weight 0/1 times vi is exact: one rounding per add, slots in increasing order
#pragma unroll
            for (int c = 0; c < CPG; c += 4) {
                fma_slots4(m1[c], m1[c + 1], m1[c + 2], m1[c + 3], e, a1[c], a1[c + 1], a1[c + 2], a1[c + 3], std::make_integer_sequence<int, IPT>{});
                fma_slots4(m2[c], m2[c + 1], m2[c + 2], m2[c + 3], e, a2[c], a2[c + 1], a2[c + 2], a2[c + 3], std::make_integer_sequence<int, IPT>{});
            }
        }
#pragma unroll
        for (int c = 0; c < CPG; ++c) {
            const double t1 = wave_sum(a1[c]), t2 = wave_sum(a2[c]);
            if (lane == 0) {
                wpart[wave][2 * c] = t1;
                wpart[wave][2 * c + 1] = t2;
            }
        }
        __syncthreads();
        if (tid < (int)(2 * ncol)) {
            double v = wpart[0][tid];
            v += wpart[1][tid];
            v += wpart[2][tid];
            v += wpart[3][tid];
            __hip_atomic_store(p.partials + (size_t)slice * BW_ROWS + 2 * c0 + tid, v, HG_RLX_AGENT);
        }
    } else if (group < ngn + nshift) {
        // the marker whose effect is not zero: vi as it would be with that effect taken out (src/BayesW.cpp:1499-1516)
        const uint8_t* colp = p.bed + (size_t)p.shifted_marker * p.stride + (lane << 2);
        double s = 0.0, s1 = 0.0, s2 = 0.0;
        for (uint32_t tg = slice; tg < ntg; tg += S) {
            const uint32_t tile = tg * BLOCK_WAVES + wave;
            double e[IPT];
            load_eps16(p.eps, tile, lane, e);
            if (pend) apply_update16_lds(*reinterpret_cast<const uint32_t*>(pendp + ((size_t)tile << 8)), atab, e);
            const uint32_t w = *reinterpret_cast<const uint32_t*>(colp + ((size_t)tile << 8));
            const uint32_t i0 = (tile << 10) + ((uint32_t)lane << 4);
#pragma unroll
            for (int k = 0; k < IPT; ++k) {
                const uint32_t c = (w >> (2 * k)) & 3u;
                const double de = (c == GC_G0) ? p.dv[0] : ((c == GC_G1) ? p.dv[1] : ((c == GC_G2) ? p.dv[2] : 0.0));
                const double t = (i0 + k < p.n_local) ? exp(p.alpha * (e[k] + de) - bw::EULER) : 0.0;
                s += t;
                s1 += (c == GC_G1) ? t : 0.0;
                s2 += (c == GC_G2) ? t : 0.0;
            }
        }
        s = wave_sum(s);
        s1 = wave_sum(s1);
        s2 = wave_sum(s2);
        if (lane == 0) {
            wpart[wave][0] = s;
            wpart[wave][1] = s1;
            wpart[wave][2] = s2;
        }
        __syncthreads();
        if (tid < 3) {
            double v = wpart[0][tid];
            v += wpart[1][tid];
            v += wpart[2][tid];
            v += wpart[3][tid];
            __hip_atomic_store(p.partials + (size_t)slice * BW_ROWS + 2 * MAX_BATCH + tid, v, HG_RLX_AGENT);
        }
    } else if (pend) {
        // the update group: the pending event's pass over eps and vi (what k_bw_refresh does on its own launch): the other eps
        // and vi buffers and the block partials of sum(vi), src/BayesW.cpp:1606-1622, :1812, :1832-1834
        const uint32_t su = (group - (ngn + nshift)) * S + slice, SU = S * p.upd_groups; // this workgroup's place among the update workgroups
        for (uint32_t tg = su; tg < ntg; tg += SU) {
            const uint32_t tile = tg * BLOCK_WAVES + wave;
            double e[IPT], v[IPT];
            load_eps16(p.eps, tile, lane, e);
            apply_update16_lds(*reinterpret_cast<const uint32_t*>(pendp + ((size_t)tile << 8)), atab, e);
            store_eps16(p.eps_out, tile, lane, e);
            const uint32_t i0 = (tile << 10) + ((uint32_t)lane << 4);
            double sv = 0.0;
#pragma unroll
            for (int k = 0; k < IPT; ++k) {
                v[k] = (i0 + k < p.n_local) ? exp(p.alpha * e[k] - bw::EULER) : 0.0;
                sv += v[k];
            }
            store_eps16(p.vi_out, tile, lane, v);
            sv = wave_sum(sv);
            __syncthreads(); // the previous tile group's partial has been read
            if (lane == 0) wpart[wave][0] = sv;
            __syncthreads();
            if (tid == 0) {
                double t = 0.0;
                for (int w2 = 0; w2 < BLOCK_WAVES; ++w2) t += wpart[w2][0];
                p.vipart_out[tg] = t;
            }
        }
    }
}

// Individuals sharded over several GPUs: this rank's row sums (slice partials per column, the shifted
// column's three, the partials of sum(vi)) in one dense buffer for the all-reduce between k_bw_sums and k_bw_tail.
// rows[2c], rows[2c+1]; rows[2*MAX_BATCH .. +2] the shifted column; rows[2*MAX_BATCH+3] = sum(vi).
__global__ __launch_bounds__(WAVE) void k_bw_rows(BwBatchParams p)
{
    const int lane = threadIdx.x;
    const uint32_t col = blockIdx.x;
    const uint32_t nb = p.ncols + (p.shifted_marker >= 0 ? 1u : 0u);
    if (col == nb) {
        double v = 0.0;
        for (uint32_t b = lane; b < p.n_vipart; b += WAVE) v += p.vipart[b];
        v = wave_sum(v);
        if (lane == 0) p.rows[2 * MAX_BATCH + 3] = v;
        return;
    }
    const bool shifted = col >= p.ncols;
    const uint32_t r0 = shifted ? 2 * MAX_BATCH : 2 * col;
    const double* row = p.partials + (size_t)lane * BW_ROWS + r0;
    const bool live = (uint32_t)lane < p.slices;
    double t0 = live ? __hip_atomic_load(row, HG_RLX_AGENT) : 0.0;
    double t1 = live ? __hip_atomic_load(row + 1, HG_RLX_AGENT) : 0.0;
    double t2 = (live && shifted) ? __hip_atomic_load(row + 2, HG_RLX_AGENT) : 0.0;
    t0 = wave_sum(t0);
    t1 = wave_sum(t1);
    t2 = wave_sum(t2);
    if (lane == 0) {
        p.rows[r0] = t0;
        p.rows[r0 + 1] = t1;
        if (shifted) p.rows[r0 + 2] = t2;
    }
}

// One wavefront per column: reduce the column's slice partials, evaluate every (component, node)
// term of the quadrature on its own lane, sum them in the reference's order, walk the components
// with the column's uniform.  The last wavefront to finish reports the first event of the batch
// straight into pinned host memory (data, fence, then the sequence number the host is polling).
__global__ __launch_bounds__(WAVE) void k_bw_tail(BwBatchParams p)
{
    __shared__ double terms[(bw::MAX_K - 1) * 24];
    __shared__ double s_ml[bw::MAX_K];
    __shared__ uint32_t s_last;
    const int lane = threadIdx.x;
    const uint32_t col = blockIdx.x;
    const uint32_t nb = p.ncols + (p.shifted_marker >= 0 ? 1u : 0u);
    const bool shifted = col >= p.ncols;
    const uint32_t S = p.slices;

    // fixed-order (wave tree) sums over the S <= 64 slices; sum(vi) from the refresh kernel's block partials
    const uint32_t r0 = shifted ? 2 * MAX_BATCH : 2 * col;
    bw::MarkerSums sums;
    if (p.rows) { // summed over the ranks already
        if (shifted) sums = bw::MarkerSums{p.rows[r0], p.rows[r0 + 1], p.rows[r0 + 2]};
        else sums = bw::MarkerSums{p.rows[2 * MAX_BATCH + 3], p.rows[r0], p.rows[r0 + 1]};
    } else {
        const double* row = p.partials + (size_t)lane * BW_ROWS + r0;
        const bool live = (uint32_t)lane < S;
        double t0 = live ? __hip_atomic_load(row, HG_RLX_AGENT) : 0.0;
        double t1 = live ? __hip_atomic_load(row + 1, HG_RLX_AGENT) : 0.0;
        double t2 = (live && shifted) ? __hip_atomic_load(row + 2, HG_RLX_AGENT) : 0.0;
        t0 = wave_sum(t0);
        t1 = wave_sum(t1);
        if (shifted) {
            t2 = wave_sum(t2);
            sums = bw::MarkerSums{t0, t1, t2};
        } else {
            double v = 0.0;
            for (uint32_t b = lane; b < p.n_vipart; b += WAVE) v += p.vipart[b];
            sums = bw::MarkerSums{wave_sum(v), t0, t1};
        }
    }
    const int marker = shifted ? p.shifted_marker : p.markers[col];
    const int grp = p.groups[marker];
    const int K = p.K, Q = p.quad, km1 = K - 1;
    const double mean = p.mave[marker], sd = p.sd[marker], alpha = p.alpha, sigmaG = p.sigmaG[grp], dj = p.sumfail[marker];
    const double vi_0 = sums.vi_sum - sums.vi_1 - sums.vi_2;
    const double exp_sum = (sums.vi_1 * (1 - 2 * mean) + 4 * (1 - mean) * sums.vi_2 + sums.vi_sum * mean * mean) / (sd * sd);
    const double* cva = p.cva + (size_t)grp * km1;
    for (int idx = lane; idx < km1 * (Q - 1); idx += WAVE) { // src/BayesW.cpp:161-169, :713-726
        const int i = idx / (Q - 1), q = idx % (Q - 1);
        const double sigma = 1.0 / sqrt(1 + alpha * alpha * sigmaG * cva[i] * exp_sum);
        terms[idx] = p.ghw[q] * bw::gh_integrand(sigma * p.ghx[q], alpha, dj, sqrt(2 * cva[i] * sigmaG), sums.vi_sum, sums.vi_2, sums.vi_1, vi_0, sd, mean / sd);
    }
    __syncthreads();
    if (lane < km1) { // the reference's order: w1 f1 + w2 f2 + ... + w_{n-1} f_{n-1} + w_n, times sigma, times pi
        const double sigma = 1.0 / sqrt(1 + alpha * alpha * sigmaG * cva[lane] * exp_sum);
        double temp = terms[lane * (Q - 1)];
        for (int q = 1; q < Q - 1; ++q) temp = temp + terms[lane * (Q - 1) + q];
        temp = temp + p.ghw[Q - 1];
        s_ml[lane + 1] = p.pi[(size_t)grp * K + lane + 1] * (sigma * temp);
    }
    if (lane == 0) s_ml[0] = p.pi[(size_t)grp * K] * bw::SQRT_PI;
    __syncthreads();
    if (lane == 0) {
        double ml[bw::MAX_K];
        for (int k = 0; k < K; ++k) ml[k] = s_ml[k];
        const int k = bw::pick_component(K, ml, p.p_unif[col]);
        p.picks[col] = k;
        __hip_atomic_store(p.col_sums + 3 * col, sums.vi_sum, HG_RLX_AGENT);
        __hip_atomic_store(p.col_sums + 3 * col + 1, sums.vi_1, HG_RLX_AGENT);
        __hip_atomic_store(p.col_sums + 3 * col + 2, sums.vi_2, HG_RLX_AGENT);
        __hip_atomic_store(p.col_k + col, k, HG_RLX_AGENT);
        if (shifted || k != 0) __hip_atomic_fetch_min(p.first_event, col, HG_RLX_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const uint32_t t = __hip_atomic_fetch_add(p.ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (t == gridDim.x - 1u) ? 1u : 0u;
        if (s_last) {
            const uint32_t ev = __hip_atomic_load(p.first_event, HG_RLX_AGENT);
            BwResult r{nb, 0, 0.0, 0.0, 0.0, 0ull};
            if (ev < nb) {
                r.event = ev;
                r.k = __hip_atomic_load(p.col_k + ev, HG_RLX_AGENT);
                r.vi_sum = __hip_atomic_load(p.col_sums + 3 * ev, HG_RLX_AGENT);
                r.vi_1 = __hip_atomic_load(p.col_sums + 3 * ev + 1, HG_RLX_AGENT);
                r.vi_2 = __hip_atomic_load(p.col_sums + 3 * ev + 2, HG_RLX_AGENT);
            }
            __hip_atomic_store(p.ticket, 0u, HG_RLX_AGENT);
            __hip_atomic_store(p.first_event, 0xffffffffu, HG_RLX_AGENT);
            __hip_atomic_store(&p.result->event, r.event, HG_RLX_SYSTEM);
            __hip_atomic_store(&p.result->k, r.k, HG_RLX_SYSTEM);
            __hip_atomic_store(&p.result->vi_sum, r.vi_sum, HG_RLX_SYSTEM);
            __hip_atomic_store(&p.result->vi_1, r.vi_1, HG_RLX_SYSTEM);
            __hip_atomic_store(&p.result->vi_2, r.vi_2, HG_RLX_SYSTEM);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
            __hip_atomic_store(&p.result->seq, p.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// diagnostic: the ARS draw of an effect on ONE lane of the device (what the event's continuation on the device would cost)
struct BwArsProbe {
    double d[9], beta_old, safe_limit;
    uint32_t seed, ndraws;
    double* out; // {100 MHz ticks, density evaluations, last draw, error}
};
__global__ __launch_bounds__(WAVE) void k_bw_ars_probe(BwArsProbe pr)
{
    if (threadIdx.x != 0) return;
    hg::GlibcRand g;
    g.seed(pr.seed);
    struct U {
        hg::GlibcRand* g;
        __device__ double operator()() { return g->uniform(); }
    } u{&g};
    bw::BetaLogDensity f{pr.d[0], pr.d[1], pr.d[2], pr.d[3], pr.d[4], pr.d[5], pr.d[6], pr.d[7], pr.d[8]};
    const double b = pr.beta_old, sl = pr.safe_limit;
    const double xinit[4] = {b - sl / 10, b, b + sl / 20, b + sl / 10};
    double last = 0.0;
    unsigned long long evals = 0;
    int err = 0;
    const unsigned long long t0 = wall_clock64();
    for (uint32_t i = 0; i < pr.ndraws && !err; ++i) {
        hg::ars::Hull hull;
        int ne = 0;
        err = hg::ars::sample(xinit, b - sl, b + sl, f, u, hull, last, ne);
        evals += (unsigned long long)ne;
    }
    const unsigned long long t1 = wall_clock64();
    pr.out[0] = (double)(t1 - t0);
    pr.out[1] = (double)evals;
    pr.out[2] = last;
    pr.out[3] = (double)err;
}

} // namespace

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
struct BwState {
    double* vi = nullptr;
    double* vi2 = nullptr; // the other vi buffer: a batch that applies a pending update writes it (k_bw_sums' update group)
    uint32_t* failspread = nullptr;
    double* vi_sum = nullptr; // device scalar
    double *d_mave = nullptr, *d_sd = nullptr, *d_sumfail = nullptr;
    double *d_cva = nullptr, *d_pi = nullptr, *d_sigmaG = nullptr;
    double *d_ghx = nullptr, *d_ghw = nullptr;
    double* d_unif = nullptr;
    double* partials = nullptr;
    BwResult* h_result = nullptr; // pinned, written by the device
    unsigned long long seq = 0;
    double* vipart = nullptr; // block partials of sum(vi) from the last refresh
    double* d_rows = nullptr; // BW_ROWS: this rank's row sums, all-reduced between the two batch kernels (several ranks only)
    int32_t* d_colk = nullptr;
    uint32_t* d_first_event = nullptr;
    hipEvent_t evk0 = nullptr, evk1 = nullptr;
    int32_t* d_picks = nullptr;
    double* d_colsums = nullptr;
    std::vector<double> mave, sd, sumfail, cva, beta;
    std::vector<int32_t> comp, fail;
    std::vector<double> unif;
    double d_total = 0.0; // number of events (sum of the failure indicator)
    int G = 0, K = 0, quad = 0;
    bool have_tables = false, have_model = false;
    hgibbs_w_sweep_stats stats{};
};

static void bw_free(BwState* b)
{
    if (!b) return;
    void* ptrs[] = {b->vi, b->vi2, b->failspread, b->vi_sum, b->d_mave, b->d_sd, b->d_sumfail, b->d_cva, b->d_pi, b->d_sigmaG, b->d_ghx, b->d_ghw,
                    b->d_unif, b->partials, b->d_picks, b->d_colsums, b->vipart, b->d_colk, b->d_first_event, b->d_rows};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    if (b->h_result) (void)hipHostFree(b->h_result);
    if (b->evk0) (void)hipEventDestroy(b->evk0);
    if (b->evk1) (void)hipEventDestroy(b->evk1);
    delete b;
}

static int bw_need(hgibbs_ctx* h, const char* who, bool tables = false, bool model = false)
{
    if (!h || !h->bed) return fail("%s: load genotypes first", who);
    if (!h->bw) return fail("%s: call hgibbs_w_init first", who);
    if (tables && !h->bw->have_tables) return fail("%s: call hgibbs_w_marker_stats first", who);
    if (model && !h->bw->have_model) return fail("%s: call hgibbs_w_set_model first", who);
    return 0;
}

// sum of `nblk` block partials in h->scratch -> device scalar dst (fixed order)
static int bw_final(hgibbs_ctx* h, uint32_t nblk, double* dst)
{
    k_final_sum<<<1, BLOCK, 0, h->stream>>>(h->scratch, nblk, 1, dst);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" {

int hgibbs_w_init(hgibbs_t h, const int32_t* failure_host)
{
    if (!h || !h->bed || !failure_host) return fail("hgibbs_w_init: load genotypes first and pass the failure indicator");
    HIP_TRY(hipSetDevice(h->device));
    if (h->bw) return fail("hgibbs_w_init: already initialised on this handle");
    for (uint32_t i = 0; i < h->n_global; ++i)
        if (failure_host[i] != 0 && failure_host[i] != 1) return fail("hgibbs_w_init: failure indicator of individual %u is %d, not 0/1", i, failure_host[i]);
    BwState* b = new BwState();
    h->bw = b;
    b->fail.assign(failure_host, failure_host + h->n_global);
    b->d_total = 0.0;
    for (uint32_t i = 0; i < h->n_global; ++i) b->d_total += (double)failure_host[i];
    const uint32_t ndw = h->n_pad / 16;
    HIP_TRY(hipMalloc(&b->vi, (size_t)h->n_pad * sizeof(double)));
    HIP_TRY(hipMemsetAsync(b->vi, 0, (size_t)h->n_pad * sizeof(double), h->stream));
    HIP_TRY(hipMalloc(&b->vi2, (size_t)h->n_pad * sizeof(double)));
    HIP_TRY(hipMemsetAsync(b->vi2, 0, (size_t)h->n_pad * sizeof(double), h->stream));
    HIP_TRY(hipMalloc(&b->failspread, (size_t)ndw * sizeof(uint32_t)));
    HIP_TRY(hipMalloc(&b->vi_sum, sizeof(double)));
    HIP_TRY(hipMalloc(&b->d_mave, (size_t)h->M * sizeof(double)));
    HIP_TRY(hipMalloc(&b->d_sd, (size_t)h->M * sizeof(double)));
    HIP_TRY(hipMalloc(&b->d_sumfail, (size_t)h->M * sizeof(double)));
    HIP_TRY(hipMalloc(&b->d_unif, (size_t)h->M * sizeof(double)));
    HIP_TRY(hipMalloc(&b->partials, (size_t)S_CAP * BW_ROWS * sizeof(double)));
    HIP_TRY(hipHostMalloc(&b->h_result, sizeof(BwResult)));
    std::memset(b->h_result, 0, sizeof(BwResult));
    HIP_TRY(hipMalloc(&b->vipart, (size_t)(h->n_pad / BLOCK_IND) * sizeof(double)));
    HIP_TRY(hipMalloc(&b->d_colk, (MAX_BATCH + 1) * sizeof(int32_t)));
    HIP_TRY(hipMalloc(&b->d_rows, (size_t)BW_ROWS * sizeof(double)));
    HIP_TRY(hipMemsetAsync(b->d_rows, 0, (size_t)BW_ROWS * sizeof(double), h->stream));
    HIP_TRY(hipMalloc(&b->d_first_event, sizeof(uint32_t)));
    HIP_TRY(hipMemsetAsync(b->d_first_event, 0xff, sizeof(uint32_t), h->stream));
    HIP_TRY(hipEventCreate(&b->evk0));
    HIP_TRY(hipEventCreate(&b->evk1));
    HIP_TRY(hipMalloc(&b->d_picks, (MAX_BATCH + 1) * sizeof(int32_t)));
    HIP_TRY(hipMalloc(&b->d_colsums, (size_t)(MAX_BATCH + 1) * 3 * sizeof(double)));
    int32_t* d_fail = nullptr;
    HIP_TRY(hipMalloc(&d_fail, (size_t)h->n_local * sizeof(int32_t)));
    HIP_TRY(hipMemcpyAsync(d_fail, failure_host + h->row_begin, (size_t)h->n_local * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
    k_bw_failspread<<<(ndw + 255) / 256, 256, 0, h->stream>>>(d_fail, h->n_local, ndw, b->failspread);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipFree(d_fail));
    b->beta.assign(h->M, 0.0);
    b->comp.assign(h->M, 0);
    return 0;
}

/* src/BayesW.cpp:1201-1232: mave as in BayesR, sd = sqrt(sum of squares / (N-1)) (the standard
 * deviation, not its inverse), sum_failure = (sum_i g_ij d_i - mave * sum_i d_i) / sd.  The counts
 * are integer work on the device; the three tables are computed from them here, in the reference's
 * expression order, and kept on both sides. */
int hgibbs_w_marker_stats(hgibbs_t h, double* mave, double* sd, double* sum_failure)
{
    if (bw_need(h, "hgibbs_w_marker_stats")) return 1;
    HIP_TRY(hipSetDevice(h->device));
    BwState* b = h->bw;
    if (!b->have_tables) {
        if (compute_stats(h)) return 1;
        const uint32_t M = h->M;
        std::vector<unsigned long long> cnt((size_t)3 * M), fc((size_t)2 * M);
        HIP_TRY(hipMemcpy(cnt.data(), h->counts, cnt.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        unsigned long long* d_fc = nullptr;
        HIP_TRY(hipMalloc(&d_fc, fc.size() * sizeof(unsigned long long)));
        k_bw_fail_counts<<<M, BLOCK, 0, h->stream>>>(h->bed, h->stride, b->failspread, h->n_pad / 16, d_fc);
        HIP_TRY(hipGetLastError());
        if (bulk_allreduce(h, d_fc, fc.size(), 1)) return 1; // individuals are sharded: counts add over the ranks
        HIP_TRY(hipMemcpyAsync(fc.data(), d_fc, fc.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        HIP_TRY(hipFree(d_fc));
        b->mave.resize(M);
        b->sd.resize(M);
        b->sumfail.resize(M);
        const uint32_t Ntot = h->n_global;
        const double dN = (double)Ntot;
        for (uint32_t i = 0; i < M; ++i) {
            const unsigned long long n1 = cnt[3 * (size_t)i], n2 = cnt[3 * (size_t)i + 1], nm = cnt[3 * (size_t)i + 2];
            const double mv = ((double)n1 + 2.0 * (double)n2) / (dN - (double)nm);
            const double tmp1 = (double)n1 * (1.0 - mv) * (1.0 - mv);
            const double tmp2 = (double)n2 * (2.0 - mv) * (2.0 - mv);
            const double tmp0 = (double)(Ntot - n1 - n2 - nm) * (0.0 - mv) * (0.0 - mv);
            b->mave[i] = mv;
            b->sd[i] = std::sqrt((tmp0 + tmp1 + tmp2) / (double)(Ntot - 1));
            const int temp_sum = (int)(fc[2 * (size_t)i] + 2ull * fc[2 * (size_t)i + 1]);
            b->sumfail[i] = ((double)temp_sum - mv * b->d_total) / b->sd[i];
        }
        HIP_TRY(hipMemcpy(b->d_mave, b->mave.data(), (size_t)M * sizeof(double), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(b->d_sd, b->sd.data(), (size_t)M * sizeof(double), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(b->d_sumfail, b->sumfail.data(), (size_t)M * sizeof(double), hipMemcpyHostToDevice));
        b->have_tables = true;
    }
    if (mave) std::copy(b->mave.begin(), b->mave.end(), mave);
    if (sd) std::copy(b->sd.begin(), b->sd.end(), sd);
    if (sum_failure) std::copy(b->sumfail.begin(), b->sumfail.end(), sum_failure);
    return 0;
}

int hgibbs_w_set_model(hgibbs_t h, int G, int K, const int32_t* groups_host, const double* mS, int quad_points)
{
    if (bw_need(h, "hgibbs_w_set_model")) return 1;
    if (G < 1 || K < 2 || !mS) return fail("hgibbs_w_set_model: need G >= 1, K >= 2 and the mixture table");
    if (K > bw::MAX_K) return fail("hgibbs_w_set_model: K = %d components, at most %d", K, bw::MAX_K);
    const double *X = nullptr, *W = nullptr;
    if (!hg_gh_lookup(quad_points, &X, &W)) return fail("Possible number of quad_points = 3,5,7,9,11,13,15,17,25 (got %d)", quad_points);
    HIP_TRY(hipSetDevice(h->device));
    BwState* b = h->bw;
    h->groups_host.assign(h->M, 0);
    if (groups_host)
        for (uint32_t i = 0; i < h->M; ++i) {
            if (groups_host[i] < 0 || groups_host[i] >= G) return fail("hgibbs_w_set_model: marker %u in group %d outside [0,%d)", i, groups_host[i], G);
            h->groups_host[i] = groups_host[i];
        }
    HIP_TRY(hipMemcpy(h->groups, h->groups_host.data(), (size_t)h->M * sizeof(int32_t), hipMemcpyHostToDevice));
    b->G = G;
    b->K = K;
    b->quad = quad_points;
    b->cva.resize((size_t)G * (K - 1));
    for (int g = 0; g < G; ++g)
        for (int k = 1; k < K; ++k) {
            if (!(mS[(size_t)g * K + k] > 0.0)) return fail("hgibbs_w_set_model: mixture value can only be strictly positive");
            b->cva[(size_t)g * (K - 1) + (k - 1)] = mS[(size_t)g * K + k];
        }
    for (double** p : {&b->d_cva, &b->d_pi, &b->d_sigmaG, &b->d_ghx, &b->d_ghw})
        if (*p) {
            HIP_TRY(hipFree(*p));
            *p = nullptr;
        }
    HIP_TRY(hipMalloc(&b->d_cva, b->cva.size() * sizeof(double)));
    HIP_TRY(hipMalloc(&b->d_pi, (size_t)G * K * sizeof(double)));
    HIP_TRY(hipMalloc(&b->d_sigmaG, (size_t)G * sizeof(double)));
    HIP_TRY(hipMalloc(&b->d_ghx, (size_t)(quad_points - 1) * sizeof(double)));
    HIP_TRY(hipMalloc(&b->d_ghw, (size_t)quad_points * sizeof(double)));
    HIP_TRY(hipMemcpy(b->d_cva, b->cva.data(), b->cva.size() * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(b->d_ghx, X, (size_t)(quad_points - 1) * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(b->d_ghw, W, (size_t)quad_points * sizeof(double), hipMemcpyHostToDevice));
    b->have_model = true;
    return 0;
}

int hgibbs_w_reduce(hgibbs_t h, int kind, int col, double p0, double p1, double p2, double* out)
{
    if (bw_need(h, "hgibbs_w_reduce")) return 1;
    if (!out || kind < 0 || kind > 3) return fail("hgibbs_w_reduce: kind %d outside [0,3] or null out", kind);
    if (kind == 2 && (!h->covX || col < 0 || col >= h->C)) return fail("hgibbs_w_reduce: covariate %d outside [0,%d)", col, h->C);
    HIP_TRY(hipSetDevice(h->device));
    const uint32_t nblk = h->n_pad / BLOCK_IND;
    k_bw_reduce<<<nblk, BLOCK, 0, h->stream>>>(h->eps[h->eps_cur], kind == 2 ? h->covX + (size_t)col * h->n_pad : nullptr, h->bw->failspread,
                                               kind, p0, p1, p2, h->n_local, h->scratch);
    if (bw_final(h, nblk, h->sums)) return 1;
    if (bulk_allreduce(h, h->sums, 1, 0)) return 1; // every rank gets the same bits: the ARS decisions stay replicated
    HIP_TRY(hipMemcpyAsync(h->scratch_host, h->sums, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    *out = h->scratch_host[0];
    return 0;
}

int hgibbs_w_refresh_vi(hgibbs_t h, double alpha)
{
    if (bw_need(h, "hgibbs_w_refresh_vi")) return 1;
    HIP_TRY(hipSetDevice(h->device));
    const uint32_t nblk = h->n_pad / BLOCK_IND;
    k_bw_refresh<<<nblk, BLOCK, 0, h->stream>>>(h->bed, h->stride, -1, 0.0, 0.0, 0.0, h->eps[h->eps_cur], h->bw->vi, alpha, h->n_local, h->bw->vipart);
    HIP_TRY(hipGetLastError());
    return 0;
}

int hgibbs_w_get_vi(hgibbs_t h, double* vi_host, double* vi_sum)
{
    if (bw_need(h, "hgibbs_w_get_vi")) return 1;
    HIP_TRY(hipSetDevice(h->device));
    if (vi_host) {
        double* tmp = nullptr;
        HIP_TRY(hipMalloc(&tmp, (size_t)h->n_local * sizeof(double)));
        k_get_eps<<<(h->n_local + 255) / 256, 256, 0, h->stream>>>(h->bw->vi, tmp, h->n_local);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(vi_host, tmp, (size_t)h->n_local * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        HIP_TRY(hipFree(tmp));
    }
    if (vi_sum) {
        k_final_sum<<<1, BLOCK, 0, h->stream>>>(h->bw->vipart, h->n_pad / BLOCK_IND, 1, h->bw->vi_sum);
        HIP_TRY(hipGetLastError());
        if (bulk_allreduce(h, h->bw->vi_sum, 1, 0)) return 1;
        HIP_TRY(hipMemcpyAsync(vi_sum, h->bw->vi_sum, sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
    }
    return 0;
}

int hgibbs_w_get_beta(hgibbs_t h, double* beta, int32_t* components)
{
    if (bw_need(h, "hgibbs_w_get_beta")) return 1;
    if (beta) std::copy(h->bw->beta.begin(), h->bw->beta.end(), beta);
    if (components) std::copy(h->bw->comp.begin(), h->bw->comp.end(), components);
    return 0;
}

int hgibbs_w_set_beta(hgibbs_t h, const double* beta, const int32_t* components)
{
    if (bw_need(h, "hgibbs_w_set_beta")) return 1;
    if (beta) h->bw->beta.assign(beta, beta + h->M);
    if (components) h->bw->comp.assign(components, components + h->M);
    return 0;
}

void hgibbs_grand_seed(hgibbs_grand_state* st, uint32_t seed)
{
    static_assert(sizeof(hgibbs_grand_state) == sizeof(hg::GlibcRand), "layout");
    reinterpret_cast<hg::GlibcRand*>(st)->seed(seed);
}

int32_t hgibbs_grand_next(hgibbs_grand_state* st) { return reinterpret_cast<hg::GlibcRand*>(st)->next(); }

int hgibbs_ars_sample(const double* xinit4, double xl, double xr, double (*logdens)(double, void*), void* data, hgibbs_grand_state* rng,
                      double* xsamp, int* neval)
{
    if (!xinit4 || !logdens || !rng || !xsamp) return fail("hgibbs_ars_sample: null argument");
    struct F {
        double (*f)(double, void*);
        void* d;
        double operator()(double x) { return f(x, d); }
    } f{logdens, data};
    struct U {
        hg::GlibcRand* g;
        double operator()() { return g->uniform(); }
    } u{reinterpret_cast<hg::GlibcRand*>(rng)};
    const double xi[4] = {xinit4[0], xinit4[1], xinit4[2], xinit4[3]};
    hg::ars::Hull hull;
    int ne = 0;
    const int err = hg::ars::sample(xi, xl, xr, f, u, hull, *xsamp, ne);
    if (neval) *neval = ne;
    if (err) fail("hgibbs_ars_sample: error %d (1003 bounds, 1004 order, 2000 log density not concave)", err);
    return err;
}

/* One pass over all markers, src/BayesW.cpp:1461-1622.  The effect/component vectors live on the
 * host side of the handle (hgibbs_w_get_beta); eps and vi on the device.  rng: dist.rng, one
 * uniform per marker; ars_rng: the libc rand() stream of the ARS draws. */
int hgibbs_w_sweep(hgibbs_t h, const int32_t* order_host, double alpha, const double* sigmaG, const double* pi, double sumSigmaG,
                   hgibbs_rng_state* rng, hgibbs_grand_state* ars_rng, int32_t* cass_host, double* beta_sqnorm, uint64_t* nnz_updates)
{
    if (bw_need(h, "hgibbs_w_sweep", true, true)) return 1;
    if (!order_host || !sigmaG || !pi || !rng || !ars_rng || !cass_host || !beta_sqnorm) return fail("hgibbs_w_sweep: null argument");
    HIP_TRY(hipSetDevice(h->device));
    BwState* b = h->bw;
    const uint32_t M = h->M;
    const int G = b->G, K = b->K;
    for (uint32_t i = 0; i < M; ++i)
        if (order_host[i] < 0 || (uint32_t)order_host[i] >= M) return fail("hgibbs_w_sweep: order[%u] = %d outside [0,%u)", i, order_host[i], M);

    // one Boost uniform per marker, in sweep order (src/BayesW.cpp:1528)
    {
        hg::Mt gen{rng->x, rng->idx};
        b->unif.resize(M);
        for (uint32_t j = 0; j < M; ++j) b->unif[j] = hg::unif01(gen);
        rng->idx = gen.idx;
    }
    HIP_TRY(hipMemcpyAsync(b->d_unif, b->unif.data(), (size_t)M * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->order, order_host, (size_t)M * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(b->d_pi, pi, (size_t)G * K * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(b->d_sigmaG, sigmaG, (size_t)G * sizeof(double), hipMemcpyHostToDevice, h->stream));
    std::fill(cass_host, cass_host + (size_t)G * K, 0);
    std::fill(beta_sqnorm, beta_sqnorm + G, 0.0);

    const uint32_t ntg = h->n_pad / BLOCK_IND, nblk = ntg;
    const uint32_t batch_cap = std::min<uint32_t>(h->batch ? h->batch : MAX_BATCH, MAX_BATCH);
    constexpr int CPG = 8;
    uint64_t nnz = 0, launches = 0, ars_draws = 0, ars_evals = 0;
    double sums_ms = 0.0;
    HIP_TRY(hipEventRecord(h->ev0, h->stream));
    uint32_t cursor = 0;
    hg::GlibcRand* gr = reinterpret_cast<hg::GlibcRand*>(ars_rng);
    // the last event's update of eps and vi, applied by the next batch's launch (or flushed behind the loop)
    int32_t pend_marker = -1;
    double pend_dv[3] = {0.0, 0.0, 0.0};
    while (cursor < M) {
        // columns with a zero effect, then (if it comes within reach) the first marker whose effect is not zero
        uint32_t ncols = 0;
        int32_t shifted = -1;
        while (cursor + ncols < M && ncols < batch_cap) {
            const int32_t mk = order_host[cursor + ncols];
            if (b->beta[mk] != 0.0) {
                shifted = mk;
                break;
            }
            ++ncols;
        }
        const uint32_t nb = ncols + (shifted >= 0 ? 1u : 0u);
        BwBatchParams p{};
        p.bed = h->bed;
        p.stride = h->stride;
        p.eps = h->eps[h->eps_cur];
        p.vi = b->vi;
        p.n_pad = h->n_pad;
        p.n_local = h->n_local;
        p.markers = h->order + cursor;
        p.ncols = ncols;
        p.shifted_marker = shifted;
        if (shifted >= 0) {
            double dv[3];
            bw::delta_values(b->beta[shifted], b->mave[shifted], b->sd[shifted], dv);
            p.dv[0] = dv[0];
            p.dv[1] = dv[1];
            p.dv[2] = dv[2];
        }
        p.alpha = alpha;
        p.pend_marker = pend_marker;
        for (int c = 0; c < 3; ++c) {
            p.pend_dv[c] = pend_dv[c];
            p.pend_f[c] = std::exp(alpha * pend_dv[c]);
        }
        p.eps_out = h->eps[h->eps_cur ^ 1u];
        p.vi_out = b->vi2;
        p.vipart_out = b->vipart;
        // the update pass costs about four streaming tiles per tile (sixteen exponentials per lane): it gets four groups of workgroups
        p.upd_groups = 4u;
        const uint32_t ngroups = (ncols + CPG - 1) / CPG + (shifted >= 0 ? 1u : 0u) + (pend_marker >= 0 ? p.upd_groups : 0u);
        uint32_t S = h->slices ? h->slices : S_CAP;
        S = std::min<uint32_t>(std::min<uint32_t>(S, S_CAP), ntg);
        S = std::max<uint32_t>(1u, std::min<uint32_t>(S, 768u / ngroups));
        p.slices = S;
        p.partials = b->partials;
        p.ticket = h->ticket;
        p.p_unif = b->d_unif + cursor;
        p.mave = b->d_mave;
        p.sd = b->d_sd;
        p.sumfail = b->d_sumfail;
        p.groups = h->groups;
        p.cva = b->d_cva;
        p.pi = b->d_pi;
        p.sigmaG = b->d_sigmaG;
        p.K = K;
        p.ghx = b->d_ghx;
        p.ghw = b->d_ghw;
        p.quad = b->quad;
        p.vipart = b->vipart;
        p.n_vipart = nblk;
        p.rows = (h->nranks > 1) ? b->d_rows : nullptr;
        p.result = b->h_result;
        p.seq = ++b->seq;
        p.picks = b->d_picks;
        p.col_k = b->d_colk;
        p.col_sums = b->d_colsums;
        p.first_event = b->d_first_event;
        if (h->w_kernel_timing) HIP_TRY(hipEventRecord(b->evk0, h->stream));
        k_bw_sums<CPG><<<S * ngroups, BLOCK, 0, h->stream>>>(p);
        if (h->w_kernel_timing) HIP_TRY(hipEventRecord(b->evk1, h->stream));
        if (pend_marker >= 0) { // the launch wrote the other eps / vi buffers: they are the current ones from here on
            h->eps_cur ^= 1u;
            std::swap(b->vi, b->vi2);
            pend_marker = -1;
        }
        if (p.rows) { // individuals sharded: the row sums add over the ranks, same bits everywhere
            k_bw_rows<<<nb + 1, WAVE, 0, h->stream>>>(p);
            HIP_TRY(hipGetLastError());
            if (bulk_allreduce(h, b->d_rows, BW_ROWS, 0)) return 1;
        }
        k_bw_tail<<<nb, WAVE, 0, h->stream>>>(p);
        HIP_TRY(hipGetLastError());
        ++launches;
        // the tail writes the result into pinned memory, the sequence number last: poll it, and fall
        // back to the stream's completion so that a failed launch cannot hang the host
        {
            volatile unsigned long long* seqp = &b->h_result->seq;
            uint32_t spins = 0;
            while (*seqp != p.seq) {
                if ((++spins & 0x3ffu) == 0) {
                    const hipError_t q = hipStreamQuery(h->stream);
                    if (q == hipSuccess) break;
                    if (q != hipErrorNotReady) return fail("hgibbs_w_sweep: %s", hipGetErrorString(q));
                }
            }
            std::atomic_thread_fence(std::memory_order_acquire);
            if (*seqp != p.seq) {
                HIP_TRY(hipStreamSynchronize(h->stream));
                if (*seqp != p.seq) return fail("hgibbs_w_sweep: the batch result never arrived (seq %llu, expected %llu)", *seqp, p.seq);
            }
        }
        if (h->w_kernel_timing) {
            HIP_TRY(hipEventSynchronize(b->evk1));
            float kms = 0.f;
            HIP_TRY(hipEventElapsedTime(&kms, b->evk0, b->evk1));
            sums_ms += kms;
        }
        const BwResult r = *b->h_result;
        if (r.event > nb) return fail("hgibbs_w_sweep: device reported event %u in a batch of %u", r.event, nb);
        // markers before the event keep a zero effect (src/BayesW.cpp:1541-1546)
        for (uint32_t t = 0; t < std::min(r.event, ncols); ++t) {
            const int32_t mk = order_host[cursor + t];
            cass_host[(size_t)h->groups_host[mk] * K] += 1;
            b->comp[mk] = 0;
        }
        if (r.event == nb) {
            cursor += nb;
            continue;
        }
        const int32_t mk = order_host[cursor + r.event];
        const int grp = h->groups_host[mk];
        const double beta_old = b->beta[mk];
        if (r.k < 0) return fail("hgibbs_w_sweep: marginal likelihoods of marker %d are not finite", mk);
        double beta_new = 0.0;
        if (r.k > 0) { // src/BayesW.cpp:1548-1590
            bw::BetaLogDensity f{alpha, sigmaG[grp], b->sumfail[mk], b->sd[mk], b->mave[mk] / b->sd[mk], b->cva[(size_t)grp * (K - 1) + (r.k - 1)],
                                 r.vi_sum - r.vi_1 - r.vi_2, r.vi_1, r.vi_2};
            const double safe_limit = 2 * std::sqrt(sumSigmaG * f.mixture_value);
            const double xinit[4] = {beta_old - safe_limit / 10, beta_old, beta_old + safe_limit / 20, beta_old + safe_limit / 10};
            struct U {
                hg::GlibcRand* g;
                double operator()() { return g->uniform(); }
            } u{gr};
            hg::ars::Hull hull;
            int ne = 0;
            const int err = hg::ars::sample(xinit, beta_old - safe_limit, beta_old + safe_limit, f, u, hull, beta_new, ne);
            ++ars_draws;
            ars_evals += (uint64_t)ne;
            if (err) return fail("Error code = %d (ARS on the effect of marker %d)", err, mk);
            beta_sqnorm[grp] += beta_new * beta_new;
        }
        cass_host[(size_t)grp * K + r.k] += 1;
        b->comp[mk] = r.k;
        b->beta[mk] = beta_new;
        const double deltaBeta = beta_old - beta_new;
        if (deltaBeta != 0.0) { // src/BayesW.cpp:1606-1622, :1812, :1832-1834
            bw::delta_values(deltaBeta, b->mave[mk], b->sd[mk], pend_dv);
            pend_marker = mk; // rides in the next batch's launch
            ++nnz;
        }
        cursor += r.event + 1;
    }
    if (pend_marker >= 0) { // the sweep's last event: its pass on its own
        k_bw_refresh<<<nblk, BLOCK, 0, h->stream>>>(h->bed, h->stride, pend_marker, pend_dv[0], pend_dv[1], pend_dv[2], h->eps[h->eps_cur], b->vi, alpha,
                                                    h->n_local, b->vipart);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipEventRecord(h->ev1, h->stream));
    HIP_TRY(hipEventSynchronize(h->ev1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    if (nnz_updates) *nnz_updates = nnz;
    b->stats.launches = launches;
    b->stats.nnz_updates = nnz;
    b->stats.ars_draws = ars_draws;
    b->stats.ars_evals = ars_evals;
    b->stats.device_ms = ms;
    b->stats.sums_kernel_ms = sums_ms;
    return 0;
}

/* the ARS draw on one device lane, timed (diagnostic; see include/hgibbs.h) */
int hgibbs_w_ars_device_probe(hgibbs_t h, const double* dens9, double beta_old, double safe_limit, uint32_t seed, uint32_t ndraws, double* us_per_draw,
                              double* evals_per_draw, double* last_draw)
{
    if (!h || !dens9 || ndraws == 0) return fail("hgibbs_w_ars_device_probe: null argument");
    HIP_TRY(hipSetDevice(h->device));
    if (ensure_scratch(h, 64)) return 1;
    BwArsProbe pr{};
    for (int i = 0; i < 9; ++i) pr.d[i] = dens9[i];
    pr.beta_old = beta_old;
    pr.safe_limit = safe_limit;
    pr.seed = seed;
    pr.ndraws = ndraws;
    pr.out = h->scratch;
    k_bw_ars_probe<<<1, WAVE, 0, h->stream>>>(pr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(h->scratch_host, h->scratch, 4 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->scratch_host[3] != 0.0) return fail("hgibbs_w_ars_device_probe: ARS error code %d on the device", (int)h->scratch_host[3]);
    if (us_per_draw) *us_per_draw = h->scratch_host[0] / 100.0 / (double)ndraws; // 100 MHz wall clock
    if (evals_per_draw) *evals_per_draw = h->scratch_host[1] / (double)ndraws;
    if (last_draw) *last_draw = h->scratch_host[2];
    return 0;
}

int hgibbs_w_last_sweep_stats(hgibbs_t h, hgibbs_w_sweep_stats* out)
{
    if (bw_need(h, "hgibbs_w_last_sweep_stats") || !out) return 1;
    *out = h->bw->stats;
    return 0;
}

/* the per-marker operator on its own: the three masked sums of marker j as the sweep would see them */
int hgibbs_w_marker_sums(hgibbs_t h, uint32_t marker, double beta_old, double alpha, double* vi_sum, double* vi_1, double* vi_2)
{
    if (bw_need(h, "hgibbs_w_marker_sums", true)) return 1;
    if (marker >= h->M) return fail("hgibbs_w_marker_sums: marker %u >= M %u", marker, h->M);
    HIP_TRY(hipSetDevice(h->device));
    BwState* b = h->bw;
    const uint32_t ntg = h->n_pad / BLOCK_IND;
    const uint32_t nblk = ntg;
    // shifted form reads eps; plain form reads vi
    double dv[3] = {0.0, 0.0, 0.0};
    if (beta_old != 0.0) bw::delta_values(beta_old, b->mave[marker], b->sd[marker], dv);
    double* part = h->scratch; // 3 rows per block
    k_bw_marker_sums<<<nblk, BLOCK, 0, h->stream>>>(h->bed, h->stride, marker, h->eps[h->eps_cur], b->vi, beta_old != 0.0, dv[0], dv[1], dv[2], alpha,
                                       h->n_local, part);
    k_final_sum<<<1, BLOCK, 0, h->stream>>>(part, nblk, 3, h->sums);
    HIP_TRY(hipGetLastError());
    if (bulk_allreduce(h, h->sums, 3, 0)) return 1;
    HIP_TRY(hipMemcpyAsync(h->scratch_host, h->sums, 3 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (vi_sum) *vi_sum = h->scratch_host[0];
    if (vi_1) *vi_1 = h->scratch_host[1];
    if (vi_2) *vi_2 = h->scratch_host[2];
    return 0;
}

} // extern "C"
