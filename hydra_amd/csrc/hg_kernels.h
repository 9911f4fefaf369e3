// hg_kernels.h -- hand-written HIP kernels (gfx950 / CDNA4, wave64) for the
// BayesRR per-marker hot path.  No MFMA: the path is a byte-decode + masked
// fp64 reduction, bound by HBM/L2 traffic and launch/hand-off latency.
//
// Data layout in HBM (per GPU, individuals shard [row_begin,row_end)):
//   bed    M columns x stride bytes, stride = n_pad/4, n_pad = N_local rounded
//          up to 4096 (one 256-thread block = 4 wave tiles of 1024 individuals).
//          Same 2 bits per individual at the same position as the PLINK file, but
//          RE-CODED once at load (k_recode_bed) so that the field IS the weight of
//          the dot product: 00 -> genotype 0, 01 -> 1, 10 -> 2, 11 -> missing
//          (PLINK's own codes, src/data.cpp:1189-1200, are 11 -> 0, 10 -> 1, 00 -> 2,
//          01 -> missing: new_hi = ~hi, new_lo = hi ^ lo, an involution away from the
//          file's bytes; hgibbs_get_bed returns the file's codes).  The streaming loop
//          then needs no code -> weight conversion at all, and the integer Gram terms
//          come from eight bit operations per dword.  Padding slots hold 11 (missing),
//          so they contribute to no sum and receive no update.
//   eps    2 x n_pad doubles (double-buffered: a launch that applies a pending
//          update reads buffer `cur` and writes `cur^1`), stored PERMUTED so
//          that the lane that holds column dword l of a wave tile (individuals
//          16l..16l+15) reads its 16 residuals with 8 fully coalesced 16-byte
//          loads:  slot s of lane l  ->  tile*1024 + (s>>1)*128 + l*2 + (s&1).
//   beta/components/acum  M-sized, replicated on every rank.
//
// Kernels
//   k_sweep_batch   the hot kernel (hg_sweep.hip.h).  One launch = a speculative batch of up to 256
//                   markers of the shuffled order, a chain of segments that end on predicted events.  Its workgroups are
//                   the update group (applies the previous launch's events to eps, stores the other eps buffer),
//                   Gram-only groups (carried columns: integer Gram terms with the pending columns and this launch's
//                   pivots instead of a second pass over eps) and fresh groups (s1 = sum g nm eps against eps + pending
//                   updates, s2 for columns with missing calls, Gram terms with the pivots).  The last workgroup of a
//                   group sums its rows over the slices in fixed order; the last group's evaluates the mixture posterior
//                   of a segment's markers in parallel, then consumes the shared MT19937 stream in marker order and accepts
//                   markers up to and including the first one whose effect changes -- later dots are stale: corrected
//                   through the Gram terms where the event was a planned pivot, else handed to the next launch as
//                   carried columns.  Exactly the sequential chain: every accepted marker saw the eps the reference's
//                   marker loop would have given it (src/BayesRRm.cpp:1709-2025).
//   k_* helpers     stats, permuted get/set, scalar add, reductions, synthetic
//                   genotypes, single-marker dot/update.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hg_rng.h"

namespace hg {

constexpr int WAVE = 64;
constexpr int IPT = 16;                  // individuals per lane per tile (one column dword)
constexpr int TILE = WAVE * IPT;         // 1024 individuals per wave tile
constexpr int BLOCK_WAVES = 4;
constexpr int BLOCK = WAVE * BLOCK_WAVES;       // 256 threads
constexpr int BLOCK_IND = TILE * BLOCK_WAVES;   // 4096 individuals per block
constexpr int MAX_BATCH = 256;           // speculative batch width upper bound (one thread per column in the draw)
constexpr int MAX_CPG = 16;              // columns per workgroup column-group (register accumulators)
constexpr int S_CAP = 64;                // max tile-group slices (gridDim.x): one partial row = 64 doubles
constexpr int MAX_K = 8;                 // mixture components incl. zero
constexpr int HT_LDS = 64;               // hyper tables are staged in LDS when G*K <= this
constexpr int NSUM = 2;                  // sums per batch column: s1 = sum g*nm*eps, s2 = sum nm*eps
constexpr int MAX_SEG = 4;               // segments per launch: each ends on a predicted event (its pivot) and hands one pending update on
constexpr int NROW = NSUM + 4;           // most rows per batch column in `totals`: s1, s2 and one Gram term per earlier pivot (MAX_SEG - 1) or, in the
                                         // two-segment build that takes missing calls through the extension, the four terms A, B, C, D
// rows per batch column of a kernel build in `totals`: s1 (for a carried column: its Gram correction G_j), s2, then per earlier pivot one
// Gram term, or A, B, C, D
__host__ __device__ constexpr int sweep_rows(int seg, int mg) { return NSUM + (seg - 1) * (mg ? 4 : 1); }
// carried columns per Gram-only workgroup, and the rows such a group publishes per column (per pending update and per earlier pivot)
// (8 where a column takes many terms -- four sums per term, or the four-segment build's seven terms: registers)
__host__ __device__ constexpr int carried_cpg(int seg, int mg) { return (mg || seg > 2) ? 8 : 16; }
__host__ __device__ constexpr int carried_rows(int seg, int mg) { return (seg + seg - 1) * (mg ? 4 : 1); }
// row block of one group (slices x this many rows) in `partials`
__host__ __device__ constexpr int group_rows(int cpg, int seg, int mg)
{
    return carried_cpg(seg, mg) * carried_rows(seg, mg) > cpg * sweep_rows(seg, mg) ? carried_cpg(seg, mg) * carried_rows(seg, mg) : cpg * sweep_rows(seg, mg);
}
constexpr int MAX_GROUPS = 1 + MAX_BATCH / 8 + MAX_BATCH / 2; // update group + Gram-only groups + fresh groups at their smallest sizes
constexpr int PROWS_CAP = 10240;         // partial rows per slice: groups x group_rows of any build (checked on the host)
constexpr int ROWS_CAP = NROW * MAX_BATCH + 8; // rows of `totals` (padded)
constexpr int AHEAD_MAX = 256;           // most columns a launch streams ahead of its batch
constexpr int MAX_RANKS = 16;             // GPUs of one node that can share the in-launch exchange
constexpr int MT_BUF = 2 * MT_N;         // current + next MT19937 block staged in LDS

struct SweepDesc {
    uint32_t cursor;        // next position in order[]
    uint32_t cur;           // which eps buffer is current
    int32_t pend_marker[MAX_SEG]; // markers whose eps update is still to be applied (in order), -1 none
    double pv[MAX_SEG][3];        // their update constants for genotype 0,1,2 (missing gets 0)
    uint32_t seg_end[MAX_SEG];    // plan of the next launch: segment s = columns [seg_end[s-1], seg_end[s]); a segment that is
                                  // followed by another one ends ON a predicted event (its pivot); later segments' dots get the
                                  // Gram correction for every earlier pivot
    uint32_t rng_idx;       // MT19937 position (0..624)
    uint32_t error;         // non-zero: 1 logL overflow abort (src/BayesRRm.cpp:1910-1913), 2 rng staging overrun, 3 peer timeout, 4 LDS base not on a 256-byte boundary
    uint64_t seq;           // batches since the handle was created (epoch of the cross-GPU exchange)
    // carried dots: the first carry_n columns of the next batch were already streamed by this launch (they lay behind the
    // event that ended it); their dots against THIS launch's residual are in SweepParams::carry, and the next launch only
    // takes their integer Gram terms with its pending columns (this launch's events)
    uint32_t carry_n;
    uint32_t carry_left;    // the first carry_left of them were left over by this launch's walk (dots in SweepParams::carry); the rest it
                            // streamed ahead (raw sums in SweepParams::ahead_raw)
    double pend_ev[MAX_SEG][3]; // (dbeta, mave, mstd) of the pending updates: what the Gram correction of a carried dot needs
};

// Counters of a sweep, behind the descriptor in memory and apart from it, so that the descriptor the draw phase hands to the next
// launch is written from registers without a load in front of it.  One thread of the launch's last workgroup updates them with a
// plain read-modify-write AFTER the hand-over (launches of a sweep are serial on the stream: no other writer).
struct SweepCounters {
    unsigned long long nnz;           // markers with deltaBeta != 0 so far
    unsigned long long launches;      // launches that did work
    unsigned long long accepted_sum;  // total accepted markers (== cursor at the end)
    unsigned long long carried_sum;   // columns carried so far
    unsigned long long streamed_sum;  // batch columns whose dot was streamed (carried ones not, ahead ones included)
    uint32_t tiles_min;     // fewest / most tile groups (4096 individuals) one workgroup streamed in a working launch of this sweep:
    uint32_t tiles_max;     // > 1 means the loop's next-tile prefetch and the accumulation across tiles ran
};

// In-launch cross-GPU exchange (xGMI peer mailboxes, IPC-mapped).  Rank r's
// mailbox holds, per parity, one row block and one flag word per source rank.
struct P2PParams {
    int nranks;                              // 0/1 = disabled
    int rank;
    double* data[MAX_RANKS];                 // data[r]  = base of rank r's mailbox rows [2][MAX_RANKS][ROWS_CAP]
    unsigned long long* flags[MAX_RANKS];    // flags[r] = base of rank r's flags      [2][MAX_RANKS]
};

struct SweepParams {
    // data
    const uint8_t* bed;
    uint64_t stride;       // bytes per column
    double* eps0;
    double* eps1;
    uint32_t n_pad;        // padded local individuals
    uint32_t n_local;      // individuals of this shard (slots [n_local, n_pad) are padding: code 11, eps 0)
    uint32_t M;
    double n_minus_1;      // (double)(N_global - 1)
    double n_total;        // (double)N_global
    double eps_sum;        // sum of eps over ALL ranks at sweep start = s2 of every column without missing calls, for the whole sweep:
                           // a marker update leaves it unchanged (columns are centred over their non-missing calls), see hgibbs_sweep
    int gram;              // 1: extend batches past the first predicted event with Gram-corrected dots
    const int32_t* order;
    // per-marker metadata gathered into SWEEP order once per sweep (k_gather_meta)
    const double* s_mave;
    const double* s_mstd;
    const double* s_bold;
    const int32_t* s_ga;   // group | adaV << 30 | has-missing-calls << 29
    // effects
    double* beta;
    int32_t* comp;
    double* acum;
    int32_t* cass;         // G*K counters
    // hyper tables, G*K each (column 0 of denom/h/sd unused)
    int K;
    int GK;                // G*K
    const double* denom;   // (N-1) + sigmaE/sigmaG[g]*cVaI[g,k]
    const double* logpi;   // log(estPi[g,k])
    const double* hlog;    // 0.5*log(sigmaG[g]/sigmaE*(N-1)*cVa[g,k] + 1)
    const double* sdk;     // sqrt(sigmaE/denom[g,k])
    double i_2sigE;        // 1/(2 sigmaE)
    // rng
    uint32_t* mt;          // 624 words
    ZigTables zig;
    // hand-off
    SweepDesc* desc;
    SweepCounters* counters;
    double* carry;         // [MAX_BATCH] dots of the carried columns (written by the draw phase, read by the next one)
    uint32_t carry_on;     // 1: launches may hand dots of already streamed columns to the next one
    uint32_t ahead_cols;   // columns a launch streams ahead of its batch while its last workgroup draws (0: none)
    double* ahead_raw;     // [2][AHEAD_MAX][2]: (s1, s2) of the columns streamed ahead, by launch parity
    double* apartials;     // [S_CAP][2 * AHEAD_MAX]: their per-slice rows
    uint32_t* aticket;     // [AHEAD_MAX / 2]: per ahead group, slices that have stored their rows
    uint32_t* aqueue;      // [2]: next item of the ahead queue, by launch parity
    double* partials;      // [S_CAP][PROWS_CAP]: per slice, one row block per group, written sc1
    double* totals;        // [ROWS_CAP]: rows summed over the slices, published per column group
    uint32_t* ticket;      // groups that have published their totals
    uint32_t* gticket;     // [MAX_GROUPS]: per group, workgroups that have stored their partials
    uint32_t* entered;     // [8] words 32 apart: workgroups of the current launch that have read its descriptor (the drawing workgroup publishes
                           // the next one only when all of them have: a workgroup dispatched late must not see the next launch's plan)
    uint32_t cols_per_group; // columns handled per blockIdx.y
    uint32_t batch_cap;      // gridDim.y * cols_per_group (LDS carve-up)
    uint32_t batch_limit;    // widest batch a launch may take (the batch option)
    uint32_t slices_max;     // most tile-group slices a launch may use (<= S_CAP)
    uint32_t resident;       // workgroups of this kernel build that are co-resident on the device (registers and LDS)
    uint32_t ext_limit;      // longest Gram-corrected extension past the first pivot
    uint32_t max_seg;        // segments a launch may plan (1..MAX_SEG)
    // multi-GPU: when non-null the kernel stops after the local reduction and
    // leaves sums[NROW*MAX_BATCH+1] for the all-reduce; k_sweep_draw continues.
    double* sums_out;
    unsigned long long* dbg; // optional stage timestamps (wall_clock64, 100 MHz)
    P2PParams p2p;
};

// ---------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------
// 64-lane sum on the VALU's DPP path (no LDS round trips): quad swaps, row
// mirrors, then row broadcasts; fixed tree, total valid in lane 63 and returned
// wave-uniform.  (DPP only moves 32 bits, so each step moves both halves.)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xF, false);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wave_sum_lane63(double v)
{
    v += dpp_f64<0xB1, 0xF>(v);  // quad_perm [1,0,3,2]
    v += dpp_f64<0x4E, 0xF>(v);  // quad_perm [2,3,0,1]
    v += dpp_f64<0x141, 0xF>(v); // row_half_mirror
    v += dpp_f64<0x140, 0xF>(v); // row_mirror: every lane of a 16-row holds the row total
    v += dpp_f64<0x142, 0xA>(v); // row_bcast15 into rows 1,3
    v += dpp_f64<0x143, 0xC>(v); // row_bcast31 into rows 2,3
    return v;                    // lane 63 = total
}

__device__ __forceinline__ double wave_sum(double v)
{
    v = wave_sum_lane63(v);
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double mask_f64(double e, int m /* 0 or -1 */)
{
    long long b = __double_as_longlong(e);
    b &= (long long)m; // sign-extended 0 / all-ones
    return __longlong_as_double(b);
}

// position of (tile-local lane l, slot s) in the permuted eps layout
__device__ __host__ __forceinline__ uint32_t eps_pos(uint32_t i)
{
    uint32_t tile = i >> 10, r = i & 1023u;
    uint32_t l = r >> 4, s = r & 15u;
    return (tile << 10) + ((s >> 1) << 7) + (l << 1) + (s & 1u);
}

// Load the 16 residuals of this lane for wave tile `tile` (permuted layout).
__device__ __forceinline__ void load_eps16(const double* __restrict__ eps, uint32_t tile, int lane, double (&e)[IPT])
{
    const double2* p = reinterpret_cast<const double2*>(eps + ((size_t)tile << 10)) + lane;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        double2 v = p[k * 64];
        e[2 * k] = v.x;
        e[2 * k + 1] = v.y;
    }
}

__device__ __forceinline__ void store_eps16(double* __restrict__ eps, uint32_t tile, int lane, const double (&e)[IPT])
{
    double2* p = reinterpret_cast<double2*>(eps + ((size_t)tile << 10)) + lane;
#pragma unroll
    for (int k = 0; k < 8; ++k) p[k * 64] = make_double2(e[2 * k], e[2 * k + 1]);
}

// device genotype codes (see the layout note at the top): the 2-bit field is the genotype, 3 = missing call
constexpr uint32_t GC_G0 = 0u, GC_G1 = 1u, GC_G2 = 2u, GC_MISS = 3u;

// PLINK byte <-> device byte (four 2-bit fields each): new_hi = ~hi, new_lo = hi ^ lo; the inverse is hi = ~new_hi, lo = hi ^ new_lo
__host__ __device__ __forceinline__ uint32_t recode_plink_to_device(uint32_t w)
{
    const uint32_t h = (w >> 1) & 0x55555555u, l = w & 0x55555555u;
    return ((~h & 0x55555555u) << 1) | (h ^ l);
}
__host__ __device__ __forceinline__ uint32_t recode_device_to_plink(uint32_t w)
{
    const uint32_t nh = (w >> 1) & 0x55555555u, nl = w & 0x55555555u;
    const uint32_t h = ~nh & 0x55555555u;
    return (h << 1) | (h ^ nl);
}

// bit-plane masks of one column dword (16 codes): bit 2s set iff code s is ...
__device__ __forceinline__ void code_masks(uint32_t w, uint32_t& m1, uint32_t& m2, uint32_t& mm)
{
    const uint32_t hi = w >> 1, lo = w;
    m1 = lo & ~hi & 0x55555555u;   // code 01 -> genotype 1
    m2 = hi & ~lo & 0x55555555u;   // code 10 -> genotype 2
    mm = hi & lo & 0x55555555u;    // code 11 -> missing
}

// weights of one column dword (16 codes), 2-bit fields: gw = g*nm in {0,1,2}
// (what dotp_lut_a*dotp_lut_b tabulate), nm = non-missing in {0,1} (dotp_lut_b).
// For a column without missing calls gw == w: the loop uses the loaded dword as it is.
__device__ __forceinline__ void code_weights(uint32_t w, uint32_t& gw, uint32_t& nm)
{
    const uint32_t mm = w & (w >> 1) & 0x55555555u; // code 11
    gw = w & ~(mm | (mm << 1));
    nm = ~mm & 0x55555555u;
}

// a8 on registers: eps_s += {v0,v1,v2,0}[g_s]
__device__ __forceinline__ void apply_update16(uint32_t w, double v0, double v1, double v2, double (&e)[IPT])
{
#pragma unroll
    for (int s = 0; s < IPT; ++s) {
        uint32_t c = (w >> (2 * s)) & 3u;
        double v = (c == GC_G0) ? v0 : ((c == GC_G1) ? v1 : ((c == GC_G2) ? v2 : 0.0));
        e[s] = e[s] + (0.0 + v);
    }
}

// LDS reads the compiler must not see.  The streaming loops keep LDS-DMA (global_load_lds) and register loads in flight
// across iterations and wait on exact vmcnt values; the compiler cannot tell which LDS bytes a DMA writes, so before any
// LDS read IT emits it waits for vmcnt(0) -- every load in flight, the prefetched tiles included.  Reads issued from inline
// assembly are invisible to that rule; their lgkmcnt wait is placed by hand.
typedef double d2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t lds_addr(const void* p) { return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p; }
template <int OFF>
__device__ __forceinline__ d2_t lds_read128(uint32_t addr)
{
    d2_t v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
__device__ __forceinline__ void lds_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// a value an earlier inline-assembly load produced may be used only behind the wait that covers it: this empty statement
// (volatile statements keep their order) makes every later use depend on a point behind that wait
template <class V>
__device__ __forceinline__ void pin_after_wait(V& v) { asm volatile("" : "+v"(v)); }

// the 16 residuals of a lane from its wave's staging tile (lane-linear 1 KiB pieces, the permuted eps layout)
__device__ __forceinline__ void lds_eps16(uint32_t tile_addr, int lane, double (&e)[IPT])
{
    const uint32_t a = tile_addr + ((uint32_t)lane << 4);
    d2_t v0 = lds_read128<0>(a), v1 = lds_read128<1024>(a), v2 = lds_read128<2048>(a), v3 = lds_read128<3072>(a);
    d2_t v4 = lds_read128<4096>(a), v5 = lds_read128<5120>(a), v6 = lds_read128<6144>(a), v7 = lds_read128<7168>(a);
    lds_wait(); // eps is in registers: the staging tile may be overwritten
    pin_after_wait(v0); pin_after_wait(v1); pin_after_wait(v2); pin_after_wait(v3);
    pin_after_wait(v4); pin_after_wait(v5); pin_after_wait(v6); pin_after_wait(v7);
    e[0] = v0.x; e[1] = v0.y; e[2] = v1.x; e[3] = v1.y; e[4] = v2.x; e[5] = v2.y; e[6] = v3.x; e[7] = v3.y;
    e[8] = v4.x; e[9] = v4.y; e[10] = v5.x; e[11] = v5.y; e[12] = v6.x; e[13] = v6.y; e[14] = v7.x; e[15] = v7.y;
}

// apply_update16_lds with such reads: tab_addr = LDS address of the 16-entry pair table of one pending update, a multiple of 256
__device__ __forceinline__ void apply_update16_asm(uint32_t w, uint32_t tab_addr, double (&e)[IPT])
{
    d2_t v[IPT / 2];
#pragma unroll
    for (int s = 0; s < IPT; s += 2) {
        // entry ((w >> 2s) & 15) of a table that starts on a 256-byte boundary: the four index bits land in bits 4..7 by one
        // shift and reach the address through an AND-OR (v_and_or_b32) instead of shift, mask, shift, add
        const uint32_t idx16 = (2 * s >= 4) ? ((w >> (2 * s - 4)) & 0xF0u) : ((w << (4 - 2 * s)) & 0xF0u);
        v[s >> 1] = lds_read128<0>(idx16 | tab_addr);
    }
    lds_wait();
#pragma unroll
    for (int s = 0; s < IPT; s += 2) {
        pin_after_wait(v[s >> 1]);
        e[s] = e[s] + v[s >> 1].x;
        e[s + 1] = e[s + 1] + v[s >> 1].y;
    }
}

// a8 inside the sweep: the same addends from a 16-entry table in LDS indexed by a PAIR of codes (4 bits of the column
// dword): field extract, one 16-byte LDS read and two adds per two individuals instead of a chain of selects.
// tab[c1 << 2 | c0] = (0.0 + v[c0], 0.0 + v[c1]); all lanes read the same 256 bytes (one entry per 4 banks: broadcast).
__device__ __forceinline__ void apply_update16_lds(uint32_t w, const double2* tab, double (&e)[IPT])
{
#pragma unroll
    for (int s = 0; s < IPT; s += 2) {
        const double2 v = tab[(w >> (2 * s)) & 15u];
        e[s] = e[s] + v.x;
        e[s + 1] = e[s + 1] + v.y;
    }
}

} // namespace hg
