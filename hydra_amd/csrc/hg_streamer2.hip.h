// Streaming workgroup of the resident sweep engine, second form ("limb dots"): the refill's dot products as integer matrix products.
//
// hg_resident.hip.h's first form gives every wave whole columns (all 1024 T individuals of the workgroup, sixteen per lane and dword) and
// takes the dot x_j'eps (a4, src/BayesRRm.cpp:1766-1809) individual by individual: field extract, int -> f64, fused multiply-add -- three
// vector instructions per individual and column, ~125 per column at T = 2, on a machine that issues one vector instruction per SIMD
// every ~4.2 clocks whatever it is: config 4's refill is bound by exactly that, not by HBM.
//
// Here a wave owns a SLICE of the individuals (128 T of them: wave w the dwords [8 T w, 8 T (w + 1)) of the workgroup's 64 T) and takes
// ALL columns against it, sixteen at a time, on v_mfma_i32_16x16x64_i8:
//   A (16 "rows" x 64 individuals, bytes): eps as a fixed-point integer, round(eps 2^44), written in seven SIGNED base-256 digits; lane (j, g)
//     holds digit j of sixteen individuals of k-group g (lanes j >= 7: zero).  The digits are made once per update of eps, by the lanes that
//     own the individuals (2 T each), and handed to the lanes that need them through a small LDS image per wave;
//   B (64 individuals x 16 columns): lane (c, g) expands ONE dword of column c -- the sixteen 2-bit codes of the same individuals -- to
//     sixteen bytes (seven instructions; codes 0 1 2, a missing call 3);
//   D (i32): lane (c, g), register r: sum over the 64 individuals of digit 4 g + r x code, column c -- exact; sum_j D_j 256^j is the
//     column's dot against the wave's slice as an integer of units 2^-44, exact again: the lane puts its four digits together and the
//     parts meet in one 8-byte LDS accumulator per position (integer adds: no order to keep).
// One step (16 columns x 64 individuals) costs seven vector instructions and one MFMA instead of 48; eps is 2 T doubles per lane instead
// of 16 T, so the update a8 (src/BayesRRm.cpp:1976-2010) costs an eighth; nothing is summed in floating point, so the dot does not
// depend on which wave took what.  What the walker sees is unchanged: the workgroup's part of s1 as a fixed-point integer of 1 / fx_scale
// units added to racc (rounded ONCE, from the exact 2^-44 sum), batch counters, Gram terms, messages (hg_resident.hip.h).
// Build MISS: a second product with A' = (code == 3) gives R = sum of eps over the missing calls by the same digits -- s1' = s1 + 3 R and R
// travel as before; no gather, no copy of eps in LDS.
// Numerics: |eps| < 64 is required (< 32 at four tiles per workgroup; standardised phenotypes: |eps| of a few units; a sweep that meets a
// larger one is refused, error 5, never wrapped) -- eps 2^44 then fits 51 bits ("x + 1.5 2^52" rounds to nearest), a column's exact sum 63.  The quantisation of eps,
// 2^-45 per individual, adds ~1e-11 to a dot at N = 500 K -- below the 1 / fx_scale (~5e-10) the parts are rounded to anyway.
// Operand maps checked with random data against integer sums: tools/ubench/mfma_limb_dot.hip.
#pragma once

namespace hg {

constexpr int RL_NG = 8;           // register sets: groups of sixteen columns a wave has in registers or on their way (128 positions; T = 4: four sets, 64 positions)
constexpr int RL_VGPR_LIMIT = 200; // v200 .. v215: the sets' (mave | mstd), v216 .. v223: the ids of the sets' next groups, v224 .. v255: the sets
constexpr int RL_VGPR_LIMIT4 = 208; // T = 4 (32 bytes of a column per lane, FOUR sets: 64 positions): v208 .. v215, v216 .. v219, v224 .. v255
// (the attribute amdgpu_num_vgpr counts the unified VGPR + AGPR file on this target: HALF the limit is what caps the compiler's own registers)
constexpr int RL_TMAX = 4;         // tiles per workgroup in this form: eps is 2 T doubles per lane, the window's codes B x 256 T bytes (B = 128 at T = 4)
constexpr int RL_EX = 44;          // eps digits: units of 2^-44
constexpr uint32_t RL_BLK = 144;   // bytes of a block of sixteen individuals in the digit image: 7 x 16 in use, padded so that the lanes' writes spread over the banks
typedef int rl_v4i __attribute__((ext_vector_type(4)));

// LDS: [0, 512) table, message, counters, stage clocks (as in the first form); (mave, mstd) of the window slots; the round's integer
// accumulators of s1 (and R); a block of zeros + the waves' digit images; the ring of window codes
__host__ __device__ constexpr size_t rl_meta_off() { return 512; }
__host__ __device__ constexpr size_t rl_acc_off(uint32_t B) { return 512 + (size_t)B * 16; }
__host__ __device__ constexpr size_t rl_acc2_off(uint32_t B) { return 512 + (size_t)B * 24; }
__host__ __device__ constexpr size_t rl_zero_off(uint32_t B) { return 512 + (size_t)B * 32; }
__host__ __device__ constexpr size_t rl_digit_off(uint32_t B) { return rl_zero_off(B) + 64; }
__host__ __device__ constexpr int rl_image_blocks(int T) { return 8 * T < 16 ? 8 * T : 16; } // blocks of sixteen individuals in a wave's digit image (T = 4: made in two halves)
__host__ __device__ constexpr size_t rl_ring_off(uint32_t B, int T) { return rl_digit_off(B) + (size_t)RS_WAVES * rl_image_blocks(T) * RL_BLK; }
__host__ __device__ constexpr size_t rl_streamer_lds(uint32_t B, int T) { return rl_ring_off(B, T) + (size_t)B * 256 * T; }
static_assert(rl_streamer_lds(RS_BMAX, RS_TMAX) <= 160 * 1024 && rl_streamer_lds(RS_BMAX / 2, RL_TMAX) <= 160 * 1024,
              "the largest window at the most tiles per workgroup fits the 160 KB of LDS of a compute unit");

// ---- the sets: named registers, loads the compiler does not see as loads (hg_resident.hip.h explains why) ----
// T = 1, 2: set r = v[224 + 4 r .. 227 + 4 r] (T = 1: the first two), the column's (mave | mstd) v[200 + 2 r, 201 + 2 r], the next id v(216 + r)
#define RL_SET_LIST(X) X(0, 224, 225, 226, 227, 200, 201, 216) X(1, 228, 229, 230, 231, 202, 203, 217) X(2, 232, 233, 234, 235, 204, 205, 218) X(3, 236, 237, 238, 239, 206, 207, 219) \
    X(4, 240, 241, 242, 243, 208, 209, 220) X(5, 244, 245, 246, 247, 210, 211, 221) X(6, 248, 249, 250, 251, 212, 213, 222) X(7, 252, 253, 254, 255, 214, 215, 223)
// T = 4: four sets, set r = v[224 + 8 r .. 231 + 8 r], (mave | mstd) v[208 + 2 r, 209 + 2 r], the next id v(216 + r)
#define RL4_SET_LIST(X) X(0, 224, 225, 226, 227, 228, 229, 230, 231, 208, 209, 216) X(1, 232, 233, 234, 235, 236, 237, 238, 239, 210, 211, 217) \
    X(2, 240, 241, 242, 243, 244, 245, 246, 247, 212, 213, 218) X(3, 248, 249, 250, 251, 252, 253, 254, 255, 214, 215, 219)
// set R takes 8 T bytes of a column per lane (T = 1: the first two registers of the set)
template <int T, int R>
__device__ __forceinline__ void rl_set_load(const uint8_t* addr)
{
    if constexpr (T == 4) {
#define RL_X(r, a, b, c, d, e, f, g, h, m0, m1, id)                                                                                     \
    if constexpr (R == r)                                                                                                              \
        asm volatile("global_load_dwordx4 v[" #a ":" #d "], %0, off\n\tglobal_load_dwordx4 v[" #e ":" #h "], %0, off offset:16" ::"v"(addr) \
                     : "memory", "v" #a, "v" #b, "v" #c, "v" #d, "v" #e, "v" #f, "v" #g, "v" #h);
        RL4_SET_LIST(RL_X)
#undef RL_X
    } else {
#define RL_X(r, a, b, c, d, m0, m1, id)                                                                                             \
    if constexpr (R == r) {                                                                                                        \
        if constexpr (T == 1) asm volatile("global_load_dwordx2 v[" #a ":" #b "], %0, off" ::"v"(addr) : "memory", "v" #a, "v" #b); \
        else asm volatile("global_load_dwordx4 v[" #a ":" #d "], %0, off" ::"v"(addr) : "memory", "v" #a, "v" #b, "v" #c, "v" #d);  \
    }
        RL_SET_LIST(RL_X)
#undef RL_X
    }
}
// the set's dwords, masked (keep: the lane's valid individuals)
template <int T, int R>
__device__ __forceinline__ void rl_set_read(uint32_t (&w)[2 * T], const uint32_t (&keep)[2 * T])
{
    if constexpr (T == 4) {
#define RL_X(r, a, b, c, d, e, f, g, h, m0, m1, id)                                                                                                  \
    if constexpr (R == r)                                                                                                                           \
        asm volatile("v_and_b32 %0, v" #a ", %8\n\tv_and_b32 %1, v" #b ", %9\n\tv_and_b32 %2, v" #c ", %10\n\tv_and_b32 %3, v" #d ", %11\n\t"          \
                     "v_and_b32 %4, v" #e ", %12\n\tv_and_b32 %5, v" #f ", %13\n\tv_and_b32 %6, v" #g ", %14\n\tv_and_b32 %7, v" #h ", %15"              \
                     : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2 * T - 6]), "=&v"(w[2 * T - 5]), "=&v"(w[2 * T - 4]), "=&v"(w[2 * T - 3]), "=&v"(w[2 * T - 2]), "=&v"(w[2 * T - 1]) \
                     : "v"(keep[0]), "v"(keep[1]), "v"(keep[2 * T - 6]), "v"(keep[2 * T - 5]), "v"(keep[2 * T - 4]), "v"(keep[2 * T - 3]), "v"(keep[2 * T - 2]), "v"(keep[2 * T - 1]));
        RL4_SET_LIST(RL_X)
#undef RL_X
    } else {
#define RL_X(r, a, b, c, d, m0, m1, id)                                                                                                          \
    if constexpr (R == r) {                                                                                                                     \
        if constexpr (T == 1) asm volatile("v_and_b32 %0, v" #a ", %2\n\tv_and_b32 %1, v" #b ", %3" : "=&v"(w[0]), "=&v"(w[1]) : "v"(keep[0]), "v"(keep[1])); \
        else asm volatile("v_and_b32 %0, v" #a ", %4\n\tv_and_b32 %1, v" #b ", %5\n\tv_and_b32 %2, v" #c ", %6\n\tv_and_b32 %3, v" #d ", %7"       \
                          : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2 * T - 2]), "=&v"(w[2 * T - 1])                                                   \
                          : "v"(keep[0]), "v"(keep[1]), "v"(keep[2 * T - 2]), "v"(keep[2 * T - 1]));                                              \
    }
        RL_SET_LIST(RL_X)
#undef RL_X
    }
}
// the id of the column this lane takes when set R is loaded next, and (wave 0) the 8 bytes that travel with the set's column of this
// lane: k-group 0 its mave, k-group 1 its mstd
template <int T, int R>
__device__ __forceinline__ void rl_id_load(const int32_t* id)
{
    if constexpr (T == 4) {
#define RL_X(r, a, b, c, d, e, f, g, h, m0, m1, idr) \
    if constexpr (R == r) asm volatile("global_load_dword v" #idr ", %0, off" ::"v"(id) : "memory", "v" #idr);
        RL4_SET_LIST(RL_X)
#undef RL_X
    } else {
#define RL_X(r, a, b, c, d, m0, m1, idr) \
    if constexpr (R == r) asm volatile("global_load_dword v" #idr ", %0, off" ::"v"(id) : "memory", "v" #idr);
        RL_SET_LIST(RL_X)
#undef RL_X
    }
}
template <int T, int R>
__device__ __forceinline__ int32_t rl_id_read()
{
    int32_t v = 0;
    if constexpr (T == 4) {
#define RL_X(r, a, b, c, d, e, f, g, h, m0, m1, idr) \
    if constexpr (R == r) asm volatile("v_mov_b32 %0, v" #idr : "=v"(v));
        RL4_SET_LIST(RL_X)
#undef RL_X
    } else {
#define RL_X(r, a, b, c, d, m0, m1, idr) \
    if constexpr (R == r) asm volatile("v_mov_b32 %0, v" #idr : "=v"(v));
        RL_SET_LIST(RL_X)
#undef RL_X
    }
    return v;
}
template <int T, int R>
__device__ __forceinline__ void rl_meta_load(const double* a)
{
    if constexpr (T == 4) {
#define RL_X(r, sa, sb, sc, sd, se, sf, sg, sh, m0, m1, idr) \
    if constexpr (R == r) asm volatile("global_load_dwordx2 v[" #m0 ":" #m1 "], %0, off" ::"v"(a) : "memory", "v" #m0, "v" #m1);
        RL4_SET_LIST(RL_X)
#undef RL_X
    } else {
#define RL_X(r, sa, sb, sc, sd, m0, m1, idr) \
    if constexpr (R == r) asm volatile("global_load_dwordx2 v[" #m0 ":" #m1 "], %0, off" ::"v"(a) : "memory", "v" #m0, "v" #m1);
        RL_SET_LIST(RL_X)
#undef RL_X
    }
}
template <int T, int R>
__device__ __forceinline__ void rl_meta_read(int& lo, int& hi)
{
    if constexpr (T == 4) {
#define RL_X(r, sa, sb, sc, sd, se, sf, sg, sh, m0, m1, idr) \
    if constexpr (R == r) asm volatile("v_mov_b32 %0, v" #m0 "\n\tv_mov_b32 %1, v" #m1 : "=&v"(lo), "=&v"(hi));
        RL4_SET_LIST(RL_X)
#undef RL_X
    } else {
#define RL_X(r, sa, sb, sc, sd, m0, m1, idr) \
    if constexpr (R == r) asm volatile("v_mov_b32 %0, v" #m0 "\n\tv_mov_b32 %1, v" #m1 : "=&v"(lo), "=&v"(hi));
        RL_SET_LIST(RL_X)
#undef RL_X
    }
}

// sixteen 2-bit codes -> four dwords of bytes: dword r, byte i = the code of individual 4 i + r of the sixteen
__device__ __forceinline__ rl_v4i rl_expand16(uint32_t x)
{
    rl_v4i z;
    z.x = (int)(x & 0x03030303u);
    z.y = (int)((x >> 2) & 0x03030303u);
    z.z = (int)((x >> 4) & 0x03030303u);
    z.w = (int)((x >> 6) & 0x03030303u);
    return z;
}
// eps as seven signed base-256 digits of round(eps 2^44): y = x + 0x808080808080 (the six low bytes biased), digit j < 6 = byte j of y
// ^ 0x80 read as int8, digit 6 = byte 6 of y
__device__ __forceinline__ void rl_digits(double e, uint32_t& lo, uint32_t& hi)
{
    const double MAGIC = 6755399441055744.0; // 1.5 2^52: x + MAGIC rounds x to the nearest integer, exact for |x| < 2^51
    const unsigned long long x = (unsigned long long)(__double_as_longlong(fma(e, (double)(1ull << RL_EX), MAGIC)) - __double_as_longlong(MAGIC));
    const unsigned long long y = (x + 0x0000808080808080ull) ^ 0x0000808080808080ull;
    lo = (uint32_t)y;
    hi = (uint32_t)(y >> 32);
}

template <int T, int DBG, int MISS>
__device__ __forceinline__ void res_streamer_limb(const ResParams& p, unsigned char* smem)
{
    constexpr int ND = 2 * T;      // dwords of a column per lane and group of sixteen columns = k-steps per group = individuals per lane
    constexpr int NG = T == 4 ? 4 : RL_NG; // register sets
    constexpr int LPB = 8 / T;     // lanes that own a block of sixteen individuals
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t wg = blockIdx.x;
    const uint32_t B = p.B, bmask = B - 1u, M = p.M;
    double* const tab1 = reinterpret_cast<double*>(smem);                               // the update's addends by window code (00, 01, 10 = missing, 11)
    unsigned long long* const lmsg = reinterpret_cast<unsigned long long*>(smem + 256); // the message, as the polling lane read it
    unsigned long long* const tacc = reinterpret_cast<unsigned long long*>(smem + 384); // [8] stage clocks of the debug build
    double2* const meta = reinterpret_cast<double2*>(smem + rl_meta_off());             // (mave, mstd) of the window slots
    unsigned long long* const acc1 = reinterpret_cast<unsigned long long*>(smem + rl_acc_off(B));  // [position - Sx] this round's refill: the column's dot, integer units of 2^-44
    unsigned long long* const acc2 = reinterpret_cast<unsigned long long*>(smem + rl_acc2_off(B)); // build MISS: the same for R
    unsigned char* const zero16 = smem + rl_zero_off(B);
    unsigned char* const dimg = smem + rl_digit_off(B) + (size_t)wave * (rl_image_blocks(T) * RL_BLK); // this wave's digit image: [block of sixteen individuals][digit][16 bytes in A's order] (T = 4: half of the blocks at a time)
    uint32_t* const ring = reinterpret_cast<uint32_t*>(smem + rl_ring_off(B, T));       // [B][64 * T] codes of the window columns (x form)
    if (tid < 16) reinterpret_cast<uint32_t*>(zero16)[tid] = 0u;
    for (uint32_t i = (uint32_t)tid; i < 2u * B; i += RS_BLOCK) acc1[i] = 0ull; // (acc1 and acc2 are adjacent)
    const bool timing = DBG && wg == 0 && tid == 0;
    unsigned long long tmark = timing ? wall_clock64() : 0ull;
    if (timing)
        for (int i = 0; i < 8; ++i) tacc[i] = 0ull;
    auto lap = [&](int i) {
        if (timing) {
            const unsigned long long now = wall_clock64();
            tacc[i] += now - tmark;
            tmark = now;
        }
    };

    // ---- who holds what ----
    // A / codes: lane (c, g) = (lane & 15, lane >> 4) takes column c of a group and the dwords D = 8 T wave + 2 T g + d, d < 2 T, of the workgroup's 64 T
    const uint32_t cl = (uint32_t)lane & 15u, gl = (uint32_t)lane >> 4;
    const uint32_t ndw = p.n_pad >> 4;
    const uint32_t Dc = (uint32_t)wave * 8u * T + gl * (uint32_t)ND;    // first dword of this lane's codes, in the workgroup
    const bool vcol = wg * 64u * T + Dc < ndw;                         // (the shard's dwords are a multiple of 256: all of the lane's or none)
    const uint32_t coff = vcol ? (wg * 64u * T + Dc) * 4u : 0u;        // byte offset into a column
    uint32_t keep[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d) {
        const uint32_t i0 = (wg * 64u * T + Dc + (uint32_t)d) * 16u;
        const uint32_t nv = (!vcol || i0 >= p.n_local) ? 0u : (p.n_local - i0 >= 16u ? 16u : p.n_local - i0);
        keep[d] = nv >= 16u ? 0xffffffffu : ((1u << (2u * nv)) - 1u);
    }
    // eps: lane l owns 2 T individuals of block bw = l / LPB of the wave's 8 T -- whole dwords of A's order (dword r = the individuals r, 4 + r,
    // 8 + r, 12 + r of the block) or, at T = 1, half of one: T = 1: dword r4 = m & 3, bytes 2 hq, 2 hq + 1 (m = l % 8, hq = m >> 2); T = 2: dword m
    // (m = l % 4); T = 4: dwords 2 m, 2 m + 1 (m = l % 2)
    const uint32_t bw = (uint32_t)lane / (uint32_t)LPB, ml = (uint32_t)lane % (uint32_t)LPB;
    const uint32_t r4 = T == 4 ? 2u * ml : (ml & 3u), hq = T == 1 ? ml >> 2 : 0u;
    const uint32_t De = wg * 64u * T + (uint32_t)wave * 8u * T + bw;   // the block's dword in the shard
    double e[ND];
    uint32_t eb[ND]; // the individuals' index within their block
    bool ev[ND];     // ... is an individual of the shard (padding stays zero)
#pragma unroll
    for (int s = 0; s < ND; ++s) {
        eb[s] = T == 4 ? (r4 + ((uint32_t)s >> 2)) + 4u * ((uint32_t)s & 3u) : r4 + 4u * (hq * (uint32_t)ND + (uint32_t)s);
        const uint32_t i = De * 16u + eb[s];
        ev[s] = De < ndw && i < p.n_local;
        e[s] = ev[s] ? p.eps[eps_pos(i)] : 0.0;
    }
    // B: lane (j, g) = (lane & 15, lane >> 4), step d: digit j of block 2 T g + d of the wave's image (lanes j >= 7: the zeros)
    rl_v4i bop[ND];
    auto digits_to_operands = [&]() {
        uint32_t lo[ND], hi[ND];
        bool big = false;
#pragma unroll
        for (int s = 0; s < ND; ++s) {
            rl_digits(e[s], lo[s], hi[s]);
            big = big || !(fabs(e[s]) < (T == 4 ? 32.0 : 64.0)); // (a column's exact sum over the workgroup's 1024 T individuals must fit 63 bits: 3 x 1024 T x |eps| 2^44)
        }
        if (big) atomicMax(&p.state->error, 5u); // out of the fixed-point range (or not finite): the sweep is refused, not wrapped
        // 4 x 4 byte transposes: digit j of four individuals in one dword (byte s = individual s of the four)
        auto tr4 = [&](unsigned char* blk, uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3, int jbase, int nj) {
            const uint32_t t0 = __builtin_amdgcn_perm(x1, x0, 0x05010400u), t1 = __builtin_amdgcn_perm(x1, x0, 0x07030602u);
            const uint32_t u0 = __builtin_amdgcn_perm(x3, x2, 0x05010400u), u1 = __builtin_amdgcn_perm(x3, x2, 0x07030602u);
            *reinterpret_cast<uint32_t*>(blk + (jbase + 0) * 16) = __builtin_amdgcn_perm(u0, t0, 0x05040100u);
            *reinterpret_cast<uint32_t*>(blk + (jbase + 1) * 16) = __builtin_amdgcn_perm(u0, t0, 0x07060302u);
            *reinterpret_cast<uint32_t*>(blk + (jbase + 2) * 16) = __builtin_amdgcn_perm(u1, t1, 0x05040100u);
            if (nj > 3) *reinterpret_cast<uint32_t*>(blk + (jbase + 3) * 16) = __builtin_amdgcn_perm(u1, t1, 0x07060302u);
        };
        auto operands = [&](uint32_t first_block) { // the lane's operand registers of the blocks that are in the image now
#pragma unroll
            for (int d = 0; d < ND; ++d) {
                const unsigned char* src = cl < 7u ? dimg + (gl * (uint32_t)ND + (uint32_t)d - first_block) * RL_BLK + cl * 16u : zero16;
                bop[d] = *reinterpret_cast<const rl_v4i*>(src);
            }
        };
        if constexpr (T == 4) {
            // the image holds sixteen blocks: the lanes 0 .. 31 own the blocks 0 .. 15 (read by the k-groups 0 and 1), the lanes 32 .. 63 the rest
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (((uint32_t)lane >> 5) == (uint32_t)h) {
                    unsigned char* const blk = dimg + (bw & 15u) * RL_BLK + r4 * 4u;
                    tr4(blk, lo[0], lo[1], lo[2], lo[3], 0, 4);
                    tr4(blk, hi[0], hi[1], hi[2], hi[3], 4, 3);
                    tr4(blk + 4, lo[ND - 4], lo[ND - 3], lo[ND - 2], lo[ND - 1], 0, 4);
                    tr4(blk + 4, hi[ND - 4], hi[ND - 3], hi[ND - 2], hi[ND - 1], 4, 3);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                if ((gl >> 1) == (uint32_t)h) operands(16u * (uint32_t)h);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // (read before the other half overwrites the image)
                __builtin_amdgcn_wave_barrier();
            }
        } else {
            unsigned char* const blk = dimg + bw * RL_BLK + r4 * 4u + hq * (uint32_t)ND;
            if constexpr (T == 2) {
                tr4(blk, lo[0], lo[1], lo[ND - 2], lo[ND - 1], 0, 4);
                tr4(blk, hi[0], hi[1], hi[ND - 2], hi[ND - 1], 4, 3);
            } else {
                // two individuals: digit j = bytes (x0.j, x1.j)
                auto tr2 = [&](const uint32_t (&x)[ND], int jbase, int nj) {
                    const uint32_t t0 = __builtin_amdgcn_perm(x[1], x[0], 0x05010400u), t1 = __builtin_amdgcn_perm(x[1], x[0], 0x07030602u);
                    *reinterpret_cast<uint16_t*>(blk + (jbase + 0) * 16) = (uint16_t)t0;
                    *reinterpret_cast<uint16_t*>(blk + (jbase + 1) * 16) = (uint16_t)(t0 >> 16);
                    *reinterpret_cast<uint16_t*>(blk + (jbase + 2) * 16) = (uint16_t)t1;
                    if (nj > 3) *reinterpret_cast<uint16_t*>(blk + (jbase + 3) * 16) = (uint16_t)(t1 >> 16);
                };
                tr2(lo, 0, 4);
                tr2(hi, 4, 3);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // (the wave's own writes: in order with its reads -- made explicit for the compiler)
            __builtin_amdgcn_wave_barrier();
            operands(0u);
        }
    };
    __syncthreads(); // (the zeros, the accumulators)
    digits_to_operands();

    uint32_t C = 0, Sx = 0, seq = 0, nev = 0;
    uint32_t kind = RS_ADVANCE, ncons = 0;
    bool gram_sent = false; // the Gram terms of the event the next message brings have been sent (on its announcement)
    bool count_due = false; // (last wave) this round's adds are out, the batch counter's is not
    bool last = M == 0;
    double dbeta = 0.0;
    // Set r holds the group H_r of sixteen positions [16 H_r, 16 H_r + 16) -- the first group = r (mod NG) that is not admitted to the
    // window in full; it is reloaded (group H_r + NG) right after the round that admits its last column.
    auto cols_landed = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
    // (the lane's array of the two, as a VALUE made once: selected inside the loop the compiler reads the pointer from the kernel's arguments
    // at a per-lane address -- a load in front of the set's loads, and its wait is for everything in flight)
    const double* mbase = gl == 0u ? p.s_mave : p.s_mstd;
    {
        uint32_t mb_lo = (uint32_t)(uintptr_t)mbase, mb_hi = (uint32_t)((uintptr_t)mbase >> 32);
        asm volatile("" : "+v"(mb_lo), "+v"(mb_hi));
        mbase = reinterpret_cast<const double*>(((uintptr_t)mb_hi << 32) | mb_lo);
    }
    auto load_set = [&](auto rtag, uint32_t G, int32_t id) __attribute__((always_inline)) {
        constexpr int r = decltype(rtag)::value;
        const uint32_t pn = 16u * G + cl;
        rl_set_load<T, r>(p.bed + ((size_t)((uint32_t)(pn < M ? id : 0) * (uint32_t)(p.stride >> 10)) << 10) + coff);
        const uint32_t pi = 16u * (G + (uint32_t)NG) + cl; // the id of the set's next column
        rl_id_load<T, r>(p.order + (pi < M ? pi : 0u));
        if (wave == 0 && gl < 2u) rl_meta_load<T, r>(mbase + (pn < M ? pn : 0u)); // what travels with the column: k-group 0 its mave, k-group 1 its mstd
    };
    {
        int32_t id0[NG];
#pragma unroll
        for (int r = 0; r < NG; ++r) {
            const uint32_t pn = 16u * (uint32_t)r + cl;
            id0[r] = p.order[pn < M ? pn : 0u];
        }
        [&]<int... R>(std::integer_sequence<int, R...>) { (load_set(std::integral_constant<int, R>{}, (uint32_t)R, id0[R]), ...); }(std::make_integer_sequence<int, NG>{});
    }

    // wait for the walker's message number seq (one lane polls; everybody else sleeps at the barrier); behind an announcement: for the
    // message that overwrites it
    auto take_message = [&](bool announced) {
        if (wg == 0 && tid == 0) p.progress[1] = ((unsigned long long)seq << 8) | 3u;
        if (tid == RS_BLOCK - WAVE) {
            const ResMsg* m = p.msg + (seq % RS_MSG);
            const unsigned long long t0 = wall_clock64();
            uint32_t npoll = 0;
            u4_t v;
            for (;;) {
                v = rs_load16(m); // (waits for everything this wave has in flight: the round's adds to the shard's accumulators have been performed)
                if (count_due) {
                    __hip_atomic_fetch_add(p.rcnt + (size_t)(wg % p.rsh) * RS_CROW, 1u, HG_RLX_AGENT);
                    count_due = false;
                }
                if (v.y == seq && ((v.x >> 12) & 0xffffu) == rs_msg_check(seq, v.z, v.w) && !(announced && ((v.x >> 28) & 7u) == (uint32_t)RS_ANNOUNCE)) break;
                if (wall_clock64() - t0 > p.timeout || ((++npoll & 255u) == 0u && __hip_atomic_load(p.progress + 2, HG_RLX_AGENT) != 0ull)) { // (or the host gave the sweep up)
                    v.x = (uint32_t)RS_ABORT << 28;
                    v.y = seq;
                    atomicMax(&p.state->error, 3u);
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
            lmsg[0] = ((unsigned long long)v.y << 32) | v.x;
            lmsg[1] = ((unsigned long long)v.w << 32) | v.z;
        }
        __syncthreads();
        const unsigned long long tag = lmsg[0];
        dbeta = __longlong_as_double((long long)lmsg[1]);
        const uint32_t kf = (uint32_t)(tag >> 28) & 0xfu;
        kind = kf & 7u;
        last = (kf & RS_LAST) != 0u;
        ncons = (uint32_t)tag & 0xfffu;
        lap(0);
        if (timing) p.trace[4 * RS_TRACE + seq % RS_TRACE] = wall_clock64();
    };

    for (;;) {
        const bool ann = kind == RS_ANNOUNCE;
        const bool upd = kind == RS_EVENT || kind == RS_PIVOT || kind == RS_ANNOUNCED;
        const bool with_gram = kind == RS_EVENT || ann || (kind == RS_ANNOUNCED && !gram_sent);
        if (DBG && wg == 0 && tid == 0) p.progress[1] = ((unsigned long long)seq << 8) | 1u;
        const uint32_t q = C + ncons - 1u;
        const uint32_t Cn = ann ? C : C + ncons;
        const uint32_t Sn = rs_window_end(Cn, B, M, p.wend_mask);
        const uint32_t nnew = Sn - Sx;
        cols_landed(); // issued at the end of the last round: nothing to wait for

        if (upd || ann) {
            const uint32_t slotq = q & bmask;
            uint32_t xq[T];
#pragma unroll
            for (int t = 0; t < T; ++t) xq[t] = ring[slotq * 64u * T + (uint32_t)lane * T + t];
            const uint32_t xqe = ring[slotq * 64u * T + (uint32_t)wave * 8u * T + bw]; // the codes of the block this lane owns individuals of
            const double2 mq = meta[slotq];
            if (upd && tid < 4) { // the update's addends by window code (x form 00 / 01 / 11 = genotype 0 / 1 / 2; 10 = missing call, addend 0)
                const double av = mq.x, sd = mq.y, db = dbeta;
                const double v0 = -(av * sd * db), v1 = db * (1.0 - av) * sd, v2 = db * (2.0 - av) * sd;
                tab1[tid] = 0.0 + (tid == 0 ? v0 : (tid == 1 ? v1 : (tid == 3 ? v2 : 0.0)));
            }
            // ---- FIRST what the walker waits for: the integer Gram terms of the window columns behind q (hg_resident.hip.h: rs_gram_terms) ----
            if (with_gram) rs_gram_terms<T, MISS>(p, ring, meta, wg, wave, lane, q, Sx, bmask, nev, xq, mq);

            if (with_gram) ++nev;
            lap(2);
            if (timing) p.trace[6 * RS_TRACE + seq % RS_TRACE] = wall_clock64();
            // ---- a8 (src/BayesRRm.cpp:1976-2010,2022,2471): eps += {v0, v1, 0, v2}[code] on the individuals this lane owns, then their digits ----
            rs_lds_barrier(); // (the table; an announcement: everybody has read the message before the polling lane writes the next one)
            if (upd) {
#pragma unroll
                for (int s = 0; s < ND; ++s) {
                    const double add = tab1[(xqe >> (2u * eb[s])) & 3u];
                    e[s] += ev[s] ? add : 0.0;
                }
                digits_to_operands();
            }
            lap(1);
            if (timing) p.trace[5 * RS_TRACE + seq % RS_TRACE] = wall_clock64();
        }
        gram_sent = ann;
        if (ann) { // the message proper takes the announcement's place: same number, same slot
            take_message(true);
            if (kind == RS_ABORT) break;
            continue;
        }

        // ---- a4 (src/BayesRRm.cpp:1766-1809) of the positions [Sx, Sn) that refill the window, against eps as it is now: the groups
        // G0 .. G1 of sixteen, every wave its slice of every column ----
        if (nnew) {
            const uint32_t G0 = Sx >> 4, G1 = (Sn - 1u) >> 4;
            for (uint32_t gb = G0; gb <= G1; gb += (uint32_t)NG) { // (more than NG groups: the first fill, a long advance)
                if (gb != G0) cols_landed();
                auto one = [&](auto rtag) __attribute__((always_inline)) {
                    constexpr int r = decltype(rtag)::value;
                    const uint32_t G = gb + (((uint32_t)r - gb) & (uint32_t)(NG - 1));
                    if (G <= G1) { // wave-uniform
                        uint32_t w[ND];
                        rl_set_read<T, r>(w, keep);
                        const uint32_t pc = 16u * G + cl;
                        const bool mine = pc >= Sx && pc < Sn;
                        bool anym = false;
                        int mlo = 0, mhi = 0;
                        if (wave == 0) rl_meta_read<T, r>(mlo, mhi); // (wave 0 fills the slots' (mave, mstd))
                        if constexpr (MISS) {
                            // a column with missing calls is known by its codes: 11 anywhere in the group's dwords of this wave (no flag needed)
                            uint32_t any3 = 0u;
#pragma unroll
                            for (int d = 0; d < ND; ++d) any3 |= w[d] & (w[d] >> 1) & 0x55555555u;
                            anym = __ballot(any3 != 0u) != 0ull;
                        }
                        rl_v4i a1 = {0, 0, 0, 0}, a2 = {0, 0, 0, 0};
                        uint32_t xf[ND];
#pragma unroll
                        for (int d = 0; d < ND; ++d) {
                            const rl_v4i z = rl_expand16(w[d]);
                            a1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(bop[d], z, a1, 0, 0, 0);
                            if constexpr (MISS) {
                                if (anym) { // wave-uniform
                                    rl_v4i zm;
                                    zm.x = z.x & (z.x >> 1);
                                    zm.y = z.y & (z.y >> 1);
                                    zm.z = z.z & (z.z >> 1);
                                    zm.w = z.w & (z.w >> 1);
                                    a2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(bop[d], zm, a2, 0, 0, 0);
                                }
                                const uint32_t hi1 = (w[d] >> 1) & 0x55555555u, mm = w[d] & hi1;
                                xf[d] = (w[d] | hi1) ^ mm; // the x form 00 / 01 / 11, a missing call the free code 10
                            } else
                                xf[d] = gram_xform(w[d]);
                        }
                        // the window keeps the x form: what the update's table and the Gram terms of every later event need, made once
                        if (mine) {
                            uint32_t* rp = ring + (pc & bmask) * 64u * T + Dc;
                            if constexpr (T == 4) {
                                *reinterpret_cast<u4_t*>(rp) = rs_u4(xf[0], xf[1], xf[ND - 6], xf[ND - 5]);
                                *reinterpret_cast<u4_t*>(rp + 4) = rs_u4(xf[ND - 4], xf[ND - 3], xf[ND - 2], xf[ND - 1]);
                            } else if constexpr (T == 2) *reinterpret_cast<u4_t*>(rp) = rs_u4(xf[0], xf[1], xf[ND - 2], xf[ND - 1]);
                            else *reinterpret_cast<uint2*>(rp) = make_uint2(xf[0], xf[1]);
                            if (wave == 0 && gl < 2u) reinterpret_cast<double*>(meta + (pc & bmask))[gl] = __hiloint2double(mhi, mlo);
                        }
                        // lane (c, g) register rr: digit 4 g + rr of column c over the wave's slice (the digits are the product's ROWS): the lane
                        // puts its four digits together -- d0 + 2^8 d1 + 2^16 d2 + 2^24 d3, times 2^32 in k-group 1 (digits 4 .. 6; groups 2 and 3
                        // hold zeros) -- and adds them to the position's accumulator: ONE instruction per group of sixteen columns, two lanes a column
                        // (integer adds in LDS: the eight waves in any order)
                        auto digits_sum = [&](const rl_v4i& a) {
                            const int t01 = a[0] + (a[1] << 8), t23 = a[2] + (a[3] << 8); // (|digit sum| < 2^17: 25 bits each)
                            return ((long long)t01 + ((long long)t23 << 16)) << (32u * (gl & 1u));
                        };
                        {
                            const bool on = mine && gl < 2u;
                            const long long v = digits_sum(a1);
                            if (on) asm volatile("ds_add_u64 %0, %1" ::"v"(lds_addr(acc1 + (pc - Sx))), "v"(v) : "memory");
                            if constexpr (MISS) {
                                if (anym) { // wave-uniform
                                    const long long v2 = digits_sum(a2);
                                    if (on && v2) asm volatile("ds_add_u64 %0, %1" ::"v"(lds_addr(acc2 + (pc - Sx))), "v"(v2) : "memory");
                                }
                            }
                        }
                        // the set's group is admitted in full: its next group leaves HBM now (waited for by hand, at the top of the next round)
                        if (16u * G + 16u <= Sn && !last) load_set(rtag, G + (uint32_t)NG, rl_id_read<T, r>());
                    }
                };
                [&]<int... R>(std::integer_sequence<int, R...>) { (one(std::integral_constant<int, R>{}), ...); }(std::make_integer_sequence<int, NG>{});
            }
        }
        lap(3);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); // the window's new columns and this round's sums are in LDS for every wave
        lap(4);
        if (DBG && wg == 0 && tid == 0) p.progress[1] = ((unsigned long long)seq << 8) | 2u;
        // One lane per refilled position rounds the column's exact sum (units of 2^-44) to the fixed-point units the workgroups' parts travel
        // in (1 / fx_scale: "x + 1.5 2^52", round to nearest) and adds it to the shard's accumulator; once the adds have been performed, one
        // add to the shard's batch counter tells the walker that this workgroup's part is in (first form: the same protocol).  The LAST wave
        // does it all -- it is the one that polls for the next message, and the wait of its first poll (vmcnt counts in order) is the wait
        // for these adds: nobody stands at the barrier while atomics drain, the count goes out behind the first poll (take_message).
        if (wave == RS_WAVES - 1) {
            for (uint32_t t = (uint32_t)lane; t < nnew; t += WAVE) {
                const double MAGIC = 6755399441055744.0;
                const double unit = p.fx_scale * (1.0 / (double)(1ull << RL_EX));
                const long long tot = (long long)acc1[t];
                acc1[t] = 0ull; // (the next round's lanes add behind this round's last barrier)
                if constexpr (MISS) {
                    const long long tot2 = (long long)acc2[t];
                    acc2[t] = 0ull;
                    const double xr = (double)tot2 * unit;
                    if (!(fabs(xr) < 2.2e15)) atomicMax(&p.state->error, 5u);
                    const long long fr = __double_as_longlong(xr + MAGIC) - __double_as_longlong(MAGIC);
                    __hip_atomic_fetch_add(p.racc2 + (size_t)(wg % p.rsh) * RS_RB + ((Sx + t) % RS_RB), (unsigned long long)fr + RS_RONE, HG_RLX_AGENT);
                }
                const double xs = (double)tot * unit;
                if (!(fabs(xs) < 2.2e15)) atomicMax(&p.state->error, 5u); // out of the fixed-point range: the sweep is refused, not wrapped
                const long long fx = __double_as_longlong(xs + MAGIC) - __double_as_longlong(MAGIC);
                __hip_atomic_fetch_add(p.racc + (size_t)(wg % p.rsh) * RS_RB + ((Sx + t) % RS_RB), (unsigned long long)fx + RS_RONE, HG_RLX_AGENT); // (the arrival in the top byte)
            }
            count_due = true;
            if (last) { // (no message follows: the count goes out here)
                wait_vmcnt<0>();
                if (lane == 0) __hip_atomic_fetch_add(p.rcnt + (size_t)(wg % p.rsh) * RS_CROW, 1u, HG_RLX_AGENT);
            }
        }
        lap(5);
        lap(6);
        if (timing) p.trace[7 * RS_TRACE + seq % RS_TRACE] = wall_clock64();
        C = Cn;
        Sx = Sn;
        if (last) break;
        lap(7);
        ++seq;
        take_message(false);
        if (kind == RS_ABORT) break;
    }

    // eps goes back to HBM in the layout the other kernels read (padding slots stay zero)
#pragma unroll
    for (int s = 0; s < ND; ++s) {
        const uint32_t i = De * 16u + eb[s];
        if (De < ndw) p.eps[eps_pos(i)] = ev[s] ? e[s] : 0.0;
    }
    if (timing)
        for (int i = 0; i < 8; ++i) p.state->t[8 + i] = tacc[i];
}

} // namespace hg
