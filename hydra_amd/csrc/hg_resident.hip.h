// hg_resident.hip.h -- the resident sweep engine: ONE launch per Gibbs sweep.
//
// The marker loop of BayesRRm::runMpiGibbs (src/BayesRRm.cpp:1709-2025) is a sequential chain: marker j+1's dot
// needs eps after marker j's update.  The batch engine (hg_sweep.hip.h) pays a kernel launch and ~25 us of
// N-independent hand-off latency per *event* (a marker whose effect changes).  This engine keeps the whole sweep
// inside one kernel and shards the INDIVIDUALS over the compute units, exactly as SURVEY.md 8(e) shards them over
// GPUs -- the compute unit is the rank:
//
//   * streaming workgroups (one per CU, 8 waves): workgroup w owns T wave tiles (T * 1024 individuals).  Every wave
//     keeps the workgroup's eps slice in REGISTERS for the whole sweep (16 * T doubles per lane), so a dot product is
//     field-extract -> int-to-f64 -> fma on registers (no LDS, no address arithmetic) and an update is 16 * T adds.
//     The last B columns (the window) stay in LDS as 2-bit codes: the event column for the update and for the
//     integer Gram terms x_j'x_q that correct the already-taken dots of the window (popcounts, exact).
//   * one walker workgroup (a CU of its own, quiet memory queue): sums the per-workgroup contributions, evaluates the
//     mixture posterior of the window's markers, consumes the shared MT19937 stream in marker order up to the first
//     event, draws its effect (a5-a7) and broadcasts (position, dbeta) as one 16-byte message.
//
// Per event one round: message -> [eps update | Gram terms of the window -> atomics | dots of the columns that refill
// the window -> atomics] -> walker.  The only cross-CU traffic of a round is one 4-byte atomic add per window column and
// one 8-byte add per refilled column from every workgroup, and one 16-byte message.  Sums over workgroups are integers (the
// Gram terms; the raw dots as 51-bit fixed point), accumulated by memory-side atomic adds: exact and order-independent
// (the chain does not depend on the launch geometry).  The Gram words carry their own arrival count in the top byte, the
// raw dots a count of completed refill batches; nothing is ever stored to an accumulator (the walker takes differences to
// what it saw last).  Every spin is bounded (ResParams::timeout): a lost peer ends the sweep with error 3, never a hang.
// Builds: T = 1, 2 wave tiles per workgroup; DBG stage clocks; MISS columns with missing calls (s2 per column, four-term
// Gram sums).  Several ranks (GPUs): one such kernel per rank, replica walkers, sums exchanged through peer mailboxes (RX_*).
#pragma once

#include "hg_sweep.hip.h"

namespace hg {

constexpr int RS_WAVES = 8;
constexpr int RS_BLOCK = RS_WAVES * WAVE; // 512 threads: two waves per SIMD
constexpr int RS_BMAX = 256;              // window capacity (columns kept in LDS as 2-bit codes)
constexpr int RS_NSH = 8;                 // shards of the Gram accumulators = waves of the walker (same-address atomics serialise at ~12 ns each)
constexpr int RS_RSH = 8;                 // shards of the raw-dot accumulators (off the critical chain)
constexpr int RS_GROW = 1024;             // u32 words from one Gram accumulator row to the next (4 KiB: rows on different channels)
constexpr int RS_CROW = 1024;             // the same for the batch counters
constexpr int RS_RB = 2 * RS_BMAX;        // raw-dot accumulators: positions mod RS_RB
constexpr int RS_MSG = 16;                // message slots (seq mod RS_MSG): the walker is never more than RS_MSG - 2 messages ahead of the slowest workgroup
constexpr int RS_PMAX = 4;                // predicted pivots per refill batch whose Gram terms are taken when a column is streamed
constexpr int RS_PFIRE = 16;              // fired pivots whose corrections some not yet arrived column still needs
constexpr int RS_NB = 256;                // refill batches the walker keeps the pivot lists of (a batch holds at least one position of the window)
constexpr int RS_TRACE = 4096;            // messages whose stamps the debug build keeps
constexpr int RS_TMAX = 2;                // most wave tiles per workgroup (16 T doubles of eps per lane, in every wave; a build for 4 spills registers)
constexpr uint32_t RS_ONE = 1u << 24;     // a Gram accumulator word carries its arrival count in bits 24..31, the sum below
constexpr uint32_t RS_LOW = RS_ONE - 1u;
constexpr unsigned long long RS_ONE64 = 1ull << 56; // build MISS: the same in an 8-byte word (the sum below is fixed point, units of 2^-RS_GFX)
constexpr int RS_GFX = 32;                // a workgroup's four-term sum is < 2^15 (four terms of at most 4 * 2048 each): 47 bits, 55 over 255 workgroups
constexpr int RS_FG = 4, RS_FN = 256;      // the walker's tabulated bound: groups it is kept for, intervals of its grid
constexpr int RX_MAXR = 8;                // ranks the engine shards over (one node of eight GPUs; more fall back to the batch engine): the walker keeps a load per peer in flight
constexpr int RS_EVENT_FLAG = 0x100;      // in comp[] during a sweep: the marker was an event (the walker wrote its component); cleared by k_res_finish

enum { RS_EVENT = 0, RS_ADVANCE = 1, RS_ABORT = 2, RS_PIVOT = 3, RS_ANNOUNCE = 4, RS_ANNOUNCED = 5, RS_LAST = 8 }; // message kinds (RS_PIVOT: an event whose Gram terms the walker already has or has asked for); RS_LAST is a flag bit
// RS_ANNOUNCE (second walker): "position C + ncons - 1 WILL be an event -- its marker's effect is non-zero, so whatever is drawn changes it --
// send its Gram terms now".  Nothing else happens: the window stays, no batch is counted.  The announcement stands in the NEXT message's
// slot with that message's number; the message itself (RS_ANNOUNCED: an event whose Gram terms a workgroup that saw the announcement
// has sent already -- one that was busy and never saw it sends them now) overwrites it when the new effect has been drawn, so the Gram
// terms' way to the walker (~3.5 us) runs beside the walker's draw instead of behind it.

// tag = seq << 32 | kind << 28 | check << 12 | ncons: the walker consumed `ncons` positions; RS_EVENT: the last of them changed its
// effect by -dbeta (dbeta = old - new)
struct ResMsg {
    unsigned long long tag;
    double dbeta;
};
// A message is ONE 16-byte store and is read by ONE 16-byte load (a single transaction either way on this hardware); the word with
// kind and count also carries 16 check bits over the other three words, so that a reader that ever saw a torn message would keep
// polling instead of acting on it.  word 0 = kind << 28 | check << 12 | positions consumed (<= RS_BMAX).
__host__ __device__ inline uint32_t rs_msg_check(uint32_t seq, uint32_t lo, uint32_t hi)
{
    uint32_t x = seq * 0x9E3779B1u ^ lo ^ (hi * 0x85EBCA77u);
    x ^= x >> 16;
    return x & 0xffffu;
}
// A workgroup's add to a raw-dot word (racc, racc2) carries its arrival in the top byte, as the Gram words do: the low 56 bits are the
// fixed-point sum (|total| < 2^52 by the scale's bound: it cannot reach the count), the top byte counts the workgroups that have added -- a
// reader takes the difference to what it saw last (the words only ever grow), and a sum that is short of an arrival is never accepted,
// whatever the batch counters say.
constexpr unsigned long long RS_RONE = 1ull << 56;
__host__ __device__ inline unsigned long long rs_raw_value(unsigned long long d) { return (unsigned long long)((long long)(d << 8) >> 8); } // the sum, sign-extended
__host__ __device__ inline uint32_t rs_raw_count(unsigned long long d) { return (uint32_t)((d - rs_raw_value(d)) >> 56) & 0xffu; }
// where the window that starts at position C ends
__host__ __device__ inline uint32_t rs_window_end(uint32_t C, uint32_t B, uint32_t M, uint32_t mask)
{
    const uint32_t e = (C + B) & ~mask;
    return e < M ? e : M;
}
__host__ __device__ inline uint32_t rs_msg_word0(uint32_t kf, uint32_t ncons, uint32_t seq, uint32_t lo, uint32_t hi) { return (kf << 28) | (rs_msg_check(seq, lo, hi) << 12) | ncons; }

struct ResState { // device -> host, written by the walker at the end of the sweep
    uint32_t cursor, rng_idx, error, pad;
    unsigned long long rounds, events, advances, nnz, chunks, refolds, pivots, predicted;
    unsigned long long shader_ticks, wall_ticks; // s_memtime and 100 MHz wall clock over the walker's life: the clock the chip held
    unsigned long long t[16]; // 100 MHz ticks: walker [0] fold [1] collect [2] evaluate [3] scan + draw [4] announce + outputs + prefetch;
                              // streaming workgroup 0: [8] poll [9] update [10] Gram [11] refill dots [12] barrier [13] raw atomics + drain [14] barrier + count [15] prefetch issue
};

struct ResParams {
    const uint8_t* bed;
    uint64_t stride;
    double* eps; // the current buffer, permuted layout (hg_kernels.h); updated in place at the end of the sweep
    uint32_t n_pad, n_local, M;
    double n_minus_1, n_total, eps_sum;
    const int32_t* order;
    const double* s_mave;
    const double* s_mstd;
    const double* s_bold;
    const int32_t* s_ga;
    double* beta;
    int32_t* comp;
    double* acum;
    int32_t* cass;
    int K, GK;
    const double* denom;
    const double* logpi;
    const double* hlog;
    const double* sdk;
    double i_2sigE;
    uint32_t* mt;
    ZigTables zig;
    uint32_t rng_idx;
    uint32_t W;   // streaming workgroups; the walker is workgroup W
    uint32_t B;   // window (power of two, <= RS_BMAX, B * T <= 512)
    uint32_t nsh, rsh;
    uint32_t* gacc;           // [2][RS_NSH] rows of RS_GROW words, RS_BMAX in use: count << 24 | sum of the workgroups' Gram terms
    unsigned long long* racc; // [RS_RSH][RS_RB]: sum of the workgroups' raw dots (51-bit fixed point) of position p at p mod RS_RB
    uint32_t* rcnt;           // [RS_RSH] words RS_CROW apart: refill batches the shard's workgroups have completed
    unsigned long long* racc2; // [RS_RSH][RS_RB] (build MISS): the same for R = sum of eps over the column's missing calls (s2 = sum of eps - R)
    unsigned long long* gacc64; // [2][RS_NSH] rows of RS_GROW 8-byte words (build MISS): count << 56 | fixed-point sum of the workgroups' four-term Gram sums
    const unsigned long long* counts; // [M][3] (n1, n2, missing) by marker, summed over the ranks
    unsigned long long* pacc; // [RS_RSH][RS_RB][2]: sums of the workgroups' Gram terms of position p (at p mod RS_RB) with its batch's pivots, two 32-bit
                              // fields to a word (pivots 0 | 1 << 32, 2 | 3 << 32: a batch adds < 2^32 to a field, and the walker decodes DIFFERENCES)
    ResMsg* msg;              // [RS_MSG]
    ResState* state;
    double fx_scale, fx_unscale; // raw dots travel as round(dot * fx_scale), |.| < 2^51 per workgroup
    unsigned long long timeout;  // 100 MHz ticks a spin may last
    unsigned long long* progress; // device memory: [0] walker (round << 8 | stage), [1] streaming workgroup 0 (message << 8 | stage): what the host
                                  // reports when its deadline passes (a store to host memory here would put a PCIe round trip in front of the next barrier)
    // progress[2]: the host's abort word (non-zero: give the sweep up -- polled where a workgroup waits); progress[3]: workgroups that have
    // arrived at the kernel's start (the rendezvous below)
    unsigned long long rdv_timeout; // 100 MHz ticks the start-of-kernel rendezvous may last
    unsigned long long* trace;   // debug_timing: [8][RS_TRACE] wall-clock stamps of the last RS_TRACE messages (tools/res_anatomy.py)
    int dbg;
    int pivots; // 1: Gram terms with predicted pivots are taken when a column is streamed (messages RS_PIVOT need no round trip)
    int tune;   // experiments (option res_tune): bits 0-1: priority of the younger wave of each SIMD (waves 4 .. 7) in the refill
    int early_advance; // second walker: a walk that has run out of dots moves the window on at once (a message that only advances) when at least this many positions have passed (0: it waits)
    uint32_t wend_mask; // 15 (second walker, B >= 32): the window ends at a multiple of sixteen positions -- min((C + B) & ~15, M) -- so that the
                        // streaming workgroups' second form admits every group of sixteen columns in ONE round (else 0: the window is C + B)
    int announce; // second walker: 1: an event that is certain (a marker with a non-zero effect) is announced to the streaming workgroups before its draw (RS_ANNOUNCE)
    int walker; // 2: the second walker (hg_walker2.hip.h: one wave walks the chain, the others serve it), else the first
    const uint32_t* pred; // the sweep positions whose marker has a non-zero effect at sweep start (predicted events), ascending, then 16 sentinels 0xffffffff
    int all_ada; // 1: no marker is frozen out (adaV all ones, the usual case): a marker's uniform is its distance from the cursor
    // several GPUs (individuals sharded over the ranks): every rank runs this kernel on its shard, the walkers are
    // replicas that decide on the SAME integer sums -- each adds its peers' parts, which arrive in its mailbox (RX_* below)
    int nranks, rank;
    unsigned char* mbox[RX_MAXR]; // mbox[r]: rank r's resident mailbox (IPC-mapped; [rank] is local memory)
    unsigned long long sweep_id;    // this sweep's epoch in the mailbox flags (the ranks count their sweeps alike)
};

// Resident mailbox of a rank, written by its peers with system-scope 8-byte stores (xGMI; uncached memory).  Indexed by the SENDER's rank:
//   gbox[src][parity][RS_BMAX] u64   sweep_id << 48 | n << 24 | the sender's sum over its workgroups of the Gram terms of event n (parity n & 1):
//                                    self-validating words, no flag
//   (build MISS: gbox holds two words per column, tag << 32 | half of the sender's 56-bit fixed-point four-term sum, tag = 1 | sweep | event;
//    rbox two more words per position for R, the sum over the column's missing calls)
//   rbox[src][RX_RING][4] u64        the sender's sum of the raw dots (fixed point, 64 bits) of position p, at p mod RX_RING, as two
//                                    words tag << 32 | low half, tag << 32 | high half with tag = 1 (never the zero of fresh memory) | sweep (11 bits) | the position's refill batch (20)
// Every word says itself what it is: no flag has to be ordered behind the data (one hop; the reader polls the words it needs).
// A peer is never more than one event ahead (it needs this rank's part of event n + 1 to get past it) nor more than a window's
// worth of positions (a position's dot needs every rank's part): two parities and a ring of four windows are enough.
constexpr uint32_t RX_RING = 1024;
constexpr size_t RX_GBOX = 0;
constexpr size_t RX_RBOX = RX_GBOX + (size_t)RX_MAXR * 2 * RS_BMAX * 16;
//   hbox[src][2] u64                 the probe launch's handshake (rs_probe_peers): "my grid is resident" and "so is everybody's, as far as I see"
constexpr size_t RX_HBOX = RX_RBOX + (size_t)RX_MAXR * RX_RING * 32;
constexpr size_t RX_BYTES = RX_HBOX + (size_t)RX_MAXR * 2 * 8;
__device__ __forceinline__ unsigned long long* rx_gbox(unsigned char* mb, int src, uint32_t par) { return reinterpret_cast<unsigned long long*>(mb + RX_GBOX) + ((size_t)src * 2 + par) * RS_BMAX * 2; } // two words per column (build MISS uses both)
__device__ __forceinline__ unsigned long long* rx_rbox(unsigned char* mb, int src) { return reinterpret_cast<unsigned long long*>(mb + RX_RBOX) + (size_t)src * RX_RING * 4; } // four words per position: the halves of s1 and (build MISS) of R
__device__ __forceinline__ unsigned long long* rx_hbox(unsigned char* mb, int src) { return reinterpret_cast<unsigned long long*>(mb + RX_HBOX) + (size_t)src * 2; }
__device__ __forceinline__ unsigned long long rx_rtag(unsigned long long sweep, uint32_t batch) { return (0x80000000ull | ((sweep & 0x7ffull) << 20) | (unsigned long long)(batch & 0xfffffu)) << 32; }

typedef uint32_t u4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(4))) uint32_t rs_cu32; // read-only global memory through the scalar cache
__device__ __forceinline__ u4_t rs_load16(const void* p)
{
    u4_t v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
// A barrier for what the workgroup shares through LDS only: __syncthreads() also waits for every store to global memory the wave has in
// flight (results, pushes to the peers' mailboxes: a round trip each) -- for nothing where no other thread of the workgroup reads them.
__device__ __forceinline__ void rs_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void rs_store16(void* p, u4_t v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ double rs_readlane(double v, int l)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
// one DPP step of a 64-bit value (every lane reads a valid lane: no old value needed)
template <int CTRL>
__device__ __forceinline__ double rs_dpp_f64(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ u4_t rs_u4(uint32_t x, uint32_t y, uint32_t z, uint32_t w)
{
    u4_t v;
    v.x = x;
    v.y = y;
    v.z = z;
    v.w = w;
    return v;
}

__host__ __device__ constexpr size_t rs_streamer_lds(uint32_t B, int T) { return 512 + (size_t)B * 97 + (size_t)B * 256 * T; }
// build MISS: a layout of its own in front of the ring -- (mave, mstd) of the window slots (16 B each), this round's refill: four
// 16-lane partial sums of s1 per position (32 B) and one 8-byte INTEGER accumulator of R = sum of eps over the column's missing calls
// (fixed point: the lanes add their parts with LDS atomics -- integers, so the order does not matter), and a copy of the workgroup's eps
// slice addressable by individual ([(16 t + slot) * 64 + lane] doubles) for the gather over a column's (few) missing calls.
__host__ __device__ constexpr size_t rs_miss_part_off(uint32_t B) { return 512 + (size_t)B * 16; }
__host__ __device__ constexpr size_t rs_miss_racc_off(uint32_t B) { return 512 + (size_t)B * 48; }
__host__ __device__ constexpr size_t rs_epsl_off(uint32_t B, int T) { return 512 + (size_t)B * 56; }
__host__ __device__ constexpr size_t rs_miss_ring_off(uint32_t B, int T) { return rs_epsl_off(B, T) + (size_t)8192 * T; }
__host__ __device__ constexpr size_t rs_streamer_lds_miss(uint32_t B, int T) { return rs_miss_ring_off(B, T) + (size_t)B * 256 * T; }
static_assert(rs_streamer_lds(RS_BMAX, RS_TMAX) <= 160 * 1024 && rs_streamer_lds_miss(RS_BMAX, RS_TMAX) <= 160 * 1024 && rs_streamer_lds_miss(RS_BMAX, 1) <= 160 * 1024,
              "the largest window at the most tiles per workgroup fits the 160 KB of LDS of a compute unit, in both builds");

// ---------------------------------------------------------------------------------------------------------------
// streaming workgroup
// ---------------------------------------------------------------------------------------------------------------
// The refill's columns land in the TOP 32 vector registers, named by hand: v224 .. v255 are outside what the compiler may allocate
// (the kernel carries a register cap of RS_VGPR_LIMIT, see below), so nothing but the instructions below ever touches them.  The loads are
// instructions the compiler does not see as loads: it would otherwise place its own wait before the first use of a loaded register,
// and the only wait it can place there is for ALL vector-memory operations in flight (vmcnt counts in order, the loads are issued
// under wave-uniform conditions it cannot count) -- for loads issued microseconds ago AND for the ones just issued for the next
// round.  Registers of its own cannot serve: it is free to copy a loop-carried value to another register right behind the
// instruction that "defined" it, i.e. before the load has landed (it does: a 16-register shuffle at the loop's back edge).  The
// wait is placed by hand, once per pass (cols_landed); a set is read only behind it.  Set R = v(224 + R) for T = 1, v[224 + 2R : 225 + 2R]
// for T = 2.  tools/asm_check_loads.py verifies on the emitted code that no other instruction names these registers.
constexpr int RS_VGPR_LIMIT = 216; // v216 .. v220: the lanes' (mave, mstd, next id), v224 .. v255: the sets
// (amdgpu_num_vgpr counts the unified VGPR + AGPR file of this target -- the compiler doubles the attribute's value: the kernels carry HALF
// the limit, which caps the registers the compiler allocates itself at the limit; the named registers in the clobber lists of the inline
// assembly keep the wave's allocation at 256.  Measured with a small kernel: amdgpu_num_vgpr(216) caps nothing, (108) caps at 216.)
#define RS_SET_LIST(X) X(0, 224, 224, 225) X(1, 225, 226, 227) X(2, 226, 228, 229) X(3, 227, 230, 231) X(4, 228, 232, 233) X(5, 229, 234, 235) X(6, 230, 236, 237) X(7, 231, 238, 239) X(8, 232, 240, 241) X(9, 233, 242, 243) X(10, 234, 244, 245) X(11, 235, 246, 247) X(12, 236, 248, 249) X(13, 237, 250, 251) X(14, 238, 252, 253) X(15, 239, 254, 255)
template <int T, int R>
__device__ __forceinline__ void rs_set_load(uint32_t voff, const uint8_t* base)
{
    static_assert(T == 1 || T == 2, "one or two dwords per lane and column");
#define RS_X(r, s1, lo, hi)                                                                                                                 \
    if constexpr (R == r) {                                                                                                                 \
        if constexpr (T == 1) asm volatile("global_load_dword v" #s1 ", %0, %1" ::"v"(voff), "s"(base) : "memory", "v" #s1);                \
        else asm volatile("global_load_dwordx2 v[" #lo ":" #hi "], %0, %1" ::"v"(voff), "s"(base) : "memory", "v" #lo, "v" #hi);            \
    }
    RS_SET_LIST(RS_X)
#undef RS_X
}
// the set's dwords, masked (keep: the lanes' valid individuals)
template <int T, int R>
__device__ __forceinline__ void rs_set_read(uint32_t (&w)[T], const uint32_t (&keep)[T])
{
#define RS_X(r, s1, lo, hi)                                                                                                                 \
    if constexpr (R == r) {                                                                                                                 \
        if constexpr (T == 1) asm volatile("v_and_b32 %0, v" #s1 ", %1" : "=v"(w[0]) : "v"(keep[0]));                                       \
        else asm volatile("v_and_b32 %0, v" #lo ", %2\n\tv_and_b32 %1, v" #hi ", %3" : "=&v"(w[0]), "=&v"(w[T - 1]) : "v"(keep[0]), "v"(keep[T - 1])); \
    }
    RS_SET_LIST(RS_X)
#undef RS_X
}

// per-lane values that travel with the sets (their loads must not make the compiler wait either): lane r keeps (mave, mstd) of set r's
// column in v[216:217], v[218:219], its effect at sweep start in v[222:223] and the marker id of the set's NEXT column in v220 -- named registers as well
__device__ __forceinline__ void rs_lane_load(const double* mave, const double* mstd, const double* bold, const int32_t* id, const int32_t* ga)
{
    asm volatile("global_load_dwordx2 v[216:217], %0, off\n\tglobal_load_dwordx2 v[218:219], %1, off\n\tglobal_load_dwordx2 v[222:223], %2, off\n\tglobal_load_dword v220, %3, off\n\t"
                 "global_load_dword v221, %4, off" ::"v"(mave),
                 "v"(mstd), "v"(bold), "v"(id), "v"(ga)
                 : "memory", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223");
}
// the columns' (group | flags) words: bit 29 = the column has missing calls
__device__ __forceinline__ int32_t rs_lane_ga()
{
    int32_t v;
    asm volatile("v_mov_b32 %0, v221" : "=v"(v));
    return v;
}
__device__ __forceinline__ double2 rs_lane_meta()
{
    int a0, a1, b0, b1;
    asm volatile("v_mov_b32 %0, v216\n\tv_mov_b32 %1, v217\n\tv_mov_b32 %2, v218\n\tv_mov_b32 %3, v219" : "=&v"(a0), "=&v"(a1), "=&v"(b0), "=&v"(b1));
    return make_double2(__hiloint2double(a1, a0), __hiloint2double(b1, b0));
}
// the column's effect at sweep start (non-zero: a predicted pivot)
__device__ __forceinline__ double rs_lane_bold()
{
    int a0, a1;
    asm volatile("v_mov_b32 %0, v222\n\tv_mov_b32 %1, v223" : "=&v"(a0), "=&v"(a1));
    return __hiloint2double(a1, a0);
}
// the lanes' next ids as an ordinary value (read behind the wait; the compiler then takes care of the readlane hazards itself)
__device__ __forceinline__ int32_t rs_lane_ids()
{
    int32_t v;
    asm volatile("v_mov_b32 %0, v220" : "=v"(v));
    return v;
}

// The 64-lane sums of sixteen accumulators, each a pair of 16-bit column sums, scattered: lane c < 32 gets the sum of column c
// (accumulator c >> 1, half c & 1).  A reduce-scatter -- at every step a lane keeps the half of its values that matches its
// next lane bit and adds its partner's -- costs ~60 instructions instead of the ~180 of sixteen separate wave sums: after four
// DPP steps (lane ^ 1, ^ 2, ^ 4, ^ 8; the last two as two mirrors each) lane l holds its row's sum of accumulator l & 15, two
// crossbar steps add the four rows, one more brings accumulator c >> 1 to lane c.
__device__ __forceinline__ uint32_t wave_sum16_rows(const uint32_t (&acc)[16], int lane)
{
    auto dpp = [](uint32_t v, auto ctrl) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, decltype(ctrl)::value, 0xF, 0xF, false); };
    using QX1 = std::integral_constant<int, 0xB1>;  // quad_perm [1,0,3,2]: lane ^ 1
    using QX2 = std::integral_constant<int, 0x4E>;  // quad_perm [2,3,0,1]: lane ^ 2
    using QX3 = std::integral_constant<int, 0x1B>;  // quad_perm [3,2,1,0]: lane ^ 3
    using HM = std::integral_constant<int, 0x141>;  // row_half_mirror: lane ^ 7
    using RM = std::integral_constant<int, 0x140>;  // row_mirror: lane ^ 15
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4, b3 = lane & 8;
    uint32_t b[8], c[4], d[2];
#pragma unroll
    for (int i = 0; i < 8; ++i) b[i] = (b0 ? acc[2 * i + 1] : acc[2 * i]) + dpp(b0 ? acc[2 * i] : acc[2 * i + 1], QX1{});
#pragma unroll
    for (int i = 0; i < 4; ++i) c[i] = (b1 ? b[2 * i + 1] : b[2 * i]) + dpp(b1 ? b[2 * i] : b[2 * i + 1], QX2{});
#pragma unroll
    for (int i = 0; i < 2; ++i) d[i] = (b2 ? c[2 * i + 1] : c[2 * i]) + dpp(dpp(b2 ? c[2 * i] : c[2 * i + 1], QX3{}), HM{}); // ^ 3 then ^ 7 = ^ 4
    uint32_t v = (b3 ? d[1] : d[0]) + dpp(dpp(b3 ? d[0] : d[1], HM{}), RM{});                                                    // ^ 7 then ^ 15 = ^ 8
    v += (uint32_t)__shfl_xor((int)v, 16, 64);
    v += (uint32_t)__shfl_xor((int)v, 32, 64);
    return v; // every lane: the wave's sum of accumulator lane & 15
}
__device__ __forceinline__ uint32_t wave_sum16_scatter(const uint32_t (&acc)[16], int lane)
{
    const uint32_t w = (uint32_t)__shfl((int)wave_sum16_rows(acc, lane), lane >> 1, 64);
    return (lane & 1) ? (w >> 16) : (w & 0xffffu);
}

// four 64-lane integer sums at once (hg_sweep.hip.h: wave_sum_u32), step by step over the four: totals returned wave-uniform
__device__ __forceinline__ void wave_sum_u32x4(uint32_t (&v)[4])
{
#define RS_STEP(ctrl, rmask)                                                                                   \
    _Pragma("unroll") for (int k = 0; k < 4; ++k) v[k] += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v[k], ctrl, rmask, 0xF, false);
    RS_STEP(0xB1, 0xF)  // quad_perm [1,0,3,2]
    RS_STEP(0x4E, 0xF)  // quad_perm [2,3,0,1]
    RS_STEP(0x141, 0xF) // row_half_mirror
    RS_STEP(0x140, 0xF) // row_mirror
    RS_STEP(0x142, 0xA) // row_bcast15
    RS_STEP(0x143, 0xC) // row_bcast31
#undef RS_STEP
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = (uint32_t)__builtin_amdgcn_readlane((int)v[k], 63);
}

// 64-lane sum of a double on the DPP path, fixed order; the total is in lane 63 (rows that a step does not write add +0.0)
__device__ __forceinline__ double rs_wave_sum_f64(double v)
{
    auto step = [&](auto ctrl, auto rmask) {
        constexpr int C = decltype(ctrl)::value, RM = decltype(rmask)::value;
        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), C, RM, 0xF, false);
        const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), C, RM, 0xF, false);
        v += __hiloint2double(hi, lo);
    };
    step(std::integral_constant<int, 0xB1>{}, std::integral_constant<int, 0xF>{});  // quad_perm [1,0,3,2]
    step(std::integral_constant<int, 0x4E>{}, std::integral_constant<int, 0xF>{});  // quad_perm [2,3,0,1]
    step(std::integral_constant<int, 0x141>{}, std::integral_constant<int, 0xF>{}); // row_half_mirror
    step(std::integral_constant<int, 0x140>{}, std::integral_constant<int, 0xF>{}); // row_mirror
    step(std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xA>{}); // row_bcast15
    step(std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xC>{}); // row_bcast31
    return v;
}

// A pivot's masks from its x form (hg_sweep.hip.h: u = [g >= 1] in the even bit, v = [g == 2] in the odd one)
__device__ __forceinline__ GramPivot gram_pivot_x(uint32_t x)
{
    const uint32_t ue = x & 0x55555555u, vo = x & 0xAAAAAAAAu;
    return GramPivot{ue | (ue << 1), vo | (vo >> 1)};
}

// The dot product's inner step for ONE column: four slots S .. S + 3 of a dword into four accumulators (field extract, int -> f64,
// fused multiply-add, issued as three groups of four: three independent instructions between a producer and its consumer).
// a_i += double((g >> 2 (4 Q + i)) & 3) * e[4 Q + i] -- the products are exact, one rounding per add.
template <int Q>
__device__ __forceinline__ void fma_col4(uint32_t g, const double (&e)[IPT], double& a0, double& a1, double& a2, double& a3)
{
    uint32_t t0, t1, t2, t3;
    double w0, w1, w2, w3;
    asm("v_bfe_u32 %[t0], %[g], %[s0], 2\n\t"
        "v_bfe_u32 %[t1], %[g], %[s1], 2\n\t"
        "v_bfe_u32 %[t2], %[g], %[s2], 2\n\t"
        "v_bfe_u32 %[t3], %[g], %[s3], 2\n\t"
        "v_cvt_f64_u32 %[w0], %[t0]\n\t"
        "v_cvt_f64_u32 %[w1], %[t1]\n\t"
        "v_cvt_f64_u32 %[w2], %[t2]\n\t"
        "v_cvt_f64_u32 %[w3], %[t3]\n\t"
        "v_fmac_f64 %[a0], %[w0], %[e0]\n\t"
        "v_fmac_f64 %[a1], %[w1], %[e1]\n\t"
        "v_fmac_f64 %[a2], %[w2], %[e2]\n\t"
        "v_fmac_f64 %[a3], %[w3], %[e3]"
        : [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3),
          [w0] "=&v"(w0), [w1] "=&v"(w1), [w2] "=&v"(w2), [w3] "=&v"(w3)
        : [g] "v"(g), [e0] "v"(e[4 * Q]), [e1] "v"(e[4 * Q + 1]), [e2] "v"(e[4 * Q + 2]), [e3] "v"(e[4 * Q + 3]), [s0] "i"(8 * Q), [s1] "i"(8 * Q + 2),
          [s2] "i"(8 * Q + 4), [s3] "i"(8 * Q + 6));
}
template <int... Q>
__device__ __forceinline__ void fma_col(uint32_t g, const double (&e)[IPT], double& a0, double& a1, double& a2, double& a3, std::integer_sequence<int, Q...>)
{
    (fma_col4<Q>(g, e, a0, a1, a2, a3), ...);
}
// The same for a whole dword (sixteen slots) as ONE instruction block -- between separate blocks the compiler pads with s_nop, and every
// instruction a wave issues costs it the same four to five clocks -- and, where the dword is the column's first (FIRST), with the
// accumulators STARTED by the first four products (v_mul_f64: no four moves of zero in front; the products are exact either way).
template <bool FIRST>
__device__ __forceinline__ void fma_dword(uint32_t g, const double (&e)[IPT], double& a0, double& a1, double& a2, double& a3)
{
    uint32_t t0, t1, t2, t3;
    double w0, w1, w2, w3;
#define HG_FD_BLOCK(S0, S1, S2, S3, E0, E1, E2, E3, OP)                                                                                    \
    "v_bfe_u32 %[t0], %[g], " #S0 ", 2\n\tv_bfe_u32 %[t1], %[g], " #S1 ", 2\n\tv_bfe_u32 %[t2], %[g], " #S2 ", 2\n\tv_bfe_u32 %[t3], %[g], " #S3 ", 2\n\t"   \
    "v_cvt_f64_u32 %[w0], %[t0]\n\tv_cvt_f64_u32 %[w1], %[t1]\n\tv_cvt_f64_u32 %[w2], %[t2]\n\tv_cvt_f64_u32 %[w3], %[t3]\n\t" OP(E0, E1, E2, E3)
#define HG_FD_FMAC(E0, E1, E2, E3) "v_fmac_f64 %[a0], %[w0], %[" #E0 "]\n\tv_fmac_f64 %[a1], %[w1], %[" #E1 "]\n\tv_fmac_f64 %[a2], %[w2], %[" #E2 "]\n\tv_fmac_f64 %[a3], %[w3], %[" #E3 "]\n\t"
#define HG_FD_MUL(E0, E1, E2, E3) "v_mul_f64 %[a0], %[w0], %[" #E0 "]\n\tv_mul_f64 %[a1], %[w1], %[" #E1 "]\n\tv_mul_f64 %[a2], %[w2], %[" #E2 "]\n\tv_mul_f64 %[a3], %[w3], %[" #E3 "]\n\t"
#define HG_FD_OPERANDS                                                                                                                     \
    [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [w0] "=&v"(w0), [w1] "=&v"(w1), [w2] "=&v"(w2), [w3] "=&v"(w3)               \
        : [g] "v"(g), [e0] "v"(e[0]), [e1] "v"(e[1]), [e2] "v"(e[2]), [e3] "v"(e[3]), [e4] "v"(e[4]), [e5] "v"(e[5]), [e6] "v"(e[6]), [e7] "v"(e[7]), [e8] "v"(e[8]),  \
          [e9] "v"(e[9]), [e10] "v"(e[10]), [e11] "v"(e[11]), [e12] "v"(e[12]), [e13] "v"(e[13]), [e14] "v"(e[14]), [e15] "v"(e[15])
    if constexpr (FIRST) {
        asm(HG_FD_BLOCK(0, 2, 4, 6, e0, e1, e2, e3, HG_FD_MUL) HG_FD_BLOCK(8, 10, 12, 14, e4, e5, e6, e7, HG_FD_FMAC) HG_FD_BLOCK(16, 18, 20, 22, e8, e9, e10, e11, HG_FD_FMAC)
                HG_FD_BLOCK(24, 26, 28, 30, e12, e13, e14, e15, HG_FD_FMAC)
            : [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2), [a3] "=&v"(a3), HG_FD_OPERANDS);
    } else {
        asm(HG_FD_BLOCK(0, 2, 4, 6, e0, e1, e2, e3, HG_FD_FMAC) HG_FD_BLOCK(8, 10, 12, 14, e4, e5, e6, e7, HG_FD_FMAC) HG_FD_BLOCK(16, 18, 20, 22, e8, e9, e10, e11, HG_FD_FMAC)
                HG_FD_BLOCK(24, 26, 28, 30, e12, e13, e14, e15, HG_FD_FMAC)
            : [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3), HG_FD_OPERANDS);
    }
#undef HG_FD_BLOCK
#undef HG_FD_FMAC
#undef HG_FD_MUL
#undef HG_FD_OPERANDS
}

// The integer Gram terms A_jq = sum_i g_ij g_iq of the window columns behind the event column q, by one streaming workgroup (both forms:
// the window's codes are [slot][64 T dwords] in either): wave w takes the columns [w Vw, (w + 1) Vw) of the V = Sx - (q + 1) behind q, all of
// the workgroup's individuals, and sends its sums itself -- memory-side atomic adds, the arrival count in the word's top byte.  xq: the
// lane's dwords of column q, mq: (mave, mstd) of q, nev: the event's number (its parity selects the accumulator row).
template <int T, int MISS>
__device__ __forceinline__ void rs_gram_terms(const ResParams& p, const uint32_t* ring, const double2* meta, uint32_t wg, int wave, int lane, uint32_t q, uint32_t Sx, uint32_t bmask,
                                              uint32_t nev, const uint32_t (&xq)[T], const double2& mq)
{
    constexpr bool with_gram = true;
    const uint32_t V = Sx - (q + 1u);
    const uint32_t Vw = (V + 7u) / 8u, i0 = (uint32_t)wave * Vw;
    if constexpr (MISS) {
    if (V && with_gram) {
        // Four sums where a call may be missing in either column (x_j'x_q = mstd_j mstd_q (A - m_q B - m_j C + m_j m_q D), sums over
        // the individuals called in both): with the missing calls' fields cleared, A = sum g_j g_q as before, B = G_j - P,
        // C = G_q - Q, D = N - nm_j - nm_q + X with P = sum of g_j over q's missing calls, Q = sum of g_q over j's, X = calls
        // missing in both -- popcounts against the missing masks.  What depends on the individuals, A + m_q P + m_j Q + m_j m_q X
        // (>= 0), is summed over the wave as a double and sent as ONE fixed-point word per column (units of 2^-RS_GFX, arrival
        // count in the top byte); the walker adds the rest from the markers' counts.
        GramPivot gp[T];
        uint32_t xqc[T], mq1[T], mq2[T];
#pragma unroll
        for (int t = 0; t < T; ++t) { // window code 10 = missing: its mask, and the x form with those fields cleared
            mq1[t] = (xq[t] >> 1) & ~xq[t] & 0x55555555u;
            mq2[t] = mq1[t] | (mq1[t] << 1);
            xqc[t] = xq[t] & ~mq2[t];
            gp[t] = gram_pivot_x(xqc[t]);
        }
        const double mqv = mq.x;
        // the four integer sums of a column meet packed (A' | P' << 16, Q | X << 16: a lane adds at most 192 / 96 / 64 / 32, the wave
        // 12288 / 6144 / 4096 / 2048), sixteen columns to a reduce-scatter: lane c < 16 then holds column c0 + c's totals, forms its
        // term and sends it
        for (uint32_t c0 = 0; c0 < Vw; c0 += 16u) { // wave-uniform
            uint32_t ap[16], qx[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const uint32_t i = i0 + c0 + (uint32_t)k;
                const uint32_t slot = (q + 1u + (i < V ? i : V - 1u)) & bmask;
                const uint32_t* rp = ring + slot * 64u * T + (uint32_t)lane * T;
                uint32_t A = 0u, P = 0u, Q = 0u, X = 0u;
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    // (the column's word is taken as it stands: its missing fields, code 10, count 1 in a popcount -- A' = A + Q and
                    // P' = P + X -- and the integers are put right behind the wave sums: no cleared copy of the word is made)
                    const uint32_t w = rp[t];
                    const uint32_t mj1 = ~(w | 0xAAAAAAAAu) & (w >> 1), mj2 = mj1 | (mj1 << 1); // (one v_bfi_b32)
                    A += gram16x(w, gp[t]);
                    P += (uint32_t)__popc(w & mq2[t]);
                    Q += (uint32_t)__popc(xqc[t] & mj2);
                    X += (uint32_t)__popc(mj1 & mq1[t]);
                }
                ap[k] = A | (P << 16);
                qx[k] = Q | (X << 16);
            }
            const uint32_t apt = wave_sum16_rows(ap, lane), qxt = wave_sum16_rows(qx, lane);
            const uint32_t mycol = i0 + c0 + ((uint32_t)lane & 15u);
            if ((uint32_t)lane < 16u && c0 + (uint32_t)lane < Vw && mycol < V) {
                const double mj = meta[(q + 1u + mycol) & bmask].x;
                const uint32_t Qs = qxt & 0xffffu, Xs = qxt >> 16, As = (apt & 0xffffu) - Qs, Ps = (apt >> 16) - Xs; // (A = A' - Q, P = P' - X: exact)
                const double mine = ((double)As + mqv * (double)Ps) + (mj * (double)Qs + (mj * mqv) * (double)Xs);
                const double MAGIC = 6755399441055744.0;
                const unsigned long long fx = (unsigned long long)(__double_as_longlong(mine * (double)(1ull << RS_GFX) + MAGIC) - __double_as_longlong(MAGIC));
                __hip_atomic_fetch_add(p.gacc64 + ((size_t)(nev & 1u) * RS_NSH + (wg % p.nsh)) * RS_GROW + ((q + 1u + mycol) & bmask), RS_ONE64 | fx, HG_RLX_AGENT);
            }
        }
    }
    } else
    if (V && with_gram) {
    GramPivot gp[T];
#pragma unroll
    for (int t = 0; t < T; ++t) gp[t] = gram_pivot_x(xq[t]);
    uint32_t acc[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0u;
#pragma unroll
    for (int cb = 0; cb < 32; cb += 8) {
        if ((uint32_t)cb < Vw) { // wave-uniform; the eight columns of a group are read together (past the end: column V - 1 again, unused)
            uint32_t wv[8][T];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const uint32_t i = i0 + (uint32_t)(cb + c);
                const uint32_t slot = (q + 1u + (i < V ? i : V - 1u)) & bmask;
                const uint32_t* rp = ring + slot * 64u * T + (uint32_t)lane * T;
#pragma unroll
                for (int t = 0; t < T; ++t) wv[c][t] = rp[t];
            }
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                uint32_t g = 0u;
#pragma unroll
                for (int t = 0; t < T; ++t) g += gram16x(wv[c][t], gp[t]);
                acc[(cb + c) >> 1] += g << (16 * (c & 1)); // a lane adds at most 64 T <= 256 per column: the 64-lane sum fits 16 bits
            }
        }
    }
    const uint32_t mine = wave_sum16_scatter(acc, lane); // lane c: the wave's Gram term of its column c
    if ((uint32_t)lane < Vw && i0 + (uint32_t)lane < V) // one instruction per wave, contiguous words (by window slot of the column, up to the wrap): count in the top byte
        __hip_atomic_fetch_add(p.gacc + ((size_t)(nev & 1u) * RS_NSH + (wg % p.nsh)) * RS_GROW + ((q + 1u + i0 + (uint32_t)lane) & bmask), RS_ONE | mine, HG_RLX_AGENT);
    }
}

template <int T, int DBG, int MISS>
__device__ __forceinline__ void res_streamer(const ResParams& p, unsigned char* smem)
{
    constexpr int RS_PF = 16; // register sets: columns one wave has in registers or on their way
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t wg = blockIdx.x;
    const uint32_t B = p.B, bmask = B - 1u, M = p.M;
    double2* const tab = reinterpret_cast<double2*>(smem);                        // pair table of the event's addends (256 B)
    unsigned long long* const lmsg = reinterpret_cast<unsigned long long*>(smem + 256); // the message, as the polling lane read it
    double2* const meta = reinterpret_cast<double2*>(smem + 512);                 // (mave, mstd) of the window slots
    double* const part = reinterpret_cast<double*>(smem + 512 + (size_t)B * 16);       // this round's refill: [position - Sx][8] sums of the wave's eight-lane groups (build MISS: [.][4] sums of its 16-lane rows)
    unsigned long long* const rlds = reinterpret_cast<unsigned long long*>(smem + rs_miss_racc_off(B)); // build MISS: [position - Sx] fixed-point sum of eps over the column's missing calls
    uint32_t* const ring = reinterpret_cast<uint32_t*>(smem + (MISS ? rs_miss_ring_off(B, T) : 512 + (size_t)B * 97)); // [B][64 * T] codes of the window columns
    double* const epsl = reinterpret_cast<double*>(smem + rs_epsl_off(B, T));           // build MISS: the eps slice by individual (see rs_epsl_off)
    uint32_t* const fin_cnt = reinterpret_cast<uint32_t*>(smem + 320);                  // waves of this round that are through their atomics
    if (tid == 0) *fin_cnt = 0u;
    const bool timing = DBG && wg == 0 && tid == 0;
    unsigned long long* const tacc = reinterpret_cast<unsigned long long*>(smem + 384); // [8] stage clocks of the debug build (in LDS: sixteen registers less)
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

    // this lane's T dwords of every column: dwords [d0, d0 + T) = individuals [16 d0, 16 (d0 + T)) of the shard
    const uint32_t d0 = (wg * 64u + (uint32_t)lane) * (uint32_t)T;
    const uint32_t ndw = p.n_pad >> 4;
    const bool vgrp = d0 < ndw; // T divides ndw (a multiple of 256): all of the lane's dwords or none
    const uint32_t voff = vgrp ? d0 * 4u : 0u;
    uint32_t keep[T];
    double e[T][IPT];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const uint32_t i0 = (d0 + (uint32_t)t) * 16u;
        const uint32_t nv = (!vgrp || i0 >= p.n_local) ? 0u : (p.n_local - i0 >= 16u ? 16u : p.n_local - i0);
        keep[t] = nv >= 16u ? 0xffffffffu : ((1u << (2u * nv)) - 1u);
#pragma unroll
        for (int s = 0; s < IPT; ++s) e[t][s] = vgrp ? p.eps[eps_pos(i0 + (uint32_t)s)] : 0.0;
    }

    // build MISS: every wave holds the same eps; wave w keeps slots w and w + 8 of the LDS copy current
    auto eps_to_lds = [&]() {
        if constexpr (MISS) {
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int sl = 0; sl < IPT; ++sl)
                    if ((sl & 7) == wave) epsl[(uint32_t)(t * IPT + sl) * 64u + (uint32_t)lane] = e[t][sl]; // wave-uniform
        }
    };
    eps_to_lds();
    if constexpr (MISS) {
        for (uint32_t i = (uint32_t)tid; i < B; i += RS_BLOCK) rlds[i] = 0ull;
        __syncthreads();
    }
    uint32_t C = 0, Sx = 0, seq = 0, nev = 0;
    uint32_t pcur = 0; // index into p.pred of the first predicted position at or behind the cursor
    const rs_cu32* const pred4 = (const rs_cu32*)p.pred; // (constant address space: uniform indices make scalar loads)
    uint32_t kind = RS_ADVANCE, ncons = 0;
    bool gram_sent = false; // the Gram terms of the event the next message brings have been sent (on its announcement)
    bool last = M == 0;
    double dbeta = 0.0;
    // Wave v streams the positions p = v (mod 8); its k-th one, v + 8 k, lives in register set k mod RS_PF from the moment it
    // leaves HBM until the message that admits it to the window arrives: the wave's next RS_PF columns are in registers or on
    // their way BEFORE they are asked for, nothing is fetched twice, and the loads that refill a register set leave right
    // after the set has been consumed -- spread over the refill's arithmetic instead of in one burst the memory queue would
    // make the wave wait for.  Lane r keeps what belongs to set r: (mave, mstd) of its column and the marker id of its NEXT one
    // (one vector load a round ahead: a scalar load per column would put a dependent round trip in front of every column load).
    uint32_t nk = 0; // columns of this wave's residue class streamed so far: set r holds k_r = nk + ((r - nk) mod RS_PF)
    auto pos_of = [&](uint32_t k) { return (uint32_t)wave + 8u * k; };
    // everything loaded for the sets has landed
    auto cols_landed = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
    // set r takes the column of the wave's k-th position; its id is in lane r of v220 (past the end: any column; the set is never used)
    auto load_set = [&](auto rtag, uint32_t k, int32_t ids) __attribute__((always_inline)) {
        constexpr int r = decltype(rtag)::value;
        const uint32_t pn = __builtin_amdgcn_readfirstlane(pos_of(k));
        const int32_t mk = __builtin_amdgcn_readlane(ids, r);
        // (the stride is a multiple of 1 KiB and a shard's BED below 4 TiB -- resident_plan -- : one 32-bit multiply and a shift instead of a
        // 64-bit multiply's four; scalar instructions cost the wave its issue slot like any other)
        rs_set_load<T, r>(voff, p.bed + ((size_t)((uint32_t)(pn < M ? mk : 0) * (uint32_t)(p.stride >> 10)) << 10));
    };
    // lane r < RS_PF: (mave, mstd) of the column of position p1 and the id of position p2
    auto lane_load = [&](uint32_t p1, uint32_t p2) { rs_lane_load(p.s_mave + (p1 < M ? p1 : 0u), p.s_mstd + (p1 < M ? p1 : 0u), p.s_bold + (p1 < M ? p1 : 0u), p.order + (p2 < M ? p2 : 0u), p.s_ga + (p1 < M ? p1 : 0u)); };
    {
        // sets 0 .. RS_PF - 1 for k = 0 .. RS_PF - 1: the ids first (ordinary load), then the columns, then the lanes' values
        const uint32_t mp = pos_of((uint32_t)lane);
        int32_t id0 = 0;
        if (lane < RS_PF) id0 = p.order[mp < M ? mp : 0u];
        [&]<int... R>(std::integer_sequence<int, R...>) { (load_set(std::integral_constant<int, R>{}, (uint32_t)R, id0), ...); }(std::make_integer_sequence<int, RS_PF>{});
        if (lane < RS_PF) lane_load(mp, pos_of((uint32_t)lane + RS_PF));
    }
    // wait for the walker's message number seq (one lane polls; everybody else sleeps at the barrier); behind an announcement: for the
    // message that overwrites it
    auto take_message = [&](bool announced) {
        if (wg == 0 && tid == 0) p.progress[1] = ((unsigned long long)seq << 8) | 3u;
        if (tid == RS_BLOCK - WAVE) { // (the last wave: it never has atomics of the refill to drain)
            const ResMsg* m = p.msg + (seq % RS_MSG);
            const unsigned long long t0 = wall_clock64();
            uint32_t npoll = 0;
            u4_t v;
            for (;;) {
                v = rs_load16(m);
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
        const bool upd = kind == RS_EVENT || kind == RS_PIVOT || kind == RS_ANNOUNCED; // RS_PIVOT: the walker has this event's Gram terms already
        const bool with_gram = kind == RS_EVENT || ann || (kind == RS_ANNOUNCED && !gram_sent); // (announced while this workgroup was busy: the announcement was overwritten unseen)
        // (progress words: a store in front of a barrier or a vmcnt(0) wait makes it wait for the store's round trip -- workgroup 0 would be the
        // slowest of all; the timed build keeps only the stores that stand in front of a poll)
        if (DBG && wg == 0 && tid == 0) p.progress[1] = ((unsigned long long)seq << 8) | 1u;
        const uint32_t q = C + ncons - 1u;
        const uint32_t Cn = ann ? C : C + ncons;
        const uint32_t Sn = rs_window_end(Cn, B, M, p.wend_mask);
        const uint32_t nnew = Sn - Sx;
        uint32_t count_w = (Sn > (uint32_t)wave ? (Sn - (uint32_t)wave + 7u) / 8u : 0u) - nk; // this wave's positions in [Sx, Sn)
        cols_landed(); // issued at the end of the last round: nothing to wait for (and no Gram atomic in flight yet; behind an announcement they went out a draw ago)

        if (upd || ann) {
            const uint32_t slotq = q & bmask;
            uint32_t xq[T];
#pragma unroll
            for (int t = 0; t < T; ++t) xq[t] = ring[slotq * 64u * T + (uint32_t)lane * T + t];
            const double2 mq = meta[slotq];
            if (upd && tid < 16) { // the update's pair table (read behind the Gram terms, below)
                const double av = mq.x, sd = mq.y, db = dbeta;
                const double v0 = -(av * sd * db), v1 = db * (1.0 - av) * sd, v2 = db * (2.0 - av) * sd;
                // (window codes: the x form 00 / 01 / 11 = genotype 0 / 1 / 2; 10 = missing call, addend 0)
                auto addend = [&](uint32_t c) { return 0.0 + ((c == 0u) ? v0 : ((c == 1u) ? v1 : ((c == 3u) ? v2 : 0.0))); };
                tab[tid] = make_double2(addend((uint32_t)tid & 3u), addend(((uint32_t)tid >> 2) & 3u));
            }

            // ---- FIRST what the walker waits for (they need the window's codes only, not eps): the integer Gram terms A_jq = sum_i g_ij g_iq of the window columns behind q (their dots were taken before this
            // update): the walker corrects them, x_j'eps_new = x_j'eps_old + dbeta mstd_j mstd_q (A_jq - N mave_j mave_q) ----
            if (with_gram) rs_gram_terms<T, MISS>(p, ring, meta, wg, wave, lane, q, Sx, bmask, nev, xq, mq);
            if (with_gram) ++nev;
            lap(2);
            if (timing) p.trace[6 * RS_TRACE + seq % RS_TRACE] = wall_clock64();
            // ---- a8 (src/BayesRRm.cpp:1976-2010,2022,2471): eps += {v0, v1, v2, 0}[code] on the registers of every wave, while the
            // Gram atomics are on their way (an LDS-only barrier: __syncthreads() would wait for them to be performed) ----
            rs_lds_barrier(); // (an announcement: everybody has read the message before the polling lane writes the next one)
            if (upd) {
#pragma unroll
                for (int t = 0; t < T; ++t) apply_update16_lds(xq[t], tab, e[t]);
                eps_to_lds(); // (build MISS: read by the refill below, behind a barrier)
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

        // ---- a4 (src/BayesRRm.cpp:1766-1809) of the columns that refill the window, against eps as it is now ----
        if constexpr (MISS) {
            if (upd) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); // every wave's share of the updated LDS copy is in place
        }
        if ((p.tune & 3) && wave >= 4) { // (wave-uniform; the priority is an immediate)
            if ((p.tune & 3) == 1) __builtin_amdgcn_s_setprio(1);
            else if ((p.tune & 3) == 2) __builtin_amdgcn_s_setprio(2);
            else __builtin_amdgcn_s_setprio(3);
        }
        const uint32_t round_k0 = nk; // this wave's columns of the round: k in [round_k0, nk) behind the passes
        uint32_t done_k0 = nk, done_m = 0; // the sets the last pass consumed: reloaded behind the raw dots
        while (count_w) {
            if (done_m) cols_landed(); // a second pass in one round (more than 8 RS_PF new columns): its columns are asked for here and now
            const uint32_t m = count_w < (uint32_t)RS_PF ? count_w : (uint32_t)RS_PF; // register sets in use this pass: k in [nk, nk + m)
            const int32_t ids = rs_lane_ids(); // lane r: marker id of set r's next column
            const int32_t gal = MISS ? rs_lane_ga() : 0; // lane r: (group | flags) of set r's column
            // one register set = one column: no arithmetic is spent on a set that is not this pass's (the sets in use are a circular
            // run of m of the 16: taken four at a time, a quarter of the work was for columns not asked for -- and waves with one quad
            // more than the others kept the workgroup's barrier waiting)
            auto one = [&](auto rtag) __attribute__((always_inline)) {
                constexpr int r = decltype(rtag)::value;
                const uint32_t kr = nk + (((uint32_t)r - nk) & (uint32_t)(RS_PF - 1));
                if (kr - nk < m) { // wave-uniform
                    double a[4];
                    uint32_t gw[T];
                    rs_set_read<T, r>(gw, keep);
                    const uint32_t pos = __builtin_amdgcn_readfirstlane(pos_of(kr));
                    const uint32_t slot = pos & bmask;
                    uint32_t* rp = ring + slot * 64u * T + (uint32_t)lane * T;
                    // the window keeps the x form (00, 01, 11 for genotype 0, 1, 2; a missing call is the free code 10): what the update's table
                    // and the Gram terms of EVERY later event need, made once
                    if constexpr (!MISS) {
#pragma unroll
                        for (int t = 0; t < T; ++t) rp[t] = gram_xform(gw[t]);
                    }
                    if constexpr (MISS) {
                        // a column with missing calls (:1785-1790).  The device code of a missing call is 11: its field would weigh 3.  Instead of
                        // clearing the fields (five instructions a dword, on the VALU that bounds this loop) the dot is taken as it stands,
                        // s1' = s1 + 3 R with R = sum of eps over the missing calls -- which s2 = sum of eps - R needs anyway -- and the walker
                        // takes 3 R off again, in integers (both travel as fixed point).  R in the reference's index-list form
                        // (src/BayesRRm.cpp:331-341): a gather over the ~1 % of individuals whose call is missing, from the LDS copy of eps.
                        if (__builtin_amdgcn_readlane(gal, r) & 0x20000000) { // wave-uniform
                            uint32_t mall = 0u; // the lane's missing calls of its dwords in ONE word: a dword's are at the even bits, tile t's shifted by t
#pragma unroll
                            for (int t = 0; t < T; ++t) {
                                const uint32_t hi = (gw[t] >> 1) & 0x55555555u, mm = gw[t] & hi;
                                rp[t] = (gw[t] | hi) ^ mm; // the x form 00 / 01 / 11, a missing call the free code 10
                                mall |= mm << t;
                            }
                            fma_dword<true>(gw[0], e[0], a[0], a[1], a[2], a[3]);
                            if constexpr (T == 2) fma_dword<false>(gw[T - 1], e[T - 1], a[0], a[1], a[2], a[3]);
                            if (mall) { // (per lane: about a quarter of the lanes at 1 % missing calls)
                                double rsum = 0.0;
                                do { // the wave goes round as often as its fullest lane needs; bit b = slot b >> 1 of tile b & 1: eps at [(16 t + s) * 64 + lane]
                                    const uint32_t b = (uint32_t)__builtin_ctz(mall);
                                    rsum += epsl[(((b & 1u) << 4) | (b >> 1)) * 64u + (uint32_t)lane];
                                    mall &= mall - 1u;
                                } while (mall);
                                const double MAGIC = 6755399441055744.0; // the lane's part as a fixed-point integer: x + 1.5 2^52 rounds to nearest
                                const double xr = fmin(fmax(rsum * p.fx_scale, -2.2e15), 2.2e15); // (a part beyond the range is held at it: the finishing thread refuses the sum)
                                // (an instruction of its own per lane -- the LDS serialises the few lanes that meet at the address; the compiler's form of the
                                // atomic is a scalar loop over the active lanes first, ten instructions a lane)
                                const unsigned long long fr = (unsigned long long)(__double_as_longlong(xr + MAGIC) - __double_as_longlong(MAGIC));
                                asm volatile("ds_add_u64 %0, %1" ::"v"(lds_addr(rlds + (pos - Sx))), "v"(fr) : "memory");
                            }
                        } else {
#pragma unroll
                            for (int t = 0; t < T; ++t) rp[t] = gram_xform(gw[t]);
                            fma_dword<true>(gw[0], e[0], a[0], a[1], a[2], a[3]);
                            if constexpr (T == 2) fma_dword<false>(gw[T - 1], e[T - 1], a[0], a[1], a[2], a[3]);
                        }
                        // the lane sums meet in the wave's four 16-lane rows (four DPP steps, fixed order); one thread per column adds the four
                        double v = (a[0] + a[1]) + (a[2] + a[3]);
                        v += rs_dpp_f64<0xB1>(v);  // quad_perm [1,0,3,2]
                        v += rs_dpp_f64<0x4E>(v);  // quad_perm [2,3,0,1]
                        v += rs_dpp_f64<0x141>(v); // row_half_mirror
                        v += rs_dpp_f64<0x140>(v); // row_mirror
                        if ((lane & 15) == 0) part[(pos - Sx) * 4u + ((uint32_t)lane >> 4)] = v;
                    } else {
                    fma_dword<true>(gw[0], e[0], a[0], a[1], a[2], a[3]);
                    if constexpr (T == 2) fma_dword<false>(gw[T - 1], e[T - 1], a[0], a[1], a[2], a[3]);
                    // the lane sums meet in eight-lane groups (three DPP steps, fixed order); the eight group sums go to LDS and
                    // are added by ONE thread per column behind the barrier (in order: the dot does not depend on which wave took it)
                    double v = (a[0] + a[1]) + (a[2] + a[3]);
                    v += rs_dpp_f64<0xB1>(v);  // quad_perm [1,0,3,2]
                    v += rs_dpp_f64<0x4E>(v);  // quad_perm [2,3,0,1]
                    v += rs_dpp_f64<0x141>(v); // row_half_mirror
                    if ((lane & 7) == 0) part[(pos - Sx) * 8u + ((uint32_t)lane >> 3)] = v;
                    }
                    // the consumed set takes its next column now: the load leaves HBM while the other sets are at work (no wait is
                    // triggered by it: it is waited for by hand, at the top of the next round)
                    if (!last) load_set(rtag, kr + (uint32_t)RS_PF, ids);
                }
            };
            [&]<int... R>(std::integer_sequence<int, R...>) { (one(std::integral_constant<int, R>{}), ...); }(std::make_integer_sequence<int, RS_PF>{});
            // (mave, mstd) of the columns just taken go to their window slots; the lanes take their next column's and the id of the one after
            {
                const uint32_t kl = nk + (((uint32_t)lane - nk) & (uint32_t)(RS_PF - 1));
                if (lane < RS_PF && kl - nk < m) {
                    meta[pos_of(kl) & bmask] = rs_lane_meta();
                    lane_load(pos_of(kl + (uint32_t)RS_PF), pos_of(kl + 2u * (uint32_t)RS_PF));
                }
            }
            done_k0 = nk;
            done_m = m;
            nk += m;
            count_w -= m;
        }
        lap(3);
        if (p.tune & 3) __builtin_amdgcn_s_setprio(0);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); // the window's new columns and this round's group sums are in LDS for every wave
        lap(4);
        if (DBG && wg == 0 && tid == 0) p.progress[1] = ((unsigned long long)seq << 8) | 2u;
        // ---- Gram terms with the batch's pivots.  A marker whose effect is non-zero at sweep start WILL change (a predicted event):
        // the first RS_PMAX of them in the window as it stands now, [Cn, Sn), are this batch's pivots -- read off the sorted list of
        // predicted positions, p.pred, by scalar loads: every wave (and the walker) finds the same ones without a word exchanged --
        // and every column of the batch behind one of them takes its integer Gram term with it here, where both columns are in LDS:
        // the walker then needs no round trip when that pivot's turn comes (message RS_PIVOT).  Each wave takes the terms of the
        // columns it streamed: two pivots to a pass, their lane sums packed into one word per column (a lane adds at most 64 T), up
        // to sixteen columns to one reduce-scatter; lane c then holds column c's two wave sums and sends them as ONE 8-byte atomic add
        // (two 32-bit fields: the walker decodes differences).  Nothing goes through LDS and nobody waits for anybody.
        if constexpr (!MISS) {
        if (p.pivots) { // (uniform)
            // pcur: index of the first predicted position >= the cursor.  A message consumes at most one of them: its event.
            const uint32_t pe0 = pred4[pcur], pe1 = pred4[pcur + 1u], pe2 = pred4[pcur + 2u], pe3 = pred4[pcur + 3u], pe4 = pred4[pcur + 4u];
            const bool hit = upd && pe0 == q;
            if (hit) ++pcur;
            const uint32_t pv[RS_PMAX] = {hit ? pe1 : pe0, hit ? pe2 : pe1, hit ? pe3 : pe2, hit ? pe4 : pe3};
            uint32_t np = 0;
#pragma unroll
            for (int ip = 0; ip < RS_PMAX; ++ip) np += pv[ip] < Sn ? 1u : 0u; // (ascending: the ones inside the window come first)
            const uint32_t ncol = nk - round_k0; // columns this wave streamed this round
            for (uint32_t h2 = 0; h2 < np && ncol; h2 += 2u) { // wave-uniform loops
                const uint32_t pvA = pv[h2 ? 2 : 0], pvB = (h2 + 1u < np) ? pv[h2 ? 3 : 1] : 0xffffffffu;
                GramPivot gA[T], gB[T];
                {
                    const uint32_t* ra = ring + (pvA & bmask) * 64u * T + (uint32_t)lane * T;
                    const uint32_t* rb = ring + ((pvB != 0xffffffffu ? pvB : pvA) & bmask) * 64u * T + (uint32_t)lane * T;
#pragma unroll
                    for (int t = 0; t < T; ++t) {
                        gA[t] = gram_pivot_x(ra[t]);
                        gB[t] = gram_pivot_x(rb[t]);
                    }
                }
                for (uint32_t c0 = 0; c0 < ncol; c0 += 16u) {
                    uint32_t acc[16];
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        acc[k] = 0u;
                        if (c0 + (uint32_t)k < ncol) { // wave-uniform
                            const uint32_t pos = pos_of(round_k0 + c0 + (uint32_t)k);
                            if (pvA < pos) {
                                const uint32_t* rp = ring + (pos & bmask) * 64u * T + (uint32_t)lane * T;
                                uint32_t w[T], ga = 0u, gb = 0u;
#pragma unroll
                                for (int t = 0; t < T; ++t) w[t] = rp[t];
#pragma unroll
                                for (int t = 0; t < T; ++t) ga += gram16x(w[t], gA[t]);
                                if (pvB < pos) {
#pragma unroll
                                    for (int t = 0; t < T; ++t) gb += gram16x(w[t], gB[t]);
                                }
                                acc[k] = ga | (gb << 16);
                            }
                        }
                    }
                    const uint32_t tot = wave_sum16_rows(acc, lane); // every lane: the wave's sums of accumulator lane & 15
                    if ((uint32_t)lane < 16u && c0 + (uint32_t)lane < ncol && tot) {
                        const uint32_t pos = pos_of(round_k0 + c0 + (uint32_t)lane);
                        const unsigned long long w = (unsigned long long)(tot & 0xffffu) | ((unsigned long long)(tot >> 16) << 32);
                        __hip_atomic_fetch_add(p.pacc + ((size_t)(wg % p.rsh) * RS_RB + (pos % RS_RB)) * 2u + (h2 >> 1), w, HG_RLX_AGENT);
                    }
                }
            }
        }
        }
        // One thread per refilled position adds the eight group sums of its column in order and sends the workgroup's part of s1 as
        // a 51-bit fixed-point integer -- taken from "x + 1.5 2^52" (round to nearest, exact for |x| < 2^51) -- by an 8-byte atomic add
        // (contiguous over the threads: one 64-byte request per eight positions): sums over workgroups are exact and do not depend on
        // the order of arrival.  Once the adds have been performed, one add to the shard's batch counter tells the walker that this
        // workgroup's part is in.
        for (uint32_t t = (uint32_t)tid; t < nnew; t += RS_BLOCK) {
            const double MAGIC = 6755399441055744.0;
            double s1;
            if constexpr (MISS) {
                const double* pp = part + t * 4u;
                s1 = ((pp[0] + pp[1]) + pp[2]) + pp[3]; // (with weight 3 on the missing calls: the walker takes 3 R off, in integers)
                const unsigned long long fr = rlds[t];
                rlds[t] = 0ull; // (the next round's lanes add behind this round's last barrier)
                if (!(fabs((double)(long long)fr) < 2.2e15)) atomicMax(&p.state->error, 5u); // out of the fixed-point range: the sweep is refused, not wrapped
                __hip_atomic_fetch_add(p.racc2 + (size_t)(wg % p.rsh) * RS_RB + ((Sx + t) % RS_RB), fr + RS_RONE, HG_RLX_AGENT);
            } else {
                const double* pp = part + t * 8u;
                s1 = pp[0];
#pragma unroll
                for (int i = 1; i < 8; ++i) s1 += pp[i];
            }
            const double xs = s1 * p.fx_scale;
            if (!(fabs(xs) < 2.2e15)) atomicMax(&p.state->error, 5u); // out of the fixed-point range (or not finite): the sweep is refused, not wrapped
            const long long fx = __double_as_longlong(xs + MAGIC) - __double_as_longlong(MAGIC);
            __hip_atomic_fetch_add(p.racc + (size_t)(wg % p.rsh) * RS_RB + ((Sx + t) % RS_RB), (unsigned long long)fx + RS_RONE, HG_RLX_AGENT);
        }
        // Only the waves that sent something wait for their atomics to be performed (~0.6 us); the last of them to be through counts the
        // batch in for the walker.  The others go straight on to the next message -- which a wave polls that has nothing to drain -- so the
        // drain hides behind the wait for the walker instead of standing in front of it.
        {
            const uint32_t nfin = p.pivots ? (uint32_t)RS_WAVES : (nnew + 63u) / 64u < 1u ? 1u : ((nnew + 63u) / 64u > (uint32_t)RS_WAVES ? (uint32_t)RS_WAVES : (nnew + 63u) / 64u);
            if ((uint32_t)wave < nfin) {
                wait_vmcnt<0>();
                if (lane == 0) {
                    const uint32_t got = __hip_atomic_fetch_add(fin_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) + 1u;
                    if (got == nfin) {
                        __hip_atomic_store(fin_cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); // (the next round's waves count behind this round's barriers)
                        __hip_atomic_fetch_add(p.rcnt + (size_t)(wg % p.rsh) * RS_CROW, 1u, HG_RLX_AGENT);
                    }
                }
            }
        }
        lap(5);
        lap(6);
        if (timing) p.trace[7 * RS_TRACE + seq % RS_TRACE] = wall_clock64();
        C = Cn;
        Sx = Sn;
        if (last) break;
        lap(7);

        // ---- wait for the walker's next message ----
        ++seq;
        take_message(false);
        if (kind == RS_ABORT) break;
    }

    // eps goes back to HBM in the layout the other kernels read (padding slots stay zero)
    if (wave == 0 && vgrp) {
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const uint32_t i0 = (d0 + (uint32_t)t) * 16u;
#pragma unroll
            for (int s = 0; s < IPT; ++s) p.eps[eps_pos(i0 + (uint32_t)s)] = (i0 + (uint32_t)s < p.n_local) ? e[t][s] : 0.0;
        }
    }
    if (timing)
        for (int i = 0; i < 8; ++i) p.state->t[8 + i] = tacc[i];
}

// ---------------------------------------------------------------------------------------------------------------
// walker workgroup
// ---------------------------------------------------------------------------------------------------------------
struct WalkShared {
    uint32_t* mt;
    double *zig_nx, *zig_ny, *htab;
    double *mave, *mstd, *bold, *dp, *dpr, *thr0, *num; // window slots (dp: Gram corrections so far, dpr: the dot as streamed)
    int32_t *marker, *grp;
    uint32_t* batch; // refill batch (= message number) that streamed the slot's column
    uint32_t* gpart; // [RS_NSH][RS_BMAX] the shards' Gram sums of the event being collected
    unsigned long long* rprev; // [RS_RB] sum over the shards of the raw-dot words as last seen (they only ever grow)
    double* tq;     // [MT_BUF] per word of the staged generator blocks: the largest log sum_l>0 exp(logL_l - logL_0) (or, without the table
                    // below, the largest max_l (logL_l - logL_0)) that cannot give an event
    double* ftab;   // [RS_FG][RS_FN + 1] per group: f(n2) = log sum_l>0 exp(c_l + n2 r_l) at n2 = k / fscale (convex: the chord is an upper bound)
    double* fscale; // [RS_FG] grid intervals per unit of n2
    double* qtab;   // [2][HT_LDS]: per (group, component) c = logpi - hlog - logpi_0 and r = 1 / (2 sigmaE denom)
    uint16_t* crank; // [RS_BMAX] rank of the window position among the markers that take a uniform (adaV), this walk
    uint32_t* bl_pos; // [RS_NB][RS_PMAX] the pivots of a refill batch (positions, in order), by batch number mod RS_NB
    uint8_t* bl_np;   // [RS_NB] how many
    uint32_t* w_pt;   // [B][RS_PMAX] window slot: its column's Gram terms with its batch's pivots (those in front of it)
    unsigned long long* pprev; // [2][RS_RB] sum over the shards of the two pivot-term words as last seen
    uint32_t* pf_pos; // [RS_PFIRE] pivots that fired while columns streamed before their update were still without their dot:
    uint32_t* pf_msg; //            position, message number, then (dbeta, mave, mstd)
    double* pf_val;   // [RS_PFIRE][3]
    double* ebn;      // [B] window slot: the new effect of an event decided there (results are written behind the message)
    uint8_t* ekq;     // [B] its component
    uint8_t* ada;
    uint8_t* fdone; // the slot's raw dot has arrived
    double *gsum, *nmis;       // [B] build MISS: the column's sum of genotypes n1 + 2 n2 and its number of missing calls (all ranks)
    unsigned long long* rprev2; // [RS_RB] build MISS: rprev for the sums over the missing calls
    unsigned long long* gpart64; // [RS_NSH][RS_BMAX] build MISS: gpart for the 8-byte words
    unsigned long long* rloc2; // [B] several ranks, build MISS: the same for R
    unsigned long long* rloc; // [B] several ranks: this rank's part of the slot's raw dot (pushed to the peers; the dot needs theirs)
    uint8_t* fpush;           // [B] ... has been taken and pushed
    unsigned char* end;
    double* fd;     // 64 doubles of scratch (event: dbeta, bnew, prob, ...)
    uint32_t* fl;   // 64 words of flags
    int32_t* lcass; // [256]
};
enum { WF_FOUND = 0, WF_Q = 1, WF_K = 2, WF_RPOS = 3, WF_ERR = 4, WF_ABORT = 5, WF_FMIN = 6, WF_RDONE = 7, WF_CAND = 8, WF_NADA = 9, WF_COVER = 10, WF_POSTED = 11 };
enum { WD_DBETA = 0, WD_BNEW = 1, WD_PROB = 2 };

__host__ __device__ inline WalkShared walk_carve(unsigned char* q, uint32_t B)
{
    WalkShared s;
    s.mt = reinterpret_cast<uint32_t*>(q); q += MT_BUF * 4;
    s.zig_nx = reinterpret_cast<double*>(q); q += 130 * 8;
    s.zig_ny = reinterpret_cast<double*>(q); q += 130 * 8;
    s.htab = reinterpret_cast<double*>(q); q += (size_t)4 * HT_LDS * 8;
    s.mave = reinterpret_cast<double*>(q); q += (size_t)B * 8;
    s.mstd = reinterpret_cast<double*>(q); q += (size_t)B * 8;
    s.bold = reinterpret_cast<double*>(q); q += (size_t)B * 8;
    s.dp = reinterpret_cast<double*>(q); q += (size_t)B * 8;
    s.dpr = reinterpret_cast<double*>(q); q += (size_t)B * 8;
    s.thr0 = reinterpret_cast<double*>(q); q += (size_t)B * 8;
    s.num = reinterpret_cast<double*>(q); q += (size_t)B * 8;
    s.rprev = reinterpret_cast<unsigned long long*>(q); q += (size_t)RS_RB * 8;
    s.pf_val = reinterpret_cast<double*>(q); q += (size_t)RS_PFIRE * 3 * 8;
    s.ebn = reinterpret_cast<double*>(q); q += (size_t)B * 8;
    s.rloc = reinterpret_cast<unsigned long long*>(q); q += (size_t)B * 8;
    s.rloc2 = reinterpret_cast<unsigned long long*>(q); q += (size_t)B * 8;
    s.gsum = reinterpret_cast<double*>(q); q += (size_t)B * 8;
    s.nmis = reinterpret_cast<double*>(q); q += (size_t)B * 8;
    s.rprev2 = reinterpret_cast<unsigned long long*>(q); q += (size_t)RS_RB * 8;
    s.gpart64 = reinterpret_cast<unsigned long long*>(q); q += (size_t)RS_NSH * RS_BMAX * 8;
    s.tq = reinterpret_cast<double*>(q); q += (size_t)MT_BUF * 8;
    s.ftab = reinterpret_cast<double*>(q); q += (size_t)RS_FG * (RS_FN + 1) * 8;
    s.fscale = reinterpret_cast<double*>(q); q += (size_t)RS_FG * 8;
    s.qtab = reinterpret_cast<double*>(q); q += (size_t)2 * HT_LDS * 8;
    s.fd = reinterpret_cast<double*>(q); q += 64 * 8;
    s.marker = reinterpret_cast<int32_t*>(q); q += (size_t)B * 4;
    s.grp = reinterpret_cast<int32_t*>(q); q += (size_t)B * 4;
    s.batch = reinterpret_cast<uint32_t*>(q); q += (size_t)B * 4;
    s.gpart = reinterpret_cast<uint32_t*>(q); q += (size_t)RS_NSH * RS_BMAX * 4;
    s.fl = reinterpret_cast<uint32_t*>(q); q += 64 * 4;
    s.lcass = reinterpret_cast<int32_t*>(q); q += 256 * 4;
    s.bl_pos = reinterpret_cast<uint32_t*>(q); q += (size_t)RS_NB * RS_PMAX * 4;
    s.w_pt = reinterpret_cast<uint32_t*>(q); q += (size_t)B * RS_PMAX * 4;
    s.pprev = reinterpret_cast<unsigned long long*>(q); q += (size_t)2 * RS_RB * 8;
    s.pf_pos = reinterpret_cast<uint32_t*>(q); q += (size_t)RS_PFIRE * 4;
    s.pf_msg = reinterpret_cast<uint32_t*>(q); q += (size_t)RS_PFIRE * 4;
    s.crank = reinterpret_cast<uint16_t*>(q); q += (size_t)RS_BMAX * 2;
    s.bl_np = q; q += RS_NB;
    s.ekq = q; q += B;
    s.ada = q; q += B;
    s.fdone = q; q += B;
    s.fpush = q; q += B;
    s.end = q;
    return s;
}
// bytes of LDS the walker needs: the end of its own carve-up (so that the two cannot disagree)
__host__ __device__ inline size_t rs_walker_lds(uint32_t B) { return (size_t)(walk_carve(nullptr, B).end - (unsigned char*)nullptr) + 64; }

// (a function of its own, not inlined: the kernel's body is then the streaming workgroup alone -- what the register budget of the
// landing registers is about, and what tools/asm_check_loads.py checks; the walker may use every register)
template <int DBG, int MISS>
__device__ __attribute__((noinline)) void res_walker(const ResParams& p, unsigned char* smem)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t B = p.B, bmask = B - 1u, M = p.M;
    const int K = p.K;
    const bool lds_tab = p.GK <= HT_LDS;
    const WalkShared sh = walk_carve(smem, B);
    const unsigned long long clk0 = __builtin_amdgcn_s_memtime(), wall0 = wall_clock64();
    unsigned long long* const tacc = reinterpret_cast<unsigned long long*>(sh.fd + 32); // [8] stage clocks of the debug build (in LDS: sixteen registers less)
    unsigned long long tmark = DBG ? wall_clock64() : 0ull;
    if (DBG && tid == 0)
        for (int i = 0; i < 8; ++i) tacc[i] = 0ull;
    auto lap = [&](int i) {
        if (DBG && tid == 0) {
            const unsigned long long now = wall_clock64();
            tacc[i] += now - tmark;
            tmark = now;
        }
    };

    // generator, tables, counters
    for (int i = tid; i < MT_N; i += RS_BLOCK) sh.mt[i] = p.mt[i];
    for (int i = tid; i < 129; i += RS_BLOCK) {
        sh.zig_nx[i] = p.zig.nx[i];
        sh.zig_ny[i] = p.zig.ny[i];
    }
    if (lds_tab)
        for (int i = tid; i < 4 * p.GK; i += RS_BLOCK) sh.htab[(i / p.GK) * HT_LDS + (i % p.GK)] = p.denom[i];
    for (int i = tid; i < 256; i += RS_BLOCK) sh.lcass[i] = 0;
    // tq[i]: with prob = the uniform of generator word i, no event iff prob <= 1 / sum_l exp(d_l) (d_l = logL_l - logL_0, d_0 = 0,
    // :1883-1921), which holds whenever (K - 1) exp(max_l d_l) <= 1 / prob - 1: max_l d_l <= log((1 / prob - 1) / (K - 1)).  The
    // margin covers the roundings of the quick form of d_l below; a marker beyond it is decided by the exact arithmetic.
    // Tighter, where the mixture tables are in LDS and the groups are few: f(n2) = log sum_l>0 exp(c_l + n2 r_l) is convex in n2 =
    // num^2, so between two grid points its chord lies above it -- no event iff f(n2) <= log(1 / prob - 1), tested against the
    // chord (an upper bound; the bound with the factor K - 1 sent a quarter of the exact phases after markers that were no events).
    // The grid covers f up to 40 + c (beyond every possible threshold: prob >= 2^-32); n2 beyond it is a candidate.
    const bool use_ftab = lds_tab && p.GK / K <= RS_FG;
    auto stage_tq = [&](int lo, int hi) {
        const double div = use_ftab ? 1.0 : (double)(K - 1);
        for (int i = lo + tid; i < hi; i += RS_BLOCK) {
            const double prob = (double)mt_temper(sh.mt[i]) * (1.0 / 4294967296.0);
            sh.tq[i] = log((1.0 / prob - 1.0) / div) - 1e-9;
        }
    };
    if (lds_tab)
        for (int i = tid; i < p.GK; i += RS_BLOCK) {
            const int g0 = (i / K) * K;
            sh.qtab[i] = (p.logpi[i] - p.hlog[i]) - p.logpi[g0];
            sh.qtab[HT_LDS + i] = (i % K) ? p.i_2sigE / p.denom[i] : 0.0;
        }
    for (int i = tid; i < RS_RB; i += RS_BLOCK) {
        sh.rprev[i] = 0ull;
        sh.rprev2[i] = 0ull;
    }
    for (int i = tid; i < 2 * RS_RB; i += RS_BLOCK) sh.pprev[i] = 0ull;
    if (tid < 64) sh.fl[tid] = 0u;
    auto tabv = [&](int which, int t) -> double { // 0 denom, 1 logpi, 2 hlog, 3 sdk
        if (lds_tab) return sh.htab[which * HT_LDS + t];
        return which == 0 ? p.denom[t] : (which == 1 ? p.logpi[t] : (which == 2 ? p.hlog[t] : p.sdk[t]));
    };

    // window slots of the positions [lo, hi): per-marker metadata, dot accumulator reset
    auto prefetch = [&](uint32_t lo, uint32_t hi, uint32_t batch) {
        for (uint32_t j = lo + (uint32_t)tid; j < hi; j += RS_BLOCK) {
            const uint32_t slot = j & bmask;
            const int ga = p.s_ga[j];
            sh.marker[slot] = p.order[j];
            sh.grp[slot] = ga & 0x0fffffff;
            sh.ada[slot] = (uint8_t)((ga & 0x40000000) ? 1 : 0);
            sh.bold[slot] = p.s_bold[j];
            sh.mave[slot] = p.s_mave[j];
            sh.mstd[slot] = p.s_mstd[j];
            if constexpr (MISS) {
                const unsigned long long* cn = p.counts + 3ull * (unsigned long long)p.order[j];
                sh.gsum[slot] = (double)(cn[0] + 2ull * cn[1]);
                sh.nmis[slot] = (double)cn[2];
            }
            sh.dp[slot] = 0.0;
            sh.fdone[slot] = 0;
            sh.fpush[slot] = 0;
            sh.batch[slot] = batch;
        }
    };

    // arrivals per shard: shards below W mod n hold one workgroup more ([0]) than the others ([1])
    const uint32_t cntG[2] = {p.W / p.nsh + (p.W % p.nsh ? 1u : 0u), p.W / p.nsh};
    const uint32_t cntR[2] = {p.W / p.rsh + (p.W % p.rsh ? 1u : 0u), p.W / p.rsh};

    uint32_t C = 0, Sx = (B < M) ? B : M, F = 0, rpos = p.rng_idx, seq = 0, nev = 0;
    u4_t gprev0 = rs_u4(0u, 0u, 0u, 0u), gprev1 = rs_u4(0u, 0u, 0u, 0u); // this lane's four words of its shard's row as last seen, per parity
    bool has_next = false, aborted = false;
    bool pendG = false;
    uint32_t gq = 0, gV = 0;          // the event whose Gram terms are still to be collected: position, window columns behind it
    double g_db = 0.0, g_mave = 0.0, g_mstd = 0.0, g_gsum = 0.0, g_nmis = 0.0;
    unsigned long long gprev64[2][4] = {{0ull, 0ull, 0ull, 0ull}, {0ull, 0ull, 0ull, 0ull}}; // build MISS: the lane's four 8-byte words as last seen, per parity
    unsigned long long n_rounds = 0, n_events = 0, n_adv = 0, n_nnz = 0, n_chunks = 0, n_refold = 0, n_pivots = 0, n_pred = 0;
    // the pivots of refill batch `batch`: the first RS_PMAX positions of the window [lo, hi) -- as it stands when the batch is streamed --
    // whose marker has a non-zero effect at sweep start (the streaming workgroups find the same list: res_streamer)
    auto batch_pivots = [&](uint32_t batch, uint32_t lo, uint32_t hi) {
        if (tid < WAVE) {
            uint32_t np = 0;
            for (uint32_t b0 = lo; p.pivots && b0 < hi && np < (uint32_t)RS_PMAX; b0 += WAVE) {
                const uint32_t pp = b0 + (uint32_t)lane;
                unsigned long long m = __ballot(pp < hi && sh.bold[pp & bmask] != 0.0);
                while (m && np < (uint32_t)RS_PMAX) {
                    const uint32_t b = (uint32_t)(__ffsll((long long)m) - 1);
                    if (lane == 0) sh.bl_pos[(batch % RS_NB) * RS_PMAX + np] = b0 + b;
                    ++np;
                    m &= m - 1ull;
                }
            }
            if (lane == 0) sh.bl_np[batch % RS_NB] = (uint8_t)np;
        }
    };
    uint32_t pf_n = 0; // fired pivots on record (sh.pf_*)
    prefetch(0u, Sx, 0u);
    __syncthreads();
    if (use_ftab) {
        const int G = p.GK / K;
        if (tid < G) {
            double rmin = 1e300, cmin = 1e300;
            for (int l = 1; l < K; ++l) {
                rmin = fmin(rmin, sh.qtab[HT_LDS + tid * K + l]);
                cmin = fmin(cmin, sh.qtab[tid * K + l]);
            }
            // (d_l = c_l + n2 r_l >= c_l: with every c_l >= -699 no exponential of the exact arithmetic underflows to its special case, :1892;
            // scale 0: every marker of the group is a candidate)
            sh.fscale[tid] = (rmin > 0.0 && rmin < 1e300 && cmin >= -699.0) ? (double)RS_FN * rmin / 40.0 : 0.0;
        }
        __syncthreads();
        for (int i = tid; i < G * (RS_FN + 1); i += RS_BLOCK) {
            const int g = i / (RS_FN + 1), k = i % (RS_FN + 1);
            const double sc = sh.fscale[g];
            const double n2 = sc > 0.0 ? (double)k / sc : 0.0;
            double sum = 0.0;
            for (int l = 1; l < K; ++l) sum += exp(sh.qtab[g * K + l] + n2 * sh.qtab[HT_LDS + g * K + l]);
            sh.ftab[i] = log(sum);
        }
    }
    stage_tq(0, MT_N);
    batch_pivots(0u, 0u, Sx);
    __syncthreads();

    // Refill batches all workgroups have completed (their raw dots are summed in racc): wave 0 reads the shards' counters.
    auto refresh_batches = [&]() {
        if (tid < WAVE) {
            uint32_t b = 0xffffffffu;
            if ((uint32_t)lane < p.rsh) b = __hip_atomic_load(p.rcnt + (size_t)lane * RS_CROW, HG_RLX_AGENT) / cntR[0 + ((uint32_t)lane < p.W % p.rsh ? 0 : 1)];
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                const uint32_t o = (uint32_t)__shfl_xor((int)b, off, 64);
                b = o < b ? o : b;
            }
            if (lane == 0) sh.fl[WF_RDONE] = b;
        }
    };
    // The raw dot of position j, once its batch is complete: sum of the shards' words (wrapping 64-bit sums of the workgroups'
    // fixed-point parts) minus what they held before this batch (position j - RS_RB used them last; no store ever touches
    // them), s1 -> x_j'eps (:1785-1790,1809 with s2 = sum of eps).  Else j is a candidate for the first position still missing.
    auto try_raw = [&](uint32_t j, uint32_t done) {
        const uint32_t slot = j & bmask;
        if (sh.fdone[slot]) return;
        if (sh.batch[slot] >= done) {
            atomicMin(&sh.fl[WF_FMIN], j);
            return;
        }
        unsigned long long tot, totR = 0ull, now1 = 0ull;
        if (p.nranks > 1) { // this rank's part was taken by push_raw (same thread, same pass); the peers' parts: two self-validating words each
            if (!sh.fpush[slot]) { // (this rank's own part was short of an arrival: next pass)
                atomicMin(&sh.fl[WF_FMIN], j);
                return;
            }
            const unsigned long long tag = rx_rtag(p.sweep_id, sh.batch[slot]);
            if constexpr (MISS) { // (the peers' parts of R first: the four words of a position are checked together)
                unsigned long long o2[RX_MAXR][2];
#pragma unroll
                for (int r = 0; r < RX_MAXR; ++r) {
                    const bool on = r < p.nranks && r != p.rank;
                    const unsigned long long* w = rx_rbox(p.mbox[p.rank], on ? r : p.rank) + 4u * (j % RX_RING) + 2u;
                    o2[r][0] = on ? __hip_atomic_load(w, HG_RLX_SYSTEM) : tag;
                    o2[r][1] = on ? __hip_atomic_load(w + 1, HG_RLX_SYSTEM) : tag;
                }
                bool all2 = true;
                totR = sh.rloc2[slot];
#pragma unroll
                for (int r = 0; r < RX_MAXR; ++r) {
                    all2 = all2 && (o2[r][0] >> 32) == (tag >> 32) && (o2[r][1] >> 32) == (tag >> 32);
                    totR += (o2[r][1] << 32) | (o2[r][0] & 0xffffffffull);
                }
                if (!all2) {
                    atomicMin(&sh.fl[WF_FMIN], j);
                    return;
                }
            }
            unsigned long long o[RX_MAXR][2];
#pragma unroll
            for (int r = 0; r < RX_MAXR; ++r) {
                const bool on = r < p.nranks && r != p.rank;
                const unsigned long long* w = rx_rbox(p.mbox[p.rank], on ? r : p.rank) + 4u * (j % RX_RING);
                o[r][0] = on ? __hip_atomic_load(w, HG_RLX_SYSTEM) : tag;
                o[r][1] = on ? __hip_atomic_load(w + 1, HG_RLX_SYSTEM) : tag;
            }
            bool all = true;
            tot = sh.rloc[slot];
#pragma unroll
            for (int r = 0; r < RX_MAXR; ++r) {
                all = all && (o[r][0] >> 32) == (tag >> 32) && (o[r][1] >> 32) == (tag >> 32);
                tot += (o[r][1] << 32) | (o[r][0] & 0xffffffffull);
            }
            if (!all) { // a peer's part is still on its way
                atomicMin(&sh.fl[WF_FMIN], j);
                return;
            }
        } else {
            unsigned long long* base = p.racc + (j % RS_RB);
            unsigned long long w[RS_RSH];
#pragma unroll
            for (int s = 0; s < RS_RSH; ++s) w[s] = (uint32_t)s < p.rsh ? __hip_atomic_load(base + (size_t)s * RS_RB, HG_RLX_AGENT) : 0ull;
            unsigned long long now = 0ull;
#pragma unroll
            for (int s = 0; s < RS_RSH; ++s) now += w[s];
            const unsigned long long d = now - sh.rprev[j % RS_RB]; // what this position's batch added (wrapping 64-bit arithmetic), every workgroup's arrival counted
            if (rs_raw_count(d) != (p.W & 0xffu)) { // (the batch counter said "complete": an add that is not there yet is waited for, never left out)
                atomicMin(&sh.fl[WF_FMIN], j);
                return;
            }
            now1 = now; // (kept as "seen" below, in build MISS only when the second word is complete as well)
            tot = rs_raw_value(d);
        }
        double s1 = (double)(long long)tot * p.fx_unscale;
        double s2 = p.eps_sum;
        if (MISS && p.nranks > 1) {
            s1 = (double)(long long)(tot - 3ull * totR) * p.fx_unscale; // (the streamed dot weighs a missing call 3: s1' = s1 + 3 R, exact integers)
            s2 -= (double)(long long)totR * p.fx_unscale;
        } else if constexpr (MISS) { // s2 = sum of eps over the column's calls = sum of eps - R
            unsigned long long* base2 = p.racc2 + (j % RS_RB);
            unsigned long long w2[RS_RSH];
#pragma unroll
            for (int s = 0; s < RS_RSH; ++s) w2[s] = (uint32_t)s < p.rsh ? __hip_atomic_load(base2 + (size_t)s * RS_RB, HG_RLX_AGENT) : 0ull;
            unsigned long long now2 = 0ull;
#pragma unroll
            for (int s = 0; s < RS_RSH; ++s) now2 += w2[s];
            const unsigned long long d2 = now2 - sh.rprev2[j % RS_RB];
            if (rs_raw_count(d2) != (p.W & 0xffu)) {
                atomicMin(&sh.fl[WF_FMIN], j);
                return;
            }
            sh.rprev2[j % RS_RB] = now2;
            const unsigned long long tot2 = rs_raw_value(d2);
            s1 = (double)(long long)(tot - 3ull * tot2) * p.fx_unscale; // (the streamed dot weighs a missing call 3: s1' = s1 + 3 R, exact integers)
            s2 -= (double)(long long)tot2 * p.fx_unscale;
        }
        if (p.nranks <= 1) sh.rprev[j % RS_RB] = now1;
        sh.dpr[slot] = sh.mstd[slot] * (s1 - sh.mave[slot] * s2);
        // the column's Gram terms with its batch's pivots (those in front of it), and the corrections of the pivots that have fired since
        // the column was streamed
        const uint32_t bt = sh.batch[slot], np = sh.bl_np[bt % RS_NB];
        const uint32_t* bp = sh.bl_pos + (bt % RS_NB) * RS_PMAX;
        if (np && bp[0] < j) { // (a pivot in front of the column: its terms were sent; all shards' two words at once, the batch's part = the difference)
            unsigned long long w[RS_RSH][2];
#pragma unroll
            for (int sidx = 0; sidx < RS_RSH; ++sidx) {
                const unsigned long long* pw = p.pacc + ((size_t)((uint32_t)sidx < p.rsh ? sidx : 0) * RS_RB + (j % RS_RB)) * 2u;
                w[sidx][0] = (uint32_t)sidx < p.rsh ? __hip_atomic_load(pw, HG_RLX_AGENT) : 0ull;
                w[sidx][1] = (uint32_t)sidx < p.rsh ? __hip_atomic_load(pw + 1, HG_RLX_AGENT) : 0ull;
            }
            unsigned long long n01 = 0ull, n23 = 0ull;
#pragma unroll
            for (int sidx = 0; sidx < RS_RSH; ++sidx) {
                n01 += w[sidx][0];
                n23 += w[sidx][1];
            }
            const unsigned long long d01 = n01 - sh.pprev[j % RS_RB], d23 = n23 - sh.pprev[RS_RB + (j % RS_RB)];
            sh.pprev[j % RS_RB] = n01;
            sh.pprev[RS_RB + (j % RS_RB)] = n23;
            sh.w_pt[slot * RS_PMAX + 0] = (uint32_t)d01;
            sh.w_pt[slot * RS_PMAX + 1] = (uint32_t)(d01 >> 32);
            sh.w_pt[slot * RS_PMAX + 2] = (uint32_t)d23;
            sh.w_pt[slot * RS_PMAX + 3] = (uint32_t)(d23 >> 32);
        }
        for (uint32_t f = 0; f < pf_n; ++f) {
            const uint32_t fq = sh.pf_pos[f];
            if (sh.pf_msg[f] > bt && fq < j) {
                for (uint32_t ip = 0; ip < np; ++ip)
                    if (bp[ip] == fq) {
                        const double xx = sh.mstd[slot] * sh.pf_val[3 * f + 2] * ((double)sh.w_pt[slot * RS_PMAX + ip] - p.n_total * (sh.mave[slot] * sh.pf_val[3 * f + 1]));
                        sh.dp[slot] += sh.pf_val[3 * f] * xx;
                    }
            }
        }
        sh.fdone[slot] = 1;
    };
    // several ranks: this rank's part of the raw dot of position j (its batch complete here), kept and pushed to every peer
    auto push_raw = [&](uint32_t j, uint32_t done) {
        const uint32_t slot = j & bmask;
        if (sh.fpush[slot] || sh.batch[slot] >= done) return;
        unsigned long long* base = p.racc + (j % RS_RB);
        unsigned long long w[RS_RSH];
#pragma unroll
        for (int s = 0; s < RS_RSH; ++s) w[s] = (uint32_t)s < p.rsh ? __hip_atomic_load(base + (size_t)s * RS_RB, HG_RLX_AGENT) : 0ull;
        unsigned long long now = 0ull;
#pragma unroll
        for (int s = 0; s < RS_RSH; ++s) now += w[s];
        const unsigned long long d = now - sh.rprev[j % RS_RB];
        if (rs_raw_count(d) != (p.W & 0xffu)) return; // (an add that has not been performed yet: taken in a later pass -- fpush stays clear)
        const unsigned long long tot = rs_raw_value(d);
        unsigned long long tot2 = 0ull;
        if constexpr (MISS) {
            unsigned long long* base2 = p.racc2 + (j % RS_RB);
            unsigned long long w2[RS_RSH];
#pragma unroll
            for (int s = 0; s < RS_RSH; ++s) w2[s] = (uint32_t)s < p.rsh ? __hip_atomic_load(base2 + (size_t)s * RS_RB, HG_RLX_AGENT) : 0ull;
            unsigned long long now2 = 0ull;
#pragma unroll
            for (int s = 0; s < RS_RSH; ++s) now2 += w2[s];
            const unsigned long long d2 = now2 - sh.rprev2[j % RS_RB];
            if (rs_raw_count(d2) != (p.W & 0xffu)) return;
            tot2 = rs_raw_value(d2);
            sh.rprev2[j % RS_RB] = now2;
            sh.rloc2[slot] = tot2;
        }
        sh.rprev[j % RS_RB] = now;
        sh.rloc[slot] = tot;
        const unsigned long long tag = rx_rtag(p.sweep_id, sh.batch[slot]);
        for (int r = 0; r < p.nranks; ++r)
            if (r != p.rank) {
                unsigned long long* w = rx_rbox(p.mbox[r], p.rank) + 4u * (j % RX_RING);
                __hip_atomic_store(w, tag | (tot & 0xffffffffull), HG_RLX_SYSTEM);
                __hip_atomic_store(w + 1, tag | (tot >> 32), HG_RLX_SYSTEM);
                if constexpr (MISS) {
                    __hip_atomic_store(w + 2, tag | (tot2 & 0xffffffffull), HG_RLX_SYSTEM);
                    __hip_atomic_store(w + 3, tag | (tot2 >> 32), HG_RLX_SYSTEM);
                }
            }
        sh.fpush[slot] = 1;
    };
    // one pass over the positions [F, Sx) that have no dot yet; afterwards F = the first one still missing
    auto fold_pass = [&]() {
        refresh_batches();
        if (tid == 0) sh.fl[WF_FMIN] = Sx;
        rs_lds_barrier();
        const uint32_t done = sh.fl[WF_RDONE];
        if (p.nranks > 1)
            for (uint32_t j = F + (uint32_t)tid; j < Sx; j += RS_BLOCK) push_raw(j, done);
        for (uint32_t j = F + (uint32_t)tid; j < Sx; j += RS_BLOCK) try_raw(j, done);
        rs_lds_barrier(); // (not __syncthreads(): the pushes to the peers need not have arrived)
        F = sh.fl[WF_FMIN];
    };

    for (;;) {
        if (C >= M) break;
        ++n_rounds;
        if (tid == 0) p.progress[0] = (n_rounds << 8) | 1u;
        // generator: the next block exists before a round can run into it
        if (!has_next && rpos + B + 96u > (uint32_t)MT_N) {
            mt_next_block(sh.mt, tid);
            stage_tq(MT_N, MT_BUF);
            has_next = true;
            __syncthreads();
        }
        // 1. Gram terms of the window columns behind the last event (the streaming workgroups' first job after a message): wave s
        // polls shard s's row, 16 bytes per lane; every word in use must carry the shard's full arrival count
        if (MISS && pendG && gV) {
            // the same with 8-byte words (four per lane, two loads): count in the top byte, below it the fixed-point sum of the
            // workgroups' four-term sums; the terms that do not depend on the individuals come from the markers' counts
            const int ws = tid >> 6;
            const uint32_t par = (nev - 1u) & 1u;
            if ((uint32_t)ws < p.nsh && 4u * (uint32_t)lane < B) {
                const unsigned long long* row = p.gacc64 + ((size_t)par * RS_NSH + (uint32_t)ws) * RS_GROW + 4u * (uint32_t)lane;
                const unsigned long long want = cntG[0 + ((uint32_t)ws < p.W % p.nsh ? 0 : 1)];
                const unsigned long long t0 = wall_clock64();
                unsigned long long v[4], d[4];
                for (;;) {
                    const u4_t a = rs_load16(row), b = rs_load16(row + 2);
                    v[0] = ((unsigned long long)a.y << 32) | a.x;
                    v[1] = ((unsigned long long)a.w << 32) | a.z;
                    v[2] = ((unsigned long long)b.y << 32) | b.x;
                    v[3] = ((unsigned long long)b.w << 32) | b.z;
                    bool ok = true;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        d[i] = v[i] - gprev64[par][i];
                        ok = ok && (((4u * (uint32_t)lane + (uint32_t)i - (gq + 1u)) & bmask) >= gV || (d[i] >> 56) == want); // (words are by window slot)
                    }
                    if (ok) break;
                    if (wall_clock64() - t0 > p.timeout) {
                        sh.fl[WF_ABORT] = 1u;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    gprev64[par][i] = v[i];
                    sh.gpart64[(uint32_t)ws * RS_BMAX + 4u * (uint32_t)lane + (uint32_t)i] = d[i] & (RS_ONE64 - 1ull);
                }
            }
            __syncthreads();
            if ((uint32_t)tid < gV) {
                unsigned long long A = 0ull;
                for (uint32_t sidx = 0; sidx < p.nsh; ++sidx) A += sh.gpart64[sidx * RS_BMAX + ((gq + 1u + (uint32_t)tid) & bmask)];
                if (p.nranks > 1) {
                    // several ranks: the 56-bit sums cross as two words each, tag << 32 | half (tag = 1 | sweep | event: self-validating)
                    const unsigned long long tag = (0x80000000ull | ((p.sweep_id & 0x7full) << 24) | (unsigned long long)(nev & 0xffffffu)) << 32;
                    for (int r = 0; r < p.nranks; ++r)
                        if (r != p.rank) {
                            unsigned long long* w = rx_gbox(p.mbox[r], p.rank, par) + 2 * tid;
                            __hip_atomic_store(w, tag | (A & 0xffffffffull), HG_RLX_SYSTEM);
                            __hip_atomic_store(w + 1, tag | (A >> 32), HG_RLX_SYSTEM);
                        }
                    const unsigned long long t0 = wall_clock64();
                    for (;;) {
                        unsigned long long o[RX_MAXR][2];
#pragma unroll
                        for (int r = 0; r < RX_MAXR; ++r) {
                            const bool on = r < p.nranks && r != p.rank;
                            const unsigned long long* w = rx_gbox(p.mbox[p.rank], on ? r : p.rank, par) + 2 * tid;
                            o[r][0] = on ? __hip_atomic_load(w, HG_RLX_SYSTEM) : tag;
                            o[r][1] = on ? __hip_atomic_load(w + 1, HG_RLX_SYSTEM) : tag;
                        }
                        bool all = true;
                        unsigned long long add = 0ull;
#pragma unroll
                        for (int r = 0; r < RX_MAXR; ++r) {
                            all = all && (o[r][0] >> 32) == (tag >> 32) && (o[r][1] >> 32) == (tag >> 32);
                            add += (o[r][1] << 32) | (o[r][0] & 0xffffffffull);
                        }
                        if (all) {
                            A += add;
                            break;
                        }
                        if (wall_clock64() - t0 > p.timeout) {
                            sh.fl[WF_ABORT] = 1u;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
                const uint32_t slot = (gq + 1u + (uint32_t)tid) & bmask;
                const double mj = sh.mave[slot], sj = sh.mstd[slot];
                const double Ad = (double)A * (1.0 / (double)(1ull << RS_GFX));
                const double both = p.n_total - sh.nmis[slot] - g_nmis; // + X: calls present in both columns
                const double xx = sj * g_mstd * (((Ad - g_mave * sh.gsum[slot]) - mj * g_gsum) + (mj * g_mave) * both);
                sh.dp[slot] += g_db * xx;
            }
        } else if (pendG && gV) {
            const int ws = tid >> 6;
            const uint32_t par = (nev - 1u) & 1u;
            if ((uint32_t)ws < p.nsh && 4u * (uint32_t)lane < B) {
                const uint32_t* row = p.gacc + ((size_t)par * RS_NSH + (uint32_t)ws) * RS_GROW + 4u * (uint32_t)lane;
                const uint32_t want = cntG[0 + ((uint32_t)ws < p.W % p.nsh ? 0 : 1)];
                const unsigned long long t0 = wall_clock64();
                // the words only ever grow (no store ever touches them: adds are performed at the memory side, and a store could
                // overtake or be overtaken by one): what this event added is the difference to what the lane saw last time
                u4_t v, d;
                for (;;) {
                    v = rs_load16(row);
                    d = par ? (v - gprev1) : (v - gprev0);
                    // (words are by window slot: one is in use iff its slot's column lies behind the event)
                    const uint32_t i = 4u * (uint32_t)lane - (gq + 1u);
                    const bool ok = ((i & bmask) >= gV || (d.x >> 24) == want) && (((i + 1u) & bmask) >= gV || (d.y >> 24) == want) &&
                                    (((i + 2u) & bmask) >= gV || (d.z >> 24) == want) && (((i + 3u) & bmask) >= gV || (d.w >> 24) == want);
                    if (ok) break;
                    if (wall_clock64() - t0 > p.timeout) {
                        sh.fl[WF_ABORT] = 1u;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                // (a word beyond gV got no add this time: its difference is zero and its previous value stays what it is)
                if (par) gprev1 = v;
                else gprev0 = v;
                uint32_t* gp = sh.gpart + (uint32_t)ws * RS_BMAX + 4u * (uint32_t)lane;
                gp[0] = d.x & RS_LOW;
                gp[1] = d.y & RS_LOW;
                gp[2] = d.z & RS_LOW;
                gp[3] = d.w & RS_LOW;
            }
            __syncthreads();
            uint32_t A = 0u;
            if ((uint32_t)tid < gV)
                for (uint32_t sidx = 0; sidx < p.nsh; ++sidx) A += sh.gpart[sidx * RS_BMAX + ((gq + 1u + (uint32_t)tid) & bmask)];
            if (p.nranks > 1) {
                // this rank's sums go to every peer's mailbox, the peers' arrive in mine (integers: the total does not depend on the
                // order).  A word says itself what it is -- sweep << 48 | event << 24 | sum, one 8-byte store -- so there is no flag
                // to order behind the data: one hop, and the reader polls the words it needs.
                const unsigned long long tag = (p.sweep_id << 48) | ((unsigned long long)(nev & 0xffffffu) << 24);
                if ((uint32_t)tid < gV) {
                    for (int r = 0; r < p.nranks; ++r)
                        if (r != p.rank) __hip_atomic_store(rx_gbox(p.mbox[r], p.rank, par) + 2 * tid, tag | (unsigned long long)A, HG_RLX_SYSTEM);
                    const unsigned long long t0 = wall_clock64();
                    for (;;) {
                        unsigned long long o[RX_MAXR];
#pragma unroll
                        for (int r = 0; r < RX_MAXR; ++r)
                            o[r] = (r < p.nranks && r != p.rank) ? __hip_atomic_load(rx_gbox(p.mbox[p.rank], r, par) + 2 * tid, HG_RLX_SYSTEM) : tag;
                        bool all = true;
                        uint32_t add = 0u;
#pragma unroll
                        for (int r = 0; r < RX_MAXR; ++r) {
                            all = all && (o[r] >> 24) == (tag >> 24);
                            add += (uint32_t)o[r] & RS_LOW;
                        }
                        if (all) {
                            A += add;
                            break;
                        }
                        if (wall_clock64() - t0 > p.timeout) {
                            sh.fl[WF_ABORT] = 1u;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
            }
            if ((uint32_t)tid < gV) {
                const uint32_t slot = (gq + 1u + (uint32_t)tid) & bmask;
                const double mj = sh.mave[slot], sj = sh.mstd[slot];
                const double xx = sj * g_mstd * ((double)A - p.n_total * (mj * g_mave));
                sh.dp[slot] += g_db * xx;
            }
        }
        pendG = false;
        if (tid == 0) sh.fl[WF_CAND] = 0xffffffffu;
        __syncthreads();
        lap(1);
        if (DBG && tid == 0) p.trace[1 * RS_TRACE + seq % RS_TRACE] = wall_clock64();
        if (sh.fl[WF_ABORT]) {
            aborted = true;
            break;
        }

        // 3. The walk.  Every position of the window that has its dot is tested in parallel, one thread each, with a bound that needs
        // no exponential: "this marker cannot be an event" (see stage_tq).  The first position that does not pass -- a marker whose
        // effect is non-zero (it WILL change), or one too close to call -- is decided by wave 0 with the exact arithmetic of the
        // reference (all thresholds of its marker, :1883-1921, the component, the draw, a7); if that says "no event" after all, the
        // next candidate is looked at.  Acum of the markers that pass is computed behind the message (step 5).
        if (DBG && tid == 0) p.progress[0] = (n_rounds << 8) | 2u;
        uint32_t base = C;
        bool found = false;
        while (!found && base < Sx) {
            if (base >= F) { // the walk needs dots that are still on their way: wait for them
                ++n_refold;
                if (tid == 0) p.progress[0] = (n_rounds << 8) | 3u;
                const unsigned long long t0 = wall_clock64();
                for (;;) {
                    __syncthreads();
                    if (tid == 0 && wall_clock64() - t0 > p.timeout) sh.fl[WF_ABORT] = 1u; // (one thread decides: the exit must be uniform)
                    fold_pass();
                    if (F > base || sh.fl[WF_ABORT]) break;
                }
                __syncthreads();
                if (sh.fl[WF_ABORT]) break;
                lap(0);
            }
            const uint32_t nA = F - base; // <= B <= 256 positions: threads 0 .. nA - 1
            ++n_chunks;
            if (DBG && tid == 0) p.progress[0] = (n_rounds << 8) | 4u;
            bool cand = false;
            uint32_t myrank = 0;
            {
                const int wv = tid >> 6;
                // rank among the markers that take a uniform: ballots of this wave's positions and of the waves before it
                uint32_t before = 0;
                unsigned long long am = 0ull;
                if (p.all_ada) { // every marker takes a uniform: the rank is the offset
                    before = (uint32_t)wv * WAVE;
                    am = ~0ull;
                }
                for (int w = 0; w <= wv && w < 4 && !p.all_ada; ++w) {
                    const uint32_t jw = (uint32_t)(w * WAVE + lane);
                    const bool a = jw < nA && sh.ada[(base + jw) & bmask] != 0;
                    const unsigned long long m = __ballot(a);
                    if (w < wv) before += (uint32_t)__popcll(m);
                    else am = m;
                }
                if ((uint32_t)tid < nA) {
                    const uint32_t slot = (base + (uint32_t)tid) & bmask;
                    const bool ada = sh.ada[slot] != 0;
                    const double bold = sh.bold[slot];
                    myrank = before + (uint32_t)__popcll(am & ((1ull << lane) - 1ull));
                    sh.crank[tid] = (uint16_t)myrank;
                    cand = bold != 0.0;
                    if (ada) {
                        const int g0 = sh.grp[slot] * K;
                        const double num = (sh.dpr[slot] + sh.dp[slot]) + bold * p.n_minus_1;
                        sh.num[slot] = num;
                        const double n2 = num * num;
                        double dmax = -1e300, dmin = 1e300;
                        for (int l = 1; l < K && !use_ftab; ++l) {
                            double c, r;
                            if (lds_tab) {
                                c = sh.qtab[g0 + l];
                                r = sh.qtab[HT_LDS + g0 + l];
                            } else {
                                c = (p.logpi[g0 + l] - p.hlog[g0 + l]) - p.logpi[g0];
                                r = p.i_2sigE / p.denom[g0 + l];
                            }
                            const double d = c + n2 * r;
                            dmax = d > dmax ? d : dmax;
                            dmin = d < dmin ? d : dmin;
                        }
                        if (use_ftab) {
                            const int g = sh.grp[slot];
                            const double x = n2 * sh.fscale[g];
                            const bool inside = x < (double)RS_FN && sh.fscale[g] > 0.0; // (NaN: outside)
                            const int k = inside ? (int)x : 0;
                            const double f0 = sh.ftab[g * (RS_FN + 1) + k], f1 = sh.ftab[g * (RS_FN + 1) + k + 1];
                            const double fup = f0 + (x - (double)k) * (f1 - f0);
                            cand = cand || !(inside && fup <= sh.tq[rpos + myrank]);
                        } else {
                            cand = cand || !(dmax <= sh.tq[rpos + myrank] && dmin >= -699.0);
                        }
                    }
                    if (cand) atomicMin(&sh.fl[WF_CAND], base + (uint32_t)tid);
                    if ((uint32_t)tid == nA - 1u) sh.fl[WF_NADA] = myrank + (ada ? 1u : 0u);
                }
            }
            __syncthreads();
            lap(2);
            // the candidates in order, until one is an event
            for (;;) {
                const uint32_t qc = sh.fl[WF_CAND];
                if (qc == 0xffffffffu) break; // uniform
                if (tid < WAVE) {
                    const uint32_t slot = qc & bmask;
                    const bool ada = sh.ada[slot] != 0;
                    const double bold = sh.bold[slot];
                    const uint32_t upos = rpos + (uint32_t)sh.crank[qc - base]; // the marker's uniform (if it takes one)
                    uint32_t pos = upos + (ada ? 1u : 0u);
                    int k = 0;
                    double bnew = 0.0;
                    uint32_t consumed = 0u, gerr = 0u;
                    if (ada) { // wave-uniform
                        const double prob = (double)mt_temper(sh.mt[upos]) * (1.0 / 4294967296.0);
                        const int g0 = sh.grp[slot] * K;
                        const double num = sh.num[slot];
                        double Lm = 0.0; // lane x < K: logL_x
                        if (lane < K) {
                            Lm = tabv(1, g0 + lane);
                            if (lane > 0) {
                                const double mk = num / tabv(0, g0 + lane);
                                Lm = tabv(1, g0 + lane) - tabv(2, g0 + lane) + mk * num * p.i_2sigE;
                            }
                        }
                        const int kk = lane >> 3, l = lane & 7;
                        const double Ll = __shfl(Lm, l, 64), Lk = __shfl(Lm, kk, 64);
                        const bool on = kk < K - 1 && l < K;
                        const double d = on ? Ll - Lk : 0.0;
                        const double ex = exp(d);
                        const bool bigp = on && l >= (kk ? kk : 1) && fabs(d) > 700.0;
                        const unsigned long long bm = __ballot(bigp);
                        // the group's sum in component order, in the group's first lane: its neighbours' terms come by DPP shifts (a
                        // crossbar shuffle per term was ~100 clocks each, on the chain)
                        double sum = ex, cur = ex;
#pragma unroll
                        for (int x = 1; x < 8; ++x) {
                            cur = rs_dpp_f64<0x101>(cur); // row_shl:1 -- lane i takes lane i + 1's
                            if (x < K) sum += cur;        // wave-uniform
                        }
                        const bool anyb = ((bm >> (lane & ~7)) & 0xffull) != 0ull;
                        const double thr = anyb ? 0.0 : 1.0 / sum; // of walk step kk, valid in the first lane of its group
                        k = K - 1;
                        double acum = 0.0;
                        bool fnd = false;
#pragma unroll
                        for (int sidx = 0; sidx < 7; ++sidx) {
                            if (sidx + 1 < K) { // wave-uniform
                                const double t = rs_readlane(thr, 8 * sidx);
                                acum = sidx ? acum + t : t;
                                if (!fnd && prob <= acum) {
                                    k = sidx;
                                    fnd = true;
                                }
                            }
                        }
                        if (lane == 0 && k > 0) {
                            LdsGen g{sh.mt, pos, has_next ? (uint32_t)MT_BUF : (uint32_t)MT_N, 0u};
                            ZigTables zt{sh.zig_nx, sh.zig_ny, p.zig.ex, p.zig.ey};
                            bnew = norm_rng_sd(g, zt, num / tabv(0, g0 + k), tabv(3, g0 + k));
                            consumed = g.pos - pos;
                            gerr = g.err;
                        }
                        bnew = rs_readlane(bnew, 0);
                        consumed = (uint32_t)__builtin_amdgcn_readlane((int)consumed, 0);
                        gerr = (uint32_t)__builtin_amdgcn_readlane((int)gerr, 0);
                    }
                    const bool is_ev = k != 0 || bold != 0.0;
                    if (lane == 0) {
                        if (is_ev) {
                            sh.fl[WF_FOUND] = 1u;
                            sh.fl[WF_Q] = qc;
                            sh.fl[WF_K] = (uint32_t)k;
                            sh.fl[WF_RPOS] = pos + consumed;
                            sh.fd[WD_BNEW] = bnew;
                            sh.fd[WD_DBETA] = bold - bnew;
                            if (gerr) sh.fl[WF_ERR] = gerr;
                            // the message leaves HERE, from the lane that drew the effect -- two barriers before the rest of the
                            // workgroup has caught up with the decision (section 4 below skips its store).  Not when the event's
                            // Gram terms may already be with the walker (pivots), nor when the flow control has to be consulted.
                            const double db_now = bold - bnew;
                            if (!p.pivots && !gerr && db_now != 0.0 && seq + 4u <= sh.fl[WF_RDONE] + (uint32_t)RS_MSG - 3u) {
                                const uint32_t nc = qc - C + 1u;
                                const uint32_t kf = (uint32_t)RS_EVENT | (C + nc >= M ? (uint32_t)RS_LAST : 0u);
                                const unsigned long long db = (unsigned long long)__double_as_longlong(db_now);
                                if (DBG) {
                                    p.trace[2 * RS_TRACE + seq % RS_TRACE] = wall_clock64();
                                    p.trace[3 * RS_TRACE + seq % RS_TRACE] = nc;
                                    p.trace[0 * RS_TRACE + (seq + 1u) % RS_TRACE] = wall_clock64();
                                }
                                rs_store16(p.msg + ((seq + 1u) % RS_MSG), rs_u4(rs_msg_word0(kf, nc, seq + 1u, (uint32_t)db, (uint32_t)(db >> 32)), seq + 1u, (uint32_t)db, (uint32_t)(db >> 32)));
                                sh.fl[WF_POSTED] = 1u;
                            }
                        } else {
                            sh.fl[WF_CAND] = 0xffffffffu; // too close to call, and no event: on to the next candidate
                        }
                    }
                }
                __syncthreads();
                if (sh.fl[WF_FOUND]) break;
                if (cand && base + (uint32_t)tid == qc) cand = false;
                if (cand) atomicMin(&sh.fl[WF_CAND], base + (uint32_t)tid);
                __syncthreads();
            }
            found = sh.fl[WF_FOUND] != 0u;
            if (found) {
                rpos = sh.fl[WF_RPOS];
            } else {
                rpos += sh.fl[WF_NADA];
                base += nA;
            }
            __syncthreads();
            if (tid == 0) sh.fl[WF_CAND] = 0xffffffffu; // (for the next pass of the walk: set behind one barrier, used behind the next)
            lap(3);
        }
        if (sh.fl[WF_ABORT] || sh.fl[WF_ERR]) {
            aborted = true;
            break;
        }

        // 4. the message
        if (DBG && tid == 0) p.progress[0] = (n_rounds << 8) | 5u;
        const uint32_t qpos = found ? sh.fl[WF_Q] : 0u;
        const uint32_t ncons = found ? qpos - C + 1u : Sx - C;
        const double dbeta = found ? sh.fd[WD_DBETA] : 0.0, bnew = found ? sh.fd[WD_BNEW] : 0.0;
        const int kq = found ? (int)sh.fl[WF_K] : 0;
        const bool is_event = found && dbeta != 0.0;
        if (is_event && sh.bold[qpos & bmask] != 0.0) ++n_pred;
        const uint32_t Cn = C + ncons;
        const bool lastmsg = Cn >= M;
        // A predicted pivot whose Gram terms came with the columns: every refill batch that has columns behind it in the window must
        // list it (wave 0 checks the batches, one per lane).  Then nobody has to be waited for: message RS_PIVOT, corrections now.
        bool pivot = false;
        if (p.pivots && is_event && sh.bold[qpos & bmask] != 0.0 && pf_n < (uint32_t)RS_PFIRE) { // uniform
            if (tid < WAVE) {
                bool ok = true;
                if (qpos + 1u < Sx) {
                    const uint32_t b_lo = sh.batch[(qpos + 1u) & bmask], nb = seq - b_lo + 1u; // batches b_lo .. seq have columns behind q
                    ok = nb <= (uint32_t)WAVE;
                    bool mine = true;
                    if ((uint32_t)lane < nb && ok) {
                        const uint32_t b = b_lo + (uint32_t)lane;
                        mine = false;
                        for (uint32_t ip = 0; ip < sh.bl_np[b % RS_NB]; ++ip) mine = mine || sh.bl_pos[(b % RS_NB) * RS_PMAX + ip] == qpos;
                    }
                    ok = ok && __ballot(!mine) == 0ull;
                }
                if (lane == 0) sh.fl[WF_COVER] = ok ? 1u : 0u;
            }
            __syncthreads();
            pivot = sh.fl[WF_COVER] != 0u;
        }
        // the walker is never more than RS_MSG - 3 messages ahead of the slowest streaming workgroup (messages without a Gram round trip
        // would otherwise let it run away): batches completed = messages taken + 1
        if (seq + 3u > sh.fl[WF_RDONE] + (uint32_t)RS_MSG - 3u) {
            const unsigned long long t0 = wall_clock64();
            for (;;) {
                __syncthreads();
                if (tid == 0 && wall_clock64() - t0 > p.timeout) sh.fl[WF_ABORT] = 1u;
                refresh_batches();
                __syncthreads();
                if (seq + 3u <= sh.fl[WF_RDONE] + (uint32_t)RS_MSG - 3u || sh.fl[WF_ABORT]) break;
            }
            if (sh.fl[WF_ABORT]) {
                aborted = true;
                break;
            }
        }
        __syncthreads();
        ++seq;
        const bool posted = tid == 0 && sh.fl[WF_POSTED] != 0u; // (thread 0 wrote it, reads it and clears it: nobody else looks)
        if (DBG && tid == 0 && !posted) {
            p.trace[2 * RS_TRACE + (seq - 1u) % RS_TRACE] = wall_clock64();
            p.trace[3 * RS_TRACE + (seq - 1u) % RS_TRACE] = ncons;
            p.trace[0 * RS_TRACE + seq % RS_TRACE] = wall_clock64();
        }
        if (tid == 0 && !posted) {
            const uint32_t kf = (pivot ? (uint32_t)RS_PIVOT : (is_event ? (uint32_t)RS_EVENT : (uint32_t)RS_ADVANCE)) | (lastmsg ? (uint32_t)RS_LAST : 0u);
            const unsigned long long db = (unsigned long long)__double_as_longlong(dbeta);
            rs_store16(p.msg + (seq % RS_MSG), rs_u4(rs_msg_word0(kf, ncons, seq, (uint32_t)db, (uint32_t)(db >> 32)), seq, (uint32_t)db, (uint32_t)(db >> 32)));
        }
        if (pivot) {
            // corrections of the window columns behind q that have their dot; the others get theirs when it arrives (try_raw)
            const uint32_t qs = qpos & bmask;
            const double mq = sh.mave[qs], sq = sh.mstd[qs];
            for (uint32_t j = qpos + 1u + (uint32_t)tid; j < F; j += RS_BLOCK) {
                const uint32_t slot = j & bmask, bt = sh.batch[slot];
                const uint32_t* bp = sh.bl_pos + (bt % RS_NB) * RS_PMAX;
                for (uint32_t ip = 0; ip < sh.bl_np[bt % RS_NB]; ++ip)
                    if (bp[ip] == qpos) {
                        const double xx = sh.mstd[slot] * sq * ((double)sh.w_pt[slot * RS_PMAX + ip] - p.n_total * (sh.mave[slot] * mq));
                        sh.dp[slot] += dbeta * xx;
                    }
            }
            if (F < Sx) { // columns behind q whose dot (streamed before this update) is still on its way
                if (tid == 0) {
                    sh.pf_pos[pf_n] = qpos;
                    sh.pf_msg[pf_n] = seq;
                    sh.pf_val[3 * pf_n] = dbeta;
                    sh.pf_val[3 * pf_n + 1] = mq;
                    sh.pf_val[3 * pf_n + 2] = sq;
                }
                ++pf_n;
            }
            ++n_pivots;
            ++n_events;
            ++n_nnz;
        } else if (is_event) {
            const uint32_t slot = qpos & bmask;
            pendG = true;
            gq = qpos;
            gV = Sx - (qpos + 1u);
            g_db = dbeta;
            g_mave = sh.mave[slot];
            g_mstd = sh.mstd[slot];
            if constexpr (MISS) {
                g_gsum = sh.gsum[slot];
                g_nmis = sh.nmis[slot];
            }
            ++nev;
            ++n_events;
            ++n_nnz;
        } else {
            ++n_adv;
        }

        // 5. results of the consumed markers: the numerator of every marker that took part goes out as it stands (Acum's slot holds it
        // until the sweep is over: k_res_finish turns it into Acum = 1 / sum_l exp(logL_l - logL_0), :1892,:1899-1905, for all markers at
        // once -- K exponentials per marker are not the walker's business while the chain waits for it); an event also writes its new
        // effect and its component (flagged: k_res_finish clears the flag, gives every other marker component 0 and counts those)
        for (uint32_t j = C + (uint32_t)tid; j < Cn; j += RS_BLOCK) {
            const uint32_t slot = j & bmask;
            if (DBG) { // the dot as streamed and its Gram corrections, by sweep position (tools/dbg_res.py)
                p.trace[8 * RS_TRACE + j % RS_TRACE] = (unsigned long long)__double_as_longlong(sh.dpr[slot]);
                p.trace[9 * RS_TRACE + j % RS_TRACE] = (unsigned long long)__double_as_longlong(sh.dp[slot]);
            }
            if (sh.ada[slot]) {
                const int marker = sh.marker[slot];
                p.acum[marker] = sh.num[slot];
                if (found && j == qpos) {
                    p.beta[marker] = bnew;
                    p.comp[marker] = kq | RS_EVENT_FLAG;
                    atomicAdd(&sh.lcass[sh.grp[slot] * K + kq], 1);
                }
            }
        }
        if (tid == 0) {
            sh.fl[WF_FOUND] = 0u;
            sh.fl[WF_POSTED] = 0u;
        }
        rs_lds_barrier(); // (the results above are stores nobody here reads)
        const uint32_t Sn = (Cn + B < M) ? Cn + B : M;
        prefetch(Sx, Sn, seq);
        if (rpos >= (uint32_t)MT_N) { // the stream crossed into the next block: it becomes the current one
            for (int i = tid; i < MT_N; i += RS_BLOCK) {
                sh.mt[i] = sh.mt[MT_N + i];
                sh.tq[i] = sh.tq[MT_N + i];
            }
            rpos -= (uint32_t)MT_N;
            has_next = false;
        }
        rs_lds_barrier();
        batch_pivots(seq, Cn, Sn);
        Sx = Sn;
        C = Cn;
        if (DBG && tid == 0) p.progress[0] = (n_rounds << 8) | 6u;
        fold_pass(); // dots of the refills that have arrived meanwhile (off the chain: the streaming workgroups are busy with the message)
        // fired pivots are on record only while a column streamed before their update is still without its dot: positions in front of
        // F all have theirs, and the batches behind are in position order
        if (F >= Sx) {
            pf_n = 0;
        } else {
            const uint32_t bF = sh.batch[F & bmask];
            uint32_t keep = 0;
            for (uint32_t f = 0; f < pf_n; ++f) keep += sh.pf_msg[f] > bF ? 1u : 0u;
            if (keep == 0u) pf_n = 0; // (entries are in message order: either a tail survives or nothing -- a partial prune compacts below)
            else if (keep < pf_n) {
                __syncthreads();
                if (tid == 0) {
                    uint32_t w = 0;
                    for (uint32_t f = 0; f < pf_n; ++f)
                        if (sh.pf_msg[f] > bF) {
                            sh.pf_pos[w] = sh.pf_pos[f];
                            sh.pf_msg[w] = sh.pf_msg[f];
                            sh.pf_val[3 * w] = sh.pf_val[3 * f];
                            sh.pf_val[3 * w + 1] = sh.pf_val[3 * f + 1];
                            sh.pf_val[3 * w + 2] = sh.pf_val[3 * f + 2];
                            ++w;
                        }
                }
                pf_n = keep;
                __syncthreads();
            }
        }
        lap(4);
    }

    if (aborted) {
        ++seq;
        if (tid == 0) {
            rs_store16(p.msg + (seq % RS_MSG), rs_u4(rs_msg_word0((uint32_t)RS_ABORT, 0u, seq, 0u, 0u), seq, 0u, 0u));
            atomicMax(&p.state->error, sh.fl[WF_ERR] ? sh.fl[WF_ERR] : 3u);
        }
    }
    __syncthreads();
    for (int i = tid; i < MT_N; i += RS_BLOCK) p.mt[i] = sh.mt[i];
    for (int i = tid; i < p.GK; i += RS_BLOCK) p.cass[i] = sh.lcass[i];
    if (tid == 0) {
        ResState* st = p.state;
        st->cursor = C;
        st->rng_idx = rpos;
        st->rounds = n_rounds;
        st->events = n_events;
        st->advances = n_adv;
        st->nnz = n_nnz;
        st->chunks = n_chunks;
        st->refolds = n_refold;
        st->pivots = n_pivots;
        st->predicted = n_pred;
        st->shader_ticks = __builtin_amdgcn_s_memtime() - clk0;
        st->wall_ticks = wall_clock64() - wall0;
        if (DBG)
            for (int i = 0; i < 8; ++i) st->t[i] = tacc[i];
    }
}

// Behind the sweep, one thread per sweep position: the marker's numerator (left in Acum's slot by the walker) becomes
// Acum = 1 / sum_l exp(logL_l - logL_0) with the reference's arithmetic and order (:1883-1905: logL_0 = logpi_0,
// logL_l = logpi_l - hlog_l + (num / denom_l) num / (2 sigmaE); a difference beyond 700 makes it 0); a marker that was no event
// has component 0 and counts for cass[group][0] (the walker counted the events); one outside the adaptive set gets effect 0, Acum 1.
__global__ __launch_bounds__(256) void k_res_finish(ResParams p)
{
    __shared__ int lc[256]; // GK <= 256 (resident_plan)
    if (p.state->error != 0u) return; // the sweep was given up: the slots hold no numerators
    for (int i = threadIdx.x; i < 256; i += blockDim.x) lc[i] = 0;
    __syncthreads();
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < p.M) {
        const int ga = p.s_ga[j], marker = p.order[j], K = p.K;
        if (ga & 0x40000000) {
            const int g0 = (ga & 0x0fffffff) * K;
            const double num = p.acum[marker];
            const double L0 = p.logpi[g0];
            double sum = 0.0;
            bool big = false;
            for (int l = 0; l < K; ++l) {
                double L = L0;
                if (l > 0) {
                    const double mk = num / p.denom[g0 + l];
                    L = p.logpi[g0 + l] - p.hlog[g0 + l] + mk * num * p.i_2sigE;
                }
                const double d = L - L0;
                sum += exp(d);
                big = big || (l >= 1 && fabs(d) > 700.0);
            }
            p.acum[marker] = big ? 0.0 : 1.0 / sum;
            const int c = p.comp[marker];
            if (c & RS_EVENT_FLAG) {
                p.comp[marker] = c & (RS_EVENT_FLAG - 1);
            } else {
                p.comp[marker] = 0;
                atomicAdd(&lc[g0], 1);
            }
        } else {
            p.beta[marker] = 0.0;
            p.acum[marker] = 1.0;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < p.GK; i += blockDim.x)
        if (lc[i]) atomicAdd(p.cass + i, lc[i]);
}

} // namespace hg

#include "hg_walker2.hip.h"
#include "hg_streamer2.hip.h"

namespace hg {

// Every workgroup of the grid waits for the others inside the sweep: all W + 1 must be resident at once.  The host checks that against
// the occupancy query, which cannot see another process (or stream) on the device -- so the kernel makes sure itself, BEFORE it touches
// eps, the effects or the generator: every workgroup counts itself in and waits, bounded, until all are there.  A grid that is only
// partly resident ends here with error 6 and nothing written (hgibbs_sweep then runs the sweep on the batch engine).
__device__ __forceinline__ bool rs_rendezvous(const ResParams& p, unsigned char* smem)
{
    volatile uint32_t* const flag = reinterpret_cast<volatile uint32_t*>(smem);
    if (threadIdx.x == 0) {
        unsigned long long* const cnt = p.progress + 3;
        __hip_atomic_fetch_add(cnt, 1ull, HG_RLX_AGENT);
        const unsigned long long t0 = wall_clock64();
        uint32_t ok = 1u;
        while (__hip_atomic_load(cnt, HG_RLX_AGENT) < (unsigned long long)p.W + 1ull) {
            if (__hip_atomic_load(p.progress + 2, HG_RLX_AGENT) != 0ull || wall_clock64() - t0 > p.rdv_timeout) {
                __hip_atomic_store(p.progress + 2, 6ull, HG_RLX_AGENT); // (the others give up at once)
                atomicMax(&p.state->error, 6u);
                ok = 0u;
                break;
            }
            __builtin_amdgcn_s_sleep(8);
        }
        if (ok && __hip_atomic_load(p.progress + 2, HG_RLX_AGENT) != 0ull) ok = 0u; // (somebody timed out while this one was arriving)
        *flag = ok;
    }
    __syncthreads();
    const bool ok = *flag != 0u;
    __syncthreads(); // (the word is the streaming workgroups' pair table and the walker's first array: nobody writes it before everybody has read it)
    return ok;
}

// Several ranks, probe launch (hgibbs.hip, resident_probe): are ALL ranks' grids resident AT THE SAME TIME?  A rank's own rendezvous cannot
// tell: two processes that share a device may have their kernels run one after the other, each grid resident at once, and each sweep
// would wait its full time-out for a peer that has not started.  The walkers shake hands through the mailboxes in two steps -- "my grid
// is resident" to every peer, then, when every peer has said so, "everybody's is, as far as I see" -- while every workgroup of the grid
// stays where it is; a rank that has every peer's second word was resident together with all of them at the moment the last first word
// was sent.  Both waits are bounded (rdv_timeout); a rank that gives up says so in state->error (6) and the host's all-reduce tells the others.
__device__ __forceinline__ void rs_probe_peers(const ResParams& p)
{
    if (threadIdx.x != 0) return; // (the workgroup keeps its compute unit as long as one wave is here)
    unsigned long long* const done = p.progress + 4;
    if (blockIdx.x == p.W) {
        const unsigned long long tag = 0x5a00000000000000ull | (p.sweep_id << 8);
        uint32_t ok = 1u;
        for (unsigned long long phase = 1ull; phase <= 2ull && ok; ++phase) {
            for (int r = 0; r < p.nranks; ++r)
                if (r != p.rank) __hip_atomic_store(rx_hbox(p.mbox[r], p.rank) + (phase - 1ull), tag | phase, HG_RLX_SYSTEM);
            const unsigned long long t0 = wall_clock64();
            for (int r = 0; r < p.nranks && ok; ++r) {
                if (r == p.rank) continue;
                while (__hip_atomic_load(rx_hbox(p.mbox[p.rank], r) + (phase - 1ull), HG_RLX_SYSTEM) != (tag | phase)) {
                    if (wall_clock64() - t0 > p.rdv_timeout) {
                        ok = 0u;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(16);
                }
            }
        }
        if (!ok) atomicMax(&p.state->error, 6u);
        __hip_atomic_store(done, 1ull, HG_RLX_AGENT);
    } else {
        const unsigned long long t0 = wall_clock64();
        while (__hip_atomic_load(done, HG_RLX_AGENT) == 0ull && wall_clock64() - t0 < 4ull * p.rdv_timeout + 100000000ull) __builtin_amdgcn_s_sleep(64);
    }
}

template <int T, int DBG, int MISS>
__global__ __launch_bounds__(RS_BLOCK) __attribute__((amdgpu_num_vgpr(RS_VGPR_LIMIT / 2))) void k_sweep_resident(ResParams p, const ResParams* pg)
{
    if (!rs_rendezvous(p, hg_smem)) return;
    if (p.M == 0xffffffffu) { // (a probe launch: is the grid resident at once, and together with the peers' -- hgibbs.hip, resident_probe)
        if (p.nranks > 1) rs_probe_peers(p);
        return;
    }
    if (blockIdx.x < p.W) res_streamer<T, DBG, MISS>(p, hg_smem);
    else if (p.walker == 2) res_walker2<DBG, MISS>(*pg); // pg: the same parameters in device memory (a reference the called function can read with scalar loads)
    else res_walker<DBG, MISS>(*pg, hg_smem);
}

// the same grid with the streaming workgroups' second form (hg_streamer2.hip.h: the refill's dots as integer matrix products)
template <int T, int DBG, int MISS>
__global__ __launch_bounds__(RS_BLOCK) __attribute__((amdgpu_num_vgpr(RL_VGPR_LIMIT / 2))) void k_sweep_limb(ResParams p, const ResParams* pg)
{
    if (!rs_rendezvous(p, hg_smem)) return;
    if (p.M == 0xffffffffu) {
        if (p.nranks > 1) rs_probe_peers(p);
        return;
    }
    if (blockIdx.x < p.W) res_streamer_limb<T, DBG, MISS>(p, hg_smem);
    else if (p.walker == 2) res_walker2<DBG, MISS>(*pg);
    else res_walker<DBG, MISS>(*pg, hg_smem);
}

// (four tiles per workgroup: 32 bytes of a column per lane and group -- more named registers, hence a register cap of its own)
template <int DBG, int MISS>
__global__ __launch_bounds__(RS_BLOCK) __attribute__((amdgpu_num_vgpr(RL_VGPR_LIMIT4 / 2))) void k_sweep_limb4(ResParams p, const ResParams* pg)
{
    if (!rs_rendezvous(p, hg_smem)) return;
    if (p.M == 0xffffffffu) {
        if (p.nranks > 1) rs_probe_peers(p);
        return;
    }
    if (blockIdx.x < p.W) res_streamer_limb<4, DBG, MISS>(p, hg_smem);
    else if (p.walker == 2) res_walker2<DBG, MISS>(*pg);
    else res_walker<DBG, MISS>(*pg, hg_smem);
}

} // namespace hg
