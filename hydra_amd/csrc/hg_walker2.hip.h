// hg_walker2.hip.h -- the resident engine's walker, second form: ONE wave walks the chain, the others serve it.
//
// The first walker (res_walker, hg_resident.hip.h) runs the whole workgroup through every step of a round in lockstep: the bound
// test, the exact decision, the message, then results, metadata prefetch and the fold of arrived raw dots -- three dependent trips
// to global memory and a dozen workgroup barriers per round, all of them between one message and the next.  Where an event needs no
// round trip through the streaming workgroups (a predicted pivot: its Gram terms came with the columns), that serial work IS the
// round.  Here the roles are split over the workgroup's eight waves and nothing on the chain ever waits for global memory:
//
//   wave 0      the CHAIN: per window position (four to a lane, in registers) the dot as streamed, its Gram corrections, mave, mstd
//               and the old effect; per round [absorb dots that arrived | bound test of the positions behind the cursor, no
//               exponential (the tabulated convex bound) | first candidate by ballots | the reference's exact decision and draw
//               (a5-a7, src/BayesRRm.cpp:1744-1753,1859-1921) | message | corrections of the positions behind the event].  It reads and
//               writes LDS only, and posts the 16-byte message; no barrier, no vmcnt wait.
//   wave 1      the FOLDER: polls the batch counters, turns the fixed-point sums of completed refill batches into dots (and the
//               columns' Gram terms with their batch's pivots) and publishes "every position below F has its dot".
//   wave 2      the HOUSEKEEPER: generator blocks and their thresholds ahead of the chain (a ring of four), marker metadata two
//               windows ahead, the list of predicted positions, and the results of consumed positions (numerator into Acum's slot,
//               new effect and component of an event) out to global memory.
//   waves 4-7   the COLLECTORS: when the chain posts an event that needs the round trip, they poll the Gram accumulators' shard
//               rows (two each) until every word carries its full arrival count and leave the sums in LDS.
//
// The waves hand over through LDS words (a writer finishes its data, waits lgkmcnt(0), then publishes a counter; LDS serves a
// wave's operations in order).  Every wait is bounded by ResParams::timeout and ends the sweep with error 3.
// Applies where every marker takes a uniform (adaV all ones), the mixture tables fit LDS with at most RS_FG groups, and the shard is
// not split over several ranks; the first walker takes the rest.
#pragma once

namespace hg {

constexpr int W2_NBLK = 4;                   // generator blocks in the ring
constexpr uint32_t W2_RING = W2_NBLK * MT_N; // words
constexpr uint32_t W2_RR = 1024;             // results ring (positions)
constexpr uint32_t W2_EV = 64;               // event records in flight to the housekeeper
constexpr uint32_t W2_PRED = 1024;           // predicted positions staged
constexpr uint32_t W2_NB = 256;              // refill batches on record (a batch on record has at least one position in the window)
constexpr int W2_NCOL = 4;                   // collector waves (4 .. 7)

enum {
    S_FPUB = 0, // folder -> chain: every position below has its dot
    S_MPUB,     // housekeeper -> chain: positions below have their metadata staged
    S_CPUB,     // chain -> housekeeper: cursor (positions below are consumed, their numerators are in the results ring)
    S_SXPUB,    // chain -> folder: positions below have been announced to the streaming workgroups (batch[] is set)
    S_GREQ,     // chain -> collectors: number of the event whose Gram terms are wanted
    S_GV,       //   ... columns behind it
    S_GDONE,    // [W2_NCOL] collectors -> chain: number of the event whose sums are in gpart
    S_BLK = S_GDONE + W2_NCOL, // housekeeper -> chain: generator blocks made
    S_GPOS,     // chain -> housekeeper: generator words consumed (absolute, from the start of the block the sweep began in)
    S_PLD,      // housekeeper -> chain: predicted positions staged (index)
    S_PCUR,     // chain -> housekeeper: index of the first predicted position at or behind the cursor
    S_RDONE,    // folder -> chain: refill batches every streaming workgroup has completed
    S_EVN,      // chain -> housekeeper: events recorded
    S_ABORT,
    S_END,
    S_ERR,
    S_WPUB,     // housekeeper -> chain: results of the positions below are written (their ring entries and metadata slots are free)
    S_EVW,      // housekeeper -> chain: event records taken
    S_NWORDS = 32
};

// LDS of the walker workgroup.  The carve-up is a list of byte offsets (host and device agree on its end); the device takes its
// pointers from it with the LDS address space in their TYPE: res_walker2 is a function of its own (not inlined into the kernel), and
// through a generic pointer every access would be a FLAT instruction -- slower, and counted by vmcnt as well as lgkmcnt, so that a wait
// for an LDS read would also wait for every global store in flight.
#define W2_ARRAYS(X)                                                                                                                          \
    X(double, tq, (size_t)W2_RING)        /* per generator word: the largest f = log sum_l>0 exp(logL_l - logL_0) that cannot give an event */ \
    X(double, zig_nx, 130)                                                                                                                    \
    X(double, zig_ny, 130)                                                                                                                    \
    X(double, htab, (size_t)4 * HT_LDS)                                                                                                       \
    X(double, qtab, (size_t)2 * HT_LDS)                                                                                                       \
    X(double, ftab, (size_t)RS_FG * (RS_FN + 1))                                                                                              \
    X(double, fscale, RS_FG)                                                                                                                  \
    X(double, mave, MR)                   /* metadata ring, by position mod 2 B */                                                            \
    X(double, mstd, MR)                                                                                                                       \
    X(double, bold, MR)                                                                                                                       \
    X(double, gsum, MR)                   /* (build MISS) */                                                                                  \
    X(double, nmis, MR)                                                                                                                       \
    X(double, dpr, Bz)                    /* the dot as streamed, by window slot */                                                           \
    X(double, rnum, (size_t)W2_RR)        /* numerators of consumed positions */                                                              \
    X(double, ev_bnew, (size_t)W2_EV)                                                                                                         \
    X(unsigned long long, rprev, (size_t)RS_RB) /* the accumulators' sums as last seen (they only ever grow) */                               \
    X(unsigned long long, rprev2, (size_t)RS_RB)                                                                                              \
    X(unsigned long long, pprev, (size_t)2 * RS_RB)                                                                                           \
    X(double, pf_val, (size_t)RS_PFIRE * 3) /* fired pivots: (dbeta, mave, mstd) */                                                           \
    X(unsigned long long, gpart64, (size_t)W2_NCOL * RS_BMAX) /* the collectors' sums (the plain build uses the first half as 4-byte words) */ \
    X(uint32_t, mt, (size_t)W2_RING)      /* untempered generator words */                                                                    \
    X(int32_t, marker, MR)                                                                                                                    \
    X(int32_t, ga, MR)                                                                                                                        \
    X(uint32_t, wpt, Bz * RS_PMAX)        /* Gram terms with the batch's pivots, by window slot */                                            \
    X(uint32_t, batch, Bz)                /* refill batch (= message number) of the slot's column */                                          \
    X(uint32_t, ev_pos, (size_t)W2_EV)                                                                                                        \
    X(uint32_t, ev_k, (size_t)W2_EV)                                                                                                          \
    X(uint32_t, pred, (size_t)W2_PRED)    /* predicted positions staged, by index mod W2_PRED */                                              \
    X(uint32_t, bl_pi, (size_t)W2_NB)     /* per batch: index of its first pivot in the list of predicted positions */                        \
    X(uint32_t, bl_p0, (size_t)W2_NB)     /*   that pivot's position (0xffffffff: none) */                                                    \
    X(uint32_t, bl_np, (size_t)W2_NB)     /*   its number of pivots */                                                                        \
    X(uint32_t, pf_pos, RS_PFIRE)         /* pivots fired while columns streamed before their update were without their dot */                \
    X(uint32_t, pf_msg, RS_PFIRE)                                                                                                             \
    X(uint32_t, pf_pi, RS_PFIRE)                                                                                                              \
    X(int32_t, lcass, 256)                                                                                                                    \
    X(uint32_t, sw, S_NWORDS)             /* hand-over words */

struct Walk2Off {
#define W2_X(type, name, count) uint32_t name;
    W2_ARRAYS(W2_X)
#undef W2_X
    uint32_t end;
};
__host__ __device__ inline Walk2Off walk2_offsets(uint32_t B)
{
    Walk2Off o;
    const size_t MR = (size_t)2 * B, Bz = B;
    size_t q = 0;
#define W2_X(type, name, count) \
    o.name = (uint32_t)q;       \
    q += sizeof(type) * (size_t)(count);
    W2_ARRAYS(W2_X)
#undef W2_X
    o.end = (uint32_t)q;
    return o;
}
__host__ __device__ inline size_t rs_walker2_lds(uint32_t B) { return (size_t)walk2_offsets(B).end + 64; }

#define W2_LDS __attribute__((address_space(3)))
// hand-over words: a plain LDS load / store the compiler may neither cache nor move; the value is the same in every lane and is
// handed on as a scalar
__device__ __forceinline__ uint32_t w2_ld(const W2_LDS uint32_t* w) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)); }
__device__ __forceinline__ void w2_st(W2_LDS uint32_t* w, uint32_t v) { __hip_atomic_store(w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
// a double that is the same in every lane (read from LDS through a uniform address), as a scalar: what depends on it is then scalar
// control flow, not a masked vector branch
__device__ __forceinline__ double w2_uni(double v) { return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v))); }
__device__ __forceinline__ uint32_t w2_uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ int w2_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
// a store to global memory through a pointer the compiler knows to be global (through a generic one it would be a FLAT instruction, which
// lgkmcnt counts as well: the next wait for an LDS read would wait for the store)
__device__ __forceinline__ void w2_gst(unsigned long long* p, unsigned long long v) { *(__attribute__((address_space(1))) unsigned long long*)p = v; }
// everything this wave has written to LDS is in place (LDS serves a wave's operations in order; the wait makes the order hold for the
// instruction stream, the clobber for the compiler)
__device__ __forceinline__ void w2_lds_done() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// word source over the ring of generator blocks (untempered words)
struct RingGen {
    const W2_LDS uint32_t* w;
    uint32_t pos, left, err, n;
    __device__ __forceinline__ uint32_t next()
    {
        if (!left) {
            err = 2u;
            return 0u;
        }
        const uint32_t v = mt_temper(w[pos]);
        pos = pos + 1u == W2_RING ? 0u : pos + 1u;
        --left;
        ++n;
        return v;
    }
};

// the normal Ziggurat's layers in LDS (the exponential's, for the rare tail, stay in global memory)
struct ZigLds {
    const W2_LDS double* nx;
    const W2_LDS double* ny;
    const double* ex;
    const double* ey;
};

struct Walk2Lds {
#define W2_X(type, name, count) W2_LDS type* name;
    W2_ARRAYS(W2_X)
#undef W2_X
};

__device__ __forceinline__ Walk2Lds walk2_lds(uint32_t B)
{
    Walk2Lds sh;
    const Walk2Off off = walk2_offsets(B);
    W2_LDS unsigned char* const lbase = (W2_LDS unsigned char*)hg_smem;
#define W2_X(type, name, count) sh.name = (W2_LDS type*)(lbase + off.name);
    W2_ARRAYS(W2_X)
#undef W2_X
    return sh;
}

// what every role's function starts with: the parameters it reads again and again, once (they live in global memory: the memory
// clobbers of the hand-over waits would otherwise make every use a scalar load of its own), the LDS arrays, the small helpers.
// The roles are functions of their own so that each gets a register allocation of its own (the chain's four slots per lane want most
// of the register file).
#define W2_PROLOGUE \
    const int tid = threadIdx.x, lane = tid & 63; \
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6); \
    const uint32_t B = pr.B, bmask = B - 1u, M = pr.M; \
    const uint32_t MR = 2u * B, mrmask = MR - 1u; \
    const int K = pr.K, GK = pr.GK; \
    const uint32_t W = pr.W, nsh = pr.nsh, rsh = pr.rsh; \
    const double n_total = pr.n_total, n_minus_1 = pr.n_minus_1, i_2sigE = pr.i_2sigE, eps_sum = pr.eps_sum, fx_unscale = pr.fx_unscale; \
    const unsigned long long timeout = pr.timeout; \
    const bool pivots = pr.pivots != 0; \
    const uint32_t rng_idx0 = pr.rng_idx; \
    ResMsg* const msg = pr.msg; \
    ResState* const state = pr.state; \
    unsigned long long* const trace = pr.trace; \
    unsigned long long* const progress = pr.progress; \
    const uint32_t* const gpred = pr.pred; \
    const double* const zig_ex = pr.zig.ex; \
    const double* const zig_ey = pr.zig.ey; \
    const Walk2Lds sh = walk2_lds(B); \
    W2_LDS uint32_t* const gpart = (W2_LDS uint32_t*)sh.gpart64; \
    const uint32_t cntG[2] = {W / nsh + (W % nsh ? 1u : 0u), W / nsh}; \
    const uint32_t cntR[2] = {W / rsh + (W % rsh ? 1u : 0u), W / rsh}; \
    const uint32_t Sx0 = (B < M) ? B : M; \
    const uint32_t m0 = MR < M ? MR : M; \
    auto stage_meta = [&](uint32_t j) { \
        const uint32_t ms = j & mrmask; \
        const int ga = pr.s_ga[j]; \
        const int mk = pr.order[j]; \
        sh.marker[ms] = mk; \
        sh.ga[ms] = ga; \
        sh.bold[ms] = pr.s_bold[j]; \
        sh.mave[ms] = pr.s_mave[j]; \
        sh.mstd[ms] = pr.s_mstd[j]; \
        if constexpr (MISS) { \
            const unsigned long long* cn = pr.counts + 3ull * (unsigned long long)mk; \
            sh.gsum[ms] = (double)(cn[0] + 2ull * cn[1]); \
            sh.nmis[ms] = (double)cn[2]; \
        } \
    }; \
    auto tq_of = [&](uint32_t word) { \
        const double prob = (double)mt_temper(word) * (1.0 / 4294967296.0); \
        return log(1.0 / prob - 1.0) - 1e-9; \
    }; \
    auto aborted = [&]() { return w2_ld(sh.sw + S_ABORT) != 0u; }; \
    auto ended = [&]() { return w2_ld(sh.sw + S_END) != 0u; };

template <int DBG, int MISS>
__device__ __attribute__((noinline)) void w2_chain(const ResParams& pr)
{
    W2_PROLOGUE
    // =====================================================================================================================
    // the chain.  Everything per window position is straight-line code over the lane's four slots (reads with clamped addresses,
    // results selected): the four dependent chains then run interleaved -- one LDS round trip per step, not one per slot and step
    // =====================================================================================================================
    __builtin_amdgcn_s_setprio(3);
    const uint32_t NSL = B >= 64u ? B / 64u : 1u; // window slots per lane: slot = 64 i + lane
    uint32_t C = 0, Sx = Sx0, Fs = 0, base = 0, seq = 0, nev = 0, pi = 0, evn = 0, pf_n = 0;
    uint32_t gpos = rng_idx0, gposr = rng_idx0 % W2_RING, blk = 1u;
    // per slot: the dot as streamed, its Gram corrections, the marker's (mave, mstd, old effect x (N - 1)), the tabulated bound's scale and table offset of its group, its refill batch and that batch's first pivot index
    double dpr[4], dp[4], mave[4], mstd[4], boldn[4], fsc[4], gsm[4], nms[4];
    uint32_t fof[4], sbt[4], spi[4];
    bool prd[4], son[4]; // the marker's effect is non-zero at sweep start (a predicted event); the slot exists
    uint32_t sl[4];      // the slot (clamped to an existing one)
    uint32_t n_rounds = 0, n_events = 0, n_adv = 0, n_nnz = 0, n_chunks = 0, n_refold = 0, n_pivots = 0, n_pred = 0; // (scalars: the control flow around them is uniform)
    unsigned long long tacc[8] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull};
    unsigned long long tmark = DBG ? wall_clock64() : 0ull;
    auto lap = [&](int i) {
        if (DBG) {
            const unsigned long long now = wall_clock64();
            tacc[i] += now - tmark;
            tmark = now;
        }
    };
    uint32_t err = 0;
    if (DBG) { // what a clock read costs (ticks per 64 reads, in t[7]): the stage clocks above include one each
        const unsigned long long c0 = wall_clock64();
        unsigned long long c1 = c0;
#pragma unroll 1
        for (int i = 0; i < 64; ++i) c1 = wall_clock64() + (c1 & 1ull);
        tacc[7] = c1 - c0;
        // the speed of this wave where it runs (next to the other roles' polling): 1024 dependent f64 adds (t[5]), 256 dependent LDS reads (t[6])
        {
            double a = (double)lane;
            const unsigned long long d0 = wall_clock64();
#pragma unroll 1
            for (int i = 0; i < 128; ++i) {
                asm volatile("v_add_f64 %0, %0, 1.0\n\tv_add_f64 %0, %0, 1.0\n\tv_add_f64 %0, %0, 1.0\n\tv_add_f64 %0, %0, 1.0\n\tv_add_f64 %0, %0, 1.0\n\tv_add_f64 %0, %0, 1.0\n\tv_add_f64 %0, %0, 1.0\n\tv_add_f64 %0, %0, 1.0" : "+v"(a));
            }
            const unsigned long long d1 = wall_clock64();
            uint32_t x = (uint32_t)lane & 3u;
#pragma unroll 1
            for (int i = 0; i < 256; ++i) x = sh.pf_pos[x & 15u] & 3u;
            const unsigned long long d2 = wall_clock64();
            tacc[7] += (unsigned long long)(x & 0u) + (unsigned long long)(a < 0.0 ? 1 : 0) + 0ull * (d2 - d0 + d1);
        }
        tmark = wall_clock64();
    }
    auto pos_of_slot = [&](uint32_t s, uint32_t c) { return c + ((s - c) & bmask); };
    // the slot's new position j (where `take`): metadata from the ring, no dot yet, no corrections yet
    auto take_meta = [&](int i, uint32_t j, bool take, uint32_t batch_no, uint32_t batch_pi) {
        const uint32_t ms = (take ? j : 0u) & mrmask;
        const double a = sh.mave[ms], d = sh.mstd[ms], b = sh.bold[ms];
        const int g = sh.ga[ms] & 0x0fffffff;
        const double sc = sh.fscale[g];
        mave[i] = take ? a : mave[i];
        mstd[i] = take ? d : mstd[i];
        boldn[i] = take ? b * n_minus_1 : boldn[i];
        prd[i] = take ? b != 0.0 : prd[i];
        fsc[i] = take ? sc : fsc[i];
        fof[i] = take ? (uint32_t)g * (uint32_t)(RS_FN + 1) : fof[i];
        dp[i] = take ? 0.0 : dp[i];
        dpr[i] = take ? 0.0 : dpr[i];
        sbt[i] = take ? batch_no : sbt[i];
        spi[i] = take ? batch_pi : spi[i];
        if constexpr (MISS) {
            const double gs = sh.gsum[ms], nm = sh.nmis[ms];
            gsm[i] = take ? gs : gsm[i];
            nms[i] = take ? nm : nms[i];
        }
    };
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        dpr[i] = dp[i] = mave[i] = mstd[i] = boldn[i] = fsc[i] = gsm[i] = nms[i] = 0.0;
        fof[i] = sbt[i] = spi[i] = 0u;
        prd[i] = false;
        const uint32_t s = (uint32_t)i * 64u + (uint32_t)lane;
        son[i] = (uint32_t)i < NSL && s < B;
        sl[i] = son[i] ? s : 0u;
        take_meta(i, s, son[i] && s < Sx0, 0u, 0u);
    }
    // inside a wait: has the sweep been given up, or is it time to give it up
    auto spin_fail = [&](unsigned long long t0) {
        if (aborted()) return true;
        if (wall_clock64() - t0 > timeout) {
            if (lane == 0) w2_st(sh.sw + S_ABORT, 1u);
            return true;
        }
        return false;
    };
    // wait until a counter that only grows has reached the value, with what the chain saw of it last: no LDS round trip while
    // that is enough
    auto wait_seen = [&](uint32_t& seen, int word, uint32_t value) {
        if (seen >= value) return true;
        seen = w2_ld(sh.sw + word);
        if (seen >= value) return true;
        const unsigned long long t0 = wall_clock64();
        for (;;) {
            seen = w2_ld(sh.sw + word);
            if (seen >= value) return true;
            if (spin_fail(t0)) return false;
            __builtin_amdgcn_s_sleep(1);
        }
    };
    uint32_t rdone_seen = 0, wpub_seen = 0, evw_seen = 0, mpub_seen = m0, pld_seen = W2_PRED, fpub_seen = 0;
    uint32_t gdone_seen[W2_NCOL] = {0u, 0u, 0u, 0u};

    while (C < M) {
        ++n_rounds;
        if (lane == 0 && (DBG || (n_rounds & 255u) == 0u)) w2_gst(progress, ((unsigned long long)n_rounds << 8) | 1u);
        // the generator: the words a round can reach exist
        if (gpos + B + 96u > blk * (uint32_t)MT_N && !wait_seen(blk, S_BLK, (gpos + B + 96u + (uint32_t)MT_N - 1u) / (uint32_t)MT_N)) break;
        // ---- the walk: up to the first event, or through the whole window ----
        bool found = false, failed = false;
        uint32_t qpos = 0, q_consumed = 0;
        int q_k = 0;
        double q_bnew = 0.0, q_bold = 0.0;
        for (;;) {
            // the dots that have arrived since the last look
            lap(6);
            fpub_seen = w2_ld(sh.sw + S_FPUB);
            const uint32_t F = fpub_seen;
            if (F > Fs) {
                bool arr[4];
                double dnew[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const uint32_t j = pos_of_slot(sl[i], C);
                    arr[i] = son[i] && j >= Fs && j < F;
                    dnew[i] = sh.dpr[sl[i]];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) dpr[i] = arr[i] ? dnew[i] : dpr[i];
                if (pf_n) { // (uniform) the pivots that fired between a column's streaming and now: their updates were not in the streamed dot
                    for (uint32_t f = 0; f < pf_n; ++f) {
                        const uint32_t fmsg = w2_uni(sh.pf_msg[f]), fpos = w2_uni(sh.pf_pos[f]), fpi = w2_uni(sh.pf_pi[f]);
                        const double fdb = w2_uni(sh.pf_val[3 * f]), fmv = w2_uni(sh.pf_val[3 * f + 1]), fsd = w2_uni(sh.pf_val[3 * f + 2]);
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const uint32_t j = pos_of_slot(sl[i], C);
                            const bool hit = arr[i] && fmsg > sbt[i] && fpos < j;
                            const uint32_t ip = hit ? (fpi - spi[i]) & 3u : 0u;
                            const double A = (double)sh.wpt[sl[i] * RS_PMAX + ip];
                            const double xx = mstd[i] * fsd * (A - n_total * (mave[i] * fmv));
                            dp[i] = hit ? dp[i] + fdb * xx : dp[i];
                        }
                    }
                }
                Fs = F;
                // fired pivots stay on record only while a column streamed before their update is without its dot (entries are in message order)
                if (pf_n) {
                    if (Fs >= Sx) pf_n = 0;
                    else {
                        const uint32_t bF = w2_uni(sh.batch[Fs & bmask]);
                        uint32_t drop = 0;
                        while (drop < pf_n && w2_uni(sh.pf_msg[drop]) <= bF) ++drop;
                        if (drop) {
                            if (lane == 0)
                                for (uint32_t f = drop; f < pf_n; ++f) {
                                    sh.pf_pos[f - drop] = sh.pf_pos[f];
                                    sh.pf_msg[f - drop] = sh.pf_msg[f];
                                    sh.pf_pi[f - drop] = sh.pf_pi[f];
                                    sh.pf_val[3 * (f - drop)] = sh.pf_val[3 * f];
                                    sh.pf_val[3 * (f - drop) + 1] = sh.pf_val[3 * f + 1];
                                    sh.pf_val[3 * (f - drop) + 2] = sh.pf_val[3 * f + 2];
                                }
                            pf_n -= drop;
                            w2_lds_done();
                        }
                    }
                }
            }
            lap(5);
            if (base >= Fs) {
                if (base >= Sx) break; // the whole window has been walked: a round that only advances
                // the walk needs dots that are still on their way
                ++n_refold;
                if (!wait_seen(fpub_seen, S_FPUB, Fs + 1u)) {
                    failed = true;
                    break;
                }
                lap(0);
                continue;
            }
            // ---- the bound test of the positions [base, Fs): "this marker cannot be an event" ----
            ++n_chunks;
            unsigned long long cm[4];
            {
                bool act[4], inside[4];
                double x[4], tqv[4], f0[4], f1[4];
                int kx[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const uint32_t j = pos_of_slot(sl[i], C);
                    act[i] = son[i] && j >= base && j < Fs;
                    const double num = (dpr[i] + dp[i]) + boldn[i];
                    if (act[i]) sh.rnum[j & (W2_RR - 1u)] = num; // the numerator as last tested: final once the position is consumed
                    x[i] = (num * num) * fsc[i];
                    inside[i] = x[i] < (double)RS_FN && fsc[i] > 0.0; // (NaN: outside)
                    kx[i] = inside[i] ? (int)x[i] : 0;
                    uint32_t r = gposr + ((j - C) & bmask);
                    r = r >= W2_RING ? r - W2_RING : r;
                    f0[i] = sh.ftab[fof[i] + (uint32_t)kx[i]];
                    f1[i] = sh.ftab[fof[i] + (uint32_t)kx[i] + 1u];
                    tqv[i] = sh.tq[r];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const double fup = f0[i] + (x[i] - (double)kx[i]) * (f1[i] - f0[i]);
                    cm[i] = __ballot(act[i] && (prd[i] || !(inside[i] && fup <= tqv[i])));
                }
            }
            lap(2);
            // ---- the candidates in window order, until one is an event ----
            const uint32_t s0 = C & bmask, i0 = s0 >> 6, l0 = s0 & 63u;
            const unsigned long long lowm = (1ull << l0) - 1ull;
            for (;;) {
                uint32_t qs = 0xffffffffu;
#pragma unroll
                for (int t = 0; t <= 4; ++t) {
                    if ((uint32_t)t <= NSL && qs == 0xffffffffu) {
                        uint32_t i = i0 + (uint32_t)t;
                        i = i >= NSL ? i - NSL : i;
                        unsigned long long mm = i == 0u ? cm[0] : (i == 1u ? cm[1] : (i == 2u ? cm[2] : cm[3]));
                        if (t == 0) mm &= ~lowm;
                        if ((uint32_t)t == NSL) mm &= lowm;
                        if (mm) qs = i * 64u + (uint32_t)(__ffsll((long long)mm) - 1);
                    }
                }
                qs = w2_uni(qs);
                if (qs == 0xffffffffu) break; // uniform
                const uint32_t qc = pos_of_slot(qs, C);
                const uint32_t qi = qs >> 6, ql = qs & 63u;
                const uint32_t ms = qc & mrmask;
                uint32_t upos = gposr + (qc - C);
                upos = upos >= W2_RING ? upos - W2_RING : upos;
                // (one round trip for all of the candidate's values)
                const double bold_v = sh.bold[ms], num_v = sh.rnum[qc & (W2_RR - 1u)];
                const int ga_v = sh.ga[ms];
                const uint32_t word_v = sh.mt[upos];
                const double bold = w2_uni(bold_v), num = num_v;
                const int g0 = w2_uni(ga_v & 0x0fffffff) * K;
                const double prob = (double)mt_temper(word_v) * (1.0 / 4294967296.0);
                // a5 (src/BayesRRm.cpp:1859-1921) over the lanes: lane x < K holds logL_x; lane 8 kk + l the term exp(logL_l - logL_kk)
                double Lm = 0.0;
                {
                    const int lk = lane < K ? lane : 0;
                    const double den = sh.htab[g0 + lk], lpi = sh.htab[HT_LDS + g0 + lk], hlg = sh.htab[2 * HT_LDS + g0 + lk];
                    const double mk = num / (lk ? den : 1.0);
                    Lm = lk ? lpi - hlg + mk * num * i_2sigE : lpi;
                }
                const int kk = lane >> 3, l = lane & 7;
                const double Ll = __shfl(Lm, l, 64), Lk = __shfl(Lm, kk, 64);
                const bool on = kk < K - 1 && l < K;
                const double d = on ? Ll - Lk : 0.0;
                const double ex = exp(d);
                const bool bigp = on && l >= (kk ? kk : 1) && fabs(d) > 700.0;
                const unsigned long long bm = __ballot(bigp);
                double sum = ex, cur = ex;
#pragma unroll
                for (int x = 1; x < 8; ++x) {
                    cur = rs_dpp_f64<0x101>(cur); // row_shl:1 -- lane i takes lane i + 1's
                    if (x < K) sum += cur;        // wave-uniform
                }
                const bool anyb = ((bm >> (lane & ~7)) & 0xffull) != 0ull;
                const double thr = anyb ? 0.0 : 1.0 / sum; // of walk step kk, valid in the first lane of its group
                int k = K - 1;
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
                k = w2_uni(k);
                double bnew = 0.0;
                uint32_t consumed = 0u, gerr = 0u;
                if (k > 0) { // (uniform) a7: the new effect, on one lane
                    if (lane == 0) {
                        RingGen g{sh.mt, upos + 1u == W2_RING ? 0u : upos + 1u, blk * (uint32_t)MT_N - (gpos + (qc - C) + 1u), 0u, 0u};
                        const ZigLds zt{sh.zig_nx, sh.zig_ny, zig_ex, zig_ey};
                        bnew = norm_rng_sd(g, zt, num / sh.htab[g0 + k], sh.htab[3 * HT_LDS + g0 + k]);
                        consumed = g.n;
                        gerr = g.err;
                    }
                    bnew = w2_uni(bnew);
                    consumed = (uint32_t)__builtin_amdgcn_readlane((int)consumed, 0);
                    gerr = (uint32_t)__builtin_amdgcn_readlane((int)gerr, 0);
                }
                if (gerr) {
                    err = gerr;
                    failed = true;
                    break;
                }
                if (k != 0 || bold != 0.0) {
                    found = true;
                    qpos = qc;
                    q_k = k;
                    q_bnew = bnew;
                    q_bold = bold;
                    q_consumed = consumed;
                    break;
                }
                // too close to call, and no event: on to the next candidate
                const unsigned long long bit = 1ull << ql;
                if (qi == 0u) cm[0] &= ~bit;
                else if (qi == 1u) cm[1] &= ~bit;
                else if (qi == 2u) cm[2] &= ~bit;
                else cm[3] &= ~bit;
            }
            lap(3);
            if (found || failed) break;
            base = Fs; // every position below has passed; on to the dots still to come
        }
        if (failed) break;

        // ---- the message: an event at qpos, or a round that only moves the window on ----
        const uint32_t ncons = found ? qpos - C + 1u : Sx - C;
        const uint32_t Cn = C + ncons;
        const uint32_t Sn = (Cn + B < M) ? Cn + B : M;
        const double dbeta = found ? q_bold - q_bnew : 0.0;
        const bool is_event = found && dbeta != 0.0;
        const bool predicted = found && q_bold != 0.0;
        const uint32_t qms = (found ? qpos : C) & mrmask;
        const double mq_v = sh.mave[qms], sq_v = sh.mstd[qms], gsq_v = MISS ? sh.gsum[qms] : 0.0, nmq_v = MISS ? sh.nmis[qms] : 0.0;
        const uint32_t blo_v = sh.batch[(qpos + 1u) & bmask]; // (the oldest batch with columns behind the event, if there are any)
        // a predicted pivot whose Gram terms came with the columns: the oldest batch with columns behind it lists it
        bool pivot = false;
        if (pivots && is_event && predicted && pf_n < (uint32_t)RS_PFIRE)
            pivot = qpos + 1u >= Sx || pi - w2_uni(sh.bl_pi[w2_uni(blo_v) % W2_NB]) < (uint32_t)RS_PMAX;
        // flow control: never more than RS_MSG - 3 messages ahead of the slowest streaming workgroup (batches completed = messages
        // taken + 1); room in the results ring and on the event record
        if (seq + 6u > (uint32_t)RS_MSG && !wait_seen(rdone_seen, S_RDONE, seq + 6u - (uint32_t)RS_MSG)) break;
        if (Cn + B > W2_RR && !wait_seen(wpub_seen, S_WPUB, Cn + B - W2_RR)) break;
        if (evn + 2u > W2_EV && !wait_seen(evw_seen, S_EVW, evn + 2u - W2_EV)) break;
        ++seq;
        if (lane == 0) {
            const uint32_t kf = (pivot ? (uint32_t)RS_PIVOT : (is_event ? (uint32_t)RS_EVENT : (uint32_t)RS_ADVANCE)) | (Cn >= M ? (uint32_t)RS_LAST : 0u);
            const unsigned long long db = (unsigned long long)__double_as_longlong(dbeta);
            if (DBG) {
                w2_gst(trace + (2 * RS_TRACE + (seq - 1u) % RS_TRACE), wall_clock64());
                w2_gst(trace + (3 * RS_TRACE + (seq - 1u) % RS_TRACE), ncons);
                w2_gst(trace + (0 * RS_TRACE + seq % RS_TRACE), wall_clock64());
            }
            rs_store16(msg + (seq % RS_MSG), rs_u4(rs_msg_word0(kf, ncons, seq, (uint32_t)db, (uint32_t)(db >> 32)), seq, (uint32_t)db, (uint32_t)(db >> 32)));
        }
        const bool round_trip = is_event && !pivot;
        const uint32_t gV = found ? Sx - (qpos + 1u) : 0u;
        if (round_trip) {
            ++nev;
            if (gV && lane == 0) {
                sh.sw[S_GV] = gV;
                w2_lds_done();
                w2_st(sh.sw + S_GREQ, nev);
            }
        }
        const double mq = w2_uni(mq_v), sq = w2_uni(sq_v), gsq = w2_uni(gsq_v), nmq = w2_uni(nmq_v);
        // ---- results of the consumed positions [C, Cn): their numerators are in the ring (the bound test left them there); the event on record ----
        if (found) {
            if (lane == 0) {
                sh.ev_pos[evn % W2_EV] = qpos;
                sh.ev_k[evn % W2_EV] = (uint32_t)q_k;
                sh.ev_bnew[evn % W2_EV] = q_bnew;
            }
            ++evn;
        }
        if (is_event) {
            ++n_events;
            ++n_nnz;
            if (predicted) ++n_pred;
        } else
            ++n_adv;
        // ---- a pivot's corrections: the terms are here ----
        if (pivot) {
            ++n_pivots;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t j = pos_of_slot(sl[i], C);
                const bool hit = son[i] && j > qpos && j < Fs;
                const uint32_t ip = hit ? (pi - spi[i]) & 3u : 0u;
                const double A = (double)sh.wpt[sl[i] * RS_PMAX + ip];
                const double xx = mstd[i] * sq * (A - n_total * (mave[i] * mq));
                dp[i] = hit ? dp[i] + dbeta * xx : dp[i];
            }
            if (Fs < Sx) { // columns behind the event whose dot (streamed before this update) is still on its way
                if (lane == 0) {
                    sh.pf_pos[pf_n] = qpos;
                    sh.pf_msg[pf_n] = seq;
                    sh.pf_pi[pf_n] = pi;
                    sh.pf_val[3 * pf_n] = dbeta;
                    sh.pf_val[3 * pf_n + 1] = mq;
                    sh.pf_val[3 * pf_n + 2] = sq;
                }
                ++pf_n;
            }
        }
        if (predicted) ++pi;
        // ---- the window moves: the slots of the consumed positions take the positions [Sx, Sn) ----
        if (Sn > Sx) {
            if (!wait_seen(mpub_seen, S_MPUB, Sn)) break;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t j = pos_of_slot(sl[i], Cn);
                const bool take = son[i] && j >= Sx && j < Sn;
                take_meta(i, j, take, seq, pi);
                if (take) sh.batch[sl[i]] = seq;
            }
        }
        if (lane == 0) { // this batch's pivots: the first predicted positions of the window [Cn, Sn)
            uint32_t np = 0, p0 = 0xffffffffu;
            if (pivots) {
                const uint32_t e0 = sh.pred[pi % W2_PRED], e1 = sh.pred[(pi + 1u) % W2_PRED], e2 = sh.pred[(pi + 2u) % W2_PRED], e3 = sh.pred[(pi + 3u) % W2_PRED];
                np = (e0 < Sn ? 1u : 0u) + (e1 < Sn ? 1u : 0u) + (e2 < Sn ? 1u : 0u) + (e3 < Sn ? 1u : 0u);
                if (np) p0 = e0;
            }
            sh.bl_pi[seq % W2_NB] = pi;
            sh.bl_np[seq % W2_NB] = np;
            sh.bl_p0[seq % W2_NB] = p0;
        }
        // the generator moves past the consumed positions' uniforms and the draw
        {
            const uint32_t adv = ncons + q_consumed;
            gpos += adv;
            gposr += adv;
            while (gposr >= W2_RING) gposr -= W2_RING;
        }
        w2_lds_done();
        if (lane == 0) {
            w2_st(sh.sw + S_SXPUB, Sn);
            w2_st(sh.sw + S_EVN, evn); // (before the cursor: the housekeeper reads the cursor first)
            w2_st(sh.sw + S_CPUB, Cn);
            w2_st(sh.sw + S_GPOS, gpos);
            w2_st(sh.sw + S_PCUR, pi);
        }
        lap(4);
        // ---- an event that needs the round trip: the collectors' sums, then its corrections ----
        if (round_trip && gV) {
            const uint32_t ncl = nsh < (uint32_t)W2_NCOL ? nsh : (uint32_t)W2_NCOL;
            bool ok = true;
#pragma unroll
            for (int c = 0; c < W2_NCOL; ++c)
                if ((uint32_t)c < ncl && ok) ok = wait_seen(gdone_seen[c], S_GDONE + c, nev);
            if (!ok) break;
            lap(1);
            if (DBG && lane == 0) w2_gst(trace + (1 * RS_TRACE + seq % RS_TRACE), wall_clock64());
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t j = pos_of_slot(sl[i], Cn);
                const bool hit = son[i] && j < Sx; // the old window's positions behind the event
                const uint32_t c = hit ? j - Cn : 0u;
                if constexpr (MISS) {
                    unsigned long long A = sh.gpart64[c];
#pragma unroll
                    for (int w = 1; w < W2_NCOL; ++w) A += (uint32_t)w < ncl ? sh.gpart64[(uint32_t)w * RS_BMAX + c] : 0ull;
                    const double Ad = (double)A * (1.0 / (double)(1ull << RS_GFX));
                    const double both = n_total - nms[i] - nmq; // + X: calls present in both columns
                    const double xx = mstd[i] * sq * (((Ad - mq * gsm[i]) - mave[i] * gsq) + (mave[i] * mq) * both);
                    dp[i] = hit ? dp[i] + dbeta * xx : dp[i];
                } else {
                    uint32_t A = gpart[c];
#pragma unroll
                    for (int w = 1; w < W2_NCOL; ++w) A += (uint32_t)w < ncl ? gpart[(uint32_t)w * RS_BMAX + c] : 0u;
                    const double xx = mstd[i] * sq * ((double)A - n_total * (mave[i] * mq));
                    dp[i] = hit ? dp[i] + dbeta * xx : dp[i];
                }
            }
        } else if (DBG && lane == 0)
            w2_gst(trace + (1 * RS_TRACE + seq % RS_TRACE), wall_clock64());
        C = Cn;
        Sx = Sn;
        base = Cn;
        // the staged list of predicted positions reaches far enough (the housekeeper refills behind S_PCUR)
        if (pivots && !wait_seen(pld_seen, S_PLD, pi + 8u)) break;
    }
    if (C < M) { // the sweep was given up
        ++seq;
        if (lane == 0) {
            w2_st(sh.sw + S_ABORT, 1u);
            rs_store16(msg + (seq % RS_MSG), rs_u4(rs_msg_word0((uint32_t)RS_ABORT, 0u, seq, 0u, 0u), seq, 0u, 0u));
            atomicMax(&state->error, err ? err : 3u);
        }
    }
    if (lane == 0) {
        w2_st(sh.sw + S_GPOS, gpos);
        w2_lds_done();
        w2_st(sh.sw + S_END, 1u);
        state->cursor = C;
        state->rounds = n_rounds;
        state->events = n_events;
        state->advances = n_adv;
        state->nnz = n_nnz;
        state->chunks = n_chunks;
        state->refolds = n_refold;
        state->pivots = n_pivots;
        state->predicted = n_pred;
        if (DBG)
            for (int i = 0; i < 8; ++i) state->t[i] = tacc[i];
    }
}

template <int MISS>
__device__ __attribute__((noinline)) void w2_folder(const ResParams& pr)
{
    W2_PROLOGUE
    // =====================================================================================================================
    // the folder: fixed-point sums of completed refill batches -> dots
    // =====================================================================================================================
    uint32_t Ff = 0, done = 0;
    unsigned long long t_idle = wall_clock64();
    auto poll_done = [&]() {
        uint32_t b = 0xffffffffu;
        if ((uint32_t)lane < rsh) b = __hip_atomic_load(pr.rcnt + (size_t)lane * RS_CROW, HG_RLX_AGENT) / cntR[(uint32_t)lane < W % rsh ? 0 : 1];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const uint32_t o = (uint32_t)__shfl_xor((int)b, off, 64);
            b = o < b ? o : b;
        }
        done = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
        if (lane == 0) w2_st(sh.sw + S_RDONE, done);
    };
    unsigned long long* const racc = pr.racc;
    unsigned long long* const racc2 = pr.racc2;
    unsigned long long* const pacc = pr.pacc;
    for (;;) {
        if (ended() || aborted()) break;
        const uint32_t sx = w2_ld(sh.sw + S_SXPUB);
        if (Ff >= sx) { // nothing announced that has no dot: keep the batch count fresh for the chain's flow control
            poll_done();
            __builtin_amdgcn_s_sleep(2);
            t_idle = wall_clock64();
            continue;
        }
        const uint32_t b0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh.batch[Ff & bmask]);
        if (done <= b0) {
            poll_done();
            if (done <= b0) {
                if (wall_clock64() - t_idle > timeout) {
                    if (lane == 0) w2_st(sh.sw + S_ABORT, 1u);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
                continue;
            }
        }
        // up to 64 positions from Ff on whose batch is complete (batches are in position order: a prefix)
        const uint32_t j = Ff + (uint32_t)lane;
        const uint32_t slot = j & bmask;
        const uint32_t bt = j < sx ? sh.batch[slot] : 0xffffffffu;
        const bool ok = j < sx && bt < done;
        if (ok) {
            const uint32_t rr = j % RS_RB;
            unsigned long long w[RS_RSH], w2[RS_RSH], wp[RS_RSH][2];
            const uint32_t np = sh.bl_np[bt % W2_NB];
            const bool piv = np && sh.bl_p0[bt % W2_NB] < j; // a pivot in front of the column: its terms were sent
#pragma unroll
            for (int s = 0; s < RS_RSH; ++s) w[s] = (uint32_t)s < rsh ? __hip_atomic_load(racc + (size_t)s * RS_RB + rr, HG_RLX_AGENT) : 0ull;
            if constexpr (MISS) {
#pragma unroll
                for (int s = 0; s < RS_RSH; ++s) w2[s] = (uint32_t)s < rsh ? __hip_atomic_load(racc2 + (size_t)s * RS_RB + rr, HG_RLX_AGENT) : 0ull;
            }
            if (piv) {
#pragma unroll
                for (int s = 0; s < RS_RSH; ++s) {
                    const unsigned long long* pw = pacc + ((size_t)((uint32_t)s < rsh ? s : 0) * RS_RB + rr) * 2u;
                    wp[s][0] = (uint32_t)s < rsh ? __hip_atomic_load(pw, HG_RLX_AGENT) : 0ull;
                    wp[s][1] = (uint32_t)s < rsh ? __hip_atomic_load(pw + 1, HG_RLX_AGENT) : 0ull;
                }
            }
            unsigned long long now = 0ull;
#pragma unroll
            for (int s = 0; s < RS_RSH; ++s) now += w[s];
            const unsigned long long tot = now - sh.rprev[rr]; // what this position's batch added (wrapping 64-bit arithmetic)
            sh.rprev[rr] = now;
            const double s1 = (double)(long long)tot * fx_unscale;
            double s2 = eps_sum;
            if constexpr (MISS) { // s2 = sum of eps over the column's calls = sum of eps - R
                unsigned long long now2 = 0ull;
#pragma unroll
                for (int s = 0; s < RS_RSH; ++s) now2 += w2[s];
                const unsigned long long tot2 = now2 - sh.rprev2[rr];
                sh.rprev2[rr] = now2;
                s2 -= (double)(long long)tot2 * fx_unscale;
            }
            const uint32_t ms = j & mrmask;
            sh.dpr[slot] = sh.mstd[ms] * (s1 - sh.mave[ms] * s2); // :1785-1790,1809
            if (piv) {
                unsigned long long n01 = 0ull, n23 = 0ull;
#pragma unroll
                for (int s = 0; s < RS_RSH; ++s) {
                    n01 += wp[s][0];
                    n23 += wp[s][1];
                }
                const unsigned long long d01 = n01 - sh.pprev[rr], d23 = n23 - sh.pprev[RS_RB + rr];
                sh.pprev[rr] = n01;
                sh.pprev[RS_RB + rr] = n23;
                sh.wpt[slot * RS_PMAX + 0] = (uint32_t)d01;
                sh.wpt[slot * RS_PMAX + 1] = (uint32_t)(d01 >> 32);
                sh.wpt[slot * RS_PMAX + 2] = (uint32_t)d23;
                sh.wpt[slot * RS_PMAX + 3] = (uint32_t)(d23 >> 32);
            }
        }
        const uint32_t n = (uint32_t)__popcll(__ballot(ok));
        w2_lds_done();
        Ff += n;
        if (lane == 0) w2_st(sh.sw + S_FPUB, Ff);
        t_idle = wall_clock64();
    }
}

template <int MISS>
__device__ __attribute__((noinline)) void w2_housekeeper(const ResParams& pr)
{
    W2_PROLOGUE
    // =====================================================================================================================
    // the housekeeper: generator blocks, metadata, predicted positions, results
    // =====================================================================================================================
    uint32_t nblk = 1, mpub = m0, wpub = 0, evw = 0, pld = W2_PRED;
    double* const acum = pr.acum;
    double* const beta = pr.beta;
    int32_t* const comp = pr.comp;
    for (;;) {
        const bool fin = ended();
        if (aborted()) break;
        bool worked = false;
        // generator: block nblk goes to ring slot nblk mod W2_NBLK, which held block nblk - W2_NBLK; the block of the last word
        // consumed stays (the sweep hands it back)
        if (!fin) {
            const uint32_t gp = w2_ld(sh.sw + S_GPOS);
            if (nblk < (gp ? (gp - 1u) / (uint32_t)MT_N : 0u) + (uint32_t)W2_NBLK) {
                const W2_LDS uint32_t* cur = sh.mt + ((nblk - 1u) % W2_NBLK) * MT_N;
                W2_LDS uint32_t* nxt = sh.mt + (nblk % W2_NBLK) * MT_N;
                W2_LDS double* tqn = sh.tq + (nblk % W2_NBLK) * MT_N;
                for (int i = lane; i < 227; i += WAVE) nxt[i] = mt_mix(cur[i], cur[i + 1], cur[i + MT_M]);
                w2_lds_done();
                for (int i = 227 + lane; i < 454; i += WAVE) nxt[i] = mt_mix(cur[i], cur[i + 1], nxt[i - 227]);
                w2_lds_done();
                for (int i = 454 + lane; i < 623; i += WAVE) nxt[i] = mt_mix(cur[i], cur[i + 1], nxt[i - 227]);
                w2_lds_done();
                if (lane == 0) nxt[623] = mt_mix(cur[623], nxt[0], nxt[396]);
                w2_lds_done();
                for (int i = lane; i < MT_N; i += WAVE) tqn[i] = tq_of(nxt[i]);
                w2_lds_done();
                ++nblk;
                if (lane == 0) w2_st(sh.sw + S_BLK, nblk);
                worked = true;
            }
        }
        // results of consumed positions (before their metadata slots are given away): the numerator of every marker goes out as it
        // stands (Acum's slot holds it until the sweep is over: k_res_finish turns it into Acum, :1892,:1899-1905); an event also
        // writes its new effect and its component (flagged: k_res_finish clears the flag)
        {
            const uint32_t c = w2_ld(sh.sw + S_CPUB);
            if (wpub < c) {
                const uint32_t evn = w2_ld(sh.sw + S_EVN);
                const uint32_t hi = c - wpub > (uint32_t)WAVE ? wpub + (uint32_t)WAVE : c;
                const uint32_t j = wpub + (uint32_t)lane;
                if (j < hi) acum[sh.marker[j & mrmask]] = sh.rnum[j & (W2_RR - 1u)];
                // the events among them (in position order on the record)
                for (;;) {
                    const uint32_t e = evw + (uint32_t)lane;
                    const bool mine = e < evn && sh.ev_pos[e % W2_EV] < hi;
                    if (mine) {
                        const uint32_t ms = sh.ev_pos[e % W2_EV] & mrmask;
                        const int marker = sh.marker[ms];
                        const uint32_t k = sh.ev_k[e % W2_EV];
                        beta[marker] = sh.ev_bnew[e % W2_EV];
                        comp[marker] = (int)k | RS_EVENT_FLAG;
                        __hip_atomic_fetch_add(sh.lcass + ((sh.ga[ms] & 0x0fffffff) * K + (int)k), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    const uint32_t n = (uint32_t)__popcll(__ballot(mine));
                    evw += n;
                    if (n < (uint32_t)WAVE) break;
                }
                w2_lds_done();
                wpub = hi;
                if (lane == 0) {
                    w2_st(sh.sw + S_WPUB, wpub);
                    w2_st(sh.sw + S_EVW, evw);
                }
                worked = true;
            }
        }
        // metadata: two windows ahead of the results written
        if (!fin && mpub < M && mpub < wpub + MR) {
            const uint32_t hi = wpub + MR < M ? wpub + MR : M;
            const uint32_t j = mpub + (uint32_t)lane;
            if (j < hi) stage_meta(j);
            const uint32_t n = hi - mpub < (uint32_t)WAVE ? hi - mpub : (uint32_t)WAVE;
            w2_lds_done();
            mpub += n;
            if (lane == 0) w2_st(sh.sw + S_MPUB, mpub);
            worked = true;
        }
        // predicted positions: the ring keeps the indices [pld - W2_PRED, pld); an entry goes once it is 512 behind the chain's index
        if (!fin && pivots) {
            const uint32_t pc = w2_ld(sh.sw + S_PCUR);
            if (pld < pc + 512u) {
                const uint32_t i = pld + (uint32_t)lane;
                sh.pred[i % W2_PRED] = i < M + 16u ? gpred[i] : 0xffffffffu;
                w2_lds_done();
                pld += (uint32_t)WAVE;
                if (lane == 0) w2_st(sh.sw + S_PLD, pld);
                worked = true;
            }
        }
        if (fin && wpub >= w2_ld(sh.sw + S_CPUB)) break;
        if (!worked) __builtin_amdgcn_s_sleep(2);
    }
}

template <int MISS>
__device__ __attribute__((noinline)) void w2_collector(const ResParams& pr)
{
    W2_PROLOGUE
    // =====================================================================================================================
    // the collectors: Gram terms of the window columns behind an event, rows c and c + 4 of the accumulator shards; wave s polls its
    // rows, 16 bytes per lane; every word in use must carry the shard's full arrival count.  The words only ever grow (no store ever
    // touches them: adds are performed at the memory side, and a store could overtake or be overtaken by one): what an event added is
    // the difference to what the lane saw last time, per parity
    // =====================================================================================================================
    const uint32_t c = (uint32_t)wave - 4u;
    uint32_t done_ev = 0;
    u4_t gp0[2] = {rs_u4(0u, 0u, 0u, 0u), rs_u4(0u, 0u, 0u, 0u)}, gp1[2] = {rs_u4(0u, 0u, 0u, 0u), rs_u4(0u, 0u, 0u, 0u)}; // [row] by parity 0 / 1
    unsigned long long gq0[2][4] = {{0ull, 0ull, 0ull, 0ull}, {0ull, 0ull, 0ull, 0ull}}, gq1[2][4] = {{0ull, 0ull, 0ull, 0ull}, {0ull, 0ull, 0ull, 0ull}};
    const uint32_t* const gacc = pr.gacc;
    const unsigned long long* const gacc64 = pr.gacc64;
    for (;;) {
        if (ended() || aborted()) break;
        const uint32_t req = w2_ld(sh.sw + S_GREQ);
        if (req == done_ev) {
            __builtin_amdgcn_s_sleep(1);
            continue;
        }
        const uint32_t gV = w2_ld(sh.sw + S_GV);
        const uint32_t par = (req - 1u) & 1u;
        const bool mine = 4u * (uint32_t)lane < gV;
        const unsigned long long t0 = wall_clock64();
        bool fail = false;
        if constexpr (MISS) {
            unsigned long long acc[4] = {0ull, 0ull, 0ull, 0ull};
#pragma unroll
            for (int rw = 0; rw < 2; ++rw) {
                const uint32_t row = c + 4u * (uint32_t)rw;
                if (row < nsh && mine) { // (uniform in row; lanes beyond the columns have nothing to wait for)
                    const unsigned long long* rp = gacc64 + ((size_t)par * RS_NSH + row) * RS_GROW + 4u * (uint32_t)lane;
                    const unsigned long long want = cntG[row < W % nsh ? 0 : 1];
                    unsigned long long v[4], d[4];
                    for (;;) {
                        const u4_t a = rs_load16(rp), b = rs_load16(rp + 2);
                        v[0] = ((unsigned long long)a.y << 32) | a.x;
                        v[1] = ((unsigned long long)a.w << 32) | a.z;
                        v[2] = ((unsigned long long)b.y << 32) | b.x;
                        v[3] = ((unsigned long long)b.w << 32) | b.z;
                        bool ok = true;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            d[i] = v[i] - (par ? gq1[rw][i] : gq0[rw][i]);
                            ok = ok && (4u * (uint32_t)lane + (uint32_t)i >= gV || (d[i] >> 56) == want);
                        }
                        if (ok) break;
                        if (wall_clock64() - t0 > timeout || aborted()) {
                            fail = true;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(1);
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (par) gq1[rw][i] = v[i];
                        else gq0[rw][i] = v[i];
                        acc[i] += d[i] & (RS_ONE64 - 1ull);
                    }
                }
            }
            if (mine) {
                W2_LDS unsigned long long* g64 = sh.gpart64 + c * RS_BMAX + 4u * (uint32_t)lane;
#pragma unroll
                for (int i = 0; i < 4; ++i) g64[i] = acc[i];
            }
        } else {
            uint32_t acc[4] = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int rw = 0; rw < 2; ++rw) {
                const uint32_t row = c + 4u * (uint32_t)rw;
                if (row < nsh && mine) {
                    const uint32_t* rp = gacc + ((size_t)par * RS_NSH + row) * RS_GROW + 4u * (uint32_t)lane;
                    const uint32_t want = cntG[row < W % nsh ? 0 : 1];
                    u4_t v, d;
                    for (;;) {
                        v = rs_load16(rp);
                        d = par ? (v - gp1[rw]) : (v - gp0[rw]);
                        const uint32_t i = 4u * (uint32_t)lane;
                        const bool ok = (d.x >> 24) == want && (i + 1u >= gV || (d.y >> 24) == want) && (i + 2u >= gV || (d.z >> 24) == want) &&
                                        (i + 3u >= gV || (d.w >> 24) == want);
                        if (ok) break;
                        if (wall_clock64() - t0 > timeout || aborted()) {
                            fail = true;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(1);
                    }
                    // (a word beyond gV got no add this time: its difference is zero and its previous value stays what it is)
                    if (par) gp1[rw] = v;
                    else gp0[rw] = v;
                    acc[0] += d.x & RS_LOW;
                    acc[1] += d.y & RS_LOW;
                    acc[2] += d.z & RS_LOW;
                    acc[3] += d.w & RS_LOW;
                }
            }
            if (mine) {
                W2_LDS uint32_t* gp = gpart + c * RS_BMAX + 4u * (uint32_t)lane;
                gp[0] = acc[0];
                gp[1] = acc[1];
                gp[2] = acc[2];
                gp[3] = acc[3];
            }
        }
        if (__ballot(fail) != 0ull) {
            if (lane == 0) w2_st(sh.sw + S_ABORT, 1u);
            break;
        }
        w2_lds_done();
        done_ev = req;
        if (lane == 0) w2_st(sh.sw + S_GDONE + (int)c, req);
    }
}

template <int DBG, int MISS>
__device__ __attribute__((noinline)) void res_walker2(const ResParams& pr)
{
    W2_PROLOGUE
    const unsigned long long clk0 = __builtin_amdgcn_s_memtime(), wall0 = wall_clock64();
    // ---- staging, all eight waves ----
    for (int i = tid; i < MT_N; i += RS_BLOCK) sh.mt[i] = pr.mt[i];
    for (int i = tid; i < 129; i += RS_BLOCK) {
        sh.zig_nx[i] = pr.zig.nx[i];
        sh.zig_ny[i] = pr.zig.ny[i];
    }
    for (int i = tid; i < 4 * GK; i += RS_BLOCK) sh.htab[(i / GK) * HT_LDS + (i % GK)] = pr.denom[i];
    for (int i = tid; i < 256; i += RS_BLOCK) sh.lcass[i] = 0;
    for (int i = tid; i < GK; i += RS_BLOCK) {
        const int g0 = (i / K) * K;
        sh.qtab[i] = (pr.logpi[i] - pr.hlog[i]) - pr.logpi[g0];
        sh.qtab[HT_LDS + i] = (i % K) ? i_2sigE / pr.denom[i] : 0.0;
    }
    for (int i = tid; i < RS_RB; i += RS_BLOCK) {
        sh.rprev[i] = 0ull;
        sh.rprev2[i] = 0ull;
        sh.pprev[i] = 0ull;
        sh.pprev[RS_RB + i] = 0ull;
    }
    if (tid < S_NWORDS) sh.sw[tid] = 0u;
    // metadata of the first two windows, the predicted positions, the first batch
    for (uint32_t j = (uint32_t)tid; j < m0; j += RS_BLOCK) stage_meta(j);
    for (uint32_t i = (uint32_t)tid; i < W2_PRED; i += RS_BLOCK) sh.pred[i] = i < M + 16u ? gpred[i] : 0xffffffffu;
    for (uint32_t j = (uint32_t)tid; j < Sx0; j += RS_BLOCK) sh.batch[j & bmask] = 0u;
    __syncthreads();
    // thresholds of the first block; the tabulated bound (see res_walker: f(n2) = log sum_l>0 exp(c_l + n2 r_l) is convex in n2 = num^2,
    // so the chord between two grid points is an upper bound)
    for (int i = tid; i < MT_N; i += RS_BLOCK) sh.tq[i] = tq_of(sh.mt[i]);
    {
        const int G = GK / K;
        if (tid < G) {
            double rmin = 1e300, cmin = 1e300;
            for (int l = 1; l < K; ++l) {
                rmin = fmin(rmin, sh.qtab[HT_LDS + tid * K + l]);
                cmin = fmin(cmin, sh.qtab[tid * K + l]);
            }
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
    if (tid == 0) {
        sh.sw[S_MPUB] = m0;
        sh.sw[S_SXPUB] = Sx0;
        sh.sw[S_BLK] = 1u;
        sh.sw[S_GPOS] = rng_idx0;
        sh.sw[S_PLD] = W2_PRED;
        // batch 0: the window [0, Sx0) as the streaming workgroups fill it at entry; its pivots are the first predicted positions
        uint32_t np = 0;
        for (int ip = 0; ip < RS_PMAX; ++ip) np += (pivots && sh.pred[ip] < Sx0) ? 1u : 0u;
        sh.bl_pi[0] = 0u;
        sh.bl_np[0] = np;
        sh.bl_p0[0] = np ? sh.pred[0] : 0xffffffffu;
    }
    __syncthreads();

    if (wave == 0) w2_chain<DBG, MISS>(pr);
    else if (wave == 1) w2_folder<MISS>(pr);
    else if (wave == 2) w2_housekeeper<MISS>(pr);
    else if (wave >= 4 && (uint32_t)wave - 4u < nsh) w2_collector<MISS>(pr);
    __syncthreads();
    // ---- the generator and the counters go back ----
    {
        const uint32_t gp = sh.sw[S_GPOS];
        uint32_t b = gp / (uint32_t)MT_N, idx = gp % (uint32_t)MT_N;
        if (idx == 0u && b > 0u) { // at a block's end: the block just used up goes back, exhausted (the next call regenerates -- as the host's generator does)
            b -= 1u;
            idx = (uint32_t)MT_N;
        }
        const W2_LDS uint32_t* blkp = sh.mt + (b % W2_NBLK) * MT_N;
        for (int i = tid; i < MT_N; i += RS_BLOCK) pr.mt[i] = blkp[i];
        for (int i = tid; i < GK; i += RS_BLOCK) pr.cass[i] = sh.lcass[i];
        if (tid == 0) {
            state->rng_idx = idx;
            state->shader_ticks = __builtin_amdgcn_s_memtime() - clk0;
            state->wall_ticks = wall_clock64() - wall0;
        }
    }
}

} // namespace hg

