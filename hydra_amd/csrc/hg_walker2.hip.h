// hg_walker2.hip.h -- the resident engine's walker, second form: ONE wave walks the chain, the others serve it.
//
// The first walker (res_walker, hg_resident.hip.h) runs the whole workgroup through every step of a round in lockstep: the bound
// test, the exact decision, the message, then results, metadata prefetch and the fold of arrived raw dots -- three dependent trips
// to global memory and a dozen workgroup barriers per round, all of them between one message and the next.  Where an event needs no
// round trip through the streaming workgroups (a predicted pivot: its Gram terms came with the columns), that serial work IS the
// round.  Here the roles are split over the workgroup's eight waves and nothing on the chain ever waits for global memory:
//
//   waves 0-3   the CHAIN: one window position per lane (wave w holds the window slots 64 w .. 64 w + 63), in registers the dot as
//               streamed, its Gram corrections, mave, mstd and the old effect.  A wave issues one instruction every five clocks or
//               so: what is parallel over the window -- absorbing dots that arrived, the Gram corrections behind an event, the
//               bound test without an exponential (the tabulated convex bound) -- runs on four waves on four SIMDs, one position per
//               lane; what is serial -- the first candidate, the reference's exact decision and draw (a5-a7,
//               src/BayesRRm.cpp:1744-1753,1859-1921), the message -- on wave 0, the DECIDER.  It tells the others what happened by a
//               RECORD in LDS (event, correction terms, how the window moves, how far dots may be absorbed), they answer with the
//               candidate masks of their slots.  When an event needs the round trip through the streaming workgroups, every chain
//               lane polls the Gram accumulators' words of ITS slot (they are indexed by window slot) in all shard rows itself.
//   wave 4      the FOLDER: polls the batch counters, turns the fixed-point sums of completed refill batches into dots (and the
//               columns' Gram terms with their batch's pivots) and publishes "every position below F has its dot".
//   wave 5      the HOUSEKEEPER: generator blocks and their thresholds ahead of the chain (a ring of four), marker metadata two
//               windows ahead, the list of predicted positions, and the results of consumed positions (numerator into Acum's slot,
//               new effect and component of an event) out to global memory.
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
constexpr int W2_NCH = 4;                    // chain waves (0 .. 3); the folder is wave 4, the housekeeper wave 5
constexpr uint32_t W2_NREC = 8;              // records in the ring (the decider waits for every answer before the next one: two would do)

enum {
    S_FPUB = 0, // folder -> decider: every position below has its dot
    S_MPUB,     // housekeeper -> chain: positions below have their metadata staged
    S_CPUB,     // decider -> housekeeper: cursor (positions below are consumed, their numerators are in the results ring)
    S_SEQPUB,   // decider -> folder: messages posted (the batches on record, bl_*)
    S_RECSEQ,   // decider -> chain: records published
    S_BLK,      // housekeeper -> decider: generator blocks made
    S_GPOS,     // decider -> housekeeper: generator words consumed (absolute, from the start of the block the sweep began in)
    S_PLD,      // housekeeper -> decider: predicted positions staged (index)
    S_PCUR,     // decider -> housekeeper: index of the first predicted position at or behind the cursor
    S_RDONE,    // folder -> decider: refill batches every streaming workgroup has completed
    S_EVN,      // decider -> housekeeper: events recorded
    S_ABORT,
    S_END,
    S_ERR,
    S_WPUB,     // housekeeper -> decider: results of the positions below are written (their ring entries and metadata slots are free)
    S_EVW,      // housekeeper -> decider: event records taken
    S_NWORDS = 32
};
// a record: what the decider tells the chain waves (words of 4 bytes)
enum {
    R_TYPE = 0, // RT_*
    R_Q,        // position of the event
    R_CN,       // the cursor behind it
    R_SXO,      // the window's end before
    R_SXN,      //   ... and after: the slots of the consumed positions take the positions [R_SXO, R_SXN)
    R_FD,       // dots may be absorbed, and positions tested, below this position
    R_SEQ,      // the message's number = the refill batch of the new positions
    R_PIN,      // index of that batch's first pivot in the list of predicted positions
    R_PIQ,      // RT_PIVOT: the event's index in that list
    R_NEV,      // RT_EVENT: its number among the events that take the round trip (parity of the Gram accumulators)
    R_GPOSR,    // the generator's ring index at the new cursor
    R_PFN,      // fired pivots on record (pf_*)
    R_DB = 12,  // (two words each) dbeta, then the event's mave, mstd, and (build MISS) its genotype sum and missing calls
    R_MQ = 14,
    R_SQ = 16,
    R_GSQ = 18,
    R_NMQ = 20,
    R_WORDS = 32
};
enum { RT_EXTEND = 0, RT_ADVANCE = 1, RT_PIVOT = 2, RT_EVENT = 3, RT_END = 4 };

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
    X(unsigned long long, rloc, Bz)       /* several ranks: this rank's part of the slot's raw dot (pushed to the peers; the dot needs theirs) */ \
    X(unsigned long long, rloc2, Bz)      /*   (build MISS) the same for R */                                                                 \
    X(uint32_t, rpushed, Bz)              /*   ... has been taken and pushed */                                                               \
    X(double, pf_val, (size_t)RS_PFIRE * 3) /* fired pivots: (dbeta, mave, mstd) */                                                           \
    X(uint32_t, ans, (size_t)W2_NCH * 4)  /* the chain waves' answers: {candidate mask (two words), records answered, -}, 16 bytes each */        \
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
    X(uint32_t, bl_sn, (size_t)W2_NB)     /*   the window's end behind its message: the batch's positions end there */                        \
    X(uint32_t, rec, (size_t)W2_NREC * R_WORDS) /* the decider's records */                                                                   \
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
    // The address of the dynamic LDS is not a constant inside a function that is not the kernel: the compiler reads it from a table in global
    // memory -- and, short of registers, reads it AGAIN wherever an array's address is needed (the chain's loop held a dozen such loads, each
    // a round trip to L2 with a wait for everything in flight in front of an LDS access).  Read once, kept as an opaque scalar.
    uint32_t lb = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(W2_LDS unsigned char*)hg_smem);
    asm volatile("" : "+s"(lb));
    W2_LDS unsigned char* const lbase = (W2_LDS unsigned char*)(uintptr_t)lb;
#define W2_X(type, name, count) sh.name = (W2_LDS type*)(lbase + off.name);
    W2_ARRAYS(W2_X)
#undef W2_X
    return sh;
}

// what every role's function starts with: the parameters it reads again and again, once (they live in global memory: the memory
// clobbers of the hand-over waits would otherwise make every use a scalar load of its own), the LDS arrays, the small helpers.
// The roles are functions of their own so that each gets a register allocation of its own (the chain's four slots per lane want most
// of the register file).
#define W2_BEXPR pr.B
#define W2_PROLOGUE \
    const int tid = threadIdx.x, lane = tid & 63; \
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6); \
    const uint32_t B = W2_BEXPR, bmask = B - 1u, M = pr.M; \
    const uint32_t MR = 2u * B, mrmask = MR - 1u; \
    const int K = pr.K, GK = pr.GK; \
    const uint32_t W = pr.W, nsh = pr.nsh, rsh = pr.rsh; \
    const double n_total = pr.n_total, n_minus_1 = pr.n_minus_1, i_2sigE = pr.i_2sigE, eps_sum = pr.eps_sum, fx_unscale = pr.fx_unscale; \
    const unsigned long long timeout = pr.timeout; \
    const bool pivots = pr.pivots != 0; \
    const uint32_t early_advance = (uint32_t)pr.early_advance; \
    const bool announce = pr.announce != 0 && pr.pivots == 0; \
    const uint32_t rng_idx0 = pr.rng_idx; \
    ResMsg* const msg = pr.msg; \
    ResState* const state = pr.state; \
    unsigned long long* const trace = pr.trace; \
    unsigned long long* const progress = pr.progress; \
    const uint32_t* const gpred = pr.pred; \
    const double* const zig_ex = pr.zig.ex; \
    const double* const zig_ey = pr.zig.ey; \
    const Walk2Lds sh = walk2_lds(B); \
    const uint32_t cntG[2] = {W / nsh + (W % nsh ? 1u : 0u), W / nsh}; \
    const uint32_t cntR[2] = {W / rsh + (W % rsh ? 1u : 0u), W / rsh}; \
    const uint32_t wend_mask = pr.wend_mask; \
    const int nranks = pr.nranks, rank = pr.rank; \
    const unsigned long long sweep_id = pr.sweep_id; \
    const uint32_t Sx0 = rs_window_end(0u, B, M, wend_mask); \
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

// (BC: the window as a compile-time constant -- 256, the size that is used where speed matters -- or 0: whatever the sweep's parameters say.
// With the constant the LDS arrays are at constant addresses and the slot arithmetic is immediate: the wave's few scalar registers are not
// spent on three dozen array bases, and a wave issues one instruction every four to five clocks whatever it is.)
template <int DBG, int MISS, int BC, int RANKS> // (RANKS: several ranks -- the exchange through the mailboxes is compiled in)
__device__ __attribute__((noinline)) void w2_chain(const ResParams& pr)
{
#undef W2_BEXPR
#define W2_BEXPR (BC ? (uint32_t)BC : pr.B)
    W2_PROLOGUE
#undef W2_BEXPR
#define W2_BEXPR pr.B
    // =====================================================================================================================
    // the chain: wave w holds the window slots 64 w + lane, one position per lane; wave 0 also decides
    // =====================================================================================================================
    __builtin_amdgcn_s_setprio(3);
    const bool decider = wave == 0;
    const uint32_t nch = B >= 64u ? B / 64u : 1u; // chain waves in use
    const uint32_t slot = (uint32_t)wave * 64u + (uint32_t)lane;
    const bool son = slot < B;
    const uint32_t sl = son ? slot : 0u;
    // the slot's position: the dot as streamed, its Gram corrections, the marker's (mave, mstd, old effect x (N - 1)), the tabulated
    // bound's scale and table offset of its group, its refill batch and that batch's first pivot index
    double dpr = 0.0, dp = 0.0, mave = 0.0, mstd = 0.0, boldn = 0.0, fsc = 0.0, gsm = 0.0, nms = 0.0;
    uint32_t fof = 0, sbt = 0, spi = 0;
    bool prd = false; // the marker's effect is non-zero at sweep start (a predicted event)
    uint32_t C = 0, Sx = 0, Fs = 0, base = 0, gposr = rng_idx0 % W2_RING; // (every chain wave follows them through the records)
    // the Gram accumulators' words of this slot as last seen, per shard row and parity (they only ever grow: what an event added is
    // the difference; no store ever touches them -- adds are performed at the memory side, and a store could overtake or be overtaken)
    // (two arrays with static accesses only -- both values read, one selected -- so that they stay in registers)
    uint32_t gpa32[RS_NSH], gpb32[RS_NSH];
    unsigned long long gpa64[RS_NSH], gpb64[RS_NSH];
#pragma unroll
    for (int r = 0; r < RS_NSH; ++r) {
        gpa32[r] = gpb32[r] = 0u;
        gpa64[r] = gpb64[r] = 0ull;
    }
    // (global address space, said so: through a generic pointer the polls are FLAT instructions, which the LDS counter counts as well)
    typedef const __attribute__((address_space(1))) uint32_t* w2_g32;
    typedef const __attribute__((address_space(1))) unsigned long long* w2_g64;
    const w2_g32 gacc = (w2_g32)pr.gacc;
    const w2_g64 gacc64 = (w2_g64)pr.gacc64;
    uint32_t rno = 0; // records taken

    // ---- the decider's own state (wave 0; uniform) ----
    uint32_t seq = 0, nev = 0, pi = 0, evn = 0, pf_n = 0, gpos = rng_idx0, blk = 1u;
    uint32_t rdone_seen = 0, wpub_seen = 0, evw_seen = 0, pld_seen = W2_PRED, fpub_seen = 0, mpub_seen = m0;
    unsigned long long cm[W2_NCH] = {0ull, 0ull, 0ull, 0ull}; // candidates among the tested positions [C, Fs), by chain wave
    uint32_t n_rounds = 0, n_events = 0, n_adv = 0, n_nnz = 0, n_chunks = 0, n_refold = 0, n_pivots = 0, n_pred = 0, n_ann = 0;
    uint32_t err = 0;
    bool failed = false;
    unsigned long long tacc[8] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull};
    unsigned long long tmark = DBG ? wall_clock64() : 0ull;
    auto lap = [&](int i) {
        if (DBG && decider) {
            const unsigned long long now = wall_clock64();
            tacc[i] += now - tmark;
            tmark = now;
        }
    };
    auto pos_of_slot = [&](uint32_t s, uint32_t c) { return c + ((s - c) & bmask); };
    // inside a wait: has the sweep been given up, or is it time to give it up
    uint32_t nspin = 0;
    auto spin_fail = [&](unsigned long long t0) {
        if (aborted()) return true;
        if (wall_clock64() - t0 > timeout || ((++nspin & 1023u) == 0u && __hip_atomic_load(progress + 2, HG_RLX_AGENT) != 0ull)) { // (or the host gave the sweep up)
            if (lane == 0) w2_st(sh.sw + S_ABORT, 1u);
            return true;
        }
        return false;
    };
    // wait until a counter that only grows has reached the value, with what this wave saw of it last: no LDS round trip while that is enough
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
    // a record goes out (decider, lane 0 writes; every field is uniform)
    auto publish = [&](uint32_t type, uint32_t q, uint32_t Cn, uint32_t Sxo, uint32_t Sxn, uint32_t Fd, uint32_t piq, double db, double mq, double sq, double gsq, double nmq) {
        if (lane == 0) {
            W2_LDS uint32_t* r = sh.rec + (rno % W2_NREC) * R_WORDS;
            r[R_TYPE] = type;
            r[R_Q] = q;
            r[R_CN] = Cn;
            r[R_SXO] = Sxo;
            r[R_SXN] = Sxn;
            r[R_FD] = Fd;
            r[R_SEQ] = seq;
            r[R_PIN] = pi;
            r[R_PIQ] = piq;
            r[R_NEV] = nev;
            r[R_GPOSR] = gpos % W2_RING;
            r[R_PFN] = pf_n;
            *(W2_LDS double*)(r + R_DB) = db;
            *(W2_LDS double*)(r + R_MQ) = mq;
            *(W2_LDS double*)(r + R_SQ) = sq;
            *(W2_LDS double*)(r + R_GSQ) = gsq;
            *(W2_LDS double*)(r + R_NMQ) = nmq;
            w2_lds_done();
            w2_st(sh.sw + S_RECSEQ, rno + 1u);
        }
    };

    if (decider) publish(RT_ADVANCE, 0u, 0u, 0u, Sx0, 0u, 0u, 0.0, 0.0, 0.0, 0.0, 0.0); // record 0: the first window

    for (;;) {
        // =================================================================================================================
        // every chain wave: take the record, bring the slot up to date, test it, answer with the candidate mask
        // =================================================================================================================
        if (!decider) {
            uint32_t seen = 0;
            if (!wait_seen(seen, S_RECSEQ, rno + 1u)) break;
        }
        const W2_LDS uint32_t* rc = sh.rec + (rno % W2_NREC) * R_WORDS;
        const uint32_t r_type = w2_uni(rc[R_TYPE]);
        if (r_type == (uint32_t)RT_END) break;
        const uint32_t r_q = w2_uni(rc[R_Q]), r_cn = w2_uni(rc[R_CN]), r_sxo = w2_uni(rc[R_SXO]), r_sxn = w2_uni(rc[R_SXN]), r_fd = w2_uni(rc[R_FD]);
        const uint32_t r_pfn = w2_uni(rc[R_PFN]);
        if (r_type == (uint32_t)RT_PIVOT) {
            // a predicted pivot: its Gram term with this column came with the column's dot.  Columns whose dot was here before this record
            // take the correction now; the others find the pivot on the list of fired ones when their dot is absorbed (below, or later)
            const uint32_t jo = pos_of_slot(sl, C);
            const double db = *(const W2_LDS double*)(rc + R_DB), mq = *(const W2_LDS double*)(rc + R_MQ), sq = *(const W2_LDS double*)(rc + R_SQ);
            const bool hit = son && jo > r_q && jo < Fs;
            const uint32_t piq = rc[R_PIQ];
            const double A = (double)sh.wpt[sl * RS_PMAX + (hit ? (piq - spi) & 3u : 0u)];
            const double xx = mstd * sq * (A - n_total * (mave * mq));
            dp = hit ? dp + db * xx : dp;
        }
        // the dots that have arrived: positions [Fs, r_fd) -- with the corrections of the pivots that fired between a column's streaming
        // and now (their updates were not in the streamed dot; the terms came with it)
        {
            const uint32_t j = pos_of_slot(sl, C);
            const bool arr = son && j >= Fs && j < r_fd;
            const double dnew = sh.dpr[sl];
            dpr = arr ? dnew : dpr;
            for (uint32_t f = 0; f < r_pfn; ++f) { // (uniform, rarely any)
                const uint32_t fmsg = sh.pf_msg[f], fpos = sh.pf_pos[f], fpi = sh.pf_pi[f];
                const double fdb = sh.pf_val[3 * f], fmv = sh.pf_val[3 * f + 1], fsd = sh.pf_val[3 * f + 2];
                const bool hit = arr && fmsg > sbt && fpos < j;
                const double A = (double)sh.wpt[sl * RS_PMAX + (hit ? (fpi - spi) & 3u : 0u)];
                const double xx = mstd * fsd * (A - n_total * (mave * fmv));
                dp = hit ? dp + fdb * xx : dp;
            }
            Fs = r_fd > Fs ? r_fd : Fs;
        }
        if (r_type != (uint32_t)RT_EXTEND) {
            const uint32_t jo = pos_of_slot(sl, C); // the slot's position in the window as it was
            const double db = *(const W2_LDS double*)(rc + R_DB), mq = *(const W2_LDS double*)(rc + R_MQ), sq = *(const W2_LDS double*)(rc + R_SQ);
            if (r_type == (uint32_t)RT_EVENT) {
                // an event that took the round trip: the streaming workgroups' Gram terms of this slot's column, all shard rows; every word
                // must carry its shard's full arrival count
                const uint32_t par = (w2_uni(rc[R_NEV]) - 1u) & 1u;
                const bool hit = son && jo > r_q && jo < r_sxo;
                const unsigned long long t0 = wall_clock64();
                bool bad = false;
                if constexpr (MISS) {
                    const w2_g64 wp = gacc64 + (size_t)par * RS_NSH * RS_GROW + sl;
                    unsigned long long A = 0ull;
                    if (hit) {
                        unsigned long long v[RS_NSH];
                        for (;;) {
#pragma unroll
                            for (int r = 0; r < RS_NSH; ++r) v[r] = (uint32_t)r < nsh ? __hip_atomic_load(wp + (size_t)r * RS_GROW, HG_RLX_AGENT) : 0ull;
                            bool ok = true;
#pragma unroll
                            for (int r = 0; r < RS_NSH; ++r) {
                                const unsigned long long pa = gpa64[r], pb = gpb64[r];
                                const unsigned long long d = v[r] - (par ? pb : pa);
                                ok = ok && ((uint32_t)r >= nsh || (d >> 56) == cntG[(uint32_t)r < W % nsh ? 0 : 1]);
                            }
                            if (ok) break;
                            if (wall_clock64() - t0 > timeout || aborted()) {
                                bad = true;
                                break;
                            }
                            __builtin_amdgcn_s_sleep(1);
                        }
#pragma unroll
                        for (int r = 0; r < RS_NSH; ++r) {
                            const unsigned long long pa = gpa64[r], pb = gpb64[r];
                            const unsigned long long d = v[r] - (par ? pb : pa);
                            A += d & (RS_ONE64 - 1ull);
                            gpa64[r] = par ? pa : v[r];
                            gpb64[r] = par ? v[r] : pb;
                        }
                    }
                    if (RANKS && hit && !bad) {
                        // several ranks: the 56-bit sums cross as two words each, tag << 32 | half (tag = 1 | sweep | event: self-validating), by window slot
                        const unsigned long long tag = (0x80000000ull | ((sweep_id & 0x7full) << 24) | (unsigned long long)(w2_uni(rc[R_NEV]) & 0xffffffu)) << 32;
                        for (int r = 0; r < nranks; ++r)
                            if (r != rank) {
                                unsigned long long* w = rx_gbox(pr.mbox[r], rank, par) + 2u * sl;
                                __hip_atomic_store(w, tag | (A & 0xffffffffull), HG_RLX_SYSTEM);
                                __hip_atomic_store(w + 1, tag | (A >> 32), HG_RLX_SYSTEM);
                            }
                        for (;;) {
                            bool all = true;
                            unsigned long long add = 0ull;
                            for (int r = 0; r < nranks; ++r)
                                if (r != rank) {
                                    const unsigned long long* w = rx_gbox(pr.mbox[rank], r, par) + 2u * sl;
                                    const unsigned long long o0 = __hip_atomic_load(w, HG_RLX_SYSTEM), o1 = __hip_atomic_load(w + 1, HG_RLX_SYSTEM);
                                    all = all && (o0 >> 32) == (tag >> 32) && (o1 >> 32) == (tag >> 32);
                                    add += (o1 << 32) | (o0 & 0xffffffffull);
                                }
                            if (all) {
                                A += add;
                                break;
                            }
                            if (wall_clock64() - t0 > timeout || aborted()) {
                                bad = true;
                                break;
                            }
                            __builtin_amdgcn_s_sleep(1);
                        }
                    }
                    const double gsq = *(const W2_LDS double*)(rc + R_GSQ), nmq = *(const W2_LDS double*)(rc + R_NMQ);
                    const double Ad = (double)A * (1.0 / (double)(1ull << RS_GFX));
                    const double both = n_total - nms - nmq; // + X: calls present in both columns
                    const double xx = mstd * sq * (((Ad - mq * gsm) - mave * gsq) + (mave * mq) * both);
                    dp = hit ? dp + db * xx : dp;
                } else {
                    const w2_g32 wp = gacc + (size_t)par * RS_NSH * RS_GROW + sl;
                    uint32_t A = 0u;
                    if (hit) {
                        uint32_t v[RS_NSH];
                        for (;;) {
#pragma unroll
                            for (int r = 0; r < RS_NSH; ++r) v[r] = (uint32_t)r < nsh ? __hip_atomic_load(wp + (size_t)r * RS_GROW, HG_RLX_AGENT) : 0u;
                            bool ok = true;
#pragma unroll
                            for (int r = 0; r < RS_NSH; ++r) {
                                const uint32_t pa = gpa32[r], pb = gpb32[r];
                                const uint32_t d = v[r] - (par ? pb : pa);
                                ok = ok && ((uint32_t)r >= nsh || (d >> 24) == cntG[(uint32_t)r < W % nsh ? 0 : 1]);
                            }
                            if (ok) break;
                            if (wall_clock64() - t0 > timeout || aborted()) {
                                bad = true;
                                break;
                            }
                            __builtin_amdgcn_s_sleep(1);
                        }
#pragma unroll
                        for (int r = 0; r < RS_NSH; ++r) {
                            const uint32_t pa = gpa32[r], pb = gpb32[r];
                            const uint32_t d = v[r] - (par ? pb : pa);
                            A += d & RS_LOW;
                            gpa32[r] = par ? pa : v[r];
                            gpb32[r] = par ? v[r] : pb;
                        }
                    }
                    if (RANKS && hit && !bad) {
                        // several ranks: this rank's sum goes to every peer's mailbox, the peers' arrive in mine -- sweep << 48 | event << 24 | sum, one
                        // 8-byte store, self-validating; by window slot
                        const unsigned long long tag = (sweep_id << 48) | ((unsigned long long)(w2_uni(rc[R_NEV]) & 0xffffffu) << 24);
                        for (int r = 0; r < nranks; ++r)
                            if (r != rank) __hip_atomic_store(rx_gbox(pr.mbox[r], rank, par) + 2u * sl, tag | (unsigned long long)A, HG_RLX_SYSTEM);
                        for (;;) {
                            bool all = true;
                            uint32_t add = 0u;
                            for (int r = 0; r < nranks; ++r)
                                if (r != rank) {
                                    const unsigned long long o = __hip_atomic_load(rx_gbox(pr.mbox[rank], r, par) + 2u * sl, HG_RLX_SYSTEM);
                                    all = all && (o >> 24) == (tag >> 24);
                                    add += (uint32_t)o & RS_LOW;
                                }
                            if (all) {
                                A += add;
                                break;
                            }
                            if (wall_clock64() - t0 > timeout || aborted()) {
                                bad = true;
                                break;
                            }
                            __builtin_amdgcn_s_sleep(1);
                        }
                    }
                    const double xx = mstd * sq * ((double)A - n_total * (mave * mq));
                    dp = hit ? dp + db * xx : dp;
                }
                if (__ballot(bad) != 0ull) {
                    if (lane == 0) w2_st(sh.sw + S_ABORT, 1u);
                    break;
                }
                if (DBG && decider && lane == 0) w2_gst(trace + (1 * RS_TRACE + w2_uni(rc[R_SEQ]) % RS_TRACE), wall_clock64());
            }
            // the window moves: the slots of the consumed positions take the positions [r_sxo, r_sxn)
            if (r_sxn > r_sxo) {
                uint32_t seen = mpub_seen;
                if (!wait_seen(seen, S_MPUB, r_sxn)) break;
                mpub_seen = seen;
                const uint32_t jn = pos_of_slot(sl, r_cn);
                const bool take = son && jn >= r_sxo && jn < r_sxn;
                const uint32_t ms = (take ? jn : 0u) & mrmask;
                const double a = sh.mave[ms], d = sh.mstd[ms], b = sh.bold[ms];
                const int g = sh.ga[ms] & 0x0fffffff;
                const double sc = sh.fscale[g];
                const uint32_t bno = rc[R_SEQ], bpi = rc[R_PIN];
                mave = take ? a : mave;
                mstd = take ? d : mstd;
                boldn = take ? b * n_minus_1 : boldn;
                prd = take ? b != 0.0 : prd;
                fsc = take ? sc : fsc;
                fof = take ? (uint32_t)g * (uint32_t)(RS_FN + 1) : fof;
                dp = take ? 0.0 : dp;
                dpr = take ? 0.0 : dpr;
                sbt = take ? bno : sbt;
                spi = take ? bpi : spi;
                if constexpr (MISS) {
                    const double gs = sh.gsum[ms], nm = sh.nmis[ms];
                    gsm = take ? gs : gsm;
                    nms = take ? nm : nms;
                }
                if (take) sh.batch[sl] = bno;
            }
            C = r_cn;
            Sx = r_sxn;
            base = r_cn;
            gposr = w2_uni(rc[R_GPOSR]);
        }
        // ---- the bound test of the positions [base, Fs): "this marker cannot be an event" (a candidate else); the numerator goes to the
        // results ring as it stands -- final once the position is consumed ----
        unsigned long long mymask;
        {
            const uint32_t j = pos_of_slot(sl, C);
            const bool act = son && j >= base && j < Fs;
            const double num = (dpr + dp) + boldn;
            if (act) sh.rnum[j & (W2_RR - 1u)] = num;
            const double x = (num * num) * fsc;
            const bool inside = x < (double)RS_FN && fsc > 0.0; // (NaN: outside)
            const int kx = inside ? (int)x : 0;
            uint32_t r = gposr + ((j - C) & bmask);
            r = r >= W2_RING ? r - W2_RING : r;
            const double f0 = sh.ftab[fof + (uint32_t)kx], f1 = sh.ftab[fof + (uint32_t)kx + 1u], tqv = sh.tq[r];
            const double fup = f0 + (x - (double)kx) * (f1 - f0);
            mymask = __ballot(act && (prd || !(inside && fup <= tqv)));
        }
        base = Fs;
        if (!decider) {
            if (lane == 0) *(W2_LDS u4_t*)(sh.ans + 4 * wave) = rs_u4((uint32_t)mymask, (uint32_t)(mymask >> 32), rno + 1u, 0u); // (one 16-byte store: mask and number together)
            ++rno;
            continue;
        }
        ++rno;
        lap(5);
        // =================================================================================================================
        // the decider: the other waves' answers, the first candidate in window order, the exact decision, the next record
        // =================================================================================================================
        if (r_type != (uint32_t)RT_EXTEND) cm[0] = cm[1] = cm[2] = cm[3] = 0ull;
        cm[0] |= mymask;
        if (nch > 1u) { // the other waves' answers: all three entries in one round trip, again until each carries this record's number
            const unsigned long long t0 = wall_clock64();
            for (;;) {
                const u4_t a1 = *(const volatile W2_LDS u4_t*)(sh.ans + 4), a2 = *(const volatile W2_LDS u4_t*)(sh.ans + 8), a3 = *(const volatile W2_LDS u4_t*)(sh.ans + 12);
                const bool ok = w2_uni(a1.z) == rno && (nch < 3u || (w2_uni(a2.z) == rno && w2_uni(a3.z) == rno));
                if (ok) {
                    cm[1] |= ((unsigned long long)w2_uni(a1.y) << 32) | w2_uni(a1.x);
                    if (nch >= 3u) {
                        cm[2] |= ((unsigned long long)w2_uni(a2.y) << 32) | w2_uni(a2.x);
                        cm[3] |= ((unsigned long long)w2_uni(a3.y) << 32) | w2_uni(a3.x);
                    }
                    break;
                }
                if (spin_fail(t0)) {
                    failed = true;
                    break;
                }
            }
        }
        if (failed) break;
        lap(1);
        // the counters the message's flow control will ask for leave LDS now, all at once, and are looked at behind the exact decision: read
        // one by one where they are needed they were three trips to LDS in front of every message (their cached values go stale every round)
        const uint32_t fc_rdone = __hip_atomic_load(sh.sw + S_RDONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP),
                       fc_wpub = __hip_atomic_load(sh.sw + S_WPUB, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP),
                       fc_evw = __hip_atomic_load(sh.sw + S_EVW, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (lane == 0 && (DBG || (n_chunks & 255u) == 0u)) w2_gst(progress, ((unsigned long long)n_rounds << 8) | 1u);
        // fired pivots stay on record only while a column streamed before their update is without its dot (entries are in message order).
        // (The chain waves read the list while they take a record; it changes only here, behind their answers.)
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
        // the generator: the words a decision can reach exist
        if (gpos + B + 96u > blk * (uint32_t)MT_N && !wait_seen(blk, S_BLK, (gpos + B + 96u + (uint32_t)MT_N - 1u) / (uint32_t)MT_N)) break;
        ++n_chunks;
        bool found = false, announced = false;
        uint32_t qpos = 0, q_consumed = 0;
        int q_k = 0;
        double q_bnew = 0.0, q_bold = 0.0, q_mq = 0.0, q_sq = 0.0, q_gsq = 0.0, q_nmq = 0.0;
        {
            const uint32_t s0 = C & bmask, i0 = s0 >> 6, l0 = s0 & 63u;
            const unsigned long long lowm = (1ull << l0) - 1ull;
            for (;;) {
                uint32_t qs = 0xffffffffu;
#pragma unroll
                for (int t = 0; t <= W2_NCH; ++t) {
                    if ((uint32_t)t <= nch && qs == 0xffffffffu) {
                        uint32_t i = i0 + (uint32_t)t;
                        i = i >= nch ? i - nch : i;
                        unsigned long long mm = i == 0u ? cm[0] : (i == 1u ? cm[1] : (i == 2u ? cm[2] : cm[3]));
                        if (t == 0) mm &= ~lowm;
                        if ((uint32_t)t == nch) mm &= lowm;
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
                const double bold_v = sh.bold[ms], num = sh.rnum[qc & (W2_RR - 1u)];
                const double cmq_v = sh.mave[ms], csq_v = sh.mstd[ms], cgsq_v = MISS ? sh.gsum[ms] : 0.0, cnmq_v = MISS ? sh.nmis[ms] : 0.0; // (for the record, if this one is the event)
                const int ga_v = sh.ga[ms];
                const uint32_t word_v = sh.mt[upos];
                const double bold = w2_uni(bold_v);
                // a marker whose effect is non-zero WILL be an event (whatever is drawn changes the effect, and nothing in front of it is one):
                // the streaming workgroups are asked for its Gram terms now -- they travel while the draw is made (RS_ANNOUNCE; the slot
                // is free: flow control as for the message itself, and never waited for here)
                if (announce && !announced && bold != 0.0 && qc + 1u < Sx && (seq + 6u <= (uint32_t)RS_MSG || rdone_seen >= seq + 6u - (uint32_t)RS_MSG)) {
                    if (lane == 0) rs_store16(msg + ((seq + 1u) % RS_MSG), rs_u4(rs_msg_word0((uint32_t)RS_ANNOUNCE, qc - C + 1u, seq + 1u, 0u, 0u), seq + 1u, 0u, 0u));
                    announced = true;
                }
                const int g0 = w2_uni(ga_v & 0x0fffffff) * K;
                const double prob = (double)mt_temper(word_v) * (1.0 / 4294967296.0);
                // a5 (src/BayesRRm.cpp:1859-1921) over the lanes: lane x < K holds logL_x; lane 8 kk + l the term exp(logL_l - logL_kk)
                // (lane x also keeps num / denom_x and the draw's standard deviation of component x: the draw below takes the chosen component's
                // from its lane -- the same quotient, not a second division and a second trip to LDS behind the walk)
                double Lm = 0.0, mk = 0.0, sdl = 0.0;
                {
                    const int lk = lane < K ? lane : 0;
                    const double den = sh.htab[g0 + lk], lpi = sh.htab[HT_LDS + g0 + lk], hlg = sh.htab[2 * HT_LDS + g0 + lk];
                    sdl = sh.htab[3 * HT_LDS + g0 + lk];
                    mk = num / (lk ? den : 1.0);
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
                    const double mean_k = rs_readlane(mk, k), sd_k = rs_readlane(sdl, k);
                    if (lane == 0) {
                        RingGen g{sh.mt, upos + 1u == W2_RING ? 0u : upos + 1u, blk * (uint32_t)MT_N - (gpos + (qc - C) + 1u), 0u, 0u};
                        const ZigLds zt{sh.zig_nx, sh.zig_ny, zig_ex, zig_ey};
                        bnew = norm_rng_sd(g, zt, mean_k, sd_k);
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
                    q_mq = w2_uni(cmq_v);
                    q_sq = w2_uni(csq_v);
                    q_gsq = w2_uni(cgsq_v);
                    q_nmq = w2_uni(cnmq_v);
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
        }
        lap(3);
        if (failed) break;
        // (every position with a dot has passed and the walk needs dots that are still on their way: if enough positions have passed, the window
        // moves on NOW -- a message that only advances -- so that the streaming workgroups refill behind it while the walk waits)
        const uint32_t early_thr = early_advance;
        const bool early = !found && Fs < Sx && early_thr != 0u && Fs - C >= early_thr;
        if (!found && Fs < Sx && !early) {
            // every position with a dot has passed: the walk needs dots that are still on their way
            ++n_refold;
            if (!wait_seen(fpub_seen, S_FPUB, Fs + 1u)) break;
            lap(0);
            publish(RT_EXTEND, 0u, C, Sx, Sx, fpub_seen < Sx ? fpub_seen : Sx, 0u, 0.0, 0.0, 0.0, 0.0, 0.0);
            continue;
        }
        // ---- the message: an event at qpos, or a round that only moves the window on ----
        const uint32_t ncons = found ? qpos - C + 1u : (early ? Fs - C : Sx - C);
        const uint32_t Cn = C + ncons;
        const uint32_t Sn = rs_window_end(Cn, B, M, wend_mask);
        const double dbeta = found ? q_bold - q_bnew : 0.0;
        const bool changed = found && dbeta != 0.0;
        const bool is_event = changed || announced; // (announced and drawn the very same effect again: an event that changes nothing -- the Gram terms have been asked for)
        const bool predicted = found && q_bold != 0.0;
        // a predicted pivot whose Gram terms came with the columns: the oldest batch with columns behind it lists it
        bool pivot = false;
        if (pivots && is_event && predicted && pf_n < (uint32_t)RS_PFIRE)
            pivot = qpos + 1u >= Sx || pi - w2_uni(sh.bl_pi[w2_uni(sh.batch[(qpos + 1u) & bmask]) % W2_NB]) < (uint32_t)RS_PMAX;
        // flow control: never more than RS_MSG - 3 messages ahead of the slowest streaming workgroup (batches completed = messages
        // taken + 1); room in the results ring and on the event record
        lap(6);
        {
            const uint32_t a = w2_uni(fc_rdone), b = w2_uni(fc_wpub), c = w2_uni(fc_evw); // (they only grow)
            rdone_seen = a > rdone_seen ? a : rdone_seen;
            wpub_seen = b > wpub_seen ? b : wpub_seen;
            evw_seen = c > evw_seen ? c : evw_seen;
        }
        if (seq + 6u > (uint32_t)RS_MSG && !wait_seen(rdone_seen, S_RDONE, seq + 6u - (uint32_t)RS_MSG)) break;
        if (Cn + B > W2_RR && !wait_seen(wpub_seen, S_WPUB, Cn + B - W2_RR)) break;
        if (evn + 2u > W2_EV && !wait_seen(evw_seen, S_EVW, evn + 2u - W2_EV)) break;
        lap(2);
        ++seq;
        ++n_rounds;
        if (lane == 0) {
            const uint32_t kf = (pivot ? (uint32_t)RS_PIVOT : (announced ? (uint32_t)RS_ANNOUNCED : (is_event ? (uint32_t)RS_EVENT : (uint32_t)RS_ADVANCE))) | (Cn >= M ? (uint32_t)RS_LAST : 0u);
            const unsigned long long db = (unsigned long long)__double_as_longlong(dbeta);
            if (DBG) {
                w2_gst(trace + (2 * RS_TRACE + (seq - 1u) % RS_TRACE), wall_clock64());
                w2_gst(trace + (3 * RS_TRACE + (seq - 1u) % RS_TRACE), ncons);
                w2_gst(trace + (0 * RS_TRACE + seq % RS_TRACE), wall_clock64());
                if (!(is_event && !pivot)) w2_gst(trace + (1 * RS_TRACE + seq % RS_TRACE), wall_clock64());
            }
            rs_store16(msg + (seq % RS_MSG), rs_u4(rs_msg_word0(kf, ncons, seq, (uint32_t)db, (uint32_t)(db >> 32)), seq, (uint32_t)db, (uint32_t)(db >> 32)));
        }
        const bool round_trip = is_event && !pivot;
        if (round_trip) ++nev;
        // (behind the message: what only the record needs)
        const double mq = q_mq, sq = q_sq, gsq = q_gsq, nmq = q_nmq; // (came with the candidate's other values; a round without an event records zeros: nobody reads them)
        // the event on record (the consumed positions' numerators are in the results ring: the bound test left them there)
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
            if (changed) ++n_nnz;
            if (predicted) ++n_pred;
            if (announced) ++n_ann;
        } else
            ++n_adv;
        const uint32_t piq = pi;
        // dots that are here by now may be absorbed with this record
        fpub_seen = w2_ld(sh.sw + S_FPUB);
        const uint32_t Fd = fpub_seen < Sx ? fpub_seen : Sx;
        if (pivot) {
            ++n_pivots;
            if (Fs < Sx) { // columns behind the event whose dot (streamed before this update) has not been absorbed: the correction waits for it
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
            sh.bl_sn[seq % W2_NB] = Sn;
        }
        // the generator moves past the consumed positions' uniforms and the draw
        gpos += ncons + q_consumed;
        w2_lds_done();
        if (lane == 0) {
            w2_st(sh.sw + S_SEQPUB, seq);
            w2_st(sh.sw + S_EVN, evn); // (before the cursor: the housekeeper reads the cursor first)
            w2_st(sh.sw + S_CPUB, Cn);
            w2_st(sh.sw + S_GPOS, gpos);
            w2_st(sh.sw + S_PCUR, pi);
        }
        if (Cn >= M) { // the sweep is over
            C = Cn;
            break;
        }
        // the record: what happened, for every chain wave (this one included)
        publish(found ? (pivot ? (uint32_t)RT_PIVOT : (is_event ? (uint32_t)RT_EVENT : (uint32_t)RT_ADVANCE)) : (uint32_t)RT_ADVANCE, found ? qpos : Cn - 1u, Cn, Sx, Sn, Fd, piq, dbeta,
                mq, sq, gsq, nmq);
        lap(4);
        // the staged list of predicted positions reaches far enough (the housekeeper refills behind S_PCUR)
        if (pivots && !wait_seen(pld_seen, S_PLD, pi + 8u)) break;
    }
    if (decider) {
        if (C < M) { // the sweep was given up
            ++seq;
            if (lane == 0) {
                w2_st(sh.sw + S_ABORT, 1u);
                rs_store16(msg + (seq % RS_MSG), rs_u4(rs_msg_word0((uint32_t)RS_ABORT, 0u, seq, 0u, 0u), seq, 0u, 0u));
                atomicMax(&state->error, err ? err : 3u);
            }
        }
        publish(RT_END, 0u, C, Sx, Sx, 0u, 0u, 0.0, 0.0, 0.0, 0.0, 0.0);
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
            state->pivots = n_pivots + n_ann; // (events whose Gram terms the walker did not have to wait a whole round trip for: predicted pivots, or announced)
            state->predicted = n_pred;
            if (DBG)
                for (int i = 0; i < 8; ++i) state->t[i] = tacc[i];
        }
    }
}

template <int MISS, int RANKS>
__device__ __attribute__((noinline)) void w2_folder(const ResParams& pr)
{
    W2_PROLOGUE
    // =====================================================================================================================
    // the folder: fixed-point sums of completed refill batches -> dots
    // =====================================================================================================================
    uint32_t Ff = 0, done = 0, fb = 0; // fb: the refill batch of position Ff (batch b = the positions [bl_sn[b - 1], bl_sn[b]) announced by message b)
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
        const uint32_t sq = w2_ld(sh.sw + S_SEQPUB);                 // messages posted
        const uint32_t sx = w2_uni(sh.bl_sn[sq % W2_NB]);             // positions announced
        while (fb < sq && Ff >= w2_uni(sh.bl_sn[fb % W2_NB])) ++fb;   // (batches without positions of their own are skipped)
        if (Ff >= sx) { // nothing announced that has no dot: keep the batch count fresh for the decider's flow control
            poll_done();
            __builtin_amdgcn_s_sleep(2);
            t_idle = wall_clock64();
            continue;
        }
        // The raw-dot words say themselves when they are complete (every workgroup's arrival in the top byte): the announced positions are
        // read as soon as they are announced, and what is complete is folded -- no wait for the batch counters (they lag the adds by the
        // streaming workgroups' drain and one more trip; they still feed the decider's flow control).  With predicted pivots the pivot words
        // have no count of their own: there the batch counter gates as before.
        poll_done();
        if (pivots && done <= fb) {
            if (wall_clock64() - t_idle > timeout) {
                if (lane == 0) w2_st(sh.sw + S_ABORT, 1u);
                break;
            }
            __builtin_amdgcn_s_sleep(1);
            continue;
        }
        // up to 128 positions from Ff on, two to a lane, all their loads in flight together: one round trip for a refill batch of the usual size
        bool okv[2], pivv[2];
        uint32_t jv[2], btv[2];
        unsigned long long w[2][RS_RSH], w2[2][RS_RSH], wp[2][RS_RSH][2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t j = Ff + (uint32_t)(64 * h + lane);
            jv[h] = j;
            uint32_t bt = fb;
            while (bt < sq && j >= sh.bl_sn[bt % W2_NB]) ++bt; // (per lane: the positions may span a few batches)
            btv[h] = bt;
            okv[h] = j < sx && (!pivots || bt < done) && (h == 0 || B >= 128u); // (two positions of a pass never share a window slot)
            const uint32_t np = sh.bl_np[bt % W2_NB];
            pivv[h] = okv[h] && np && sh.bl_p0[bt % W2_NB] < j; // a pivot in front of the column: its terms were sent
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t rr = jv[h] % RS_RB;
#pragma unroll
            for (int s = 0; s < RS_RSH; ++s) w[h][s] = (okv[h] && (uint32_t)s < rsh) ? __hip_atomic_load(racc + (size_t)s * RS_RB + rr, HG_RLX_AGENT) : 0ull;
            if constexpr (MISS) {
#pragma unroll
                for (int s = 0; s < RS_RSH; ++s) w2[h][s] = (okv[h] && (uint32_t)s < rsh) ? __hip_atomic_load(racc2 + (size_t)s * RS_RB + rr, HG_RLX_AGENT) : 0ull;
            }
            if (pivv[h]) {
#pragma unroll
                for (int s = 0; s < RS_RSH; ++s) {
                    const unsigned long long* pw = pacc + ((size_t)((uint32_t)s < rsh ? s : 0) * RS_RB + rr) * 2u;
                    wp[h][s][0] = (uint32_t)s < rsh ? __hip_atomic_load(pw, HG_RLX_AGENT) : 0ull;
                    wp[h][s][1] = (uint32_t)s < rsh ? __hip_atomic_load(pw + 1, HG_RLX_AGENT) : 0ull;
                }
            }
        }
        // which of them are complete: the difference to what was seen last carries W arrivals (both words in build MISS); positions are
        // folded in order, so only the complete PREFIX counts (a later position that happens to be complete waits for its turn)
        unsigned long long nowv[2], now2v[2], totv[2], tot2v[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t rr = jv[h] % RS_RB;
            nowv[h] = now2v[h] = 0ull;
#pragma unroll
            for (int s = 0; s < RS_RSH; ++s) nowv[h] += w[h][s];
            const unsigned long long d = nowv[h] - sh.rprev[rr];
            bool full = rs_raw_count(d) == (W & 0xffu);
            unsigned long long d2 = 0ull;
            if constexpr (MISS) {
#pragma unroll
                for (int s = 0; s < RS_RSH; ++s) now2v[h] += w2[h][s];
                d2 = now2v[h] - sh.rprev2[rr];
                full = full && rs_raw_count(d2) == (W & 0xffu);
            }
            totv[h] = rs_raw_value(d);
            tot2v[h] = rs_raw_value(d2);
            if constexpr (RANKS != 0) {
                // several ranks: this rank's part is taken once (the words are "seen" from then on), kept by window slot and pushed to every peer
                // -- two tagged halves per sum, tag = 1 | sweep | the position's refill batch --; the dot needs the peers' parts from this
                // rank's own mailbox, each word validating itself
                const uint32_t slot = jv[h] & bmask;
                const unsigned long long tag = rx_rtag(sweep_id, btv[h]);
                if (okv[h] && !sh.rpushed[slot] && full) {
                    sh.rprev[rr] = nowv[h];
                    sh.rloc[slot] = totv[h];
                    if constexpr (MISS) {
                        sh.rprev2[rr] = now2v[h];
                        sh.rloc2[slot] = tot2v[h];
                    }
                    for (int r = 0; r < nranks; ++r)
                        if (r != rank) {
                            unsigned long long* pw = rx_rbox(pr.mbox[r], rank) + 4u * (jv[h] % RX_RING);
                            __hip_atomic_store(pw, tag | (totv[h] & 0xffffffffull), HG_RLX_SYSTEM);
                            __hip_atomic_store(pw + 1, tag | (totv[h] >> 32), HG_RLX_SYSTEM);
                            if constexpr (MISS) {
                                __hip_atomic_store(pw + 2, tag | (tot2v[h] & 0xffffffffull), HG_RLX_SYSTEM);
                                __hip_atomic_store(pw + 3, tag | (tot2v[h] >> 32), HG_RLX_SYSTEM);
                            }
                        }
                    sh.rpushed[slot] = 1u;
                }
                full = okv[h] && sh.rpushed[slot] != 0u;
                if (full) {
                    unsigned long long t1 = sh.rloc[slot], t2 = MISS ? sh.rloc2[slot] : 0ull;
                    for (int r = 0; r < nranks; ++r)
                        if (r != rank) {
                            const unsigned long long* pw = rx_rbox(pr.mbox[rank], r) + 4u * (jv[h] % RX_RING);
                            const unsigned long long o0 = __hip_atomic_load(pw, HG_RLX_SYSTEM), o1 = __hip_atomic_load(pw + 1, HG_RLX_SYSTEM);
                            full = full && (o0 >> 32) == (tag >> 32) && (o1 >> 32) == (tag >> 32);
                            t1 += (o1 << 32) | (o0 & 0xffffffffull);
                            if constexpr (MISS) {
                                const unsigned long long o2 = __hip_atomic_load(pw + 2, HG_RLX_SYSTEM), o3 = __hip_atomic_load(pw + 3, HG_RLX_SYSTEM);
                                full = full && (o2 >> 32) == (tag >> 32) && (o3 >> 32) == (tag >> 32);
                                t2 += (o3 << 32) | (o2 & 0xffffffffull);
                            }
                        }
                    totv[h] = t1;
                    tot2v[h] = t2;
                }
            }
            okv[h] = okv[h] && full;
        }
        {
            const unsigned long long b0 = __ballot(okv[0]);
            const uint32_t n0p = b0 == ~0ull ? (uint32_t)WAVE : (uint32_t)(__ffsll((long long)~b0) - 1); // leading complete positions of the first 64
            okv[0] = okv[0] && (uint32_t)lane < n0p;
            const unsigned long long b1 = n0p == (uint32_t)WAVE ? __ballot(okv[1]) : 0ull;
            const uint32_t n1p = b1 == ~0ull ? (uint32_t)WAVE : (uint32_t)(__ffsll((long long)~b1) - 1);
            okv[1] = okv[1] && n0p == (uint32_t)WAVE && (uint32_t)lane < n1p;
            pivv[0] = pivv[0] && okv[0];
            pivv[1] = pivv[1] && okv[1];
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (okv[h]) {
                const uint32_t j = jv[h], slot = j & bmask, rr = j % RS_RB;
                const unsigned long long tot = totv[h]; // what this position's batch added (wrapping 64-bit arithmetic, the arrivals taken off; several ranks: all of them)
                if (RANKS) sh.rpushed[slot] = 0u; // (the slot's next position starts afresh)
                else sh.rprev[rr] = nowv[h];
                double s1 = (double)(long long)tot * fx_unscale;
                double s2 = eps_sum;
                if constexpr (MISS) { // s2 = sum of eps over the column's calls = sum of eps - R; the streamed dot weighs a missing call 3: s1' = s1 + 3 R
                    const unsigned long long tot2 = tot2v[h];
                    if (!RANKS) sh.rprev2[rr] = now2v[h];
                    s1 = (double)(long long)(tot - 3ull * tot2) * fx_unscale; // (exact: integers)
                    s2 -= (double)(long long)tot2 * fx_unscale;
                }
                const uint32_t ms = j & mrmask;
                sh.dpr[slot] = sh.mstd[ms] * (s1 - sh.mave[ms] * s2); // :1785-1790,1809
                if (pivv[h]) {
                    unsigned long long n01 = 0ull, n23 = 0ull;
#pragma unroll
                    for (int s = 0; s < RS_RSH; ++s) {
                        n01 += wp[h][s][0];
                        n23 += wp[h][s][1];
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
        }
        const uint32_t n0 = (uint32_t)__popcll(__ballot(okv[0]));
        const uint32_t n = n0 + (n0 == (uint32_t)WAVE ? (uint32_t)__popcll(__ballot(okv[1])) : 0u);
        if (n == 0u) { // (the first announced position is still short of an arrival)
            if (wall_clock64() - t_idle > timeout) {
                if (lane == 0) w2_st(sh.sw + S_ABORT, 1u);
                break;
            }
            __builtin_amdgcn_s_sleep(1);
            continue;
        }
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
    for (uint32_t i = (uint32_t)tid; i < B; i += RS_BLOCK) sh.rpushed[i] = 0u;
    if (tid < S_NWORDS) sh.sw[tid] = 0u;
    if (tid < W2_NCH * 4) sh.ans[tid] = 0u;
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
        sh.bl_sn[0] = Sx0;
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

    if (wave < W2_NCH) {
        if ((uint32_t)wave < (B >= 64u ? B / 64u : 1u)) {
            if (pr.nranks > 1) w2_chain<DBG, MISS, 0, 1>(pr);
            else if (B == 256u) w2_chain<DBG, MISS, 256, 0>(pr);
            else w2_chain<DBG, MISS, 0, 0>(pr);
        }
    } else if (wave == 4) {
        if (pr.nranks > 1) w2_folder<MISS, 1>(pr);
        else w2_folder<MISS, 0>(pr);
    }
    else if (wave == 5) w2_housekeeper<MISS>(pr);
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

