// hgibbs.hip -- device operators behind include/hgibbs.h (layer 1 of the C ABI).
// Written for gfx950 only.  See hg_kernels.h for the data layout and kernels.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/hgibbs.h"
#include "hg_kernels.h"
#include "hg_sweep.hip.h"
#include "hg_resident.hip.h"

using namespace hg;

// ---------------------------------------------------------------------------
// error plumbing (fail-stop, like check_mpi/check_malloc src/mpi_utils.hpp:19-36)
// ---------------------------------------------------------------------------
static thread_local std::string g_err;

static int fail(const char* fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return 1;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) return fail("%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e_)); \
    } while (0)

#define NCCL_TRY(expr)                                                                          \
    do {                                                                                        \
        ncclResult_t r_ = (expr);                                                               \
        if (r_ != ncclSuccess) return fail("%s:%d %s -> %s", __FILE__, __LINE__, #expr, ncclGetErrorString(r_)); \
    } while (0)

extern "C" void hgibbs_set_error_(const char* msg) { g_err = msg; }

// ---------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------
struct BwState; // BayesW side of the handle (hg_bayesw.hip.h)
static void bw_free(BwState* b);

struct hgibbs_ctx {
    int device = 0;
    int num_cu = 256;
    BwState* bw = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;

    // sharding
    int nranks = 1, rank = 0;
    ncclComm_t comm = nullptr;
    hgibbs_allreduce_fn ext_fn = nullptr; // bulk reductions through the caller's transport
    void* ext_user = nullptr;
    // in-launch exchange: own mailbox + IPC-mapped peers
    unsigned char* mbox = nullptr;
    void* peer_base[MAX_RANKS] = {};
    bool p2p_ready = false, p2p_enabled = true;
    uint64_t batch_seq = 0;

    // data
    uint32_t n_global = 0, n_local = 0, n_pad = 0, M = 0, row_begin = 0;
    uint64_t stride = 0;
    uint8_t* bed = nullptr;
    double* eps[2] = {nullptr, nullptr};
    uint32_t eps_cur = 0;
    double *mave = nullptr, *mstd = nullptr;
    unsigned long long* counts = nullptr; // 3*M: n1, n2, nmiss (global after all-reduce)
    bool have_stats = false;
    bool any_missing = false; // some column has missing calls
    double missing_col_frac = 0.0; // fraction of columns with missing calls (decides which build of the sweep kernel runs)
    int gram_missing = -1; // carry columns with missing calls through the extension: -1 auto, 0 never, 1 whenever possible

    // covariates: C columns of n_pad doubles in the permuted eps layout
    double* covX = nullptr;
    int C = 0;

    // model / effects
    int G = 0, K = 0;
    int32_t* groups = nullptr;
    std::vector<int32_t> groups_host;
    std::vector<double> cVa, cVaI;
    double* beta = nullptr;
    int32_t* comp = nullptr;
    double* acum = nullptr;

    // sweep scratch
    int32_t* order = nullptr;
    uint8_t* adaV = nullptr;
    double *s_mave = nullptr, *s_mstd = nullptr, *s_bold = nullptr;
    int32_t* s_ga = nullptr;
    uint32_t* pred = nullptr;     // resident engine: the sweep positions whose marker has a non-zero effect at sweep start, ascending, then 16 sentinels
    uint32_t* pred_cnt = nullptr; // per 4096 positions: how many of them (k_pred_*)
    unsigned long long* dbg = nullptr; // 8 words, device
    bool debug_timing = false;
    bool w_kernel_timing = false; // BayesW: HIP events around every k_bw_sums launch (bench.py's roofline leg)
    int32_t* cass = nullptr;
    double* tables = nullptr; // 4 * G*K
    uint32_t* mt = nullptr;
    double* zig = nullptr; // 129+129+257+257
    SweepDesc* desc = nullptr;
    SweepDesc* desc_host = nullptr; // pinned
    double* partials = nullptr;
    double* totals = nullptr;
    double* carry = nullptr; // MAX_BATCH dots handed from one launch to the next
    double* ahead_raw = nullptr; // the ahead phase's buffers (hg_kernels.h: SweepParams)
    double* apartials = nullptr;
    uint32_t* aticket = nullptr; // AHEAD_MAX / 2 group tickets, then the two queue counters
    int ahead = -1;          // option ahead: columns streamed ahead per launch (-1 = auto)
    int carry_on = -1;       // option carry: -1 auto (on for shards of 400 000 individuals and more), 0 off, 1 on
    uint32_t* ticket = nullptr; // word 0: the launch-wide ticket; words 16.. : one per column group
    double* sums = nullptr;    // 3*MAX_BATCH+1 (multi-GPU exchange buffer)
    double* scratch = nullptr; // reductions
    size_t scratch_n = 0;
    double* scratch_host = nullptr; // pinned, 4096 doubles
    double* beta_host = nullptr;    // pinned, M doubles (lazy)

    // options
    uint32_t batch = 0; // 0 = auto: 256 for shards of >= 20k individuals or several ranks, else 128
    uint32_t cols_per_group = 4; // measured (N = 50k / 200k / 500k): 4 beats 8 by 10 % / 7 % / 2 % (more, lighter workgroups: four waves per SIMD)
    int chunk = 0; // launches per host check (0 = adaptive)
    uint32_t slices = 0; // gridDim.x of the sweep (0 = auto)
    uint32_t ext_limit = 256;
    uint32_t max_seg = 0; // segments (predicted events) one launch chains through; 0 = auto: 4 (the wider kernel tier) for shards of up to
                          // 300k individuals (400k when every launch also pays a cross-GPU exchange), where fewer, longer launches win
                          // (N = 200k: 4.77 vs 4.61 M markers/s, 250k: 3.75 vs 3.59 M, 350k: equal, 500k: 3.05 vs 3.11 M), else 2
    bool gram = true; // Gram-corrected continuation past the first predicted event
    bool use_graph = false; // replay the sweep's launches from a captured graph
    bool force_split = false; // run dots -> all-reduce -> draw as separate launches even on one rank

    // resident engine (hg_resident.hip.h): one launch per sweep, individuals sharded over the compute units
    int engine = 0;           // option engine: 0 auto (resident where it applies), 1 batch engine (k_sweep_batch), 2 resident (refused where it does not apply)
    bool res_all_ada = false;            // this sweep's adaV has no zero
    unsigned long long res_sweep_id = 0; // resident sweeps so far: the epoch of the cross-rank mailbox flags
    bool engine_pinned = false; // an option of the batch engine was set while engine = 0: auto means the batch engine then
    uint32_t window = 0;      // option window: columns kept in LDS per streaming workgroup (0 auto; power of two <= 256)
    uint32_t res_cus = 0;     // option res_cus: compute units the resident engine may use (0 = all)
    int res_walker = 0;       // option walker: 0 auto (the second where it applies), 1 the first walker, 2 the second (hg_walker2.hip.h; refused where it does not apply)
    bool res_attr_set[20] = {}; // the resident kernels whose LDS opt-in has been made on THIS handle's device
    bool res_dead = false;     // a resident kernel did not come back even after the abort word: the stream (and the handle) cannot be used any more
    uint32_t res_probed_w = 0;     // several ranks: the grid size the probe launch has found resident together with the peers' (0: not yet)
    unsigned long long res_probe_id = 0;
    bool res_not_resident = false; // a resident grid was found partly resident (another process on the device): engine 0 means the batch engine from then on
    int res_early = 24;       // option early_advance (ResParams::early_advance)
    int res_announce = 1;     // option announce (ResParams::announce)
    int res_wend = 1;         // option window_end16 (ResParams::wend_mask)
    int res_walker2_ranks = 1; // option walker2_ranks: several ranks run the second walker too (0: the first, as in round 3)
    double eps_abs_bound = 0.0; // (sum of eps^8)^(1/8) >= max |eps| as of the last reduce_eps_all
    int res_refill = 0;       // option refill: the streaming workgroups' form -- 1 first (hg_resident.hip.h: fused multiply-adds), 2 second (hg_streamer2.hip.h: integer matrix products), 0 auto
    int res_tune = 0;         // option res_tune: experiments of the resident kernel (ResParams::tune)
    int res_pivots = 0;       // option pivots: Gram terms with predicted pivots at streaming time (no round trip for those events)
    unsigned char* res_acc = nullptr; // Gram + raw-dot accumulators, batch counters
    ResMsg* res_msg = nullptr;
    ResState* res_state = nullptr;
    ResState* res_state_host = nullptr; // pinned
    ResParams* res_params = nullptr;      // the sweep's parameters in device memory (the walker reads them there) and their pinned staging copy
    ResParams* res_params_host = nullptr;
    unsigned long long* res_progress = nullptr; // device, written by the kernel while it runs
    unsigned long long* res_progress_host = nullptr; // pinned copy, fetched on a second stream when the host's deadline passes
    hipStream_t aux_stream = nullptr;
    unsigned long long* res_trace = nullptr; // [8][RS_TRACE], debug_timing
    double res_timeout_s = 2.0;
    double res_deadline_s = 0.0; // option res_deadline_ms: the host's deadline for a resident sweep (0 = derived)

    hgibbs_sweep_stats stats{};
};

static int ensure_scratch(hgibbs_ctx* h, size_t n)
{
    if (h->scratch_n >= n) return 0;
    if (h->scratch) HIP_TRY(hipFree(h->scratch));
    HIP_TRY(hipMalloc(&h->scratch, n * sizeof(double)));
    h->scratch_n = n;
    return 0;
}

static constexpr size_t RES_GACC_BYTES = (size_t)2 * RS_NSH * RS_GROW * 4, RES_RACC_BYTES = (size_t)RS_RSH * RS_RB * 8, RES_RCNT_BYTES = (size_t)RS_RSH * RS_CROW * 4, RES_PACC_BYTES = (size_t)RS_RSH * RS_RB * 2 * 8;
static constexpr size_t RES_RACC2_OFF = RES_GACC_BYTES + RES_RACC_BYTES + RES_RCNT_BYTES + RES_PACC_BYTES, RES_GACC64_BYTES = (size_t)2 * RS_NSH * RS_GROW * 8;
static constexpr size_t RES_ACC_BYTES = RES_RACC2_OFF + RES_RACC_BYTES + RES_GACC64_BYTES; // (+ the MISS build's second raw sums and its 8-byte Gram words)
static constexpr size_t MBOX_DATA_BYTES = (size_t)2 * MAX_RANKS * ROWS_CAP * sizeof(double);
static constexpr size_t MBOX_BATCH_BYTES = MBOX_DATA_BYTES + (size_t)2 * MAX_RANKS * sizeof(unsigned long long); // the batch engine's rows and flags
static constexpr size_t MBOX_RES_OFF = (MBOX_BATCH_BYTES + 4095) / 4096 * 4096;                                  // behind them the resident engine's mailbox (hg_resident.hip.h, RX_*)
static constexpr size_t MBOX_BYTES = MBOX_RES_OFF + RX_BYTES;

// Sum a device buffer over ranks (rare, bulk): RCCL when a communicator exists,
// else the caller's transport on a host copy.  dtype 0 = f64, 1 = u64.
static int bulk_allreduce(hgibbs_ctx* h, void* dptr, size_t count, int dtype)
{
    if (h->nranks <= 1 && !h->comm) return 0;
    if (h->comm) {
        NCCL_TRY(ncclAllReduce(dptr, dptr, count, dtype == 0 ? ncclDouble : ncclUint64, ncclSum, h->comm, h->stream));
        return 0;
    }
    if (!h->ext_fn) return fail("multi-rank handle without a transport: call hgibbs_comm_init or hgibbs_comm_init_external");
    std::vector<unsigned char> buf(count * 8);
    HIP_TRY(hipMemcpyAsync(buf.data(), dptr, buf.size(), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->ext_fn(h->ext_user, buf.data(), count, dtype)) return fail("external all-reduce callback failed");
    HIP_TRY(hipMemcpyAsync(dptr, buf.data(), buf.size(), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return 0;
}

// ---------------------------------------------------------------------------
// helper kernels
// ---------------------------------------------------------------------------
namespace {

// counter-based hash for synthetic genotypes (splitmix64 finaliser)
__host__ __device__ inline uint64_t mix64(uint64_t z)
{
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

// One thread per byte (4 individuals) of a column.  Global row index decides
// the draw, so any sharding reproduces the same matrix.
__global__ void k_synth_bed(uint8_t* bed, uint64_t stride, uint32_t n_local, uint32_t row_begin, uint32_t marker0,
                            uint64_t seed, uint32_t miss_thr)
{
    const uint32_t marker = marker0 + blockIdx.y;
    const uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= stride) return;
    // allele frequency p_j ~ U(0.01, 0.5); integer thresholds on a 32-bit uniform
    const uint32_t up = (uint32_t)(mix64(seed ^ (0xA5A5A5A5ull + (uint64_t)marker * 0x100000001b3ull)) >> 32);
    const double p = 0.01 + 0.49 * ((double)up * (1.0 / 4294967296.0));
    const double q0 = (1.0 - p) * (1.0 - p);
    const double q1 = q0 + 2.0 * p * (1.0 - p);
    const uint32_t t0 = (uint32_t)(q0 * 4294967296.0);
    const uint32_t t1 = (uint32_t)(q1 * 4294967296.0);
    uint32_t byte = 0;
    for (int s = 0; s < 4; ++s) {
        const uint64_t il = b * 4 + s;
        uint32_t code = GC_MISS; // padding
        if (il < n_local) {
            const uint64_t ig = (uint64_t)row_begin + il;
            const uint64_t hsh = mix64(seed + (uint64_t)marker * 0x9E3779B1ull + ig * 0xD1B54A32D192ED03ull);
            const uint32_t ug = (uint32_t)(hsh >> 32), um = (uint32_t)hsh;
            if (um < miss_thr) code = GC_MISS;
            else code = (ug < t0) ? GC_G0 : ((ug < t1) ? GC_G1 : GC_G2); // device codes (hg_kernels.h): the field is the genotype
        }
        byte |= code << (2 * s);
    }
    bed[(uint64_t)marker * stride + b] = (uint8_t)byte;
}

// Set every 2-bit slot at local index >= n_local to the missing code.
__global__ void k_fix_padding(uint8_t* bed, uint64_t stride, uint32_t n_local, uint32_t M)
{
    const uint32_t marker = blockIdx.x * blockDim.x + threadIdx.x;
    if (marker >= M) return;
    const uint32_t bfirst = n_local >> 2;
    if ((n_local & 3u) && bfirst < stride) {
        uint8_t v = bed[(uint64_t)marker * stride + bfirst];
        for (uint32_t s = n_local & 3u; s < 4; ++s) v = (uint8_t)((v & ~(3u << (2 * s))) | (1u << (2 * s)));
        bed[(uint64_t)marker * stride + bfirst] = v;
    }
}

// PLINK codes -> device codes (hg_kernels.h: the 2-bit field becomes the genotype itself), in place, dword-wise over
// the whole padded buffer; run once after the file's bytes are in HBM.  HBM-bound one-pass kernel.
__global__ void k_recode_bed(uint32_t* __restrict__ bed, uint64_t ndwords)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ndwords; i += stride) bed[i] = recode_plink_to_device(bed[i]);
}

// NA-phenotype row removal: gather the kept 2-bit fields of a full column into
// the local packed column.  src_idx[i] = source individual of local slot i.
__global__ void k_compact_bed(const uint8_t* __restrict__ src, uint64_t src_stride, const uint32_t* __restrict__ src_idx,
                              uint8_t* __restrict__ dst, uint64_t dst_stride, uint32_t n_local, uint32_t mcount)
{
    const uint32_t marker = blockIdx.y;
    const uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= dst_stride || marker >= mcount) return;
    uint32_t byte = 0;
    for (int s = 0; s < 4; ++s) {
        const uint64_t il = b * 4 + s;
        uint32_t code = 1u;
        if (il < n_local) {
            const uint32_t i = src_idx[il];
            code = (src[(uint64_t)marker * src_stride + (i >> 2)] >> (2 * (i & 3u))) & 3u;
        }
        byte |= code << (2 * s);
    }
    dst[(uint64_t)marker * dst_stride + b] = (uint8_t)byte;
}

// per-marker genotype counts over the local shard: one block per marker
__global__ __launch_bounds__(BLOCK) void k_counts(const uint8_t* __restrict__ bed, uint64_t stride, uint32_t n_pad,
                                                  uint32_t n_local, unsigned long long* __restrict__ counts, uint32_t M)
{
    __shared__ unsigned int sh[3][BLOCK_WAVES];
    const uint32_t marker = blockIdx.x;
    const uint32_t* col = reinterpret_cast<const uint32_t*>(bed + (uint64_t)marker * stride);
    const uint32_t nw = (uint32_t)(stride >> 2);
    unsigned int c1 = 0, c2 = 0, cm = 0;
    for (uint32_t i = threadIdx.x; i < nw; i += BLOCK) {
        uint32_t m1, m2, mm;
        code_masks(col[i], m1, m2, mm);
        c1 += __popc(m1);
        c2 += __popc(m2);
        cm += __popc(mm);
    }
    for (int off = 32; off >= 1; off >>= 1) {
        c1 += __shfl_xor(c1, off, 64);
        c2 += __shfl_xor(c2, off, 64);
        cm += __shfl_xor(cm, off, 64);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        sh[0][wave] = c1;
        sh[1][wave] = c2;
        sh[2][wave] = cm;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t1 = 0, t2 = 0, tm = 0;
        for (int w = 0; w < BLOCK_WAVES; ++w) {
            t1 += sh[0][w];
            t2 += sh[1][w];
            tm += sh[2][w];
        }
        counts[3ull * marker] = t1;
        counts[3ull * marker + 1] = t2;
        counts[3ull * marker + 2] = tm - (unsigned long long)(n_pad - n_local); // padding slots are coded missing
    }
}

// a2: src/BayesRRm.cpp:1502-1507 from the (global) counts
__global__ void k_stats(const unsigned long long* __restrict__ counts, uint32_t N, double* mave, double* mstd, uint32_t M)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    const double n1 = (double)counts[3ull * i], n2 = (double)counts[3ull * i + 1], nm = (double)counts[3ull * i + 2];
    const double dN = (double)N;
    const double av = (n1 + 2.0 * n2) / (dN - nm);
    const double tmp1 = n1 * (1.0 - av) * (1.0 - av);
    const double tmp2 = n2 * (2.0 - av) * (2.0 - av);
    const unsigned long long n0 = (unsigned long long)N - counts[3ull * i] - counts[3ull * i + 1] - counts[3ull * i + 2];
    const double tmp0 = (double)n0 * (0.0 - av) * (0.0 - av);
    mave[i] = av;
    mstd[i] = sqrt((double)(N - 1) / (tmp0 + tmp1 + tmp2));
}

// per-marker metadata in sweep order (one pass per sweep): the draw phase then
// reads contiguous, order-independent rows
__global__ void k_gather_meta(const int32_t* __restrict__ order, const double* __restrict__ mave, const double* __restrict__ mstd,
                              const double* __restrict__ beta, const int32_t* __restrict__ groups, const uint8_t* __restrict__ adaV,
                              const unsigned long long* __restrict__ counts, double* s_mave, double* s_mstd, double* s_bold, int32_t* s_ga, uint32_t M)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= M) return;
    const int m = order[j];
    s_mave[j] = mave[m];
    s_mstd[j] = mstd[m];
    s_bold[j] = beta[m];
    s_ga[j] = groups[m] | (adaV[m] ? 0x40000000 : 0) | (counts[3ull * m + 2] ? 0x20000000 : 0);
}

// The sweep positions whose marker's effect is non-zero at sweep start (it WILL change: a predicted event), ascending, followed by 16
// sentinels 0xffffffff: the resident engine's streaming workgroups and its walker both read the window's pivots off this list.
// Three small launches: per chunk of 4096 positions a count, one workgroup's exclusive scan of the counts, an order-preserving scatter.
constexpr uint32_t PRED_CHUNK = 4096;
__global__ __launch_bounds__(256) void k_pred_count(const double* __restrict__ s_bold, uint32_t M, uint32_t* cnt)
{
    __shared__ uint32_t wsum[4];
    const uint32_t base = blockIdx.x * PRED_CHUNK;
    uint32_t n = 0;
    for (uint32_t i = 0; i < PRED_CHUNK / 256; ++i) {
        const uint32_t j = base + i * 256 + threadIdx.x;
        n += (uint32_t)__popcll(__ballot(j < M && s_bold[j] != 0.0));
    }
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = n; // (every lane of a wave holds the wave's count)
    __syncthreads();
    if (threadIdx.x == 0) cnt[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}
__global__ __launch_bounds__(1024) void k_pred_scan(uint32_t* cnt, uint32_t nchunk)
{
    __shared__ uint32_t tot[1024];
    const uint32_t per = (nchunk + 1023u) / 1024u, lo = threadIdx.x * per;
    uint32_t s = 0;
    for (uint32_t i = lo; i < lo + per && i < nchunk; ++i) s += cnt[i];
    tot[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int t = 0; t < 1024; ++t) {
            const uint32_t v = tot[t];
            tot[t] = run;
            run += v;
        }
        cnt[nchunk] = run; // the list's length
    }
    __syncthreads();
    uint32_t run = tot[threadIdx.x];
    for (uint32_t i = lo; i < lo + per && i < nchunk; ++i) {
        const uint32_t v = cnt[i];
        cnt[i] = run;
        run += v;
    }
}
__global__ __launch_bounds__(256) void k_pred_scatter(const double* __restrict__ s_bold, uint32_t M, const uint32_t* __restrict__ cnt, uint32_t nchunk, uint32_t* pred)
{
    __shared__ uint32_t wsum[4];
    const uint32_t base = blockIdx.x * PRED_CHUNK;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t run = cnt[blockIdx.x];
    for (uint32_t i = 0; i < PRED_CHUNK / 256; ++i) {
        const uint32_t j = base + i * 256 + threadIdx.x;
        const bool f = j < M && s_bold[j] != 0.0;
        const unsigned long long m = __ballot(f);
        __syncthreads(); // (the previous pass's wsum has been read)
        if (lane == 0) wsum[wave] = (uint32_t)__popcll(m);
        __syncthreads();
        uint32_t before = 0;
        for (int w = 0; w < wave; ++w) before += wsum[w];
        if (f) pred[run + before + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = j;
        run += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    }
    if (blockIdx.x == 0 && threadIdx.x < 16) pred[cnt[nchunk] + threadIdx.x] = 0xffffffffu;
}

__global__ void k_set_eps(double* eps, const double* __restrict__ src, uint32_t n_local, uint32_t n_pad)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pad) return;
    eps[eps_pos(i)] = (i < n_local) ? src[i] : 0.0;
}

__global__ void k_get_eps(const double* __restrict__ eps, double* dst, uint32_t n_local)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_local) return;
    dst[i] = eps[eps_pos(i)];
}

// eps_i += c for real individuals only (padding stays 0)
__global__ void k_add_scalar(double* eps, double c, uint32_t n_local, uint32_t n_pad)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_local) return;
    const uint32_t p = eps_pos(i);
    eps[p] = eps[p] + c;
}

// per-block partial (sum, sqn, sum of eps^8) in natural individual order inside the block
// (the third: max |eps| <= (sum eps^8)^(1/8) -- how the host knows that every residual is inside the range of the streaming workgroups'
// digits, hg_streamer2.hip.h, without a reduction of its own)
__global__ __launch_bounds__(BLOCK) void k_reduce_eps(const double* __restrict__ eps, uint32_t n_pad, double* partial)
{
    __shared__ double sh[3][BLOCK_WAVES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t tile = blockIdx.x * BLOCK_WAVES + wave;
    double e[IPT];
    load_eps16(eps, tile, lane, e);
    double s = 0.0, q = 0.0, o = 0.0;
#pragma unroll
    for (int i = 0; i < IPT; ++i) {
        const double e2 = e[i] * e[i], e4 = e2 * e2;
        s += e[i];
        q += e2;
        o += e4 * e4;
    }
    s = wave_sum(s);
    q = wave_sum(q);
    o = wave_sum(o);
    if (lane == 0) {
        sh[0][wave] = s;
        sh[1][wave] = q;
        sh[2][wave] = o;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double ts = 0.0, tq = 0.0, to = 0.0;
        for (int w = 0; w < BLOCK_WAVES; ++w) {
            ts += sh[0][w];
            tq += sh[1][w];
            to += sh[2][w];
        }
        partial[3 * blockIdx.x] = ts;
        partial[3 * blockIdx.x + 1] = tq;
        partial[3 * blockIdx.x + 2] = to;
    }
}

// fixed-order final reduction of `rows` interleaved partial rows: out[r] = sum_b partial[b*rows + r]
__global__ __launch_bounds__(BLOCK) void k_final_sum(const double* __restrict__ partial, uint32_t nblk, uint32_t rows, double* out)
{
    __shared__ double sh[BLOCK_WAVES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t r = 0; r < rows; ++r) {
        double v = 0.0;
        for (uint32_t b = threadIdx.x; b < nblk; b += BLOCK) v += partial[(size_t)b * rows + r];
        v = wave_sum(v);
        if (lane == 0) sh[wave] = v;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
            for (int w = 0; w < BLOCK_WAVES; ++w) t += sh[w];
            out[r] = t;
        }
        __syncthreads();
    }
}

// single-marker masked sums (S1,S2,SM,Sall) per block -> partial[b*4 + {0..3}]
__global__ __launch_bounds__(BLOCK) void k_dot_one(const uint8_t* __restrict__ bed, uint64_t stride, uint32_t marker,
                                                   const double* __restrict__ eps, double* partial)
{
    __shared__ double sh[4][BLOCK_WAVES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t tile = blockIdx.x * BLOCK_WAVES + wave;
    double e[IPT];
    load_eps16(eps, tile, lane, e);
    const uint32_t w = *reinterpret_cast<const uint32_t*>(bed + (size_t)marker * stride + ((size_t)tile << 8) + (lane << 2));
    uint32_t m1, m2, mm;
    code_masks(w, m1, m2, mm);
    double s1 = 0.0, s2 = 0.0, sm = 0.0, sa = 0.0;
#pragma unroll
    for (int s = 0; s < IPT; ++s) {
        s1 += mask_f64(e[s], ((int)(m1 << (31 - 2 * s))) >> 31);
        s2 += mask_f64(e[s], ((int)(m2 << (31 - 2 * s))) >> 31);
        sm += mask_f64(e[s], ((int)(mm << (31 - 2 * s))) >> 31);
        sa += e[s];
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    sm = wave_sum(sm);
    sa = wave_sum(sa);
    if (lane == 0) {
        sh[0][wave] = s1;
        sh[1][wave] = s2;
        sh[2][wave] = sm;
        sh[3][wave] = sa;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        double t = 0.0;
        for (int wv = 0; wv < BLOCK_WAVES; ++wv) t += sh[threadIdx.x][wv];
        partial[4 * blockIdx.x + threadIdx.x] = t;
    }
}

// covariate column (row-major host matrix) -> permuted device column
__global__ void k_set_cov(double* dst, const double* __restrict__ X, uint32_t n_local, uint32_t n_pad, int C, int c)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pad) return;
    dst[eps_pos(i)] = (i < n_local) ? X[(size_t)i * C + c] : 0.0;
}

// per-block partial of sum_k x_k * (eps_k + g * x_k)   (src/BayesRRm.cpp:2666-2668)
__global__ __launch_bounds__(BLOCK) void k_cov_dot(const double* __restrict__ x, const double* __restrict__ eps, double g, double* partial)
{
    __shared__ double sh[BLOCK_WAVES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t tile = blockIdx.x * BLOCK_WAVES + wave;
    double e[IPT], xv[IPT];
    load_eps16(eps, tile, lane, e);
    load_eps16(x, tile, lane, xv);
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < IPT; ++i) s += xv[i] * (e[i] + g * xv[i]);
    s = wave_sum(s);
    if (lane == 0) sh[wave] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < BLOCK_WAVES; ++w) t += sh[w];
        partial[blockIdx.x] = t;
    }
}

// eps_k = eps_k + d * x_k   (src/BayesRRm.cpp:2673-2676)
__global__ __launch_bounds__(BLOCK) void k_cov_update(const double* __restrict__ x, double* eps, double dgam)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t tile = blockIdx.x * BLOCK_WAVES + wave;
    double e[IPT], xv[IPT];
    load_eps16(eps, tile, lane, e);
    load_eps16(x, tile, lane, xv);
#pragma unroll
    for (int i = 0; i < IPT; ++i) e[i] = e[i] + dgam * xv[i];
    store_eps16(eps, tile, lane, e);
}

__global__ __launch_bounds__(BLOCK) void k_update_one(const uint8_t* __restrict__ bed, uint64_t stride, uint32_t marker,
                                                      double* eps, double v0, double v1, double v2)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t tile = blockIdx.x * BLOCK_WAVES + wave;
    double e[IPT];
    load_eps16(eps, tile, lane, e);
    const uint32_t w = *reinterpret_cast<const uint32_t*>(bed + (size_t)marker * stride + ((size_t)tile << 8) + (lane << 2));
    apply_update16(w, v0, v1, v2, e);
    store_eps16(eps, tile, lane, e);
}

} // namespace

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
extern "C" {

const char* hgibbs_last_error(void) { return g_err.c_str(); }
int hgibbs_version(void) { return 1; }

int hgibbs_create(int device_id, hgibbs_t* out)
{
    if (!out) return fail("hgibbs_create: null out");
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (ndev <= 0) return fail("hgibbs_create: no HIP device (this library has no CPU fallback)");
    if (device_id < 0 || device_id >= ndev) return fail("hgibbs_create: device %d out of range (%d devices)", device_id, ndev);
    HIP_TRY(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device_id));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail("hgibbs_create: device %d is %s, this library is built for gfx950 only", device_id, prop.gcnArchName);
    hgibbs_ctx* h = new hgibbs_ctx();
    h->device = device_id;
    h->num_cu = prop.multiProcessorCount;
    HIP_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreate(&h->ev0));
    HIP_TRY(hipEventCreate(&h->ev1));
    HIP_TRY(hipMalloc(&h->mt, MT_N * sizeof(uint32_t)));
    HIP_TRY(hipMalloc(&h->zig, (129 + 129 + 257 + 257) * sizeof(double)));
    HIP_TRY(hipMemcpy(h->zig, HG_ZIG_NORMAL_X, 129 * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->zig + 129, HG_ZIG_NORMAL_Y, 129 * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->zig + 258, HG_ZIG_EXP_X, 257 * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->zig + 515, HG_ZIG_EXP_Y, 257 * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMalloc(&h->desc, sizeof(SweepDesc) + sizeof(SweepCounters)));      // descriptor, then the sweep's counters
    HIP_TRY(hipHostMalloc(&h->desc_host, sizeof(SweepDesc) + sizeof(SweepCounters)));
    HIP_TRY(hipMalloc(&h->ticket, (16 + MAX_GROUPS + 256) * sizeof(uint32_t)));
    HIP_TRY(hipMemset(h->ticket, 0, (16 + MAX_GROUPS + 256) * sizeof(uint32_t)));
    HIP_TRY(hipMalloc(&h->totals, (size_t)ROWS_CAP * sizeof(double)));
    HIP_TRY(hipMalloc(&h->carry, (size_t)MAX_BATCH * sizeof(double)));
    HIP_TRY(hipMalloc(&h->ahead_raw, (size_t)2 * AHEAD_MAX * 2 * sizeof(double)));
    HIP_TRY(hipMalloc(&h->apartials, (size_t)S_CAP * 2 * AHEAD_MAX * sizeof(double)));
    HIP_TRY(hipMalloc(&h->aticket, (AHEAD_MAX / 2 + 4) * sizeof(uint32_t)));
    HIP_TRY(hipMemset(h->aticket, 0, (AHEAD_MAX / 2 + 4) * sizeof(uint32_t)));
    HIP_TRY(hipMalloc(&h->sums, (NROW * MAX_BATCH + 1) * sizeof(double)));
    HIP_TRY(hipMalloc(&h->dbg, 48 * sizeof(unsigned long long)));
    HIP_TRY(hipMemset(h->dbg, 0, 48 * sizeof(unsigned long long)));
    HIP_TRY(hipHostMalloc(&h->scratch_host, 4096 * sizeof(double)));
    HIP_TRY(hipMalloc(&h->res_acc, RES_ACC_BYTES));
    HIP_TRY(hipMalloc(&h->res_msg, RS_MSG * sizeof(ResMsg)));
    HIP_TRY(hipMalloc(&h->res_state, sizeof(ResState)));
    HIP_TRY(hipHostMalloc(&h->res_state_host, sizeof(ResState)));
    HIP_TRY(hipMalloc(&h->res_params, sizeof(ResParams)));
    HIP_TRY(hipHostMalloc(&h->res_params_host, sizeof(ResParams)));
    HIP_TRY(hipMalloc(&h->res_progress, 16 * sizeof(unsigned long long)));
    HIP_TRY(hipHostMalloc(&h->res_progress_host, 16 * sizeof(unsigned long long)));
    HIP_TRY(hipStreamCreateWithFlags(&h->aux_stream, hipStreamNonBlocking));
    HIP_TRY(hipMalloc(&h->res_trace, (size_t)10 * RS_TRACE * sizeof(unsigned long long)));
    HIP_TRY(hipMemset(h->res_trace, 0, (size_t)10 * RS_TRACE * sizeof(unsigned long long)));
    *out = h;
    return 0;
}

int hgibbs_destroy(hgibbs_t h)
{
    if (!h) return 0;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    if (h->comm) ncclCommDestroy(h->comm);
    bw_free(h->bw);
    for (int r = 0; r < MAX_RANKS; ++r)
        if (h->peer_base[r] && h->peer_base[r] != h->mbox) (void)hipIpcCloseMemHandle(h->peer_base[r]);
    if (h->mbox) (void)hipFree(h->mbox);
    void* ptrs[] = {h->bed, h->eps[0], h->eps[1], h->mave, h->mstd, h->counts, h->groups, h->beta, h->comp, h->acum, h->order,
                    h->adaV, h->covX, h->s_mave, h->s_mstd, h->s_bold, h->s_ga, h->pred, h->pred_cnt, h->dbg, h->cass, h->tables, h->mt, h->zig, h->desc, h->partials, h->totals, h->ticket, h->sums, h->scratch, h->carry, h->ahead_raw, h->apartials, h->aticket, h->res_acc, h->res_msg, h->res_state, h->res_trace};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    if (h->desc_host) (void)hipHostFree(h->desc_host);
    if (h->scratch_host) (void)hipHostFree(h->scratch_host);
    if (h->res_state_host) (void)hipHostFree(h->res_state_host);
    if (h->res_params) (void)hipFree(h->res_params);
    if (h->res_params_host) (void)hipHostFree(h->res_params_host);
    if (h->res_progress) (void)hipFree(h->res_progress);
    if (h->res_progress_host) (void)hipHostFree(h->res_progress_host);
    if (h->aux_stream) (void)hipStreamDestroy(h->aux_stream);
    if (h->beta_host) (void)hipHostFree(h->beta_host);
    (void)hipEventDestroy(h->ev0);
    (void)hipEventDestroy(h->ev1);
    (void)hipStreamDestroy(h->stream);
    delete h;
    return 0;
}

int hgibbs_comm_unique_id(void* id128)
{
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    NCCL_TRY(ncclGetUniqueId(&id));
    std::memcpy(id128, &id, sizeof id);
    return 0;
}

int hgibbs_comm_init(hgibbs_t h, int nranks, int rank, const void* id128)
{
    if (!h) return fail("null handle");
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail("hgibbs_comm_init: bad rank %d of %d", rank, nranks);
    h->nranks = nranks;
    h->rank = rank;
    if (nranks == 1 && !id128) return 0; // a 1-rank communicator is only built on request (tests of the split path)
    HIP_TRY(hipSetDevice(h->device));
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof id);
    NCCL_TRY(ncclCommInitRank(&h->comm, nranks, id, rank));
    return 0;
}

extern "C" int hgibbs_comm_init_external(hgibbs_t h, int nranks, int rank, hgibbs_allreduce_fn fn, void* user)
{
    if (!h) return fail("null handle");
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail("hgibbs_comm_init_external: bad rank %d of %d", rank, nranks);
    if (nranks > 1 && !fn) return fail("hgibbs_comm_init_external: null callback");
    h->nranks = nranks;
    h->rank = rank;
    h->ext_fn = fn;
    h->ext_user = user;
    return 0;
}

extern "C" int hgibbs_p2p_export(hgibbs_t h, void* handle64)
{
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
    if (!h || !handle64) return fail("hgibbs_p2p_export: null argument");
    HIP_TRY(hipSetDevice(h->device));
    if (!h->mbox) {
        // uncached (fine-grained) device memory: remote GPUs' stores must be visible to local loads
        void* p = nullptr;
        hipError_t e = hipExtMallocWithFlags(&p, MBOX_BYTES, hipDeviceMallocUncached);
        hipIpcMemHandle_t probe;
        if (e != hipSuccess || hipIpcGetMemHandle(&probe, p) != hipSuccess) {
            (void)hipGetLastError();
            if (e == hipSuccess) (void)hipFree(p);
            HIP_TRY(hipMalloc(&p, MBOX_BYTES));
        }
        h->mbox = (unsigned char*)p;
        HIP_TRY(hipMemset(h->mbox, 0, MBOX_BYTES));
    }
    hipIpcMemHandle_t hd;
    HIP_TRY(hipIpcGetMemHandle(&hd, h->mbox));
    std::memcpy(handle64, &hd, sizeof hd);
    return 0;
}

extern "C" int hgibbs_p2p_import(hgibbs_t h, const void* handles)
{
    if (!h || !handles) return fail("hgibbs_p2p_import: null argument");
    if (h->nranks < 2) return fail("hgibbs_p2p_import: initialise the ranks first (hgibbs_comm_init / hgibbs_comm_init_external)");
    if (h->nranks > MAX_RANKS) return fail("hgibbs_p2p_import: at most %d ranks", MAX_RANKS);
    if (!h->mbox) return fail("hgibbs_p2p_import: call hgibbs_p2p_export first");
    HIP_TRY(hipSetDevice(h->device));
    for (int r = 0; r < h->nranks; ++r) {
        if (r == h->rank) {
            h->peer_base[r] = h->mbox;
            continue;
        }
        hipIpcMemHandle_t hd;
        std::memcpy(&hd, (const unsigned char*)handles + (size_t)r * 64, sizeof hd);
        void* p = nullptr;
        HIP_TRY(hipIpcOpenMemHandle(&p, hd, hipIpcMemLazyEnablePeerAccess));
        h->peer_base[r] = p;
    }
    h->p2p_ready = true;
    return 0;
}

static int alloc_problem(hgibbs_ctx* h, uint32_t n_global, uint32_t n_local, uint32_t M, uint32_t row_begin)
{
    if (n_local == 0 || M == 0) return fail("empty problem: n_local=%u M=%u", n_local, M);
    if (h->bed) return fail("data already loaded on this handle");
    h->n_global = n_global;
    h->n_local = n_local;
    h->M = M;
    h->row_begin = row_begin;
    h->n_pad = (uint32_t)(((uint64_t)n_local + BLOCK_IND - 1) / BLOCK_IND * BLOCK_IND);
    h->stride = h->n_pad / 4;
    HIP_TRY(hipMalloc(&h->bed, (size_t)M * h->stride));
    for (int b = 0; b < 2; ++b) {
        HIP_TRY(hipMalloc(&h->eps[b], (size_t)h->n_pad * sizeof(double)));
        HIP_TRY(hipMemsetAsync(h->eps[b], 0, (size_t)h->n_pad * sizeof(double), h->stream));
    }
    HIP_TRY(hipMalloc(&h->mave, (size_t)M * sizeof(double)));
    HIP_TRY(hipMalloc(&h->mstd, (size_t)M * sizeof(double)));
    HIP_TRY(hipMalloc(&h->counts, (size_t)M * 3 * sizeof(unsigned long long)));
    HIP_TRY(hipMalloc(&h->beta, (size_t)M * sizeof(double)));
    HIP_TRY(hipMalloc(&h->comp, (size_t)M * sizeof(int32_t)));
    HIP_TRY(hipMalloc(&h->acum, (size_t)M * sizeof(double)));
    HIP_TRY(hipMalloc(&h->order, (size_t)M * sizeof(int32_t)));
    HIP_TRY(hipMalloc(&h->adaV, (size_t)M));
    HIP_TRY(hipMalloc(&h->s_mave, (size_t)M * sizeof(double)));
    HIP_TRY(hipMalloc(&h->s_mstd, (size_t)M * sizeof(double)));
    HIP_TRY(hipMalloc(&h->s_bold, (size_t)M * sizeof(double)));
    HIP_TRY(hipMalloc(&h->s_ga, (size_t)M * sizeof(int32_t)));
    HIP_TRY(hipMalloc(&h->pred, ((size_t)M + 16) * sizeof(uint32_t)));
    HIP_TRY(hipMalloc(&h->pred_cnt, ((size_t)(M + PRED_CHUNK - 1) / PRED_CHUNK + 1) * sizeof(uint32_t)));
    HIP_TRY(hipMalloc(&h->groups, (size_t)M * sizeof(int32_t)));
    HIP_TRY(hipMemsetAsync(h->beta, 0, (size_t)M * sizeof(double), h->stream));
    HIP_TRY(hipMemsetAsync(h->comp, 0, (size_t)M * sizeof(int32_t), h->stream));
    HIP_TRY(hipMemsetAsync(h->acum, 0, (size_t)M * sizeof(double), h->stream));
    HIP_TRY(hipMemsetAsync(h->groups, 0, (size_t)M * sizeof(int32_t), h->stream));
    const uint32_t nblk_x = h->n_pad / BLOCK_IND;
    HIP_TRY(hipMalloc(&h->partials, (size_t)PROWS_CAP * S_CAP * sizeof(double)));
    if (ensure_scratch(h, (size_t)nblk_x * 4 + 4096)) return 1;
    h->eps_cur = 0;
    h->have_stats = false;
    return 0;
}

int hgibbs_load_bed(hgibbs_t h, const uint8_t* bed_host, uint64_t stride_in, uint32_t n_total, uint32_t M,
                    const uint8_t* keep_host, uint32_t row_begin, uint32_t row_end, uint32_t n_global)
{
    if (!h || !bed_host) return fail("hgibbs_load_bed: null argument");
    if (row_end <= row_begin) return fail("hgibbs_load_bed: empty row range");
    if (stride_in < ((uint64_t)n_total + 3) / 4) return fail("hgibbs_load_bed: stride_in %llu < ceil(%u/4)", (unsigned long long)stride_in, n_total);
    HIP_TRY(hipSetDevice(h->device));
    const uint32_t n_local = row_end - row_begin;
    // validate everything before anything is allocated on the handle
    std::vector<uint32_t> kept;
    if (!keep_host) {
        if (row_end > n_total) return fail("hgibbs_load_bed: row_end %u > n_total %u", row_end, n_total);
        if (row_begin & 3u) return fail("hgibbs_load_bed: row_begin %u must be a multiple of 4", row_begin);
    } else {
        kept.reserve(n_total);
        for (uint32_t i = 0; i < n_total; ++i)
            if (keep_host[i]) kept.push_back(i);
        if (row_end > kept.size()) return fail("hgibbs_load_bed: row_end %u > kept individuals %zu", row_end, kept.size());
    }
    if (n_global < 2) return fail("hgibbs_load_bed: n_global must be at least 2");
    if (alloc_problem(h, n_global, n_local, M, row_begin)) return 1;
    HIP_TRY(hipMemsetAsync(h->bed, 0x55, (size_t)M * h->stride, h->stream));

    if (!keep_host) {
        const size_t width = ((size_t)n_local + 3) / 4;
        HIP_TRY(hipMemcpy2DAsync(h->bed, h->stride, bed_host + (row_begin >> 2), stride_in, width, M, hipMemcpyHostToDevice, h->stream));
        k_fix_padding<<<(M + 255) / 256, 256, 0, h->stream>>>(h->bed, h->stride, n_local, M);
        HIP_TRY(hipGetLastError());
    } else {
        // NA rows dropped: build the kept-row index list, then gather on the device in slabs of columns
        // (the two staging buffers are released on every exit, the error returns included)
        struct Staging {
            uint32_t* idx = nullptr;
            uint8_t* src = nullptr;
            ~Staging()
            {
                if (src) (void)hipFree(src);
                if (idx) (void)hipFree(idx);
            }
        } st;
        HIP_TRY(hipMalloc(&st.idx, (size_t)n_local * sizeof(uint32_t)));
        HIP_TRY(hipMemcpyAsync(st.idx, kept.data() + row_begin, (size_t)n_local * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));
        const uint32_t slab = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(M, (256ull << 20) / std::max<uint64_t>(1, stride_in)));
        HIP_TRY(hipMalloc(&st.src, (size_t)slab * stride_in));
        for (uint32_t m0 = 0; m0 < M; m0 += slab) {
            const uint32_t mc = std::min(slab, M - m0);
            HIP_TRY(hipMemcpyAsync(st.src, bed_host + (size_t)m0 * stride_in, (size_t)mc * stride_in, hipMemcpyHostToDevice, h->stream));
            dim3 grid((uint32_t)((h->stride + 255) / 256), mc);
            k_compact_bed<<<grid, 256, 0, h->stream>>>(st.src, stride_in, st.idx, h->bed + (size_t)m0 * h->stride, h->stride, n_local, mc);
            HIP_TRY(hipGetLastError());
        }
        HIP_TRY(hipStreamSynchronize(h->stream)); // before ~Staging frees what the kernels read
    }
    // the file's bytes (padding included: PLINK's missing code) are in place: re-code them for the kernels
    k_recode_bed<<<4096, 256, 0, h->stream>>>(reinterpret_cast<uint32_t*>(h->bed), (uint64_t)M * h->stride / 4);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(h->stream));
    return 0;
}

int hgibbs_synth_bed(hgibbs_t h, uint32_t n_global, uint32_t M, uint32_t row_begin, uint32_t row_end, uint64_t seed,
                     double missing_rate)
{
    if (!h) return fail("null handle");
    if (row_end <= row_begin || row_end > n_global) return fail("hgibbs_synth_bed: bad row range");
    HIP_TRY(hipSetDevice(h->device));
    if (alloc_problem(h, n_global, row_end - row_begin, M, row_begin)) return 1;
    double mr = missing_rate < 0 ? 0 : (missing_rate > 1 ? 1 : missing_rate);
    const uint32_t miss_thr = (uint32_t)std::min(4294967295.0, mr * 4294967296.0);
    // grid.y is limited to 65535: synthesise in slabs of markers
    for (uint32_t m0 = 0; m0 < M; m0 += 32768) {
        const uint32_t mc = std::min<uint32_t>(32768, M - m0);
        dim3 grid((uint32_t)((h->stride + 255) / 256), mc);
        k_synth_bed<<<grid, 256, 0, h->stream>>>(h->bed, h->stride, h->n_local, row_begin, m0, seed, miss_thr);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipStreamSynchronize(h->stream));
    return 0;
}

int hgibbs_dims(hgibbs_t h, uint32_t* n_global, uint32_t* n_local, uint32_t* M, uint32_t* row_begin)
{
    if (!h) return fail("hgibbs_dims: null handle");
    if (n_global) *n_global = h->n_global;
    if (n_local) *n_local = h->n_local;
    if (M) *M = h->M;
    if (row_begin) *row_begin = h->row_begin;
    return 0;
}

int hgibbs_get_bed(hgibbs_t h, uint32_t m0, uint32_t mcount, uint8_t* out_host, uint64_t out_stride)
{
    if (!h || !h->bed) return fail("hgibbs_get_bed: no data");
    if ((uint64_t)m0 + mcount > h->M) return fail("hgibbs_get_bed: marker range out of bounds");
    HIP_TRY(hipSetDevice(h->device));
    const size_t width = ((size_t)h->n_local + 3) / 4;
    if (out_stride < width) return fail("hgibbs_get_bed: out_stride too small");
    HIP_TRY(hipMemcpy2D(out_host, out_stride, h->bed + (size_t)m0 * h->stride, h->stride, width, mcount, hipMemcpyDeviceToHost));
    // device codes -> the file's (PLINK) codes; slots past n_local in the last byte come back as PLINK's missing code
    for (uint32_t m = 0; m < mcount; ++m) {
        uint8_t* row = out_host + (size_t)m * out_stride;
        for (size_t b = 0; b < width; ++b) row[b] = (uint8_t)recode_device_to_plink(row[b]);
    }
    return 0;
}

static int compute_stats(hgibbs_ctx* h)
{
    if (h->have_stats) return 0;
    k_counts<<<h->M, BLOCK, 0, h->stream>>>(h->bed, h->stride, h->n_pad, h->n_local, h->counts, h->M);
    HIP_TRY(hipGetLastError());
    if (bulk_allreduce(h, h->counts, (size_t)h->M * 3, 1)) return 1;
    k_stats<<<(h->M + 255) / 256, 256, 0, h->stream>>>(h->counts, h->n_global, h->mave, h->mstd, h->M);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(h->stream));
    {
        std::vector<unsigned long long> c((size_t)h->M * 3);
        HIP_TRY(hipMemcpy(c.data(), h->counts, c.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        uint64_t nmc = 0;
        for (uint32_t i = 0; i < h->M; ++i) nmc += c[3ull * i + 2] != 0 ? 1u : 0u;
        h->any_missing = nmc != 0;
        h->missing_col_frac = (double)nmc / (double)h->M;
    }
    h->have_stats = true;
    return 0;
}

int hgibbs_marker_stats(hgibbs_t h, double* mave_host, double* mstd_host, uint64_t* n1_host, uint64_t* n2_host, uint64_t* nmiss_host)
{
    if (!h || !h->bed) return fail("hgibbs_marker_stats: no data loaded");
    HIP_TRY(hipSetDevice(h->device));
    if (compute_stats(h)) return 1;
    if (mave_host) HIP_TRY(hipMemcpy(mave_host, h->mave, (size_t)h->M * sizeof(double), hipMemcpyDeviceToHost));
    if (mstd_host) HIP_TRY(hipMemcpy(mstd_host, h->mstd, (size_t)h->M * sizeof(double), hipMemcpyDeviceToHost));
    if (n1_host || n2_host || nmiss_host) {
        std::vector<unsigned long long> c((size_t)h->M * 3);
        HIP_TRY(hipMemcpy(c.data(), h->counts, c.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        for (uint32_t i = 0; i < h->M; ++i) {
            if (n1_host) n1_host[i] = c[3ull * i];
            if (n2_host) n2_host[i] = c[3ull * i + 1];
            if (nmiss_host) nmiss_host[i] = c[3ull * i + 2];
        }
    }
    return 0;
}

int hgibbs_set_residual(hgibbs_t h, const double* eps_host)
{
    if (!h || !h->bed) return fail("hgibbs_set_residual: no data loaded");
    HIP_TRY(hipSetDevice(h->device));
    double* tmp = nullptr;
    HIP_TRY(hipMalloc(&tmp, (size_t)h->n_local * sizeof(double)));
    HIP_TRY(hipMemcpyAsync(tmp, eps_host, (size_t)h->n_local * sizeof(double), hipMemcpyHostToDevice, h->stream));
    k_set_eps<<<(h->n_pad + 255) / 256, 256, 0, h->stream>>>(h->eps[h->eps_cur], tmp, h->n_local, h->n_pad);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipFree(tmp));
    return 0;
}

int hgibbs_get_residual(hgibbs_t h, double* eps_host)
{
    if (!h || !h->bed) return fail("hgibbs_get_residual: no data loaded");
    HIP_TRY(hipSetDevice(h->device));
    double* tmp = nullptr;
    HIP_TRY(hipMalloc(&tmp, (size_t)h->n_local * sizeof(double)));
    k_get_eps<<<(h->n_local + 255) / 256, 256, 0, h->stream>>>(h->eps[h->eps_cur], tmp, h->n_local);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(eps_host, tmp, (size_t)h->n_local * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipFree(tmp));
    return 0;
}

// sum of eps and of eps^2 over all ranks, fixed order: out[0], out[1]
static int reduce_eps_all(hgibbs_ctx* h, double out[2])
{
    const uint32_t nblk = h->n_pad / BLOCK_IND;
    k_reduce_eps<<<nblk, BLOCK, 0, h->stream>>>(h->eps[h->eps_cur], h->n_pad, h->scratch);
    k_final_sum<<<1, BLOCK, 0, h->stream>>>(h->scratch, nblk, 3, h->sums);
    HIP_TRY(hipGetLastError());
    if (bulk_allreduce(h, h->sums, 3, 0)) return 1;
    HIP_TRY(hipMemcpyAsync(h->scratch_host, h->sums, 3 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    out[0] = h->scratch_host[0];
    out[1] = h->scratch_host[1];
    h->eps_abs_bound = std::pow(std::max(h->scratch_host[2], 0.0), 0.125); // >= max |eps| over all ranks (not finite: no bound)
    return 0;
}

int hgibbs_reduce_eps(hgibbs_t h, double* sum, double* sqn)
{
    if (!h || !h->bed) return fail("hgibbs_reduce_eps: no data loaded");
    HIP_TRY(hipSetDevice(h->device));
    double r[2];
    if (reduce_eps_all(h, r)) return 1;
    if (sum) *sum = r[0];
    if (sqn) *sqn = r[1];
    return 0;
}

int hgibbs_add_scalar(hgibbs_t h, double c)
{
    if (!h || !h->bed) return fail("hgibbs_add_scalar: no data loaded");
    HIP_TRY(hipSetDevice(h->device));
    k_add_scalar<<<(h->n_local + 255) / 256, 256, 0, h->stream>>>(h->eps[h->eps_cur], c, h->n_local, h->n_pad);
    HIP_TRY(hipGetLastError());
    return 0;
}

int hgibbs_update_marker(hgibbs_t h, uint32_t marker, double dbeta)
{
    if (!h || !h->bed) return fail("hgibbs_update_marker: no data loaded");
    if (marker >= h->M) return fail("hgibbs_update_marker: marker %u >= M %u", marker, h->M);
    HIP_TRY(hipSetDevice(h->device));
    if (compute_stats(h)) return 1;
    if (dbeta == 0.0) return 0;
    double ms[2];
    HIP_TRY(hipMemcpy(&ms[0], h->mave + marker, sizeof(double), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(&ms[1], h->mstd + marker, sizeof(double), hipMemcpyDeviceToHost));
    const double v0 = -(ms[0] * ms[1] * dbeta);
    const double v1 = dbeta * (1.0 - ms[0]) * ms[1];
    const double v2 = dbeta * (2.0 - ms[0]) * ms[1];
    k_update_one<<<h->n_pad / BLOCK_IND, BLOCK, 0, h->stream>>>(h->bed, h->stride, marker, h->eps[h->eps_cur], v0, v1, v2);
    HIP_TRY(hipGetLastError());
    return 0;
}

int hgibbs_dot_marker(hgibbs_t h, uint32_t marker, double* num)
{
    if (!h || !h->bed || !num) return fail("hgibbs_dot_marker: bad argument");
    if (marker >= h->M) return fail("hgibbs_dot_marker: marker %u >= M %u", marker, h->M);
    HIP_TRY(hipSetDevice(h->device));
    if (compute_stats(h)) return 1;
    const uint32_t nblk = h->n_pad / BLOCK_IND;
    k_dot_one<<<nblk, BLOCK, 0, h->stream>>>(h->bed, h->stride, marker, h->eps[h->eps_cur], h->scratch);
    k_final_sum<<<1, BLOCK, 0, h->stream>>>(h->scratch, nblk, 4, h->sums);
    HIP_TRY(hipGetLastError());
    if (bulk_allreduce(h, h->sums, 4, 0)) return 1;
    HIP_TRY(hipMemcpyAsync(h->scratch_host, h->sums, 4 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(h->scratch_host + 4, h->mave + marker, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(h->scratch_host + 5, h->mstd + marker, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    const double S1 = h->scratch_host[0], S2 = h->scratch_host[1], SM = h->scratch_host[2], Sall = h->scratch_host[3];
    double dp = 0.0;
    dp += S1 * 1.0;
    dp += S2 * 2.0;
    double syt = Sall;
    syt -= SM;
    dp -= (h->scratch_host[4] * syt);
    dp *= h->scratch_host[5];
    *num = dp;
    return 0;
}

int hgibbs_set_covariates(hgibbs_t h, const double* X_host, int C)
{
    if (!h || !h->bed) return fail("hgibbs_set_covariates: load genotypes first");
    if (C < 0 || (C > 0 && !X_host)) return fail("hgibbs_set_covariates: bad argument");
    HIP_TRY(hipSetDevice(h->device));
    if (h->covX) HIP_TRY(hipFree(h->covX));
    h->covX = nullptr;
    h->C = C;
    if (C == 0) return 0;
    HIP_TRY(hipMalloc(&h->covX, (size_t)C * h->n_pad * sizeof(double)));
    double* tmp = nullptr;
    HIP_TRY(hipMalloc(&tmp, (size_t)h->n_local * C * sizeof(double)));
    HIP_TRY(hipMemcpyAsync(tmp, X_host, (size_t)h->n_local * C * sizeof(double), hipMemcpyHostToDevice, h->stream));
    for (int c = 0; c < C; ++c)
        k_set_cov<<<(h->n_pad + 255) / 256, 256, 0, h->stream>>>(h->covX + (size_t)c * h->n_pad, tmp, h->n_local, h->n_pad, C, c);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipFree(tmp));
    return 0;
}

int hgibbs_cov_dot(hgibbs_t h, int c, double gamma_old, double* num_f)
{
    if (!h || !h->covX || !num_f) return fail("hgibbs_cov_dot: no covariates set");
    if (c < 0 || c >= h->C) return fail("hgibbs_cov_dot: covariate %d outside [0,%d)", c, h->C);
    HIP_TRY(hipSetDevice(h->device));
    const uint32_t nblk = h->n_pad / BLOCK_IND;
    k_cov_dot<<<nblk, BLOCK, 0, h->stream>>>(h->covX + (size_t)c * h->n_pad, h->eps[h->eps_cur], gamma_old, h->scratch);
    k_final_sum<<<1, BLOCK, 0, h->stream>>>(h->scratch, nblk, 1, h->sums);
    HIP_TRY(hipGetLastError());
    if (bulk_allreduce(h, h->sums, 1, 0)) return 1;
    HIP_TRY(hipMemcpyAsync(h->scratch_host, h->sums, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    *num_f = h->scratch_host[0];
    return 0;
}

int hgibbs_cov_update(hgibbs_t h, int c, double dgamma)
{
    if (!h || !h->covX) return fail("hgibbs_cov_update: no covariates set");
    if (c < 0 || c >= h->C) return fail("hgibbs_cov_update: covariate %d outside [0,%d)", c, h->C);
    HIP_TRY(hipSetDevice(h->device));
    k_cov_update<<<h->n_pad / BLOCK_IND, BLOCK, 0, h->stream>>>(h->covX + (size_t)c * h->n_pad, h->eps[h->eps_cur], dgamma);
    HIP_TRY(hipGetLastError());
    return 0;
}

int hgibbs_set_model(hgibbs_t h, int G, int K, const int32_t* groups_host, const double* cVa_host, const double* cVaI_host)
{
    if (!h || !h->bed) return fail("hgibbs_set_model: load data first");
    if (G < 1 || K < 2 || K > MAX_K) return fail("hgibbs_set_model: need G>=1 and 2<=K<=%d (got G=%d K=%d)", MAX_K, G, K);
    HIP_TRY(hipSetDevice(h->device));
    h->G = G;
    h->K = K;
    h->groups_host.assign(h->M, 0);
    if (groups_host) {
        for (uint32_t i = 0; i < h->M; ++i) {
            if (groups_host[i] < 0 || groups_host[i] >= G) return fail("hgibbs_set_model: groups[%u]=%d outside [0,%d)", i, groups_host[i], G);
            h->groups_host[i] = groups_host[i];
        }
    }
    HIP_TRY(hipMemcpy(h->groups, h->groups_host.data(), (size_t)h->M * sizeof(int32_t), hipMemcpyHostToDevice));
    h->cVa.assign(cVa_host, cVa_host + (size_t)G * K);
    h->cVaI.assign(cVaI_host, cVaI_host + (size_t)G * K);
    if (h->cass) HIP_TRY(hipFree(h->cass));
    if (h->tables) HIP_TRY(hipFree(h->tables));
    HIP_TRY(hipMalloc(&h->cass, (size_t)G * K * sizeof(int32_t)));
    HIP_TRY(hipMalloc(&h->tables, (size_t)4 * G * K * sizeof(double)));
    return 0;
}

int hgibbs_set_beta(hgibbs_t h, const double* beta_host)
{
    if (!h || !h->bed) return fail("hgibbs_set_beta: no data loaded");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipMemcpy(h->beta, beta_host, (size_t)h->M * sizeof(double), hipMemcpyHostToDevice));
    return 0;
}

int hgibbs_set_components(hgibbs_t h, const int32_t* components_host)
{
    if (!h || !h->bed || !components_host) return fail("hgibbs_set_components: bad argument");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipMemcpy(h->comp, components_host, (size_t)h->M * sizeof(int32_t), hipMemcpyHostToDevice));
    return 0;
}

int hgibbs_get_beta(hgibbs_t h, double* beta_host, int32_t* components_host, double* acum_host)
{
    if (!h || !h->bed) return fail("hgibbs_get_beta: no data loaded");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (beta_host) HIP_TRY(hipMemcpy(beta_host, h->beta, (size_t)h->M * sizeof(double), hipMemcpyDeviceToHost));
    if (components_host) HIP_TRY(hipMemcpy(components_host, h->comp, (size_t)h->M * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (acum_host) HIP_TRY(hipMemcpy(acum_host, h->acum, (size_t)h->M * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}

int hgibbs_beta_sqnorm(hgibbs_t h, double* bsq_host)
{
    if (!h || !h->bed || h->G < 1) return fail("hgibbs_beta_sqnorm: model not set");
    HIP_TRY(hipSetDevice(h->device));
    // M doubles cross PCIe once per iteration (they are needed on the host for
    // the .bet output anyway); the sum then runs sequentially in marker order,
    // exactly as src/BayesRRm.cpp:2496-2499.
    if (!h->beta_host) HIP_TRY(hipHostMalloc(&h->beta_host, (size_t)h->M * sizeof(double)));
    HIP_TRY(hipMemcpyAsync(h->beta_host, h->beta, (size_t)h->M * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    for (int g = 0; g < h->G; ++g) bsq_host[g] = 0.0;
    for (uint32_t i = 0; i < h->M; ++i) bsq_host[h->groups_host[i]] += h->beta_host[i] * h->beta_host[i];
    return 0;
}

int hgibbs_set_option(hgibbs_t h, const char* name, int64_t value)
{
    if (!h || !name) return fail("hgibbs_set_option: null argument");
    // an option that shapes the batch engine's launches asks for that engine (unless the caller has chosen one)
    for (const char* o : {"cols_per_group", "slices", "gram_missing", "graph", "max_seg", "ext_limit", "gram", "carry", "ahead", "force_split", "chunk"})
        if (!std::strcmp(name, o) && h->engine == 0) h->engine_pinned = true;
    if (!std::strcmp(name, "batch") && value != 0 && h->engine == 0) h->engine_pinned = true; // a batch width (0 = auto names none)
    if (!std::strcmp(name, "batch")) {
        if (value < 0 || value > MAX_BATCH) return fail("batch must be in [0,%d] (0 = auto)", MAX_BATCH);
        h->batch = (uint32_t)value;
    } else if (!std::strcmp(name, "cols_per_group")) {
        if (value != 2 && value != 4 && value != 8 && value != 16) return fail("cols_per_group must be 2, 4, 8 or 16");
        h->cols_per_group = (uint32_t)value;
    } else if (!std::strcmp(name, "slices")) {
        if (value < 0 || value > S_CAP) return fail("slices must be in [0,%d] (0 = auto)", S_CAP);
        h->slices = (uint32_t)value;
    } else if (!std::strcmp(name, "gram_missing")) {
        h->gram_missing = (int)value;
    } else if (!std::strcmp(name, "graph")) {
        h->use_graph = value != 0;
    } else if (!std::strcmp(name, "max_seg")) {
        if (value < 0 || value > MAX_SEG) return fail("hgibbs_set_option: max_seg %lld outside [0,%d]", (long long)value, MAX_SEG);
        h->max_seg = (uint32_t)value;
    } else if (!std::strcmp(name, "ext_limit")) {
        if (value < 0 || value > MAX_BATCH) return fail("ext_limit must be in [0,%d]", MAX_BATCH);
        h->ext_limit = (uint32_t)value;
    } else if (!std::strcmp(name, "gram")) {
        h->gram = value != 0;
    } else if (!std::strcmp(name, "carry")) {
        h->carry_on = value < 0 ? -1 : (value != 0 ? 1 : 0);
    } else if (!std::strcmp(name, "ahead")) {
        if (value < -1 || value > AHEAD_MAX) return fail("ahead must be in [-1,%d] (-1 = auto)", AHEAD_MAX);
        h->ahead = (int)value;

    } else if (!std::strcmp(name, "engine")) {
        if (value < 0 || value > 2) return fail("engine must be 0 (auto), 1 (batch) or 2 (resident)");
        h->engine = (int)value;
    } else if (!std::strcmp(name, "window")) {
        if (value != 0 && (value < 8 || value > RS_BMAX || (value & (value - 1)))) return fail("window must be 0 (auto) or a power of two in [8,%d]", RS_BMAX);
        h->window = (uint32_t)value;
    } else if (!std::strcmp(name, "pivots")) {
        h->res_pivots = value != 0;
    } else if (!std::strcmp(name, "early_advance")) {
        if (value < 0 || value > 255) return fail("early_advance must be in [0,255]");
        h->res_early = (int)value;
    } else if (!std::strcmp(name, "refill")) {
        if (value < 0 || value > 2) return fail("refill must be 0 (auto), 1 (fused multiply-adds) or 2 (integer matrix products)");
        h->res_refill = (int)value;
    } else if (!std::strcmp(name, "walker2_ranks")) {
        h->res_walker2_ranks = value != 0;
    } else if (!std::strcmp(name, "window_end16")) {
        h->res_wend = value != 0;
    } else if (!std::strcmp(name, "announce")) {
        h->res_announce = value != 0;
    } else if (!std::strcmp(name, "res_tune")) {
        h->res_tune = (int)value;
    } else if (!std::strcmp(name, "walker")) {
        if (value < 0 || value > 2) return fail("walker must be 0 (auto), 1 (first) or 2 (second)");
        h->res_walker = (int)value;
    } else if (!std::strcmp(name, "res_cus")) {
        if (value < 0 || value > 4096) return fail("res_cus must be in [0,4096]");
        h->res_cus = (uint32_t)value;
    } else if (!std::strcmp(name, "res_deadline_ms")) {
        h->res_deadline_s = (double)value * 1e-3;
    } else if (!std::strcmp(name, "res_timeout_ms")) {
        if (value < 1) return fail("res_timeout_ms must be positive");
        h->res_timeout_s = (double)value * 1e-3;
    } else if (!std::strcmp(name, "p2p")) {
        h->p2p_enabled = value != 0;
    } else if (!std::strcmp(name, "force_split")) {
        h->force_split = value != 0;
    } else if (!std::strcmp(name, "w_kernel_timing")) {
        h->w_kernel_timing = value != 0;
    } else if (!std::strcmp(name, "debug_timing")) {
        h->debug_timing = value != 0;
    } else if (!std::strcmp(name, "chunk")) {
        h->chunk = (int)value;
    } else {
        return fail("hgibbs_set_option: unknown option '%s'", name);
    }
    return 0;
}

/* stage timestamps (100 MHz ticks) of the LAST launch that did work: [0] first
 * workgroup at loop entry, [1] last arriver past the ticket, [2] partials
 * reduced, [3] posteriors done, [4] descriptor written.  Diagnostic only. */
int hgibbs_debug_times(hgibbs_t h, uint64_t* out8)
{
    if (!h || !out8) return fail("null argument");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipMemcpy(out8, h->dbg, 48 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemset(h->dbg, 0, 48 * sizeof(unsigned long long)));
    return 0;
}

int hgibbs_resident_trace(hgibbs_t h, uint64_t* out, uint64_t words)
{
    if (!h || !out) return fail("null argument");
    HIP_TRY(hipSetDevice(h->device));
    const uint64_t n = std::min<uint64_t>(words, (uint64_t)10 * RS_TRACE);
    HIP_TRY(hipMemcpy(out, h->res_trace, n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return 0;
}

int hgibbs_stream_ceiling(hgibbs_t h, uint64_t bytes, int reps, double* gbps)
{
    if (!h || !gbps || bytes < (1u << 20) || reps < 1) return fail("hgibbs_stream_ceiling: bad argument");
    HIP_TRY(hipSetDevice(h->device));
    void *a = nullptr, *b = nullptr;
    HIP_TRY(hipMalloc(&a, bytes));
    HIP_TRY(hipMalloc(&b, bytes));
    HIP_TRY(hipMemsetAsync(a, 1, bytes, h->stream));
    HIP_TRY(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, h->stream)); // warm-up
    HIP_TRY(hipEventRecord(h->ev0, h->stream));
    for (int i = 0; i < reps; ++i) HIP_TRY(hipMemcpyAsync((i & 1) ? a : b, (i & 1) ? b : a, bytes, hipMemcpyDeviceToDevice, h->stream));
    HIP_TRY(hipEventRecord(h->ev1, h->stream));
    HIP_TRY(hipEventSynchronize(h->ev1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    HIP_TRY(hipFree(a));
    HIP_TRY(hipFree(b));
    *gbps = 2.0 * (double)bytes * reps / (ms * 1e-3) / 1e9;
    return 0;
}

int hgibbs_last_sweep_stats(hgibbs_t h, hgibbs_sweep_stats* out)
{
    if (!h || !out) return fail("null argument");
    *out = h->stats;
    return 0;
}

} // extern "C"

// ---- the resident engine (hg_resident.hip.h) -------------------------------------------------------------------
struct ResPlan {
    bool ok = false;
    int T = 1;        // wave tiles per streaming workgroup
    uint32_t W = 0;   // streaming workgroups (+ 1 walker)
    uint32_t B = 0;   // window
};

static int resident_refill(const hgibbs_ctx* h);
// nullptr when the resident engine can run this handle's sweeps, else the reason why not
static const char* resident_plan(hgibbs_ctx* h, ResPlan* pl)
{
    pl->ok = false;
    if (h->nranks > 1 && !(h->p2p_ready && h->p2p_enabled)) return "several ranks without peer mailboxes (hgibbs_p2p_import): the RCCL / host exchange lives in the batch engine";
    if (h->nranks > RX_MAXR) return "more than eight ranks";
    if (h->force_split) return "force_split";
    if (h->G * h->K > 256 || h->K > MAX_K || h->K < 2) return "mixture size";
    const uint32_t cus = h->res_cus ? std::min<uint32_t>(h->res_cus, (uint32_t)h->num_cu) : (uint32_t)h->num_cu;
    if (cus < 2) return "fewer than two compute units";
    const uint32_t ntile = h->n_pad / TILE;
    uint32_t T = (ntile + (cus - 1) - 1) / (cus - 1);
    if (T > (uint32_t)RS_TMAX) {
        // four tiles per workgroup: the second form of the streaming workgroups only (eps is 2 T doubles per lane there; the window
        // shrinks to 128 columns to make room for their codes), one rank, no predicted pivots, a residual inside the digits' range
        const bool four = T <= (uint32_t)RL_TMAX && h->nranks <= 1 && resident_refill(h) == 2 && (h->res_refill == 2 || h->eps_abs_bound < 16.0); // (half the range of T <= 2)
        if (!four) return T <= (uint32_t)RL_TMAX ? "more than 2048 individuals per compute unit need the second form of the streaming workgroups on one rank (option refill, |eps| < 32)"
                                                  : "more individuals than the compute units hold (4096 each)";
        T = (uint32_t)RL_TMAX;
    }
    if ((h->stride & 1023u) || (uint64_t)h->M * (h->stride >> 10) >= (1ull << 32)) return "a shard's BED columns are addressed in 32 bits of KiB";
    pl->T = (int)T;
    pl->W = (ntile + T - 1) / T;
    uint32_t B = h->window ? h->window : (uint32_t)RS_BMAX;
    while (B * T > 512u) B >>= 1;
    pl->B = B;
    pl->ok = true;
    return nullptr;
}

// the streaming workgroups' form of this handle's resident sweeps (option refill; env HGIBBS_REFILL for handles that do not set it)
static int resident_refill(const hgibbs_ctx* h)
{
    static const int env_refill = std::getenv("HGIBBS_REFILL") ? std::atoi(std::getenv("HGIBBS_REFILL")) : 0;
    const int want = h->res_refill ? h->res_refill : env_refill;
    if (want == 0 && h->res_pivots) return 1; // (predicted pivots -- off by default -- take their Gram terms in the first form's refill only)
    // the second form holds eps as round(eps 2^44) in seven signed digits: |eps| < 64 (RL_EX, hg_streamer2.hip.h).  Left to itself the
    // library takes the first form for a sweep that starts with a residual beyond 32 (or with no bound at all); asked for by option, the
    // kernel's own check refuses such a sweep (error 5)
    if (want == 0 && !(h->eps_abs_bound < 32.0)) return 1;
    return want == 1 ? 1 : 2;
}
static size_t resident_streamer_lds(const hgibbs_ctx* h, const ResPlan& pl)
{
    if (resident_refill(h) == 2 || pl.T == RL_TMAX) return rl_streamer_lds(pl.B, pl.T);
    return h->any_missing ? rs_streamer_lds_miss(pl.B, pl.T) : rs_streamer_lds(pl.B, pl.T);
}

// the resident kernel of a plan (T tiles per workgroup, stage clocks, missing-call build), with its LDS opt-in made on this handle's device
static int resident_kernel(hgibbs_ctx* h, const ResPlan& pl, void (**out)(ResParams, const ResParams*))
{
    void (*kern)(ResParams, const ResParams*) = nullptr;
    const bool dbg = h->debug_timing;
    const bool miss = h->any_missing; // the build that keeps s2 per column and the four-term Gram sums
    const bool limb = resident_refill(h) == 2 || pl.T == RL_TMAX;
    if (limb) {
        switch (pl.T) {
        case 4: kern = miss ? (dbg ? k_sweep_limb4<1, 1> : k_sweep_limb4<0, 1>) : (dbg ? k_sweep_limb4<1, 0> : k_sweep_limb4<0, 0>); break;
        case 1: kern = miss ? (dbg ? k_sweep_limb<1, 1, 1> : k_sweep_limb<1, 0, 1>) : (dbg ? k_sweep_limb<1, 1, 0> : k_sweep_limb<1, 0, 0>); break;
        default: kern = miss ? (dbg ? k_sweep_limb<2, 1, 1> : k_sweep_limb<2, 0, 1>) : (dbg ? k_sweep_limb<2, 1, 0> : k_sweep_limb<2, 0, 0>); break;
        }
    } else {
        switch (pl.T) {
        case 1: kern = miss ? (dbg ? k_sweep_resident<1, 1, 1> : k_sweep_resident<1, 0, 1>) : (dbg ? k_sweep_resident<1, 1, 0> : k_sweep_resident<1, 0, 0>); break;
        default: kern = miss ? (dbg ? k_sweep_resident<2, 1, 1> : k_sweep_resident<2, 0, 1>) : (dbg ? k_sweep_resident<2, 1, 0> : k_sweep_resident<2, 0, 0>); break;
        }
    }
    const int ai = (pl.T == RL_TMAX ? 16 : (limb ? 8 : 0) + (pl.T == 1 ? 0 : 1) * 4) + (dbg ? 2 : 0) + (miss ? 1 : 0);
    if (!h->res_attr_set[ai]) { // (per handle: the attribute belongs to the function ON A DEVICE)
        HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        h->res_attr_set[ai] = true;
    }
    *out = kern;
    return 0;
}

// Several ranks: is this rank's resident grid resident at once, and at the same time as every peer's?  A launch of the sweep kernel that
// ends behind its start-of-kernel rendezvous and the walkers' handshake through the mailboxes (rs_probe_peers; M = 0xffffffff -- nothing
// else runs), then one scalar all-reduce: the ranks run the resident engine only if every rank says yes.  Asked once per grid size.  (One
// rank finds out by itself, inside the sweep's own launch: hgibbs_sweep falls back then.)  Returns non-zero when some rank's grid is not
// resident -- e.g. ranks that share one device and whose kernels the device runs one after the other.
static int resident_probe(hgibbs_ctx* h, const ResPlan& pl)
{
    void (*kern)(ResParams, const ResParams*) = nullptr;
    if (resident_kernel(h, pl, &kern)) return 1;
    ResParams p{};
    p.W = pl.W;
    p.B = pl.B;
    p.M = 0xffffffffu; // (probe: see k_sweep_resident)
    p.state = h->res_state;
    p.progress = h->res_progress;
    // (the ranks launch their probes within the skew of a host-side all-reduce -- but a rank's first launch of the kernel may load its code
    // object first: the peers' handshake is given a second, not the 0.1 s of a grid's own rendezvous)
    p.rdv_timeout = (unsigned long long)(std::min(h->res_timeout_s, 1.0) * 1e8);
    p.nranks = h->nranks > 1 ? h->nranks : 1;
    p.rank = h->nranks > 1 ? h->rank : 0;
    for (int r = 0; r < RX_MAXR; ++r) p.mbox[r] = (h->nranks > 1 && r < h->nranks) ? (unsigned char*)h->peer_base[r] + MBOX_RES_OFF : nullptr;
    p.sweep_id = ++h->res_probe_id; // (the ranks probe alike: the handshake's words of an earlier probe never match)
    const size_t lds = std::max(resident_streamer_lds(h, pl), std::max(rs_walker_lds(pl.B), rs_walker2_lds(pl.B)));
    if (hipMemsetAsync(h->res_progress, 0, 16 * sizeof(unsigned long long), h->stream) != hipSuccess || hipMemsetAsync(h->res_state, 0, sizeof(ResState), h->stream) != hipSuccess) return 1;
    kern<<<dim3(pl.W + 1), RS_BLOCK, lds, h->stream>>>(p, h->res_params);
    if (hipGetLastError() != hipSuccess) return 1;
    if (hipMemcpyAsync(h->res_state_host, h->res_state, sizeof(ResState), hipMemcpyDeviceToHost, h->stream) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess) return 1;
    h->scratch_host[0] = h->res_state_host->error ? 1.0 : 0.0;
    if (hipMemcpyAsync(h->sums, h->scratch_host, sizeof(double), hipMemcpyHostToDevice, h->stream) != hipSuccess) return 1;
    if (bulk_allreduce(h, h->sums, 1, 0)) return 1;
    if (hipMemcpyAsync(h->scratch_host, h->sums, sizeof(double), hipMemcpyDeviceToHost, h->stream) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess) return 1;
    if (h->scratch_host[0] != 0.0) {
        h->res_not_resident = true;
        return 1;
    }
    h->res_probed_w = pl.W + 1; // (this grid has been seen resident together with every peer's: later sweeps of the same plan launch without asking again)
    return 0;
}

static int sweep_resident(hgibbs_ctx* h, const ResPlan& pl, double sigmaE, hgibbs_rng_state* rng, int32_t* cass_host, uint64_t* nnz_updates)
{
    const int G = h->G, K = h->K;
    static const bool timing = std::getenv("HGIBBS_TIMING") != nullptr;
    const auto t_0 = std::chrono::steady_clock::now();
    double r0[2];
    if (reduce_eps_all(h, r0)) return 1; // (sum, sum of squares) over all individuals
    ResParams p{};
    p.bed = h->bed;
    p.stride = h->stride;
    p.eps = h->eps[h->eps_cur];
    p.n_pad = h->n_pad;
    p.n_local = h->n_local;
    p.M = h->M;
    p.n_minus_1 = (double)(h->n_global - 1);
    p.n_total = (double)h->n_global;
    p.eps_sum = r0[0]; // s2 of every column (none has missing calls here), see hgibbs_sweep
    p.order = h->order;
    p.s_mave = h->s_mave;
    p.s_mstd = h->s_mstd;
    p.s_bold = h->s_bold;
    p.s_ga = h->s_ga;
    p.beta = h->beta;
    p.comp = h->comp;
    p.acum = h->acum;
    p.cass = h->cass;
    p.K = K;
    p.GK = G * K;
    p.denom = h->tables;
    p.logpi = h->tables + (size_t)G * K;
    p.hlog = h->tables + (size_t)2 * G * K;
    p.sdk = h->tables + (size_t)3 * G * K;
    p.i_2sigE = 1.0 / (2.0 * sigmaE);
    p.mt = h->mt;
    p.zig = ZigTables{h->zig, h->zig + 129, h->zig + 258, h->zig + 515};
    p.rng_idx = rng->idx;
    p.W = pl.W;
    p.B = pl.B;
    p.nsh = std::min<uint32_t>(RS_NSH, pl.W);
    p.rsh = std::min<uint32_t>(RS_RSH, pl.W);
    p.gacc = reinterpret_cast<uint32_t*>(h->res_acc);
    p.racc = reinterpret_cast<unsigned long long*>(h->res_acc + RES_GACC_BYTES);
    p.rcnt = reinterpret_cast<uint32_t*>(h->res_acc + RES_GACC_BYTES + RES_RACC_BYTES);
    p.pacc = reinterpret_cast<unsigned long long*>(h->res_acc + RES_GACC_BYTES + RES_RACC_BYTES + RES_RCNT_BYTES);
    p.racc2 = reinterpret_cast<unsigned long long*>(h->res_acc + RES_RACC2_OFF);
    p.gacc64 = reinterpret_cast<unsigned long long*>(h->res_acc + RES_RACC2_OFF + RES_RACC_BYTES);
    p.counts = h->counts;
    p.msg = h->res_msg;
    p.state = h->res_state;
    {
        // A workgroup's part of a raw dot travels as a fixed-point integer of at most 51 bits (units of 1 / fx_scale), the sum over the
        // workgroups as a wrapping 64-bit integer.  |sum_i g_i eps_i| <= 2 sqrt(n sum eps^2) over the workgroup's n = 1024 T
        // individuals (Cauchy-Schwarz, with the sum of squares of ALL individuals); the scale leaves a factor 8 for what the
        // sweep's own updates add, and a contribution that would not fit is refused by the kernel (error 5), never wrapped.
        // Resolution: 1 / fx_scale (~1e-10 at config 4 against dots of order 10^2..10^3)
        // (several ranks: the same scale everywhere -- the largest workgroup any rank may have; the sum of squares is over all ranks)
        const double bound = 16.0 * std::sqrt(1024.0 * (h->nranks > 1 ? RS_TMAX : pl.T) * std::max(r0[1], 1e-300)) + 1.0;
        int ex = 50 - (int)std::ceil(std::log2(bound));
        ex = std::max(-40, std::min(ex, 60));
        p.fx_scale = std::ldexp(1.0, ex);
        p.fx_unscale = std::ldexp(1.0, -ex);
    }
    p.timeout = (unsigned long long)(h->res_timeout_s * 1e8);
    p.rdv_timeout = (unsigned long long)(std::min(h->res_timeout_s, 0.1) * 1e8); // the grid's workgroups start within microseconds of each other -- or not at all
    p.dbg = h->debug_timing ? 1 : 0;
    p.pivots = (h->nranks > 1 || h->any_missing || resident_refill(h) == 2 || pl.T == RL_TMAX) ? 0 : h->res_pivots; // (the pivot terms have no cross-rank exchange and no four-term form)
    p.nranks = h->nranks > 1 ? h->nranks : 1;
    p.rank = h->nranks > 1 ? h->rank : 0;
    for (int r = 0; r < RX_MAXR; ++r) p.mbox[r] = (h->nranks > 1 && r < h->nranks) ? (unsigned char*)h->peer_base[r] + MBOX_RES_OFF : nullptr;
    p.all_ada = h->res_all_ada ? 1 : 0;
    {
        // the second walker: every marker takes a uniform, the mixture tables and the tabulated bound fit its LDS, one rank
        const bool w2_ok = h->res_all_ada && p.GK <= HT_LDS && G <= RS_FG && (p.nranks == 1 || h->res_walker2_ranks);
        if (h->res_walker == 2 && !w2_ok) return fail("hgibbs_sweep: the second walker does not apply (frozen markers, more than %d groups or %d table entries, or several ranks with walker2_ranks = 0)", RS_FG, HT_LDS);
        static const int env_walker = std::getenv("HGIBBS_WALKER") ? std::atoi(std::getenv("HGIBBS_WALKER")) : 0; // (test runs: the default walker of handles that do not set the option)
        const int want = h->res_walker ? h->res_walker : env_walker;
        p.walker = (want != 1 && w2_ok) ? 2 : 1; // (auto: the second walker where it applies)
    }
    p.pred = h->pred;
    p.tune = h->res_tune;
    p.early_advance = p.nranks > 1 ? 0 : h->res_early; // (several ranks: the replicas must send the same messages -- an advance that depends on when the dots arrive would not be)
    p.announce = h->res_announce;
    p.wend_mask = (p.walker == 2 && pl.B >= 32u && h->res_wend) ? 15u : 0u;
    {
        // the predicted events of this sweep, in sweep order (read by the streaming workgroups and by the walker)
        const uint32_t nchunk = (h->M + PRED_CHUNK - 1) / PRED_CHUNK;
        k_pred_count<<<nchunk, 256, 0, h->stream>>>(h->s_bold, h->M, h->pred_cnt);
        k_pred_scan<<<1, 1024, 0, h->stream>>>(h->pred_cnt, nchunk);
        k_pred_scatter<<<nchunk, 256, 0, h->stream>>>(h->s_bold, h->M, h->pred_cnt, nchunk, h->pred);
        HIP_TRY(hipGetLastError());
    }
    p.sweep_id = ++h->res_sweep_id; // every rank runs the same sweeps on the resident engine (agreed in hgibbs_sweep): the counters stay equal
    p.trace = h->res_trace;
    p.progress = h->res_progress;
    HIP_TRY(hipMemsetAsync(h->res_progress, 0, 16 * sizeof(unsigned long long), h->stream));

    HIP_TRY(hipMemsetAsync(h->res_acc, 0, RES_ACC_BYTES, h->stream));
    HIP_TRY(hipMemsetAsync(h->res_msg, 0, RS_MSG * sizeof(ResMsg), h->stream));
    HIP_TRY(hipMemsetAsync(h->res_state, 0, sizeof(ResState), h->stream));
    const size_t lds = std::max(resident_streamer_lds(h, pl), p.walker == 2 ? rs_walker2_lds(pl.B) : rs_walker_lds(pl.B));
    void (*kern)(ResParams, const ResParams*) = nullptr;
    if (resident_kernel(h, pl, &kern)) return 1;
    {
        // every workgroup of the grid waits for the others: all of them must be resident at once
        int per_cu = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, RS_BLOCK, lds));
        if (per_cu < 1 || (uint64_t)per_cu * (uint64_t)h->num_cu < (uint64_t)pl.W + 1)
            return fail("hgibbs_sweep: the resident grid of %u workgroups does not fit the device (%d per compute unit, %d units)", pl.W + 1, per_cu, h->num_cu);
    }
    if (std::getenv("HGIBBS_DEBUG"))
        std::fprintf(stderr, "[hgibbs] resident sweep: T %d, %u streaming workgroups, window %u, LDS %zu B, fixed-point scale 2^%d\n", pl.T, pl.W, pl.B, lds, (int)std::log2(p.fx_scale));
    HIP_TRY(hipEventRecord(h->ev0, h->stream));
    *h->res_params_host = p;
    HIP_TRY(hipMemcpyAsync(h->res_params, h->res_params_host, sizeof(ResParams), hipMemcpyHostToDevice, h->stream));
    kern<<<dim3(pl.W + 1), RS_BLOCK, lds, h->stream>>>(p, h->res_params);
    HIP_TRY(hipGetLastError());
    k_res_finish<<<dim3((h->M + 255u) / 256u), 256, 0, h->stream>>>(p); // numerators -> Acum, components and cass of the markers that were no event
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(h->ev1, h->stream));
    HIP_TRY(hipMemcpyAsync(h->res_state_host, h->res_state, sizeof(ResState), hipMemcpyDeviceToHost, h->stream));
    {
        // the kernel bounds every wait of its own; the host's deadline is the last line of defence (a kernel that does not come
        // back is reported with where its walker and its first streaming workgroup stand, not waited for)
        // A healthy sweep keeps moving: the deadline counts from the walker's last sign of life (its round counter, fetched on a second
        // stream once a second), not from the launch -- a dense model on a throttled device may legitimately take minutes.
        auto t_start = std::chrono::steady_clock::now();
        auto t_probe = t_start;
        unsigned long long last_round = ~0ull;
        const double limit_s = h->res_deadline_s > 0.0 ? h->res_deadline_s : 30.0 + 20.0 * h->res_timeout_s;
        for (;;) {
            const hipError_t q = hipStreamQuery(h->stream);
            if (q == hipSuccess) break;
            if (q != hipErrorNotReady) return fail("hgibbs_sweep: resident engine: %s", hipGetErrorString(q));
            const auto now = std::chrono::steady_clock::now();
            const double el = std::chrono::duration<double>(now - t_start).count();
            if (std::chrono::duration<double>(now - t_probe).count() > std::min(1.0, 0.25 * limit_s)) {
                t_probe = now;
                if (hipMemcpyAsync(h->res_progress_host, h->res_progress, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->aux_stream) == hipSuccess &&
                    hipStreamSynchronize(h->aux_stream) == hipSuccess && (h->res_progress_host[0] >> 8) != last_round) {
                    last_round = h->res_progress_host[0] >> 8;
                    t_start = now; // the walker has moved on
                    continue;
                }
            }
            if (el > limit_s) {
                // no progress for limit_s: tell the kernel to give up (its workgroups poll the word wherever they wait), give it the time its own
                // bounded waits need, and report where it stood; a kernel that does not even do that has taken the stream with it
                const unsigned long long* pr = h->res_progress_host;
                const unsigned long long w0 = pr[0], w1 = pr[1];
                h->res_progress_host[8] = 3ull;
                (void)hipMemcpyAsync(h->res_progress + 2, h->res_progress_host + 8, sizeof(unsigned long long), hipMemcpyHostToDevice, h->aux_stream);
                (void)hipStreamSynchronize(h->aux_stream);
                const auto t_abort = std::chrono::steady_clock::now();
                bool back = false;
                while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t_abort).count() < 5.0 + 3.0 * h->res_timeout_s) {
                    if (hipStreamQuery(h->stream) != hipErrorNotReady) {
                        back = true;
                        break;
                    }
                    std::this_thread::sleep_for(std::chrono::milliseconds(2));
                }
                if (!back) h->res_dead = true;
                return fail("hgibbs_sweep: the resident kernel made no progress for %.0f s: walker at round %llu stage %llu, streaming workgroup 0 at message %llu stage %llu; it was told to stop and %s", el,
                            (unsigned long long)(w0 >> 8), (unsigned long long)(w0 & 255u), (unsigned long long)(w1 >> 8), (unsigned long long)(w1 & 255u),
                            back ? "did (the sweep's results are void)" : "did NOT come back: this handle cannot be used any more");
            }
            if (el > 0.002) std::this_thread::sleep_for(std::chrono::microseconds(el > 0.5 ? 2000 : 50));
        }
    }
    const auto t_kernel = std::chrono::steady_clock::now();
    const ResState& st = *h->res_state_host;
    if (st.error == 6u) { // the grid was only partly resident (another process or stream on the device): nothing has been touched
        h->res_not_resident = true;
        fail("hgibbs_sweep: the resident grid of %u workgroups was not resident at once (the device is shared)", pl.W + 1);
        return 2;
    }
    if (st.error)
        return fail("hgibbs_sweep: resident engine abort code %u at cursor %u (2 = rng staging overrun, 3 = a workgroup timed out, 5 = raw dot outside the fixed-point range)", st.error, st.cursor);
    if (st.cursor != h->M) return fail("hgibbs_sweep: resident engine stopped at cursor %u of %u", st.cursor, h->M);
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    rng->idx = st.rng_idx;
    HIP_TRY(hipMemcpy(rng->x, h->mt, MT_N * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (cass_host) HIP_TRY(hipMemcpy(cass_host, h->cass, (size_t)G * K * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (nnz_updates) *nnz_updates = st.nnz;
    hgibbs_sweep_stats& s = h->stats;
    s = hgibbs_sweep_stats{};
    s.launches = 1;
    s.nnz_updates = st.nnz;
    s.device_ms = ms;
    s.working_launches = st.rounds;
    s.kernel_ms_avg = st.rounds ? ms / (double)st.rounds : 0.0;
    s.accepted_markers = st.cursor;
    s.streamed_columns = h->M;
    s.tiles_per_workgroup_min = s.tiles_per_workgroup_max = (uint32_t)pl.T;
    s.engine = 2;
    s.walker = (uint32_t)p.walker;
    s.refill = (uint32_t)(pl.T == RL_TMAX ? 2 : resident_refill(h));
    s.rounds = st.rounds;
    s.events = st.events;
    s.advances = st.advances;
    s.chunks = st.chunks;
    s.refolds = st.refolds;
    s.pivots = st.pivots;
    s.predicted = st.predicted;
    for (int i = 0; i < 16; ++i) s.ticks[i] = st.t[i];
    s.shader_mhz = st.wall_ticks ? 100.0 * (double)st.shader_ticks / (double)st.wall_ticks : 0.0;
    {
        double r1[2];
        if (reduce_eps_all(h, r1)) return 1;
        s.eps_sum_drift = std::fabs(r1[0] - r0[0]);
    }
    if (timing) {
        auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        std::fprintf(stderr, "[hgibbs] resident sweep: host until the kernel is back %.3f ms (device %.3f), results + drift %.3f\n", ms(t_0, t_kernel), (double)s.device_ms,
                     ms(t_kernel, std::chrono::steady_clock::now()));
    }
    return 0;
}

extern "C" {

int hgibbs_sweep(hgibbs_t h, const int32_t* order_host, double sigmaE, const double* sigmaG_host, const double* estPi_host,
                 const uint8_t* adaV_host, hgibbs_rng_state* rng, int32_t* cass_host, uint64_t* nnz_updates)
{
    if (!h || !h->bed) return fail("hgibbs_sweep: no data loaded");
    if (h->res_dead) return fail("hgibbs_sweep: a resident kernel of this handle never came back: the handle cannot be used any more");
    if (h->G < 1) return fail("hgibbs_sweep: model not set");
    if (!order_host || !sigmaG_host || !estPi_host || !adaV_host || !rng) return fail("hgibbs_sweep: null argument");
    const auto t_prep0 = std::chrono::steady_clock::now();
    HIP_TRY(hipSetDevice(h->device));
    if (compute_stats(h)) return 1;
    const int G = h->G, K = h->K;
    const uint32_t M = h->M;
    for (uint32_t i = 0; i < M; ++i)
        if (order_host[i] < 0 || (uint32_t)order_host[i] >= M) return fail("hgibbs_sweep: order[%u]=%d outside [0,%u)", i, order_host[i], M);
    if (rng->idx > (uint32_t)MT_N) return fail("hgibbs_sweep: rng idx %u > 624", rng->idx);

    // per-sweep hyper tables (the marker-independent factors of src/BayesRRm.cpp:1721-1723,1750,1875,1901)
    const double dNm1 = (double)(h->n_global - 1);
    std::vector<double> tab((size_t)4 * G * K, 0.0);
    double* denom = tab.data();
    double* logpi = denom + (size_t)G * K;
    double* hlog = logpi + (size_t)G * K;
    double* sdk = hlog + (size_t)G * K;
    for (int g = 0; g < G; ++g) {
        const double sigE_G = sigmaE / sigmaG_host[g];
        const double sigG_E = sigmaG_host[g] / sigmaE;
        for (int k = 0; k < K; ++k) {
            logpi[g * K + k] = log(estPi_host[g * K + k]);
            if (k >= 1) {
                denom[g * K + k] = dNm1 + sigE_G * h->cVaI[g * K + k];
                hlog[g * K + k] = 0.5 * log(sigG_E * dNm1 * h->cVa[g * K + k] + 1.0);
                sdk[g * K + k] = sqrt(sigmaE / denom[g * K + k]);
            }
        }
    }
    HIP_TRY(hipMemcpyAsync(h->tables, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->order, order_host, (size_t)M * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->adaV, adaV_host, (size_t)M, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->mt, rng->x, MT_N * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemsetAsync(h->cass, 0, (size_t)G * K * sizeof(int32_t), h->stream));
    HIP_TRY(hipMemsetAsync(h->ticket, 0, (16 + MAX_GROUPS + 256) * sizeof(uint32_t), h->stream));
    HIP_TRY(hipMemsetAsync(h->aticket, 0, (AHEAD_MAX / 2 + 4) * sizeof(uint32_t), h->stream));
    k_gather_meta<<<(M + 255) / 256, 256, 0, h->stream>>>(h->order, h->mave, h->mstd, h->beta, h->groups, h->adaV, h->counts, h->s_mave, h->s_mstd,
                                                       h->s_bold, h->s_ga, M);
    HIP_TRY(hipGetLastError());

    const uint32_t cpg = h->cols_per_group;
    const uint32_t batch = h->batch ? h->batch : ((h->n_local >= 20000u || h->nranks > 1) ? 256u : 128u);
    const uint32_t ngroups = (batch + cpg - 1) / cpg;
    SweepDesc d0{};
    d0.cursor = 0;
    for (int q = 0; q < MAX_SEG; ++q) {
        d0.pend_marker[q] = -1;
        d0.seg_end[q] = batch; // first launch: one segment, planned blind; its tail plans the rest
    }
    d0.cur = h->eps_cur;
    d0.rng_idx = rng->idx;
    d0.seq = h->batch_seq;
    *h->desc_host = d0;
    SweepCounters* const cnt_host = reinterpret_cast<SweepCounters*>(h->desc_host + 1);
    *cnt_host = SweepCounters{};
    cnt_host->tiles_min = 0xffffffffu;
    HIP_TRY(hipMemcpyAsync(h->desc, h->desc_host, sizeof(SweepDesc) + sizeof(SweepCounters), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream)); // staging buffers are on the host stack / pageable

    // which engine runs this sweep
    ResPlan plan{};
    {
        const char* why = resident_plan(h, &plan);
        if (h->engine == 1 || (h->engine == 0 && h->engine_pinned)) plan.ok = false;
        if (h->engine == 0 && h->res_not_resident && plan.ok) {
            plan.ok = false;
            why = "an earlier resident grid was not resident at once (the device is shared)";
        }
        if (h->nranks > 1) {
            // the ranks run ONE engine: the resident one only if every rank can (a rank with a larger shard, another option or no
            // mailbox would otherwise wait for peers that are in the other engine's exchange)
            h->scratch_host[0] = plan.ok ? 0.0 : 1.0;
            HIP_TRY(hipMemcpyAsync(h->sums, h->scratch_host, sizeof(double), hipMemcpyHostToDevice, h->stream));
            if (bulk_allreduce(h, h->sums, 1, 0)) return 1;
            HIP_TRY(hipMemcpyAsync(h->scratch_host, h->sums, sizeof(double), hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(hipStreamSynchronize(h->stream));
            if (h->scratch_host[0] != 0.0) {
                if (plan.ok) why = "another rank cannot run it";
                plan.ok = false;
            }
            // (every rank has the same answer now) a rank whose grid is not resident at once must be known BEFORE its peers wait for it
            if (plan.ok && h->res_probed_w != plan.W + 1 && resident_probe(h, plan)) {
                plan.ok = false;
                why = "some rank's resident grid is not resident at once (a shared device)";
            }
        }
        if (h->engine == 2 && !plan.ok) return fail("hgibbs_sweep: the resident engine does not apply: %s", why ? why : "the batch engine was asked for by an option");
        if (std::getenv("HGIBBS_DEBUG"))
            std::fprintf(stderr, "[hgibbs] engine: %s%s%s\n", plan.ok ? "resident" : "batch", why ? " -- resident refused: " : "", why ? why : "");
    }
    if (std::getenv("HGIBBS_TIMING"))
        std::fprintf(stderr, "[hgibbs] sweep preparation (checks, tables, order / adaV upload, metadata gather) %.3f ms\n",
                     std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_prep0).count());
    h->res_all_ada = std::memchr(adaV_host, 0, (size_t)M) == nullptr;
    if (plan.ok) {
        const int rc = sweep_resident(h, plan, sigmaE, rng, cass_host, nnz_updates);
        // 2: the grid was found partly resident at the kernel's start (a shared device) and NOTHING was touched: with engine = 0 this sweep and
        // the following ones run on the batch engine (several ranks have agreed on that before the launch: resident_probe)
        if (rc != 2 || h->engine == 2 || h->nranks > 1) return rc ? 1 : 0;
        if (std::getenv("HGIBBS_DEBUG")) std::fprintf(stderr, "[hgibbs] engine: the resident grid was not resident at once -- this sweep and the following ones run on the batch engine\n");
    }

    SweepParams p{};
    p.bed = h->bed;
    p.stride = h->stride;
    p.eps0 = h->eps[0];
    p.eps1 = h->eps[1];
    p.n_pad = h->n_pad;
    p.n_local = h->n_local;
    p.M = M;
    p.n_minus_1 = dNm1;
    p.n_total = (double)h->n_global;
    double eps_sum_start = 0.0;
    {
        // s2 = sum_i nm_i eps_i of a column WITHOUT missing calls is the plain sum of eps (src/BayesRRm.cpp:1788 with
        // every nm_i = 1).  A marker update adds mstd (g_i - mave) nm_i dbeta to eps_i, and mave is the mean of g over the
        // marker's non-missing calls, so the update's sum over i is zero: the sum of eps is the same before and after
        // every update of the sweep up to the rounding of the adds (~1e-16 of |eps|, random sign; far below the 1e-9
        // the dots are held to).  It is therefore reduced once per sweep, in fixed order and over all ranks, instead of
        // once per launch by the streaming loop.  Columns with missing calls keep their own masked sum per marker.
        double r[2];
        if (reduce_eps_all(h, r)) return 1;
        p.eps_sum = r[0];
        eps_sum_start = r[0];
    }
    p.gram = h->gram ? 1 : 0;
    p.order = h->order;
    p.beta = h->beta;
    p.comp = h->comp;
    p.acum = h->acum;
    p.cass = h->cass;
    p.K = K;
    p.GK = G * K;
    p.s_mave = h->s_mave;
    p.s_mstd = h->s_mstd;
    p.s_bold = h->s_bold;
    p.s_ga = h->s_ga;
    p.dbg = (h->debug_timing && h->cols_per_group == 4) ? h->dbg : nullptr;
    p.denom = h->tables;
    p.logpi = h->tables + (size_t)G * K;
    p.hlog = h->tables + (size_t)2 * G * K;
    p.sdk = h->tables + (size_t)3 * G * K;
    p.i_2sigE = 1.0 / (2.0 * sigmaE);
    p.mt = h->mt;
    p.zig = ZigTables{h->zig, h->zig + 129, h->zig + 258, h->zig + 515};
    p.desc = h->desc;
    p.counters = reinterpret_cast<SweepCounters*>(h->desc + 1);
    p.partials = h->partials;
    p.ticket = h->ticket;
    p.gticket = h->ticket + 16;
    p.entered = h->ticket + 16 + MAX_GROUPS;
    p.totals = h->totals;
    p.carry = h->carry;
    p.ahead_raw = h->ahead_raw;
    p.apartials = h->apartials;
    p.aticket = h->aticket;
    p.aqueue = h->aticket + AHEAD_MAX / 2;
    // the carry term is a 16-bit field per lane like the other Gram partials; a carried column group may run on a single slice
    // carried dots pay where re-streaming a column costs more than the Gram-only group, its reductions and the extra work of
    // the group stage and of the draw phase: measured (round 2) -2 % at N = 200 K, -1.5 % at 350 K, +1 % at 500 K, +4 % at
    // 500 K with missing calls in every column (the fresh dot of such a column costs twice as much), -10 % at 50 K
    const bool carry_auto = h->n_local >= 400000u;
    p.carry_on = ((h->carry_on < 0 ? carry_auto : h->carry_on != 0) && h->gram) ? 1u : 0u; // (16-bit Gram fields: p.gram falls to 0 where they could overflow)
    p.ahead_cols = (uint32_t)(h->ahead >= 0 ? h->ahead : 0); // columns streamed ahead of the batch behind the hand-off (needs carry; off on the split path)
    p.cols_per_group = cpg;
    p.batch_cap = ngroups * cpg;
    p.batch_limit = batch;
    p.ext_limit = h->ext_limit;
    p.max_seg = h->max_seg ? std::min<uint32_t>(h->max_seg, MAX_SEG) : ((h->n_local <= (h->nranks > 1 ? 400000u : 300000u)) ? 4u : 2u);
    // two builds of the kernel: tier 2 (one Gram term, two pending updates: lean registers, 3 workgroups per CU) and
    // tier 4 (three Gram terms, four pending updates; cols_per_group 4 or 8 only)
    // data with missing calls: the two-segment build that carries the four Gram terms (A, B, C, D) takes such columns
    // through the extension; the four-segment build would stop its chain at the first of them
    // (that costs three more popcount sums per extension column and word, and a column with missing calls needs its own
    // s2 pass anyway: measured +10 % at N = 100 K whatever the share of such columns, +2 % at N = 500 K when every column
    // has missing calls -- the worst case, where the streaming loop dominates the launch)
    const bool mg_wanted = h->gram_missing != 0;
    const bool mg = mg_wanted && h->gram && h->any_missing && (cpg == 4 || cpg == 8) && (h->max_seg == 0 || h->max_seg == 2);
    const int tier = (!mg && p.max_seg > 2 && h->gram && (cpg == 4 || cpg == 8)) ? 4 : 2;
    if (p.max_seg > (uint32_t)tier) p.max_seg = (uint32_t)tier;
    const int nr = sweep_rows(tier, mg ? 1 : 0);
    const size_t lds = sweep_lds_bytes(p.batch_cap, cpg, K, nr);
    // 160 KiB of LDS per compute unit, handed out in 1280-byte granules: three workgroups fit only up to 42 granules each
    if (std::getenv("HGIBBS_DEBUG")) std::fprintf(stderr, "[hgibbs] sweep LDS %zu B = %zu granules of 1280 B (3 per CU up to 42)\n", lds, (lds + 1279) / 1280);
    const bool use_p2p = h->nranks > 1 && h->p2p_ready && h->p2p_enabled && !h->force_split;
    const bool split = (h->nranks > 1 && !use_p2p) || h->force_split;
    if (h->nranks > 1 && split && !h->comm && !h->ext_fn)
        return fail("hgibbs_sweep: %d ranks but no per-batch transport (hgibbs_comm_init for RCCL, hgibbs_comm_init_external or hgibbs_p2p_import)", h->nranks);
    p.sums_out = split ? h->sums : nullptr;
    p.p2p.nranks = use_p2p ? h->nranks : 0;
    p.p2p.rank = h->rank;
    for (int r = 0; r < MAX_RANKS; ++r) {
        p.p2p.data[r] = (use_p2p && r < h->nranks) ? (double*)h->peer_base[r] : nullptr;
        p.p2p.flags[r] = (use_p2p && r < h->nranks) ? (unsigned long long*)((unsigned char*)h->peer_base[r] + MBOX_DATA_BYTES) : nullptr;
    }

    const uint32_t ntg = h->n_pad / BLOCK_IND;
    const uint32_t S = std::min<uint32_t>(h->slices ? h->slices : S_CAP, ntg);
    p.slices_max = S;
    // most groups a launch can have: the update group, Gram-only groups of carried columns, fresh groups (a batch is at most
    // `batch` columns, split between the last two kinds)
    const uint32_t ccg = (uint32_t)carried_cpg(tier, mg ? 1 : 0);
    const uint32_t groups_max = 1u + (batch + ccg - 1) / ccg + ngroups;
    if (groups_max > (uint32_t)MAX_GROUPS || (uint64_t)groups_max * group_rows((int)cpg, tier, mg ? 1 : 0) > (uint64_t)PROWS_CAP)
        return fail("hgibbs_sweep: %u groups of %d partial rows exceed the partial buffer", groups_max, group_rows((int)cpg, tier, mg ? 1 : 0));
    uint64_t total_launches = 0;
    // the build of the kernel this sweep runs
    void (*kern)(SweepParams) = nullptr;
    const bool nomiss = !h->any_missing;
    const bool dbg = h->debug_timing && cpg == 4; // stage timestamps exist in the builds of the default column count only
    if (mg) {
        kern = (cpg == 4) ? (dbg ? k_sweep_batch<4, 2, 1, 0, 1> : k_sweep_batch<4, 2, 1>) : k_sweep_batch<8, 2, 1>;
    } else if (tier == 4) {
        if (cpg == 4) kern = nomiss ? (dbg ? k_sweep_batch<4, 4, 0, 1, 1> : k_sweep_batch<4, 4, 0, 1>) : (dbg ? k_sweep_batch<4, 4, 0, 0, 1> : k_sweep_batch<4, 4, 0>);
        else kern = k_sweep_batch<8, 4, 0>;
    } else {
        switch (cpg) {
        case 2: kern = k_sweep_batch<2, 2, 0>; break;
        case 4: kern = nomiss ? (dbg ? k_sweep_batch<4, 2, 0, 1, 1> : k_sweep_batch<4, 2, 0, 1>) : (dbg ? k_sweep_batch<4, 2, 0, 0, 1> : k_sweep_batch<4, 2, 0>); break;
        case 8: kern = k_sweep_batch<8, 2, 0>; break;
        default: kern = k_sweep_batch<16, 2, 0>; break;
        }
    }
    // the active workgroups of a launch must be co-resident (a second round of workgroups would double the streaming
    // phase): how many fit depends on the build's registers and on this launch's LDS size (K, batch capacity, rows)
    {
        int per_cu = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, BLOCK, lds));
        if (per_cu < 1) return fail("hgibbs_sweep: the sweep kernel does not fit a compute unit with %zu bytes of LDS", lds);
        p.resident = (uint32_t)per_cu * (uint32_t)h->num_cu;
        if (std::getenv("HGIBBS_DEBUG")) std::fprintf(stderr, "[hgibbs] sweep build: tier %d mg %d cpg %u, LDS %zu B, %d workgroups per CU, %u resident\n", tier, (int)mg, cpg, lds, per_cu, p.resident);
    }
    // the Gram partials are 16-bit fields per lane (64 per tile at most): a launch uses at least
    // min(slices, resident / groups) slices, so a lane sees at most ntg / that many tiles -- refuse the chain of
    // segments where that could overflow
    {
        const uint32_t s_min = std::max<uint32_t>(1u, std::min<uint32_t>(S, p.resident / std::max<uint32_t>(1u, groups_max)));
        if ((ntg + s_min - 1) / s_min > 1000u) p.gram = 0;
    }
    // A launch's active workgroups are the first S * nactive <= resident ones in dispatch order (S is chosen on the device
    // as resident / groups): the grid never needs more than `resident` workgroups.  Workgroups beyond the active ones leave
    // at once, but each still has to be given its LDS first -- thousands of them queue for the slots the active ones
    // hold and keep the kernel alive after its last workgroup has drawn.
    const dim3 grid(std::min<uint32_t>(S * groups_max, p.resident));
    auto launch_one = [&]() { kern<<<grid, BLOCK, lds, h->stream>>>(p); };
    // launch-bound inner loop: the launches of one sweep are identical (all state travels through the
    // descriptor), so GRAPH_N of them are captured once per sweep into a graph and replayed
    constexpr int GRAPH_N = 64;
    hipGraph_t graph = nullptr;
    hipGraphExec_t gexec = nullptr;
    if (h->use_graph && !split) {
        HIP_TRY(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < GRAPH_N; ++i) launch_one();
        HIP_TRY(hipStreamEndCapture(h->stream, &graph));
        HIP_TRY(hipGraphInstantiate(&gexec, graph, nullptr, nullptr, 0));
    }
    HIP_TRY(hipEventRecord(h->ev0, h->stream));
    double avg_accept = std::max(1.0, (double)batch * 0.5);
    for (;;) {
        const SweepDesc& dh = *h->desc_host;
        const uint32_t remaining = M - std::min(M, dh.cursor);
        int n = h->chunk > 0 ? h->chunk : (int)std::min<double>(2048.0, std::max(8.0, 1.25 * remaining / avg_accept + 2));
        if (gexec) {
            n = (n + GRAPH_N - 1) / GRAPH_N * GRAPH_N;
            for (int i = 0; i < n; i += GRAPH_N) HIP_TRY(hipGraphLaunch(gexec, h->stream));
        }
        for (int i = 0; i < n && !gexec; ++i) {
            launch_one();
            if (split) {
                // the exchange step of src/BayesRRm.cpp:2456, on the batch rows: RCCL in-stream, or the caller's transport
                // (hydra's own MPI_Allreduce, say) on a host copy -- a stream round trip per batch, the parity baseline
                if (h->comm) NCCL_TRY(ncclAllReduce(h->sums, h->sums, nr * MAX_BATCH, ncclDouble, ncclSum, h->comm, h->stream));
                else if (h->nranks > 1 && bulk_allreduce(h, h->sums, (size_t)nr * MAX_BATCH, 0)) return 1;
                if (mg) k_sweep_draw<2, 1><<<1, BLOCK, lds, h->stream>>>(p);
                else if (tier == 4) k_sweep_draw<4, 0><<<1, BLOCK, lds, h->stream>>>(p);
                else k_sweep_draw<2, 0><<<1, BLOCK, lds, h->stream>>>(p);
            }
        }
        total_launches += (uint64_t)n;
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(h->desc_host, h->desc, sizeof(SweepDesc) + sizeof(SweepCounters), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        if (dh.error) return fail("hgibbs_sweep: device abort code %u at cursor %u (1 = logL overflow, 2 = rng staging overrun, 3 = peer exchange timed out, 4 = LDS base misaligned)", dh.error, dh.cursor);
        if (cnt_host->launches > 0) avg_accept = std::max(1.0, (double)cnt_host->accepted_sum / (double)cnt_host->launches);
        if (dh.cursor >= M && dh.pend_marker[0] < 0) break;
        // every launch accepts at least one marker or flushes a pending update: a sweep that needs more
        // than 2M launches (plus one chunk of overshoot) is stuck, not slow
        if (total_launches > 2ull * M + 4096)
            return fail("hgibbs_sweep: no progress after %llu launches (cursor %u of %u)", (unsigned long long)total_launches, dh.cursor, M);
    }
    HIP_TRY(hipEventRecord(h->ev1, h->stream));
    HIP_TRY(hipEventSynchronize(h->ev1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    if (gexec) (void)hipGraphExecDestroy(gexec);
    if (graph) (void)hipGraphDestroy(graph);

    h->eps_cur = h->desc_host->cur;
    h->batch_seq = h->desc_host->seq;
    rng->idx = h->desc_host->rng_idx;
    HIP_TRY(hipMemcpy(rng->x, h->mt, MT_N * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (cass_host) HIP_TRY(hipMemcpy(cass_host, h->cass, (size_t)G * K * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (nnz_updates) *nnz_updates = cnt_host->nnz;
    h->stats.launches = total_launches;
    h->stats.nnz_updates = cnt_host->nnz;
    h->stats.carried_columns = cnt_host->carried_sum;
    if (std::getenv("HGIBBS_DEBUG")) std::fprintf(stderr, "[hgibbs] sweep: %llu launches, %llu columns carried\n", (unsigned long long)total_launches, (unsigned long long)cnt_host->carried_sum);
    h->stats.device_ms = ms;
    // launches that did work (the chunks the host enqueues overshoot the end of the sweep by a few launches that find
    // nothing left and return at once): the average below is taken over the working ones
    h->stats.working_launches = cnt_host->launches;
    h->stats.accepted_markers = cnt_host->accepted_sum;
    h->stats.streamed_columns = cnt_host->streamed_sum;
    h->stats.tiles_per_workgroup_min = cnt_host->tiles_max ? cnt_host->tiles_min : 0u;
    h->stats.tiles_per_workgroup_max = cnt_host->tiles_max;
    h->stats.kernel_ms_avg = cnt_host->launches ? ms / (double)cnt_host->launches : 0.0;
    h->stats.engine = 1;
    h->stats.rounds = h->stats.events = h->stats.advances = h->stats.chunks = h->stats.refolds = h->stats.pivots = 0;
    h->stats.shader_mhz = 0.0;
    for (int i = 0; i < 16; ++i) h->stats.ticks[i] = 0;
    {
        // s2 of the columns without missing calls was the sum of eps at sweep start for the whole sweep: what the sum is now says
        // how far the updates' roundings have moved it (src/BayesRRm.cpp:331 re-sums eps per marker)
        double r[2];
        if (reduce_eps_all(h, r)) return 1;
        h->stats.eps_sum_drift = std::fabs(r[0] - eps_sum_start);
    }
    return 0;
}

} // extern "C"

#include "hg_bayesw.hip.h"
