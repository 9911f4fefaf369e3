// One column of the resident engine's refill (hg_resident.hip.h, res_streamer::one at T = 2), rebuilt piece by piece: what each piece costs a wave
// at two waves per SIMD (512-thread workgroups, one per CU), in cycles per column.
//   bit 0: the 96 bfe / cvt / fmac triples + the lane sum     bit 1: the x form to the LDS ring (ds_write2_b32)
//   bit 2: the eight-lane group sums (3 DPP steps) + the store of the group sum     bit 3: the next column's load (global_load_dwordx2, streaming 125 KB columns)
//   bit 4: the scalar address arithmetic of that load (readlane + 64-bit multiply-add)
// build: hipcc --offload-arch=gfx950 -O3 -o refill_col refill_col.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <type_traits>
extern __shared__ unsigned char smem[];
template <int Q>
__device__ __forceinline__ void fma4(uint32_t g, const double* e, double& a0, double& a1, double& a2, double& a3)
{
    uint32_t t0, t1, t2, t3;
    double w0, w1, w2, w3;
    asm("v_bfe_u32 %[t0], %[g], %[s0], 2\n v_bfe_u32 %[t1], %[g], %[s1], 2\n v_bfe_u32 %[t2], %[g], %[s2], 2\n v_bfe_u32 %[t3], %[g], %[s3], 2\n"
        "v_cvt_f64_u32 %[w0], %[t0]\n v_cvt_f64_u32 %[w1], %[t1]\n v_cvt_f64_u32 %[w2], %[t2]\n v_cvt_f64_u32 %[w3], %[t3]\n"
        "v_fmac_f64 %[a0], %[w0], %[e0]\n v_fmac_f64 %[a1], %[w1], %[e1]\n v_fmac_f64 %[a2], %[w2], %[e2]\n v_fmac_f64 %[a3], %[w3], %[e3]"
        : [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [w0] "=&v"(w0), [w1] "=&v"(w1), [w2] "=&v"(w2),
          [w3] "=&v"(w3)
        : [g] "v"(g), [e0] "v"(e[4 * Q]), [e1] "v"(e[4 * Q + 1]), [e2] "v"(e[4 * Q + 2]), [e3] "v"(e[4 * Q + 3]), [s0] "i"(8 * Q), [s1] "i"(8 * Q + 2), [s2] "i"(8 * Q + 4), [s3] "i"(8 * Q + 6));
}
template <int CTRL>
__device__ __forceinline__ double dpp64(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
template <int MODE, int UNR>
__global__ __launch_bounds__(512) void k(unsigned long long* out, double* sink, const uint8_t* bed, size_t stride, const int* order, int ncol, const double* ein)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double e[2][16];
    for (int t = 0; t < 2; ++t)
        for (int s = 0; s < 16; ++s) e[t][s] = ein[(t * 16 + s + lane) & 63];
    uint32_t* ring = reinterpret_cast<uint32_t*>(smem + 16384);
    double* part = reinterpret_cast<double*>(smem);
    const uint32_t voff = (blockIdx.x * 64u + lane) * 8u;
    uint32_t g0 = lane * 2654435761u, g1 = g0 ^ 0x9e3779b9u;
    double pad[80];
    if (MODE & 32)
        for (int i = 0; i < 80; ++i) pad[i] = ein[(i + lane) & 63];
    if (MODE & 64) asm volatile("v_mov_b32 v255, 0" ::: "v255"); // (the wave is allocated all 256 registers: two waves fill the SIMD's file)
    double tot = 0.0;
    unsigned long long c0 = __builtin_amdgcn_s_memtime();
    auto cols = [&](auto copy) __attribute__((always_inline)) {
        asm volatile("; copy %0" ::"i"(decltype(copy)::value));
#pragma unroll UNR
        for (int c = wave; c < ncol; c += 8) {
            double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
            if (MODE & 8) { // the column's dwords: loaded one column-of-this-wave ahead (plain loads: the compiler waits where they are used)
                int id = c;
                if (MODE & 16) id = __builtin_amdgcn_readfirstlane(order[c]);
                const uint2 v = *reinterpret_cast<const uint2*>(bed + (size_t)id * stride + voff);
                g0 ^= v.x;
                g1 ^= v.y;
            }
            if (MODE & 2) {
                uint32_t* rp = ring + (uint32_t)(c & 255) * 128u + lane * 2;
                rp[0] = g0 | ((g0 >> 1) & 0x55555555u);
                rp[1] = g1 | ((g1 >> 1) & 0x55555555u);
            }
            if (MODE & 1) {
                fma4<0>(g0, e[0], a0, a1, a2, a3);
                fma4<1>(g0, e[0], a0, a1, a2, a3);
                fma4<2>(g0, e[0], a0, a1, a2, a3);
                fma4<3>(g0, e[0], a0, a1, a2, a3);
                fma4<0>(g1, e[1], a0, a1, a2, a3);
                fma4<1>(g1, e[1], a0, a1, a2, a3);
                fma4<2>(g1, e[1], a0, a1, a2, a3);
                fma4<3>(g1, e[1], a0, a1, a2, a3);
            }
            double v = (a0 + a1) + (a2 + a3);
            if (MODE & 4) {
                v += dpp64<0xB1>(v);
                v += dpp64<0x4E>(v);
                v += dpp64<0x141>(v);
                if ((lane & 7) == 0) part[(c & 255) * 8 + (lane >> 3)] = v;
            }
            tot += v;
            if (MODE & 32) {
#pragma unroll
                for (int i = 0; i < 80; i += 16) asm volatile("" : "+v"(pad[i]));
            }
            g0 = g0 * 1664525u + 1013904223u; // (another column's codes)
            g1 = g1 * 1664525u + 1013904223u;
        }
    };
    if (MODE & 128) { // every wave runs a copy of its own of the loop: eight instruction streams at eight different addresses
        switch (wave) {
        case 0: cols(std::integral_constant<int, 0>{}); break;
        case 1: cols(std::integral_constant<int, 1>{}); break;
        case 2: cols(std::integral_constant<int, 2>{}); break;
        case 3: cols(std::integral_constant<int, 3>{}); break;
        case 4: cols(std::integral_constant<int, 4>{}); break;
        case 5: cols(std::integral_constant<int, 5>{}); break;
        case 6: cols(std::integral_constant<int, 6>{}); break;
        default: cols(std::integral_constant<int, 7>{}); break;
        }
    } else
        cols(std::integral_constant<int, 8>{});
    unsigned long long c1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) atomicMax(out, c1 - c0); // the SLOWEST wave of the workgroup (the older wave of a SIMD is served first)
    if (MODE & 32)
        for (int i = 0; i < 80; ++i) tot += pad[i];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = tot;
}
template <int MODE, int UNR>
void run(const char* name, const uint8_t* bed, size_t stride, const int* order, int ncol, const double* ein, unsigned long long* out, double* sink)
{
    for (int rep = 0; rep < 2; ++rep) {
        hipMemset(out, 0, 8);
        k<MODE, UNR><<<245, 512, 160 * 1024>>>(out, sink, bed, stride, order, ncol, ein);
        hipDeviceSynchronize();
    }
    unsigned long long c;
    hipMemcpy(&c, out, 8, hipMemcpyDeviceToHost);
    printf("%-70s %.0f cycles per column and wave\n", name, (double)c / (ncol / 8.0));
}
int main()
{
    const int ncol = 16384;
    const size_t stride = (size_t)245 * 64 * 8; // 245 workgroups x 64 lanes x 2 dwords: every lane of the grid reads inside its column
    uint8_t* bed;
    int* order;
    double *ein, *sink;
    unsigned long long* out;
    hipMalloc(&bed, stride * ncol);
    hipMemset(bed, 0x5a, stride * ncol);
    hipMalloc(&order, ncol * 4);
    int* oh = new int[ncol];
    for (int i = 0; i < ncol; ++i) oh[i] = (int)(((long long)i * 7919) % ncol);
    hipMemcpy(order, oh, ncol * 4, hipMemcpyHostToDevice);
    hipMalloc(&ein, 64 * 8);
    {
        double eh[64];
        for (int i = 0; i < 64; ++i) eh[i] = 0.37 * (i - 31.5) + 1e-3 * i * i; // (real residuals, not zeros)
        hipMemcpy(ein, eh, 64 * 8, hipMemcpyHostToDevice);
    }
    hipMalloc(&sink, 8 * 512 * 256);
    hipMalloc(&out, 8);
#define RUN(M, N) RUNU(M, 1, N)
#define RUNU(M, U, N) hipFuncSetAttribute((const void*)k<M, U>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); run<M, U>(N, bed, stride, order, ncol, ein, out, sink);
    RUN(1, "triples + lane sum");
    RUN(3, "+ x form to the LDS ring");
    RUN(7, "+ group sums (DPP) and their store");
    RUN(15, "+ the column's load (in order)");
    RUN(31, "+ shuffled order (readlane, 64-bit address arithmetic)");
    RUN(71, "triples, ring, group sums in a wave that is allocated 256 registers");
    RUN(135, "triples, ring, group sums: every wave in a copy of its own of the code");
    RUNU(135, 16, "... the same, unrolled 16 x");
    RUNU(7, 16, "triples, ring, group sums: body unrolled 16 x (~18 KB of code)");
    RUNU(7, 64, "... unrolled 64 x (~70 KB of code: more than the instruction cache)");
    return 0;
}
