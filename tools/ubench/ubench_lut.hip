// Diagnostic micro-benchmark: a pair-table inner step (4-bit code -> LDS table of e0*w0 + e1*w1 -> add)
// against the bfe/cvt/fma triple the sweep uses now.  Cycles per (column, pair of individuals) per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define NCOL 16
template <int MODE>
__global__ __launch_bounds__(256) void k(double* out, const unsigned* cols, unsigned n)
{
    __shared__ double tab[4][16 * 64]; // per wave: [code][lane]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double* T = tab[wave];
    double a[NCOL];
    unsigned w[NCOL];
#pragma unroll
    for (int c = 0; c < NCOL; ++c) {
        a[c] = 0.0;
        w[c] = cols[(blockIdx.x * NCOL + c) * 256 + threadIdx.x];
    }
    double e0 = 1.0 + lane * 1e-3, e1 = 2.0 - lane * 1e-3;
    for (unsigned it = 0; it < n; ++it) {
#pragma unroll
        for (int slot = 0; slot < 8; ++slot) {
            if (MODE == 1) {
                // build the 16-entry table of this pair of individuals
                const double t0[4] = {2.0 * e0, 0.0, e0, 0.0}, t1[4] = {2.0 * e1, 0.0, e1, 0.0};
#pragma unroll
                for (int hi = 0; hi < 4; ++hi)
#pragma unroll
                    for (int lo = 0; lo < 4; ++lo) T[(hi * 4 + lo) * 64 + lane] = t0[lo] + t1[hi];
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int c = 0; c < NCOL; ++c) a[c] += T[((w[c] >> (4 * slot)) & 15u) * 64 + lane];
            } else {
#pragma unroll
                for (int c = 0; c < NCOL; ++c) {
                    a[c] = __builtin_fma((double)((w[c] >> (4 * slot)) & 3u), e0, a[c]);
                    a[c] = __builtin_fma((double)((w[c] >> (4 * slot + 2)) & 3u), e1, a[c]);
                }
            }
            e0 += 1e-9;
            e1 -= 1e-9;
        }
#pragma unroll
        for (int c = 0; c < NCOL; ++c) w[c] = w[c] * 1664525u + 1013904223u;
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < NCOL; ++c) s += a[c];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE>
void run(const char* name, int wps)
{
    double* out; unsigned* cols;
    const int blocks = 256 * wps;
    hipMalloc(&out, (size_t)blocks * 256 * 8); hipMalloc(&cols, (size_t)blocks * NCOL * 256 * 4);
    hipMemset(cols, 0x5a, (size_t)blocks * NCOL * 256 * 4);
    const unsigned n = 400;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<blocks, 256>>>(out, cols, 4);
    hipEventRecord(a); k<MODE><<<blocks, 256>>>(out, cols, n); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    // per SIMD: wps waves, each n * 8 slots * NCOL (column, pair) steps
    const double steps = (double)n * 8 * NCOL * wps;
    printf("%-22s waves/SIMD %d: %.2f cycles per (column, pair of individuals) per SIMD @2.4GHz\n", name, wps, ms * 1e6 / steps * 2.4);
    hipFree(out); hipFree(cols);
}
int main()
{
    for (int w : {1, 2, 3}) { run<0>("bfe+cvt+fma x2", w); run<1>("pair table (LDS)", w); }
    return 0;
}
