// Diagnostic micro-benchmark: issue cost of the VALU instructions the sweep's inner loop is made of.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 64
template <int MODE>
__global__ void k(double* out, unsigned n, unsigned seed)
{
    double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    unsigned g0 = seed + threadIdx.x, g1 = g0 * 3, g2 = g0 * 5, g3 = g0 * 7;
    double e = 1.0 + threadIdx.x * 1e-9;
    for (unsigned it = 0; it < n; ++it) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
            if (MODE == 0) { // fma only
                asm volatile("v_fmac_f64 %0, %4, %5\n v_fmac_f64 %1, %4, %5\n v_fmac_f64 %2, %4, %5\n v_fmac_f64 %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(e), "v"(e));
            } else if (MODE == 1) { // cvt only
                asm volatile("v_cvt_f64_u32 %0, %4\n v_cvt_f64_u32 %1, %5\n v_cvt_f64_u32 %2, %6\n v_cvt_f64_u32 %3, %7" : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3) : "v"(g0), "v"(g1), "v"(g2), "v"(g3));
            } else if (MODE == 2) { // bfe only
                asm volatile("v_bfe_u32 %0, %4, 4, 2\n v_bfe_u32 %1, %5, 4, 2\n v_bfe_u32 %2, %6, 4, 2\n v_bfe_u32 %3, %7, 4, 2" : "=v"(g0), "=v"(g1), "=v"(g2), "=v"(g3) : "v"(g0), "v"(g1), "v"(g2), "v"(g3));
            } else if (MODE == 3) { // the triple
                unsigned t0, t1, t2, t3; double w0, w1, w2, w3;
                asm volatile("v_bfe_u32 %8, %12, 4, 2\n v_bfe_u32 %9, %13, 4, 2\n v_bfe_u32 %10, %14, 4, 2\n v_bfe_u32 %11, %15, 4, 2\n"
                             "v_cvt_f64_u32 %4, %8\n v_cvt_f64_u32 %5, %9\n v_cvt_f64_u32 %6, %10\n v_cvt_f64_u32 %7, %11\n"
                             "v_fmac_f64 %0, %4, %16\n v_fmac_f64 %1, %5, %16\n v_fmac_f64 %2, %6, %16\n v_fmac_f64 %3, %7, %16"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "=&v"(w0), "=&v"(w1), "=&v"(w2), "=&v"(w3), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
                             : "v"(g0), "v"(g1), "v"(g2), "v"(g3), "v"(e));
            } else if (MODE == 4) { // v_add_f64
                asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(e));
            } else if (MODE == 5) { // v_cvt_f32_u32 + v_cvt_f64_f32
                float f0, f1, f2, f3;
                asm volatile("v_cvt_f32_u32 %4, %8\n v_cvt_f32_u32 %5, %9\n v_cvt_f32_u32 %6, %10\n v_cvt_f32_u32 %7, %11\n"
                             "v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %5\n v_cvt_f64_f32 %2, %6\n v_cvt_f64_f32 %3, %7"
                             : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=&v"(f0), "=&v"(f1), "=&v"(f2), "=&v"(f3) : "v"(g0), "v"(g1), "v"(g2), "v"(g3));
            } else if (MODE == 6) { // and_b32 (full rate int)
                asm volatile("v_and_b32 %0, %0, %4\n v_and_b32 %1, %1, %4\n v_and_b32 %2, %2, %4\n v_and_b32 %3, %3, %4" : "+v"(g0), "+v"(g1), "+v"(g2), "+v"(g3) : "v"(seed));
            } else if (MODE == 7) { // v_ldexp_f64
                asm volatile("v_ldexp_f64 %0, %0, %4\n v_ldexp_f64 %1, %1, %4\n v_ldexp_f64 %2, %2, %4\n v_ldexp_f64 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(g0 & 1));
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + g0 + g1 + g2 + g3;
}
template <int MODE>
void run(const char* name, int waves_per_simd)
{
    double* out; hipMalloc(&out, 1 << 24);
    const unsigned n = 2000;
    dim3 grid(256 * waves_per_simd), block(256); // 256-thread blocks: 4 waves = 1 per SIMD per block
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<grid, block>>>(out, 10, 1);
    hipEventRecord(a); k<MODE><<<grid, block>>>(out, n, 1); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double inst_per_wave = (double)n * REP * (MODE == 3 ? 12 : (MODE == 5 ? 8 : 4));
    // per SIMD: waves_per_simd waves, each inst_per_wave instructions, in ms
    const double ns_per_inst_per_simd = ms * 1e6 / (inst_per_wave * waves_per_simd);
    printf("%-28s waves/SIMD %d: %.3f ns per wave-instruction per SIMD (%.1f cycles @2.4GHz)\n", name, waves_per_simd, ns_per_inst_per_simd, ns_per_inst_per_simd * 2.4);
    hipFree(out);
}
int main()
{
    for (int w : {1, 2, 4}) {
        run<0>("v_fmac_f64", w); run<4>("v_add_f64", w); run<1>("v_cvt_f64_u32", w); run<2>("v_bfe_u32", w); run<6>("v_and_b32", w);
        run<5>("cvt_f32_u32+cvt_f64_f32 (x2)", w); run<7>("v_ldexp_f64", w); run<3>("bfe+cvt+fma triple (x3)", w);
    }
    return 0;
}
