// The refill's inner block (hg_resident.hip.h, fma_col4): 4 x v_bfe_u32, 4 x v_cvt_f64_u32, 4 x v_fmac_f64 per four genotypes -- cycles per block
// at one and two waves per SIMD, with the interleave depth (4 or 8 independent chains) and the register banks of the f64 operands varied.
// build: hipcc --offload-arch=gfx950 -O3 -o refill_block refill_block.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP 32
template <int V>
__global__ void k(unsigned long long* out, double* sink, int iters, const double* ein)
{
    double e0 = ein[threadIdx.x & 7], e1 = e0 + 1, e2 = e0 + 2, e3 = e0 + 3, e4 = e0 + 4, e5 = e0 + 5, e6 = e0 + 6, e7 = e0 + 7;
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;
    uint32_t g = threadIdx.x * 2654435761u;
    unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
            if (V == 0) { // depth 4, as in the kernel (compiler-chosen registers)
                uint32_t t0, t1, t2, t3;
                double w0, w1, w2, w3;
                asm volatile("v_bfe_u32 %[t0], %[g], 0, 2\n v_bfe_u32 %[t1], %[g], 2, 2\n v_bfe_u32 %[t2], %[g], 4, 2\n v_bfe_u32 %[t3], %[g], 6, 2\n"
                             "v_cvt_f64_u32 %[w0], %[t0]\n v_cvt_f64_u32 %[w1], %[t1]\n v_cvt_f64_u32 %[w2], %[t2]\n v_cvt_f64_u32 %[w3], %[t3]\n"
                             "v_fmac_f64 %[a0], %[w0], %[e0]\n v_fmac_f64 %[a1], %[w1], %[e1]\n v_fmac_f64 %[a2], %[w2], %[e2]\n v_fmac_f64 %[a3], %[w3], %[e3]"
                             : [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [w0] "=&v"(w0), [w1] "=&v"(w1),
                               [w2] "=&v"(w2), [w3] "=&v"(w3)
                             : [g] "v"(g), [e0] "v"(e0), [e1] "v"(e1), [e2] "v"(e2), [e3] "v"(e3));
            }
            if (V == 1) { // depth 8
                uint32_t t0, t1, t2, t3, t4, t5, t6, t7;
                double w0, w1, w2, w3, w4, w5, w6, w7;
                asm volatile("v_bfe_u32 %[t0], %[g], 0, 2\n v_bfe_u32 %[t1], %[g], 2, 2\n v_bfe_u32 %[t2], %[g], 4, 2\n v_bfe_u32 %[t3], %[g], 6, 2\n"
                             "v_bfe_u32 %[t4], %[g], 8, 2\n v_bfe_u32 %[t5], %[g], 10, 2\n v_bfe_u32 %[t6], %[g], 12, 2\n v_bfe_u32 %[t7], %[g], 14, 2\n"
                             "v_cvt_f64_u32 %[w0], %[t0]\n v_cvt_f64_u32 %[w1], %[t1]\n v_cvt_f64_u32 %[w2], %[t2]\n v_cvt_f64_u32 %[w3], %[t3]\n"
                             "v_cvt_f64_u32 %[w4], %[t4]\n v_cvt_f64_u32 %[w5], %[t5]\n v_cvt_f64_u32 %[w6], %[t6]\n v_cvt_f64_u32 %[w7], %[t7]\n"
                             "v_fmac_f64 %[a0], %[w0], %[e0]\n v_fmac_f64 %[a1], %[w1], %[e1]\n v_fmac_f64 %[a2], %[w2], %[e2]\n v_fmac_f64 %[a3], %[w3], %[e3]\n"
                             "v_fmac_f64 %[a4], %[w4], %[e4]\n v_fmac_f64 %[a5], %[w5], %[e5]\n v_fmac_f64 %[a6], %[w6], %[e6]\n v_fmac_f64 %[a7], %[w7], %[e7]"
                             : [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3), [a4] "+v"(a4), [a5] "+v"(a5), [a6] "+v"(a6), [a7] "+v"(a7), [t0] "=&v"(t0), [t1] "=&v"(t1),
                               [t2] "=&v"(t2), [t3] "=&v"(t3), [t4] "=&v"(t4), [t5] "=&v"(t5), [t6] "=&v"(t6), [t7] "=&v"(t7), [w0] "=&v"(w0), [w1] "=&v"(w1), [w2] "=&v"(w2), [w3] "=&v"(w3),
                               [w4] "=&v"(w4), [w5] "=&v"(w5), [w6] "=&v"(w6), [w7] "=&v"(w7)
                             : [g] "v"(g), [e0] "v"(e0), [e1] "v"(e1), [e2] "v"(e2), [e3] "v"(e3), [e4] "v"(e4), [e5] "v"(e5), [e6] "v"(e6), [e7] "v"(e7));
            }
            if (V == 2) { // depth 4, hand-named registers: every fmac's three operands in conflicting banks as in the kernel (acc/e or w/e share banks)
                asm volatile("v_bfe_u32 v99, %[g], 0, 2\n v_bfe_u32 v104, %[g], 2, 2\n v_bfe_u32 v105, %[g], 4, 2\n v_bfe_u32 v106, %[g], 6, 2\n"
                             "v_cvt_f64_u32 v[78:79], v99\n v_cvt_f64_u32 v[80:81], v104\n v_cvt_f64_u32 v[100:101], v105\n v_cvt_f64_u32 v[102:103], v106\n"
                             "v_fmac_f64 v[120:121], v[78:79], v[110:111]\n v_fmac_f64 v[122:123], v[80:81], v[108:109]\n v_fmac_f64 v[126:127], v[100:101], v[114:115]\n v_fmac_f64 v[116:117], v[102:103], v[112:113]"
                             :: [g] "v"(g) : "v99", "v104", "v105", "v106", "v78", "v79", "v80", "v81", "v100", "v101", "v102", "v103", "v120", "v121", "v122", "v123", "v126", "v127", "v116", "v117");
            }
            if (V == 3) { // depth 4, hand-named registers: no two f64 operands of an fmac in the same banks is impossible with 3 x 2 dwords over 4 banks; acc/w share instead
                asm volatile("v_bfe_u32 v99, %[g], 0, 2\n v_bfe_u32 v104, %[g], 2, 2\n v_bfe_u32 v105, %[g], 4, 2\n v_bfe_u32 v106, %[g], 6, 2\n"
                             "v_cvt_f64_u32 v[76:77], v99\n v_cvt_f64_u32 v[80:81], v104\n v_cvt_f64_u32 v[100:101], v105\n v_cvt_f64_u32 v[104:105], v106\n"
                             "v_fmac_f64 v[120:121], v[76:77], v[110:111]\n v_fmac_f64 v[124:125], v[80:81], v[114:115]\n v_fmac_f64 v[128:129], v[100:101], v[118:119]\n v_fmac_f64 v[132:133], v[104:105], v[122:123]"
                             :: [g] "v"(g) : "v99", "v104", "v105", "v106", "v76", "v77", "v80", "v81", "v100", "v101", "v120", "v121", "v124", "v125", "v128", "v129", "v132", "v133");
            }
        }
    }
    unsigned long long c1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) atomicMax(out, c1 - c0); // the SLOWEST wave of the workgroup (the older wave of a SIMD is served first)
    sink[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
template <int V>
void run(const char* name, int per_block)
{
    unsigned long long* out;
    double *sink, *ein;
    hipMalloc(&out, 8);
    hipMalloc(&sink, 8 * 1024 * 512);
    hipMalloc(&ein, 64);
    hipMemset(ein, 0, 64);
    for (int wps = 1; wps <= 2; ++wps) {
        const int threads = 256 * wps, iters = 200;
        k<V><<<256, threads>>>(out, sink, iters, ein);
        hipDeviceSynchronize();
        hipMemset(out, 0, 8);
        k<V><<<256, threads>>>(out, sink, iters, ein);
        hipDeviceSynchronize();
        unsigned long long c;
        hipMemcpy(&c, out, 8, hipMemcpyDeviceToHost);
        printf("%-44s %d wave(s)/SIMD: %.2f cycles per instruction per wave, %.2f per SIMD\n", name, wps, (double)c / (iters * REP * (double)per_block), (double)c / (iters * REP * (double)per_block * wps));
    }
}
int main()
{
    run<0>("depth 4, compiler's registers", 12);
    run<1>("depth 8, compiler's registers", 24);
    run<2>("depth 4, the kernel's bank pattern", 12);
    run<3>("depth 4, accumulator and weight share banks", 12);
    return 0;
}
