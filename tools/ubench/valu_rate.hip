// Issue cost of the streaming loop's instructions on gfx950: cycles per wave-instruction, 1 and 2 waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 64
template <int OP>
__global__ void k(unsigned long long* out, double* sink, int iters)
{
    double a0 = threadIdx.x, a1 = 1.5, a2 = 2.5, a3 = 3.5, e = 1.000001;
    uint32_t g = threadIdx.x * 2654435761u, t0, t1, t2, t3;
    double w0 = 0, w1 = 0, w2 = 0, w3 = 0;
    unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
            if (OP == 0) asm volatile("v_bfe_u32 %0, %4, 2, 2\n v_bfe_u32 %1, %4, 4, 2\n v_bfe_u32 %2, %4, 6, 2\n v_bfe_u32 %3, %4, 8, 2" : "=v"(t0), "=v"(t1), "=v"(t2), "=v"(t3) : "v"(g));
            if (OP == 1) asm volatile("v_cvt_f64_u32 %0, %4\n v_cvt_f64_u32 %1, %4\n v_cvt_f64_u32 %2, %4\n v_cvt_f64_u32 %3, %4" : "=v"(w0), "=v"(w1), "=v"(w2), "=v"(w3) : "v"(g));
            if (OP == 2) asm volatile("v_fmac_f64 %0, %4, %5\n v_fmac_f64 %1, %4, %5\n v_fmac_f64 %2, %4, %5\n v_fmac_f64 %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(w0), "v"(e));
            if (OP == 3) asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(e));
            if (OP == 4) asm volatile("v_mul_u32_u24 %0, %4, %4\n v_mul_u32_u24 %1, %4, %4\n v_mul_u32_u24 %2, %4, %4\n v_mul_u32_u24 %3, %4, %4" : "=v"(t0), "=v"(t1), "=v"(t2), "=v"(t3) : "v"(g));
            if (OP == 5) asm volatile("v_cndmask_b32 %0, %4, %4, vcc\n v_cndmask_b32 %1, %4, %4, vcc\n v_cndmask_b32 %2, %4, %4, vcc\n v_cndmask_b32 %3, %4, %4, vcc" : "=v"(t0), "=v"(t1), "=v"(t2), "=v"(t3) : "v"(g) : "vcc");
            if (OP == 6) asm volatile("v_cvt_f64_i32 %0, %4\n v_cvt_f64_i32 %1, %4\n v_cvt_f64_i32 %2, %4\n v_cvt_f64_i32 %3, %4" : "=v"(w0), "=v"(w1), "=v"(w2), "=v"(w3) : "v"(g));
            if (OP == 7) asm volatile("v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %4\n v_cvt_f64_f32 %2, %4\n v_cvt_f64_f32 %3, %4" : "=v"(w0), "=v"(w1), "=v"(w2), "=v"(w3) : "v"(g));
            if (OP == 8) asm volatile("v_dot4_i32_i8 %0, %4, %4, %0\n v_dot4_i32_i8 %1, %4, %4, %1\n v_dot4_i32_i8 %2, %4, %4, %2\n v_dot4_i32_i8 %3, %4, %4, %3" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3) : "v"(g));
            if (OP == 9) asm volatile("v_mul_f64 %0, %4, %5\n v_mul_f64 %1, %4, %5\n v_mul_f64 %2, %4, %5\n v_mul_f64 %3, %4, %5" : "=v"(w0), "=v"(w1), "=v"(w2), "=v"(w3) : "v"(a0), "v"(e));
            if (OP == 10) asm volatile("v_ldexp_f64 %0, %4, %5\n v_ldexp_f64 %1, %4, %5\n v_ldexp_f64 %2, %4, %5\n v_ldexp_f64 %3, %4, %5" : "=v"(w0), "=v"(w1), "=v"(w2), "=v"(w3) : "v"(a0), "v"(g));
            if (OP == 11) asm volatile("v_bcnt_u32_b32 %0, %4, %0\n v_bcnt_u32_b32 %1, %4, %1\n v_bcnt_u32_b32 %2, %4, %2\n v_bcnt_u32_b32 %3, %4, %3" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3) : "v"(g));
        }
    }
    unsigned long long c1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) atomicMax(out, c1 - c0); // the SLOWEST wave of the workgroup (the older wave of a SIMD is served first)
    sink[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + w0 + w1 + w2 + w3 + t0 + t1 + t2 + t3;
}
template <int OP>
void run(const char* name)
{
    unsigned long long* out;
    double* sink;
    hipMalloc(&out, 8);
    hipMalloc(&sink, 8 * 1024 * 512);
    for (int waves_per_simd = 1; waves_per_simd <= 2; ++waves_per_simd) {
        const int threads = 256 * waves_per_simd, iters = 200;
        k<OP><<<256, threads>>>(out, sink, iters);
        hipDeviceSynchronize();
        hipMemset(out, 0, 8);
        k<OP><<<256, threads>>>(out, sink, iters);
        hipDeviceSynchronize();
        unsigned long long c;
        hipMemcpy(&c, out, 8, hipMemcpyDeviceToHost);
        printf("%-16s %d wave(s)/SIMD: %.2f cycles per wave-instruction (per SIMD: %.2f)\n", name, waves_per_simd, (double)c / (iters * REP * 4.0), (double)c / (iters * REP * 4.0 * waves_per_simd));
    }
}
int main()
{
    run<0>("v_bfe_u32");
    run<1>("v_cvt_f64_u32");
    run<2>("v_fmac_f64");
    run<3>("v_add_f64");
    run<4>("v_mul_u32_u24");
    run<5>("v_cndmask_b32");
    run<6>("v_cvt_f64_i32");
    run<7>("v_cvt_f64_f32");
    run<8>("v_dot4_i32_i8");
    run<9>("v_mul_f64");
    run<10>("v_ldexp_f64");
    run<11>("v_bcnt_u32_b32");
    return 0;
}
