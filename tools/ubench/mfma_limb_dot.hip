// The refill's dot products as ONE integer matrix product ("limb dots"): sixteen columns of 2-bit codes against
// eps held as a fixed-point integer of seven signed 8-bit digits, on v_mfma_i32_16x16x64_i8 -- exact, and a fraction of the vector
// instructions of the per-individual convert + fused multiply-add form.  This program checks the operand maps the kernel relies on with
// random data against the host's integer sums (A: lane (j, g) = digit j of sixteen individuals of k-group g; B: lane (c, g) = column c, the
// codes of the SAME sixteen individuals; D: lane (c, g) register r = digit 4 g + r of column c -- a lane puts its four digits together
// itself) and times the inner step.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_limb_dot mfma_limb_dot.hip ; run: ./mfma_limb_dot
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
constexpr int STEPS = 4; // k-steps per wave and group of sixteen columns: 4 x 64 = 256 individuals

// sixteen 2-bit codes -> four dwords of bytes: dword r, byte i = the code of individual 4 i + r
__device__ __forceinline__ v4i expand16(uint32_t x)
{
    v4i z;
    z.x = (int)(x & 0x03030303u);
    z.y = (int)((x >> 2) & 0x03030303u);
    z.z = (int)((x >> 4) & 0x03030303u);
    z.w = (int)((x >> 6) & 0x03030303u);
    return z;
}

// codes: [16 columns][STEPS][4 groups] dwords; digits: [STEPS][4 groups][16 lanes j] v4i (digit j of the sixteen individuals, bytes in the
// order of expand16); out: [16 columns] 64-bit sums
__global__ void k_dot(const uint32_t* codes, const v4i* digits, long long* out, int reps, unsigned long long* clocks)
{
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    v4i acc = {0, 0, 0, 0};
    uint32_t x[STEPS];
    v4i b[STEPS];
    for (int s = 0; s < STEPS; ++s) {
        x[s] = codes[(c * STEPS + s) * 4 + g];
        b[s] = digits[(s * 4 + g) * 16 + c]; // (as the A operand: lane (j = c, g))
    }
    const unsigned long long t0 = wall_clock64();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int s = 0; s < STEPS; ++s) acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(b[s], expand16(x[s]), acc, 0, 0, 0);
        if (r + 1 < reps) { // (keep the loop honest: the codes change, the sum of the last pass is what is checked)
            acc = v4i{0, 0, 0, 0};
#pragma unroll
            for (int s = 0; s < STEPS; ++s) x[s] = x[s] * 1u + 0u;
        }
    }
    const unsigned long long t1 = wall_clock64();
    // lane (c, g): digits 4 g .. 4 g + 3 of column c: d0 + 2^8 d1 + 2^16 d2 + 2^24 d3, times 2^32 in k-group 1 (groups 2, 3: zeros)
    {
        const int t01 = acc[0] + (acc[1] << 8), t23 = acc[2] + (acc[3] << 8);
        long long v = ((long long)t01 + ((long long)t23 << 16)) << (32 * (g & 1));
        if (g >= 2) v = 0;
        v += __shfl_xor(v, 16, 64);
        if (g == 0) out[c] = v;
    }
    if (lane == 0) clocks[0] = t1 - t0;
}

int main()
{
    const int NI = 64 * STEPS; // individuals
    std::vector<uint32_t> codes(16 * STEPS * 4);
    std::vector<int8_t> code_of(16 * NI);
    std::vector<long long> eps(NI);
    srand(7);
    for (int c = 0; c < 16; ++c)
        for (int s = 0; s < STEPS; ++s)
            for (int g = 0; g < 4; ++g) {
                uint32_t x = 0;
                for (int b = 0; b < 16; ++b) {
                    const uint32_t cd = (uint32_t)(rand() % 4);
                    x |= cd << (2 * b);
                    code_of[c * NI + (s * 4 + g) * 16 + b] = (int8_t)cd;
                }
                codes[(c * STEPS + s) * 4 + g] = x;
            }
    for (int i = 0; i < NI; ++i) eps[i] = ((long long)rand() << 20 ^ (long long)rand()) % (1ll << 50) * ((rand() & 1) ? 1 : -1);
    // signed digits: y = x + 0x008080808080 (six low bytes biased), d_j = (y_j ^ 0x80) as int8 for j < 6, d_6 = byte 6 of y as int8
    std::vector<int8_t> dig((size_t)STEPS * 4 * 16 * 16, 0);
    for (int s = 0; s < STEPS; ++s)
        for (int g = 0; g < 4; ++g)
            for (int b = 0; b < 16; ++b) {
                const long long x = eps[(s * 4 + g) * 16 + b];
                const unsigned long long y = (unsigned long long)x + 0x0000808080808080ull;
                for (int j = 0; j < 7; ++j) {
                    const uint8_t yb = (uint8_t)(y >> (8 * j));
                    const int8_t d = (int8_t)(j < 6 ? (yb ^ 0x80u) : yb);
                    const int r = b & 3, i = b >> 2; // byte i of dword r (expand16's order)
                    dig[(((size_t)(s * 4 + g) * 16 + j) * 16) + r * 4 + i] = d;
                }
            }
    uint32_t* d_codes;
    v4i* d_dig;
    long long* d_out;
    unsigned long long* d_clk;
    hipMalloc(&d_codes, codes.size() * 4);
    hipMalloc(&d_dig, dig.size());
    hipMalloc(&d_out, 16 * 8);
    hipMalloc(&d_clk, 8);
    hipMemcpy(d_codes, codes.data(), codes.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_dig, dig.data(), dig.size(), hipMemcpyHostToDevice);
    k_dot<<<1, 64>>>(d_codes, d_dig, d_out, 1, d_clk);
    long long out[16];
    hipMemcpy(out, d_out, sizeof(out), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int c = 0; c < 16; ++c) {
        long long ref = 0;
        for (int i = 0; i < NI; ++i) ref += (long long)code_of[c * NI + i] * eps[i];
        if (ref != out[c]) {
            ++bad;
            printf("column %d: device %lld host %lld\n", c, out[c], ref);
        }
    }
    printf("sixteen columns x %d individuals, seven signed digits: %s\n", NI, bad ? "MISMATCH" : "exact");
    const int reps = 4096;
    k_dot<<<1, 64>>>(d_codes, d_dig, d_out, reps, d_clk);
    unsigned long long clk;
    hipMemcpy(&clk, d_clk, 8, hipMemcpyDeviceToHost);
    printf("one wave alone: %.1f clocks (100 MHz ticks x 24) per step of expand + MFMA (= 16 columns x 64 individuals)\n", (double)clk * 24.0 / reps / STEPS);
    return bad ? 1 : 0;
}
