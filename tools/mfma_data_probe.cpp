// Diagnostic: does the rate of v_mfma_f32_16x16x32_f16 depend on the operand DATA when the whole chip runs it?  512 workgroups x 256
// threads (two waves per SIMD on every CU), 16 independent accumulators per wave, operands rotated through 8 register pairs filled
// with: zeros; small "activation-like" values; second-part-like values (~2^-12 of the first); uniform random finite fp16; random bit
// patterns (NaN / Inf / subnormals included).  Reports core clocks per MFMA (s_memtime), wall time per MFMA and the clock the chip held
// (clocks / wall).  Answers whether a k-loop's time follows its MFMAs' switching activity (power) rather than their count.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define MF_V(ACC, A, B) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(ACC) : "v"(A), "v"(B))

__global__ __launch_bounds__(256, 2) void probe(const u32x4* __restrict__ ops, float* out, unsigned long long* stamps, int iters) {
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    u32x4 a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = ops[(i * 2) * 256 + threadIdx.x];
        b[i] = ops[(i * 2 + 1) * 256 + threadIdx.x];
    }
    __syncthreads();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) MF_V(acc[i], a[i & 7], b[(i + (i >> 3)) & 7]);
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) r += acc[i][0] + acc[i][3];
    if (r == 12345.678f) out[0] = r;
    if (threadIdx.x == 0) {
        stamps[2 * blockIdx.x] = c1 - c0;
        stamps[2 * blockIdx.x + 1] = r1 - r0;
    }
}

static unsigned short f2h(float f) { _Float16 h = (_Float16)f; unsigned short u; __builtin_memcpy(&u, &h, 2); return u; }

int main() {
    const int nwg = 512, iters = 40000;
    u32x4* dops; float* dout; unsigned long long* dst;
    (void)hipMalloc(&dops, 16 * 256 * 16); (void)hipMalloc(&dout, 4); (void)hipMalloc(&dst, nwg * 16);
    const char* names[5] = {"zeros", "activation-like N(0,1) values", "second-part-like (2^-12 of N(0,1))", "uniform random finite fp16", "random bit patterns"};
    srand(1);
    for (int mode = 0; mode < 5; ++mode) {
        std::vector<unsigned short> h(16 * 256 * 8);
        for (size_t i = 0; i < h.size(); ++i) {
            float g = 0.f;
            for (int k = 0; k < 12; ++k) g += (float)rand() / RAND_MAX;
            g -= 6.f;
            if (mode == 0) h[i] = 0;
            else if (mode == 1) h[i] = f2h(g);
            else if (mode == 2) h[i] = f2h(g * (1.0f / 4096.0f));
            else if (mode == 3) { unsigned short u = (unsigned short)(rand() & 0xffff); if (((u >> 10) & 31) == 31) u &= 0xbfff; if (((u >> 10) & 31) == 0) u |= 0x0400; h[i] = u; }
            else h[i] = (unsigned short)(rand() & 0xffff);
        }
        (void)hipMemcpy(dops, h.data(), h.size() * 2, hipMemcpyHostToDevice);
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(probe, dim3(nwg), dim3(256), 0, 0, dops, dout, dst, iters);
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> st(nwg * 2);
        (void)hipMemcpy(st.data(), dst, nwg * 16, hipMemcpyDeviceToHost);
        double clk = 0, wall = 0;
        for (int i = 0; i < nwg; ++i) { clk += (double)st[2 * i]; wall += (double)st[2 * i + 1]; }
        clk /= nwg; wall /= nwg;   // wall in 10 ns ticks
        const double n = (double)iters * 16;
        printf("%-36s %6.2f clocks per MFMA per wave (%5.2f per SIMD), %6.2f ns per MFMA per SIMD, clock %.2f GHz\n", names[mode], clk / n, clk / n / 2,
               wall * 10.0 / n / 2, clk / (wall * 10.0));
    }
    return 0;
}
