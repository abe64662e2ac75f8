// Diagnostic: v_mfma_f32_16x16x32_bf16 issue rate of ONE wave per SIMD as a function of how many independent accumulators are
// cycled through (distance between two MFMAs that accumulate into the same registers), accumulators in VGPRs (what hipcc
// picks when a kernel fits 256 registers) or pinned to AGPRs.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC, bool AGPR>
__global__ __launch_bounds__(256, 1) void probe(float* out, long long* clk, int iters) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3f80 + threadIdx.x + i); b[i] = (short)(0x3f00 + 3 * threadIdx.x + i); }
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                if (AGPR) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
                else acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
            }
    }
    const long long t1 = clock64();
    float r = 0.f;
    for (int i = 0; i < NACC; ++i) r += acc[i][0] + acc[i][2];
    if (r == 12345.678f) out[0] = r;
    if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = t1 - t0;
}

template <int NACC, bool AGPR>
static void run() {
    float* d; long long* c; (void)hipMalloc(&d, 4); (void)hipMalloc(&c, 8);
    const int iters = 20000;
    hipLaunchKernelGGL((probe<NACC, AGPR>), dim3(256), dim3(256), 0, 0, d, c, iters);
    hipLaunchKernelGGL((probe<NACC, AGPR>), dim3(256), dim3(256), 0, 0, d, c, iters);
    (void)hipDeviceSynchronize();
    long long h = 0; (void)hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    printf("%2d accumulators, %s: %6.2f clocks per MFMA\n", NACC, AGPR ? "AGPR" : "VGPR", (double)h / (iters * 2.0 * NACC));
    (void)hipFree(d); (void)hipFree(c);
}

int main() {
    run<1, false>(); run<1, true>();
    run<2, false>(); run<2, true>();
    run<3, false>(); run<3, true>();
    run<4, false>(); run<4, true>();
    run<6, false>(); run<6, true>();
    run<8, false>(); run<8, true>();
    run<12, false>(); run<12, true>();
    return 0;
}
