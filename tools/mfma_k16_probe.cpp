// Diagnostic: cycles per v_mfma_f32_16x16x16_f16 vs v_mfma_f32_16x16x32_f16 on gfx950 (AGPR accumulators, 8 independent
// chains, 2 waves per SIMD so the pipe is saturated; s_memtime).  Decides whether conv_0 (K = 9) should use the K = 16 form.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

template <int K32>
__global__ __launch_bounds__(256, 2) void probe(float* out, long long* clk, int iters) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(1 + (threadIdx.x + i) % 7); b[i] = (_Float16)(1 + (3 * threadIdx.x + i) % 5); }
    f16x4 a4 = {a[0], a[1], a[2], a[3]}, b4 = {b[0], b[1], b[2], b[3]};
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            if (K32) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[t]) : "v"(a), "v"(b));
            else asm volatile("v_mfma_f32_16x16x16_f16 %0, %1, %2, %0" : "+a"(acc[t]) : "v"(a4), "v"(b4));
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float r = 0;
    for (int i = 0; i < 8; ++i) r += acc[i][0];
    if (r == 12345.678f) out[0] = r;
    if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = t1 - t0;
}
template <int K32>
static void run(const char* name) {
    float* d; long long* c; hipMalloc(&d, 4); hipMalloc(&c, 8);
    const int iters = 20000;
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((probe<K32>), dim3(512), dim3(256), 0, 0, d, c, iters);
    hipDeviceSynchronize();
    long long h = 0; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    // two waves per SIMD issue 2 x 8 MFMAs per loop trip
    printf("%-28s %6.2f clocks per MFMA (pipe-limited, 2 waves/SIMD)\n", name, (double)h / (iters * 16.0));
    hipFree(d); hipFree(c);
}
int main() {
    run<1>("v_mfma_f32_16x16x32_f16");
    run<0>("v_mfma_f32_16x16x16_f16");
    return 0;
}
