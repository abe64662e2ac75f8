// Diagnostic: bf16 MFMA (v_mfma_f32_16x16x32_bf16) rate and co-execution with fp32 VALU from another wave per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(512, 2) void probe(float* out, int n_mfma, int n_valu) {
    const int w = threadIdx.x >> 6;
    float r = 0.f;
    if (w < 4) {
        f32x4 acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0, 0, 0, 0};
        bf16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3f80 + threadIdx.x + i); b[i] = (short)(0x3f00 + 3 * threadIdx.x + i); }
        for (int it = 0; it < n_mfma; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][3];
    } else {
        float v[16];
        for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 0.001f + i;
        const float m = 1.0000001f, c = 1e-7f;
        for (int it = 0; it < n_valu; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = __builtin_fmaf(v[i], m, c);
        }
        for (int i = 0; i < 16; ++i) r += v[i];
    }
    if (r == 12345.678f) out[0] = r;
}
static float run(int nm, int nv) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float* d; hipMalloc(&d, 4);
    hipLaunchKernelGGL(probe, dim3(256), dim3(512), 0, 0, d, nm, nv);
    hipEventRecord(a);
    hipLaunchKernelGGL(probe, dim3(256), dim3(512), 0, 0, d, nm, nv);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); hipFree(d); return ms;
}
int main() {
    const int NM = 40000, NV = 40000;   // 640k bf16 MFMAs per wave, 640k v_fma per wave
    float tm = run(NM, 0), tv = run(0, NV), tb = run(NM, NV);
    double flops = 256.0 * 4 * NM * 16.0 * 2 * 16 * 16 * 32;
    printf("bf16 16x16x32: mfma only %.3f ms (%.0f TFLOP/s, %.1f cyc/MFMA at 2.4GHz), valu only %.3f ms, both %.3f ms (sum %.3f, max %.3f)\n",
           tm, flops / tm * 1e-9, tm * 1e-3 * 2.4e9 / (NM * 16.0), tv, tb, tm + tv, tm > tv ? tm : tv);
    return 0;
}
