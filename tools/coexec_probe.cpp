// Diagnostic: can VALU fp32 FMAs overlap fp32-input MFMAs (v_mfma_f32_16x16x4_f32) issued by ANOTHER wave on the
// same SIMD?  512-thread blocks, one per CU: waves 0-3 run an MFMA loop, waves 4-7 a v_fma / v_pk_fma loop.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(512, 2) void probe(float* out, int n_mfma, int n_valu, int valu_kind) {
    const int w = threadIdx.x >> 6;
    float r = 0.f;
    if (w < 4) {
        f32x4 acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0, 0, 0, 0};
        float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
        for (int it = 0; it < n_mfma; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][3];
    } else if (valu_kind == 0) {
        float v[16];
        for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 0.001f + i;
        const float m = 1.0000001f, c = 1e-7f;
        for (int it = 0; it < n_valu; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = __builtin_fmaf(v[i], m, c);
        }
        for (int i = 0; i < 16; ++i) r += v[i];
    } else {
        f32x2 v[16];
        for (int i = 0; i < 16; ++i) v[i] = (f32x2){threadIdx.x * 0.001f + i, 1.f * i};
        const f32x2 m = {1.0000001f, 0.9999999f}, c = {1e-7f, 2e-7f};
        for (int it = 0; it < n_valu; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = __builtin_elementwise_fma(v[i], m, c);
        }
        for (int i = 0; i < 16; ++i) r += v[i].x + v[i].y;
    }
    if (r == 12345.678f) out[0] = r;
}
static float run(int nm, int nv, int kind) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float* d; hipMalloc(&d, 4);
    hipLaunchKernelGGL(probe, dim3(256), dim3(512), 0, 0, d, nm, nv, kind);   // warm-up
    hipEventRecord(a);
    hipLaunchKernelGGL(probe, dim3(256), dim3(512), 0, 0, d, nm, nv, kind);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); hipFree(d); return ms;
}
int main() {
    const int NM = 20000, NV = 40000;   // 320k MFMAs (x32 cyc) per wave vs 640k VALU (x4 cyc issue) per wave
    for (int kind = 0; kind < 2; ++kind) {
        float tm = run(NM, 0, kind), tv = run(0, NV, kind), tb = run(NM, NV, kind);
        printf("valu_kind %d (%s): mfma only %.3f ms, valu only %.3f ms, both %.3f ms  (sum %.3f, max %.3f)\n", kind,
               kind ? "v_pk_fma_f32" : "v_fma_f32", tm, tv, tb, tm + tv, tm > tv ? tm : tv);
    }
    return 0;
}
