// Diagnostic: does v_mfma_f32_16x16x32_f16 keep fp16 subnormal inputs (needed by the two-part fp16 split of fp32 values)?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__global__ void probe(float* out, float aval, float bval) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)0.f; b[i] = (_Float16)0.f; }
    // A[row = lane & 15][k = 8 (lane >> 4) .. +7], B[k][col = lane & 15]: put one non-zero at k = 0 of every row / column
    if ((threadIdx.x >> 4) == 0) { a[0] = (_Float16)aval; b[0] = (_Float16)bval; }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = c[0];
}
int main() {
    float* d; (void)hipMalloc(&d, 4);
    const float cases[][2] = {{1.0f, 1.0f}, {3.0e-6f, 1.0f}, {1.0f, 3.0e-6f}, {6.0e-8f, 1024.0f}, {3.0e-6f, 3.0e-6f}, {5.96046448e-8f, 1.0f}};
    for (auto& cs : cases) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, cs[0], cs[1]);
        float h = 0; (void)hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        const double want = (double)(float)(_Float16)cs[0] * (double)(float)(_Float16)cs[1];
        printf("a=%.9g b=%.9g  mfma=%.9g  exact product of the fp16 values=%.9g\n", cs[0], cs[1], h, want);
    }
    return 0;
}
