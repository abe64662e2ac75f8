// Diagnostic: two-part fp16 split by convert / subtract / convert against the packed-convert + mixed-precision-FMA form.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* x, unsigned* ref, unsigned* got, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    const float a = x[2 * i], b = x[2 * i + 1];
    const f16x2 h = {(_Float16)a, (_Float16)b};
    const f16x2 l = {(_Float16)(a - (float)h[0]), (_Float16)(b - (float)h[1])};
    ref[2 * i] = __builtin_bit_cast(unsigned, h);
    ref[2 * i + 1] = __builtin_bit_cast(unsigned, l);
    const unsigned hh = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){a, b}, f16x2));
    unsigned ll;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(ll) : "v"(hh), "v"(a));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(ll) : "v"(hh), "v"(b));
    got[2 * i] = hh;
    got[2 * i + 1] = ll;
}
int main() {
    const int n = 1 << 20;
    float* hx = (float*)malloc(n * 4);
    srand(1);
    for (int i = 0; i < n; ++i) {
        const float u = (float)rand() / RAND_MAX * 2.f - 1.f;
        const int e = rand() % 40 - 30;      // magnitudes 2^-30 .. 2^9
        hx[i] = i % 97 == 0 ? 0.f : ldexpf(u, e);
    }
    float* dx; unsigned *dr, *dg;
    hipMalloc(&dx, n * 4); hipMalloc(&dr, n * 4); hipMalloc(&dg, n * 4);
    hipMemcpy(dx, hx, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 2 / 256), dim3(256), 0, 0, dx, dr, dg, n);
    unsigned* hr = (unsigned*)malloc(n * 4); unsigned* hg = (unsigned*)malloc(n * 4);
    hipMemcpy(hr, dr, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hg, dg, n * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; ++i)
        if (hr[i] != hg[i]) {
            if (bad < 10) printf("mismatch at %d (%s): x = %g %g  ref %08x got %08x\n", i, i & 1 ? "low parts" : "high parts", hx[i & ~1], hx[i | 1], hr[i], hg[i]);
            ++bad;
        }
    printf("%d mismatching words of %d\n", bad, n);
    return 0;
}
