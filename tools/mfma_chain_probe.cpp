// Diagnostic: what is a dependent v_mfma_f32_16x16x32_f16 worth when the whole chip is at its power cap?  512 workgroups x 256 threads (two
// waves per SIMD on every CU), 16 accumulators per wave, activation-like fp16 operands; 48 MFMAs per loop body in the order
// acc[(q / L) % 16]: L = 1 deals them out round-robin (every MFMA reads its C from the register file), L = 3 is the three-term chain of the
// conv kernels (two of three take C from the MFMA in front of them), L = 6 / 12 longer chains.  Reports core clocks and wall time per
// MFMA and the clock the chip held.  hipcc -O3 --offload-arch=gfx950 tools/mfma_chain_probe.cpp -o tools/mfma_chain_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define MF_V(ACC, A, B) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(ACC) : "v"(A), "v"(B))

template <int L>
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
        for (int q = 0; q < 48; ++q) MF_V(acc[(q / L) % 16], a[q & 7], b[(q + (q >> 3)) & 7]);
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

template <int L>
static void run(const u32x4* dops, float* dout, unsigned long long* dst, int nwg, int iters) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<L>, dim3(nwg), dim3(256), 0, 0, dops, dout, dst, iters);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(probe<L>, dim3(nwg), dim3(256), 0, 0, dops, dout, dst, iters);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ev_ms = 0.f;
    (void)hipEventElapsedTime(&ev_ms, e0, e1);
    std::vector<unsigned long long> st(nwg * 2);
    (void)hipMemcpy(st.data(), dst, nwg * 16, hipMemcpyDeviceToHost);
    double clk = 0, wall = 0;
    for (int i = 0; i < nwg; ++i) { clk += (double)st[2 * i]; wall += (double)st[2 * i + 1]; }
    clk /= nwg; wall /= nwg;   // wall in 10 ns ticks
    const double n = (double)iters * 48;
    // (the launch's own duration is the ground truth; the per-workgroup stamps agree as long as every workgroup is resident from start to end,
    //  which needs all 16 accumulators in use: a kernel with one accumulator is placed in shifts and its stamps read 12 clocks per MFMA)
    printf("chain length %2d: %6.2f clocks per MFMA per wave (%5.2f per SIMD), %6.3f ns per MFMA per SIMD in the kernel, %6.3f by hipEvent, clock %.2f GHz\n", L, clk / n,
           clk / n / 2, wall * 10.0 / n / 2, ev_ms * 1e6 / n / 2, clk / (wall * 10.0));
}

int main() {
    const int nwg = 512, iters = 15000;
    u32x4* dops; float* dout; unsigned long long* dst;
    (void)hipMalloc(&dops, 16 * 256 * 16); (void)hipMalloc(&dout, 4); (void)hipMalloc(&dst, nwg * 16);
    srand(1);
    std::vector<unsigned short> h(16 * 256 * 8);
    for (size_t i = 0; i < h.size(); ++i) {
        float g = 0.f;
        for (int k = 0; k < 12; ++k) g += (float)rand() / RAND_MAX;
        h[i] = f2h(g - 6.f);
    }
    (void)hipMemcpy(dops, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    for (int round = 0; round < 2; ++round) {
        run<1>(dops, dout, dst, nwg, iters);
        run<2>(dops, dout, dst, nwg, iters);
        run<3>(dops, dout, dst, nwg, iters);
        run<6>(dops, dout, dst, nwg, iters);
        run<12>(dops, dout, dst, nwg, iters);
    }
    return 0;
}
