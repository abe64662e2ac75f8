// Diagnostic: where do the blocks of a 512-block, 2-per-CU launch land?  Prints HW_ID fields per block.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256, 2) void probe(unsigned* out, unsigned long long* t) {
    extern __shared__ float lds[];
    if (threadIdx.x == 0) {
        out[blockIdx.x * 2 + 0] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));   // HW_ID, all 32 bits
        out[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));   // XCC_ID[3:0]
        t[blockIdx.x] = __builtin_amdgcn_s_memrealtime();
    }
    lds[threadIdx.x] = 0.f;
    for (int i = 0; i < 200; ++i) __builtin_amdgcn_s_sleep(127);   // stay resident ~0.7 ms so all 512 blocks co-reside
}
int main() {
    const int n = 512;
    unsigned* d; unsigned long long* dt;
    hipMalloc(&d, n * 8); hipMalloc(&dt, n * 8);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 77760);
    hipLaunchKernelGGL(probe, dim3(n), dim3(256), 77760, 0, d, dt);
    hipDeviceSynchronize();
    std::vector<unsigned> h(n * 2); std::vector<unsigned long long> ht(n);
    hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost); hipMemcpy(ht.data(), dt, n * 8, hipMemcpyDeviceToHost);
    unsigned long long t0 = ht[0]; for (auto v : ht) if (v < t0) t0 = v;
    for (int b = 0; b < n; ++b) {
        unsigned v = h[b * 2];
        printf("blk %3d xcc %u se %u sh %u cu %2u simd %u wave %2u tg %2u t+%llu\n", b, h[b * 2 + 1] & 15, (v >> 13) & 7, (v >> 12) & 1,
               (v >> 8) & 15, (v >> 4) & 3, v & 15, (v >> 16) & 15, ht[b] - t0);
    }
    return 0;
}
