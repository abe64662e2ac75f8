// What does a captured  hipMemsetD32Async(word) -> kernel(atomicAdd(word))  pair do when its graph is replayed back to back WITHOUT a
// synchronisation between the replays?  (Round 3 saw a GPU memory fault on such a replay loop of kws_forward_wav; its persistent kernels
// then had a memset node in front of each of them.)  Harmless by construction: the kernel only draws a ticket and logs it, range-checked.
//   hipcc -O2 --offload-arch=gfx950 tools/graph_memset_probe.cpp -o tools/graph_memset_probe && tools/graph_memset_probe [dot-file]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            std::printf("FAILED %s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);      \
            return 1;                                                                           \
        }                                                                                       \
    } while (0)

__global__ void draw(unsigned* word, unsigned* logpos, unsigned* log, unsigned cap, unsigned spin_ticks) {
    if (threadIdx.x != 0) return;
    const unsigned t = atomicAdd(word, 1u);
    const unsigned i = atomicAdd(logpos, 1u);
    if (i < cap) log[i] = t;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz
    while (__builtin_amdgcn_s_memrealtime() - t0 < spin_ticks) __builtin_amdgcn_s_sleep(8);
}

int main(int argc, char** argv) {
    const int G = 64, R = 20, START = 512;
    const unsigned SENT = 0xA5A5A5A5u, spin = 5000;   // 50 us per workgroup
    unsigned *blk, *logpos, *log;
    CK(hipMalloc(&blk, 64 * 4));
    CK(hipMalloc(&logpos, 4));
    CK(hipMalloc(&log, (size_t)R * G * 4));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipGraph_t graph;
    hipGraphExec_t exec;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    CK(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(blk + 16), START, 1, s));
    hipLaunchKernelGGL(draw, dim3(G), dim3(64), 0, s, blk + 16, logpos, log, (unsigned)(R * G), spin);
    CK(hipStreamEndCapture(s, &graph));
    size_t nn = 0;
    CK(hipGraphGetNodes(graph, nullptr, &nn));
    std::vector<hipGraphNode_t> nodes(nn);
    CK(hipGraphGetNodes(graph, nodes.data(), &nn));
    std::printf("captured graph: %zu nodes\n", nn);
    for (size_t i = 0; i < nn; ++i) {
        hipGraphNodeType ty;
        CK(hipGraphNodeGetType(nodes[i], &ty));
        size_t nd = 0;
        CK(hipGraphNodeGetDependencies(nodes[i], nullptr, &nd));
        std::printf("  node %zu: type %d (%s), %zu dependencies\n", i, (int)ty,
                    ty == hipGraphNodeTypeKernel ? "kernel" : (ty == hipGraphNodeTypeMemset ? "memset" : "other"), nd);
        if (ty == hipGraphNodeTypeMemset) {
            hipMemsetParams mp{};
            CK(hipGraphMemsetNodeGetParams(nodes[i], &mp));
            std::printf("    memset: dst %p (word at %p) value %u elementSize %u width %zu height %zu pitch %zu\n", mp.dst, (void*)(blk + 16),
                        mp.value, mp.elementSize, mp.width, mp.height, mp.pitch);
        }
    }
    if (argc > 1) CK(hipGraphDebugDotPrint(graph, argv[1], hipGraphDebugDotFlagsVerbose));
    CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));

    for (int mode = 0; mode < 2; ++mode) {   // 0: a synchronisation after every replay; 1: back to back
        std::vector<unsigned> init(64, SENT);
        CK(hipMemcpy(blk, init.data(), 64 * 4, hipMemcpyHostToDevice));
        CK(hipMemset(logpos, 0, 4));
        CK(hipMemset(log, 0xff, (size_t)R * G * 4));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, s));
        for (int r = 0; r < R; ++r) {
            CK(hipGraphLaunch(exec, s));
            if (mode == 0) CK(hipStreamSynchronize(s));
        }
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned> hl((size_t)R * G), hb(64);
        CK(hipMemcpy(hl.data(), log, hl.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hb.data(), blk, 64 * 4, hipMemcpyDeviceToHost));
        int bad_groups = 0, sent_bad = 0;
        for (int r = 0; r < R; ++r) {
            std::vector<unsigned> g(hl.begin() + (size_t)r * G, hl.begin() + (size_t)(r + 1) * G);
            std::sort(g.begin(), g.end());
            bool ok = true;
            for (int i = 0; i < G; ++i) ok = ok && g[i] == (unsigned)(START + i);
            if (!ok) {
                if (bad_groups < 4) std::printf("    replay %d: tickets %u .. %u (want %d .. %d)\n", r, g.front(), g.back(), START, START + G - 1);
                ++bad_groups;
            }
        }
        for (int i = 0; i < 64; ++i)
            if (i != 16 && hb[i] != SENT) ++sent_bad;
        std::printf("%s: %d replays in %.3f ms (%.1f us each; one replay's kernel spins 50 us); replays with wrong tickets: %d; word after: %u; "
                    "sentinel words overwritten: %d\n",
                    mode == 0 ? "synchronised" : "back-to-back", R, ms, 1e3 * ms / R, bad_groups, hb[16], sent_bad);
    }
    return 0;
}
