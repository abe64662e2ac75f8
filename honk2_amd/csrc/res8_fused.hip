// res8 forward, fully fused for gfx950: (101,40) feature map -> 12 logits, activations never leave the CU.
//
// Replaces ResNet.forward for config/resnet/res8.json (reference model/resnet.py:38-60):
//   conv_0 3x3 (1->45) -> ReLU -> AvgPool(4,3) -> 6 x [conv 3x3 (45->45) -> ReLU -> (+prev on even i) -> BN]
//   -> spatial mean -> Linear(45, n_labels).
//
// MI355X mapping (one 256-thread workgroup = 4 waves per clip, 2 workgroups per CU, persistent over clips)
//   * fp32-input MFMA and fp32 VALU share a datapath on gfx950 (tools/coexec_probe.cpp: times add), so every VALU
//     instruction costs matrix-pipe time.  All convolutions therefore run on v_mfma_f32_16x16x4_f32 (1024 MACs per
//     issue slot) and the epilogues are two or three VALU ops per output.
//   * The 45x25x13 activation map lives in LDS as [48 planes][27 rows x 14 cols (+pad) = 400 words], zero halo
//     included (the right halo column of row y is the left halo column of row y+1); planes 45..47 are spare so a
//     16-row tile store needs no bounds test.  BatchNorm is applied when a layer's output is WRITTEN (one FMA from
//     an LDS (scale, shift) table), so the halo stays exactly zero as the reference's padding is.
//   * conv_0 + ReLU + AvgPool(4,3): the feature map is staged (zero top/left halo) in the part of the map region
//     conv_0's own output does not need yet; for each of the 12 positions of the pooling window an implicit GEMM
//     (M = 45->48, K = 9->12, N = 16 pooled positions per tile) is accumulated from a zero C, ReLU'd and summed, so
//     the pooled map appears directly in accumulator layout -- which is exactly prev_x for the first residual.
//   * conv_i (i = 1..6): implicit GEMM M = 45 (3 tiles), N = 325 positions (21 tiles), K = 405 walked as 9 taps x
//     11 groups of 4 input channels (one ds_read_b32 per B fragment, tap shift and channel group are immediates) +
//     3 steps that gather channel 44 of all 9 taps: 102 k-steps, 0.7 % padding.  A fragments (weights pre-packed on
//     the host in fragment order) stream from L2, one 16-byte load per lane per 4 k-steps, one group ahead.
//   * Each wave owns 5 position tiles x 3 channel tiles + one tile of position-tile 20: 16 accumulators.  A layer's
//     whole output stays in registers until every wave has finished reading the input map, then overwrites it in
//     place; the residual also stays in registers across layers.  LDS holds ONE map (76.8 KB): two workgroups per CU.
//   * the spatial mean is a 16-lane xor-shuffle reduction + a 4-wave LDS combine; Linear(45, n) by n threads.
#include "kws_internal.h"

namespace kws {

namespace {
constexpr int R8_PLANES = 48;
constexpr int ACT_WORDS = R8_PLANES * R8_CS;   // 19200
constexpr int RED_WORDS = 4 * 48;
constexpr int BNT_WORDS = R8_LAYERS * 96;      // per layer: scale[48], shift[48]
constexpr int LDS_WORDS = ACT_WORDS + RED_WORDS + 48 + BNT_WORDS;   // 20016 words = 80 064 B
constexpr int FS = 41;                         // staged feature row stride: 40 bins + left zero column
constexpr int FEAT_WORDS = 102 * FS;           // rows -1..100

// one k-step: 6 B fragments (5 own position tiles + tile 20) against 3 + 1 A fragments -> 16 MFMAs
#define R8_STEP(A0, A1, A2, AX, BADDR)                                                              \
    {                                                                                               \
        float b_[6];                                                                                \
        _Pragma("unroll") for (int j = 0; j < 6; ++j) b_[j] = act[BADDR];                            \
        _Pragma("unroll") for (int j = 0; j < 5; ++j) {                                              \
            acc[j][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(A0, b_[j], acc[j][0], 0, 0, 0);         \
            acc[j][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(A1, b_[j], acc[j][1], 0, 0, 0);         \
            acc[j][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(A2, b_[j], acc[j][2], 0, 0, 0);         \
        }                                                                                           \
        accx = __builtin_amdgcn_mfma_f32_16x16x4f32(AX, b_[5], accx, 0, 0, 0);                       \
    }

// first k-step of a layer: C is the literal 0, so the 64 accumulator registers need no clearing
#define R8_STEP0(A0, A1, A2, AX, BADDR)                                                             \
    {                                                                                               \
        const f32x4 z_ = (f32x4){0.f, 0.f, 0.f, 0.f};                                               \
        float b_[6];                                                                                \
        _Pragma("unroll") for (int j = 0; j < 6; ++j) b_[j] = act[BADDR];                            \
        _Pragma("unroll") for (int j = 0; j < 5; ++j) {                                              \
            acc[j][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(A0, b_[j], z_, 0, 0, 0);                \
            acc[j][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(A1, b_[j], z_, 0, 0, 0);                \
            acc[j][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(A2, b_[j], z_, 0, 0, 0);                \
        }                                                                                           \
        accx = __builtin_amdgcn_mfma_f32_16x16x4f32(AX, b_[5], z_, 0, 0, 0);                         \
    }

// ReLU as ONE VALU op: a signed-integer max of the bit pattern with 0 (negative floats and -0 have the sign bit set
// and map to +0, positive floats are unchanged).  fmaxf / elementwise_max / med3 all make hipcc emit a canonicalising
// v_max x,x in front, and every VALU slot costs matrix-pipe time on this chip.
__device__ __forceinline__ float relu1(float x) {
    const int b = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, b > 0 ? b : 0);
}
__device__ __forceinline__ f32x4 relu4(f32x4 v) { return (f32x4){relu1(v[0]), relu1(v[1]), relu1(v[2]), relu1(v[3])}; }

struct Ctx {
    float* act;
    float* red;
    float* mvec;
    const float* bnt;
    int tid, lane, w, g, pcol, mx;
    int qn[6];
    bool xvalid;
};

// conv_{layer+1} + ReLU (+ residual when EVEN) + BN; LAST: + spatial mean + Linear instead of the write-back
template <bool EVEN, bool LAST>
__device__ __forceinline__ void r8_layer(const Res8Params& p, const Ctx& c, int layer, int clip, f32x4 (&prev)[5][3],
                                         f32x4& prevx) {
    float* act = c.act;
    const int g = c.g, mx = c.mx;
    f32x4 acc[5][3], accx;
    const f32x4* A = p.apk + (size_t)layer * R8_GROUPS * 3 * 64 + c.lane;
    {   // input channel 44 first: k-slot g of step q is tap (ky = q, kx = g); slot g = 3 carries a zero weight
        const f32x4* Al = A + (size_t)(27 * 3) * 64;
        const f32x4 l0 = Al[0], l1 = Al[64], l2 = Al[128], lx = Al[mx * 64];
        R8_STEP0(l0[0], l1[0], l2[0], lx[0], c.qn[j] + 44 * R8_CS + (0 - 1) * R8_RS + g - 1)
        R8_STEP(l0[1], l1[1], l2[1], lx[1], c.qn[j] + 44 * R8_CS + (1 - 1) * R8_RS + g - 1)
        R8_STEP(l0[2], l1[2], l2[2], lx[2], c.qn[j] + 44 * R8_CS + (2 - 1) * R8_RS + g - 1)
    }
    f32x4 n0 = A[0], n1 = A[64], n2 = A[128], nx = A[mx * 64];
    for (int tap = ((KWS_DBG(p.debug & 2)) ? 9 : 0); tap < 9; ++tap) {
        const int ty = tap / 3;
        const int tapoff = (ty - 1) * R8_RS + (tap - 3 * ty - 1) + g * R8_CS;
        int bt[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) bt[j] = c.qn[j] + tapoff;
#pragma unroll
        for (int grp = 0; grp < 3; ++grp) {
            const f32x4 a0 = n0, a1 = n1, a2 = n2, ax = nx;
            const f32x4* An = A + (size_t)((tap * 3 + grp + 1) * 3) * 64;   // one group ahead (index 27 on the last: valid, unused)
            n0 = An[0]; n1 = An[64]; n2 = An[128]; nx = An[mx * 64];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int step = 4 * grp + q;
                if (step < 11) R8_STEP(a0[q], a1[q], a2[q], ax[q], bt[j] + step * (4 * R8_CS))
            }
        }
    }

    // ---- epilogue: ReLU, residual (reference: even i), BatchNorm as one FMA
    const float* bt = c.bnt + layer * 96 + 4 * g;
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        const f32x4 sc = *reinterpret_cast<const f32x4*>(bt + 16 * m);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(bt + 48 + 16 * m);
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = relu1(acc[j][m][r]);
                if (EVEN) {
                    v += prev[j][m][r];
                    prev[j][m][r] = v;
                }
                acc[j][m][r] = fmaf(v, sc[r], sh[r]);
            }
    }
    {
        const f32x4 sc = *reinterpret_cast<const f32x4*>(bt + 16 * mx);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(bt + 48 + 16 * mx);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = relu1(accx[r]);
            if (EVEN) {
                v += prevx[r];
                prevx[r] = v;
            }
            accx[r] = fmaf(v, sc[r], sh[r]);
        }
    }

    __syncthreads();  // every wave has finished reading this layer's input map
    if (!LAST) {
        // tiles 0..19 hold real positions only; rows 45..47 of channel tile 2 are zeros into spare planes
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
            for (int m = 0; m < 3; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) act[(16 * m + 4 * g + r) * R8_CS + c.qn[j]] = acc[j][m][r];
        if (c.xvalid) {
#pragma unroll
            for (int r = 0; r < 4; ++r) act[(16 * mx + 4 * g + r) * R8_CS + c.qn[5]] = accx[r];
        }
        __syncthreads();
    } else {
        // ---- spatial mean + Linear(45, n_labels)
#pragma unroll
        for (int m = 0; m < 3; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[0][m][r] + acc[1][m][r] + acc[2][m][r] + acc[3][m][r] + acc[4][m][r];
                v += __shfl_xor(v, 8);
                v += __shfl_xor(v, 4);
                v += __shfl_xor(v, 2);
                v += __shfl_xor(v, 1);
                if (c.pcol == 0) c.red[c.w * 48 + 16 * m + 4 * g + r] = v;
            }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = c.xvalid ? accx[r] : 0.f;
            v += __shfl_xor(v, 8);
            v += __shfl_xor(v, 4);
            v += __shfl_xor(v, 2);
            v += __shfl_xor(v, 1);
            if (c.pcol == 0 && c.w < 3) c.red[c.w * 48 + 16 * mx + 4 * g + r] += v;
        }
        __syncthreads();
        if (c.tid < 48)
            c.mvec[c.tid] = (c.red[c.tid] + c.red[48 + c.tid] + c.red[96 + c.tid] + c.red[144 + c.tid]) / (float)R8_NPOS;
        __syncthreads();
        if (c.tid < p.n_labels) {
            const float* wr = p.out_w + c.tid * R8_C;
            float o = 0.f;
            for (int ch = 0; ch < R8_C; ++ch) o = fmaf(wr[ch], c.mvec[ch], o);
            p.logits[(size_t)clip * p.n_labels + c.tid] = o + p.out_b[c.tid];
        }
    }
}
}  // namespace

size_t res8_lds_bytes() { return ((size_t)LDS_WORDS * sizeof(float) + 15) & ~(size_t)15; }

__global__ __launch_bounds__(256, 2) void res8_kernel(Res8Params p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    Ctx c;
    c.act = lds;
    c.red = lds + ACT_WORDS;
    c.mvec = c.red + RED_WORDS;
    float* bnt = c.mvec + 48;
    c.bnt = bnt;
    float* act = c.act;

    const int tid = threadIdx.x;
    c.tid = tid;
    c.lane = tid & 63;
    c.w = __builtin_amdgcn_readfirstlane(tid >> 6);
    c.g = c.lane >> 4;
    c.pcol = c.lane & 15;
    c.mx = c.w < 2 ? c.w : 2;  // channel tile of this wave's extra (position-tile 20) accumulator
    const int w = c.w, g = c.g, pcol = c.pcol, mx = c.mx, lane = c.lane;

    for (int i = tid; i < ACT_WORDS; i += 256) act[i] = 0.f;
    for (int i = tid; i < BNT_WORDS; i += 256) bnt[i] = p.bn_tab[i];

    int lb[6];          // staged-feature cell of the top-left input of this lane's pooling window, per tile
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int nt = j < 5 ? 5 * w + j : 20;
        const int n = 16 * nt + pcol;
        const int nn = n < R8_NPOS ? n : R8_NPOS - 1;
        const int y = nn / W8_W;
        const int x = nn - y * W8_W;
        c.qn[j] = (y + 1) * R8_RS + x + 1;
        lb[j] = 4 * y * FS + 3 * x;     // cell of feature (4y-1, 3x-1) in the staged map (row -1 / col -1 are zeros)
    }
    c.xvalid = w < 3 && (16 * 20 + pcol) < R8_NPOS;   // this lane's slot of the extra tile is a real output
    int koff[3];        // conv_0 k-slot (tap) offset inside the staged map for k = 4 s + g; k >= 9 is padding
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const int k = 4 * s + g;
        koff[s] = k < 9 ? (k / 3) * FS + (k - 3 * (k / 3)) : 0;
    }

    for (int clip = blockIdx.x; clip < p.B; clip += gridDim.x) {
        __syncthreads();  // previous clip's tail has consumed red/mvec/act; first clip: zero-fill complete

        // ------------------------------------------------------------ stage the (101, 40) feature map, zero halo
        {
            const f32x4* f4 = reinterpret_cast<const f32x4*>(p.feat + (size_t)clip * p.T * p.F);
            f32x4 v[4];
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int q4 = it * 256 + tid;
                if (q4 < 1010) v[it] = f4[q4];
            }
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int q4 = it * 256 + tid;
                if (q4 < 1010) {
                    const int idx = 4 * q4;
                    const int cell = idx + idx / 40 + FS + 1;      // (r+1)*41 + c + 1 with r = idx/40, c = idx%40
                    act[cell] = v[it][0];
                    act[cell + 1] = v[it][1];
                    act[cell + 2] = v[it][2];
                    act[cell + 3] = v[it][3];
                }
            }
            if (tid < FS) act[tid] = 0.f;                          // row -1
            if (tid < 101) act[(tid + 1) * FS] = 0.f;              // column -1
        }
        __syncthreads();

        // ------------------------------------------------------------ conv_0 + ReLU + AvgPool(4,3) on the matrix cores
        f32x4 prev[5][3], prevx;
        {
            float a0[3][3];
#pragma unroll
            for (int m = 0; m < 3; ++m)
#pragma unroll
                for (int s = 0; s < 3; ++s) a0[m][s] = p.w0a[(m * 3 + s) * 64 + lane];
            const float ax0 = p.w0a[(mx * 3 + 0) * 64 + lane], ax1 = p.w0a[(mx * 3 + 1) * 64 + lane],
                        ax2 = p.w0a[(mx * 3 + 2) * 64 + lane];
            const f32x4 zero = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int ad0 = lb[j] + koff[0], ad1 = lb[j] + koff[1], ad2 = lb[j] + koff[2];
                f32x4 s0 = zero, s1 = zero, s2 = zero;
                if (!(KWS_DBG(p.debug & 1))) {
#pragma unroll
                    for (int wp = 0; wp < 6; ++wp) {     // pooling-window positions two at a time: the second
                        f32x4 cc[2][3];                  // position's MFMAs cover the first one's result latency
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const int wi = 2 * wp + h, oy = wi / 3, ox = wi - 3 * oy;
                            const float b0 = act[ad0 + oy * FS + ox], b1 = act[ad1 + oy * FS + ox],
                                        b2 = act[ad2 + oy * FS + ox];
#pragma unroll
                            for (int m = 0; m < 3; ++m) cc[h][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[m][0], b0, zero, 0, 0, 0);
#pragma unroll
                            for (int m = 0; m < 3; ++m) cc[h][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[m][1], b1, cc[h][m], 0, 0, 0);
#pragma unroll
                            for (int m = 0; m < 3; ++m) cc[h][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[m][2], b2, cc[h][m], 0, 0, 0);
                        }
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            s0 += relu4(cc[h][0]);
                            s1 += relu4(cc[h][1]);
                            s2 += relu4(cc[h][2]);
                        }
                    }
                }
                prev[j][0] = s0 * (1.0f / 12.0f);
                prev[j][1] = s1 * (1.0f / 12.0f);
                prev[j][2] = s2 * (1.0f / 12.0f);
            }
            {
                const int ad0 = lb[5] + koff[0], ad1 = lb[5] + koff[1], ad2 = lb[5] + koff[2];
                f32x4 sx = zero;
                if (!(KWS_DBG(p.debug & 1))) {
#pragma unroll
                    for (int oy = 0; oy < 4; ++oy)
#pragma unroll
                        for (int ox = 0; ox < 3; ++ox) {
                            f32x4 cx = __builtin_amdgcn_mfma_f32_16x16x4f32(ax0, act[ad0 + oy * FS + ox], zero, 0, 0, 0);
                            cx = __builtin_amdgcn_mfma_f32_16x16x4f32(ax1, act[ad1 + oy * FS + ox], cx, 0, 0, 0);
                            cx = __builtin_amdgcn_mfma_f32_16x16x4f32(ax2, act[ad2 + oy * FS + ox], cx, 0, 0, 0);
                            sx += relu4(cx);
                        }
                }
                prevx = sx * (1.0f / 12.0f);
            }
        }
        __syncthreads();  // every wave is done with the staged features

        // ------------------------------------------------------------ pooled map -> LDS; restore the zero cells the
        //                                                               staging dirtied (halo + padding of planes 0..10)
#pragma unroll
        for (int i = 0; i < 4; ++i) act[p.zcells[i * 256 + tid]] = 0.f;
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
            for (int m = 0; m < 3; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) act[(16 * m + 4 * g + r) * R8_CS + c.qn[j]] = prev[j][m][r];
        if (c.xvalid) {
#pragma unroll
            for (int r = 0; r < 4; ++r) act[(16 * mx + 4 * g + r) * R8_CS + c.qn[5]] = prevx[r];
        }
        __syncthreads();

        // ------------------------------------------------------------ conv_1 .. conv_6
        r8_layer<false, false>(p, c, 0, clip, prev, prevx);
        r8_layer<true, false>(p, c, 1, clip, prev, prevx);
        r8_layer<false, false>(p, c, 2, clip, prev, prevx);
        r8_layer<true, false>(p, c, 3, clip, prev, prevx);
        r8_layer<false, false>(p, c, 4, clip, prev, prevx);
        r8_layer<true, true>(p, c, 5, clip, prev, prevx);
    }
}

hipError_t launch_res8(const Res8Params& p, int grid, hipStream_t s) {
    static DeviceOnce attr_once;
    if (attr_once.first()) {
        hipError_t e = hipFuncSetAttribute((const void*)res8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
        if (e != hipSuccess) return e;
    }
    if (p.B <= 0) return hipSuccess;
    // p.debug bit 2 (timing experiments only): pad LDS so that only one workgroup fits a CU
    const size_t lds = (KWS_DBG(p.debug & 4)) ? (size_t)100 * 1024 : res8_lds_bytes();
    hipLaunchKernelGGL(res8_kernel, dim3((unsigned)grid), dim3(256), lds, s, p);
    return hipGetLastError();
}

// Host: pack one conv_i weight (45,45,3,3) into fragment order.  float4 index (grp*3 + m)*64 + lane, component q:
//   row = lane & 15 -> cout = 16 m + row; k-slot g = lane >> 4
//   grp < 27: tap = grp/3, step = 4 (grp%3) + q (step 11 = padding), cin = 4 step + g
//   grp = 27: cin = 44, tap (ky = q, kx = g); q = 3 and g = 3 are padding
void pack_res8_layer(const float* wt, float* dst) {
    for (int grp = 0; grp < R8_GROUPS; ++grp)
        for (int m = 0; m < 3; ++m)
            for (int lane = 0; lane < 64; ++lane)
                for (int q = 0; q < 4; ++q) {
                    const int co = 16 * m + (lane & 15);
                    const int g = lane >> 4;
                    float v = 0.f;
                    if (co < R8_C) {
                        if (grp < 27) {
                            const int tap = grp / 3, step = 4 * (grp % 3) + q;
                            if (step < 11) v = wt[((size_t)co * R8_C + (4 * step + g)) * 9 + tap];
                        } else if (q < 3 && g < 3) {
                            v = wt[((size_t)co * R8_C + 44) * 9 + q * 3 + g];
                        }
                    }
                    dst[(((size_t)grp * 3 + m) * 64 + lane) * 4 + q] = v;
                }
}

// Host: conv_0 weight (45,1,3,3) -> A fragments [m][s][lane]: cout = 16 m + (lane & 15), tap k = 4 s + (lane >> 4)
void pack_res8_conv0(const float* wt, float* dst) {
    for (int m = 0; m < 3; ++m)
        for (int s = 0; s < 3; ++s)
            for (int lane = 0; lane < 64; ++lane) {
                const int co = 16 * m + (lane & 15), k = 4 * s + (lane >> 4);
                dst[(m * 3 + s) * 64 + lane] = (co < R8_C && k < 9) ? wt[co * 9 + k] : 0.f;
            }
}

// Host: the 1024-entry list of map cells (planes 0..10) that are NOT interior positions, i.e. the cells that must be
// zero again after the feature staging used that region (825 cells, padded with repeats of cell 0).
void build_res8_zero_cells(int* dst) {
    int n = 0;
    for (int pl = 0; pl <= (FEAT_WORDS - 1) / R8_CS; ++pl)
        for (int q = 0; q < R8_CS; ++q) {
            const bool interior = q / R8_RS >= 1 && q / R8_RS <= R8_H && q % R8_RS >= 1 && q < (R8_H + 1) * R8_RS;
            if (!interior) dst[n++] = pl * R8_CS + q;
        }
    while (n < 1024) dst[n++] = 0;
}

}  // namespace kws
