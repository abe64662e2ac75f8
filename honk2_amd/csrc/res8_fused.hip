// res8 forward, fully fused for gfx950: (101,40) feature map -> 12 logits, activations never leave the CU.
//
// Replaces ResNet.forward for config/resnet/res8.json (reference model/resnet.py:38-60):
//   conv_0 3x3 (1->45) -> ReLU -> AvgPool(4,3) -> 6 x [conv 3x3 (45->45) -> ReLU -> (+prev on even i) -> BN]
//   -> spatial mean -> Linear(45, n_labels).
//
// MI355X mapping (one 256-thread workgroup = 4 waves per clip, 2 workgroups per CU, persistent over clips)
//   * The 45x25x13 activation map lives in LDS as [channel][27 rows x 14 cols], zero halo included (the
//     right halo column of row y is the left halo column of row y+1), channel stride 400 words.  BatchNorm is
//     applied when a layer's output is WRITTEN, so the halo stays exactly zero as the reference's padding is.
//   * conv_i (i = 1..6) is an implicit GEMM on the fp32 matrix cores (v_mfma_f32_16x16x4_f32):
//       M = 45 output channels (3 tiles of 16), N = 325 positions (21 tiles of 16), K = 405 = 9 taps x 45.
//     K is walked as 9 taps x 11 groups of 4 input channels (one ds_read_b32 per B fragment, the tap shift and
//     the channel group are an immediate offset) + 3 steps that gather channel 44 of all 9 taps: 102 k-steps,
//     0.7 % padding.  A fragments (weights, pre-packed on the host in fragment order) stream from L2 as one
//     16-byte load per lane per 4 k-steps and never touch LDS.
//   * Each wave owns 5 position tiles x 3 channel tiles + one tile of position-tile 20: 16 accumulators.  A
//     layer's whole output stays in registers until every wave has finished reading the input map, then
//     overwrites it in place; the residual ("prev_x") also stays in registers across layers.  LDS holds ONE
//     activation map (72 KB), which is what lets two workgroups share a CU and overlap one's VALU phases
//     (conv_0/pool, epilogues) with the other's MFMA phase.
//   * conv_0 + ReLU + AvgPool runs on the VALU with lane = pooled position (30 inputs held in registers,
//     reused for all channels), two output channels per packed FMA, weights from scalar registers.
//   * the spatial mean is a 16-lane xor-shuffle reduction + a 4-wave LDS combine.
#include "kws_internal.h"

namespace kws {

namespace {
constexpr int R8_PLANES = 48;                 // 45 channels + 3 all-zero planes so every store of a 16-row tile is in bounds
constexpr int ACT_WORDS = R8_PLANES * R8_CS;  // 19200
constexpr int RED_WORDS = 4 * 48;
constexpr int LDS_WORDS = ACT_WORDS + RED_WORDS + 48;

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
}  // namespace

size_t res8_lds_bytes() { return ((size_t)LDS_WORDS * sizeof(float) + 15) & ~(size_t)15; }

__global__ __launch_bounds__(256, 2) void res8_kernel(Res8Params p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* act = lds;
    float* red = lds + ACT_WORDS;
    float* mvec = red + RED_WORDS;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4;
    const int pcol = lane & 15;
    const int mx = w < 2 ? w : 2;  // channel tile of this wave's extra (position-tile 20) accumulator

    for (int i = tid; i < ACT_WORDS; i += 256) act[i] = 0.f;

    // Two workgroups share a CU and would otherwise run in lockstep (same work per clip), so one's VALU phases
    // (conv_0/pool, epilogues) never hide under the other's MFMA phase.  Delay the workgroup in the odd
    // threadgroup slot of its CU by roughly a third of a clip, once.  HW_REG_HW_ID[19:16] = TG_ID.  Speed only:
    // any value of the register gives correct results.
    if ((__builtin_amdgcn_s_getreg(4 | (16 << 6) | (3 << 11)) & 1) != 0)
        for (int i = 0; i < p.stagger_sleeps; ++i) __builtin_amdgcn_s_sleep(127);

    int qn[6];          // LDS cell of this lane's position in each of the wave's 6 position tiles
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int nt = j < 5 ? 5 * w + j : 20;
        const int n = 16 * nt + pcol;
        const int nn = n < R8_NPOS ? n : R8_NPOS - 1;
        const int y = nn / W8_W;
        const int x = nn - y * W8_W;
        qn[j] = (y + 1) * R8_RS + x + 1;
    }
    const bool xvalid = w < 3 && (16 * 20 + pcol) < R8_NPOS;   // this lane's slot of the extra tile is a real output

    for (int clip = blockIdx.x; clip < p.B; clip += gridDim.x) {
        __syncthreads();  // previous clip's tail has consumed red/mvec; zero-fill (first clip) is complete

        // ------------------------------------------------------------ conv_0 + ReLU + AvgPool(4,3) on the VALU
        {
            const float* feat = p.feat + (size_t)clip * p.T * p.F;
            const int pair_beg = 6 * w;
            const int pair_end = w < 3 ? 6 * w + 6 : 23;
            for (int pass = 0; pass < ((p.debug & 1) ? 0 : 6); ++pass) {
                const int pos = pass * 64 + lane;
                const bool pv = pos < R8_NPOS;
                const int pp = pv ? pos : R8_NPOS - 1;
                const int py = pp / W8_W;
                const int px = pp - py * W8_W;
                float in[6][5];
#pragma unroll
                for (int r = 0; r < 6; ++r)
#pragma unroll
                    for (int c = 0; c < 5; ++c) {
                        const int row = 4 * py - 1 + r;
                        const int col = 3 * px - 1 + c;
                        const bool ok = row >= 0 && col >= 0;
                        const float v = feat[(ok ? row : 0) * p.F + (ok ? col : 0)];
                        in[r][c] = ok ? v : 0.f;
                    }
                const int cell = (py + 1) * R8_RS + px + 1;
                for (int pr = pair_beg; pr < pair_end; ++pr) {
                    const f32x2* wp = reinterpret_cast<const f32x2*>(p.w0) + pr * 9;
                    f32x2 wk[9];
#pragma unroll
                    for (int t = 0; t < 9; ++t) wk[t] = wp[t];
                    f32x2 sum = (f32x2){0.f, 0.f};
#pragma unroll
                    for (int oy = 0; oy < 4; ++oy)
#pragma unroll
                        for (int ox = 0; ox < 3; ++ox) {
                            f32x2 a = (f32x2){0.f, 0.f};
#pragma unroll
                            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                                for (int kx = 0; kx < 3; ++kx) {
                                    const float xv = in[oy + ky][ox + kx];
                                    a = __builtin_elementwise_fma(wk[ky * 3 + kx], (f32x2){xv, xv}, a);
                                }
                            sum.x += fmaxf(a.x, 0.f);
                            sum.y += fmaxf(a.y, 0.f);
                        }
                    if (pv) {   // channel 45 (pair 22, second half) has zero weights and lands in a spare plane
                        act[(2 * pr) * R8_CS + cell] = sum.x / 12.0f;
                        act[(2 * pr + 1) * R8_CS + cell] = sum.y / 12.0f;
                    }
                }
            }
        }
        __syncthreads();

        // ------------------------------------------------------------ prev_x <- pooled map, accumulator layout
        f32x4 prev[5][3], prevx;
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
            for (int m = 0; m < 3; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) prev[j][m][r] = act[(16 * m + 4 * g + r) * R8_CS + qn[j]];
#pragma unroll
        for (int r = 0; r < 4; ++r) prevx[r] = act[(16 * mx + 4 * g + r) * R8_CS + qn[5]];

        // ------------------------------------------------------------ conv_1 .. conv_6 on the matrix cores
        for (int layer = 0; layer < R8_LAYERS; ++layer) {
            f32x4 acc[5][3], accx;
#pragma unroll
            for (int j = 0; j < 5; ++j)
#pragma unroll
                for (int m = 0; m < 3; ++m) acc[j][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
            accx = (f32x4){0.f, 0.f, 0.f, 0.f};

            // A fragments: one 16-byte load per lane per (group of 4 k-steps, channel tile); software-pipelined one
            // group ahead so the L2 latency of group i+1 hides under the 48..64 MFMAs of group i.
            const f32x4* A = p.apk + (size_t)layer * R8_GROUPS * 3 * 64 + lane;
            f32x4 n0 = A[0], n1 = A[64], n2 = A[128], nx = A[mx * 64];
            for (int tap = 0; tap < ((p.debug & 2) ? 0 : 9); ++tap) {
                const int ty = tap / 3;
                const int tapoff = (ty - 1) * R8_RS + (tap - 3 * ty - 1) + g * R8_CS;
                int bt[6];
#pragma unroll
                for (int j = 0; j < 6; ++j) bt[j] = qn[j] + tapoff;
#pragma unroll
                for (int grp = 0; grp < 3; ++grp) {
                    const f32x4 a0 = n0, a1 = n1, a2 = n2, ax = nx;
                    const f32x4* An = A + (size_t)((tap * 3 + grp + 1) * 3) * 64;   // group 27 = the channel-44 group
                    n0 = An[0]; n1 = An[64]; n2 = An[128]; nx = An[mx * 64];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int step = 4 * grp + q;
                        if (step < 11) R8_STEP(a0[q], a1[q], a2[q], ax[q], bt[j] + step * (4 * R8_CS))
                    }
                }
            }
            // input channel 44: k-slot g of step q is tap (ky = q, kx = g); slot g = 3 carries a zero weight
#pragma unroll
            for (int q = 0; q < 3; ++q) R8_STEP(n0[q], n1[q], n2[q], nx[q], qn[j] + 44 * R8_CS + (q - 1) * R8_RS + g - 1)

            // ---- epilogue: ReLU, residual on even layers (reference i = layer + 1), BatchNorm on write
            const bool even = (layer & 1) != 0;
            const float* bm = p.bn_mean + layer * 48 + 4 * g;
            const float* br = p.bn_rstd + layer * 48 + 4 * g;
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                const f32x4 mu = *reinterpret_cast<const f32x4*>(bm + 16 * m);
                const f32x4 rs = *reinterpret_cast<const f32x4*>(br + 16 * m);
#pragma unroll
                for (int j = 0; j < 5; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = fmaxf(acc[j][m][r], 0.f);
                        const float vr = v + prev[j][m][r];
                        v = even ? vr : v;
                        prev[j][m][r] = even ? vr : prev[j][m][r];
                        acc[j][m][r] = (v - mu[r]) * rs[r];
                    }
            }
            {
                const f32x4 mu = *reinterpret_cast<const f32x4*>(bm + 16 * mx);
                const f32x4 rs = *reinterpret_cast<const f32x4*>(br + 16 * mx);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = fmaxf(accx[r], 0.f);
                    const float vr = v + prevx[r];
                    v = even ? vr : v;
                    prevx[r] = even ? vr : prevx[r];
                    accx[r] = (v - mu[r]) * rs[r];
                }
            }

            __syncthreads();  // every wave has finished reading this layer's input map
            if (layer < R8_LAYERS - 1) {
                // tiles 0..19 hold real positions only; rows 45..47 of channel tile 2 are zeros into spare planes
#pragma unroll
                for (int j = 0; j < 5; ++j)
#pragma unroll
                    for (int m = 0; m < 3; ++m)
#pragma unroll
                        for (int r = 0; r < 4; ++r) act[(16 * m + 4 * g + r) * R8_CS + qn[j]] = acc[j][m][r];
                if (xvalid) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) act[(16 * mx + 4 * g + r) * R8_CS + qn[5]] = accx[r];
                }
                __syncthreads();
            } else {
                // -------------------------------------------------------- spatial mean + Linear(45, n_labels)
#pragma unroll
                for (int m = 0; m < 3; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = acc[0][m][r] + acc[1][m][r] + acc[2][m][r] + acc[3][m][r] + acc[4][m][r];
                        v += __shfl_xor(v, 8);
                        v += __shfl_xor(v, 4);
                        v += __shfl_xor(v, 2);
                        v += __shfl_xor(v, 1);
                        if (pcol == 0) red[w * 48 + 16 * m + 4 * g + r] = v;
                    }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = xvalid ? accx[r] : 0.f;
                    v += __shfl_xor(v, 8);
                    v += __shfl_xor(v, 4);
                    v += __shfl_xor(v, 2);
                    v += __shfl_xor(v, 1);
                    if (pcol == 0 && w < 3) red[w * 48 + 16 * mx + 4 * g + r] += v;
                }
                __syncthreads();
                if (tid < 48) mvec[tid] = (red[tid] + red[48 + tid] + red[96 + tid] + red[144 + tid]) / (float)R8_NPOS;
                __syncthreads();
                if (tid < p.n_labels) {
                    const float* wr = p.out_w + tid * R8_C;
                    float o = 0.f;
                    for (int c = 0; c < R8_C; ++c) o = fmaf(wr[c], mvec[c], o);
                    p.logits[(size_t)clip * p.n_labels + tid] = o + p.out_b[tid];
                }
            }
        }
    }
}

hipError_t launch_res8(const Res8Params& p, int grid, hipStream_t s) {
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)res8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    if (p.B <= 0) return hipSuccess;
    // p.debug bit 2 (timing experiments only): pad LDS so that only one workgroup fits a CU
    const size_t lds = (p.debug & 4) ? (size_t)100 * 1024 : res8_lds_bytes();
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

}  // namespace kws
