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
constexpr int ACT_WORDS = R8_C * R8_CS;   // 18000
constexpr int RED_WORDS = 4 * 48;
constexpr int LDS_WORDS = ACT_WORDS + RED_WORDS + 48;
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

    int qn[6];          // LDS cell of this lane's position in each of the wave's 6 position tiles
    bool nvalid[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int nt = j < 5 ? 5 * w + j : 20;
        const int n = 16 * nt + pcol;
        nvalid[j] = n < R8_NPOS;
        const int nn = n < R8_NPOS ? n : R8_NPOS - 1;
        const int y = nn / W8_W;
        const int x = nn - y * W8_W;
        qn[j] = (y + 1) * R8_RS + x + 1;
    }

    for (int clip = blockIdx.x; clip < p.B; clip += gridDim.x) {
        __syncthreads();  // previous clip's tail has consumed red/mvec; zero-fill (first clip) is complete

        // ------------------------------------------------------------ conv_0 + ReLU + AvgPool(4,3) on the VALU
        {
            const float* feat = p.feat + (size_t)clip * p.T * p.F;
            const int pair_beg = 6 * w;
            const int pair_end = w < 3 ? 6 * w + 6 : 23;
            for (int pass = 0; pass < 6; ++pass) {
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
                    if (pv) {
                        act[(2 * pr) * R8_CS + cell] = sum.x / 12.0f;
                        if (2 * pr + 1 < R8_C) act[(2 * pr + 1) * R8_CS + cell] = sum.y / 12.0f;
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
                for (int r = 0; r < 4; ++r) {
                    const int co = 16 * m + 4 * g + r;
                    prev[j][m][r] = co < R8_C ? act[co * R8_CS + qn[j]] : 0.f;
                }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = 16 * mx + 4 * g + r;
            prevx[r] = co < R8_C ? act[co * R8_CS + qn[5]] : 0.f;
        }

        float msum[3][4], msumx[4];
#pragma unroll
        for (int m = 0; m < 3; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) msum[m][r] = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) msumx[r] = 0.f;

        // ------------------------------------------------------------ conv_1 .. conv_6 on the matrix cores
        for (int layer = 0; layer < R8_LAYERS; ++layer) {
            f32x4 acc[5][3], accx;
#pragma unroll
            for (int j = 0; j < 5; ++j)
#pragma unroll
                for (int m = 0; m < 3; ++m) acc[j][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
            accx = (f32x4){0.f, 0.f, 0.f, 0.f};

            const f32x4* A = p.apk + (size_t)layer * R8_GROUPS * 3 * 64 + lane;
            for (int tap = 0; tap < 9; ++tap) {
                const int ty = tap / 3;
                const int tapoff = (ty - 1) * R8_RS + (tap - 3 * ty - 1) + g * R8_CS;
                int bt[6];
#pragma unroll
                for (int j = 0; j < 6; ++j) bt[j] = qn[j] + tapoff;
#pragma unroll
                for (int grp = 0; grp < 3; ++grp) {
                    const f32x4* Ag = A + (size_t)((tap * 3 + grp) * 3) * 64;
                    const f32x4 a0 = Ag[0], a1 = Ag[64], a2 = Ag[128], ax = Ag[mx * 64];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int step = 4 * grp + q;
                        if (step < 11) {
                            float b[6];
#pragma unroll
                            for (int j = 0; j < 6; ++j) b[j] = act[bt[j] + step * (4 * R8_CS)];
#pragma unroll
                            for (int j = 0; j < 5; ++j) {
                                acc[j][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[q], b[j], acc[j][0], 0, 0, 0);
                                acc[j][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[q], b[j], acc[j][1], 0, 0, 0);
                                acc[j][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[q], b[j], acc[j][2], 0, 0, 0);
                            }
                            accx = __builtin_amdgcn_mfma_f32_16x16x4f32(ax[q], b[5], accx, 0, 0, 0);
                        }
                    }
                }
            }
            {   // input channel 44: k-slot g of step q is tap (ky = q, kx = g); slot g = 3 carries a zero weight
                const f32x4* Ag = A + (size_t)(27 * 3) * 64;
                const f32x4 a0 = Ag[0], a1 = Ag[64], a2 = Ag[128], ax = Ag[mx * 64];
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    float b[6];
#pragma unroll
                    for (int j = 0; j < 6; ++j) b[j] = act[qn[j] + 44 * R8_CS + (q - 1) * R8_RS + g - 1];
#pragma unroll
                    for (int j = 0; j < 5; ++j) {
                        acc[j][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[q], b[j], acc[j][0], 0, 0, 0);
                        acc[j][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[q], b[j], acc[j][1], 0, 0, 0);
                        acc[j][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[q], b[j], acc[j][2], 0, 0, 0);
                    }
                    accx = __builtin_amdgcn_mfma_f32_16x16x4f32(ax[q], b[5], accx, 0, 0, 0);
                }
            }

            // ---- epilogue: ReLU, residual on even layers (reference i = layer + 1), BatchNorm on write
            const bool even = (layer & 1) != 0;
            const bool last = layer == R8_LAYERS - 1;
            const float* bm = p.bn_mean + layer * 48 + 4 * g;
            const float* br = p.bn_rstd + layer * 48 + 4 * g;
            f32x4 mu[3], rs[3];
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                mu[m] = *reinterpret_cast<const f32x4*>(bm + 16 * m);
                rs[m] = *reinterpret_cast<const f32x4*>(br + 16 * m);
            }
            const f32x4 mux = *reinterpret_cast<const f32x4*>(bm + 16 * mx);
            const f32x4 rsx = *reinterpret_cast<const f32x4*>(br + 16 * mx);

#pragma unroll
            for (int j = 0; j < 5; ++j)
#pragma unroll
                for (int m = 0; m < 3; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = fmaxf(acc[j][m][r], 0.f);
                        if (even) {
                            v += prev[j][m][r];
                            prev[j][m][r] = v;
                        }
                        acc[j][m][r] = (v - mu[m][r]) * rs[m][r];
                    }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = fmaxf(accx[r], 0.f);
                if (even) {
                    v += prevx[r];
                    prevx[r] = v;
                }
                accx[r] = (v - mux[r]) * rsx[r];
            }

            __syncthreads();  // every wave has finished reading this layer's input map
            if (!last) {
#pragma unroll
                for (int j = 0; j < 5; ++j)
#pragma unroll
                    for (int m = 0; m < 3; ++m)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int co = 16 * m + 4 * g + r;
                            if (co < R8_C) act[co * R8_CS + qn[j]] = acc[j][m][r];   // tiles 0..19 hold real positions only
                        }
                if (w < 3) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int co = 16 * mx + 4 * g + r;
                        if (co < R8_C && nvalid[5]) act[co * R8_CS + qn[5]] = accx[r];
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < 5; ++j)
#pragma unroll
                    for (int m = 0; m < 3; ++m)
#pragma unroll
                        for (int r = 0; r < 4; ++r) msum[m][r] += acc[j][m][r];
#pragma unroll
                for (int r = 0; r < 4; ++r) msumx[r] = nvalid[5] ? accx[r] : 0.f;
            }
            __syncthreads();
        }

        // ------------------------------------------------------------ spatial mean + Linear(45, n_labels)
#pragma unroll
        for (int m = 0; m < 3; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = msum[m][r];
                v += __shfl_xor(v, 8);
                v += __shfl_xor(v, 4);
                v += __shfl_xor(v, 2);
                v += __shfl_xor(v, 1);
                if (pcol == 0) red[w * 48 + 16 * m + 4 * g + r] = v;
            }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = msumx[r];
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

hipError_t launch_res8(const Res8Params& p, int grid, hipStream_t s) {
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)res8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)res8_lds_bytes());
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    if (p.B <= 0) return hipSuccess;
    hipLaunchKernelGGL(res8_kernel, dim3((unsigned)grid), dim3(256), res8_lds_bytes(), s, p);
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
