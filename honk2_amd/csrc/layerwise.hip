// Layer-by-layer kernels for every model the fused res8 path does not cover: res15 / res26 / narrow variants /
// hey_snips (reference model/resnet.py:38-60, any n_layers, n_feature_maps, dilation, pool) and the cnn-* family
// (reference model/cnn.py:79-107: rectangular valid convs with stride and bias, MaxPool, up to four Linears).
//
// One implicit-GEMM kernel on the fp32 matrix cores (v_mfma_f32_16x16x4_f32) serves all of them:
//   M = output channels (tiles of 16, MT tiles per wave), N = B*Ho*Wo output positions flattened over the batch
//   (64 per wave, 256 per workgroup), K = taps x input channels.
//   * K order (ky, kx, cin padded to 4) for Cin > 1: within a tap the four k-slots of an MFMA are four consecutive
//     input channels; (ky, kx padded to 4) for Cin == 1 (every conv_0, and every Linear, which is run as a 1 x K
//     "conv" over the flattened feature vector): the four k-slots are four consecutive columns.
//   * fp32 VALU instructions cost matrix-pipe time on gfx950 (see res8_fused.hip), so the inner loop has none:
//     B fragments are buffer loads whose address is (per-tap lane register) + (scalar channel-group offset) and whose
//     hardware range check returns 0 for padding taps (out-of-bounds lanes carry an offset past the buffer);
//     A fragments are buffer loads at lane*4 + scalar step offset.  The next k-step's fragments are requested
//     before the current step's MFMAs.
//   * BatchNorm of the PREVIOUS layer never appears in the loop: its scale is folded into this layer's weights on
//     the host, and its shift -- which must NOT be applied to zero-padding taps -- becomes a 16-entry "border bias"
//     per output channel, indexed by which of the top / bottom / left / right tap rows are in bounds.
//   * ReLU, bias, border bias and the residual add are fused into the epilogue.  With `accumulate` the output buffer
//     already holds prev_x and becomes the new prev_x, so a ResNet needs two activation buffers.
//   Activations are fp32 (B, C, H, W) in the caller-provided workspace.
#include "kws_internal.h"

#include <algorithm>

namespace kws {

namespace {
constexpr int OOB = (int)0x80000000;   // byte offset beyond any activation buffer (they are <= 1 GiB)
__device__ __forceinline__ float bload(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
}  // namespace

template <int MT, bool KX>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvGeom gm, ConvArgs a) {
    if (range_gate_closed(a.rg)) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4;
    const int pcol = lane & 15;
    const int npc = gm.Ho * gm.Wo;
    const long long ntot = (long long)gm.B * npc;
    const long long n0 = ((long long)blockIdx.x * 4 + w) * 64;
    const int hw = gm.H * gm.W;
    const bool padded = gm.ph > 0 || gm.pw > 0;

    bool valid[4];
    int iy0[4], ix0[4], inb[4], pos[4], bidx[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const long long n = n0 + 16 * j + pcol;
        valid[j] = n < ntot;
        const long long nn = valid[j] ? n : ntot - 1;
        const int b = (int)(nn / npc);
        const int ps = (int)(nn - (long long)b * npc);
        const int oy = ps / gm.Wo;
        const int ox = ps - oy * gm.Wo;
        bidx[j] = b;
        pos[j] = ps;
        iy0[j] = oy * gm.sh - gm.ph;
        ix0[j] = ox * gm.sw - gm.pw;
        inb[j] = b * gm.Cin * hw;
    }

    const __amdgpu_buffer_rsrc_t rin =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), 0, (int)((size_t)gm.B * gm.Cin * hw * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rwt = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.apk + (size_t)blockIdx.y * gm.ksteps * MT * 64), 0, gm.ksteps * MT * 256, 0x00020000);

    // Outer index t ("tap"): the unit whose lane addresses / bounds are computed once (VALU); inner index: steps that
    // only advance a SCALAR byte offset.  Unpadded Cin == 1 convs and Linears need no bounds at all, so their whole
    // K range is one "tap" and the (ky, kx-group) walk is scalar arithmetic.
    const bool flat = KX && !padded;
    const int ntaps = KX ? (padded ? gm.kh * gm.inner_steps : 1) : gm.kh * gm.kw;
    // split-K (flat mode only): this workgroup covers k-steps [s_begin, s_begin + ncg)
    const int s_begin = (flat && gm.ksplit > 1) ? (int)blockIdx.z * gm.ksteps_split : 0;
    const int ncg = KX ? (padded ? 1 : min(gm.ksteps - s_begin, (gm.ksplit > 1 ? gm.ksteps_split : gm.ksteps)))
                       : gm.inner_steps;
    const int sstride = (KX ? 4 * gm.dw : 4 * hw) * 4;                       // bytes per inner step
    const int rowjump = flat ? (gm.dh * gm.W - gm.inner_steps * 4 * gm.dw) * 4 : 0;   // extra bytes when kx wraps to the next ky

    int voff[4];
    auto tap_voff = [&](int t) {
        if (KX) {
            const int ky = padded ? t / gm.inner_steps : 0;
            const int kxl = (padded ? 4 * (t - ky * gm.inner_steps) : 0) + g;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int iy = iy0[j] + ky * gm.dh;
                const int ix = ix0[j] + kxl * gm.dw;
                bool ok = valid[j];
                if (padded) ok = ok && iy >= 0 && iy < gm.H && ix >= 0 && ix < gm.W && kxl < gm.kw;
                voff[j] = ok ? (inb[j] + iy * gm.W + ix) * 4 : OOB;
            }
        } else {
            const int ky = t / gm.kw;
            const int kx = t - ky * gm.kw;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int iy = iy0[j] + ky * gm.dh;
                const int ix = ix0[j] + kx * gm.dw;
                const bool ok = valid[j] && iy >= 0 && iy < gm.H && ix >= 0 && ix < gm.W;
                voff[j] = ok ? (inb[j] + iy * gm.W + ix + g * hw) * 4 : OOB;
            }
        }
    };

    f32x4 acc[MT][4];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[m][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int avoff = lane * 4;
    int sa = s_begin * MT * 256;   // scalar byte offset of the current k-step's A fragments (steps are consecutive across taps)
    const int ky_begin = flat ? s_begin / gm.inner_steps : 0;
    const int xg_begin = flat ? s_begin - ky_begin * gm.inner_steps : 0;

#define LW_LOAD(AR, BR)                                                                     \
    {                                                                                       \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) BR[j] = bload(rin, voff[j], sb);      \
        _Pragma("unroll") for (int m = 0; m < MT; ++m) AR[m] = bload(rwt, avoff + m * 256, sa); \
        __builtin_amdgcn_sched_barrier(0); /* keep the prefetch ABOVE the MFMAs it overlaps */ \
    }
#define LW_MFMA(AR, BR)                                                                     \
    {                                                                                       \
        _Pragma("unroll") for (int m = 0; m < MT; ++m)                                      \
            _Pragma("unroll") for (int j = 0; j < 4; ++j)                                   \
                acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(AR[m], BR[j], acc[m][j], 0, 0, 0); \
        __builtin_amdgcn_sched_barrier(0);                                                  \
    }
#define LW_ADVANCE()                                                                        \
    {                                                                                       \
        sb += sstride;                                                                      \
        sa += MT * 256;                                                                     \
        if (flat && ++xg == gm.inner_steps) {                                               \
            xg = 0;                                                                         \
            sb += rowjump;                                                                  \
        }                                                                                   \
    }

    for (int t = 0; t < ntaps; ++t) {
        tap_voff(t);
        // Ping-pong pipeline over this tap's ncg steps: the loads of step i+1 are issued, straight-line, before the
        // MFMAs of step i; the last one or two steps are peeled so every load in the loop is a real one.
        int sb = (ky_begin * gm.dh * gm.W + xg_begin * 4 * gm.dw) * 4, xg = xg_begin, i = 0;
        float a0[MT], b0[4], a1[MT], b1[4];
        LW_LOAD(a0, b0)
        while (i + 2 < ncg) {
            LW_ADVANCE()
            LW_LOAD(a1, b1)
            LW_MFMA(a0, b0)
            LW_ADVANCE()
            LW_LOAD(a0, b0)
            LW_MFMA(a1, b1)
            i += 2;
        }
        if (i + 1 < ncg) {
            LW_ADVANCE()
            LW_LOAD(a1, b1)
            LW_MFMA(a0, b0)
            LW_MFMA(a1, b1)
        } else {
            LW_MFMA(a0, b0)
        }
        sa += MT * 256;   // first step of the next tap
    }
#undef LW_LOAD
#undef LW_MFMA
#undef LW_ADVANCE

    // border class of each position (3x3 "same" convs only): bit0 top row of taps in bounds, bit1 bottom, bit2 left, bit3 right
    int bmask[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
        bmask[j] = (iy0[j] >= 0 ? 1 : 0) | (iy0[j] + 2 * gm.dh < gm.H ? 2 : 0) | (ix0[j] >= 0 ? 4 : 0) |
                   (ix0[j] + 2 * gm.dw < gm.W ? 8 : 0);

    // epilogue: D[row = 4g + r][col = pcol]; rows are output channels
    float amax = 0.f;   // largest magnitude stored (fp16 range guard)
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = ((int)blockIdx.y * MT + m) * 16 + 4 * g + r;
            if (co >= gm.Cout) continue;
            const float bias = a.bias ? a.bias[co] : 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (!valid[j]) continue;
                if (gm.ksplit > 1) {   // raw partial sum; bias / ReLU are applied by the reduce kernel
                    a.partial[((size_t)blockIdx.z * gm.B + bidx[j]) * gm.Cout * npc + (size_t)co * npc + pos[j]] = acc[m][j][r];
                    continue;
                }
                float v = acc[m][j][r] + bias;
                if (a.border) v += a.border[bmask[j] * gm.Cout + co];
                if (gm.relu) v = fmaxf(v, 0.f);
                const size_t idx = ((size_t)bidx[j] * gm.Cout + co) * npc + pos[j];
                if (gm.accumulate) v += a.out[idx];
                a.out[idx] = v;
                amax = fmaxf(amax, fabsf(v));
            }
        }
    range_note(a.rg, amax);
}

int choose_mt(int mtiles) {
    int best = 1, waste = 1 << 30;
    for (int mt = 4; mt >= 1; --mt) {
        const int wst = (mtiles + mt - 1) / mt * mt - mtiles;
        if (wst < waste) {
            waste = wst;
            best = mt;
        }
    }
    return best;
}

template <int MT>
static hipError_t launch_conv_mt(const ConvGeom& g, const ConvArgs& a, hipStream_t s) {
    const long long ntot = (long long)g.B * g.Ho * g.Wo;
    dim3 grid((unsigned)((ntot + 255) / 256), (unsigned)((g.mtiles + MT - 1) / MT), (unsigned)(g.ksplit > 1 ? g.ksplit : 1));
    if (g.kx_inner)
        hipLaunchKernelGGL((conv_igemm_kernel<MT, true>), grid, dim3(256), 0, s, g, a);
    else
        hipLaunchKernelGGL((conv_igemm_kernel<MT, false>), grid, dim3(256), 0, s, g, a);
    return hipGetLastError();
}

hipError_t launch_conv(const ConvGeom& g, const ConvArgs& a, hipStream_t s) {
    if (g.B <= 0) return hipSuccess;
    switch (g.MT) {
        case 1: return launch_conv_mt<1>(g, a, s);
        case 2: return launch_conv_mt<2>(g, a, s);
        case 3: return launch_conv_mt<3>(g, a, s);
        case 4: return launch_conv_mt<4>(g, a, s);
    }
    return hipErrorInvalidValue;
}

// Host: weights (Cout, Cin, kh, kw) -> [mgroup][kstep][MT][lane]; lane = (k-slot g << 4) | row.
void pack_conv_weights(const ConvGeom& g, const float* w, std::vector<float>& dst) {
    const int mgroups = (g.mtiles + g.MT - 1) / g.MT;
    dst.assign((size_t)mgroups * g.ksteps * g.MT * 64, 0.f);
    for (int mg = 0; mg < mgroups; ++mg)
        for (int ks = 0; ks < g.ksteps; ++ks)
            for (int m = 0; m < g.MT; ++m)
                for (int lane = 0; lane < 64; ++lane) {
                    const int co = (mg * g.MT + m) * 16 + (lane & 15);
                    const int slot = lane >> 4;
                    int c, ky, kx;
                    if (g.kx_inner) {
                        ky = ks / g.inner_steps;
                        kx = 4 * (ks % g.inner_steps) + slot;
                        c = 0;
                    } else {
                        const int tap = ks / g.inner_steps;
                        ky = tap / g.kw;
                        kx = tap % g.kw;
                        c = 4 * (ks % g.inner_steps) + slot;
                    }
                    float v = 0.f;
                    if (co < g.Cout && c < g.Cin && kx < g.kw && ky < g.kh)
                        v = w[(((size_t)co * g.Cin + c) * g.kh + ky) * g.kw + kx];
                    dst[(((size_t)mg * g.ksteps + ks) * g.MT + m) * 64 + lane] = v;
                }
}

// ------------------------------------------------------------------------------------------------ split-K reduce
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                                            const float* __restrict__ bias, int ksplit, long long total,
                                                            int Cout, int npc, int relu, RangeGate rg) {
    if (range_gate_closed(rg)) return;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    float v = bias ? bias[(int)((i / npc) % Cout)] : 0.f;
    // fixed order: deterministic.  Eight partials are requested before the first of them is added: with one load per add every one of the
    // ~50 additions waited for its own round trip (r3: 18 -> 4 us per 1 024-clip Linear of cnn-trad-pool2)
    for (int z = 0; z < ksplit; z += 8) {
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = partial[(size_t)min(z + u, ksplit - 1) * total + i];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (z + u < ksplit) v += t[u];
    }
    v = relu ? fmaxf(v, 0.f) : v;
    out[i] = v;
    range_note(rg, fabsf(v));
}

hipError_t launch_splitk_reduce(const float* partial, float* out, const float* bias, int ksplit, long long total,
                                int Cout, int npc, int relu, hipStream_t s, RangeGate rg) {
    if (total <= 0) return hipSuccess;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, partial, out, bias,
                       ksplit, total, Cout, npc, relu, rg);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ pooling
// AvgPool2d / MaxPool2d with stride == kernel, floor mode (reference model/resnet.py:34, model/cnn.py:30,43).
__global__ __launch_bounds__(256) void pool_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                   long long total, int H, int W, int Ho, int Wo, int kh, int kw,
                                                   int is_max, RangeGate rg) {
    if (range_gate_closed(rg)) return;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int ox = (int)(i % Wo);
    const long long t = i / Wo;
    const int oy = (int)(t % Ho);
    const long long plane = t / Ho;
    const float* src = in + (plane * H + (long long)oy * kh) * W + (long long)ox * kw;
    float v = is_max ? -INFINITY : 0.f;
    for (int y = 0; y < kh; ++y)
        for (int x = 0; x < kw; ++x) {
            const float s = src[y * W + x];
            v = is_max ? fmaxf(v, s) : v + s;
        }
    v = is_max ? v : v / (float)(kh * kw);
    out[i] = v;
    range_note(rg, fabsf(v));
}

// fp16 range guard for tensors the library does not produce itself (the feature maps handed to a CNN)
__global__ __launch_bounds__(256) void range_check_kernel(const float* __restrict__ x, long long n, RangeGate rg) {
    float amax = 0.f;
    bool odd = false;   // a NaN: fmaxf would drop it
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float v = x[i];
        odd |= v != v;
        amax = fmaxf(amax, fabsf(v));
    }
    range_note(rg, odd ? INFINITY : amax);
}

hipError_t launch_range_check(const float* x, long long n, hipStream_t s, RangeGate rg) {
    if (n <= 0 || !rg.flag) return hipSuccess;
    const unsigned grid = (unsigned)std::min<long long>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(range_check_kernel, dim3(grid), dim3(256), 0, s, x, n, rg);
    return hipGetLastError();
}

hipError_t launch_pool(const float* in, float* out, int planes, int H, int W, int kh, int kw, int is_max,
                       hipStream_t s, RangeGate rg) {
    const int Ho = H / kh, Wo = W / kw;
    const long long total = (long long)planes * Ho * Wo;
    if (total <= 0) return hipSuccess;
    hipLaunchKernelGGL(pool_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, in, out, total, H, W, Ho,
                       Wo, kh, kw, is_max, rg);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ mean + linear
// ResNet tail (reference model/resnet.py:57-59) with the last BatchNorm folded in: mean(BN(x)) == BN(mean(x)).
__global__ __launch_bounds__(256) void mean_linear_kernel(const float* __restrict__ x, float* __restrict__ logits,
                                                          int C, int HW, const float* mean, const float* rstd,
                                                          const float* __restrict__ wt, const float* __restrict__ bias,
                                                          int n_out, RangeGate rg) {
    extern __shared__ float mv[];
    if (range_gate_closed(rg)) return;
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    for (int c = w; c < C; c += 4) {
        const float* src = x + ((size_t)b * C + c) * HW;
        float s = 0.f;
        for (int i = lane; i < HW; i += 64) s += src[i];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
        if (lane == 0) {
            float m = s / (float)HW;
            if (mean) m = (m - mean[c]) * rstd[c];
            mv[c] = m;
        }
    }
    __syncthreads();
    for (int o = threadIdx.x; o < n_out; o += 256) {
        float v = 0.f;
        for (int c = 0; c < C; ++c) v = fmaf(wt[o * C + c], mv[c], v);
        logits[(size_t)b * n_out + o] = v + bias[o];
    }
}

hipError_t launch_mean_linear(const float* x, float* logits, int B, int C, int HW, const float* mean,
                              const float* rstd, const float* w, const float* bias, int n_out, hipStream_t s, RangeGate rg) {
    if (B <= 0) return hipSuccess;
    hipLaunchKernelGGL(mean_linear_kernel, dim3((unsigned)B), dim3(256), (size_t)C * sizeof(float), s, x, logits, C,
                       HW, mean, rstd, w, bias, n_out, rg);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ evaluation tail
// ce_loss (loss_function.py:6-9) + Acc.accumulate (metric/acc.py:14-24) + PerClassAcc.accumulate
// (metric/per_class_acc.py:14-45) in one pass: stats = [correct, total, per-class correct[n], per-class total[n], bad].
// A target outside [0, n) (the reference's F.cross_entropy raises on one) is never used as an index: the clip is skipped
// and counted in stats[2 + 2n], which the caller checks after its one device-to-host copy.  NaN logits take the argmax
// as in torch.argmax (the first NaN wins).
__global__ __launch_bounds__(256) void eval_tail_kernel(const float* __restrict__ logits,
                                                        const int64_t* __restrict__ target, int B, int n,
                                                        unsigned long long* stats, double* loss_sum) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    double loss = 0.0;
    if (b < B) {
        const float* z = logits + (size_t)b * n;
        const int64_t t64 = target[b];
        if (t64 < 0 || t64 >= (int64_t)n) {
            atomicAdd(&stats[2 + 2 * n], 1ull);
        } else {
            const int t = (int)t64;
            int arg = 0;
            float zmax = z[0];
            for (int i = 1; i < n; ++i)
                if (z[i] > zmax || (z[i] != z[i] && zmax == zmax)) {   // first maximum wins, NaN counts as the maximum
                    zmax = z[i];
                    arg = i;
                }
            double se = 0.0;
            for (int i = 0; i < n; ++i) se += exp((double)z[i] - (double)zmax);
            loss = log(se) - ((double)z[t] - (double)zmax);
            const bool hit = arg == t;
            atomicAdd(&stats[1], 1ull);
            atomicAdd(&stats[2 + n + t], 1ull);
            if (hit) {
                atomicAdd(&stats[0], 1ull);
                atomicAdd(&stats[2 + t], 1ull);
            }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) loss += __shfl_xor(loss, off);
    if ((threadIdx.x & 63) == 0 && loss != 0.0) atomicAdd(loss_sum, loss);
}

hipError_t launch_eval_tail(const float* logits, const int64_t* target, int B, int n_labels, int64_t* stats,
                            double* loss_sum, hipStream_t s) {
    if (B <= 0) return hipSuccess;
    hipLaunchKernelGGL(eval_tail_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, s, logits, target, B,
                       n_labels, reinterpret_cast<unsigned long long*>(stats), loss_sum);
    return hipGetLastError();
}

}  // namespace kws
