// Layer-wise implicit-GEMM convolution with fp32-accurate products on the bf16 matrix cores ("bf16x6").
//
// Same contract, tensor layouts (fp32 (B, C, H, W) activations) and epilogue as conv_igemm_kernel in layerwise.hip
// -- it is a drop-in replacement for the cases it supports: any conv with Cin > 1, and UNPADDED convs with Cin == 1
// (every cnn-* conv_0 and every Linear).  See res8_bf16x6.hip for why: the fp32-input MFMA runs at 1/16 of the bf16
// MFMA rate and blocks the VALU; six bf16 MFMA terms of three-way bf16 splits give the same (slightly better)
// accuracy at 6/16 of the cost, and the VALU work of splitting co-executes with other waves' bf16 MFMAs.
//
//   * K is walked in blocks of 8: 8 consecutive input channels of one tap (Cin > 1), or 8 consecutive kernel columns of
//     one kernel row (Cin == 1).  One v_mfma_f32_16x16x32_bf16 consumes 4 blocks (one per 16-lane group g), so k-step s
//     covers blocks 4s..4s+3; lane group g decodes ITS block to (tap, channel block) / (row, column block).
//   * A (weights): split into three bf16 parts on the host, fragment order [mgroup][k-step][MT][part][lane][8].
//   * B (activations): each lane gathers its 8 fp32 values with 8 buffer loads at (per-block lane offset) + (scalar
//     element stride); out-of-bounds taps carry an offset past the buffer and read 0.  The values are split into three
//     bf16 parts in registers (44 VALU per fragment) right before use; the NEXT k-step's raw values and weights are
//     requested before the current step's MFMAs.
//   * per k-step and wave: MT x 4 tiles x 6 terms MFMAs.
#include "kws_internal.h"

#include <cstring>

namespace kws {

namespace {
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int OOB = (int)0x80000000;

__device__ __forceinline__ float bload(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ u32x4 bload4(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ unsigned pack2(float a, float b) {
    const bf16x2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float lo_f(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float hi_f(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }

// 8 fp32 values -> three bf16x8 fragments (x = h + m + l to 24 bits)
__device__ __forceinline__ void split8(const float (&x)[8], u32x4 (&out)[3]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float a = x[2 * i], b = x[2 * i + 1];
        const unsigned h = pack2(a, b);
        const float ra = a - lo_f(h), rb = b - hi_f(h);
        const unsigned m = pack2(ra, rb);
        const unsigned l = pack2(ra - lo_f(m), rb - hi_f(m));
        out[0][i] = h;
        out[1][i] = m;
        out[2][i] = l;
    }
}

// 8 fp32 values -> two fp16x8 fragments (x = h + l to 22 bits; exact products in three terms, see res8_f16x3.hip)
// (one packed convert + one mixed-precision FMA per value: l = fp16(x - float(h)), exact difference, same bits as
// convert-back / subtract / convert at a third of the instructions -- this kernel is bound by exactly this VALU work)
__device__ __forceinline__ void split8_f16(const float (&x)[8], u32x4 (&out)[3]) {
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_){x[2 * i], x[2 * i + 1]}, f16x2));
        unsigned l;
        asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(h), "v"(x[2 * i]));
        // the fragment goes straight into an MFMA: hipcc pads no hazards for instructions inside an asm statement, and a VALU
        // result needs two wait states before a matrix instruction may read it (without them the last pair's second part
        // was stale in the multi-pass pooling variants, where nothing else sits in between: logits off by 1e-2)
        asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\ts_nop 1" : "+v"(l) : "v"(h), "v"(x[2 * i + 1]));
        out[0][i] = h;
        out[1][i] = l;
    }
}

#define XMFH(A_, B_, C_) C_ = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, A_), __builtin_bit_cast(f16x8, B_), C_, 0, 0, 0)
#define XMF(A_, B_, C_) C_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A_), __builtin_bit_cast(bf16x8, B_), C_, 0, 0, 0)
#define XMF6(A3, B3, C_)            \
    if (F16) {                      \
        if (TERMS >= 3) {           \
            XMFH(A3[1], B3[0], C_); \
            XMFH(A3[0], B3[1], C_); \
        }                           \
        XMFH(A3[0], B3[0], C_);     \
    } else {                        \
        if (TERMS == 6) {           \
            XMF(A3[2], B3[0], C_);  \
            XMF(A3[1], B3[1], C_);  \
            XMF(A3[0], B3[2], C_);  \
        }                           \
        if (TERMS >= 3) {           \
            XMF(A3[1], B3[0], C_);  \
            XMF(A3[0], B3[1], C_);  \
        }                           \
        XMF(A3[0], B3[0], C_);      \
    }
}  // namespace

// MULTI: pooling windows of more than four members (several passes over K with a running maximum)
// F16 (the fp32-accurate default): two-part fp16 operands, three terms, weights scaled by 2^S (gm.x_inv_scale = 2^-S is applied
// in the epilogue); otherwise bf16 parts with TERMS = 6 / 3 / 1 products.
template <int MT, bool KX, int TERMS, bool MULTI, bool F16>
__global__ __launch_bounds__(256, 2) void conv_bf16x6_kernel(ConvGeom gm, ConvArgs a) {
    constexpr int LP = F16 ? 2 : 3;   // weight parts per fragment group
    if (range_gate_closed(a.rg)) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4;
    const int pcol = lane & 15;
    // Fused max-pool mode (gm.pool_h * gm.pool_w > 1, stride = window, floor): a wave owns 16 POOLED positions and its
    // position tiles are members of their windows -- up to four per pass over K, windows of 6 / 9 members in passes of three
    // -- so the maximum is taken over accumulators of one lane and the un-pooled map never exists.  Otherwise a wave owns
    // 64 consecutive output positions (one pass).
    const int nmem = gm.pool_h * gm.pool_w > 1 ? gm.pool_h * gm.pool_w : 0;   // 0: no pooling
    const int per_pass = nmem == 0 ? 4 : (nmem <= 4 ? nmem : (nmem % 3 == 0 ? 3 : 4));
    const int npass = MULTI ? (nmem + per_pass - 1) / per_pass : 1;
    const int Hq = nmem ? gm.Ho / gm.pool_h : gm.Ho, Wq = nmem ? gm.Wo / gm.pool_w : gm.Wo;
    const int npc = Hq * Wq;
    const long long ntot = (long long)gm.B * npc;
    const long long n0 = ((long long)blockIdx.x * 4 + w) * (nmem ? 16 : 64);
    const int hw = gm.H * gm.W;

    bool valid[4];
    int iy0[4], ix0[4], inb[4], pos[4], bidx[4];
    int ntile = 4;   // position tiles in use in the current pass (uniform)
    auto setup_tiles = [&](int pass) {
        ntile = nmem ? min(per_pass, nmem - pass * per_pass) : 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long long n = nmem ? n0 + pcol : n0 + 16 * j + pcol;
            valid[j] = n < ntot && j < ntile;
            const long long nn = n < ntot ? n : ntot - 1;
            const int b = (int)(nn / npc);
            const int ps = (int)(nn - (long long)b * npc);
            int oy = ps / Wq;
            int ox = ps - oy * Wq;
            if (nmem) {
                const int mi = min(pass * per_pass + j, nmem - 1);
                const int dy = mi / gm.pool_w;
                oy = oy * gm.pool_h + dy;
                ox = ox * gm.pool_w + (mi - dy * gm.pool_w);
            }
            bidx[j] = b;
            pos[j] = ps;
            iy0[j] = oy * gm.sh - gm.ph;
            ix0[j] = ox * gm.sw - gm.pw;
            inb[j] = b * gm.Cin * hw;
        }
    };
    setup_tiles(0);

    // Linears (one kernel row over a flat input whose rows are 16-byte aligned): a block is fetched with two 16-byte loads.  Eight
    // 4-byte loads of 16 clips 100 KB apart touched 64 cache lines per instruction: the first Linear of the cnn-* models (20 - 42 k
    // inputs) ran at 1.6 TB/s, bound by the vector-memory path's line rate.
    const bool vec8 = KX && conv_x_vec8(gm);
    const bool in16 = F16 && TERMS == 1 && vec8 && gm.in_f16 && (gm.W & 7) == 0;   // fp16 input cells (rows of whole 16-byte blocks): conv_x_in16_ok, checked by the launcher
    const __amdgpu_buffer_rsrc_t rin =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), 0, (int)((size_t)gm.B * gm.Cin * hw * (in16 ? 2 : 4)), 0x00020000);
    const int steps = gm.x_ksteps;
    const int s_begin = gm.ksplit > 1 ? (int)blockIdx.z * gm.ksteps_split : 0;
    const int s_end = gm.ksplit > 1 ? min(steps, s_begin + gm.ksteps_split) : steps;
    const __amdgpu_buffer_rsrc_t rwt = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned short*>(a.apk16 + (size_t)blockIdx.y * steps * MT * LP * 64 * 8), 0, steps * MT * LP * 1024,
        0x00020000);

    // inner/outer split of the block index: blocks per "row" (tap for Cin > 1, kernel row for Cin == 1)
    const int nbr = gm.x_blocks_per_row;
    const int nblocks = gm.x_blocks;
    const int estride = (KX ? gm.dw : hw) * 4;   // byte stride between the 8 elements of a block

    f32x4 acc[MT][4];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[m][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // this lane group's block at k-step s: bi = 4 s + g -> (row r, block-in-row c)
    auto block_voff = [&](int s, int (&voff)[4]) {
        const int bi = 4 * s + g;
        const int r = bi / nbr, c = bi - r * nbr;
        const bool bok = bi < nblocks;
        if (KX) {   // unpadded, Cin == 1: row = ky, block = 8 kernel columns
#pragma unroll
            for (int j = 0; j < 4; ++j)
                voff[j] = (valid[j] && bok) ? (inb[j] + (iy0[j] + r * gm.dh) * gm.W + ix0[j] + 8 * c * gm.dw) * (in16 ? 2 : 4) : OOB;
        } else {    // row = tap (ky, kx), block = 8 input channels
            const int ky = r / gm.kw, kx = r - ky * gm.kw;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int iy = iy0[j] + ky * gm.dh, ix = ix0[j] + kx * gm.dw;
                const bool ok = valid[j] && bok && iy >= 0 && iy < gm.H && ix >= 0 && ix < gm.W;
                voff[j] = ok ? (inb[j] + iy * gm.W + ix + 8 * c * hw) * 4 : OOB;
            }
        }
    };

    float raw0[4][8], raw1[4][8];

    // raw fp32 B values of k-step S: requested one whole step ahead (they feed VALU work, which needs them early)
#define XLOADB(RAW, S)                                                                                \
    {                                                                                                 \
        int voff_[4];                                                                                 \
        block_voff(S, voff_);                                                                         \
        if (KX && vec8 && in16) {   /* fp16 cells: the eight k-slots are ONE 16-byte load, already the operand */ \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                           \
                const u32x4 h_ = bload4(rin, voff_[j], 0);                                            \
                _Pragma("unroll") for (int e = 0; e < 4; ++e) RAW[j][e] = __builtin_bit_cast(float, (unsigned)h_[e]); \
            }                                                                                         \
        } else if (KX && vec8) {   /* Linear: the eight k-slots are 32 contiguous, 16-byte aligned bytes */  \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                           \
                const u32x4 lo_ = bload4(rin, voff_[j], 0), hi_ = bload4(rin, voff_[j], 16);          \
                _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                       \
                    const unsigned wl_ = lo_[e], wh_ = hi_[e];                                        \
                    RAW[j][e] = __builtin_bit_cast(float, wl_);                                       \
                    RAW[j][4 + e] = __builtin_bit_cast(float, wh_);                                   \
                }                                                                                     \
            }                                                                                         \
        } else {                                                                                      \
            _Pragma("unroll") for (int j = 0; j < 4; ++j)                                             \
                _Pragma("unroll") for (int e = 0; e < 8; ++e) RAW[j][e] = bload(rin, voff_[j], e * estride); \
        }                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                            \
    }
    // weights of k-step S are requested here: the split of the first tile (VALU) covers their L2 latency
#define XCOMPUTE(RAW, S)                                                                              \
    {                                                                                                 \
        u32x4 wa_[MT][3];                                                                             \
        _Pragma("unroll") for (int m = 0; m < MT; ++m)                                                \
            _Pragma("unroll") for (int pt = 0; pt < LP; ++pt)                                         \
                wa_[m][pt] = bload4(rwt, lane * 16 + (m * LP + pt) * 1024, (S) * MT * LP * 1024);     \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                               \
            if (j >= ntile) continue;          /* windows with fewer members than tiles (uniform branch) */ \
            u32x4 bs_[3];                                                                             \
            if (in16) {                                                                               \
                _Pragma("unroll") for (int e = 0; e < 4; ++e) bs_[0][e] = __builtin_bit_cast(unsigned, RAW[j][e]); \
                bs_[1] = bs_[0];                                                                      \
            } else if (F16) split8_f16(RAW[j], bs_); else split8(RAW[j], bs_);                        \
            _Pragma("unroll") for (int m = 0; m < MT; ++m) { XMF6(wa_[m], bs_, acc[m][j]) }           \
        }                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                            \
    }

    const float pinf = opaque_pinf();   // (vmax_f32)
    f32x4 best[MULTI ? MT : 1];   // running maximum over the window members
#pragma unroll
    for (int m = 0; m < (MULTI ? MT : 1); ++m) best[m] = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll 1
    for (int pass = 0; pass < npass; ++pass) {
        if (pass > 0) {
            setup_tiles(pass);
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[m][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        int s = s_begin;
        XLOADB(raw0, s)
        while (s + 2 < s_end) {
            XLOADB(raw1, s + 1)
            XCOMPUTE(raw0, s)
            XLOADB(raw0, s + 2)
            XCOMPUTE(raw1, s + 1)
            s += 2;
        }
        if (s + 1 < s_end) {
            XLOADB(raw1, s + 1)
            XCOMPUTE(raw0, s)
            XCOMPUTE(raw1, s + 1)
        } else {
            XCOMPUTE(raw0, s)
        }
        if (MULTI) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (j < ntile)
#pragma unroll
                        for (int r = 0; r < 4; ++r) best[MULTI ? m : 0][r] = vmax_f32(best[MULTI ? m : 0][r], acc[m][j][r], pinf);
        }
    }
#undef XLOADB
#undef XCOMPUTE

    if (gm.out_cl) {   // channels-last output (B, npc, out_cp) for conv_band.hip: fp32 or fp16 cells, exact zeros in the channel padding
        float amax_cl = 0.f;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int co0 = ((int)blockIdx.y * MT + m) * 16 + 4 * g;
            if (co0 >= gm.out_cp) continue;
            f32x4 bv;
#pragma unroll
            for (int r = 0; r < 4; ++r) bv[r] = (a.bias && co0 + r < gm.Cout) ? a.bias[co0 + r] : 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (nmem ? (j > 0 || n0 + pcol >= ntot) : !valid[j]) continue;
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float x;
                    if (!nmem) x = acc[m][j][r];
                    else if (MULTI) x = best[MULTI ? m : 0][r];
                    else {
                        x = acc[m][0][r];
#pragma unroll
                        for (int jj = 1; jj < 4; ++jj)
                            if (jj < nmem) x = vmax_f32(x, acc[m][jj][r], pinf);
                    }
                    x = fmaf(x, gm.x_inv_scale, bv[r]);
                    if (gm.relu) x = fmaxf(x, 0.f);
                    v[r] = co0 + r < gm.Cout ? x : 0.f;
                    amax_cl = fmaxf(amax_cl, fabsf(v[r]));
                }
                const size_t o = ((size_t)bidx[j] * npc + pos[j]) * gm.out_cp + co0;
                if (gm.out_cl == 1) *reinterpret_cast<f32x4*>(a.out + o) = v;
                else {
                    typedef float f32x2_ __attribute__((ext_vector_type(2)));
                    typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
                    const u32x2_ pk = {__builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_){v[0], v[1]}, f16x2)),
                                       __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_){v[2], v[3]}, f16x2))};
                    *reinterpret_cast<u32x2_*>(reinterpret_cast<unsigned short*>(a.out) + o) = pk;
                }
            }
        }
        range_note(a.rg, amax_cl);
        return;
    }

    int bmask[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
        bmask[j] = (iy0[j] >= 0 ? 1 : 0) | (iy0[j] + 2 * gm.dh < gm.H ? 2 : 0) | (ix0[j] >= 0 ? 4 : 0) |
                   (ix0[j] + 2 * gm.dw < gm.W ? 8 : 0);

    float amax = 0.f;   // largest magnitude stored (fp16 range guard; split-K partials are checked by the reduce kernel)
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = ((int)blockIdx.y * MT + m) * 16 + 4 * g + r;
            if (co >= gm.Cout) continue;
            const float bias = a.bias ? a.bias[co] : 0.f;
            if (nmem) {   // max over the window, then bias + ReLU (both monotone, so the order does not matter)
                if (n0 + pcol >= ntot) continue;
                float v;
                if (MULTI) {
                    v = best[MULTI ? m : 0][r];
                } else {
                    v = acc[m][0][r];
#pragma unroll
                    for (int j = 1; j < 4; ++j)
                        if (j < nmem) v = vmax_f32(v, acc[m][j][r], pinf);
                }
                v = fmaf(v, gm.x_inv_scale, bias);   // 2^-S > 0 commutes with the maximum
                if (gm.relu) v = fmaxf(v, 0.f);
                a.out[((size_t)bidx[0] * gm.Cout + co) * npc + pos[0]] = v;
                amax = fmaxf(amax, fabsf(v));
                continue;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (!valid[j]) continue;
                if (gm.ksplit > 1) {
                    a.partial[((size_t)blockIdx.z * gm.B + bidx[j]) * gm.Cout * npc + (size_t)co * npc + pos[j]] =
                        acc[m][j][r] * gm.x_inv_scale;
                    continue;
                }
                float v = fmaf(acc[m][j][r], gm.x_inv_scale, bias);
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

template <int MT>
static hipError_t launch_x_mt(const ConvGeom& g, const ConvArgs& a, hipStream_t s) {
    const bool pooled = g.pool_h * g.pool_w > 1;
    const long long ntot = pooled ? (long long)g.B * (g.Ho / g.pool_h) * (g.Wo / g.pool_w) : (long long)g.B * g.Ho * g.Wo;
    const int per_wg = pooled ? 64 : 256;
    dim3 grid((unsigned)((ntot + per_wg - 1) / per_wg), (unsigned)((g.mtiles + MT - 1) / MT), (unsigned)(g.ksplit > 1 ? g.ksplit : 1));
    const bool multi = pooled && g.pool_h * g.pool_w > 4;
#define X_LAUNCH(KX_, T_, M_)                                                                                      \
    do {                                                                                                           \
        if (g.x_f16 && g.x_terms == 1) hipLaunchKernelGGL((conv_bf16x6_kernel<MT, KX_, 1, M_, true>), grid, dim3(256), 0, s, g, a); \
        else if (g.x_f16) hipLaunchKernelGGL((conv_bf16x6_kernel<MT, KX_, 3, M_, true>), grid, dim3(256), 0, s, g, a);  \
        else hipLaunchKernelGGL((conv_bf16x6_kernel<MT, KX_, T_, M_, false>), grid, dim3(256), 0, s, g, a);        \
    } while (0)
#define X_LAUNCH_T(T_)                                         \
    do {                                                       \
        if (multi) {                                           \
            if (MT > 3) return hipErrorInvalidValue;           \
            if (g.kx_inner) X_LAUNCH(true, T_, (MT <= 3));     \
            else X_LAUNCH(false, T_, (MT <= 3));               \
        } else {                                               \
            if (g.kx_inner) X_LAUNCH(true, T_, false);         \
            else X_LAUNCH(false, T_, false);                   \
        }                                                      \
    } while (0)
    if (g.x_terms == 1) X_LAUNCH_T(1);
    else if (g.x_terms == 3) X_LAUNCH_T(3);
    else X_LAUNCH_T(6);
#undef X_LAUNCH_T
#undef X_LAUNCH
    return hipGetLastError();
}

// Cin == 1: only unpadded convs whose 8-wide column blocks never run past a row of a caller-owned tensor
// (kw % 8 == 0), or Linears (H == 1; their input is always a workspace buffer with slack behind it).
bool conv_bf16x6_supported(const ConvGeom& g) {
    if (!g.kx_inner) return true;
    return g.ph == 0 && g.pw == 0 && g.dw == 1 && (g.kw % 8 == 0 || g.kh == 1);
}

hipError_t launch_conv_bf16x6(const ConvGeom& g, const ConvArgs& a, hipStream_t s) {
    if (g.B <= 0) return hipSuccess;
    if (g.pool_h * g.pool_w > 1 && (g.pool_h * g.pool_w > 16 || g.ksplit > 1 || g.accumulate || a.border ||
                                    g.Ho < g.pool_h || g.Wo < g.pool_w))
        return hipErrorInvalidValue;
    if (g.out_cl && (g.ksplit > 1 || g.accumulate || a.border || g.out_cp % 16 || g.out_cp < g.Cout || g.out_cp > g.mtiles * 16))
        return hipErrorInvalidValue;
    if (g.in_f16 && !conv_x_in16_ok(g)) return hipErrorInvalidValue;   // the kernel would read the fp16 cells as fp32
    switch (g.x_mt) {
        case 1: return launch_x_mt<1>(g, a, s);
        case 2: return launch_x_mt<2>(g, a, s);
        case 3: return launch_x_mt<3>(g, a, s);
        case 4: return launch_x_mt<4>(g, a, s);
    }
    return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------------------------- host packing
namespace {
unsigned short bf16_rne_h(float x) {
    unsigned u;
    std::memcpy(&u, &x, 4);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
float bf16_to_f_h(unsigned short h) {
    const unsigned u = (unsigned)h << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}
}  // namespace

void conv_bf16x6_geometry(ConvGeom& g) {
    if (g.kx_inner) {
        g.x_blocks_per_row = (g.kw + 7) / 8;
        g.x_blocks = g.kh * g.x_blocks_per_row;
    } else {
        g.x_blocks_per_row = (g.Cin + 7) / 8;
        g.x_blocks = g.kh * g.kw * g.x_blocks_per_row;
    }
    g.x_ksteps = (g.x_blocks + 3) / 4;
    // Channel tiles per wave.  Per k-step a wave pays the gather + split of its four B fragments once (~800 cycles of VALU
    // issue) and 4 x 6 MFMAs (384 cycles) per channel tile; the grid repeats that for every group of MT tiles, so few
    // groups matter more than a fully used last group (Cout = 78: MT = 3 over 2 groups beats MT = 1 over 5).
    int best = 1;
    long long best_cost = -1;
    for (int mt = 1; mt <= (g.x_mt_cap > 0 ? g.x_mt_cap : 4); ++mt) {
        const long long cost = (long long)((g.mtiles + mt - 1) / mt) * (800 + 384 * mt);
        if (best_cost < 0 || cost < best_cost) {
            best_cost = cost;
            best = mt;
        }
    }
    g.x_mt = best;
}

// fp16 variant: weights times `scale` -> [mgroup][k-step][MT][part 2][lane][8 fp16], same block order
void pack_conv_weights_f16x3(const ConvGeom& g, const float* w, float scale, std::vector<unsigned short>& dst) {
    const int mgroups = (g.mtiles + g.x_mt - 1) / g.x_mt;
    dst.assign((size_t)mgroups * g.x_ksteps * g.x_mt * 2 * 64 * 8, 0);
    for (int mg = 0; mg < mgroups; ++mg)
        for (int s = 0; s < g.x_ksteps; ++s)
            for (int m = 0; m < g.x_mt; ++m)
                for (int lane = 0; lane < 64; ++lane) {
                    const int co = (mg * g.x_mt + m) * 16 + (lane & 15);
                    const int bi = 4 * s + (lane >> 4);
                    const int r = bi / g.x_blocks_per_row, cb = bi % g.x_blocks_per_row;
                    for (int e = 0; e < 8; ++e) {
                        float v = 0.f;
                        if (bi < g.x_blocks && co < g.Cout) {
                            if (g.kx_inner) {
                                const int ky = r, kx = 8 * cb + e;
                                if (kx < g.kw) v = w[((size_t)co * g.kh + ky) * g.kw + kx];
                            } else {
                                const int ky = r / g.kw, kx = r % g.kw, ci = 8 * cb + e;
                                if (ci < g.Cin) v = w[(((size_t)co * g.Cin + ci) * g.kh + ky) * g.kw + kx];
                            }
                        }
                        v *= scale;
                        const unsigned short h = f16_rne_host(v);
                        const unsigned short l = f16_rne_host(v - f16_to_f_host(h));
                        dst[(((((size_t)mg * g.x_ksteps + s) * g.x_mt + m) * 2 + 0) * 64 + lane) * 8 + e] = h;
                        dst[(((((size_t)mg * g.x_ksteps + s) * g.x_mt + m) * 2 + 1) * 64 + lane) * 8 + e] = l;
                    }
                }
}

// weights (Cout, Cin, kh, kw) -> [mgroup][k-step][MT][part][lane][8 bf16]; lane = (block slot g << 4) | row
void pack_conv_weights_bf16x6(const ConvGeom& g, const float* w, std::vector<unsigned short>& dst) {
    const int mgroups = (g.mtiles + g.x_mt - 1) / g.x_mt;
    dst.assign((size_t)mgroups * g.x_ksteps * g.x_mt * 3 * 64 * 8, 0);
    for (int mg = 0; mg < mgroups; ++mg)
        for (int s = 0; s < g.x_ksteps; ++s)
            for (int m = 0; m < g.x_mt; ++m)
                for (int lane = 0; lane < 64; ++lane) {
                    const int co = (mg * g.x_mt + m) * 16 + (lane & 15);
                    const int bi = 4 * s + (lane >> 4);
                    const int r = bi / g.x_blocks_per_row, cb = bi % g.x_blocks_per_row;
                    for (int e = 0; e < 8; ++e) {
                        float v = 0.f;
                        if (bi < g.x_blocks && co < g.Cout) {
                            if (g.kx_inner) {
                                const int ky = r, kx = 8 * cb + e;
                                if (kx < g.kw) v = w[((size_t)co * g.kh + ky) * g.kw + kx];
                            } else {
                                const int ky = r / g.kw, kx = r % g.kw, ci = 8 * cb + e;
                                if (ci < g.Cin) v = w[(((size_t)co * g.Cin + ci) * g.kh + ky) * g.kw + kx];
                            }
                        }
                        const unsigned short h = bf16_rne_h(v);
                        const float r1 = v - bf16_to_f_h(h);
                        const unsigned short mm = bf16_rne_h(r1);
                        const unsigned short l = bf16_rne_h(r1 - bf16_to_f_h(mm));
                        const unsigned short parts[3] = {h, mm, l};
                        for (int pt = 0; pt < 3; ++pt)
                            dst[(((((size_t)mg * g.x_ksteps + s) * g.x_mt + m) * 3 + pt) * 64 + lane) * 8 + e] = parts[pt];
                    }
                }
}

}  // namespace kws
