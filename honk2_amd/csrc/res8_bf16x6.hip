// res8 forward, fully fused, with fp32-accurate products formed on the bf16 matrix cores ("bf16x6").
//
// Same function as res8_fused.hip (reference model/resnet.py:38-60 for config/resnet/res8.json) and the same
// dataflow -- activations never leave the CU, each wave owns 5 position tiles x 3 channel tiles + one tile of
// position-tile 20, a layer's output and the residual stay in registers -- but conv_1..conv_6 use
// v_mfma_f32_16x16x32_bf16 instead of v_mfma_f32_16x16x4_f32:
//   * gfx950's fp32-input MFMA runs at the fp32 VALU rate (157 TFLOP/s) and blocks the VALU while it does
//     (tools/coexec_probe.cpp); the bf16 MFMA is 16x faster per MAC and co-executes with VALU work
//     (tools/coexec_probe_bf16.cpp: 2.0 PFLOP/s sustained).  Every fp32 operand x is split into three bf16 parts
//     x = x1 + x2 + x3 (24 mantissa bits in total; bf16 has fp32's exponent range, so no scaling or overflow
//     concerns) and a product a*b is accumulated as the six terms a3b1 + a2b2 + a1b3 + a2b1 + a1b2 + a1b1; each
//     bf16 x bf16 product is exact in the fp32 accumulator and the dropped terms are <= 2^-24 |ab|, so the result is
//     as accurate as an fp32 FMA chain (CPU emulation: rms error 1.1e-7 vs 2.6e-7 for a plain fp32 GEMM of this shape).
//     Cost: 6 MFMAs of 16 cycles for 16x16x32 MACs = 3 cycles per 16x16x4 block instead of 32.
//   * LDS holds the activation map as [384 cells][3 parts][48 channels] bf16 (288 B per cell, 110 KB, zero halo
//     cells included), so one B fragment (8 consecutive input channels of one tap for 16 positions) is one
//     ds_read_b128 per part, with no VALU work in the loop; the fp32 -> 3 x bf16 split happens once per output
//     element in the epilogue (where ReLU / residual / BatchNorm are still done in fp32).
//   * K = 9 taps x 6 blocks of 8 channels = 54 blocks -> 14 k-steps (2 padding blocks with zero weights).  Weights
//     are split and packed on the host in fragment order [layer][k-step][channel tile][part][lane] (16 B per lane).
//   * one workgroup of 4 waves per CU (110 KB of LDS), one wave per SIMD with the whole 512-entry register file:
//     A and B fragments of the next k-step are requested before the 96 MFMAs (1 536 cycles) of the current one.
//   conv_0 (K = 9) keeps the fp32-input MFMA path of res8_fused.hip.
#include "kws_internal.h"

#include <cstring>

namespace kws {

namespace {
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int CELL_B = 288;                           // bytes per map cell: 3 parts x 48 channels x 2
constexpr int PART_B = 96;
constexpr int MAP_BYTES = 384 * CELL_B;               // 110 592
constexpr int FS = 41;                                // staged feature row stride (fp32 words)
constexpr int FEAT_BYTES = ((102 * FS * 4 + 15) / 16) * 16;
constexpr int RED_OFF = MAP_BYTES + FEAT_BYTES;       // fp32 words from here on
constexpr int BNT_WORDS = R8_LAYERS * 96;
constexpr int X_LDS_BYTES = RED_OFF + (4 * 48 + 48 + BNT_WORDS) * 4;
constexpr int KSTEPS = R8X_KSTEPS;                    // 14
constexpr int A_STEP = 3 * 3 * 64;                    // u32x4 per k-step: [channel tile][part][lane]

__device__ __forceinline__ float relu1(float x) {
    const int b = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, b > 0 ? b : 0);
}
__device__ __forceinline__ f32x4 relu4(f32x4 v) { return (f32x4){relu1(v[0]), relu1(v[1]), relu1(v[2]), relu1(v[3])}; }
__device__ __forceinline__ unsigned pack2(float a, float b) {
    const bf16x2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float lo_f(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float hi_f(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }

// split 4 consecutive channels into three bf16 parts and store them at byte address `addr` (+ part * 96)
__device__ __forceinline__ void store_split(char* lds, int addr, f32x4 v) {
    u32x2 h, m, l;
    h[0] = pack2(v[0], v[1]);
    h[1] = pack2(v[2], v[3]);
    const float r0 = v[0] - lo_f(h[0]), r1 = v[1] - hi_f(h[0]), r2 = v[2] - lo_f(h[1]), r3 = v[3] - hi_f(h[1]);
    m[0] = pack2(r0, r1);
    m[1] = pack2(r2, r3);
    l[0] = pack2(r0 - lo_f(m[0]), r1 - hi_f(m[0]));
    l[1] = pack2(r2 - lo_f(m[1]), r3 - hi_f(m[1]));
    *reinterpret_cast<u32x2*>(lds + addr) = h;
    *reinterpret_cast<u32x2*>(lds + addr + PART_B) = m;
    *reinterpret_cast<u32x2*>(lds + addr + 2 * PART_B) = l;
}

#define MF(A_, B_, C_) C_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A_), __builtin_bit_cast(bf16x8, B_), C_, 0, 0, 0)
// six-term product, small terms first
#define MF6(A3, B_, C_)        \
    if (TERMS == 6) {          \
        MF(A3[2], B_[0], C_);  \
        MF(A3[1], B_[1], C_);  \
        MF(A3[0], B_[2], C_);  \
    }                          \
    if (TERMS >= 3) {          \
        MF(A3[1], B_[0], C_);  \
        MF(A3[0], B_[1], C_);  \
    }                          \
    MF(A3[0], B_[0], C_);

struct XCtx {
    char* lds;
    float* red;
    float* mvec;
    const float* bnt;
    int tid, lane, w, g, pcol, mx;
    int qb[6];   // byte address of this lane's cell (part 0, channel 0) in each of the wave's 6 position tiles
    bool xvalid;
};

struct AFrags {
    u32x4 a[3][3];   // [channel tile][part]
    u32x4 ax[3];     // extra tile's channel tile
};
struct BFrag {
    u32x4 p[3];      // the three bf16 parts of one position tile's B fragment
};

// byte offset (within the map) of this lane's k-slot at k-step s: block bi = 4 s + g -> tap = bi / 6, channel block bi % 6
__device__ __forceinline__ int step_boff(int s, int g) {
    int bi = 4 * s + g;
    bi = bi < 54 ? bi : 53;   // blocks 54, 55 are zero-weight padding: re-read a valid block
    const int tap = bi / 6, cblk = bi - 6 * tap, ty = tap / 3, tx = tap - 3 * ty;
    return ((ty - 1) * R8_RS + (tx - 1)) * CELL_B + cblk * 16;
}
__device__ __forceinline__ void load_a(AFrags& f, const u32x4* A, int s, int mx) {
    const u32x4* As = A + (size_t)s * A_STEP;
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int pt = 0; pt < 3; ++pt) f.a[m][pt] = As[(m * 3 + pt) * 64];
#pragma unroll
    for (int pt = 0; pt < 3; ++pt) f.ax[pt] = As[(mx * 3 + pt) * 64];
}
__device__ __forceinline__ void load_b(BFrag& b, const char* lds, int addr) {
#pragma unroll
    for (int pt = 0; pt < 3; ++pt) b.p[pt] = *reinterpret_cast<const u32x4*>(lds + addr + pt * PART_B);
}

// One k-step: 16 tiles x 6 terms = 96 MFMAs.  B fragments are fetched ONE position tile ahead (three ds_read_b128 in
// flight -- the LDS counter is 4 bits, a whole k-step's 18 reads cannot be outstanding); `bnext_addr0` is tile 0 of
// the NEXT k-step.  A fragments of the next k-step were requested by the caller before this step's MFMAs.
#define X_STEP(AF, BOFF, BOFF_NEXT)                                                   \
    {                                                                                 \
        _Pragma("unroll") for (int j = 0; j < 6; ++j) {                               \
            BFrag& bcur = (j & 1) ? bb1 : bb0;                                        \
            BFrag& bnxt = (j & 1) ? bb0 : bb1;                                        \
            load_b(bnxt, c.lds, j < 5 ? c.qb[j + 1] + (BOFF) : c.qb[0] + (BOFF_NEXT)); \
            __builtin_amdgcn_sched_barrier(0);                                        \
            if (j < 5) {                                                              \
                _Pragma("unroll") for (int m = 0; m < 3; ++m) { MF6(AF.a[m], bcur.p, acc[j][m]) } \
            } else {                                                                  \
                MF6(AF.ax, bcur.p, accx)                                              \
            }                                                                         \
            __builtin_amdgcn_sched_barrier(0);                                        \
        }                                                                             \
    }

template <int TERMS, bool EVEN, bool LAST>
__device__ __forceinline__ void x_layer(const Res8xParams& p, const XCtx& c, int layer, int clip, f32x4 (&prev)[5][3],
                                        f32x4& prevx) {
    const int g = c.g, mx = c.mx;
    f32x4 acc[5][3], accx;
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int m = 0; m < 3; ++m) acc[j][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    accx = (f32x4){0.f, 0.f, 0.f, 0.f};

    const u32x4* A = reinterpret_cast<const u32x4*>(p.apk6) + (size_t)layer * KSTEPS * A_STEP + c.lane;
    AFrags fa0, fa1;
    BFrag bb0, bb1;   // ping-pong over position tiles; 6 tiles per step keeps the parity aligned across steps
    load_a(fa0, A, 0, mx);
    load_b(bb0, c.lds, c.qb[0] + step_boff(0, g));
#define X_PAIR(S)                                                                                                  \
    {                                                                                                              \
        const int o0 = step_boff((S), g), o1 = step_boff((S) + 1, g),                                              \
                  o2 = step_boff((S) + 2 < KSTEPS ? (S) + 2 : KSTEPS - 1, g);                                      \
        load_a(fa1, A, (S) + 1, mx);                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                         \
        X_STEP(fa0, o0, o1)                                                                                        \
        load_a(fa0, A, (S) + 2 < KSTEPS ? (S) + 2 : KSTEPS - 1, mx); /* last one is a harmless re-read */          \
        __builtin_amdgcn_sched_barrier(0);                                                                         \
        X_STEP(fa1, o1, o2)                                                                                        \
    }
    if (!(KWS_DBG(p.debug & 2))) {
        if (TERMS == 1) {   // not MFMA-bound: left to the unroller this variant only spills
#pragma unroll 1
            for (int s = 0; s < KSTEPS; s += 2) X_PAIR(s)
        } else {
            for (int s = 0; s < KSTEPS; s += 2) X_PAIR(s)
        }
    }
#undef X_PAIR

    // ---- epilogue in fp32: ReLU, residual (reference: even i), BatchNorm as one FMA
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
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
            for (int m = 0; m < 3; ++m) store_split(c.lds, c.qb[j] + (16 * m + 4 * g) * 2, acc[j][m]);
        if (c.xvalid) store_split(c.lds, c.qb[5] + (16 * mx + 4 * g) * 2, accx);
        __syncthreads();
    } else {
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

size_t res8x_lds_bytes() { return (size_t)X_LDS_BYTES; }

template <int TERMS>
__global__ __launch_bounds__(256, 1) void res8x_kernel(Res8xParams p) {
    extern __shared__ __attribute__((aligned(16))) char ldsb[];
    XCtx c;
    c.lds = ldsb;
    float* feat_s = reinterpret_cast<float*>(ldsb + MAP_BYTES);
    c.red = reinterpret_cast<float*>(ldsb + RED_OFF);
    c.mvec = c.red + 4 * 48;
    float* bnt = c.mvec + 48;
    c.bnt = bnt;

    const int tid = threadIdx.x;
    c.tid = tid;
    c.lane = tid & 63;
    c.w = __builtin_amdgcn_readfirstlane(tid >> 6);
    c.g = c.lane >> 4;
    c.pcol = c.lane & 15;
    c.mx = c.w < 2 ? c.w : 2;
    const int w = c.w, g = c.g, pcol = c.pcol, mx = c.mx, lane = c.lane;

    for (int i = tid; i < MAP_BYTES / 4; i += 256) reinterpret_cast<unsigned*>(ldsb)[i] = 0u;   // zero halo, for good
    for (int i = tid; i < BNT_WORDS; i += 256) bnt[i] = p.bn_tab[i];

    int lb[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int nt = j < 5 ? 5 * w + j : 20;
        const int n = 16 * nt + pcol;
        const int nn = n < R8_NPOS ? n : R8_NPOS - 1;
        const int y = nn / W8_W;
        const int x = nn - y * W8_W;
        c.qb[j] = ((y + 1) * R8_RS + x + 1) * CELL_B;
        lb[j] = 4 * y * FS + 3 * x;
    }
    c.xvalid = w < 3 && (16 * 20 + pcol) < R8_NPOS;
    int koff[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const int k = 4 * s + g;
        koff[s] = k < 9 ? (k / 3) * FS + (k - 3 * (k / 3)) : 0;
    }

    // The (101, 40) feature map of a clip is staged as fp32 with a zero top row / left column.  Only conv_0 reads it, so the
    // NEXT clip's map is requested (into registers) while conv_0 of the current clip runs and is written to LDS right after
    // it: a clip never waits for its own features.
    auto feat_load = [&](int clip, f32x4 (&v)[4]) {
        const f32x4* f4 = reinterpret_cast<const f32x4*>(p.feat + (size_t)clip * p.T * p.F);
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int q4 = it * 256 + tid;
            v[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (q4 < 1010) v[it] = f4[q4];
        }
    };
    auto feat_store = [&](const f32x4 (&v)[4]) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int q4 = it * 256 + tid;
            if (q4 < 1010) {
                const int idx = 4 * q4;
                const int cell = idx + idx / 40 + FS + 1;
                feat_s[cell] = v[it][0];
                feat_s[cell + 1] = v[it][1];
                feat_s[cell + 2] = v[it][2];
                feat_s[cell + 3] = v[it][3];
            }
        }
    };
    if ((int)blockIdx.x < p.B) {
        f32x4 v0[4];
        feat_load(blockIdx.x, v0);
        feat_store(v0);
        if (tid < FS) feat_s[tid] = 0.f;
        if (tid < 101) feat_s[(tid + 1) * FS] = 0.f;
    }

    for (int clip = blockIdx.x; clip < p.B; clip += gridDim.x) {
        __syncthreads();  // previous clip's tail has consumed red/mvec and the maps; this clip's features are in place
        const int clip_next = clip + (int)gridDim.x;

        // ---- conv_0 + ReLU + AvgPool(4,3): fp32-input MFMA (K = 9), result in accumulator layout = prev_x
        f32x4 prev[5][3], prevx;
        f32x4 vnext[4];
        {
            float a0[3][3];
#pragma unroll
            for (int m = 0; m < 3; ++m)
#pragma unroll
                for (int s = 0; s < 3; ++s) a0[m][s] = p.w0a[(m * 3 + s) * 64 + lane];
            const float ax0 = p.w0a[(mx * 3 + 0) * 64 + lane], ax1 = p.w0a[(mx * 3 + 1) * 64 + lane],
                        ax2 = p.w0a[(mx * 3 + 2) * 64 + lane];
            // after the (L2-resident) conv_0 weights: vector loads retire in order
            if (clip_next < p.B) feat_load(clip_next, vnext);
            const f32x4 zero = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int ad0 = lb[j] + koff[0], ad1 = lb[j] + koff[1], ad2 = lb[j] + koff[2];
                f32x4 s0 = zero, s1 = zero, s2 = zero;
                if (!(KWS_DBG(p.debug & 1))) {
#pragma unroll
                    for (int wp = 0; wp < 6; ++wp) {
                        f32x4 cc[2][3];
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const int wi = 2 * wp + h, oy = wi / 3, ox = wi - 3 * oy;
                            const float b0 = feat_s[ad0 + oy * FS + ox], b1 = feat_s[ad1 + oy * FS + ox],
                                        b2 = feat_s[ad2 + oy * FS + ox];
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
                            f32x4 cx = __builtin_amdgcn_mfma_f32_16x16x4f32(ax0, feat_s[ad0 + oy * FS + ox], zero, 0, 0, 0);
                            cx = __builtin_amdgcn_mfma_f32_16x16x4f32(ax1, feat_s[ad1 + oy * FS + ox], cx, 0, 0, 0);
                            cx = __builtin_amdgcn_mfma_f32_16x16x4f32(ax2, feat_s[ad2 + oy * FS + ox], cx, 0, 0, 0);
                            sx += relu4(cx);
                        }
                }
                prevx = sx * (1.0f / 12.0f);
            }
        }
        // the maps are idle here (the previous clip finished behind the barrier at the top of the loop)
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
            for (int m = 0; m < 3; ++m) store_split(c.lds, c.qb[j] + (16 * m + 4 * g) * 2, prev[j][m]);
        if (c.xvalid) store_split(c.lds, c.qb[5] + (16 * mx + 4 * g) * 2, prevx);
        __syncthreads();
        if (clip_next < p.B) feat_store(vnext);   // every wave is past conv_0: the staging area is free

        x_layer<TERMS, false, false>(p, c, 0, clip, prev, prevx);
        x_layer<TERMS, true, false>(p, c, 1, clip, prev, prevx);
        x_layer<TERMS, false, false>(p, c, 2, clip, prev, prevx);
        x_layer<TERMS, true, false>(p, c, 3, clip, prev, prevx);
        x_layer<TERMS, false, false>(p, c, 4, clip, prev, prevx);
        x_layer<TERMS, true, true>(p, c, 5, clip, prev, prevx);
    }
}

hipError_t launch_res8x(const Res8xParams& p, int grid, hipStream_t s) {
    static DeviceOnce attr_once;
    if (attr_once.first()) {
        hipError_t e = hipFuncSetAttribute((const void*)res8x_kernel<6>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)res8x_lds_bytes());
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute((const void*)res8x_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)res8x_lds_bytes());
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute((const void*)res8x_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)res8x_lds_bytes());
        if (e != hipSuccess) return e;
    }
    if (p.B <= 0) return hipSuccess;
    if (p.terms == 1)
        hipLaunchKernelGGL(res8x_kernel<1>, dim3((unsigned)grid), dim3(256), res8x_lds_bytes(), s, p);
    else if (p.terms == 3)
        hipLaunchKernelGGL(res8x_kernel<3>, dim3((unsigned)grid), dim3(256), res8x_lds_bytes(), s, p);
    else
        hipLaunchKernelGGL(res8x_kernel<6>, dim3((unsigned)grid), dim3(256), res8x_lds_bytes(), s, p);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------- host packing
namespace {
unsigned short bf16_rne(float x) {
    unsigned u;
    std::memcpy(&u, &x, 4);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
float bf16_to_f(unsigned short h) {
    const unsigned u = (unsigned)h << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}
}  // namespace

// conv_i weight (45,45,3,3) -> [k-step 14][channel tile 3][part 3][lane 64][8 bf16]:
//   cout = 16 m + (lane & 15); block bi = 4 s + (lane >> 4): tap = bi / 6, input channels 8 (bi % 6) .. +7; bi >= 54: zeros
void pack_res8x_layer(const float* wt, unsigned short* dst) {
    for (int s = 0; s < KSTEPS; ++s)
        for (int m = 0; m < 3; ++m)
            for (int lane = 0; lane < 64; ++lane) {
                const int co = 16 * m + (lane & 15), bi = 4 * s + (lane >> 4);
                for (int j = 0; j < 8; ++j) {
                    float v = 0.f;
                    if (bi < 54 && co < R8_C) {
                        const int tap = bi / 6, ci = 8 * (bi % 6) + j;
                        if (ci < R8_C) v = wt[((size_t)co * R8_C + ci) * 9 + tap];
                    }
                    const unsigned short h = bf16_rne(v);
                    const float r1 = v - bf16_to_f(h);
                    const unsigned short mpart = bf16_rne(r1);
                    const unsigned short l = bf16_rne(r1 - bf16_to_f(mpart));
                    const unsigned short parts[3] = {h, mpart, l};
                    for (int pt = 0; pt < 3; ++pt)
                        dst[((((size_t)s * 3 + m) * 3 + pt) * 64 + lane) * 8 + j] = parts[pt];
                }
            }
}

}  // namespace kws
