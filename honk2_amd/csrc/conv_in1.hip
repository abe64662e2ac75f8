// First convolution of the cnn-* models from an LDS image of the clip (reference model/cnn.py:46-62: Conv2d(1, C, (kh, 8), stride
// (sh, sw), no padding) + bias -> ReLU -> MaxPool(ph, pw)): one input channel, 15-32 x 8 taps, 54-336 output channels.
//
// The generic layer-wise kernel gathers the eight kernel columns of every B fragment with eight 4-byte loads and splits them into fp16
// parts in registers, once per (k-step, position tile, channel group): ~800 cycles of vector work per k-step against 384 cycles of
// MFMAs -- it is bound by that work, the matrix pipe is 8-20 % busy (profiles/r02).  Here a workgroup owns ONE clip: its 101 x 40
// feature map goes to LDS once, already split into fp16 parts, and every B fragment is one ds_read2_b64 per part.
//
//   * A lane's eight k-slots are eight consecutive samples of one feature row starting at ANY column, but an LDS read of 2-byte
//     elements needs 8-byte alignment: the image is stored four times, copy s shifted by s elements, so that the window starting
//     at column x is 8-byte aligned in copy x mod 4 (4 x 2 parts x 9 KB).  Copies start 32 bytes apart mod 256: 16 consecutive
//     windows (four per copy) fall on 16 different 8-byte bank pairs.
//   * K = kh x 8: k-step s = kernel rows 4 s .. 4 s + 3 (one per 16-lane group), KS = ceil(kh / 4) k-steps of
//     v_mfma_f32_16x16x32_f16.  A wave keeps the weight fragments of MH channel tiles for ALL k-steps in registers (KS x MH x
//     parts x 4 VGPRs), walks every fourth position tile of the clip with them, then takes the next channel group.
//   * MaxPool is reduced in the accumulators as in the generic kernel: a position tile = 16 POOLED positions, its window members
//     are computed one after the other with a running maximum; bias + ReLU after the maximum (both monotone).
//   * Output: channels-last (B, Hq, Wq, Cp) cells, fp32 or fp16 (`fp16` dtype), exact zeros in the channel padding -- what
//     conv_band.hip and the column-permuted first Linear read.
#include "kws_internal.h"

namespace kws {

namespace {
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2_ __attribute__((ext_vector_type(2)));

#define IMF(A_, B_, C_) C_ = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, A_), __builtin_bit_cast(f16x8, B_), C_, 0, 0, 0)

// h = fp16(x) pairs and l = fp16(x - h) pairs of eight fp32 values (conv3x3_tile.hip; the results go to LDS)
__device__ __forceinline__ void in1_split8(const float (&x)[8], unsigned (&h)[4], unsigned (&l)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        h[i] = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_){x[2 * i], x[2 * i + 1]}, f16x2));
        unsigned lo;
        asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(h[i]), "v"(x[2 * i]));
        asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(h[i]), "v"(x[2 * i + 1]));
        l[i] = lo;
    }
}
}  // namespace

constexpr int IN1_COPY_ALIGN = 256, IN1_COPY_SKEW = 32, IN1_TAIL = 1024;

// bytes of one shifted copy of a (T, F) map: rows of F + 4 halfs, rounded so that consecutive copies start 32 bytes apart mod 256
__host__ __device__ inline int in1_copy_bytes(int T, int F) {
    const int raw = T * (F + 4) * 2 + 16;
    return (raw + IN1_COPY_ALIGN - 1 - IN1_COPY_SKEW) / IN1_COPY_ALIGN * IN1_COPY_ALIGN + IN1_COPY_SKEW;
}

// KS: k-steps (ceil(kh / 4)); MH: channel tiles per pass; TERMS: 3 (two-part operands) or 1 (fp16 products, fp16 cells out); FC: the map's width when it is
// the 40 mel bands of every shipped config (the k-steps' LDS offsets become immediates of the fragment reads), 0: any width
template <int KS, int MH, int TERMS, int FC>
__global__ __launch_bounds__(256, 2) void conv_in1_kernel(In1ConvParams p) {
    constexpr int NP = TERMS >= 3 ? 2 : 1;
    extern __shared__ __align__(16) char lds[];
    if (range_gate_closed(p.rg)) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, pcol = lane & 15;
    const int b = blockIdx.x;
    const int RS = FC ? FC + 4 : p.F + 4;        // row stride in halfs
    const int copyb = in1_copy_bytes(p.T, p.F);
    const int partb = 4 * copyb;

    // ---------------------------------------------------------------- stage the clip: 4 shifted copies x NP parts
    // item = (row, group of four columns 4k..4k+3): the thread reads columns 4k..4k+7 and writes, for every shift s, the 8-byte group
    // {x[4k+s] .. x[4k+s+3]} to copy s at group index k of its row (copy s holds element x at half index row RS + x - s)
    {
        const float* src = p.feat + (size_t)b * p.T * p.F;
        const int gpr = p.F / 4, nitem = p.T * gpr, nel = p.T * p.F;
        unsigned amax_u = 0u;     // largest |x| as a bit pattern: orders like the magnitudes, and a NaN (which fmaxf would drop) sorts above every number
        for (int i = tid; i < nitem; i += 256) {
            const int row = i / gpr, k = i - row * gpr;
            const int e0 = row * p.F + 4 * k;
            float x[8];
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(src + e0);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(src + min(e0 + 4, nel - 4));   // (past the row's end: never read back)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                x[e] = v0[e];
                x[4 + e] = v1[e];
                amax_u = max(amax_u, __builtin_bit_cast(unsigned, v0[e]) & 0x7fffffffu);
            }
            unsigned h[4], l[4];
            in1_split8(x, h, l);
            const int off = (row * RS + 4 * k) * 2;
#pragma unroll
            for (int pt = 0; pt < NP; ++pt) {
                const unsigned(&q)[4] = pt ? l : h;
                // pairs (x0 x1)(x2 x3)(x4 x5)(x6 x7); odd shifts re-pair with a 16-bit funnel shift
                const unsigned o1 = __builtin_amdgcn_alignbit(q[1], q[0], 16), o2 = __builtin_amdgcn_alignbit(q[2], q[1], 16),
                               o3 = __builtin_amdgcn_alignbit(q[3], q[2], 16);
                char* base = lds + pt * partb + off;
                *reinterpret_cast<u32x2*>(base) = (u32x2){q[0], q[1]};
                *reinterpret_cast<u32x2*>(base + copyb) = (u32x2){o1, o2};
                *reinterpret_cast<u32x2*>(base + 2 * copyb) = (u32x2){q[1], q[2]};
                *reinterpret_cast<u32x2*>(base + 3 * copyb) = (u32x2){o2, o3};
            }
        }
        // the feature maps come from the caller: checked like any stored activation (a NaN counts as out of range), so run_cnn launches no range_check_kernel in front of this kernel
        range_note(p.rg, __builtin_bit_cast(float, amax_u));
        // Kernel rows past kh (zero weights; kh = 15, 21) of the last output rows read up to three rows past the map: the gap behind
        // every copy and the tail behind the last one must hold finite values (0 x NaN would poison the accumulator, and the
        // running maximum silently drops a NaN member)
        if (tid < IN1_TAIL / 16) *reinterpret_cast<u32x4*>(lds + NP * partb + tid * 16) = (u32x4){0u, 0u, 0u, 0u};
        const int used = p.T * RS * 2, gap8 = (copyb - used) / 8;
        for (int i = tid; i < NP * 4 * gap8; i += 256) {
            const int c = i / gap8, k = i - c * gap8;
            *reinterpret_cast<u32x2*>(lds + c * copyb + used + 8 * k) = (u32x2){0u, 0u};
        }
    }
    __syncthreads();

    const int npq = p.Hq * p.Wq;                 // pooled positions per clip
    const int ntile = (npq + 15) / 16;
    const int ngroup = (p.mtiles + MH - 1) / MH;
    const u32x4* A = reinterpret_cast<const u32x4*>(p.apk) + lane;   // [group][k-step][MH][2 parts][64]
    const int kstep_b = 4 * RS * 2;              // bytes per k-step (four feature rows)
    float amax = 0.f;
    const int step_y = 64 / p.Wq, step_x = 64 - step_y * p.Wq;      // four waves x 16 positions between a wave's tiles
    const int row_step_b = p.sh * RS * 2;
    const float pinf = opaque_pinf();
    const float relu_lo = p.relu ? 0.f : -INFINITY;      // ReLU as one maximum, no select

    for (int cg = 0; cg < ngroup; ++cg) {
        u32x4 a[KS][MH][NP];
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int m = 0; m < MH; ++m)
#pragma unroll
                for (int pt = 0; pt < NP; ++pt) a[s][m][pt] = A[(((size_t)cg * KS + s) * MH + m) * 2 * 64 + pt * 64];
        f32x4 bv[MH];   // bias of this lane's four channels per tile
#pragma unroll
        for (int m = 0; m < MH; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = (cg * MH + m) * 16 + 4 * g + r;
                bv[m][r] = co < p.Cout ? p.bias[co] : 0.f;
            }
        // this lane's pooled position of tile t: (oy, ox) walks 64 positions per tile of the wave (one division per channel group, not one per tile)
        int oy = (w * 16 + pcol) / p.Wq, ox = (w * 16 + pcol) - oy * p.Wq;
        for (int t = w; t < ntile; t += 4, oy += step_y, ox += step_x) {
            if (ox >= p.Wq) {
                ox -= p.Wq;
                ++oy;
            }
            const int ps = t * 16 + pcol;
            const bool inside = ps < npq;                       // (lanes past the map compute its last position and store nothing)
            const int oyq = inside ? oy : p.Hq - 1, oxq = inside ? ox : p.Wq - 1;
            const int rowb0 = (oyq * p.ph * p.sh + g) * (RS * 2), xb0 = oxq * p.pw * p.sw;
            f32x4 best[MH];
#pragma unroll
            for (int m = 0; m < MH; ++m) best[m] = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            for (int dy = 0; dy < p.ph; ++dy)
            for (int dx = 0; dx < p.pw; ++dx) {
                const int x0 = xb0 + dx * p.sw;
                const char* bp = lds + (x0 & 3) * copyb + (x0 & ~3) * 2 + rowb0 + dy * row_step_b;
                f32x4 acc[MH];
#pragma unroll
                for (int m = 0; m < MH; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    u32x4 bf[NP];
#pragma unroll
                    for (int pt = 0; pt < NP; ++pt) {
                        const u32x2 lo = *reinterpret_cast<const u32x2*>(bp + pt * partb + s * kstep_b);
                        const u32x2 hi = *reinterpret_cast<const u32x2*>(bp + pt * partb + s * kstep_b + 8);
                        bf[pt] = (u32x4){lo[0], lo[1], hi[0], hi[1]};
                    }
#pragma unroll
                    for (int m = 0; m < MH; ++m) {
                        if (TERMS >= 3) {
                            IMF(a[s][m][1], bf[0], acc[m]);
                            IMF(a[s][m][0], bf[1], acc[m]);
                        }
                        IMF(a[s][m][0], bf[0], acc[m]);
                        if (TERMS >= 3) __builtin_amdgcn_sched_barrier(0);   // the chain stays whole (res8_f16x3.hip, R8H_FENCE)
                    }
                }
#pragma unroll
                for (int m = 0; m < MH; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) best[m][r] = vmax_f32(best[m][r], acc[m][r], pinf);
            }
            if (ps >= npq) continue;
            const size_t cell = ((size_t)b * npq + ps) * p.Cp;
#pragma unroll
            for (int m = 0; m < MH; ++m) {
                const int co0 = (cg * MH + m) * 16 + 4 * g;
                if (co0 >= p.Cp) continue;
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float x = fmaf(best[m][r], p.inv_scale, bv[m][r]);   // 2^-S > 0 commutes with the maximum
                    v[r] = fmaxf(x, relu_lo);      // (channels past Cout: zero weights and a zero bias give the exact zero the padding must hold)
                }
                // largest magnitude, two values per v_max3_f32 (|x| is an input modifier; values straight out of an FMA / maximum: nothing to canonicalise.  A NaN
                // can only come from a NaN or an infinity in the features, and the staging loop above has flagged that)
                amax = __builtin_fmaxf(__builtin_fmaxf(amax, __builtin_fabsf(v[0])), __builtin_fabsf(v[1]));
                amax = __builtin_fmaxf(__builtin_fmaxf(amax, __builtin_fabsf(v[2])), __builtin_fabsf(v[3]));
                if (p.out_f16) {
                    const u32x2 pk = {__builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_){v[0], v[1]}, f16x2)),
                                      __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_){v[2], v[3]}, f16x2))};
                    *reinterpret_cast<u32x2*>(reinterpret_cast<unsigned short*>(p.out) + cell + co0) = pk;
                } else {
                    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + cell + co0) = v;
                }
            }
        }
    }
    range_note(p.rg, amax);
}

size_t conv_in1_lds_bytes(int T, int F, int parts) { return (size_t)parts * 4 * in1_copy_bytes(T, F) + IN1_TAIL; }

// the layer fits this kernel: Cin == 1, eight kernel columns, no padding / dilation, pooling window <= 16 members, image in LDS
bool conv_in1_supported(const ConvGeom& g, int ph, int pw) {
    const int ks = (g.kh + 3) / 4;
    return g.Cin == 1 && g.kw == 8 && g.ph == 0 && g.pw == 0 && g.dh == 1 && g.dw == 1 && g.W % 4 == 0 && g.W >= 8 &&
           (ks == 4 || ks == 5 || ks == 6 || ks == 8) && ph >= 1 && pw >= 1 && ph * pw <= 16 && g.Ho >= ph && g.Wo >= pw &&
           conv_in1_lds_bytes(g.H, g.W, 2) <= 80 * 1024 - 256 &&
           3 * (g.W + 4) * 2 + 16 <= IN1_TAIL;   // the zero-weight kernel rows of the last output rows read this far past the last copy: keep it inside the zeroed tail
}

// weights (Cout, 1, kh, 8) x scale -> two fp16 parts, [group of MH tiles][k-step][MH][part][lane][8]; lane = (kernel row 4 s + g) << 4 | co
void pack_conv_in1_weights(int Cout, int kh, int mh, const float* w, float scale, std::vector<unsigned short>& dst) {
    const int ks = (kh + 3) / 4, mtiles = (Cout + 15) / 16, ngroup = (mtiles + mh - 1) / mh;
    dst.assign((size_t)ngroup * ks * mh * 2 * 64 * 8, 0);
    for (int cg = 0; cg < ngroup; ++cg)
        for (int s = 0; s < ks; ++s)
            for (int m = 0; m < mh; ++m)
                for (int lane = 0; lane < 64; ++lane) {
                    const int co = (cg * mh + m) * 16 + (lane & 15), ky = 4 * s + (lane >> 4);
                    for (int e = 0; e < 8; ++e) {
                        float v = 0.f;
                        if (co < Cout && ky < kh) v = w[((size_t)co * kh + ky) * 8 + e] * scale;
                        const unsigned short h = f16_rne_host(v);
                        const unsigned short l = f16_rne_host(v - f16_to_f_host(h));
                        const size_t base = ((((size_t)cg * ks + s) * mh + m) * 2) * 64 * 8;
                        dst[base + (size_t)lane * 8 + e] = h;
                        dst[base + 64 * 8 + (size_t)lane * 8 + e] = l;
                    }
                }
}

template <int KS, int FC>
static hipError_t launch_in1_ks(const In1ConvParams& p, hipStream_t s) {
    const size_t lds = conv_in1_lds_bytes(p.T, p.F, p.terms >= 3 ? 2 : 1);
    auto k3 = conv_in1_kernel<KS, IN1_MH3, 3, FC>;
    auto k1 = conv_in1_kernel<KS, IN1_MH1, 1, FC>;
    static DeviceOnce attr_once;   // per instantiation pair: allow > 64 KB of dynamic LDS
    if (attr_once.first()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k3), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k1), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
    }
    if (p.terms >= 3) hipLaunchKernelGGL(k3, dim3((unsigned)p.B), dim3(256), lds, s, p);
    else hipLaunchKernelGGL(k1, dim3((unsigned)p.B), dim3(256), lds, s, p);
    return hipGetLastError();
}

hipError_t launch_conv_in1(const In1ConvParams& p, hipStream_t s) {
    if (p.B <= 0) return hipSuccess;
    if ((p.terms != 3 && p.terms != 1) || p.F % 4 || p.Cp % 16 || p.Cp < p.Cout || p.Hq < 1 || p.Wq < 1) return hipErrorInvalidValue;
    switch ((p.kh + 3) / 4) {
        case 4: return p.F == 40 ? launch_in1_ks<4, 40>(p, s) : launch_in1_ks<4, 0>(p, s);
        case 5: return p.F == 40 ? launch_in1_ks<5, 40>(p, s) : launch_in1_ks<5, 0>(p, s);
        case 6: return p.F == 40 ? launch_in1_ks<6, 40>(p, s) : launch_in1_ks<6, 0>(p, s);
        case 8: return p.F == 40 ? launch_in1_ks<8, 40>(p, s) : launch_in1_ks<8, 0>(p, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace kws
