// LDS-staged convolution for the second conv of the cnn-* models (reference model/cnn.py:46-62: Conv2d(C0, C1, (kh, kw), stride 1,
// no padding) + bias -> ReLU, pool_1 is the identity in every shipped config): 64-94 input channels, 5-10 x 4 taps, a 35-82 x 8-16 map.
//
// The generic layer-wise kernel gathers every B fragment from global memory: each input value is fetched once per tap (40 times for
// cnn-trad-pool2), 4 bytes per load instruction, and with 172 MB of input per 1 024 clips those re-reads miss L2 -- 3 GB of fabric
// traffic per launch at 4.5 TB/s with the matrix pipe 15 % busy (profiles/r02/final_cnn_fp16_summary.json).  Here a workgroup owns a
// BAND of R output rows of one clip: the R + kh - 1 input rows it needs are copied to LDS once (channels-last cells, split into two
// fp16 parts on the way in, or copied as they are when the tensor already holds fp16 -- `fp16` dtype), and every tap is served from
// there as ready-made B fragments: one ds_read_b128 per part, position tile and k-step, no VALU.
//
//   * in: channels-last (B, H, W, Cpi), Cpi = input channels padded to 16 (the producer -- the generic kernel's channels-last
//     epilogue, layerwise_bf16x6.hip -- writes exact zeros there); out: channels-last fp32 (B, Ho, Wo, Cpo); the Linear that
//     follows reads it with its weight columns permuted to (position, channel) order on the host.
//   * LDS cell = Cpi fp16 + 16 B of padding: the stride in dwords is 4 x odd, which spreads the 16 positions of a tile over all 64
//     banks for ds_read_b128 (tools/lds_bank_model.py).  Two parts = two planes.
//   * K order (tap, 8-channel block), four blocks per v_mfma_f32_16x16x32_f16 (one per 16-lane group); a k-step's (tap, block) of
//     each lane group is a byte offset from a table in LDS.  Weights: fp16 parts scaled by 2^S in fragment order from L2 (shared by
//     every workgroup), one k-step ahead.
//   * Waves 2 x 2: wave (wm, wn) owns half of the channel tiles (MH = 2 or 3) and half of the band's position tiles (<= 4): per
//     k-step MH x NP weight fragments (vector memory) + NT x NP activation fragments (LDS) feed MH x NT x terms MFMAs -- the split
//     that keeps both operand paths below the matrix pipe's time for three-term products.
#include "kws_internal.h"

namespace kws {

namespace {
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// two fp16 parts of four fp32 values (conv3x3_tile.hip: one packed convert + one mixed-precision FMA per value; LDS stores follow)
__device__ __forceinline__ void band_split4(f32x4 x, u32x2 (&out)[2]) {
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_){x[2 * i], x[2 * i + 1]}, f16x2));
        unsigned l;
        asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(h), "v"(x[2 * i]));
        asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(h), "v"(x[2 * i + 1]));
        out[0][i] = h;
        out[1][i] = l;
    }
}
#define BMF(A_, B_, C_) C_ = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, A_), __builtin_bit_cast(f16x8, B_), C_, 0, 0, 0)
}  // namespace

constexpr int BAND_NT = 4;   // position tiles per wave (half a band)

// MH: channel tiles per wave (the layer has up to 2 MH); TERMS: 3 (two-part operands, fp32-accurate) or 1 (fp16 tensor in, one part)
template <int MH, int TERMS>
__global__ __launch_bounds__(256, 2) void conv_band_kernel(BandConvParams p) {
    constexpr int NP = TERMS >= 3 ? 2 : 1;
    constexpr bool S16 = TERMS == 1;
    extern __shared__ __align__(16) char lds[];
    if (range_gate_closed(p.rg)) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w & 1, wn = w >> 1;
    const int g = lane >> 4, pcol = lane & 15;
    const int band = (int)blockIdx.x % p.nbands, b = (int)blockIdx.x / p.nbands;
    const int r0 = band * p.R;                       // first output row = first input row of the band
    const int rows_out = min(p.R, p.Ho - r0);
    const int rows_in = p.R + p.kh - 1;
    const int cellb = p.Cpi * 2 + 16;                // LDS bytes per cell and part
    const int ncell = rows_in * p.W;
    const int plane = ncell * cellb;
    const int ktab_off = NP * plane;                 // int[(ksteps + 2) * 4]
    const int nbi = p.Cpi / 8;

    // ---------------------------------------------------------------- k-step table + staging
    for (int i = tid; i < (p.ksteps + 2) * 4; i += 256) {
        const int tap = i / nbi, cb = i - tap * nbi;
        const int ky = tap / p.kw, kx = tap - ky * p.kw;
        reinterpret_cast<int*>(lds + ktab_off)[i] = tap < p.kh * p.kw ? (ky * p.W + kx) * cellb + cb * 16 : 0;   // (padding blocks carry zero weights)
    }
    {
        const int nq = S16 ? p.Cpi / 8 : p.Cpi / 4;              // 16-byte chunks per global cell
        const int gcell = S16 ? p.Cpi * 2 : p.Cpi * 4;
        const char* src = reinterpret_cast<const char*>(p.in) + ((size_t)b * p.H + r0) * p.W * gcell;
        const int nchunk = ncell * nq;
        const int last = ((p.H - r0) * p.W) * nq - 1;            // rows past the input feed only outputs that are never stored: clamp
        constexpr int UNR = 6;
        for (int c0 = tid; c0 < nchunk; c0 += UNR * 256) {
            f32x4 v[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) v[u] = *reinterpret_cast<const f32x4*>(src + (size_t)min(c0 + u * 256, last) * 16);
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int c = c0 + u * 256;
                if (c < nchunk) {
                    const int cell = c / nq, q = c - cell * nq;
                    if (S16) {
                        *reinterpret_cast<f32x4*>(lds + cell * cellb + q * 16) = v[u];
                    } else {
                        u32x2 pr[2];
                        band_split4(v[u], pr);
                        *reinterpret_cast<u32x2*>(lds + cell * cellb + q * 8) = pr[0];
                        *reinterpret_cast<u32x2*>(lds + plane + cell * cellb + q * 8) = pr[1];
                    }
                }
            }
        }
    }

    // this lane's output positions: tile j of this wave = positions (wn * BAND_NT + j) * 16 + pcol of the band, row-major over (R, Wo)
    const int npos = rows_out * p.Wo;
    int lbase[BAND_NT], opos[BAND_NT];
    const int ntile = min(BAND_NT, max(0, (npos + 15) / 16 - wn * BAND_NT));   // tiles of this wave that hold any valid position (uniform)
#pragma unroll
    for (int j = 0; j < BAND_NT; ++j) {
        const int ps = (wn * BAND_NT + j) * 16 + pcol;
        const int pc = min(ps, npos - 1);
        const int oy = pc / p.Wo, ox = pc - oy * p.Wo;
        lbase[j] = (oy * p.W + ox) * cellb;
        opos[j] = ps < npos ? (r0 + oy) * p.Wo + ox : -1;
    }
    __syncthreads();

    // ---------------------------------------------------------------- k-loop
    f32x4 acc[MH][BAND_NT];
#pragma unroll
    for (int m = 0; m < MH; ++m)
#pragma unroll
        for (int j = 0; j < BAND_NT; ++j) acc[m][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const u32x4* A = reinterpret_cast<const u32x4*>(p.apk) + (size_t)(wm * MH) * 2 * 64 + lane;   // [k-step][2 MH tiles][2 parts][64]
    const int* ktab = reinterpret_cast<const int*>(lds + ktab_off) + g;
    constexpr int ASTEP = 2 * MH * 2 * 64;   // u32x4 per k-step

#define BLOADA(AR, S)                                                                                      \
    {                                                                                                      \
        _Pragma("unroll") for (int m = 0; m < MH; ++m)                                                     \
            _Pragma("unroll") for (int pt = 0; pt < NP; ++pt) AR[m][pt] = A[(size_t)(S) * ASTEP + (m * 2 + pt) * 64]; \
    }
#define BLOADB(BR, J, KOFF)                                                                                \
    {                                                                                                      \
        const int ad_ = lbase[J] + (KOFF);                                                                 \
        _Pragma("unroll") for (int pt = 0; pt < NP; ++pt) BR[pt] = *reinterpret_cast<const u32x4*>(lds + pt * plane + ad_); \
    }
#define BTERMS(AR, BR, C_)                    \
    {                                         \
        if (TERMS >= 3) {                     \
            BMF(AR[1], BR[0], C_);            \
            BMF(AR[0], BR[1], C_);            \
        }                                     \
        BMF(AR[0], BR[0], C_);                \
    }
    // one k-step: B fragments one position tile ahead; BX holds tile 0 on entry, and tile 0 of the next step (offset KNEXT) on exit
    // (BAND_NT is even: the two buffers keep their roles from step to step)
#define BSTEP(AR, KCUR, KNEXT)                                                                             \
    {                                                                                                      \
        _Pragma("unroll") for (int j = 0; j < BAND_NT; ++j) {                                              \
            u32x4 (&cur_)[NP] = (j & 1) ? bb1 : bb0;                                                       \
            u32x4 (&nxt_)[NP] = (j & 1) ? bb0 : bb1;                                                       \
            if (j + 1 < BAND_NT) BLOADB(nxt_, j + 1, KCUR) else BLOADB(nxt_, 0, KNEXT)                     \
            __builtin_amdgcn_sched_barrier(0);                                                             \
            if (j < ntile) {                                                                               \
                _Pragma("unroll") for (int m = 0; m < MH; ++m) BTERMS(AR[m], cur_, acc[m][j])              \
            }                                                                                              \
            __builtin_amdgcn_sched_barrier(0);                                                             \
        }                                                                                                  \
    }
    static_assert(BAND_NT % 2 == 0, "fragment buffers keep their roles");
    u32x4 a0[MH][NP], a1[MH][NP], bb0[NP], bb1[NP];
    if (ntile > 0) {
        int kc = ktab[0];
        BLOADA(a0, 0)
        BLOADB(bb0, 0, kc)
        for (int s = 0; s < p.ksteps; s += 2) {
            int kn = ktab[4 * (s + 1)];
            if (s + 1 < p.ksteps) BLOADA(a1, s + 1)
            __builtin_amdgcn_sched_barrier(0);
            BSTEP(a0, kc, kn)
            if (s + 1 >= p.ksteps) break;
            kc = kn;
            kn = ktab[4 * (s + 2)];
            if (s + 2 < p.ksteps) BLOADA(a0, s + 2)
            __builtin_amdgcn_sched_barrier(0);
            BSTEP(a1, kc, kn)
            kc = kn;
        }
    }
#undef BLOADA
#undef BLOADB
#undef BTERMS
#undef BSTEP

    // ---------------------------------------------------------------- epilogue: bias, ReLU, channels-last fp32 stores
    float amax = 0.f;
    float* const outb = p.out + (size_t)b * p.Ho * p.Wo * p.Cpo;
#pragma unroll
    for (int m = 0; m < MH; ++m) {
        const int co0 = (wm * MH + m) * 16 + 4 * g;
        if (co0 >= p.Cpo) continue;
        f32x4 bv;
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[r] = co0 + r < p.Cout ? p.bias[co0 + r] : 0.f;
#pragma unroll
        for (int j = 0; j < BAND_NT; ++j) {
            if (j >= ntile || opos[j] < 0) continue;
            f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float x = fmaf(acc[m][j][r], p.inv_scale, bv[r]);
                if (p.relu) x = fmaxf(x, 0.f);
                v[r] = co0 + r < p.Cout ? x : 0.f;
                amax = fmaxf(amax, fabsf(v[r]));
            }
            *reinterpret_cast<f32x4*>(outb + (size_t)opos[j] * p.Cpo + co0) = v;
        }
    }
    range_note(p.rg, amax);
}

size_t conv_band_lds_bytes(int Cpi, int W, int kh, int kw, int R, int parts) {
    const int nbi = Cpi / 8, ksteps = (kh * kw * nbi + 3) / 4;
    return (size_t)parts * (R + kh - 1) * W * (Cpi * 2 + 16) + (size_t)(ksteps + 2) * 16;
}

// Rows per band: the largest R whose band fits half a CU's LDS with two-part cells and 2 x BAND_NT position tiles, weighted by how
// full its tiles are.  0: the layer does not fit this kernel.
int conv_band_rows(int Cin, int Cout, int H, int W, int kh, int kw) {
    const int Cpi = (Cin + 15) / 16 * 16, Ho = H - kh + 1, Wo = W - kw + 1;
    if (Cin < 16 || Cout > 96 || Ho < 1 || Wo < 1 || Wo > 16 * 2 * BAND_NT) return 0;
    int best = 0;
    double best_eff = 0.0;
    for (int R = 1; R <= Ho && R * Wo <= 16 * 2 * BAND_NT; ++R) {
        if (conv_band_lds_bytes(Cpi, W, kh, kw, R, 2) > 80 * 1024 - 256) break;
        // useful MFMA share: positions over the tile slots of the slower wave pair, rows over the rows the bands cover, and the
        // halo rows each band re-reads count against small R through the staging cost
        const int nb = (Ho + R - 1) / R;
        const int tiles = (R * Wo + 15) / 16, slots = 2 * ((tiles + 1) / 2);
        const double eff = (double)(Ho * Wo) / ((double)nb * slots * 16) * ((double)R / (R + 0.25 * (kh - 1)));
        if (eff > best_eff) {
            best_eff = eff;
            best = R;
        }
    }
    return best;
}

// weights (Cout, Cin, kh, kw) x scale -> two fp16 parts, [k-step][2 MH channel tiles][part][lane][8]; block 4 s + (lane >> 4) = (tap, 8-channel block)
void pack_conv_band_weights(int Cin, int Cout, int kh, int kw, const float* w, float scale, std::vector<unsigned short>& dst) {
    const int Cpi = (Cin + 15) / 16 * 16, nbi = Cpi / 8, ksteps = (kh * kw * nbi + 3) / 4;
    const int mh = conv_band_mh(Cout), mtt = 2 * mh;
    dst.assign((size_t)ksteps * mtt * 2 * 64 * 8, 0);
    for (int s = 0; s < ksteps; ++s)
        for (int m = 0; m < mtt; ++m)
            for (int lane = 0; lane < 64; ++lane) {
                const int co = m * 16 + (lane & 15), bi = 4 * s + (lane >> 4);
                const int tap = bi / nbi, cb = bi % nbi, ky = tap / kw, kx = tap % kw;
                for (int e = 0; e < 8; ++e) {
                    const int ci = 8 * cb + e;
                    float v = 0.f;
                    if (tap < kh * kw && co < Cout && ci < Cin) v = w[(((size_t)co * Cin + ci) * kh + ky) * kw + kx] * scale;
                    const unsigned short h = f16_rne_host(v);
                    const unsigned short l = f16_rne_host(v - f16_to_f_host(h));
                    dst[((((size_t)s * mtt + m) * 2 + 0) * 64 + lane) * 8 + e] = h;
                    dst[((((size_t)s * mtt + m) * 2 + 1) * 64 + lane) * 8 + e] = l;
                }
            }
}

template <int MH, int TERMS>
static hipError_t launch_band_k(const BandConvParams& p, hipStream_t s) {
    const size_t lds = conv_band_lds_bytes(p.Cpi, p.W, p.kh, p.kw, p.R, TERMS >= 3 ? 2 : 1);
    auto k = conv_band_kernel<MH, TERMS>;
    static DeviceOnce attr_once;   // per instantiation: allow > 64 KB of dynamic LDS
    if (attr_once.first()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k, dim3((unsigned)(p.B * p.nbands)), dim3(256), lds, s, p);
    return hipGetLastError();
}

hipError_t launch_conv_band(const BandConvParams& p, hipStream_t s) {
    if (p.B <= 0) return hipSuccess;
    const int mh = conv_band_mh(p.Cout);
    if (p.R < 1 || p.Cpi % 16 || p.Cpo % 16 || p.Cpo < p.Cout || p.Cpo > 32 * mh || p.R * p.Wo > 16 * 2 * BAND_NT ||
        (p.terms != 3 && p.terms != 1) || conv_band_lds_bytes(p.Cpi, p.W, p.kh, p.kw, p.R, 2) > 160 * 1024 - 512)
        return hipErrorInvalidValue;
    if (mh == 2) return p.terms == 3 ? launch_band_k<2, 3>(p, s) : launch_band_k<2, 1>(p, s);
    return p.terms == 3 ? launch_band_k<3, 3>(p, s) : launch_band_k<3, 1>(p, s);
}

}  // namespace kws
