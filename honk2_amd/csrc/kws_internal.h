// Internal declarations shared by the C-ABI translation unit and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <atomic>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/kws.h"

// Experiment switches -- the KWS_R8_DEBUG / KWS_T3_DEBUG ablation bits (results wrong by construction), the *_TIMING phase stamps (a host
// synchronisation and a device-to-host copy inside a compute call), the grid / chunk sizing knobs and the fault-injection hook -- exist only
// in a `make EXPERIMENTS=1` build (tools/ uses it).  The default build reads none of those environment variables: `experiment_int` /
// `experiment_str` return their defaults and KWS_DBG(x) is the constant 0, so the branches behind it are compiled out of every kernel.
#ifdef KWS_EXPERIMENTS
#define KWS_DBG(x) ((x) != 0)
#else
#define KWS_DBG(x) false
#endif

namespace kws {

inline int experiment_int(const char* name, int dflt) {
#ifdef KWS_EXPERIMENTS
    const char* v = std::getenv(name);
    return v ? std::atoi(v) : dflt;
#else
    (void)name;
    return dflt;
#endif
}
inline const char* experiment_str(const char* name) {
#ifdef KWS_EXPERIMENTS
    return std::getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

// ---------------------------------------------------------------- host-side fp16 helpers (weight packing of the fp16 paths)
// fp32 -> fp16 bits, round to nearest even; subnormals and overflow handled (weights are finite)
inline unsigned short f16_rne_host(float x) {
    unsigned u;
    std::memcpy(&u, &x, 4);
    const unsigned sign = (u >> 16) & 0x8000u;
    const int e = (int)((u >> 23) & 0xffu) - 127 + 15;
    unsigned m = u & 0x7fffffu;
    if (((u >> 23) & 0xffu) == 0) return (unsigned short)sign;            // fp32 zero / subnormal -> 0
    if (e >= 31) return (unsigned short)(sign | 0x7c00u);                  // overflow -> inf
    if (e <= 0) {                                                          // fp16 subnormal (or underflow to 0)
        if (e < -10) return (unsigned short)sign;
        m |= 0x800000u;
        const int shift = 14 - e;
        const unsigned q = m >> shift, rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
        return (unsigned short)(sign | (q + ((rem > half || (rem == half && (q & 1u))) ? 1u : 0u)));
    }
    const unsigned q = m >> 13, rem = m & 0x1fffu;
    unsigned r = ((unsigned)e << 10) | q;
    if (rem > 0x1000u || (rem == 0x1000u && (q & 1u))) ++r;               // a carry into the exponent is the right result
    return (unsigned short)(sign | r);
}
inline float f16_to_f_host(unsigned short h) {
    const unsigned sign = (unsigned)(h & 0x8000u) << 16;
    const int e = (h >> 10) & 0x1f;
    const unsigned m = h & 0x3ffu;
    float f;
    if (e == 0) {
        f = std::ldexp((float)m, -24);
        return sign ? -f : f;
    }
    const unsigned u = sign | ((unsigned)(e - 15 + 127) << 23) | (m << 13);
    std::memcpy(&f, &u, 4);
    return f;
}
// power-of-two scale that brings the largest |weight| of a layer into [128, 256): the second fp16 part of every weight
// then stays in (or near) the normal range
inline float weight_scale_pow2(const float* w, size_t n) {
    float mx = 0.f;
    for (size_t i = 0; i < n; ++i) mx = std::fmax(mx, std::fabs(w[i]));
    if (!(mx > 0.f) || !std::isfinite(mx)) return 1.f;
    int ex;
    std::frexp(mx, &ex);                 // mx = f * 2^ex, f in [0.5, 1)
    return std::ldexp(1.f, 8 - ex);
}

// `static DeviceOnce once;  if (once.first()) { hipFuncSetAttribute(...) }`: true the first time it is asked on the
// current HIP device (function attributes such as the dynamic-LDS limit are per device; a process may drive several)
struct DeviceOnce {
    std::atomic<unsigned long long> done{0};
    bool first() {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return true;
        const unsigned long long bit = 1ull << dev;
        return (done.fetch_or(bit) & bit) == 0;
    }
};

// Kernels that address LDS through absolute 32-bit integers (conv3x3_tile.hip, conv_band.hip) assume their dynamic LDS starts at LDS
// address 0, i.e. that the kernel holds NO static __shared__.  Checked on the host, once per instantiation and device, where the dynamic
// LDS limit is raised: a static array added to such a translation unit fails the first launch with an error code instead of aborting on
// the GPU (the kernels keep a __builtin_trap() behind the same condition as a last line of defence).
inline hipError_t allow_big_lds_at_base_zero(const void* kernel, int bytes = 160 * 1024) {
    hipFuncAttributes fa{};
    hipError_t e = hipFuncGetAttributes(&fa, kernel);
    if (e != hipSuccess) return e;
    if (fa.sharedSizeBytes != 0) return hipErrorInvalidValue;
    return hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------- fp16 range guard of the layer-wise plans
// The fp32-accurate default splits activations into two fp16 parts, which cannot hold |x| > 65 504 (the reference is fp32
// throughout, model/resnet.py:38-60, model/cnn.py:79-107).  Every kernel that STORES an activation another kernel will
// split notes the largest magnitude it wrote: at or above KWS_RANGE_LIMIT it sets one device word (an
// infinity is caught as such, before it can turn into a NaN).  After the
// chunk's fp16 pass the same plan is launched again on three-part bf16 operands (fp32's exponent range, six terms) with
// `gated` = 1: those kernels return at once unless the word is set.  No host round trip; trained models never trip it.
struct RangeGate {
    unsigned* flag;   // device word (zeroed before the chunk's first pass), or nullptr: no guard
    int gated;        // 0: note what is stored; 1: run only if *flag != 0
};
constexpr float KWS_RANGE_LIMIT = 32768.f;
#if defined(__HIPCC__)
// max(a, b) as ONE instruction: v_med3_f32(a, b, +inf) with the +inf in a register the compiler cannot see through (opaque_pinf(): call it once per
// kernel, outside the loops).  fmaxf() itself compiles to three instructions -- hipcc canonicalises both operands first (v_max x, x, x) in case one is a
// signalling NaN -- 48 instead of 16 per window member in the MaxPool reductions; fmed3 with a CONSTANT +inf is folded back into that fmax.  A NaN operand
// loses, as with fmaxf.  (Not inline asm: the hazard recogniser pads nothing for an asm that reads an MFMA result.)
__device__ __forceinline__ float opaque_pinf() {
    float v = __builtin_inff();
    asm volatile("" : "+v"(v));
    return v;
}
__device__ __forceinline__ float vmax_f32(float a, float b, float pinf) { return __builtin_amdgcn_fmed3f(a, b, pinf); }
__device__ __forceinline__ bool range_gate_closed(const RangeGate& rg) {
    return rg.gated && rg.flag && __builtin_nontemporal_load(rg.flag) == 0u;
}
// Work queue of a persistent kernel: q[0] = tickets drawn so far, q[1] = workgroups that have retired, both zero between
// launches.  A workgroup starts on unit blockIdx.x, then draws tickets (unit = gridDim.x + atomicAdd(q, 1)) until it draws one
// past the end; ONE thread of it then calls queue_retire.  Every workgroup's last draw has returned before it retires, so the
// workgroup that retires last knows no draw is outstanding and puts both words back to zero: the launch leaves the queue as
// it found it, and needs no memset in front of it (a memset is a separate graph node; this is why there is none).
__device__ __forceinline__ void queue_retire(unsigned* q) {
    if (atomicAdd(q + 1, 1u) == gridDim.x - 1) {
        atomicExch(q, 0u);
        atomicExch(q + 1, 0u);
    }
}
__device__ __forceinline__ void range_note(const RangeGate& rg, float amax) {
    if (!rg.gated && rg.flag && !(amax < KWS_RANGE_LIMIT)) *rg.flag = 1u;   // (a NaN flags too) same value from every writer: a benign race
}
#endif

// ---------------------------------------------------------------- front end (frontend.hip)
constexpr int FE_NFFT = 480;
constexpr int FE_HOP = 160;
constexpr int FE_ROWS = 128;        // DFT bins 0..127 (four GEMMs of 64 rows: {Re, Im} x {even, odd} bins)
constexpr int FE_NT = 7;            // 16-frame column tiles per workgroup chunk
constexpr int FE_FRAMES = FE_NT * 16;
constexpr int FE_STEPS = 30;        // k-steps of 4 columns: the twice-folded DFT has K = 120
constexpr int FE_GROUPS = 8;        // k-steps packed 4 per float4 (last group half used)
constexpr int FE_PSTRIDE = 116;     // row stride (words) of the power tile in LDS
constexpr size_t FE_TABLE_FLOATS = (size_t)4 * FE_GROUPS * 4 * 64 * 4;

struct FrontendParams {
    const float* wav;     // (B, n_samples) fp32, or nullptr when pcm is given
    const short* pcm;     // (B, n_samples) int16 PCM, or nullptr
    const float* noise;   // (B, n_samples) additive noise clips, or nullptr
    float noise_pct;
    float* feat;          // (B, T, n_mels)
    const f32x4* dft;     // packed cos/sin table, FE_TABLE_FLOATS floats
    const float* hann;    // (FE_STEPS*4, 2): h[j], h[240-j]
    const float* melw;    // (n_mels, FE_ROWS) dense mel weights over bins 0..127
    const int* mel_lo;    // first / one-past-last non-zero bin per mel band
    const int* mel_hi;
    int B, n_samples, T, n_mels, chunks;
    int mel_maxw;         // widest mel filter in bins (fast path keeps <= 16 weights in registers)
    long long clip_stride;   // samples between the starts of consecutive clips (n_samples for a packed batch)
    const void* dft16;    // fp16 two-part cos/sin fragments (frontend_f16x3.hip)
    const float* consts16;   // FE16_CONST_WORDS: h[j] (128), h[240-j] (128)
    // mel stage of frontend_f16_kernel = a banded GEMM on the fp32-input matrix cores: band tile m (16 bands) covers the
    // k-steps mel_fb[m] .. mel_fb[m] + mel_ns[m] - 1 of four DFT bins each; mel_a holds their A fragments, step after step
    // over m = 0, 1, 2: [step][lane] = W[16 m + (lane & 15)][4 (fb + s) + (lane >> 4)]
    const float* mel_a;
    int mel_fb[3], mel_ns[3];
    unsigned* queue;         // two device words, zero between launches: frontend_f16_kernel's unit queue (queue_retire above)
};
constexpr int FE16_MAX_MELS = 48, FE16_MAX_STEPS = 64;   // what frontend_f16_kernel's mel stage holds: three band tiles, 64 k-steps
constexpr int FE16_CONST_WORDS = 256;
// host side: the banded-GEMM table of a filterbank (n_mels, FE_ROWS) with per-band bin ranges [lo, hi); false if it does not fit
bool build_mel_gemm_table(const std::vector<float>& wts, const std::vector<int>& lo, const std::vector<int>& hi, int n_mels,
                          std::vector<float>& tab, int (&fb)[3], int (&ns)[3]);
// streaming windows whose shift is a multiple of the hop: edge frames + copy of the shared rows (frontend.hip)
struct WindowEdgeParams {
    const float* stream;        // the long waveform; window i starts at sample i * shift
    const float* global_feat;   // (1 + n_stream / hop, n_mels): features of the stream taken as one clip
    float* feat;                // (n_windows, T, n_mels)
    const float* hann;          // (480) periodic Hann
    const f32x2* trig;          // (480) cos, sin of 2 pi j / 480
    const float* melw;
    const int* mel_lo;
    const int* mel_hi;
    int window, shift, n_windows, T, n_mels;
};
hipError_t launch_window_edges(const WindowEdgeParams& p, hipStream_t s);
void build_edge_tables(std::vector<float>& hann, std::vector<float>& trig);
size_t frontend_lds_bytes(int T);
hipError_t launch_frontend(const FrontendParams& p, hipStream_t s);
void build_dft_table(std::vector<float>& dft, std::vector<float>& hann);  // host side, double precision trig
hipError_t launch_frontend_f16(const FrontendParams& p, int n_cu, hipStream_t s);
int frontend_f16_frames();      // frames per unit (chunk) of frontend_f16_kernel
void build_dft_table_f16(std::vector<unsigned>& tab, std::vector<float>& hann2);

// ---------------------------------------------------------------- fused res8 (res8_fused.hip)
constexpr int R8_C = 45, R8_H = 25, W8_W = 13, R8_NPOS = 325;
constexpr int R8_RS = 14;           // LDS row stride: 13 columns + one shared zero halo column
constexpr int R8_CS = 400;          // LDS channel stride (27*14+1 = 379 used, 400 = 16 mod 32 banks)
constexpr int R8_LAYERS = 6;
constexpr int R8_GROUPS = 28;       // 9 taps x 3 float4 groups (11 steps + 1 pad) + 1 leftover group
constexpr size_t R8_APK_FLOATS = (size_t)R8_LAYERS * R8_GROUPS * 3 * 64 * 4;

struct Res8Params {
    const float* feat;    // (B, 101, 40)
    float* logits;        // (B, n_labels)
    const float* w0a;     // conv_0 weight as A fragments [3 channel tiles][3 k-steps][64 lanes]
    const f32x4* apk;     // packed conv_1..6 weights
    const float* bn_tab;  // (6, 96): per layer scale[48] = rstd, shift[48] = -mean*rstd (channels padded with 0)
    const int* zcells;    // 1024 LDS cells to re-zero after the feature staging (build_res8_zero_cells)
    const float* out_w;   // (n_labels, 45)
    const float* out_b;   // (n_labels)
    int B, T, F, n_labels;
    int debug;            // KWS_R8_DEBUG bits, timing experiments only (results are wrong when set): 1 skip conv_0,
                          // 2 skip the MFMA loop, 4 one workgroup per CU
};
size_t res8_lds_bytes();
hipError_t launch_res8(const Res8Params& p, int grid, hipStream_t s);
void pack_res8_layer(const float* w /*45x45x3x3*/, float* dst /*R8_GROUPS*3*64*4*/);
void pack_res8_conv0(const float* w /*45x9*/, float* dst /*3*3*64*/);
void build_res8_zero_cells(int* dst /*1024*/);

// ---------------------------------------------------------------- fused res8, bf16x6 matrix path (res8_bf16x6.hip)
constexpr int R8X_KSTEPS = 14;      // 9 taps x 6 blocks of 8 input channels = 54 blocks, 4 per k-step of v_mfma_f32_16x16x32_bf16
constexpr size_t R8X_APK_SHORTS = (size_t)R8_LAYERS * R8X_KSTEPS * 3 * 3 * 64 * 8;

struct Res8xParams {
    const float* feat;    // (B, 101, 40)
    float* logits;        // (B, n_labels)
    const float* w0a;     // conv_0 weight as fp32 A fragments (pack_res8_conv0)
    const void* apk6;     // conv_1..6 weights split into three bf16 parts, fragment order (pack_res8x_layer)
    const float* bn_tab;  // (6, 96): per layer scale[48], shift[48]
    const float* out_w;   // (n_labels, 45)
    const float* out_b;   // (n_labels)
    int B, T, F, n_labels;
    int debug;            // timing experiments only: 1 skip conv_0, 2 skip the MFMA loop
    int terms;            // 6: fp32-accurate products; 3: KWS_DTYPE_BF16X3
};
size_t res8x_lds_bytes();
hipError_t launch_res8x(const Res8xParams& p, int grid, hipStream_t s);
void pack_res8x_layer(const float* w /*45x45x3x3*/, unsigned short* dst /*R8X_KSTEPS*3*3*64*8*/);

// ---------------------------------------------------------------- fused res8, fp16 three-term matrix path (res8_f16x3.hip)
constexpr int R8H_ASTEPS = R8X_KSTEPS + 1;   // the 14 k-steps + the merged fragments of the last one (res8_f16x3.hip)
constexpr size_t R8H_APK_SHORTS = (size_t)R8_LAYERS * R8H_ASTEPS * 3 * 2 * 64 * 8;

struct Res8hParams {
    const float* feat;    // (B, 101, 40)
    float* logits;        // (B, n_labels)
    const void* w0h;      // conv_0 weight * 2^S0 as two fp16 parts, one 16x16x32 A fragment per channel tile (pack_res8h_conv0)
    float inv_scale0;     // 2^-S0
    const void* apk2;     // conv_1..6 weights (previous BatchNorm folded in) * 2^S split into two fp16 parts, fragment order (pack_res8h_layer)
    const float* bn_tab;  // (96): the LAST BatchNorm's scale[48], shift[48] (applied to the 45 channel means in the tail)
    const float* out_w;   // (n_labels, 45)
    const float* out_b;   // (n_labels)
    float inv_scale[R8_LAYERS];   // 2^-S per layer
    float kappa[R8_LAYERS];       // value of the constant channel (slot 45) in the map layer l writes: the scale of that map
    int B, T, F, n_labels;
    int debug;            // timing experiments only: 1 skip conv_0, 2 skip the MFMA loop
    int terms;            // 3: fp32-accurate products; 1: plain fp16 operands (KWS_DTYPE_F16)
    unsigned* queue;      // two device words, zero between launches: the clip queue (queue_retire)
    const int* feat_shift;   // per clip: power of two its features are staged down by (launch_feat_shift), or nullptr = 0
};
size_t res8h_lds_bytes();
// shift[b] = 0 while max |feat[b]| <= 2^14, else ceil(log2(max)) - 14: the fused kernel stages features as fp16 pairs
hipError_t launch_feat_shift(const float* feat, int B, int n_per_clip, int* shift, hipStream_t s);
hipError_t launch_res8h(const Res8hParams& p, int grid, hipStream_t s);
void pack_res8h_layer(const float* w /*45 x 46 x 3x3: input channel 45 = the folded BatchNorm shift*/, float scale, unsigned short* dst /*R8H_ASTEPS*3*2*64*8*/);
void pack_res8h_conv0(const float* w /*45x9*/, float scale, unsigned short* dst /*(3*2 + 3)*64*8*/);

// ---------------------------------------------------------------- layer-wise kernels (layerwise.hip)
struct ConvGeom {
    int B;                 // clips in this launch
    int Cin, H, W;         // input map
    int Cout, Ho, Wo;      // output map
    int kh, kw, sh, sw, ph, pw, dh, dw;
    int kx_inner;          // 1: K = (ky, kx padded to 4), needs Cin == 1;  0: K = (ky, kx, cin padded to 4)
    int inner_steps;       // ceil(kw/4) or ceil(Cin/4)
    int ksteps;            // kh*inner_steps (kx_inner) or kh*kw*inner_steps
    int mtiles;            // ceil(Cout/16)
    int MT;                // m-tiles per wave (1..4); grid.y = ceil(mtiles/MT)
    int relu, accumulate;  // epilogue: ReLU; out += value (residual kept in the output buffer)
    int x_blocks_per_row;  // bf16x6 kernel: blocks of 8 per tap (ceil(Cin/8)) or per kernel row (ceil(kw/8))
    int x_blocks;          // total K blocks of 8
    int x_ksteps;          // ceil(x_blocks / 4): k-steps of v_mfma_f32_16x16x32_bf16
    int x_mt;              // bf16 kernel: channel tiles per wave (conv_bf16x6_geometry)
    int x_mt_cap;          // > 0: upper bound for x_mt (3 for convs whose pooling window needs several passes)
    int x_terms;           // bf16 parts: 6 (fp32-accurate), 3 (KWS_DTYPE_BF16X3) or 1 (KWS_DTYPE_BF16) terms per product
    int x_f16;             // 1: two-part fp16 operands, three terms (the fp32-accurate default); weights carry 2^S
    float x_inv_scale;     // 2^-S (1 when not x_f16), applied to the accumulator in the epilogue
    int pool_h, pool_w;    // conv_bf16x6_kernel only: fused MaxPool window (stride = window, floor), <= 16 members; 0 / 1 = none
    int ksplit;            // > 1: K range split over blockIdx.z (flat Cin == 1 mode only); partial sums go to ConvArgs::partial
    int ksteps_split;      // k-steps per split
    int out_cl;            // conv_bf16x6_kernel only: 1 = write channels-last fp32 (B, npc, out_cp), 2 = the same as fp16 (for conv_band.hip);
    int out_cp;            // channels of a cell (Cout padded to 16, zeros in the padding); no split-K / accumulate / border then
    int in_f16;            // conv_bf16x6_kernel, Linears over 16-byte aligned rows with single-term fp16 products only: the input holds fp16 values (conv_band.hip's
                           // out_f16 cells) -- the operand the kernel would have rounded an fp32 input to, loaded as it is
};

// conv_bf16x6_kernel reads its input as fp16 cells (ConvGeom::in_f16) only in this geometry -- a Linear over 16-byte aligned rows with
// single-term fp16 products -- and otherwise as fp32: ONE predicate for the kernel, the launcher (which refuses in_f16 where it does not
// hold: the kernel would silently read fp16 cells as fp32) and the host that sets the flag.
__host__ __device__ inline bool conv_x_vec8(const ConvGeom& g) {
    return g.kx_inner && g.kh == 1 && g.dw == 1 && g.H == 1 && g.ph == 0 && g.pw == 0 && (g.W & 3) == 0 && g.sw == 1 && g.Wo == 1;
}
__host__ __device__ inline bool conv_x_in16_ok(const ConvGeom& g) {
    return g.x_f16 && g.x_terms == 1 && conv_x_vec8(g) && (g.W & 7) == 0;
}

struct ConvArgs {
    const float* in;       // (B, Cin, H, W)
    float* out;            // (B, Cout, Ho, Wo)
    const float* apk;      // packed weights [mgroup][kstep][MT][64] (previous layer's BN scale already folded in)
    const unsigned short* apk16;   // the same weights split into three bf16 parts for conv_bf16x6_kernel, or nullptr
    const float* bias;     // (Cout) or nullptr
    float* partial;        // (ksplit, B, Cout, Ho*Wo) raw partial sums when ksplit > 1 (then reduced by launch_splitk_reduce)
    const float* border;   // (16, Cout) or nullptr: previous layer's BN shift summed over the in-bounds taps, by border class
    RangeGate rg;          // fp16 range guard (zero-initialised: none)
};
hipError_t launch_conv(const ConvGeom& g, const ConvArgs& a, hipStream_t s);
void pack_conv_weights(const ConvGeom& g, const float* w /*Cout,Cin,kh,kw*/, std::vector<float>& dst);
int choose_mt(int mtiles);
// bf16x6 variant (layerwise_bf16x6.hip): same contract; supported for Cin > 1 and for unpadded Cin == 1 convs / Linears
bool conv_bf16x6_supported(const ConvGeom& g);
void conv_bf16x6_geometry(ConvGeom& g);
hipError_t launch_conv_bf16x6(const ConvGeom& g, const ConvArgs& a, hipStream_t s);
void pack_conv_weights_bf16x6(const ConvGeom& g, const float* w, std::vector<unsigned short>& dst);
void pack_conv_weights_f16x3(const ConvGeom& g, const float* w, float scale, std::vector<unsigned short>& dst);
// LDS-tiled 3x3 "same" conv with power-of-two dilation over channels-last fp32 tensors in sub-map layouts
// (conv3x3_tile.hip)
constexpr int T3_TILE_P = 192;         // three bf16 parts per LDS cell (KWS_MATRIX_PARTS=bf16): 3 position tiles per wave
constexpr int T3_TILE_P_F16 = 320;     // two parts or one (fp16 default, bf16x3, the 16-bit dtypes), 41-48 channels: 5 position tiles per wave
constexpr int T3_TILE_P_F16_NARROW = 192;   // the same, 17-24 channels: 3 tiles per wave, three workgroups per CU (+14 % measured)
// parts of an operand that the LDS cell holds = parts that take part in the products (terms: 6 / 3 / 1 products per fp32 product)
constexpr int t3_lds_parts(bool f16, int terms) { return f16 ? (terms >= 3 ? 2 : 1) : (terms == 6 ? 3 : (terms == 3 ? 2 : 1)); }
#ifndef T3_S16_TILE
#define T3_S16_TILE T3_TILE_P_F16      // A/B knob: positions per workgroup with 16-bit tensors (one part), 41-48 channels
#endif
#ifndef T3_S16_WGS
#define T3_S16_WGS 3                   // A/B knob: workgroups per CU the single-part kernel is compiled for
#endif
constexpr int t3_tile_positions(int parts, int nb) {
    return parts >= 3 ? T3_TILE_P : (nb <= 3 ? T3_TILE_P_F16_NARROW : (parts == 1 ? T3_S16_TILE : T3_TILE_P_F16));
}
struct TileConvParams {
    // Activations are fp32, except with single-term products (terms == 1: the `bf16` / `fp16` dtypes), where the three tensors hold
    // 16-bit values of the operand type (bf16, or fp16 when f16 is set): staging is then a plain copy and HBM traffic halves.
    const void* in;        // CL tensor in layout(2^ld_in): [clip][y mod d][x mod d][ceil(H/d)][ceil(W/d)][cp]
    void* out;             // CL tensor, written in layout(2^ld_out)
    const void* res;       // residual CL tensor in layout(2^ld_res), or nullptr
    const unsigned short* apk16;   // f16: pack_conv3x3_tile_weights_f16; else pack_conv_weights_bf16x6 with MT = all tiles
    const float* border;   // (16, C padded to 8) border-bias table, zeros in the padding, or nullptr
    int B, H, W, Cout;
    int ld_in, ld_out, ld_res;
    int Hs, Ws;            // ceil(H / d_in), ceil(W / d_in)
    int total;             // B * d_in^2 * Hs * Ws cells of the input layout
    int terms;             // bf16 parts only: 6 / 3 / 1 terms per product
    int f16;               // 1: two-part fp16 operands, three terms (fp32-accurate default)
    float inv_scale;       // 2^-S of the fp16 weights (1 for bf16)
    RangeGate rg;          // fp16 range guard (zero-initialised: none)
    // Per-cell metadata of ONE clip in the input layout (build_tile_conv_table): for cell q, {tap mask | border class << 9 | valid << 13,
    // cell of the same position in the output layout, in the residual layout, 0}; cells per clip of the three layouts.  What a lane
    // used to decode per tile (two divisions, the sub-lattice arithmetic, nine tap tests, two layout computations: ~180 vector
    // instructions per position) is one 16-byte load.
    const int* postab;
    int cpc_in, cpc_out, cpc_res;
    unsigned long long* dbg_ts;   // KWS_T3_TIMING (with a -DT3_TIMING build): 8 stamps per workgroup for the first 8192 workgroups, or nullptr
    int debug;             // KWS_T3_DEBUG, timing experiments only (results are wrong when set): 1 skip the k-loop, 2 skip the
                           // staging loads, 4 skip the output stores, 8 skip the residual read, 16 skip the k-loop's weight-fragment loads
};
bool conv3x3_tile_supported(int C, int Cout, int Ws);
// host side: the metadata table of one clip for a layer reading layout(2^ld_in), writing layout(2^ld_out), residual in layout(2^ld_res)
void build_tile_conv_table(int H, int W, int ld_in, int ld_out, int ld_res, std::vector<int>& tab, int& cpc_in, int& cpc_out, int& cpc_res);
void pack_conv3x3_tile_weights_f16(int C, const float* w /*C,C,3,3*/, float scale, std::vector<unsigned short>& dst);
hipError_t launch_conv3x3_tile(const TileConvParams& p, int C, hipStream_t s);
// Two consecutive layers of equal dilation (odd i, i + 1) in one kernel, 16-bit tensors only (conv3x3_tile.hip)
struct PairConvParams {
    const void* in;        // x_{i-1}, CL 16-bit tensor in layout(2^ld): conv_i's input AND conv_{i+1}'s residual
    void* out;             // x_{i+1}, written in layout(2^ld_out)
    const unsigned short* apk_a;   // conv_i / conv_{i+1} weights in the fragment order of the single-layer kernel
    const unsigned short* apk_b;
    const float* border_a; // (16, C padded to 8) border-bias tables, or nullptr
    const float* border_b;
    int B, H, W, Cout;
    int ld, ld_out;
    int Hs, Ws, total;     // sub-map size and cells of the input layout (as TileConvParams)
    int f16;               // operand / tensor type: fp16 (1) or bf16 (0)
    float inv_scale_a, inv_scale_b;   // 2^-S of the fp16 weights (1 for bf16)
    RangeGate rg;
    const int* postab;     // conv_{i+1}'s per-cell table (build_tile_conv_table): both layers share the input layout
    int cpc_in, cpc_out;
    int debug;             // KWS_T3_DEBUG, timing experiments only (results are wrong when set): 1 skip conv_i's k-loop, 64 skip conv_{i+1}'s, 2 skip the staging loads, 4 skip the output stores
};
int conv3x3_pair_tile(int C, int Ws);   // output positions per workgroup, 0 = geometry not supported
hipError_t launch_conv3x3_pair(const PairConvParams& p, int C, hipStream_t s);
// THREE consecutive layers of equal dilation (a, a + 1, a + 2) in one kernel, 16-bit tensors only (conv3x3_tile.hip): res15's runs
// (4,5,6) (7,8,9) (10,11,12) and every run of hey_snips.  Two shapes, by the parity of a (reference model/resnet.py:46-55):
//   a even:  x_a = relu(conv_a(in)) + res  ->  y = relu(conv_{a+1}(x_a))  ->  out = relu(conv_{a+2}(y)) + x_a
//            (in = y_{a-1}; res = x_{a-2} comes from memory in the layout it was written in; x_a never leaves the CU)
//   a odd:   y = relu(conv_a(in))  ->  x_{a+1} = relu(conv_{a+1}(y)) + in  ->  out = relu(conv_{a+2}(x_{a+1}))
//            (in = x_{a-1}; x_{a+1} is ALSO stored, to out2 in the input's own layout: it is the next even layer's residual)
struct TripleConvParams {
    const void* in;        // CL 16-bit tensor in layout(2^ld)
    const void* res;       // a even: x_{a-2} in layout(2^ld_res) (cells through the table); a odd: nullptr
    void* out;             // the third layer's output, written in layout(2^ld_out)
    void* out2;            // a odd: x_{a+1}, written in layout(2^ld) (cell = flattened position); a even: nullptr
    const unsigned short* apk[3];   // weights in the fragment order of the single-layer kernel
    const float* border[3];         // (16, C padded to 8) border-bias tables, or nullptr
    float inv_scale[3];             // 2^-S of the fp16 weights (1 for bf16)
    int first_even;        // parity of a
    int B, H, W, Cout;
    int ld, ld_out;
    int Hs, Ws, total;     // sub-map size and cells of the input layout (as TileConvParams)
    int f16;               // operand / tensor type: fp16 (1) or bf16 (0)
    RangeGate rg;
    const int* postab;     // build_tile_conv_table(H, W, ld, ld_out, ld_res): the three layers share the input layout
    int cpc_in, cpc_out, cpc_res;
    int wrap_clips;        // set by the launcher: whole clips added to a (possibly negative) halo position before it is decoded
    unsigned long long* dbg_ts;   // KWS_T3_TIMING (with a -DT3_TIMING build): 12 stamps per wave for the first 8192 workgroups, or nullptr
    int debug;             // KWS_T3_DEBUG, timing experiments only (results wrong): 1 / 64 / 128 skip the first / second / third k-loop, 2 skip the staging loads, 4 skip the output stores
};
bool conv3x3_triple_supported(int C, int Ws, bool first_even);
hipError_t launch_conv3x3_triple(const TripleConvParams& p, int C, hipStream_t s);
// (r5) The same runs -- three consecutive layers of equal dilation, or one layer -- as a persistent, weight-stationary STREAM (conv3x3_stream.hip):
// one workgroup per CU, a layer's weights in one wave's registers, the layers of a run as three waves 112 positions apart on LDS rings, a loader
// wave in front of them.  Tensors, tables, parities and results are the triple / single tile kernels' (bit for bit).
struct StreamConvParams {
    const void* in;        // CL 16-bit tensor in layout(2^ld)
    const void* res;       // first layer even: x_{a-2} in layout(2^ld_res) (cells through the table); else nullptr
    void* out;             // the last layer's output, written in layout(2^ld_out) (cells through the table)
    void* out2;            // three layers, a odd: x_{a+1}, written in the input's own layout (cell = flattened position); else nullptr
    const unsigned short* apk[3];   // weights in the fragment order of the single-layer tile kernel
    const float* border[3];         // (16, 48) border-bias tables, or nullptr
    float inv_scale[3];             // 2^-S of the fp16 weights (1 for bf16)
    int first_even;        // parity of a
    int n_layers;          // 1 or 3
    int B;                 // clips (for the 32-bit offset check only)
    int Ws, total;         // sub-map width and cells of the input layout (as TileConvParams)
    int f16;               // operand / tensor type: fp16 (1) or bf16 (0)
    RangeGate rg;
    const int* postab;     // build_tile_conv_table(H, W, ld, ld_out, ld_res)
    int cpc_in, cpc_out, cpc_res;
    int span;              // set by the launcher: positions per workgroup (a multiple of 16)
};
bool conv3x3_stream_supported(int C, int Ws);
hipError_t launch_conv3x3_stream(const StreamConvParams& p, int C, int n_cu, hipStream_t s);
// First conv of the cnn-* models from an LDS image of the clip (conv_in1.hip): Cin == 1, kw == 8, no padding, fused MaxPool
struct In1ConvParams {
    const float* feat;     // (B, T, F) fp32 feature maps
    void* out;             // channels-last (B, Hq, Wq, Cp): fp32, or fp16 when out_f16
    const unsigned short* apk;   // pack_conv_in1_weights with the kernel's MH (IN1_MH3 for terms == 3, IN1_MH1 for terms == 1)
    const float* bias;     // (Cout)
    int B, T, F, Cout, Cp, mtiles;
    int kh, sh, sw, ph, pw;   // kernel rows, strides, pooling window (1 x 1: none)
    int Hq, Wq;            // pooled output map
    int terms;             // 3: two-part fp16 operands (fp32-accurate); 1: plain fp16 products
    float inv_scale;       // 2^-S of the weights
    int relu, out_f16;
    RangeGate rg;
};
constexpr int IN1_MH3 = 2, IN1_MH1 = 4;   // channel tiles a wave keeps in registers (three-term / single-term products)
bool conv_in1_supported(const ConvGeom& g, int ph, int pw);
size_t conv_in1_lds_bytes(int T, int F, int parts);
void pack_conv_in1_weights(int Cout, int kh, int mh, const float* w, float scale, std::vector<unsigned short>& dst);
hipError_t launch_conv_in1(const In1ConvParams& p, hipStream_t s);
// LDS-staged band convolution for the second conv of the cnn-* models (conv_band.hip): stride 1, no padding, bias + ReLU
struct BandConvParams {
    const void* in;        // channels-last (B, H, W, Cpi): fp32, or fp16 when terms == 1; channels >= Cin hold zeros
    float* out;            // channels-last fp32 (B, Ho, Wo, Cpo), zeros in the padding
    const unsigned short* apk;   // pack_conv_band_weights
    const float* bias;     // (Cout)
    int B, H, W, Cpi, Ho, Wo, Cout, Cpo;
    int kh, kw, ksteps;    // ksteps = ceil(kh * kw * Cpi / 32)
    int R, nbands;         // output rows per workgroup (conv_band_plan), ceil(Ho / R)
    int Wl, PS, ntiles;    // LDS row stride in cells, cells per block plane (multiple of 16), position tiles per band (<= 8)
    const int* postab;     // [ntiles][16] x {LDS cell of the position, oy << 16 | ox, or -1 for a pad lane} (BandPlan::tab)
    unsigned long long* dbg_ts;   // KWS_BAND_TIMING (with a -DBAND_TIMING build): phase stamps of the first 8192 workgroups, or nullptr
    int terms;             // 3: two-part fp16 operands (fp32-accurate); 1: one part (`fp16` dtype, fp16 input tensor)
    float inv_scale;       // 2^-S of the weights
    int relu;
    int out_f16;           // 1: `out` holds fp16 cells (round to nearest even) for a consumer that would round them to fp16 anyway (the first Linear with single-term products)
    RangeGate rg;
};
constexpr int conv_band_mh(int Cout) { return ((Cout + 15) / 16 + 1) / 2; }   // channel tiles per wave (two wave rows)
struct BandPlan {
    int R = 0, Wl = 0, PS = 0, ntiles = 0;
    std::vector<int> tab;
};
bool conv_band_plan(int Cin, int Cout, int H, int W, int kh, int kw, int parts, BandPlan& out);   // parts: 2 fp32-accurate products, 1 fp16 tensors; false: layer not supported
size_t conv_band_lds_bytes(int Cpi, int kh, int kw, int PS, int parts);
void pack_conv_band_weights(int Cin, int Cout, int kh, int kw, const float* w, float scale, std::vector<unsigned short>& dst);
hipError_t launch_conv_band(const BandConvParams& p, hipStream_t s);
// conv_cols.hip: the same layer on fp16 tensors with column tiles, persistent and double-buffered (one workgroup per CU)
struct ColsConvParams {
    const unsigned short* in;    // channels-last fp16 cells (B, H, W, Cpi)
    void* out;                   // channels-last (B, Ho, Wo, Cpo): fp16 cells (out_f16) or fp32, zeros in the padding
    const unsigned short* apk;   // pack_conv_cols_weights
    const float* bias;           // (Cout)
    int B, H, W, Cpi, Ho, Wo, Cout, Cpo;
    int kh;                      // kernel rows (four kernel columns)
    int nbands;                  // ceil(Ho / 16): bands of 16 output rows, the last one moved up to end on row Ho - 1
    float inv_scale;             // 2^-S of the weights
    int relu, out_f16;
    RangeGate rg;
    unsigned* queue;             // two device words, zero between launches (kws_internal.h, queue_retire): the unit counter
    unsigned long long* dbg_ts;  // KWS_BAND_TIMING (with a -DCOLS_TIMING build): phase stamps of the first 512 workgroups' units, or nullptr
};
bool conv_cols_supported(int Cin, int Cout, int H, int W, int kh, int kw);
void pack_conv_cols_weights(int Cin, int Cout, int kh, const float* w, float scale, std::vector<unsigned short>& dst);
hipError_t launch_conv_cols(const ColsConvParams& p, int n_cu, hipStream_t s);
// fp32 (B, C, H, W) -> pooled (stride = window, floor; 1 x 1 = transpose only) channels-last (B, H/kh, W/kw, cp) fp32
hipError_t launch_nchw_to_cl(const float* in, float* out, int B, int C, int H, int W, int kh, int kw, int is_max, int cp,
                             hipStream_t s, RangeGate rg = RangeGate{nullptr, 0});
// element type of a channels-last activation tensor of the tiled plan
enum ClType { CL_F32 = 0, CL_BF16 = 1, CL_F16 = 2 };
// conv_0 (3x3, pad 1) + ReLU [+ AvgPool(kh, kw)] of a ResNet straight into the channels-last tensor; w9 = weights as [9 taps][cp]
hipError_t launch_conv0_cl(const float* feat, const float* w9, void* out, int cl_type, int B, int T, int F, int kh, int kw, int cp,
                           hipStream_t s, RangeGate rg = RangeGate{nullptr, 0});
hipError_t launch_mean_linear_cl(const void* x, int cl_type, float* logits, int B, int C, int cp, int HW, const float* mean,
                                 const float* rstd, const float* w, const float* bias, int n_out, hipStream_t s,
                                 RangeGate rg = RangeGate{nullptr, 0});
// out[i] = (relu?)(bias[co] + sum_z partial[z][i]) for i over (B, Cout, npc)
hipError_t launch_splitk_reduce(const float* partial, float* out, const float* bias, int ksplit, long long total,
                                int Cout, int npc, int relu, hipStream_t s, RangeGate rg = RangeGate{nullptr, 0});

hipError_t launch_pool(const float* in, float* out, int planes /*B*C*/, int H, int W, int kh, int kw, int is_max,
                       hipStream_t s, RangeGate rg = RangeGate{nullptr, 0});
// sets *rg.flag when x holds a value the fp16 parts cannot carry (the caller-provided features of a CNN)
hipError_t launch_range_check(const float* x, long long n, hipStream_t s, RangeGate rg);
// logits[b] = W * ((mean_hw(x[b]) - mean) * rstd) + bias     (mean/rstd may be nullptr = identity)
hipError_t launch_mean_linear(const float* x, float* logits, int B, int C, int HW, const float* mean,
                              const float* rstd, const float* w, const float* bias, int n_out, hipStream_t s,
                              RangeGate rg = RangeGate{nullptr, 0});
hipError_t launch_eval_tail(const float* logits, const int64_t* target, int B, int n_labels, int64_t* stats,
                            double* loss_sum, hipStream_t s);

}  // namespace kws
