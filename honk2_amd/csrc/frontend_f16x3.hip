// MFCC front end on the fp16 matrix cores (default): wav -> 2*ln(mel power).  Persistent workgroups (two per CU) walk
// the units (clip, 112-frame chunk).
//
// Same algorithm and same twice-folded DFT as frontend.hip (reference utils/audio_processor.py:18-30, looped per clip
// by data_loader/audio_data_loader.py:23-35), but the four 64 x 120 GEMMs run as three-term fp16
// products on v_mfma_f32_16x16x32_f16 (the scheme of res8_f16x3.hip): 336 MFMAs of 16 cycles per wave and unit instead
// of 840 fp32-input MFMAs of 32 cycles.
//   * A = cos / sin table x 2^7, split on the host into two fp16 parts, in fragment order; a wave streams its GEMM's
//     fragments of the next k-step (K = 120 -> 4 steps of 32) from L2 while it works on the current one.
//   * B = the folded, windowed samples  hj*(x[j] +- x[480-j]) +- (1-hj)*(x[240-j] +- x[240+j]), built in fp32 exactly
//     as in frontend.hip, scaled by a per-unit power of two (2^13 for |x| < 2, less for louder input, so the fp16
//     range cannot overflow whatever the samples are), split into two fp16 parts in registers.  A lane's eight k-slots
//     are eight consecutive j, so the four operand streams are 128-bit LDS reads: two for each forward stream, two plus
//     one word for each mirrored stream (LDS index = i + 4*floor(i/160): 16-byte blocks never straddle the padding).
//     Tile column n holds frame fcol(n) -- even frames on lanes 0-3 / 12-15, odd ones on lanes 4-11 -- which makes the
//     16-lane groups a ds_read_b128 is served in conflict-free.  The k order is free, so column j = 0 (weight 0) and
//     columns 121..127 (weight 0) simply read whatever finite samples sit there.
//   * |X|^2 goes to an LDS tile P[bin][frame] over the dead sample image (each cell has one Re and one Im owner); the
//     mel stage is a banded GEMM on the fp32-input MFMA whose accumulator layout is the output layout (a lane holds
//     four consecutive bands of a frame): log and x2 happen in registers, rows leave as 16-byte stores.
//   * Latency: the next unit's samples are requested (buffer loads, 76 registers the accumulators no longer need)
//     before the mel stage and written to LDS after the output tile has been read; the unit's rows are stored last,
//     because waiting for loads would otherwise wait for younger stores (one vmcnt).
//   Inputs whose rows are not 16-byte aligned, and filterbanks that do not fit the LDS tables, take frontend.hip.
#include "kws_internal.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>

namespace kws {

namespace {
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int XS = 164;                                        // LDS words per hop: 160 samples + 4
// FE16_NT = 4 (64-frame units, 48 KB of LDS, 168 registers: three workgroups per CU) was measured: 3.79 ms against 3.26 ms
// for 65 536 clips -- 8 tiles of work per clip instead of 7, twice the A-fragment traffic, per-unit costs paid twice.
#ifndef FE16_NT
#define FE16_NT 7
#endif
#ifndef FE16_ABLATE   // timing experiments: 1 no power tile, 2 one k-step group per band tile in the mel stage, 4 no B-fragment builds in the k-loop, 8 / 16 the two mirrored single-word reads conflict-free / gone (results wrong)
#define FE16_ABLATE 0
#endif
#ifndef FE16_ORDER
#define FE16_ORDER 0
#endif
#ifndef FE16_ASPREAD  // 1: the k-loop's A-fragment requests spread over four tiles of a k-step (A/B knob)
#define FE16_ASPREAD 1
#endif
#ifndef FE16_ISSUE    // where the next unit's sample loads are requested: 0 in one go behind the power tile, 1 in three parts through the mel stage
#define FE16_ISSUE 1
#endif
constexpr int NTT = FE16_NT;                                   // 16-frame tiles per unit
constexpr int FRM = 16 * NTT;                                  // frames per unit
constexpr int WG_PER_CU = NTT <= 4 ? 3 : 2;
constexpr int X_LEN = 160 * (FRM - 1) + FE_NFFT;               // staged samples: FRM + 2 hop blocks
constexpr int X_WORDS = X_LEN + 4 * ((X_LEN + 159) / 160);
constexpr int TILE_WORDS = 16 * XS;                            // LDS words between consecutive 16-frame tiles
constexpr int HW_WORDS = 256;                                  // h[j], h[240-j] for j = 0..127
constexpr float A_SCALE = 128.f;
constexpr int PS = FRM + 6 - (FRM % 4);                        // row stride (words) of the power tile P[bin][frame]: = 2 mod 4
static_assert(PS % 4 == 2 && PS >= FRM, "rows of lane groups 0 and 1 must start 16 banks apart");
constexpr int IMG_WORDS = X_WORDS > FE_ROWS * PS ? X_WORDS : FE_ROWS * PS;   // the power tile replaces the (dead) sample image
constexpr int MEL_SLOTS = (NTT + 3) / 4;                       // frame tiles per wave in the mel stage: tiles w, w + 4, ...
constexpr int CONST_WORDS = FE16_CONST_WORDS;
constexpr int MELW_WORDS = (IMG_WORDS - FE_ROWS * PS) / 1024 * 1024;   // what the image region leaves behind the power tile, in whole passes of the workgroup
constexpr int MELW_ITERS = MELW_WORDS / 1024;
static_assert(MELW_ITERS >= 1, "no room for the mel table behind the power tile");
}  // namespace

__device__ __forceinline__ int fx_idx(int i) { return i + 4 * (i / 160); }

// phase timestamps for tools/fe_phases.py: build with -DFE16_TIMING (they overwrite part of the feature rows)
#ifdef FE16_TIMING
#define FE16_TS_DECL unsigned long long ts[13];
#define FE16_TS(i) ts[i] = __builtin_readcyclecounter();
#else
#define FE16_TS_DECL
#define FE16_TS(i)
#endif

// MODE: 0 = fp32 samples, 1 = 16-bit PCM, 2 = PCM + additive noise clip
template <int MODE>
__global__ __launch_bounds__(256, WG_PER_CU) void frontend_f16_kernel(FrontendParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* hw = lds + IMG_WORDS;                                     // [window 256]: FE16_CONST_WORDS
    unsigned* red = reinterpret_cast<unsigned*>(hw + CONST_WORDS);
    int* const next_unit = reinterpret_cast<int*>(red + 4);          // the unit this workgroup takes next (see the loop)
    const int tid0 = threadIdx.x;
    const int n = p.n_samples;
    const int nunits = p.B * p.chunks;                               // a unit = (clip, 112-frame chunk)
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

    // ---- constants (the window halves): once per workgroup
#pragma unroll
    for (int i = 0; i < (CONST_WORDS + 255) / 256; ++i)
        if (i * 256 + tid0 < CONST_WORDS) hw[i * 256 + tid0] = p.consts16[i * 256 + tid0];

    // ---- sample loads of a unit, issued without waiting.  Padded index q of the chunk is sample org + q.  Thread
    //      (bs, off) = (tid / 40, tid % 40), tid < 240, owns group `off` (four samples) of hop block 6 * it + bs in
    //      iteration it: 19 iterations cover the image's 114 blocks exactly, the LDS address is a per-thread base + an
    //      immediate and the global offset a per-thread base + a scalar (buffer loads: groups outside the clip
    //      return 0 and are not stored).  The <= 60 groups of the reflected head and the <= 61 of the tail that
    //      existing frames reach are re-read element-wise by threads 0..123 ("edge groups").  Groups past that belong
    //      to frames beyond the clip's end, whose columns are never stored: they are not staged at all.  The loads of
    //      unit i+1 are issued before the mel stage of unit i and land in registers the accumulators no longer need.
    constexpr int len4 = X_LEN / 4;
    constexpr int iters = X_LEN / 160 / 6;
    static_assert(iters * 6 * 160 == X_LEN, "19 iterations x 6 hop blocks");
    unsigned raw[iters][MODE == 0 ? 4 : (MODE == 1 ? 2 : 6)];   // fp32: four samples; PCM: four int16 [, four noise samples]
    unsigned eraw[4][MODE == 2 ? 2 : 1];              // edge group: sample bits [, noise bits] per element
    auto edge_group = [&](int tid, int t0, int org) -> int {   // padded group this thread re-reads element-wise, or -1
        const int qt = (n - org) >> 2;                // first group that reaches past the clip's end
        int qe = -1;
        if (tid < 60) qe = t0 == 0 ? tid : -1;
        else if (tid < 124) qe = qt + tid - 60;
        return (qe >= 0 && qe < len4 && (org + 4 * qe < 0 || org + 4 * qe + 3 >= n)) ? qe : -1;
    };
    auto issue = [&](int unit, int tid, int it0 = 0, int it1 = iters, bool edges = true) {
        const int clip = unit / p.chunks;
        const int t0 = (unit - clip * p.chunks) * FRM;
        const int org = 160 * t0 - FE_NFFT / 2;
        const size_t cbase = (size_t)clip * p.clip_stride;
        const int bs = tid / 40;
        const int g0 = org + 160 * bs + 4 * (tid - 40 * bs);      // first sample of this thread's group in iteration 0
        const int qe = edge_group(tid, t0, org);
        int ex[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int sx = org + 4 * max(qe, 0) + e;
            sx = sx < 0 ? -sx : sx;
            sx = sx >= n ? 2 * (n - 1) - sx : sx;
            ex[e] = max(0, min(sx, n - 1));
        }
        if (MODE == 0) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wav + cbase), 0, n * 4, 0x00020000);
#pragma unroll
            for (int it = 0; it < iters; ++it) {
                if (it < it0 || it >= it1) continue;
                const u32x4 v = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, 4 * g0 + 3840 * it, 0, 0));
#pragma unroll
                for (int e = 0; e < 4; ++e) raw[it][e] = v[e];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (edges) eraw[e][0] = __builtin_amdgcn_raw_buffer_load_b32(rs, 4 * ex[e], 0, 0);
        } else {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<short*>(p.pcm + cbase), 0, n * 2, 0x00020000);
            const __amdgpu_buffer_rsrc_t rz =       // only MODE 2 reads through it
                __builtin_amdgcn_make_buffer_rsrc(MODE == 2 ? const_cast<float*>(p.noise + cbase) : nullptr, 0, MODE == 2 ? n * 4 : 0, 0x00020000);
#pragma unroll
            for (int it = 0; it < iters; ++it) {
                if (it < it0 || it >= it1) continue;
                const u32x2 v = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, 2 * g0 + 1920 * it, 0, 0));
                raw[it][0] = v[0];
                raw[it][1] = v[1];
                if (MODE == 2) {
                    const u32x4 z = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rz, 4 * g0 + 3840 * it, 0, 0));
#pragma unroll
                    for (int e = 0; e < 4; ++e) raw[it][(MODE == 2 ? 2 : 0) + e] = z[e];
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (!edges) continue;
                eraw[e][0] = (unsigned)(int)(short)__builtin_amdgcn_raw_buffer_load_b16(rs, 2 * ex[e], 0, 0);
                if (MODE == 2) eraw[e][MODE == 2 ? 1 : 0] = __builtin_amdgcn_raw_buffer_load_b32(rz, 4 * ex[e], 0, 0);
            }
        }
    };
    auto finish = [&](float v, unsigned zbits) -> float {
#pragma clang fp contract(off)   // two roundings, like numpy's `data += noise * noise_pct` in float32 (no FMA)
        if (MODE == 2) {
            const float t = __builtin_bit_cast(float, zbits) * p.noise_pct;
            v = v + t;
        }
        return v;
    };

    // ---- write a unit's (pre-loaded) reflect-padded samples to the LDS image and reduce max |x| per wave
    auto stage = [&](int unit, int tid) {
        const int lane = tid & 63;
        const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int clip = unit / p.chunks;
        const int t0 = (unit - clip * p.chunks) * FRM;
        const int org = 160 * t0 - FE_NFFT / 2;
        const int bs = tid / 40;
        const int off = tid - 40 * bs;
        const int g0 = org + 160 * bs + 4 * off;
        float* lbase = lds + XS * bs + 4 * off;
        float mxf = 0.f;
        auto put = [&](float* dst, f32x4 v) {
            *reinterpret_cast<f32x4*>(dst) = v;
            mxf = __builtin_fmaxf(__builtin_fmaxf(mxf, __builtin_fabsf(v[0])), __builtin_fabsf(v[1]));
            mxf = __builtin_fmaxf(__builtin_fmaxf(mxf, __builtin_fabsf(v[2])), __builtin_fabsf(v[3]));
        };
        constexpr bool is_pcm = MODE != 0;
#pragma unroll
        for (int it = 0; it < iters; ++it) {
            const int s0 = g0 + 960 * it;
            if (tid < 240 && s0 >= 0 && s0 + 3 < n) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    v[e] = is_pcm ? finish((float)(short)(raw[it][(e >> 1) & (MODE ? 1 : 3)] >> (16 * (e & 1))) * (1.0f / 32768.0f), MODE == 2 ? raw[it][(MODE == 2 ? 2 : 0) + e] : 0u)
                                  : __builtin_bit_cast(float, raw[it][MODE == 0 ? e : 0]);
                put(lbase + 6 * XS * it, v);
            }
        }
        const int qe = edge_group(tid, t0, org);
        if (qe >= 0) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                v[e] = is_pcm ? finish((float)(int)eraw[e][0] * (1.0f / 32768.0f), eraw[e][MODE == 2 ? 1 : 0]) : __builtin_bit_cast(float, eraw[e][0]);
            put(lds + fx_idx(4 * qe), v);
        }
        unsigned v = __builtin_bit_cast(unsigned, mxf);       // non-negative floats order like their bit patterns
        // wave maximum in six DPP steps (quad swaps, half-row and row mirrors, the two row broadcasts): lane 63 holds it
        v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true));
        v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true));
        v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, true));
        v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, true));
        v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, true));
        v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, true));
        if (lane == 63) red[w] = v;
    };

    // Units are handed out by a device-wide counter (the first unit is blockIdx.x, ticket t is unit gridDim.x + t): the two
    // workgroups of a CU do not progress at the same rate (res8_f16x3.hip), a fixed stride leaves the favoured one idle
    // at the end.  The counter is read at the top of an iteration and published through one LDS word well before the
    // prefetch of the next unit needs it.  The counter RESETS ITSELF (kws_internal.h, queue_retire): no memset in front
    // of the launch, so the launch is one graph node whose state does not depend on another node's.
    int unit = blockIdx.x;
    if (unit < nunits) {
        issue(unit, tid0);
        stage(unit, tid0);
    }
#pragma unroll 1
    while (unit < nunits) {
    int taken = 0;
    if (tid0 == 0) taken = (int)(gridDim.x + atomicAdd(p.queue, 1u));
    int tid = tid0;
    asm volatile("" : "+v"(tid));      // lane-derived values are recomputed per unit instead of living across the loop
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4;
    const int pcol = lane & 15;
    const int clip = unit / p.chunks;
    const int chunk = unit - clip * p.chunks;
    const int t0 = chunk * FRM;
    const int nfr = min(FRM, p.T - t0);
    FE16_TS_DECL
    FE16_TS(0)

    // ---- this wave's A fragments of k-step 0 (in flight during the staging below)
    const u32x4* tab = static_cast<const u32x4*>(p.dft16) + (size_t)w * 4 * 8 * 64 + lane;
    u32x4 an[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        an[m][0] = tab[(2 * m) * 64];
        an[m][1] = tab[(2 * m + 1) * 64];
    }

    FE16_TS(1)
    __syncthreads();
    FE16_TS(2)

    // ---- power-of-two scale of the B operand: |b| <= 2 max|x| must stay below 2^15 after scaling
    const unsigned mxall = max(max(red[0], red[1]), max(red[2], red[3]));
    const int ex = max((int)(mxall >> 23) - 127, 0);
    const float b_scale = __builtin_bit_cast(float, (unsigned)(127 + 13 - ex) << 23);
    const float post = __builtin_bit_cast(float, (unsigned)(127 - 20 + ex) << 23);   // 1 / (b_scale * A_SCALE)

    f32x4 acc[4][NTT];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int j = 0; j < NTT; ++j) acc[m][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const float sgn = w >= 2 ? -1.f : 1.f;                   // s: Re rows use sums, Im rows differences
    const float tsg = (w == 1 || w == 2) ? -1.f : 1.f;       // t: sign of the (240-j) half
    // column n of a tile is frame fr(n): the 16-lane groups a ds_read_b128 is served in pair lanes 0-3 / 12-15 of one
    // k-group with lanes 4-11 of the next (8 words further on); with even frames on the former and odd frames on the
    // latter the 16 lanes of a group start in 16 different 4-bank slots
    const int fcol = pcol < 4 ? 2 * pcol : (pcol < 12 ? 2 * pcol - 7 : 2 * pcol - 16);
    const int fb = XS * fcol;

    // x[240] of every frame (the j = 0 / 240 pair of the Re rows), read before the k-loop: seven dependent LDS round trips
    // would otherwise sit between the last MFMA and the epilogue
    float c240[NTT];
#pragma unroll
    for (int j = 0; j < NTT; ++j) c240[j] = w < 2 ? tsg * lds[fb + 244 + j * TILE_WORDS] : 0.f;

    // ---- k-loop, software-pipelined by hand: the B fragment of step idx + 1 (LDS reads, fold, window, split: ~44
    //      VALU) is built in the shadow of the 12 MFMAs of step idx.  (Explicit one-MFMA-to-four-VALU scheduling
    //      groups were tried: no faster, and the group solver's compile time explodes on a block this size.)
    struct KStep {
        int pa, pd, pb, pbs, pc, pcs;
#if FE16_ABLATE & 8
        int pms_abl;
#endif
        float hj[8], hc[8];
    };
    auto setup = [&](int s, KStep& k) {
        const int j0 = 32 * s + 8 * g;
        k.pa = fb + j0;
        k.pd = fb + 240 + j0 + (240 + j0 >= 320 ? 8 : 4);
        k.pb = fb + 472 - j0 + 8;
        k.pbs = fb + 480 - j0 + 8 - (j0 == 0 ? 1 : 0);
        k.pc = fb + 232 - j0 + (232 - j0 >= 160 ? 4 : 0);
        k.pcs = fb + 240 - j0 + (240 - j0 >= 160 ? 4 : 0);
#if FE16_ABLATE & 8
        k.pms_abl = (lane & 63) + 128 * s;
#endif
        const f32x4 h0 = *reinterpret_cast<const f32x4*>(hw + j0), h1 = *reinterpret_cast<const f32x4*>(hw + j0 + 4);
        const f32x4 c0 = *reinterpret_cast<const f32x4*>(hw + 128 + j0), c1 = *reinterpret_cast<const f32x4*>(hw + 132 + j0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            k.hj[e] = h0[e] * b_scale;
            k.hj[4 + e] = h1[e] * b_scale;
            k.hc[e] = c0[e] * (b_scale * tsg);
            k.hc[4 + e] = c1[e] * (b_scale * tsg);
        }
    };
    auto build = [&](const KStep& k, int j, u32x4& bh, u32x4& bl) {
        const float* base = lds + j * TILE_WORDS;
        const f32x4 A0 = *reinterpret_cast<const f32x4*>(base + k.pa), A1 = *reinterpret_cast<const f32x4*>(base + k.pa + 4);
        const f32x4 D0 = *reinterpret_cast<const f32x4*>(base + k.pd), D1 = *reinterpret_cast<const f32x4*>(base + k.pd + 4);
        const f32x4 M0 = *reinterpret_cast<const f32x4*>(base + k.pb), M1 = *reinterpret_cast<const f32x4*>(base + k.pb + 4);
        const f32x4 N0 = *reinterpret_cast<const f32x4*>(base + k.pc), N1 = *reinterpret_cast<const f32x4*>(base + k.pc + 4);
#if FE16_ABLATE & 8      // (r5) what the four-way conflict of these two single-word reads costs: 8 = read them at one-bank-per-lane addresses instead (wrong words,
                        // same instruction count); 16 = no reads at all (the M0 / N0 words that are there anyway: an upper bound for any bpermute / carry scheme)
        const float MS = lds[j * TILE_WORDS + k.pms_abl], NS = lds[j * TILE_WORDS + k.pms_abl + 64];
#elif FE16_ABLATE & 16
        const float MS = M0[0], NS = N0[0];
#else
        const float MS = base[k.pbs], NS = base[k.pcs];
#endif
        const float xa[8] = {A0[0], A0[1], A0[2], A0[3], A1[0], A1[1], A1[2], A1[3]};
        const float xd[8] = {D0[0], D0[1], D0[2], D0[3], D1[0], D1[1], D1[2], D1[3]};
        const float xb[8] = {MS, M1[3], M1[2], M1[1], M1[0], M0[3], M0[2], M0[1]};
        const float xc[8] = {NS, N1[3], N1[2], N1[1], N1[0], N0[3], N0[2], N0[1]};
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
#ifdef FE16_PK_MATH     // packed fp32 math: half the fold / window instructions, but each pair has to sit in adjacent registers
            const f32x2 u = {fmaf(sgn, xb[e], xa[e]), fmaf(sgn, xb[e + 1], xa[e + 1])};
            const f32x2 v = {fmaf(sgn, xd[e], xc[e]), fmaf(sgn, xd[e + 1], xc[e + 1])};
            const f32x2 q = (f32x2){k.hc[e], k.hc[e + 1]} * v;
            const f32x2 b = __builtin_elementwise_fma((f32x2){k.hj[e], k.hj[e + 1]}, u, q);
#else                   // plain FMAs: no register pairing (the mirrored streams arrive in descending order), no packed-op issue cost
            float b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float u = fmaf(sgn, xb[e + i], xa[e + i]);
                const float v = fmaf(sgn, xd[e + i], xc[e + i]);
                b[i] = fmaf(k.hj[e + i], u, k.hc[e + i] * v);
            }
#endif
            const f16x2 h = __builtin_convertvector((f32x2){b[0], b[1]}, f16x2);   // one v_cvt_pk_f16_f32 (round to nearest even)
            const unsigned hp = __builtin_bit_cast(unsigned, h);
            // l = fp16(b - h): the mixed-precision FMA reads h from its half of the packed register and writes the
            // rounded fp16 result into the low / high half of the destination
            unsigned lp;
            asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lp) : "v"(hp), "v"(b[0]));
            // (s_nop 1: two wait states between a VALU result and a matrix instruction reading it -- hipcc pads nothing for
            // instructions inside an asm statement; the fragment is built one step ahead, this is the belt to those braces)
            asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\ts_nop 1" : "+v"(lp) : "v"(hp), "v"(b[1]));
            bh[e >> 1] = hp;
            bl[e >> 1] = lp;
        }
    };
    {
        KStep ks;
        setup(0, ks);
        u32x4 bh_c, bl_c;
        build(ks, 0, bh_c, bl_c);
        u32x4 a[4][2];
#pragma unroll
        for (int idx = 0; idx < 4 * NTT; ++idx) {
            const int s = idx / NTT, j = idx - s * NTT;
            if (j == 0) {
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    a[m][0] = an[m][0];
                    a[m][1] = an[m][1];
                }
#if !FE16_ASPREAD
                if (s + 1 < 4) {
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        an[m][0] = tab[((s + 1) * 8 + 2 * m) * 64];
                        an[m][1] = tab[((s + 1) * 8 + 2 * m + 1) * 64];
                    }
                }
#endif
            }
#if FE16_ASPREAD   // the next k-step's eight fragments requested over the first four tiles of this one (2 per tile), not in one burst
            if (s + 1 < 4 && j < 4) {
                an[j][0] = tab[((s + 1) * 8 + 2 * j) * 64];
                an[j][1] = tab[((s + 1) * 8 + 2 * j + 1) * 64];
            }
#endif
            u32x4 bh_n = bh_c, bl_n = bl_c;
            if (idx + 1 < 4 * NTT) {
                const int s1 = (idx + 1) / NTT, j1 = (idx + 1) - s1 * NTT;
                if (j1 == 0) setup(s1, ks);
                if (!(FE16_ABLATE & 4)) build(ks, j1, bh_n, bl_n);     // (4: one B fragment for the whole k-loop -- no LDS reads, folds or splits)
            }
#define FMF(A_, B_, C_) C_ = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, A_), __builtin_bit_cast(f16x8, B_), C_, 0, 0, 0)
#if FE16_ORDER == 1     // chain-major: the three terms of an accumulator back to back
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                FMF(a[m][1], bh_c, acc[m][j]);
                FMF(a[m][0], bl_c, acc[m][j]);
                FMF(a[m][0], bh_c, acc[m][j]);
            }
#else
#pragma unroll
            for (int m = 0; m < 4; ++m) FMF(a[m][1], bh_c, acc[m][j]);
#pragma unroll
            for (int m = 0; m < 4; ++m) FMF(a[m][0], bl_c, acc[m][j]);
#pragma unroll
            for (int m = 0; m < 4; ++m) FMF(a[m][0], bh_c, acc[m][j]);
#endif
#undef FMF
            bh_c = bh_n;
            bl_c = bl_n;
        }
    }

    FE16_TS(3)
    // ---- undo the scales; the (j = 0, j = 240) pair: Re X[k] += (-1)^k x[240] (h[240] = 1, h[0] = 0); square
    f32x2 sq[4][NTT][2];
#pragma unroll
    for (int j = 0; j < NTT; ++j) {
        const float c = c240[j];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const f32x2 x = __builtin_elementwise_fma((f32x2){acc[m][j][2 * hh], acc[m][j][2 * hh + 1]}, (f32x2){post, post}, (f32x2){c, c});
                sq[m][j][hh] = x * x;
            }
    }
    FE16_TS(8)
    if (tid == 0) *next_unit = taken;
    // the mel stage's A table (a multiple of 1 024 fp32 words) goes to the tail of the image region that the power tile leaves free, once the
    // samples are dead: global_load_lds_dwordx4 (gfx950: memory -> LDS without passing through registers; lane i of a wave lands at base + 16 i),
    // waited for in front of the barrier that precedes the mel stage, which reads it as ds_read_b32 (one bank per lane)
    float* const melw_lds = lds + FE_ROWS * PS;
    const int mel_words = ((p.mel_ns[0] + p.mel_ns[1] + p.mel_ns[2]) * 64 + 1023) & ~1023;   // (build_mel_gemm_table pads the table to that)
    __syncthreads();  // every wave is done with the sample image
    FE16_TS(9)
#pragma unroll
    for (int i = 0; i < MELW_ITERS; ++i)
        if (i * 1024 < mel_words)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.mel_a + i * 1024 + w * 256 + lane * 4),
                                             (__attribute__((address_space(3))) void*)(melw_lds + i * 1024 + w * 256), 16, 0, 0);

    // ---- power tile P[bin][frame] = Re^2 + Im^2 (row stride PS words).  Each cell has one Re and one Im owner: in
    //      the first half-phase the Re waves store their row tiles 0-1 and the Im waves their row tiles 2-3, in the
    //      second each adds the other half into the cells its partner stored (a + b = b + a exactly; plain
    //      read-add-write: ds_add_f32 was measured 8x slower).
    const int kpar = w & 1;   // waves 0,2: even bins; 1,3: odd bins
    float* pt = lds + (8 * g + kpar) * PS + fcol;
#define P_PHASE(M0, OP)                                                                        \
    _Pragma("unroll") for (int mm = 0; mm < 2; ++mm)                                           \
    _Pragma("unroll") for (int j = 0; j < NTT; ++j)                                          \
    _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                            \
        float* cell = pt + (2 * (16 * ((M0) + mm) + r)) * PS + 16 * j;                         \
        const float v = sq[(M0) + mm][j][r >> 1][r & 1];                                       \
        OP;                                                                                    \
    }
#if FE16_ABLATE & 1   // timing experiment: no power tile (results wrong)
    {
        float keep = 0.f;     // (every square stays live: one add per value instead of the LDS traffic)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int j = 0; j < NTT; ++j) keep += (sq[m][j][0][0] + sq[m][j][0][1]) + (sq[m][j][1][0] + sq[m][j][1][1]);
        if (keep == 12345.f) *pt = keep;
    }
    __syncthreads();
#else
    if (w < 2) { P_PHASE(0, *cell = v) } else { P_PHASE(2, *cell = v) }
    __syncthreads();
    FE16_TS(10)
    {   // (r4) reads first, then the adds and stores, one row tile (28 cells) at a time: written as `*cell += v` the compiler kept every read behind the
        // store in front of it -- 56 dependent LDS round trips, 5 - 6 k cycles of a unit's ~45 k
        float got[NTT][4];
#define P_TILE(M_, OP)                                                                         \
    _Pragma("unroll") for (int j = 0; j < NTT; ++j)                                            \
    _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                            \
        float* cell = pt + (2 * (16 * (M_) + r)) * PS + 16 * j;                                \
        const float v = sq[M_][j][r >> 1][r & 1];                                              \
        OP;                                                                                    \
    }
#define P_READ(M0) { P_TILE(M0, (got[j][r] = *cell, (void)v)) }
#define P_ADD(M0) { P_TILE(M0, *cell = got[j][r] + v) }
        if (w < 2) {
            P_READ(2)
            P_ADD(2)
            P_READ(3)
            P_ADD(3)
        } else {
            P_READ(0)
            P_ADD(0)
            P_READ(1)
            P_ADD(1)
        }
#undef P_TILE
#undef P_READ
#undef P_ADD
    }
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's part of the mel table has landed (requested two barriers ago)
#undef P_PHASE
    FE16_TS(11)
    __syncthreads();
    FE16_TS(12)
    const int nxt = *next_unit;
#if FE16_ISSUE == 0
    issue(min(nxt, nunits - 1), tid);   // the accumulators are dead: prefetch the next unit's samples
#elif FE16_ISSUE == 1                   // in three parts, one in front of each band tile of the mel stage
    issue(min(nxt, nunits - 1), tid, 0, 7, false);
#endif

    FE16_TS(4)
    // ---- mel + log + "DCT of length 1" (x2).  mel = W (bands x bins) . P (bins x frames) is a banded GEMM: band tile m (16
    //      bands) only meets the bins of its own filters, p.mel_ns[m] k-steps of four bins (33 steps for the 40-band bank
    //      instead of 3 x 32).  It runs on the fp32-input MFMA (exact fp32 products and sums; the pipe is otherwise idle
    //      here): wave w takes frame tiles w and w + 4, one A fragment (global, 8 KB table, L1-resident) and one ds_read_b32
    //      per tile and step.  The result leaves in accumulator layout -- a lane holds four consecutive bands of one
    //      frame, 16 bytes of the output row -- so there is no transposed tile, no second pass over LDS and one barrier
    //      less than with the band-per-lane FMA chains this replaces (8-10 k + 4.5 k cycles per unit -> ~4 k).
    f32x4 macc[MEL_SLOTS][3];
#pragma unroll
    for (int q = 0; q < MEL_SLOTS; ++q)
#pragma unroll
        for (int m = 0; m < 3; ++m) macc[q][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    {
        // (r4) the A fragments come from the LDS copy of the table made above: read from global memory right in front of its own MFMAs, every
        // group of four k-steps waited for an L2 round trip (9 groups x ~600 cycles of a unit's ~45 k)
        const float* atab = melw_lds + lane;
        int pcol_off[MEL_SLOTS];
#pragma unroll
        for (int q = 0; q < MEL_SLOTS; ++q) pcol_off[q] = 16 * min(w + 4 * q, NTT - 1) + pcol;   // a tile past the end re-reads the last one
        int step = 0;
#pragma unroll
        for (int m = 0; m < 3; ++m) {
#if FE16_ISSUE == 1
            if (m == 1) issue(min(nxt, nunits - 1), tid, 7, 13, false);
            if (m == 2) issue(min(nxt, nunits - 1), tid, 13, iters, true);
            __builtin_amdgcn_sched_barrier(0);
#endif
            const int ns = p.mel_ns[m];
            const float* prow = lds + (4 * p.mel_fb[m] + g) * PS;
            for (int s0 = 0; s0 < ((FE16_ABLATE & 2) ? 4 : ns); s0 += 4, step += 4) {   // the host pads every tile to a multiple of four steps
                float a[4], x[4][MEL_SLOTS];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    a[u] = atab[(step + u) * 64];
#pragma unroll
                    for (int q = 0; q < MEL_SLOTS; ++q) x[u][q] = prow[4 * (s0 + u) * PS + pcol_off[q]];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int q = 0; q < MEL_SLOTS; ++q)
                        macc[q][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], x[u][q], macc[q][m], 0, 0, 0);
            }
        }
    }
    FE16_TS(5)
    // log and x2 in registers: D[band 16 m + 4 g + r][frame 16 (w + 4 q) + pcol].  v_log_f32 flushes subnormal inputs; powers
    // that small (below 1e-30: digital near-silence) are rare, so ONE test covers all of a lane's values and the exact
    // logf() runs for the lanes that need it, instead of a branch per value
    f32x4 outv[MEL_SLOTS][3];
    bool tiny = false;
#pragma unroll
    for (int q = 0; q < MEL_SLOTS; ++q)
#pragma unroll
        for (int m = 0; m < 3; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = macc[q][m][r];
                tiny |= v > 0.f && v < 1e-30f;
                outv[q][m][r] = v >= 1e-30f ? 1.3862943611198906f * __builtin_amdgcn_logf(v) : 2.0f * v;   // 2 ln 2 log2 v (bare v_log_f32)
            }
    if (__builtin_expect(tiny, 0)) {
#pragma unroll 1
        for (int i = 0; i < MEL_SLOTS * 3; ++i)
#pragma unroll 1
            for (int r = 0; r < 4; ++r) {
                const int q = i / 3, m = i - 3 * q;
                float v = 0.f;
#pragma unroll
                for (int qq = 0; qq < MEL_SLOTS; ++qq)          // (register arrays: select with compile-time indices)
#pragma unroll
                    for (int mm = 0; mm < 3; ++mm)
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr)
                            if (qq == q && mm == m && rr == r) v = macc[qq][mm][rr];
                if (v > 0.f && v < 1e-30f) {
                    const float fix = 2.0f * logf(v);
#pragma unroll
                    for (int qq = 0; qq < MEL_SLOTS; ++qq)
#pragma unroll
                        for (int mm = 0; mm < 3; ++mm)
#pragma unroll
                            for (int rr = 0; rr < 4; ++rr)
                                if (qq == q && mm == m && rr == r) outv[qq][mm][rr] = fix;
                }
            }
    }
    FE16_TS(6)
    FE16_TS(7)
    __syncthreads();   // every wave has read the power tile: the next unit's samples may overwrite it
    if (nxt < nunits) stage(nxt, tid);
    {   // the unit's rows are stored last: waiting for the prefetched samples must not wait for these stores (one vmcnt)
        float* dst = p.feat + ((size_t)clip * p.T + t0) * p.n_mels;
#pragma unroll
        for (int q = 0; q < MEL_SLOTS; ++q) {
            const int jt = w + 4 * q, frame = 16 * jt + pcol;
            if (jt < NTT && frame < nfr) {
#pragma unroll
                for (int m = 0; m < 3; ++m)
                    if (16 * m + 4 * g < p.n_mels) *reinterpret_cast<f32x4*>(dst + (size_t)frame * p.n_mels + 16 * m + 4 * g) = outv[q][m];
            }
        }
    }
#ifdef FE16_TIMING
    if (lane == 0) {
        const unsigned long long t8 = __builtin_readcyclecounter();
        unsigned long long* o = reinterpret_cast<unsigned long long*>(p.feat + ((size_t)clip * p.T + t0) * p.n_mels + 40 * (1 + w * 10));
        for (int i = 0; i < 8; ++i) o[i] = ts[i];
        o[8] = t8;
        for (int i = 8; i < 13; ++i) o[i + 1] = ts[i];
    }
#endif
    unit = nxt;
    }
    if (tid0 == 0) queue_retire(p.queue);
}

int frontend_f16_frames() { return FRM; }

// Banded mel GEMM (see the kernel's mel stage): k-steps of four bins per 16-band tile; n_mels must be a multiple of four
// (a lane stores four consecutive bands with one 16-byte store) and at most 48.
bool build_mel_gemm_table(const std::vector<float>& wts, const std::vector<int>& lo, const std::vector<int>& hi, int n_mels,
                          std::vector<float>& tab, int (&fb)[3], int (&ns)[3]) {
    if (n_mels <= 0 || n_mels > FE16_MAX_MELS || n_mels % 4) return false;
    int total = 0;
    for (int m = 0; m < 3; ++m) {
        int b_lo = FE_ROWS, b_hi = 0;
        for (int f = 16 * m; f < std::min(n_mels, 16 * m + 16); ++f)
            if (hi[f] > lo[f]) {
                b_lo = std::min(b_lo, lo[f]);
                b_hi = std::max(b_hi, hi[f]);
            }
        fb[m] = b_hi > b_lo ? b_lo / 4 : 0;
        ns[m] = b_hi > b_lo ? ((b_hi + 3) / 4 - fb[m] + 3) / 4 * 4 : 0;     // padded to a multiple of four steps (zero weights)
        if (4 * (fb[m] + ns[m]) > FE_ROWS) fb[m] = FE_ROWS / 4 - ns[m];      // the padding goes in front instead
        if (fb[m] < 0) return false;
        total += ns[m];
    }
    if (total > FE16_MAX_STEPS || total * 64 > MELW_WORDS) return false;     // (the kernel keeps the table in LDS behind its power tile)
    tab.assign(((size_t)std::max(total, 1) * 64 + 1023) / 1024 * 1024, 0.f);   // whole passes of the workgroup's copy loop
    int step = 0;
    for (int m = 0; m < 3; ++m)
        for (int s = 0; s < ns[m]; ++s, ++step)
            for (int lane = 0; lane < 64; ++lane) {
                const int f = 16 * m + (lane & 15), k = 4 * (fb[m] + s) + (lane >> 4);
                if (f < n_mels && k < FE_ROWS) tab[(size_t)step * 64 + lane] = wts[(size_t)f * FE_ROWS + k];
            }
    return true;
}

size_t frontend_f16_lds_bytes() {
    static_assert(FE_ROWS * PS <= IMG_WORDS, "power tile must fit the image region");
    const size_t words = (size_t)IMG_WORDS + CONST_WORDS + 8;
    return ((words * sizeof(float)) + 15) & ~(size_t)15;
}

hipError_t launch_frontend_f16(const FrontendParams& p, int n_cu, hipStream_t s) {
    static DeviceOnce attr_once;
    const size_t lds = frontend_f16_lds_bytes();
    if (attr_once.first()) {
        for (const void* k : {(const void*)frontend_f16_kernel<0>, (const void*)frontend_f16_kernel<1>, (const void*)frontend_f16_kernel<2>}) {
            hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
    }
    if (p.B <= 0) return hipSuccess;
    const long long units = (long long)p.B * p.chunks;
    static const int wgs_env = experiment_int("KWS_FE_WGS_PER_CU", 0);
    const int wgs = wgs_env > 0 && wgs_env < WG_PER_CU ? wgs_env : WG_PER_CU;
    const dim3 grid((unsigned)std::min<long long>(units, (long long)wgs * n_cu));      // persistent
    if (p.wav) hipLaunchKernelGGL(frontend_f16_kernel<0>, grid, dim3(256), lds, s, p);
    else if (!p.noise) hipLaunchKernelGGL(frontend_f16_kernel<1>, grid, dim3(256), lds, s, p);
    else hipLaunchKernelGGL(frontend_f16_kernel<2>, grid, dim3(256), lds, s, p);
    return hipGetLastError();
}

// A fragments: [GEMM w][k-step][row tile][part][lane] x 4 words (k-slots 8g .. 8g+7 of row 16m + (lane & 15));
// hann2 = h[j] (128), h[240-j] (128)
void build_dft_table_f16(std::vector<unsigned>& tab, std::vector<float>& hann2) {
    tab.assign((size_t)4 * 4 * 4 * 2 * 64 * 4, 0u);
    hann2.assign(HW_WORDS, 0.f);
    const double two_pi = 6.283185307179586476925286766559;
    for (int w = 0; w < 4; ++w)
        for (int s = 0; s < 4; ++s)
            for (int m = 0; m < 4; ++m)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 8; ++e) {
                        const int k = 2 * (16 * m + (lane & 15)) + (w & 1);
                        const int j = 32 * s + 8 * (lane >> 4) + e;
                        double v = 0.0;
                        if (j >= 1 && j <= 120) {
                            const int ph = (int)(((long long)k * j) % FE_NFFT);   // exact angle reduction
                            const double ang = two_pi * ph / FE_NFFT;
                            v = (w < 2 ? std::cos(ang) : std::sin(ang)) * (j == 120 ? 0.5 : 1.0) * A_SCALE;
                        }
                        const float vf = (float)v;
                        const unsigned short h = f16_rne_host(vf);
                        const unsigned short l = f16_rne_host(vf - f16_to_f_host(h));
                        const size_t base = (((((size_t)w * 4 + s) * 4 + m) * 2) * 64 + lane) * 4 + (e >> 1);
                        tab[base] |= (unsigned)h << (16 * (e & 1));
                        tab[base + 64 * 4] |= (unsigned)l << (16 * (e & 1));
                    }
    for (int j = 0; j < 128; ++j) {
        hann2[j] = (float)(0.5 - 0.5 * std::cos(two_pi * j / FE_NFFT));
        hann2[128 + j] = (float)(0.5 - 0.5 * std::cos(two_pi * (240 - j) / FE_NFFT));
    }
}

}  // namespace kws
