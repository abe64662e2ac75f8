// MFCC front end for gfx950: wav -> 2*ln(mel power), one workgroup per (clip, 112-frame chunk).
//
// Replaces AudioProcessor.compute_mfccs (reference utils/audio_processor.py:18-30) looped per clip by
// AudioDataLoader.collate_fn (data_loader/audio_data_loader.py:23-35).  Algorithm:
// reflect-pad n_fft/2, frames of 480 at hop 160, periodic Hann, |rFFT|^2, Slaney mel, v>0 ? ln v : v, x2.
//
// MI355X mapping
//   * The 480-point real DFT of every Hann-windowed frame xw[n] = h[n] x[n] runs on the fp32 matrix cores
//     (v_mfma_f32_16x16x4_f32), folded TWICE so that K shrinks from 480 to 120 per output:
//       fold 1 (n <-> 480-n):  e[j] = xw[j] + xw[480-j],  o[j] = xw[j] - xw[480-j]          j = 1..239
//       fold 2 (j <-> 240-j):  Re X[k] = sum_{j=1..120} (e[j] + (-1)^k e[240-j]) cos(2 pi k j/480)   (+-x[240])
//                              Im X[k] = sum_{j=1..120} (o[j] - (-1)^k o[240-j]) sin(2 pi k j/480)
//     (the j = 120 column pairs with itself and carries weight 1/2; the j = 0 / 240 pair reduces to +-x[240] and
//     is added in the epilogue).  That is four GEMMs -- {Re, Im} x {even k, odd k} -- of 64 rows x 120: 3 360 MFMAs
//     per clip instead of 13 440 for the plain 480-term form.  h[240-j] = 1 - h[j], so a B fragment is
//       hj*(x[j] + s x[480-j]) + t*(1-hj)*(x[240-j] + s x[240+j]),   (s, t) = (+,+) (+,-) (-,-) (-,+)
//     = 4 LDS reads + 4 VALU for 4 MFMAs.  Wave w owns GEMM w (4 row tiles x 7 frame tiles = 28 accumulators).
//   * The clip is staged ONCE in LDS with reflect padding applied; LDS index = i + 2*ceil(i/160), so the 16 frames
//     (stride 160 samples -> 162 words) x 2 k-slots read by a 32-lane half hit 32 distinct banks.  All four read
//     addresses are (per-step lane register) + (frame-tile immediate): no address arithmetic per fragment.
//   * |X|^2: the two Re waves store re^2 into the power tile (which overwrites the clip image), the two Im waves add
//     im^2 (each cell has exactly one Re and one Im owner, so a plain read-modify-write after a barrier suffices).  The 40 x (3..13 non-zero) mel sum, the log and the x2 ("DCT" of length 1) run per
//     output element and are stored fully coalesced ((T,40) rows are contiguous).
#include "kws_internal.h"

#include <cmath>

namespace kws {

namespace {
constexpr int SIG_WORDS = 18468;              // fe_idx(160*111 + 479) + 1: all 112 frames of a chunk
constexpr int HW_WORDS = FE_STEPS * 4 * 2;    // (hj, 1-hj) per (k-step, k-slot)
constexpr int TILE_STRIDE = 16 * 162;         // LDS words between consecutive 16-frame tiles
}  // namespace

__device__ __forceinline__ int fe_idx(int i) { return i + 2 * ((i + 159) / 160); }

__global__ __launch_bounds__(256, 2) void frontend_kernel(FrontendParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* hw = lds + SIG_WORDS;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4;
    const int pcol = lane & 15;
    const int clip = blockIdx.x / p.chunks;
    const int chunk = blockIdx.x - clip * p.chunks;
    const int t0 = chunk * FE_FRAMES;
    const int nfr = min(FE_FRAMES, p.T - t0);
    const int n = p.n_samples;

    // ---- stage the reflect-padded samples of all 112 frames of this chunk (padded index i = 160*t0 + li);
    //      frames past the end of the clip read clamped samples and are never stored
    {
        // input is either fp32 waveforms, or 16-bit PCM (x / 32768, the on-disk format the reference decodes with
        // librosa.load) with an optional additive noise clip: x += noise * noise_pct (dataset/gsc_dataset.py:163-174)
        // clip_stride == n_samples for a packed batch; smaller for overlapping windows of one long stream read in place
        const float* src = p.wav ? p.wav + (size_t)clip * p.clip_stride : nullptr;
        const short* pcm = p.pcm ? p.pcm + (size_t)clip * p.clip_stride : nullptr;
        const float* nz = p.noise ? p.noise + (size_t)clip * p.clip_stride : nullptr;
        auto sample = [&](int sx) -> float {
#pragma clang fp contract(off)   // two roundings, like numpy's `data += noise * noise_pct` in float32 (no FMA)
            float v = pcm ? (float)pcm[sx] * (1.0f / 32768.0f) : src[sx];
            if (nz) {
                const float t = nz[sx] * p.noise_pct;
                v = v + t;
            }
            return v;
        };
        constexpr int len4 = (160 * (FE_FRAMES - 1) + FE_NFFT) / 4;     // 4560 groups of 4 samples
        constexpr int iters = (len4 + 255) / 256;                       // 18: all loads are issued before any store
        const bool vec_ok = (p.clip_stride & 3) == 0 && !pcm && !nz;    // fp32 clip rows 16-byte aligned
        f32x4 val[iters];
#pragma unroll
        for (int it = 0; it < iters; ++it) {
            const int q4 = it * 256 + tid;
            const int li = 4 * q4;
            const int s0 = 160 * t0 + li - FE_NFFT / 2;
            if (q4 < len4 && vec_ok && s0 >= 0 && s0 + 3 < n) {
                val[it] = *reinterpret_cast<const f32x4*>(src + s0);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    int sx = s0 + e;
                    sx = sx < 0 ? -sx : sx;
                    sx = sx >= n ? 2 * (n - 1) - sx : sx;
                    sx = max(0, min(sx, n - 1));
                    val[it][e] = sample(sx);
                }
            }
        }
#pragma unroll
        for (int it = 0; it < iters; ++it) {
            const int q4 = it * 256 + tid;
            if (q4 < len4) {
                const int li = 4 * q4;                       // li % 4 == 0: only element 0 can sit on a 160-boundary
                const int f0 = fe_idx(li), f1 = fe_idx(li + 1);
                lds[f0] = val[it][0];
                lds[f1] = val[it][1];
                lds[f1 + 1] = val[it][2];
                lds[f1 + 2] = val[it][3];
            }
        }
        if (tid < HW_WORDS) hw[tid] = p.hann[tid];
    }
    __syncthreads();

    f32x4 acc[4][FE_NT];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int j = 0; j < FE_NT; ++j) acc[m][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const float sgn = w >= 2 ? -1.f : 1.f;                   // s: Re rows use sums, Im rows differences
    const float tsg = (w == 1 || w == 2) ? -1.f : 1.f;       // t: sign of the (240-j) half
    const int bp = 162 * pcol + g;                           // forward reads  x[j], x[240+j]
    const int bm = 162 * pcol + 3 - g;                       // mirrored reads x[480-j], x[240-j]

    const f32x4* tab = p.dft + (size_t)w * FE_GROUPS * 4 * 64 + lane;
    f32x4 n0 = tab[0], n1 = tab[64], n2 = tab[128], n3 = tab[192];
    for (int grp = 0; grp < FE_GROUPS; ++grp) {
        const f32x4 a0 = n0, a1 = n1, a2 = n2, a3 = n3;
        if (grp + 1 < FE_GROUPS) {
            const f32x4* tn = tab + (size_t)(grp + 1) * 4 * 64;
            n0 = tn[0]; n1 = tn[64]; n2 = tn[128]; n3 = tn[192];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int s = 4 * grp + q;
            if (s < FE_STEPS) {
                // j = 4s + g + 1;  in-frame LDS offset of sample i is i + 2*ceil(i/160)
                const int pa = bp + 4 * s + 3;                                       // i = j        (1..120)
                const int pd = bp + 4 * s + 245 + (s >= 20 ? 2 : 0);                 // i = 240 + j  (241..360)
                const int pb = bm + 482 - 4 * s;                                     // i = 480 - j  (360..479)
                const int pc = bm + 238 - 4 * s + ((4 * s + g + 1) < 80 ? 2 : 0);    // i = 240 - j  (120..239)
                const float hj = hw[(4 * s + g) * 2];
                const float hc = tsg * hw[(4 * s + g) * 2 + 1];
#pragma unroll
                for (int j = 0; j < FE_NT; ++j) {
                    const float xa = lds[pa + j * TILE_STRIDE];
                    const float xb = lds[pb + j * TILE_STRIDE];
                    const float xc = lds[pc + j * TILE_STRIDE];
                    const float xd = lds[pd + j * TILE_STRIDE];
                    const float u = fmaf(sgn, xb, xa);
                    const float v = fmaf(sgn, xd, xc);
                    const float b = fmaf(hj, u, hc * v);
                    acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[q], b, acc[0][j], 0, 0, 0);
                    acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[q], b, acc[1][j], 0, 0, 0);
                    acc[2][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[q], b, acc[2][j], 0, 0, 0);
                    acc[3][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a3[q], b, acc[3][j], 0, 0, 0);
                }
            }
        }
    }

    // ---- the (j = 0, j = 240) pair: Re X[k] += (-1)^k x[240] (h[240] = 1, h[0] = 0)
    if (w < 2) {
#pragma unroll
        for (int j = 0; j < FE_NT; ++j) {
            const float c = tsg * lds[162 * pcol + 244 + j * TILE_STRIDE];
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[m][j][r] += c;
        }
    }
    __syncthreads();  // every wave is done with the sample image

    // ---- power tile P[bin][frame] (row stride 116 words): Re waves store, then Im waves accumulate
    const int kpar = w & 1;   // waves 0,2: even bins; 1,3: odd bins
    if (w < 2) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int j = 0; j < FE_NT; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = acc[m][j][r];
                    lds[(2 * (16 * m + 4 * g + r) + kpar) * FE_PSTRIDE + 16 * j + pcol] = v * v;
                }
    }
    __syncthreads();
    if (w >= 2) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int j = 0; j < FE_NT; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = acc[m][j][r];
                    lds[(2 * (16 * m + 4 * g + r) + kpar) * FE_PSTRIDE + 16 * j + pcol] += v * v;   // sole writer of this cell
                }
    }
    __syncthreads();

    // ---- mel + log + "DCT of length 1" (x2).  Thread = one mel band for a strip of frames: its (<= 16) filter
    //      weights stay in registers; consecutive threads write consecutive bands of a frame -> coalesced stores.
    float* dst = p.feat + ((size_t)clip * p.T + t0) * p.n_mels;
    if (p.mel_maxw <= 16) {
        const int rows = 256 / p.n_mels;                    // frames handled per sweep (6 for 40 bands)
        const int tr = tid / p.n_mels;
        const int f = tid - tr * p.n_mels;
        if (tr < rows) {
            const int lo = p.mel_lo[f], hi = p.mel_hi[f];
            float wv[16];
            int row[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int k = lo + i;
                wv[i] = k < hi ? p.melw[f * FE_ROWS + k] : 0.f;
                row[i] = (k < FE_ROWS ? k : FE_ROWS - 1) * FE_PSTRIDE;
            }
            for (int tl = tr; tl < nfr; tl += rows) {
                float v = 0.f;
#pragma unroll
                for (int i = 0; i < 16; ++i) v = fmaf(wv[i], lds[row[i] + tl], v);
                const float lg = v >= 1e-30f ? __logf(v) : (v > 0.f ? logf(v) : v);
                dst[tl * p.n_mels + f] = 2.0f * lg;
            }
        }
    } else {   // unusually wide filters: generic path
        const int nout = nfr * p.n_mels;
        for (int i = tid; i < nout; i += 256) {
            const int tl = i / p.n_mels;
            const int f = i - tl * p.n_mels;
            const float* wrow = p.melw + f * FE_ROWS;
            float v = 0.f;
            for (int k = p.mel_lo[f]; k < p.mel_hi[f]; ++k) v = fmaf(wrow[k], lds[k * FE_PSTRIDE + tl], v);
            dst[i] = 2.0f * (v > 0.f ? logf(v) : v);
        }
    }
}

size_t frontend_lds_bytes(int) {
    size_t words = (size_t)SIG_WORDS + HW_WORDS;
    const size_t pw = (size_t)FE_ROWS * FE_PSTRIDE;
    if (pw > words) words = pw;
    return ((words * sizeof(float)) + 15) & ~(size_t)15;
}

hipError_t launch_frontend(const FrontendParams& p, hipStream_t s) {
    static DeviceOnce attr_once;
    const size_t lds = frontend_lds_bytes(p.T);
    if (attr_once.first()) {
        hipError_t e = hipFuncSetAttribute((const void*)frontend_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds);
        if (e != hipSuccess) return e;
    }
    if (p.B <= 0) return hipSuccess;
    hipLaunchKernelGGL(frontend_kernel, dim3((unsigned)(p.B * p.chunks)), dim3(256), lds, s, p);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ streaming windows
// Sliding windows of one long stream (reference dataset/dataset_utils.py:20-98) whose shift is a multiple of the hop share
// their STFT: frame t of window n is centred on stream sample shift * n + 160 t, and for 2 <= t <= T - 3 its 480 samples
// lie inside the window, so it is simply frame (shift / 160) * n + t of the stream taken as ONE long clip ("global"
// frames G, computed once by frontend_kernel).  Only the two first and two last frames of a window touch its reflect
// padding and are its own.  One workgroup per window: the four edge frames by a direct DFT on the vector units (4 x 128
// bins x 480 terms), their mel / log rows, and a copy of the T - 4 shared rows out of G.
__global__ __launch_bounds__(256) void window_edges_kernel(WindowEdgeParams p) {
    __shared__ f32x4 y[FE_NFFT];            // y[i] = windowed sample i of the four edge frames
    __shared__ f32x2 trig[FE_NFFT];         // cos, sin of 2 pi j / 480
    __shared__ f32x4 part[2][2][FE_ROWS];   // [half of the sum][re / im][bin] x four frames
    __shared__ f32x4 pw[FE_ROWS];           // power of the four frames
    const int tid = threadIdx.x;
    const int win = blockIdx.x;
    const float* src = p.stream + (size_t)win * p.shift;
    const int n = p.window, T = p.T;
    const int ft[4] = {0, 1, T - 2, T - 1};
    for (int i = tid; i < FE_NFFT; i += 256) {
        trig[i] = p.trig[i];
        const float hv = p.hann[i];
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int sx = FE_HOP * ft[e] - FE_NFFT / 2 + i;   // reflect padding about the window's own ends
            sx = sx < 0 ? -sx : sx;
            sx = sx >= n ? 2 * (n - 1) - sx : sx;
            sx = max(0, min(sx, n - 1));
            v[e] = hv * src[sx];
        }
        y[i] = v;
    }
    __syncthreads();
    {   // thread = (bin k, half of the 480 terms); re += y cos, im -= y sin
        const int k = tid & (FE_ROWS - 1), half = tid >> 7;
        f32x4 re = (f32x4){0.f, 0.f, 0.f, 0.f}, im = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int i0 = half * (FE_NFFT / 2);
        int idx = (int)(((long long)k * i0) % FE_NFFT);
        for (int i = i0; i < i0 + FE_NFFT / 2; ++i) {
            const f32x2 cs = trig[idx];
            const f32x4 v = y[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                re[e] = fmaf(v[e], cs[0], re[e]);
                im[e] = fmaf(v[e], cs[1], im[e]);
            }
            idx += k;
            idx = idx >= FE_NFFT ? idx - FE_NFFT : idx;
        }
        part[half][0][k] = re;
        part[half][1][k] = im;
    }
    __syncthreads();
    if (tid < FE_ROWS) {
        const f32x4 re = part[0][0][tid] + part[1][0][tid], im = part[0][1][tid] + part[1][1][tid];
        pw[tid] = re * re + im * im;
    }
    __syncthreads();
    float* dst = p.feat + (size_t)win * T * p.n_mels;
    for (int o = tid; o < 4 * p.n_mels; o += 256) {
        const int e = o / p.n_mels, f = o - e * p.n_mels;
        const float* wrow = p.melw + f * FE_ROWS;
        float v = 0.f;
        for (int k = p.mel_lo[f]; k < p.mel_hi[f]; ++k) v = fmaf(wrow[k], pw[k][e], v);
        const float lg = v >= 1e-30f ? __logf(v) : (v > 0.f ? logf(v) : v);
        dst[ft[e] * p.n_mels + f] = 2.0f * lg;
    }
    // shared rows 2 .. T - 3 are rows g0 + 2 .. of G, contiguous in both tensors
    const size_t g0 = (size_t)win * (p.shift / FE_HOP);
    const float* grow = p.global_feat + (g0 + 2) * p.n_mels;
    float* drow = dst + 2 * p.n_mels;
    const int ncopy = (T - 4) * p.n_mels;
    if (((p.n_mels * 2) & 3) == 0 && (((g0 + 2) * p.n_mels) & 3) == 0 && ((T * p.n_mels) & 3) == 0) {
        for (int i = tid; i < ncopy / 4; i += 256)
            reinterpret_cast<f32x4*>(drow)[i] = reinterpret_cast<const f32x4*>(grow)[i];
        for (int i = (ncopy & ~3) + tid; i < ncopy; i += 256) drow[i] = grow[i];
    } else {
        for (int i = tid; i < ncopy; i += 256) drow[i] = grow[i];
    }
}

hipError_t launch_window_edges(const WindowEdgeParams& p, hipStream_t s) {
    if (p.n_windows <= 0) return hipSuccess;
    hipLaunchKernelGGL(window_edges_kernel, dim3((unsigned)p.n_windows), dim3(256), 0, s, p);
    return hipGetLastError();
}

// Host: plain periodic Hann window and the unit circle in 480 steps (double-precision trig).
void build_edge_tables(std::vector<float>& hann, std::vector<float>& trig) {
    hann.resize(FE_NFFT);
    trig.resize(2 * FE_NFFT);
    const double two_pi = 6.283185307179586476925286766559;
    for (int i = 0; i < FE_NFFT; ++i) {
        hann[i] = (float)(0.5 - 0.5 * std::cos(two_pi * i / FE_NFFT));
        trig[2 * i] = (float)std::cos(two_pi * i / FE_NFFT);
        trig[2 * i + 1] = (float)std::sin(two_pi * i / FE_NFFT);
    }
}

// Host: packed A operands and the Hann table.
//   dft float4 index ((w*FE_GROUPS + grp)*4 + mt)*64 + lane, component q:
//     row r = 16 mt + (lane & 15); bin k = 2 r + (w & 1); column j = 4 (4 grp + q) + (lane >> 4) + 1  (1..120)
//     w < 2: cos(2 pi k j/480), w >= 2: sin(2 pi k j/480); the j = 120 column is halved (it pairs with itself)
//   hann[(4 s + g)*2 + {0,1}] = h[j], h[240 - j] for j = 4 s + g + 1
void build_dft_table(std::vector<float>& dft, std::vector<float>& hann) {
    dft.assign(FE_TABLE_FLOATS, 0.f);
    hann.assign(FE_STEPS * 4 * 2, 0.f);
    const double two_pi = 6.283185307179586476925286766559;
    for (int w = 0; w < 4; ++w)
        for (int grp = 0; grp < FE_GROUPS; ++grp)
            for (int mt = 0; mt < 4; ++mt)
                for (int lane = 0; lane < 64; ++lane)
                    for (int q = 0; q < 4; ++q) {
                        const int s = 4 * grp + q;
                        if (s >= FE_STEPS) continue;
                        const int k = 2 * (16 * mt + (lane & 15)) + (w & 1);
                        const int j = 4 * s + (lane >> 4) + 1;
                        const int ph = (int)(((long long)k * j) % FE_NFFT);   // exact angle reduction
                        const double ang = two_pi * ph / FE_NFFT;
                        double v = w < 2 ? std::cos(ang) : std::sin(ang);
                        if (j == 120) v *= 0.5;
                        dft[((((size_t)w * FE_GROUPS + grp) * 4 + mt) * 64 + lane) * 4 + q] = (float)v;
                    }
    for (int s = 0; s < FE_STEPS; ++s)
        for (int g = 0; g < 4; ++g) {
            const int j = 4 * s + g + 1;
            hann[(4 * s + g) * 2 + 0] = (float)(0.5 - 0.5 * std::cos(two_pi * j / FE_NFFT));
            hann[(4 * s + g) * 2 + 1] = (float)(0.5 - 0.5 * std::cos(two_pi * (240 - j) / FE_NFFT));
        }
}

}  // namespace kws
