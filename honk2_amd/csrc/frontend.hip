// MFCC front end for gfx950: wav -> 2*ln(mel power), one workgroup per (clip, 112-frame chunk).
//
// Replaces AudioProcessor.compute_mfccs (reference utils/audio_processor.py:18-30) looped per clip by
// AudioDataLoader.collate_fn (data_loader/audio_data_loader.py:23-35).  Algorithm:
// reflect-pad n_fft/2, frames of 480 at hop 160, periodic Hann, |rFFT|^2, Slaney mel, v>0 ? ln v : v, x2.
//
// MI355X mapping
//   * The windowed 480-point real DFT of every frame is one fp32 GEMM on the matrix cores
//     (v_mfma_f32_16x16x4_f32).  The Hann window is symmetric, so the transform is folded once:
//       Re X[k] = sum_{j=1..240} h[j] cos(2 pi k j/480) (x[j] + x[480-j])      (j = 240 carries weight 1/2)
//       Im X[k] = sum_{j=1..239} h[j] sin(2 pi k j/480) (x[j] - x[480-j])
//     which halves K to 240.  A = packed table (rows = bins 0..127, Re and Im), B = folded samples
//     (columns = 16 frames), built on the fly from the reflect-padded clip staged once in LDS.
//   * wave w owns bins 32w..32w+31 (2 Re + 2 Im row tiles) for all 7 frame tiles: 28 accumulators, so the
//     power |X|^2 = Re^2 + Im^2 is formed in registers with no exchange.
//   * LDS image of the clip uses index i + 2*ceil(i/160): a frame step of 160 samples becomes 162 words, so the
//     16 frames x 2 k-slots a 32-lane half reads hit 32 distinct banks.
//   * the power tile replaces the clip in LDS; the 40x(3..13 non-zero) mel sum, the log and the x2 run per
//     output element and are stored fully coalesced ((T,40) rows are contiguous).
#include "kws_internal.h"

#include <cmath>

namespace kws {

__device__ __forceinline__ int fe_idx(int i) { return i + 2 * ((i + 159) / 160); }

__global__ __launch_bounds__(256, 2) void frontend_kernel(FrontendParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = tid >> 6;
    const int g = lane >> 4;
    const int pcol = lane & 15;
    const int clip = blockIdx.x / p.chunks;
    const int chunk = blockIdx.x - clip * p.chunks;
    const int t0 = chunk * FE_FRAMES;
    const int nfr = min(FE_FRAMES, p.T - t0);
    const int n = p.n_samples;

    // The two workgroups resident on a CU start together and would stay in lockstep (staging and mel phases of
    // one never under the MFMA phase of the other); delay the odd threadgroup slot (HW_REG_HW_ID[19:16]) of the
    // first dispatch wave by about half a workgroup's lifetime.  Later workgroups inherit the offset.  Speed only.
    if (blockIdx.x < 512 && (__builtin_amdgcn_s_getreg(4 | (16 << 6) | (3 << 11)) & 1) != 0)
        for (int i = 0; i < p.stagger_sleeps; ++i) __builtin_amdgcn_s_sleep(127);

    // ---- stage the reflect-padded samples this chunk needs: padded index i = 160*t0 + li
    {
        const float* src = p.wav + (size_t)clip * n;
        const int len = 160 * (nfr - 1) + FE_NFFT;
        for (int li = tid; li < len; li += 256) {
            int s = 160 * t0 + li - FE_NFFT / 2;
            s = s < 0 ? -s : s;
            s = s >= n ? 2 * (n - 1) - s : s;
            lds[fe_idx(li)] = src[s];
        }
    }
    __syncthreads();

    f32x4 acc[4][FE_NT];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int j = 0; j < FE_NT; ++j) acc[m][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int base[FE_NT];
#pragma unroll
    for (int j = 0; j < FE_NT; ++j) base[j] = 162 * min(16 * j + pcol, nfr - 1);

    const f32x4* tab = p.dft + (size_t)w * FE_S4 * 4 * 64 + lane;
    for (int s4 = 0; s4 < FE_S4; ++s4) {
        f32x4 a[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) a[m] = tab[(s4 * 4 + m) * 64];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int kk = 16 * s4 + 4 * q + g;                 // folded pair index, j = kk + 1
            const int foff = 3 + kk + (kk >= 160 ? 2 : 0);       // fe_idx(160 t + 1 + kk) - 162 t
            const int moff = 479 - kk + (kk <= 158 ? 6 : 4);     // fe_idx(160 t + 479 - kk) - 162 t
#pragma unroll
            for (int j = 0; j < FE_NT; ++j) {
                const float x1 = lds[base[j] + foff];
                const float x2 = lds[base[j] + moff];
                const float e = x1 + x2;
                const float o = x1 - x2;
                acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0][q], e, acc[0][j], 0, 0, 0);
                acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1][q], e, acc[1][j], 0, 0, 0);
                acc[2][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2][q], o, acc[2][j], 0, 0, 0);
                acc[3][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3][q], o, acc[3][j], 0, 0, 0);
            }
        }
    }
    __syncthreads();  // every wave is done with the sample image

    // ---- power tile P[bin][frame] into LDS (row stride 116 words: conflict-free for the 4 row groups)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int j = 0; j < FE_NT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float re = acc[mi][j][r], im = acc[2 + mi][j][r];
                lds[(32 * w + 16 * mi + 4 * g + r) * FE_PSTRIDE + 16 * j + pcol] = re * re + im * im;
            }
    __syncthreads();

    // ---- mel + log + "DCT of length 1" (x2); (frame, band) order == memory order of feat
    float* dst = p.feat + ((size_t)clip * p.T + t0) * p.n_mels;
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

size_t frontend_lds_bytes(int T) {
    const int nfr = T < FE_FRAMES ? T : FE_FRAMES;
    const int li = 160 * (nfr - 1) + FE_NFFT - 1;
    size_t sig = (size_t)(li + 2 * ((li + 159) / 160) + 1);
    size_t pw = (size_t)FE_ROWS * FE_PSTRIDE;
    size_t words = sig > pw ? sig : pw;
    return ((words * sizeof(float)) + 15) & ~(size_t)15;
}

hipError_t launch_frontend(const FrontendParams& p, hipStream_t s) {
    static bool attr_done = false;
    const size_t lds = frontend_lds_bytes(p.T);
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)frontend_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)frontend_lds_bytes(FE_FRAMES));
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    if (p.B <= 0) return hipSuccess;
    hipLaunchKernelGGL(frontend_kernel, dim3((unsigned)(p.B * p.chunks)), dim3(256), lds, s, p);
    return hipGetLastError();
}

// Host: packed A operand.  Float4 index ((w*15 + s4)*4 + mt)*64 + lane, component q:
//   bin k = 32w + 16(mt&1) + (lane&15), pair j = 16 s4 + 4 q + (lane>>4) + 1,
//   mt<2: h[j] cos(2 pi k j/480) (x 1/2 at j = 240),  mt>=2: h[j] sin(2 pi k j/480).
void build_dft_table(std::vector<float>& out) {
    out.assign(FE_TABLE_FLOATS, 0.f);
    const double two_pi = 6.283185307179586476925286766559;
    for (int w = 0; w < 4; ++w)
        for (int s4 = 0; s4 < FE_S4; ++s4)
            for (int mt = 0; mt < 4; ++mt)
                for (int lane = 0; lane < 64; ++lane)
                    for (int q = 0; q < 4; ++q) {
                        const int k = 32 * w + 16 * (mt & 1) + (lane & 15);
                        const int j = 16 * s4 + 4 * q + (lane >> 4) + 1;
                        const double h = 0.5 - 0.5 * std::cos(two_pi * j / FE_NFFT);
                        // reduce the angle exactly in integers before calling cos/sin
                        const int ph = (int)(((long long)k * j) % FE_NFFT);
                        const double ang = two_pi * ph / FE_NFFT;
                        double v = mt < 2 ? h * std::cos(ang) : h * std::sin(ang);
                        if (j == FE_NFFT / 2) v = mt < 2 ? 0.5 * v : 0.0;
                        out[((((size_t)w * FE_S4 + s4) * 4 + mt) * 64 + lane) * 4 + q] = (float)v;
                    }
}

}  // namespace kws
