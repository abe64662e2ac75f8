// Second convolution of the two-conv cnn-* models on fp16 tensors (`fp16` dtype; reference model/cnn.py:46-62: Conv2d(C0, C1, (kh, 4), stride 1, no
// padding) + bias -> ReLU, pool_1 the identity): a persistent form of conv_band.hip's idea with COLUMN tiles.
//
// conv_band_kernel (one workgroup per band, position tiles by residue class) is bound by two things (tools/band_phases.py): its k-loop
// reads one 1 KB B fragment from LDS per two MFMAs -- 96 of the CU's 128 B/clk at 76 % of the matrix rate -- and the two workgroups of a CU stage, compute
// and store in phase, so the matrix pipe idles through staging and epilogue (9.2 of 31.7 us).  Here:
//
//   * a position tile = 16 consecutive output ROWS of one output COLUMN x (lane = row).  The B fragment of tile x for tap (dy, dx) and channel quad cq is
//     the 16 cells (row + dy, x + dx) -- a function of the input column c = x + dx alone: one fragment F(c) per (dy, cq) sweep serves kw = 4 taps of up to
//     four tiles.  A wave owns ONE channel tile and ALL Wo <= 14 columns: per sweep Wo + 3 LDS reads and 4 weight fragments feed 4 Wo MFMAs (cnn-trad-pool2: 52
//     MFMAs per 16 LDS reads and 4 KB of weights, where conv_band_kernel's wave reads 8 + 2 KB per 16; every one of the 32 x 13 outputs sits in a tile:
//     1 040 MFMAs per wave and band instead of 1 280);
//   * workgroups of four waves (one per channel tile), TWO per CU, each persistent over (clip, band) units drawn from a device-wide counter,
//     with ONE LDS image: the next unit's 25 input rows are requested with global_load_lds_dwordx4 (memory -> LDS, no registers) once every wave has left the
//     k-loop and land while the epilogue runs -- and while the CU's other workgroup computes.  (The first version, one double-buffered workgroup per CU, ran
//     its 1 120 MFMAs per unit in 30 k cycles with every load ablated: a lone wave per SIMD does not reach the pipe's rate.)  The image is four planes, one per
//     lane group, rows an odd number of 16-byte slots apart: fragment reads without bank conflicts (cols_image_bytes); the DMA fills it in any order it likes,
//     since every lane names its own 16 source bytes;
//   * one set of fragment registers, refilled in place a sweep ahead; four sets of weight fragments, requested two sweeps ahead; a tile's four taps as one chain on its accumulator (cols_kloop).
//
// in: channels-last fp16 cells (B, H, W, Cpi = 64); out: channels-last (B, Ho, Wo, Cpo) fp16 or fp32 cells, exact zeros in the channel padding; weights
// x 2^S as fp16 fragments in sweep order (pack_conv_cols_weights).  Same products in a different summation order than conv_band_kernel (K runs (dy, cq, dx)
// instead of (tap, block)): the results agree to fp32 rounding, not bit for bit; each output's own order does not depend on the batch or on its tile.
#include "kws_internal.h"

namespace kws {

namespace {
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef const u32x4 __attribute__((address_space(3))) * cols_lds_u32x4_ptr;
__device__ __forceinline__ u32x4 cols_lds_read16(int addr) { return *reinterpret_cast<cols_lds_u32x4_ptr>((unsigned)addr); }
#define CMF(A_, B_, C_) C_ = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, A_), __builtin_bit_cast(f16x8, B_), C_, 0, 0, 0)

#ifndef COLS_ABLATE
#define COLS_ABLATE 0      // timing experiments (results wrong): 1 no weight-fragment loads in the sweeps, 2 no LDS fragment reads, 4 no epilogue, 8 no image DMA, 16 no MFMAs
#endif
constexpr int COLS_KW = 4;
constexpr int COLS_ROWS = 16;      // output rows per band = lanes of a position tile
constexpr int COLS_MT = 4;         // channel tiles of the layer = waves of a workgroup (49 - 64 output channels)
constexpr int COLS_NXMAX = 14;     // output columns (tiles per wave)

// The k-loop of a wave that owns NXW output columns (x0 .. x0 + NXW - 1) and MH channel tiles; CB = bytes per cell, NQ = channel quads per cell.
// Sweep t = dy * NQ + cq; fragment c of sweep t is read at fbase + dy * rsb + cq * 16 + c * NQ * 16 (plane g of the image, cols_image_bytes).  ONE set of NC fragment registers: column c's register is
// refilled with the next sweep's column c as soon as this sweep's MFMAs on it are issued (a sweep -- ~900 cycles -- ahead of its use); the weight fragments
// have four sets, requested two sweeps ahead.  In: a0 / a1 = the weights of sweeps 0 / 1 (requested, maybe in flight); out: the same for the next unit.
template <int MH, int NXW, int CB, int NQ>
__device__ __forceinline__ void cols_kloop(const __amdgpu_buffer_rsrc_t ars, const int avoff, const int fbase, const int rsb, const int kh,
                                           f32x4 (&acc)[MH][COLS_NXMAX], u32x4 (&a0)[COLS_KW][MH], u32x4 (&a1)[COLS_KW][MH]) {
    constexpr int NC = NXW + COLS_KW - 1;
    constexpr int FS = NQ * 16;                           // bytes between the fragments of neighbouring columns in an LDS plane
    constexpr int ASWEEP_B = COLS_KW * COLS_MT * 1024;    // bytes of weight fragments per sweep: [dx][4 channel tiles][64 lanes] x 16 B
    const int nsweep = kh * NQ;
    u32x4 f[NC];
    // (sweep numbers are wave-uniform, but hipcc does not always see it: a scalar offset it takes for a vector costs a waterfall loop around every load)
    auto load_a = [&](u32x4 (&ar)[COLS_KW][MH], int i, int so) {      // fragment i = dx * MH + m of the sweep whose fragments start at byte so
        const int dx = i / MH, m = i - dx * MH;
        ar[dx][m] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(ars, avoff + (dx * COLS_MT + m) * 1024, so, 0));
    };
    auto faddr = [&](int t) {
        const int dy = t / NQ, cq = t - dy * NQ;
        return fbase + __builtin_amdgcn_readfirstlane(dy * rsb + cq * 16);
    };
    {
        const int ad = faddr(0);
#pragma unroll
        for (int c = 0; c < NC; ++c) f[c] = cols_lds_read16(ad + c * FS);
    }
    // one sweep: the MFMAs of (ac, f) tile by tile -- a tile's four taps as ONE chain on its accumulator (at the power cap a dependent MFMA is cheaper than an
    // independent one: it takes C from the MFMA in front of it) -- with the next sweep's weights into an and its fragments into f between them
    auto sweep = [&](const u32x4 (&ac)[COLS_KW][MH], u32x4 (&an)[COLS_KW][MH], int tn, int adn) {
        const int so = __builtin_amdgcn_readfirstlane(tn * ASWEEP_B);
#ifndef COLS_APS
#define COLS_APS (COLS_KW * MH)     // weight fragments requested per tile: all of them in front of the first one -- every tile needs all four taps' fragments, so the last one requested is needed as early as the first
#endif
        constexpr int APS = COLS_APS;
#pragma unroll
        for (int x = 0; x < NXW; ++x) {
            if (!(COLS_ABLATE & 1)) {
#pragma unroll
                for (int i = x * APS; i < (x + 1) * APS && i < COLS_KW * MH; ++i) load_a(an, i, so);
            } else if (x == 0) {
#pragma unroll
                for (int i = 0; i < COLS_KW * MH; ++i) an[i / MH][i % MH] = ac[i / MH][i % MH];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < MH; ++m) {
#pragma unroll
                for (int dx = 0; dx < COLS_KW; ++dx) {
                    if (!(COLS_ABLATE & 16)) CMF(ac[dx][m], f[x + dx], acc[m][x]);
                    else acc[m][x][0] += __builtin_bit_cast(float, ac[dx][m][0] ^ f[x + dx][0]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (!(COLS_ABLATE & 2)) f[x] = cols_lds_read16(adn + x * FS);      // (column x is done with: tiles x - 3 .. x were its users)
        }
        if (!(COLS_ABLATE & 2)) {
#pragma unroll
            for (int c = NXW; c < NC; ++c) f[c] = cols_lds_read16(adn + c * FS);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // Four weight sets, requested two sweeps ahead (an L2 round trip under load is longer than one sweep: with two sets, one sweep ahead, the loop spent 18 % of
    // its time waiting for weights, -DCOLS_ABLATE=1).  The sweep count is a multiple of four (conv_cols_supported); past the last sweep the requests wrap to
    // the next unit's first two, which are back in a0 / a1 then.
    u32x4 b0[COLS_KW][MH], b1[COLS_KW][MH];
    auto wrap = [&](int t) { return t < nsweep ? t : t - nsweep; };
    auto fwrap = [&](int t) { return faddr(t < nsweep ? t : nsweep - 1); };
    for (int t = 0; t < nsweep; t += 4) {
        sweep(a0, b0, t + 2, fwrap(t + 1));
        sweep(a1, b1, t + 3, fwrap(t + 2));
        sweep(b0, a0, wrap(t + 4), fwrap(t + 3));
        sweep(b1, a1, wrap(t + 5), fwrap(t + 4));
    }
}
}  // namespace

// LDS image: four PLANES, one per lane group g (a ds_read_b128 is conflict-free iff the 16 rows of a fragment differ mod 16 slots AND the lane groups' slots
// coincide: blocks of one cell 16 bytes apart cost every second LDS cycle, measured); plane g holds, per input row, slot col * NQ + cq = channel block 4 cq + g
// of cell (row, col), rows one spare slot apart (an odd number of slots), planes a multiple of 16 slots apart.
__host__ __device__ inline int cols_row_slots(int W, int Cpi) { return W * ((Cpi + 31) / 32) + 1; }
__host__ __device__ inline int cols_plane_slots(int W, int Cpi, int kh) { return ((COLS_ROWS + kh - 1) * cols_row_slots(W, Cpi) + 15) / 16 * 16; }
__host__ __device__ inline int cols_image_bytes(int W, int Cpi, int kh) { return (4 * cols_plane_slots(W, Cpi, kh) * 16 + 1023) / 1024 * 1024; }
// + one word per 16-byte slot (where the slot's bytes sit in memory relative to the band's first input row) + the next unit's number
static size_t cols_lds_bytes(int W, int Cpi, int kh) { return (size_t)cols_image_bytes(W, Cpi, kh) * 5 / 4 + 16; }

template <int MH, int CPI16>
__global__ __launch_bounds__(256, 2) void conv_cols_kernel(ColsConvParams p) {
    constexpr int CB = CPI16 * 32;             // bytes per fp16 cell
    constexpr int NQ = (CPI16 + 1) / 2;        // channel quads (32 channels = one k-step) per cell
    extern __shared__ __align__(16) char lds[];
    if (range_gate_closed(p.rg)) return;
    if ((unsigned)reinterpret_cast<uintptr_t>(lds) != 0u) __builtin_trap();     // integer LDS addresses below: dynamic LDS at 0 (allow_big_lds_at_base_zero)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, pcol = lane & 15;
    const int rb = p.W * CB;                   // bytes per input row in memory
    const int slr = cols_row_slots(p.W, CPI16 * 16), rsb = slr * 16;      // slots / bytes per row of an LDS plane
    const int ps = cols_plane_slots(p.W, CPI16 * 16, p.kh);
    const int rows_in = COLS_ROWS + p.kh - 1;
    const int npiece = cols_image_bytes(p.W, CPI16 * 16, p.kh) / 1024;

    // (clip, band) units: the first one by workgroup index, the rest drawn from a device-wide counter -- the two workgroups of a CU do not progress at the
    // same rate (arbitration favours the older waves: with four units each the favoured one was done after 70 us and the other ran its last two alone)
    const int nunit = p.B * p.nbands;
    if ((int)blockIdx.x >= nunit) return;      // (the launcher keeps the grid within the units)
    int* const next_slot = reinterpret_cast<int*>(lds + npiece * 1024 * 5 / 4);
    const int u_begin = (int)blockIdx.x;

    auto unit_rows = [&](int u, int& b, int& r0, int& row_lo) {
        b = u / p.nbands;
        const int band = u - b * p.nbands;
        r0 = min(band * COLS_ROWS, p.Ho - COLS_ROWS);    // the last band is moved up to end on the last row; rows the band before it owns are not stored again
        row_lo = band * COLS_ROWS - r0;
    };
    // the image of unit u: this wave's 1 KB pieces w, w + 4, ...; LDS slot s = 64 piece + lane is slot s % ps of plane s / ps.  Where each slot's 16 bytes sit
    // relative to the band's first input row does not depend on the unit: computed once, kept in LDS behind the image (one word per lane and piece, written
    // and read by the same lane: 13 - 20 registers otherwise).  The spare slot of a row and the slots past a plane repeat a neighbour's bytes: finite values
    // that nothing reads.
    int* const dtab = reinterpret_cast<int*>(lds + npiece * 1024);
    for (int pc = w; pc < npiece; pc += 4) {
        const int sl = pc * 64 + lane;
        const int gq = min(sl / ps, 3), sp = sl - (sl / ps) * ps;
        const int rw = min(sp / slr, rows_in - 1), j = min(sp - (sp / slr) * slr, slr - 2);
        const int col = j / NQ, cq = j - col * NQ;
        dtab[pc * 64 + lane] = rw * rb + col * CB + (cq * 4 + gq) * 16;
    }
    auto dma_unit = [&](int u) {
        if ((COLS_ABLATE & 8) && u != u_begin) return;
        int b, r0, lo;
        unit_rows(u, b, r0, lo);
        const char* src = reinterpret_cast<const char*>(p.in) + ((size_t)b * p.H + r0) * rb;
        for (int pc = w; pc < npiece; pc += 4) {
            const __attribute__((address_space(1))) void* gsrc = (const __attribute__((address_space(1))) void*)(src + dtab[pc * 64 + lane]);
            __builtin_amdgcn_global_load_lds(gsrc, (__attribute__((address_space(3))) void*)(lds + pc * 1024), 16, 0, 0);
        }
    };
    dma_unit(u_begin);

    // this wave's columns: the first half of the Wo output columns or the rest
    const int x0 = 0, nxw = p.Wo;              // (every wave takes all columns; an earlier split -- channel halves x column halves -- loaded every weight fragment twice per workgroup: 116 -> 106 us)
    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.apk), 0, p.kh * NQ * (COLS_KW * COLS_MT * 1024), 0x00020000);
    const int avoff = lane * 16 + w * MH * 1024;
    const int fbase = g * ps * 16 + pcol * rsb + x0 * (NQ * 16);
    unsigned amax_u = 0u;
    const float relu_lo = p.relu ? 0.f : -INFINITY;      // ReLU as one maximum, no select
    f32x4 bias_v[MH];      // (loaded here, not in the epilogue: a load there waits for the image pieces requested just before it -- vmcnt counts in order)
#pragma unroll
    for (int m = 0; m < MH; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = (w * MH + m) * 16 + 4 * g + r;
            bias_v[m][r] = co < p.Cout ? p.bias[co] : 0.f;
        }
    u32x4 a0[COLS_KW][MH], a1[COLS_KW][MH];
#pragma unroll
    for (int i = 0; i < COLS_KW * MH; ++i) {
        a0[i / MH][i % MH] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(ars, avoff + ((i / MH) * COLS_MT + i % MH) * 1024, 0, 0));
        a1[i / MH][i % MH] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(ars, avoff + ((i / MH) * COLS_MT + i % MH) * 1024, COLS_KW * COLS_MT * 1024, 0));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

#ifdef COLS_TIMING   // 100 MHz wall-clock stamps of this workgroup's phases, per unit (tools/cols_phases.py)
    unsigned long long cts[5];
    int nts = 0;
#define COLS_TS(i) cts[i] = __builtin_amdgcn_s_memrealtime();
#else
#define COLS_TS(i)
#endif
    for (int u = u_begin, nu = 0; u < nunit; u = nu) {
        COLS_TS(0)
        if (tid == 0) *next_slot = (int)(gridDim.x + atomicAdd(p.queue, 1u));      // the unit after this one: known to everyone behind the k-loop's barrier
        f32x4 acc[MH][COLS_NXMAX];
#pragma unroll
        for (int m = 0; m < MH; ++m)
#pragma unroll
            for (int x = 0; x < COLS_NXMAX; ++x) acc[m][x] = (f32x4){0.f, 0.f, 0.f, 0.f};
#define COLS_CASE(N) case N: cols_kloop<MH, N, CB, NQ>(ars, avoff, fbase, rsb, p.kh, acc, a0, a1); break;
        switch (nxw) {
            COLS_CASE(14) COLS_CASE(13) COLS_CASE(12) COLS_CASE(11) COLS_CASE(10) COLS_CASE(9) COLS_CASE(8) COLS_CASE(7) COLS_CASE(6) COLS_CASE(5) COLS_CASE(4)
            default: break;
        }
#undef COLS_CASE
        COLS_TS(1)
        __syncthreads();      // every wave has issued its last MFMA on this image (the re-reads still in flight are never used)
        COLS_TS(2)
        nu = __builtin_amdgcn_readfirstlane(*next_slot);
        if (nu < nunit) dma_unit(nu);      // ... so the next one may land while the epilogue runs (and the CU's other workgroup computes)

        // ---------------------------------------------------------------- epilogue: bias, ReLU, channels-last cells
        int b, r0, row_lo;
        unit_rows(u, b, r0, row_lo);
        const bool row_ok = pcol >= row_lo && !((COLS_ABLATE & 4) && u != u_begin);
        const size_t obase = ((size_t)b * p.Ho + r0 + pcol) * p.Wo + x0;
#pragma unroll
        for (int m = 0; m < MH; ++m) {
            const int co0 = (w * MH + m) * 16 + 4 * g;
            if (co0 >= p.Cpo) continue;
            const f32x4 bv = bias_v[m];
#pragma unroll
            for (int x = 0; x < COLS_NXMAX; ++x) {
                if (x >= nxw) continue;
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(fmaf(acc[m][x][r], p.inv_scale, bv[r]), relu_lo);      // (channels past Cout: zero weights, zero bias -> the exact zero the padding holds)
                {
                    const float s0 = v[0], s1 = v[1], s2 = v[2], s3 = v[3];      // largest magnitude as a bit pattern (a NaN sorts above every number), two values per v_max3_u32
                    amax_u = max(max(amax_u, __builtin_bit_cast(unsigned, s0) & 0x7fffffffu), __builtin_bit_cast(unsigned, s1) & 0x7fffffffu);
                    amax_u = max(max(amax_u, __builtin_bit_cast(unsigned, s2) & 0x7fffffffu), __builtin_bit_cast(unsigned, s3) & 0x7fffffffu);
                }
                if (!row_ok) continue;
                const size_t o = (obase + x) * p.Cpo + co0;
                if (p.out_f16) {
                    const u32x2 pk = {__builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){v[0], v[1]}, f16x2)),
                                      __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){v[2], v[3]}, f16x2))};
                    *reinterpret_cast<u32x2*>(reinterpret_cast<unsigned short*>(p.out) + o) = pk;
                } else {
                    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + o) = v;
                }
            }
        }
        COLS_TS(3)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();      // everyone's pieces of the next image have landed
#ifdef COLS_TIMING
        COLS_TS(4)
        if (p.dbg_ts && lane == 0 && blockIdx.x < 512 && nts < 8) {
            unsigned long long* o = p.dbg_ts + (((size_t)blockIdx.x * 4 + w) * 8 + nts++) * 8;
            for (int i = 0; i < 5; ++i) o[i] = cts[i];
            o[5] = ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) << 32) | (unsigned)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));
        }
#endif
    }
    if (tid == 0) queue_retire(p.queue);
    range_note(p.rg, __builtin_bit_cast(float, amax_u));
}

// the layer fits: four kernel columns, 64 input channels, a multiple of four sweeps (an even number of kernel rows), four tiles of output channels, 4 .. 14 output columns, at least one full band of rows, both images in LDS
bool conv_cols_supported(int Cin, int Cout, int H, int W, int kh, int kw) {
    const int Cpi = (Cin + 15) / 16 * 16, Ho = H - kh + 1, Wo = W - kw + 1;
    if (kw != COLS_KW || Cpi != 64 || (kh * ((Cpi + 31) / 32)) % 4 || (Cout + 15) / 16 != COLS_MT || Wo < 4 || Wo > COLS_NXMAX || Ho < COLS_ROWS) return false;
    if (cols_lds_bytes(W, Cpi, kh) > 80 * 1024 - 256) return false;      // two workgroups per CU
    // rows computed / rows needed (the last band overlaps the one before it): not below 0.85
    const int nb = (Ho + COLS_ROWS - 1) / COLS_ROWS;
    return (double)Ho / (nb * COLS_ROWS) >= 0.85;
}

// weights (Cout, Cin, kh, 4) x scale -> fp16, [dy][cq][dx][4 channel tiles][lane][8]; lane = (g << 4) | co, slot e of lane group g = channel (4 cq + g) 8 + e
void pack_conv_cols_weights(int Cin, int Cout, int kh, const float* w, float scale, std::vector<unsigned short>& dst) {
    const int Cpi = (Cin + 15) / 16 * 16, nq = (Cpi + 31) / 32, mtt = COLS_MT;
    dst.assign((size_t)kh * nq * COLS_KW * mtt * 64 * 8, 0);
    for (int dy = 0; dy < kh; ++dy)
        for (int cq = 0; cq < nq; ++cq)
            for (int dx = 0; dx < COLS_KW; ++dx)
                for (int m = 0; m < mtt; ++m)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int co = m * 16 + (lane & 15), gq = lane >> 4;
                        for (int e = 0; e < 8; ++e) {
                            const int ci = (4 * cq + gq) * 8 + e;
                            float v = 0.f;
                            if (co < Cout && ci < Cin) v = w[(((size_t)co * Cin + ci) * kh + dy) * COLS_KW + dx] * scale;
                            dst[(((((size_t)dy * nq + cq) * COLS_KW + dx) * mtt + m) * 64 + lane) * 8 + e] = f16_rne_host(v);
                        }
                    }
}

static hipError_t launch_cols_k(const ColsConvParams& p, int n_cu, hipStream_t s) {
    auto k = conv_cols_kernel<1, 4>;
    static DeviceOnce attr_once;
    if (attr_once.first()) {
        hipError_t e = allow_big_lds_at_base_zero(reinterpret_cast<const void*>(k));
        if (e != hipSuccess) return e;
    }
    const int nunit = p.B * p.nbands;
    const size_t lds = cols_lds_bytes(p.W, p.Cpi, p.kh);
    hipLaunchKernelGGL(k, dim3((unsigned)std::min(nunit, 2 * std::max(n_cu, 1))), dim3(256), lds, s, p);
    return hipGetLastError();
}

hipError_t launch_conv_cols(const ColsConvParams& p, int n_cu, hipStream_t s) {
    if (p.B <= 0) return hipSuccess;
    if (!p.queue || p.Cpi != 64 || (p.kh * 2) % 4 || (p.Cout + 15) / 16 != COLS_MT || p.Cpo % 16 || p.Cpo < p.Cout || p.Cpo > 16 * COLS_MT || p.Ho != p.H - p.kh + 1 || p.Wo != p.W - COLS_KW + 1 || p.Wo < 4 || p.Wo > COLS_NXMAX || p.Ho < COLS_ROWS ||
        p.nbands != (p.Ho + COLS_ROWS - 1) / COLS_ROWS || cols_lds_bytes(p.W, p.Cpi, p.kh) > 80 * 1024 - 256)
        return hipErrorInvalidValue;
    return launch_cols_k(p, n_cu, s);
}

}  // namespace kws
