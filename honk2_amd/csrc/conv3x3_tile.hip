// LDS-tiled 3x3 "same" convolution (any power-of-two dilation) over channels-last fp32 activations: the conv_i
// layers of every ResNet that does not fit the fully fused res8 kernel (res15, res26, narrow variants, hey_snips).
// Reference: model/resnet.py:24-31, 44-56 (Conv2d(C, C, 3, padding=d, dilation=d, bias=False) -> ReLU -> (+ prev_x on
// even i) -> BatchNorm(affine=False)).
//
// The generic layer-wise kernels (layerwise*.hip) read every input value nine times (once per tap) from L2 and split
// it into bf16 parts each time; they are bound by that traffic / VALU work, not by the matrix pipe.  Here a workgroup
// copies the cells its 192 output positions need into LDS ONCE, splitting each fp32 value into its three bf16 parts
// on the way in, and all nine taps are served from LDS as ready-made B fragments, as in the fused res8 kernel.
//
//   * Tensors are "CL": [cell][channel padded to 8] fp32 (C = 45 -> 192 B per cell), cells in "layout(d)":
//         [clip][y mod d][x mod d][y / d][x / d]
//     A conv with dilation d only couples positions with equal (y mod d, x mod d), so in layout(d) it is a DENSE 3x3 conv
//     on d*d independent sub-maps of ceil(H/d) x ceil(W/d) cells: the halo of a tile is one row + one cell on each side
//     whatever the dilation.  A layer writes the layout its consumer wants straight from its epilogue and reads the
//     residual in the layout it was written in, so there is no reshuffling pass.  Sub-maps are padded to a common size;
//     padded cells and cells past the tensor are replaced by zeros while staging; taps that leave the sub-map read a
//     shared zero cell.
//   * One workgroup = 192 consecutive positions of the flattened layout (12 position tiles, 3 per wave) plus a halo of
//     Ws + 1 cells on each side; LDS cell = [part 0..2][channel] bf16 (288 B), <= 79 KB, two workgroups per CU, so one
//     stages / stores while the other's waves keep the matrix pipe busy.  All of a thread's staging loads are in flight
//     together; the residual is requested before the k-loop.
//   * K order (tap, 8-channel block), 4 blocks per v_mfma_f32_16x16x32_bf16, six bf16 x bf16 terms per fp32-accurate
//     product (res8_bf16x6.hip); weights pre-split on the host (pack_conv_weights_bf16x6) and read per wave from L2, one
//     k-step ahead; B fragments are three ds_read_b128 per position tile, one tile ahead.
//   * Epilogue: border bias (the previous BatchNorm's shift over the in-bounds taps, layerwise.hip), ReLU, residual,
//     16-byte stores (4 channels) into layout(d_next).
#include "kws_internal.h"

namespace kws {

namespace {
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int TILE_P = T3_TILE_P;   // output positions per workgroup

__device__ __forceinline__ unsigned pack2(float a, float b) {
    const bf16x2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float lo_f(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float hi_f(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }
__device__ __forceinline__ float relu1(float x) {
    const int b = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, b > 0 ? b : 0);
}
// four fp32 values -> three parts of four bf16 each (x = h + m + l to 24 bits)
__device__ __forceinline__ void split4(f32x4 x, u32x2 (&out)[3]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
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

// q / dv for 0 <= q < 2^24 with a precomputed reciprocal (exact after one correction step); rem receives q % dv
__device__ __forceinline__ int fdiv(int q, int dv, float inv, int& rem) {
    int t = (int)((float)q * inv);
    int r = q - t * dv;
    if (r < 0) {
        --t;
        r += dv;
    } else if (r >= dv) {
        ++t;
        r -= dv;
    }
    rem = r;
    return t;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the vector-memory counter, i.e. it would
// make every tile wait for its own epilogue stores and for the weight loads already in flight for the next tile.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// cell index of image position (b, y, x) in layout(2^ld) of an H x W map
__device__ __forceinline__ int layout_cell(int b, int y, int x, int ld, int H, int W) {
    const int d = 1 << ld, Hs = (H + d - 1) >> ld, Ws = (W + d - 1) >> ld;
    const int sub = ((y & (d - 1)) << ld) | (x & (d - 1));
    return (((b << (2 * ld)) + sub) * Hs + (y >> ld)) * Ws + (x >> ld);
}

#define TMF(A_, B_, C_) C_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A_), __builtin_bit_cast(bf16x8, B_), C_, 0, 0, 0)
#define TMF6(A3, B3, C_)       \
    if (TERMS == 6) {          \
        TMF(A3[2], B3[0], C_); \
        TMF(A3[1], B3[1], C_); \
        TMF(A3[0], B3[2], C_); \
    }                          \
    TMF(A3[1], B3[0], C_);     \
    TMF(A3[0], B3[1], C_);     \
    TMF(A3[0], B3[0], C_);
}  // namespace

// NB: 8-channel blocks per cell (3: C <= 24, 6: C <= 48); MT: 16-channel output tiles (2 / 3)
//
// One persistent workgroup of 8 waves per CU, two roles, two LDS tile buffers:
//   waves 0..3 ("matrix waves", one per SIMD): k-loop + epilogue of tile k from buffer k & 1;
//   waves 4..7 ("staging waves", one per SIMD): copy + split tile k + 1 into buffer (k + 1) & 1 meanwhile.
// One barrier per tile hands a filled buffer to the matrix waves and a drained one to the staging waves.  (Two
// independent workgroups per CU run in lock-step -- both stage, then both share the matrix pipe -- and overlap nothing:
// measured, the phases simply added up.)
template <int NB, int MT, int TERMS>
__global__ __launch_bounds__(512, 1) void conv3x3_tile_kernel(TileConvParams p) {
    constexpr int CELL = NB * 48, PART = NB * 16;   // LDS cell: 3 parts x NB*8 channels x 2 B
    constexpr int GCELL = NB * 32;                  // global cell: NB*8 channels x 4 B
    constexpr int STEPS = (9 * NB + 3) / 4;
    constexpr int NP = TERMS == 6 ? 3 : 2;
    constexpr int NQ = NB * 2;         // 4-channel quads per cell
    constexpr int NGRP = 256 / NQ;     // cells copied per pass (21 / 42)
    constexpr int UNR = NB == 6 ? 14 : 7;  // passes in flight together: a whole tile for W <= 40
    extern __shared__ __align__(16) char lds_all[];

    const int tid = threadIdx.x & 255;
    const int role = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8);
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4;
    const int pcol = lane & 15;
    const int ld = p.ld_in, d = 1 << ld, dmask = d - 1;
    const int Hs = p.Hs, Ws = p.Ws;
    const int ncell = TILE_P + 2 * Ws + 2;
    const int zero_off = ncell * CELL;
    const int buf_bytes = (ncell + 1) * CELL;
    const float* const bord = reinterpret_cast<const float*>(lds_all + 2 * buf_bytes);   // (16, NB*8) border-bias table
    // (STEPS_E + 1, 4) x {byte offset of (tap, channel block) relative to the centre cell, 1 << tap (0 for padding blocks)}:
    // lane group g's K block at k-step s is bi = 4 s + g -> tap = bi / NB, channel block bi % NB
    constexpr int STEPS_E = STEPS + (STEPS & 1);
    const u32x2* const steptab = reinterpret_cast<const u32x2*>(lds_all + 2 * buf_bytes + 16 * NB * 8 * 4);
    const float inv_ws = 1.0f / (float)Ws, inv_hs = 1.0f / (float)Hs;
    const int ntiles = (p.total + TILE_P - 1) / TILE_P;
    const int n_my = ((int)blockIdx.x < ntiles) ? (ntiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;

    if (role == 1) {
        // ------------------------------------------------------------ staging waves
        const int qd = tid % NQ, grp = tid / NQ;
        if (tid < CELL / 16) {
            *reinterpret_cast<u32x4*>(lds_all + zero_off + tid * 16) = (u32x4){0u, 0u, 0u, 0u};
            *reinterpret_cast<u32x4*>(lds_all + buf_bytes + zero_off + tid * 16) = (u32x4){0u, 0u, 0u, 0u};
        }
        for (int i = tid; i < 16 * NB * 8; i += 256)
            reinterpret_cast<float*>(lds_all + 2 * buf_bytes)[i] = p.border ? p.border[i] : 0.f;
        if (tid < (STEPS_E + 1) * 4) {
            const int bi = tid;   // = 4 s + g
            const int tap = bi / NB, cblk = bi - tap * NB, ty = tap / 3, tx = tap - 3 * ty;
            u32x2 e;
            e[0] = (unsigned)(((ty - 1) * Ws + (tx - 1)) * CELL + cblk * 16);
            e[1] = tap < 9 ? 1u << tap : 0u;
            reinterpret_cast<u32x2*>(lds_all + 2 * buf_bytes + 16 * NB * 8 * 4)[tid] = e;
        }
        const char* src = reinterpret_cast<const char*>(p.in);
        const bool prof = p.prof && blockIdx.x == 0 && tid == 0;
        long long ps0 = 0, ps1 = 0, ps2 = 0;
        for (int k = 0; k < n_my; ++k) {
            const long long c0 = prof ? clock64() : 0;
            char* lds = lds_all + (k & 1) * buf_bytes;
            const int P0 = ((int)blockIdx.x + k * (int)gridDim.x) * TILE_P;
            // cells [P0 - Ws - 1, P0 + TILE_P + Ws + 1): fp32 quads -> three bf16 parts; padded / outside cells -> zeros
            if (grp < NGRP) {
                for (int i0 = grp; i0 < ncell; i0 += UNR * NGRP) {
                    f32x4 v[UNR];
#pragma unroll
                    for (int u = 0; u < UNR; ++u) {
                        // Cells outside the tensor or in the padding of a sub-map are never tapped (tmask below), so
                        // whatever is copied for them is irrelevant: clamp the address instead of testing.
                        const int q = min(max(P0 - Ws - 1 + i0 + u * NGRP, 0), p.total - 1);
                        v[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
                        if (!(p.debug & 2)) v[u] = *reinterpret_cast<const f32x4*>(src + (size_t)q * GCELL + qd * 16);
                    }
#pragma unroll
                    for (int u = 0; u < UNR; ++u) {
                        const int i = i0 + u * NGRP;
                        if (i < ncell && !(p.debug & 16)) {
                            u32x2 pr[3];
                            split4(v[u], pr);
#pragma unroll
                            for (int pt = 0; pt < 3; ++pt) *reinterpret_cast<u32x2*>(lds + i * CELL + pt * PART + qd * 8) = pr[pt];
                        }
                    }
                }
            }
            const long long c1 = prof ? clock64() : 0;
            lds_barrier();   // buffer k & 1 is full; the matrix waves have drained buffer (k + 1) & 1
            if (prof) {
                const long long c2 = clock64();
                ps0 += c1 - c0;
                ps1 += c2 - c1;
                ps2 += 1;
            }
        }
        if (prof) {
            atomicAdd((unsigned long long*)p.prof + 8, (unsigned long long)ps0);
            atomicAdd((unsigned long long*)p.prof + 9, (unsigned long long)ps1);
            atomicAdd((unsigned long long*)p.prof + 10, (unsigned long long)ps2);
        }
        return;
    }

    // ---------------------------------------------------------------- matrix waves
    const u32x4* A = reinterpret_cast<const u32x4*>(p.apk16) + lane;
    const char* const resp = reinterpret_cast<const char*>(p.res);
    char* const outp = reinterpret_cast<char*>(p.out);


#define TLOADB(BR, ADDR)                                                                              \
    {                                                                                                 \
        const int ad_ = (ADDR);                                                                       \
        _Pragma("unroll") for (int pt = 0; pt < NP; ++pt) BR[pt] = *reinterpret_cast<const u32x4*>(lds + ad_ + pt * PART); \
    }
#define TLOADA(AR, S)                                                                                 \
    {                                                                                                 \
        _Pragma("unroll") for (int m = 0; m < MT; ++m)                                                \
            _Pragma("unroll") for (int pt = 0; pt < NP; ++pt) AR[m][pt] = A[(((S) * MT + m) * 3 + pt) * 64]; \
    }
    u32x4 a0[MT][NP], a1[MT][NP];
    if (n_my > 0) {
        TLOADA(a0, 0)
        TLOADA(a1, 1)
    }

    const bool prof = p.prof && blockIdx.x == 0 && tid == 0;
    long long pm[4] = {0, 0, 0, 0};
    for (int k = 0; k < n_my; ++k) {
        const long long c0 = prof ? clock64() : 0;
        const char* lds = lds_all + (k & 1) * buf_bytes;
        const int P0 = ((int)blockIdx.x + k * (int)gridDim.x) * TILE_P;

        // this lane's three output positions
        int lbase[3], tmask[3], ob[3], oy[3], ox[3];
        bool valid[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int local = (w * 3 + j) * 16 + pcol;
            const int P = P0 + local;
            int xs, ys;
            const int t = fdiv(P, Ws, inv_ws, xs);
            const int m = fdiv(t, Hs, inv_hs, ys);
            const int sub = m & (d * d - 1);
            ob[j] = m >> (2 * ld);
            oy[j] = (ys << ld) + (sub >> ld);
            ox[j] = (xs << ld) + (sub & dmask);
            valid[j] = P < p.total && oy[j] < p.H && ox[j] < p.W;
            // a tap is live when it stays inside the sub-map AND lands on a real image position (sub-maps are padded to a
            // common size; the padding and everything outside the tensor is never read as an operand)
            const int r0 = sub >> ld, c0 = sub & dmask;
            int rowok = 0, colok = 0;
#pragma unroll
            for (int kk = 0; kk < 3; ++kk) {
                const int yy = ys + kk - 1, xx = xs + kk - 1;
                if (yy >= 0 && yy < Hs && (yy << ld) + r0 < p.H) rowok |= 1 << kk;
                if (xx >= 0 && xx < Ws && (xx << ld) + c0 < p.W) colok |= 1 << kk;
            }
            int mk = 0;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
                if ((rowok >> ky) & 1) mk |= colok << (3 * ky);
            tmask[j] = mk;
            lbase[j] = (local + Ws + 1) * CELL;
        }

        // residual and border bias of this lane's outputs: requested now, consumed in the epilogue
        f32x4 resv[3][MT];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int bmask = (oy[j] - d >= 0 ? 1 : 0) | (oy[j] + d < p.H ? 2 : 0) | (ox[j] - d >= 0 ? 4 : 0) |
                              (ox[j] + d < p.W ? 8 : 0);
            const size_t rcell = resp ? (size_t)layout_cell(ob[j], oy[j], ox[j], p.ld_res, p.H, p.W) * GCELL : 0;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int co0 = m * 16 + 4 * g;
                resv[j][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (valid[j] && co0 < NB * 8) {
                    if (resp) resv[j][m] = *reinterpret_cast<const f32x4*>(resp + rcell + co0 * 4);
                    (void)bmask;
                }
            }
        }

        f32x4 acc[3][MT];
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[j][m] = (f32x4){0.f, 0.f, 0.f, 0.f};

        auto b_addr = [&](int j, unsigned tapbit, int off) {
            return (tmask[j] & tapbit) ? lbase[j] + off : zero_off;   // padding blocks have no tap bit
        };
    // One k-step.  A wave has no MFMA queue: whatever it issues between two MFMAs is a bubble in the matrix pipe unless it
    // fits in the ~12 issue cycles a 16-cycle MFMA leaves free.  So the step's other work is hooked in BETWEEN the MFMAs
    // of the accumulation chains and pinned there (sched_barrier after every unit):
    //   chain of channel tile 0 : address + three ds_read_b128 of the NEXT position tile's B fragment
    //   chains of tiles 1.. (last position tile of the step only): reload the weight fragments whose last use has passed
    //   (channel tile m - 1) with step s + 2; the last channel tile's reload follows the step.
    // BX / BY are the two B buffers: BX holds position tile 0 on entry, BY the next step's tile 0 on exit.
#define SB __builtin_amdgcn_sched_barrier(0)
#define LDSV(AD_) (*reinterpret_cast<const u32x4*>(lds + (AD_)))
    // one term of the product for every channel tile: MT independent MFMAs (dependent MFMAs issue at half rate, so a
    // chain on ONE accumulator must never run back to back), a hook after each
#define TROW(AR, BC, J, PA, PB, H0, H1, H2)                                           \
    TMF(AR[0][PA], BC[PB], acc[J][0]); SB; H0; SB;                                    \
    TMF(AR[1][PA], BC[PB], acc[J][1]); SB; H1; SB;                                    \
    if (MT > 2) { TMF(AR[MT - 1][PA], BC[PB], acc[J][MT - 1]); SB; }                  \
    H2; SB;
#define ALD(AR, M_, PT_, SN, DOA) \
    if ((DOA) && (M_) < MT) AR[(M_) < MT ? (M_) : 0][PT_] = A[(((SN) * MT + (M_)) * 3 + (PT_)) * 64]
    // terms, small first: a3b1 a2b2 a1b3 a2b1 a1b2 a1b1 (parts are indexed from 0).  A part's registers are free for
    // the reload as soon as its last term has been issued: part 2 after the first row, part 1 after the fourth.
#define TTILE(J, AR, BC, BN, HA, HB, SN, DOA)                                                                    \
    if (TERMS == 6) {                                                                                            \
        TROW(AR, BC, J, NP - 1, 0, BN[0] = LDSV(nad), BN[1] = LDSV(nad + PART), BN[NP - 1] = LDSV(nad + 2 * PART)) \
        TROW(AR, BC, J, 1, 1, ALD(AR, 0, NP - 1, SN, DOA), ALD(AR, 1, NP - 1, SN, DOA), ALD(AR, 2, NP - 1, SN, DOA)) \
        TROW(AR, BC, J, 0, NP - 1, HA, HB, (void)0)                                                              \
        TROW(AR, BC, J, 1, 0, (void)0, (void)0, (void)0)                                                         \
        TROW(AR, BC, J, 0, 1, ALD(AR, 0, 1, SN, DOA), ALD(AR, 1, 1, SN, DOA), ALD(AR, 2, 1, SN, DOA))            \
        TROW(AR, BC, J, 0, 0, (void)0, (void)0, (void)0)                                                         \
    } else {                                                                                                     \
        TROW(AR, BC, J, 1, 0, BN[0] = LDSV(nad), BN[1] = LDSV(nad + PART), (void)0)                              \
        TROW(AR, BC, J, 0, 1, ALD(AR, 0, 1, SN, DOA), ALD(AR, 1, 1, SN, DOA), ALD(AR, 2, 1, SN, DOA))            \
        TROW(AR, BC, J, 0, 0, HA, HB, (void)0)                                                                   \
    }                                                                                                            \
    ALD(AR, 0, 0, SN, DOA); ALD(AR, 1, 0, SN, DOA); ALD(AR, 2, 0, SN, DOA); SB;
    // `nad` is the LDS address of the B fragment to prefetch next (always one position tile ahead); it is computed one
    // tile before it is used: during tile 0 for tile 2, during tile 1 for the next step's tile 0 (step table entry
    // requested in the same tile), during tile 2 for the next step's tile 1.
#define TSTEP(AR, BX, BY, SNEXT, SN)                                                                             \
    {                                                                                                            \
        u32x2 st_;                                                                                               \
        TTILE(0, AR, BX, BY, nad = b_addr(2, st_c[1], (int)st_c[0]), st_ = steptab[(SNEXT) * 4 + g], SN, false)  \
        TTILE(1, AR, BY, BX, nad = b_addr(0, st_[1], (int)st_[0]), (void)0, SN, false)                           \
        TTILE(2, AR, BX, BY, nad = b_addr(1, st_[1], (int)st_[0]), (void)0, SN, !(p.debug & 8))                  \
        st_c = st_;                                                                                              \
    }

        // The weight fragments form one continuous stream over the tiles: a0 / a1 hold steps s / s + 1, and a step's
        // buffer is reloaded with step s + 2 while its last MFMAs are issued -- wrapping to the NEXT tile's steps 0 / 1
        // at the end, so those loads are older than this tile's epilogue stores (loads and stores retire in order: a
        // wait on a younger load would also wait for the stores).  An odd step count is padded with a skipped step.
        u32x4 bb0[NP], bb1[NP];
        const long long c1 = prof ? clock64() : 0;
        lds_barrier();   // tile k is in buffer k & 1
        const long long c2 = prof ? clock64() : 0;
        u32x2 st_c = steptab[g];
        TLOADB(bb0, b_addr(0, st_c[1], (int)st_c[0]))
        int nad = b_addr(1, st_c[1], (int)st_c[0]);
#pragma unroll 1
        for (int s = (p.debug & 1) ? STEPS_E : 0; s < STEPS_E; s += 2) {
            TSTEP(a0, bb0, bb1, s + 1, (s + 2 == STEPS_E ? 0 : s + 2))
            if ((STEPS & 1) == 0 || s + 1 < STEPS) {
                TSTEP(a1, bb1, bb0, s + 2, (s + 3 >= STEPS_E ? s + 3 - STEPS_E : s + 3))
            } else {
                TLOADA(a1, 1)   // skipped padding step: its buffer still has to receive the next tile's step 1
                SB;
            }
        }

        const long long c3 = prof ? clock64() : 0;
        // epilogue
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            if (!valid[j] || (p.debug & 4)) continue;
            const int bmask = (oy[j] - d >= 0 ? 1 : 0) | (oy[j] + d < p.H ? 2 : 0) | (ox[j] - d >= 0 ? 4 : 0) |
                              (ox[j] + d < p.W ? 8 : 0);
            const size_t ocell = (size_t)layout_cell(ob[j], oy[j], ox[j], p.ld_out, p.H, p.W) * GCELL;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int co0 = m * 16 + 4 * g;
                if (co0 >= NB * 8) continue;
                f32x4 v;
                const f32x4 bb = *reinterpret_cast<const f32x4*>(bord + bmask * (NB * 8) + co0);   // rows padded to NB*8, zeros past Cout
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float x = relu1(acc[j][m][r] + bb[r]);
                    if (resp) x += resv[j][m][r];
                    v[r] = co0 + r < p.Cout ? x : 0.f;   // padded channels hold exact zeros
                }
                *reinterpret_cast<f32x4*>(outp + ocell + co0 * 4) = v;
            }
        }
        if (prof) {
            const long long c4 = clock64();
            pm[0] += c1 - c0;
            pm[1] += c2 - c1;
            pm[2] += c3 - c2;
            pm[3] += c4 - c3;
        }
    }
    if (prof)
        for (int i = 0; i < 4; ++i) atomicAdd((unsigned long long*)p.prof + i, (unsigned long long)pm[i]);
#undef TLOADB
#undef TLOADA
#undef TSTEP
#undef TTILE
#undef TROW
#undef ALD
#undef LDSV
#undef SB
}

// two tile buffers, each (TILE_P + 2 Ws + 2) cells + one zero cell
// + border-bias table + k-step table
size_t conv3x3_tile_lds_bytes(int cp, int Ws) { return (size_t)2 * (T3_TILE_P + 2 * Ws + 3) * cp * 6 + (size_t)16 * cp * 4 + 16 * 4 * 8; }

bool conv3x3_tile_supported(int C, int Cout, int Ws) {
    const int cp = (C + 7) / 8 * 8;
    return C == Cout && (cp == 24 || cp == 48) && conv3x3_tile_lds_bytes(cp, Ws) <= 160 * 1024;
}

template <int NB, int MT>
static hipError_t launch_t3(const TileConvParams& p, hipStream_t s) {
    const unsigned ntiles = (unsigned)((p.total + T3_TILE_P - 1) / T3_TILE_P);
    const unsigned grid = ntiles < (unsigned)p.n_cu ? ntiles : (unsigned)p.n_cu;   // one persistent workgroup per CU
    const size_t lds = conv3x3_tile_lds_bytes(NB * 8, p.Ws);
    auto k6 = conv3x3_tile_kernel<NB, MT, 6>;
    auto k3 = conv3x3_tile_kernel<NB, MT, 3>;
    static bool attr_done = false;   // per instantiation: allow > 64 KB of dynamic LDS
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k6), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(k3), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    if (p.terms == 3)
        hipLaunchKernelGGL(k3, dim3(grid), dim3(512), lds, s, p);
    else
        hipLaunchKernelGGL(k6, dim3(grid), dim3(512), lds, s, p);
    return hipGetLastError();
}

hipError_t launch_conv3x3_tile(const TileConvParams& p, int C, hipStream_t s) {
    if (p.total <= 0) return hipSuccess;
    const int cp = (C + 7) / 8 * 8;
    // positions are decoded with fp32 reciprocals (exact below 2^24) and byte offsets are 32-bit
    if (!conv3x3_tile_supported(C, p.Cout, p.Ws) || (long long)p.total + T3_TILE_P + 2 * p.Ws + 2 >= (1 << 24) ||
        (long long)p.total * cp * 4 >= (1LL << 31))
        return hipErrorInvalidValue;
    if (cp == 48) return launch_t3<6, 3>(p, s);
    return launch_t3<3, 2>(p, s);
}

// ------------------------------------------------------------------------------------------------ fp32 NCHW -> CL
// (B, C, H, W) fp32 -> pooled (window kh x kw, stride = window, floor; mode 0 average, 1 max; 1 x 1 = plain transpose)
// channels-last (B, H/kh, W/kw, cp) fp32, i.e. layout(1).  One thread: one output position x four channels.
__global__ __launch_bounds__(256) void nchw_to_cl_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                         long long total, int C, int H, int W, int Hp, int Wp, int kh,
                                                         int kw, int is_max, int cp) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int nq = cp / 4;
    const int ox = (int)(i % Wp);
    long long t = i / Wp;
    const int oy = (int)(t % Hp);
    t /= Hp;
    const int q = (int)(t % nq);
    const long long b = t / nq;
    f32x4 v;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int c = 4 * q + r;
        float x = 0.f;
        if (c < C) {
            const float* src = in + ((b * C + c) * H + (long long)oy * kh) * W + (long long)ox * kw;
            x = is_max ? -INFINITY : 0.f;
            for (int y = 0; y < kh; ++y)
                for (int xx = 0; xx < kw; ++xx) {
                    const float sv = src[y * W + xx];
                    x = is_max ? fmaxf(x, sv) : x + sv;
                }
            if (!is_max) x = x / (float)(kh * kw);
        }
        v[r] = x;
    }
    *reinterpret_cast<f32x4*>(out + ((b * Hp + oy) * (long long)Wp + ox) * cp + q * 4) = v;
}

hipError_t launch_nchw_to_cl(const float* in, float* out, int B, int C, int H, int W, int kh, int kw, int is_max, int cp,
                             hipStream_t s) {
    const int Hp = H / kh, Wp = W / kw;
    const long long total = (long long)B * (cp / 4) * Hp * Wp;
    if (total <= 0) return hipSuccess;
    hipLaunchKernelGGL(nchw_to_cl_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, in, out, total, C, H, W,
                       Hp, Wp, kh, kw, is_max, cp);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ mean + linear on CL
// ResNet tail (reference model/resnet.py:57-59), last BatchNorm folded in: mean(BN(x)) == BN(mean(x)).
// One workgroup per clip; x is (B, HW, cp) fp32 in layout(1).
__global__ __launch_bounds__(256) void mean_linear_cl_kernel(const float* __restrict__ x, float* __restrict__ logits,
                                                             int C, int cp, int HW, const float* mean, const float* rstd,
                                                             const float* __restrict__ wt, const float* __restrict__ bias,
                                                             int n_out) {
    extern __shared__ float sm[];   // [nsl][cp] partial sums, then [cp] means
    const int b = blockIdx.x;
    const int nsl = 256 / cp > 0 ? 256 / cp : 1;   // cell slices summed in parallel
    const int c = threadIdx.x % cp, sl = threadIdx.x / cp;
    const float* base = x + (size_t)b * HW * cp;
    if (sl < nsl) {
        float s = 0.f;
        for (int i = sl; i < HW; i += nsl) s += base[(size_t)i * cp + c];
        sm[sl * cp + c] = s;
    }
    __syncthreads();
    float* mv = sm + nsl * cp;
    for (int cc = threadIdx.x; cc < C; cc += 256) {
        float s = 0.f;
        for (int k = 0; k < nsl; ++k) s += sm[k * cp + cc];
        float m = s / (float)HW;
        if (mean) m = (m - mean[cc]) * rstd[cc];
        mv[cc] = m;
    }
    __syncthreads();
    for (int o = threadIdx.x; o < n_out; o += 256) {
        float v = 0.f;
        for (int cc = 0; cc < C; ++cc) v = fmaf(wt[o * C + cc], mv[cc], v);
        logits[(size_t)b * n_out + o] = v + bias[o];
    }
}

hipError_t launch_mean_linear_cl(const float* x, float* logits, int B, int C, int cp, int HW, const float* mean,
                                 const float* rstd, const float* w, const float* bias, int n_out, hipStream_t s) {
    if (B <= 0) return hipSuccess;
    if (cp > 256) return hipErrorInvalidValue;
    const int nsl = 256 / cp;
    hipLaunchKernelGGL(mean_linear_cl_kernel, dim3((unsigned)B), dim3(256), (size_t)(nsl + 1) * cp * sizeof(float), s, x,
                       logits, C, cp, HW, mean, rstd, w, bias, n_out);
    return hipGetLastError();
}

}  // namespace kws
