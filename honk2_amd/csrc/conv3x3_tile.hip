// LDS-tiled 3x3 "same" convolution (any power-of-two dilation) over channels-last fp32 activations: the conv_i
// layers of every ResNet that does not fit the fully fused res8 kernel (res15, res26, narrow variants, hey_snips).
// Reference: model/resnet.py:24-31, 44-56 (Conv2d(C, C, 3, padding=d, dilation=d, bias=False) -> ReLU -> (+ prev_x on
// even i) -> BatchNorm(affine=False)).
//
// The generic layer-wise kernels (layerwise*.hip) read every input value nine times (once per tap) from L2 and split
// it into 16-bit parts each time; they are bound by that traffic / VALU work, not by the matrix pipe.  Here a workgroup
// copies the cells its output positions need into LDS ONCE, splitting each fp32 value into its parts on the way in, and
// all nine taps are served from LDS as ready-made B fragments, as in the fused res8 kernel.
//
//   * Tensors are "CL": [cell][channel padded to 8] (C = 45 -> 48 channels), cells in "layout(d)":
//         [clip][y mod d][x mod d][y / d][x / d]
//     fp32 (192 B per cell) in the fp32-accurate and bf16x3 modes; the 16-bit operand type itself (96 B) with single-term
//     products (`bf16` / `fp16` dtypes: staging is then a plain copy).
//     A conv with dilation d only couples positions with equal (y mod d, x mod d), so in layout(d) it is a DENSE 3x3 conv
//     on d*d independent sub-maps of ceil(H/d) x ceil(W/d) cells: the halo of a tile is one row + one cell on each side
//     whatever the dilation.  A layer writes the layout its consumer wants straight from its epilogue and reads the
//     residual in the layout it was written in, so there is no reshuffling pass.  Sub-maps are padded to a common size;
//     taps that leave the sub-map, land in its padding or outside the tensor read a shared zero cell (per-position tap
//     mask), so what staging copies for such cells never matters.  Everything that depends on a position only (tap mask,
//     border class, validity, its cell in the output and residual layouts) comes from a per-layer table of one clip's cells
//     built on the host (build_tile_conv_table).
//   * One workgroup = 320 consecutive positions of the flattened layout (20 position tiles, 5 per wave; 192 = 3 per wave
//     with three-part bf16 cells or <= 24 channels) plus a halo of Ws + 1 cells on each side; LDS cell = [part][channel]
//     (two fp16 parts: 192 B; one part: 96 B; three bf16 parts: 288 B), <= 80 KB, two or three workgroups per CU.
//     Workgroups that share an XCD take a contiguous run of tiles (halo re-reads hit that L2).  All of a thread's
//     staging loads are in flight together; the residual is requested before the k-loop.
//   * K order (tap, 8-channel block), 4 blocks per v_mfma_f32_16x16x32_{f16,bf16}; three fp16 x fp16 terms per
//     fp32-accurate product by default (res8_f16x3.hip), six bf16 terms in the range-free form, three / one in the reduced
//     dtypes; weights pre-split on the host and read per wave from L2, one k-step ahead; B fragments are one ds_read_b128
//     per part and position tile, one tile ahead; the (tap, offset) of a k-step comes from a 512-byte LDS table.
//   * Epilogue: border bias (the previous BatchNorm's shift over the in-bounds taps) from a table in LDS, ReLU, residual,
//     16-byte (fp32) / 8-byte (16-bit) channels-last stores into layout(d_next); the largest stored magnitude feeds the fp16
//     range guard.
// conv0_cl_kernel (conv_0 + ReLU [+ AvgPool] straight into a CL tensor) and mean_linear_cl_kernel (the tail) live here too.
#include "kws_internal.h"

#include <cstdlib>

namespace kws {

namespace {
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef int int2_ __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
// B-fragment reads take 32-bit LDS addresses (res8_f16x3.hip): `lds + offset` costs a vector add of the `lds` symbol per read, a plain LDS
// integer does not.  The kernels below hold only dynamic LDS, which therefore starts at LDS address 0 (checked once per workgroup).
typedef const u32x4 __attribute__((address_space(3))) * t3_lds_u32x4_ptr;
__device__ __forceinline__ u32x4 t3_lds_read16(int addr) { return *reinterpret_cast<t3_lds_u32x4_ptr>((unsigned)addr); }
__device__ __forceinline__ void t3_require_lds_base_zero(const char* lds) {
    if ((unsigned)reinterpret_cast<uintptr_t>(lds) != 0u) __builtin_trap();
}


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

// four fp32 values -> four 16-bit values (round to nearest even) of a 16-bit channels-last tensor: fp16 or bf16
template <bool F16>
__device__ __forceinline__ u32x2 cl_pack4(f32x4 v) {
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    if (F16) return (u32x2){__builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_){v[0], v[1]}, f16x2)),
                            __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_){v[2], v[3]}, f16x2))};
    return (u32x2){pack2(v[0], v[1]), pack2(v[2], v[3])};
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

// two fp16 parts (x = h + l to 22 bits) of four fp32 values: one packed convert + one mixed-precision FMA per value
// (l = fp16(x - float(h)); the difference is exact in fp32, so this equals convert-back / subtract / convert bit for bit:
// tools/split_probe.cpp).  The results go to LDS stores, not straight into an MFMA (no hazard padding needed).
__device__ __forceinline__ void split4_f16(f32x4 x, u32x2 (&out)[2]) {
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

#define TMFH(A_, B_, C_) C_ = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, A_), __builtin_bit_cast(f16x8, B_), C_, 0, 0, 0)
#define TMF(A_, B_, C_) C_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A_), __builtin_bit_cast(bf16x8, B_), C_, 0, 0, 0)
// all terms of one (channel tile, position tile) product, small terms first: fp16 a2b1 a1b2 a1b1; bf16 a3b1 a2b2 a1b3 a2b1
// a1b2 a1b1 (six), the last three (KWS_DTYPE_BF16X3) or the last one (KWS_DTYPE_BF16)
#define TMF6(A3, B3, C_)           \
    if (F16) {                     \
        if (TERMS >= 3) {          \
            TMFH(A3[1], B3[0], C_);\
            TMFH(A3[0], B3[1], C_);\
        }                          \
        TMFH(A3[0], B3[0], C_);    \
    } else {                       \
        if (TERMS == 6) {          \
            TMF(A3[2], B3[0], C_); \
            TMF(A3[1], B3[1], C_); \
            TMF(A3[0], B3[2], C_); \
        }                          \
        if (TERMS >= 3) {          \
            TMF(A3[1], B3[0], C_); \
            TMF(A3[0], B3[1], C_); \
        }                          \
        TMF(A3[0], B3[0], C_);     \
    }
}  // namespace

template <int CELL>
__device__ __forceinline__ void stage_cells_dma(const char* src, int first, int n_cells, int total, char* lds_dst, int w, int lane);   // (defined with the pair kernel below)
#ifndef T3_DMA_STAGE
#define T3_DMA_STAGE 1
#endif

// NB: 8-channel blocks per cell (3: C <= 24, 6: C <= 48); MT: 16-channel output tiles (2 / 3).
// F16 (the default, fp32-accurate): operands are two-part fp16 splits, three terms per product (res8_f16x3.hip; weights
// arrive scaled by 2^S, the accumulator is scaled back in the epilogue's FMA); the LDS cell shrinks to 2 x NB*8 x 2 B, which
// buys 320-position tiles (5 per wave).  Otherwise bf16 parts with TERMS = 6 / 3 / 1 products (reduced-precision dtypes).
//
// TERMS == 1 (the `bf16` / `fp16` dtypes): the activation tensors themselves hold the 16-bit operand type, so staging is a plain
// 16-byte copy into 96-byte LDS cells, the residual and the output move half the bytes, and three workgroups fit a CU.  The layer
// was bound by HBM-side traffic, not by its MFMAs (profiles/r02/v9_res15_bf16_summary.json: 4.3 TB/s, matrix pipe 13 % busy).
template <int NB, int MT, int TERMS, bool F16>
__global__ __launch_bounds__(256, TERMS == 1 ? T3_S16_WGS : (F16 && NB <= 3) ? 3 : 2) void conv3x3_tile_kernel(TileConvParams p) {
    constexpr bool S16 = TERMS == 1;                // 16-bit activation tensors
    constexpr int WP = F16 ? 2 : 3;                 // parts per weight fragment group (as packed on the host)
    constexpr int NP = t3_lds_parts(F16, TERMS);    // parts that take part in the products = parts per LDS cell
    constexpr int CELL = NB * 16 * NP, PART = NB * 16;
    constexpr int GCELL = S16 ? NB * 16 : NB * 32;  // global cell: NB*8 channels x 2 or 4 B
    constexpr int STEPS = (9 * NB + 3) / 4;
    constexpr int TILE_P = t3_tile_positions(NP, NB);
    constexpr int JT = TILE_P / 64;                 // position tiles per wave (5 / 3)
    constexpr int NQ = S16 ? NB : NB * 2;           // 16-byte chunks per global cell (8 channels of 16 bits / 4 of fp32)
    constexpr int NGRP = 256 / NQ;     // cells copied per pass
    constexpr int UNR = S16 ? (NB == 6 ? (TILE_P <= 192 ? 7 : (TILE_P <= 320 ? 10 : 13)) : 4) : (NB == 6 ? 10 : 5);  // staging passes in flight together
    extern __shared__ __align__(16) char lds[];
    if (range_gate_closed(p.rg)) return;
    t3_require_lds_base_zero(lds);
#ifdef T3_TIMING   // 100 MHz wall-clock stamps of this workgroup's phases (tools/t3_phases.py)
    unsigned long long t3ts[12];
#define T3_TS(i) t3ts[i] = __builtin_amdgcn_s_memrealtime();
#else
#define T3_TS(i)
#endif
    T3_TS(0)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4;
    const int pcol = lane & 15;
    const int ld = p.ld_in, d = 1 << ld, dmask = d - 1;
    const int Hs = p.Hs, Ws = p.Ws;
    T3_TS(9)
    // Consecutive tiles share their halo cells: blocks that share an XCD (blockIdx mod 8) take one contiguous run of tiles, so a
    // halo re-read hits that XCD's L2 instead of going out to the fabric again (bijective for any grid size).
    int tile_id = (int)blockIdx.x;
    if (!(KWS_DBG(p.debug & 32))) {
        const int nwg = (int)gridDim.x, xcd = tile_id & 7, q8 = nwg >> 3, r8 = nwg & 7;
        tile_id = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (tile_id >> 3);
    }
    const int P0 = tile_id * TILE_P;
    const int ncell = TILE_P + 2 * Ws + 2;
    // LDS: [zero cell][ncell tile cells][k-step table 512 B][border table].  The zero cell sits at offset 0, so a dead tap's address is just
    // (address & 0) (r3: one vector instruction less per fragment address than "zero_off + (... & mask)")
    const int tab_off = (ncell + 1) * CELL;
    const int border_off = tab_off + 512;
    const float inv_ws = 1.0f / (float)Ws, inv_hs = 1.0f / (float)Hs;

    // ---------------------------------------------------------------- this lane's output positions: one table entry each
    // (requested first, consumed after the staging loads have been issued: one round trip instead of two in a row)
    int lbase[JT], tmask[JT], ocl[JT], rcl[JT];   // tmask: tap mask | border class << 9 | valid << 13; ocl / rcl: output / residual cell
    const float inv_cpc = 1.0f / (float)p.cpc_in;
    (void)inv_ws; (void)inv_hs; (void)dmask; (void)Hs;
    i32x4 pe[JT];
    int pb[JT];
#pragma unroll
    for (int j = 0; j < JT; ++j) {
        const int local = (w * JT + j) * 16 + pcol;
        const int P = min(P0 + local, p.total - 1);
        int q;
        pb[j] = fdiv(P, p.cpc_in, inv_cpc, q);
        pe[j] = *reinterpret_cast<const i32x4*>(p.postab + 4 * q);
        lbase[j] = (local + Ws + 2) * CELL;     // (+ 1: the zero cell in front of the tile)
    }
    T3_TS(8)
    T3_TS(1)
    // ---------------------------------------------------------------- stage cells [P0 - Ws - 1, P0 + TILE_P + Ws + 1)
    {
        const int qd = tid % NQ, grp = tid / NQ;
        if (tid < CELL / 16) *reinterpret_cast<u32x4*>(lds + tid * 16) = (u32x4){0u, 0u, 0u, 0u};
        if (tid < 4 * (STEPS + 2)) {   // k-step table (see the k-loop): entry [s][g], two spare steps for the look-ahead
            const int bi = tid, tap = bi / NB, cblk = bi - tap * NB, ty = tap / 3, tx = tap - 3 * ty;
            // (zero-weight padding blocks, tap >= 9, test bit 31 of the mask word, which is never set: they read the zero cell)
            reinterpret_cast<int2_*>(lds + tab_off)[tid] = (int2_){((ty - 1) * Ws + (tx - 1)) * CELL + cblk * 16, tap < 9 ? tap : 31};
        }
        // the border-bias table (16 border classes x NB*8 channels) goes to LDS: read from global memory in the epilogue, every one of
        // its loads waited -- vmcnt counts loads and stores alike -- for the previous block's output store as well (6.5 of 18 us per tile)
        if (tid < 32 * NB) {
            f32x4 bv = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (p.border) bv = *reinterpret_cast<const f32x4*>(p.border + 4 * tid);
            *reinterpret_cast<f32x4*>(lds + border_off + 16 * tid) = bv;
        }
        if (S16 && T3_DMA_STAGE && !(KWS_DBG(p.debug & 2))) {     // 16-bit tensors: the tile is a flat copy, memory -> LDS without registers (stage_cells_dma)
            stage_cells_dma<CELL>(reinterpret_cast<const char*>(p.in), P0 - Ws - 1, ncell, p.total, lds + CELL, w, lane);
        } else if (grp < NGRP) {
            const char* src = reinterpret_cast<const char*>(p.in);
            for (int i0 = grp; i0 < ncell; i0 += UNR * NGRP) {
                f32x4 v[UNR];
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    // Cells outside the tensor or in the padding of a sub-map are never tapped (tmask above), so whatever
                    // is copied for them is irrelevant: the address is clamped instead of tested.
                    const int q = min(max(P0 - Ws - 1 + i0 + u * NGRP, 0), p.total - 1);
                    v[u] = (KWS_DBG(p.debug & 2)) ? (f32x4){1.f, 2.f, 3.f, 4.f} : *reinterpret_cast<const f32x4*>(src + (size_t)q * GCELL + qd * 16);
                }
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    const int i = i0 + u * NGRP;
                    if (i < ncell) {
                        if (S16) {           // the tensor already holds the operand type: eight channels, one 16-byte store
                            *reinterpret_cast<f32x4*>(lds + (i + 1) * CELL + qd * 16) = v[u];
                        } else if (NP == 2 && !F16) {   // bf16x3: the two leading bf16 parts
                            u32x2 pr[3];
                            split4(v[u], pr);
#pragma unroll
                            for (int pt = 0; pt < 2; ++pt) *reinterpret_cast<u32x2*>(lds + (i + 1) * CELL + pt * PART + qd * 8) = pr[pt];
                        } else if (F16) {
                            u32x2 pr[2];
                            split4_f16(v[u], pr);
#pragma unroll
                            for (int pt = 0; pt < 2; ++pt) *reinterpret_cast<u32x2*>(lds + (i + 1) * CELL + pt * PART + qd * 8) = pr[pt];
                        } else {
                            u32x2 pr[3];
                            split4(v[u], pr);
#pragma unroll
                            for (int pt = 0; pt < 3; ++pt) *reinterpret_cast<u32x2*>(lds + (i + 1) * CELL + pt * PART + qd * 8) = pr[pt];
                        }
                    }
                }
            }
        }
    }
    // the table entries have arrived with the staging loads; residual values of this lane's outputs: requested now, consumed in
    // the epilogue
#pragma unroll
    for (int j = 0; j < JT; ++j) {
        const int local = (w * JT + j) * 16 + pcol;
        tmask[j] = P0 + local < p.total ? pe[j][0] : 0;
        ocl[j] = pb[j] * p.cpc_out + pe[j][1];
        rcl[j] = pb[j] * p.cpc_res + pe[j][2];
    }
    f32x4 resv[S16 ? 1 : JT][S16 ? 1 : MT];
    u32x2 resh[S16 ? JT : 1][S16 ? MT : 1];   // 16-bit tensors: four values in two words
    const char* const resp = reinterpret_cast<const char*>(p.res);
    if (resp && !(KWS_DBG(p.debug & 8))) {
#pragma unroll
        for (int j = 0; j < JT; ++j) {
            const size_t rcell = (size_t)rcl[j] * GCELL;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int co0 = m * 16 + 4 * g;
                const bool live = ((tmask[j] >> 13) & 1) && co0 < NB * 8;
                if (S16) {
                    resh[j][m] = (u32x2){0u, 0u};
                    if (live) resh[j][m] = *reinterpret_cast<const u32x2*>(resp + rcell + co0 * 2);
                } else {
                    resv[j][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    if (live) resv[j][m] = *reinterpret_cast<const f32x4*>(resp + rcell + co0 * 4);
                }
            }
        }
    }

    if (S16 && T3_DMA_STAGE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's part of the tile has landed in LDS
    T3_TS(2)
    __syncthreads();
    T3_TS(3)

    const u32x4* A = reinterpret_cast<const u32x4*>(p.apk16) + lane;

    f32x4 acc[JT][MT];
#pragma unroll
    for (int j = 0; j < JT; ++j)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[j][m] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // This lane group's K block at k-step s: bi = 4 s + g -> (tap, channel block).  (tap, byte offset of that tap and block relative
    // to the centre cell) comes from a small LDS table written once per workgroup -- computing it in the loop cost
    // two integer divisions per step and, with a five-instruction select per fragment address, 146 vector instructions per k-step
    // against 45 MFMAs: the k-loop was bound by its own address arithmetic (11 us per tile where the MFMAs need 5).
    // A tap that leaves the sub-map or the tensor (mask bit clear; always for the zero-weight padding blocks, tap >= 9) reads the
    // shared zero cell at LDS offset 0: address = (lbase + off) & -(bit).
    const int2_* const ktab = reinterpret_cast<const int2_*>(lds + tab_off) + g;
    auto b_addr = [&](int j, int2_ e) {
        const int m = __builtin_amdgcn_sbfe(tmask[j], e[1], 1);      // 0 or -1
        return (lbase[j] + e[0]) & m;
    };
#define TLOADB(BR, ADDR)                                                                              \
    {                                                                                                 \
        const int ad_ = (ADDR);                                                                       \
        _Pragma("unroll") for (int pt = 0; pt < NP; ++pt) BR[pt] = t3_lds_read16(ad_ + pt * PART);                    \
    }
#define TLOADA(AR, S)                                                                                 \
    {                                                                                                 \
        _Pragma("unroll") for (int m = 0; m < MT; ++m)                                                \
            _Pragma("unroll") for (int pt = 0; pt < NP; ++pt) AR[m][pt] = A[(((S) * MT + m) * WP + pt) * 64]; \
    }
    // one k-step: B fragments are fetched one position tile ahead (tile 0 of the next step during the last tile); BX / BY
    // are the two fragment buffers, BX holding tile 0 on entry; JT is odd, so BY holds the next step's tile 0 on exit
#define TSTEP(AR, BX, BY, TAPN)                                                                       \
    {                                                                                                 \
        _Pragma("unroll") for (int j = 0; j < JT; ++j) {                                              \
            u32x4 (&cur_)[NP] = (j & 1) ? BY : BX;                                                    \
            u32x4 (&nxt_)[NP] = (j & 1) ? BX : BY;                                                    \
            TLOADB(nxt_, j + 1 < JT ? b_addr(j + 1, e_c) : b_addr(0, TAPN))                           \
            __builtin_amdgcn_sched_barrier(0);                                                        \
            _Pragma("unroll") for (int m = 0; m < MT; ++m) { TMF6(AR[m], cur_, acc[j][m]) if (TERMS >= 3) __builtin_amdgcn_sched_barrier(0); } \
            __builtin_amdgcn_sched_barrier(0);                                                        \
        }                                                                                             \
    }
    static_assert(JT % 2 == 1, "the fragment buffers swap roles every k-step");

    u32x4 a0[MT][NP], a1[MT][NP], bb0[NP], bb1[NP];
    if (KWS_DBG(p.debug & 16)) TLOADA(a1, 1)   // (timing experiment: the k-loop then re-uses the first two steps' weight fragments)
    int2_ e_c = ktab[0];
    TLOADA(a0, 0)
    TLOADB(bb0, b_addr(0, e_c))
    for (int s = (KWS_DBG(p.debug & 1)) ? STEPS : 0; s < STEPS; s += 2) {
        int2_ e_n = ktab[4 * (s + 1)];
        if (s + 1 < STEPS && !(KWS_DBG(p.debug & 16))) TLOADA(a1, s + 1)
        __builtin_amdgcn_sched_barrier(0);
        TSTEP(a0, bb0, bb1, e_n)
        if (s + 1 >= STEPS) break;
        e_c = e_n;
        e_n = ktab[4 * (s + 2)];
        if (s + 2 < STEPS && !(KWS_DBG(p.debug & 16))) TLOADA(a0, s + 2)
        __builtin_amdgcn_sched_barrier(0);
        TSTEP(a1, bb1, bb0, e_n)
        e_c = e_n;
    }
#undef TLOADB
#undef TLOADA
#undef TSTEP

    T3_TS(4)
    // ---------------------------------------------------------------- epilogue
    char* const outp = reinterpret_cast<char*>(p.out);
    float amax = 0.f;   // largest magnitude stored (fp16 range guard)
#pragma unroll
    for (int j = 0; j < JT; ++j) {
        if (!((tmask[j] >> 13) & 1)) continue;
        const int bmask = (tmask[j] >> 9) & 15;
        const size_t ocell = (size_t)ocl[j] * GCELL;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int co0 = m * 16 + 4 * g;
            if (co0 >= NB * 8) continue;
            f32x4 v, bb = (f32x4){0.f, 0.f, 0.f, 0.f};
            bb = *reinterpret_cast<const f32x4*>(lds + border_off + (bmask * (NB * 8) + co0) * 4);   // rows padded to NB*8, zeros past Cout
            f32x4 rv = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (resp) {
                if (!S16) rv = resv[S16 ? 0 : j][S16 ? 0 : m];
                else if (F16) {
                    // (scalar conversions on purpose: __builtin_bit_cast of a vector ELEMENT reads element 0 with hipcc 7.2)
                    const u32x2 rw = resh[S16 ? j : 0][S16 ? m : 0];
                    rv[0] = (float)__builtin_bit_cast(_Float16, (unsigned short)(rw[0] & 0xffffu));
                    rv[1] = (float)__builtin_bit_cast(_Float16, (unsigned short)(rw[0] >> 16));
                    rv[2] = (float)__builtin_bit_cast(_Float16, (unsigned short)(rw[1] & 0xffffu));
                    rv[3] = (float)__builtin_bit_cast(_Float16, (unsigned short)(rw[1] >> 16));
                } else {
                    const unsigned a = resh[S16 ? j : 0][S16 ? m : 0][0], b = resh[S16 ? j : 0][S16 ? m : 0][1];
                    rv = (f32x4){lo_f(a), hi_f(a), lo_f(b), hi_f(b)};
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                // inv_scale = 2^-S of the fp16 weights (1 for bf16).  No test on the channel index: the padded output channels meet zero weight
                // rows, a zero border bias and the zero padded channels of the residual, so they come out as the exact zeros the next layer's
                // K padding needs (r3: three VALU per value less); rv is 0 without a residual (x + 0 = x exactly, x >= +0)
                v[r] = relu1(fmaf(acc[j][m][r], p.inv_scale, bb[r])) + rv[r];
                amax = fmaxf(amax, fabsf(v[r]));
            }
            if (KWS_DBG(p.debug & 4)) continue;
            if (!S16) *reinterpret_cast<f32x4*>(outp + ocell + co0 * 4) = v;
            else *reinterpret_cast<u32x2*>(outp + ocell + co0 * 2) = cl_pack4<F16>(v);
        }
    }
    range_note(p.rg, amax);
#ifdef T3_TIMING
    T3_TS(5)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    T3_TS(6)
    if (p.dbg_ts && tid == 0 && blockIdx.x < 8192) {     // wave 0 only: 32 words per workgroup
        unsigned long long* o = p.dbg_ts + (size_t)blockIdx.x * 32;
        for (int i = 0; i < 7; ++i) o[i] = t3ts[i];
        o[8] = t3ts[8];
        o[9] = t3ts[9];
        o[7] = ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) << 32) | (unsigned)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));
    }
#endif
}

// one tile buffer (TILE_P + 2 Ws + 2 cells + the zero cell)
size_t conv3x3_tile_lds_bytes(int cp, int Ws, int parts) {
    return (size_t)(t3_tile_positions(parts, cp / 8) + 2 * Ws + 3) * cp * 2 * parts + 512 + 16 * cp * 4;   // + the k-step and border tables
}

bool conv3x3_tile_supported(int C, int Cout, int Ws) {
    const int cp = (C + 7) / 8 * 8;
    return C == Cout && (cp == 24 || cp == 48) && conv3x3_tile_lds_bytes(cp, Ws, 3) <= 160 * 1024 - 512 &&
           conv3x3_tile_lds_bytes(cp, Ws, 2) <= 160 * 1024 - 512;
}

template <int NB, int MT, int TERMS, bool F16>
static hipError_t launch_t3k(const TileConvParams& p, hipStream_t s) {
    constexpr int tile = t3_tile_positions(t3_lds_parts(F16, TERMS), NB);
    const unsigned grid = (unsigned)((p.total + tile - 1) / tile);
    const size_t lds = conv3x3_tile_lds_bytes(NB * 8, p.Ws, t3_lds_parts(F16, TERMS));
    auto k = conv3x3_tile_kernel<NB, MT, TERMS, F16>;
    static DeviceOnce attr_once;   // per instantiation: allow > 64 KB of dynamic LDS
    if (attr_once.first()) {
        hipError_t e = allow_big_lds_at_base_zero(reinterpret_cast<const void*>(k));
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, s, p);
    return hipGetLastError();
}

template <int NB, int MT>
static hipError_t launch_t3(const TileConvParams& p, hipStream_t s) {
    if (p.f16 && p.terms == 1) return launch_t3k<NB, MT, 1, true>(p, s);
    if (p.f16) return launch_t3k<NB, MT, 3, true>(p, s);
    if (p.terms == 1) return launch_t3k<NB, MT, 1, false>(p, s);
    if (p.terms == 3) return launch_t3k<NB, MT, 3, false>(p, s);
    return launch_t3k<NB, MT, 6, false>(p, s);
}

hipError_t launch_conv3x3_tile(const TileConvParams& p, int C, hipStream_t s) {
    if (p.total <= 0) return hipSuccess;
    const int cp = (C + 7) / 8 * 8;
    // positions are decoded with fp32 reciprocals (exact below 2^24) and byte offsets are 32-bit
    if (!conv3x3_tile_supported(C, p.Cout, p.Ws) || (long long)p.total + T3_TILE_P_F16 + 2 * p.Ws + 2 >= (1 << 24) ||
        (long long)p.total * cp * 4 >= (1LL << 31))
        return hipErrorInvalidValue;
    if (cp == 48) return launch_t3<6, 3>(p, s);
    return launch_t3<3, 2>(p, s);
}

// Weights (Cout, Cin, 3, 3) times `scale` -> two fp16 parts in the fragment order of pack_conv_weights_bf16x6 with all channel
// tiles in one group: [k-step][channel tile][part 2][lane][8]; block bi = 4 s + (lane >> 4) = (tap, 8-channel block)
void pack_conv3x3_tile_weights_f16(int C, const float* w, float scale, std::vector<unsigned short>& dst) {
    const int nb = (C + 7) / 8, mt = (C + 15) / 16, steps = (9 * nb + 3) / 4;
    dst.assign((size_t)steps * mt * 2 * 64 * 8, 0);
    for (int s = 0; s < steps; ++s)
        for (int m = 0; m < mt; ++m)
            for (int lane = 0; lane < 64; ++lane) {
                const int co = m * 16 + (lane & 15), bi = 4 * s + (lane >> 4);
                const int tap = bi / nb, cb = bi % nb;
                for (int e = 0; e < 8; ++e) {
                    const int ci = 8 * cb + e;
                    float v = 0.f;
                    if (bi < 9 * nb && co < C && ci < C) v = w[((size_t)co * C + ci) * 9 + tap] * scale;
                    const unsigned short h = f16_rne_host(v);
                    const unsigned short l = f16_rne_host(v - f16_to_f_host(h));
                    dst[((((size_t)s * mt + m) * 2 + 0) * 64 + lane) * 8 + e] = h;
                    dst[((((size_t)s * mt + m) * 2 + 1) * 64 + lane) * 8 + e] = l;
                }
            }
}

// Per-cell metadata of one clip (see TileConvParams::postab): the decode the kernel used to do per position and tile
void build_tile_conv_table(int H, int W, int ld_in, int ld_out, int ld_res, std::vector<int>& tab, int& cpc_in, int& cpc_out, int& cpc_res) {
    auto cells = [&](int ld) { const int d = 1 << ld; return d * d * ((H + d - 1) >> ld) * ((W + d - 1) >> ld); };
    auto cell_of = [&](int y, int x, int ld) {
        const int d = 1 << ld, Hs = (H + d - 1) >> ld, Ws = (W + d - 1) >> ld;
        const int sub = ((y & (d - 1)) << ld) | (x & (d - 1));
        return (sub * Hs + (y >> ld)) * Ws + (x >> ld);
    };
    cpc_in = cells(ld_in);
    cpc_out = cells(ld_out);
    cpc_res = cells(ld_res);
    const int d = 1 << ld_in, Hs = (H + d - 1) >> ld_in, Ws = (W + d - 1) >> ld_in;
    tab.assign((size_t)4 * cpc_in, 0);
    for (int q = 0; q < cpc_in; ++q) {
        const int xs = q % Ws, t = q / Ws, ys = t % Hs, sub = t / Hs;
        const int r0 = sub >> ld_in, c0 = sub & (d - 1);
        const int y = (ys << ld_in) + r0, x = (xs << ld_in) + c0;
        const bool valid = y < H && x < W;
        int mk = 0;
        for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < 3; ++kx) {
                const int yy = ys + ky - 1, xx = xs + kx - 1;
                // a tap is live when it stays inside the sub-map AND lands on a real image position
                if (yy >= 0 && yy < Hs && (yy << ld_in) + r0 < H && xx >= 0 && xx < Ws && (xx << ld_in) + c0 < W) mk |= 1 << (3 * ky + kx);
            }
        const int bmask = (y - d >= 0 ? 1 : 0) | (y + d < H ? 2 : 0) | (x - d >= 0 ? 4 : 0) | (x + d < W ? 8 : 0);
        tab[4 * q + 0] = mk | (bmask << 9) | ((valid ? 1 : 0) << 13);
        tab[4 * q + 1] = valid ? cell_of(y, x, ld_out) : 0;
        tab[4 * q + 2] = valid ? cell_of(y, x, ld_res) : 0;
    }
}

// ------------------------------------------------------------------------------------------------ two layers in one kernel
// conv_i (odd i) -> ReLU -> [BatchNorm folded into conv_{i+1}] -> conv_{i+1} -> ReLU -> + x_{i-1}, for consecutive layers of EQUAL
// dilation on 16-bit tensors (the `bf16` / `fp16` dtypes: res15's pairs (1,2) (5,6) (7,8) (11,12), reference model/resnet.py:20-26,
// 46-55).  Those layers are bound by their bytes, not by their MFMAs (profiles/r02/final_res15_bf16_summary.json: 3.1 TB/s, matrix pipe
// 25 % busy), and the odd layer's output y_i has exactly one consumer.  So a workgroup stages the input tile with a halo of TWO
// rows, computes y_i on the output tile + ONE halo row straight into a second LDS region (same 96-byte cells, same rounding to
// the tensor type as the store it replaces: results are bit-identical to the two-kernel form), and runs conv_{i+1} from there.
// x_{i-1} is at once conv_i's input and conv_{i+1}'s residual, and in layout(d) both layers address it identically: the residual
// is read from the staged tile, not from memory.  Per pair: read 1.5 - 1.9 x + write 1 x instead of (1.26 + 1) + (1.26 + 1 + 1) tensor
// passes, one launch instead of two; the halo row of y_i is computed twice (the matrix pipe has the room).
// JTB: output position tiles per wave (TILE = 64 JTB positions); a wave takes up to 5 tiles of the intermediate map (<= 320 cells).
// (r4) A 16-bit channels-last tile is the SAME bytes in memory and in LDS (consecutive cells, 96 B each): staging it is a flat copy, done by
// global_load_lds_dwordx4 (gfx950: memory -> LDS without passing through registers; lane i of a wave-instruction lands at base + 16 i).  No register
// round trip, no ds_write instructions, and the border tables' loads, the table entries and the first weight fragments share the tile's one memory
// round trip instead of queueing behind it (tools/t3x_phases.py: staging was 5 - 7 us of a triple workgroup's 20 - 26).  Cells outside [0, total) are
// clamped, not tested: they are never tapped.  The caller waits (s_waitcnt vmcnt(0)) in front of its barrier.
#ifndef T3_DMA_STAGE
#define T3_DMA_STAGE 1
#endif
template <int CELL>
__device__ __forceinline__ void stage_cells_dma(const char* src, int first, int n_cells, int total, char* lds_dst, int w, int lane) {
    constexpr int CH = CELL / 16;
    const int nch = n_cells * CH;
    for (int c0 = w * 64; c0 < nch; c0 += 256) {     // (wave-uniform: the LDS base goes through M0)
        const int c = c0 + lane;
        const int i = min(c / CH, n_cells - 1), qd = c - (c / CH) * CH;
        const int q = min(max(first + i, 0), total - 1);
        if (c < nch)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)q * CELL + qd * 16),
                                             (__attribute__((address_space(3))) void*)(lds_dst + c0 * 16), 16, 0, 0);
    }
}

#ifndef PAIR_APF
#define PAIR_APF 3   // k-steps of weight-fragment look-ahead in the pair kernel (A/B knob)
#endif
#ifndef PAIR_NOFENCE     // A/B knob: 1 = no scheduling fences inside the pair / triple k-loops
#define PAIR_NOFENCE 0
#endif
#if PAIR_NOFENCE
#define PAIR_FENCE
#else
#define PAIR_FENCE __builtin_amdgcn_sched_barrier(0);
#endif
#ifndef PAIR_READ_SLOT   // A/B knob: where a tile's look-ahead fragment read sits (see pair_kloop)
#define PAIR_READ_SLOT 0
#endif
#ifndef PAIR_BPF
#define PAIR_BPF 1   // position tiles of B-fragment look-ahead in the pair / triple kernels (A/B knob)
#endif
template <int MT, bool F16>
__device__ __forceinline__ void pair_first_frags(const __amdgpu_buffer_rsrc_t ars, int avoff, u32x4 (&a)[MT]) {
#pragma unroll
    for (int m = 0; m < MT; ++m) a[m] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(ars, avoff, (m * (F16 ? 2 : 3)) * 1024, 0));
}
template <int NB, int MT, bool F16, int JT>
__device__ __forceinline__ void pair_kloop(const char* lds, const int2_* ktab, const __amdgpu_buffer_rsrc_t ars,
                                           const int avoff, const u32x4 (&a_first)[MT], const int (&lbase)[JT], const int (&tmask)[JT],
                                           f32x4 (&acc)[JT][MT]) {
    constexpr int WP = F16 ? 2 : 3;                 // parts per weight fragment group as packed on the host (part 0 is used)
    constexpr int STEPS = (9 * NB + 3) / 4;
    auto b_addr = [&](int j, int2_ e) {
        const int m = __builtin_amdgcn_sbfe(tmask[j], e[1], 1);      // 0 or -1: a dead tap reads the shared zero cell at LDS offset 0
        return (lbase[j] + e[0]) & m;
    };
    auto load_a = [&](u32x4 (&ar)[MT], int s) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
            ar[m] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(ars, avoff, ((s * MT + m) * WP) * 1024, 0));
    };
    // Weight fragments are requested PAIR_APF k-steps ahead: a k-step is only JT x MT MFMAs (144 - 240 clocks), far less than the L2 round
    // trip the fragments take (two layers' weights do not stay in the CU's L1), and with one step of look-ahead every step waited for it.
    constexpr int APF = PAIR_APF;
    constexpr int BPF = PAIR_BPF < JT ? PAIR_BPF : JT;      // B fragments are requested BPF position tiles ahead (ring of BPF + 1 buffers)
    constexpr int TOTAL = STEPS * JT;
    u32x4 a[APF + 1][MT], bb[BPF + 1];
    int2_ e_c = ktab[0];
#pragma unroll
    for (int m = 0; m < MT; ++m) a[0][m] = a_first[m];   // k-step 0's fragments: requested by the caller ahead of its barrier
#pragma unroll
    for (int u = 1; u < APF; ++u) load_a(a[u], u < STEPS ? u : STEPS - 1);
#pragma unroll
    for (int u = 0; u < BPF; ++u) bb[u] = t3_lds_read16(b_addr(u, e_c));
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        int2_ e_n = e_c;
        if (s + 1 < STEPS) e_n = ktab[4 * (s + 1)];
        if (s + APF < STEPS) load_a(a[(s + APF) % (APF + 1)], s + APF);
        PAIR_FENCE
#pragma unroll
        for (int j = 0; j < JT; ++j) {
            const int t = s * JT + j;
#if PAIR_READ_SLOT == 0      // the next fragment's address + read in front of the tile's MFMAs
            if (t + BPF < TOTAL) {
                const int jn = (j + BPF) % JT;
                bb[(t + BPF) % (BPF + 1)] = t3_lds_read16(b_addr(jn, j + BPF < JT ? e_c : e_n));
            }
            PAIR_FENCE
#endif
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                if (s == 0) acc[j][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (F16) TMFH(a[s % (APF + 1)][m], bb[t % (BPF + 1)], acc[j][m]);
                else TMF(a[s % (APF + 1)][m], bb[t % (BPF + 1)], acc[j][m]);
#if PAIR_READ_SLOT == 1      // ... or behind the first one: their issue falls into that MFMA's shadow
                if (m == 0) {
                    PAIR_FENCE
                    if (t + BPF < TOTAL) {
                        const int jn = (j + BPF) % JT;
                        bb[(t + BPF) % (BPF + 1)] = t3_lds_read16(b_addr(jn, j + BPF < JT ? e_c : e_n));
                    }
                    PAIR_FENCE
                }
#endif
            }
            PAIR_FENCE
        }
        e_c = e_n;
    }
}

// JTB: output position tiles per wave (TILE = 64 JTB positions); WGS: workgroups per CU the kernel is sized for -- 2: five intermediate tiles per wave
// (<= 320 cells of y_i); 3: four (<= 256), for tiles + halos small enough that three images fit a CU (the fused kernel is a chain of dependent
// phases -- stage, conv_i, barrier, conv_{i+1}, store -- and wants the occupancy more than the tile size).  (An eight-wave workgroup on the same
// tile was measured slower: twice the weight-fragment traffic.)
template <int NB, int MT, bool F16, int JTB, int WGS>
__global__ __launch_bounds__(256, WGS) void conv3x3_pair_kernel(PairConvParams p) {
    constexpr int CELL = NB * 16;                   // one 16-bit part: LDS cell = global cell
    constexpr int STEPS = (9 * NB + 3) / 4;
    constexpr int NT = 256;
    constexpr int TILE_P = 64 * JTB, JTA = WGS == 3 ? 4 : 5;
    constexpr int NQ = NB, NGRP = NT / NQ;
    extern __shared__ __align__(16) char lds[];
    if (range_gate_closed(p.rg)) return;
    t3_require_lds_base_zero(lds);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, pcol = lane & 15;
    const int Ws = p.Ws, halo = Ws + 1;
    int tile_id = (int)blockIdx.x;
    {   // blocks that share an XCD take a contiguous run of tiles (halo re-reads hit that L2)
        const int nwg = (int)gridDim.x, xcd = tile_id & 7, q8 = nwg >> 3, r8 = nwg & 7;
        tile_id = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (tile_id >> 3);
    }
    const int P0 = tile_id * TILE_P;
    const int n_in = TILE_P + 4 * halo, n_mid = TILE_P + 2 * halo;      // cells of the input tile / of y_i
    // LDS: [zero cell][n_in input cells][n_mid cells of y_i][k-step table 512 B][border tables]
    const int mid_off = (n_in + 1) * CELL;
    const int tab_off = mid_off + n_mid * CELL;
    const int border_off = tab_off + 512;           // [2][16 classes][NB*8] floats
    const float inv_cpc = 1.0f / (float)p.cpc_in;

    // ---- table entries of this lane's positions: outputs (conv_b) and intermediate cells (conv_a); both layers share the layout
    i32x4 peb[JTB];
    int pbb[JTB];
#pragma unroll
    for (int j = 0; j < JTB; ++j) {
        const int P = min(P0 + (w * JTB + j) * 16 + pcol, p.total - 1);
        int q;
        pbb[j] = fdiv(P, p.cpc_in, inv_cpc, q);
        peb[j] = *reinterpret_cast<const i32x4*>(p.postab + 4 * q);
    }
    int pea[JTA];
#pragma unroll
    for (int j = 0; j < JTA; ++j) {
        const int lm = (w * JTA + j) * 16 + pcol;
        const int Pm = P0 - halo + lm;
        int q;
        (void)fdiv(min(max(Pm, 0), p.total - 1), p.cpc_in, inv_cpc, q);
        pea[j] = p.postab[4 * q];
    }
    // ---- stage cells [P0 - 2 halo, P0 + TILE_P + 2 halo), tables
    {
        const int qd = tid % NQ, grp = tid / NQ;
        if (T3_DMA_STAGE && !(KWS_DBG(p.debug & 2))) stage_cells_dma<CELL>(reinterpret_cast<const char*>(p.in), P0 - 2 * halo, n_in, p.total, lds + CELL, w, lane);
        if (tid < CELL / 16) *reinterpret_cast<u32x4*>(lds + tid * 16) = (u32x4){0u, 0u, 0u, 0u};
        if (tid < 4 * (STEPS + 2)) {
            const int bi = tid, tap = bi / NB, cblk = bi - tap * NB, ty = tap / 3, tx = tap - 3 * ty;
            reinterpret_cast<int2_*>(lds + tab_off)[tid] = (int2_){((ty - 1) * Ws + (tx - 1)) * CELL + cblk * 16, tap < 9 ? tap : 31};
        }
        for (int t = tid; t < 2 * 32 * NB; t += NT) {
            const int which = t / (32 * NB), r = t - which * 32 * NB;
            const float* src = which ? p.border_b : p.border_a;
            f32x4 bv = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (src) bv = *reinterpret_cast<const f32x4*>(src + 4 * r);
            *reinterpret_cast<f32x4*>(lds + border_off + 16 * t) = bv;
        }
        if (T3_DMA_STAGE && !(KWS_DBG(p.debug & 2))) {
            // (requested at the top of this block)
        } else if (grp < NGRP) {
            const char* src = reinterpret_cast<const char*>(p.in);
            constexpr int UNR = 10;   // (TILE_P + 4 halo) / NGRP cells per pass: all of a thread's loads in flight together
            for (int i0 = grp; i0 < n_in; i0 += UNR * NGRP) {
                f32x4 v[UNR];
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    const int q = min(max(P0 - 2 * halo + i0 + u * NGRP, 0), p.total - 1);   // (cells that are never tapped: clamped, not tested)
                    v[u] = (KWS_DBG(p.debug & 2)) ? (f32x4){1.f, 2.f, 3.f, 4.f} : *reinterpret_cast<const f32x4*>(src + (size_t)q * CELL + qd * 16);
                }
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    const int i = i0 + u * NGRP;
                    if (i < n_in) *reinterpret_cast<f32x4*>(lds + (i + 1) * CELL + qd * 16) = v[u];
                }
            }
        }
    }
    const __amdgpu_buffer_rsrc_t ars_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.apk_a), 0, STEPS * MT * (F16 ? 2 : 3) * 1024, 0x00020000);
    const __amdgpu_buffer_rsrc_t ars_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.apk_b), 0, STEPS * MT * (F16 ? 2 : 3) * 1024, 0x00020000);
    u32x4 afirst[MT];
    pair_first_frags<MT, F16>(ars_a, lane * 16, afirst);   // in flight across the barrier
    if (T3_DMA_STAGE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's part of the tile has landed in LDS
    __syncthreads();

    const int2_* const ktab = reinterpret_cast<const int2_*>(lds + tab_off) + g;
    float amax = 0.f;
    // ---------------------------------------------------------------- conv_i on the intermediate cells -> LDS
    {
        int lbase[JTA], tmask[JTA];
#pragma unroll
        for (int j = 0; j < JTA; ++j) {
            const int lm = (w * JTA + j) * 16 + pcol;
            const int Pm = P0 - halo + lm;
            lbase[j] = (lm + halo + 1) * CELL;                               // its cell in the input tile (behind the zero cell)
            tmask[j] = (lm < n_mid && Pm >= 0 && Pm < p.total) ? pea[j] : 0;   // outside: no live tap, never read by conv_b either
        }
        f32x4 acc[JTA][MT];
        if (KWS_DBG(p.debug & 1)) {
#pragma unroll
            for (int j = 0; j < JTA; ++j)
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[j][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
        } else
            pair_kloop<NB, MT, F16, JTA>(lds, ktab, ars_a, lane * 16, afirst, lbase, tmask, acc);
        pair_first_frags<MT, F16>(ars_b, lane * 16, afirst);   // conv_{i+1}'s first fragments: in flight across the epilogue and the barrier
#pragma unroll
        for (int j = 0; j < JTA; ++j) {
            const int lm = (w * JTA + j) * 16 + pcol;
            // cells outside the tensor and the padding positions of partial sub-maps are skipped (lane mask, no per-value select): every tap of
            // conv_{i+1} that would land on one is dead in the consumer's own mask and reads the zero cell instead
            if (lm >= n_mid || !((tmask[j] >> 13) & 1)) continue;
            const int bmask = (tmask[j] >> 9) & 15;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int co0 = m * 16 + 4 * g;
                if (co0 >= NB * 8) continue;
                const f32x4 bb = *reinterpret_cast<const f32x4*>(lds + border_off + (bmask * (NB * 8) + co0) * 4);
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] = relu1(fmaf(acc[j][m][r], p.inv_scale_a, bb[r]));     // (padded channels are exact zeros by construction: see conv3x3_tile_kernel)
                    amax = fmaxf(amax, fabsf(v[r]));
                }
                *reinterpret_cast<u32x2*>(lds + mid_off + lm * CELL + co0 * 2) = cl_pack4<F16>(v);
            }
        }
    }
    __syncthreads();
    // ---------------------------------------------------------------- conv_{i+1} from the intermediate cells, + x_{i-1}, -> memory
    {
        int lbase[JTB], tmask[JTB];
#pragma unroll
        for (int j = 0; j < JTB; ++j) {
            const int local = (w * JTB + j) * 16 + pcol;
            lbase[j] = mid_off + (local + halo) * CELL;
            tmask[j] = P0 + local < p.total ? peb[j][0] : 0;
        }
        f32x4 acc[JTB][MT];
        if (KWS_DBG(p.debug & 64)) {
#pragma unroll
            for (int j = 0; j < JTB; ++j)
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[j][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
        } else
            pair_kloop<NB, MT, F16, JTB>(lds, ktab, ars_b, lane * 16, afirst, lbase, tmask, acc);
        char* const outp = reinterpret_cast<char*>(p.out);
#pragma unroll
        for (int j = 0; j < JTB; ++j) {
            if (!((tmask[j] >> 13) & 1)) continue;
            const int local = (w * JTB + j) * 16 + pcol;
            const int bmask = (tmask[j] >> 9) & 15;
            const size_t ocell = (size_t)(pbb[j] * p.cpc_out + peb[j][1]) * CELL;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int co0 = m * 16 + 4 * g;
                if (co0 >= NB * 8) continue;
                const f32x4 bb = *reinterpret_cast<const f32x4*>(lds + border_off + ((16 + bmask) * (NB * 8) + co0) * 4);
                const u32x2 rw = *reinterpret_cast<const u32x2*>(lds + (local + 2 * halo + 1) * CELL + co0 * 2);   // x_{i-1} at this position
                f32x4 rv;
                if (F16) {
                    rv[0] = (float)__builtin_bit_cast(_Float16, (unsigned short)(rw[0] & 0xffffu));
                    rv[1] = (float)__builtin_bit_cast(_Float16, (unsigned short)(rw[0] >> 16));
                    rv[2] = (float)__builtin_bit_cast(_Float16, (unsigned short)(rw[1] & 0xffffu));
                    rv[3] = (float)__builtin_bit_cast(_Float16, (unsigned short)(rw[1] >> 16));
                } else {
                    const unsigned a = rw[0], b = rw[1];
                    rv = (f32x4){lo_f(a), hi_f(a), lo_f(b), hi_f(b)};
                }
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] = relu1(fmaf(acc[j][m][r], p.inv_scale_b, bb[r])) + rv[r];
                    amax = fmaxf(amax, fabsf(v[r]));
                }
                if (!(KWS_DBG(p.debug & 4))) *reinterpret_cast<u32x2*>(outp + ocell + co0 * 2) = cl_pack4<F16>(v);
            }
        }
    }
    range_note(p.rg, amax);
}

size_t conv3x3_pair_lds_bytes(int cp, int Ws, int tile) {
    return (size_t)(2 * tile + 6 * (Ws + 1) + 1) * cp * 2 + 512 + 2 * 16 * cp * 4;
}
// is the pair kernel available for sub-maps Ws cells wide?  (returns the largest tile it would use, 0 = no)
int conv3x3_pair_tile(int C, int Ws) {
    const int cp = (C + 7) / 8 * 8;
    if (cp != 48) return 0;
    if (256 + 2 * (Ws + 1) <= 320 && conv3x3_pair_lds_bytes(cp, Ws, 256) <= 80 * 1024 - 256) return 256;
    if (192 + 2 * (Ws + 1) <= 320 && conv3x3_pair_lds_bytes(cp, Ws, 192) <= 80 * 1024 - 256) return 192;
    return 0;
}

template <bool F16, int JTB, int WGS>
static hipError_t launch_pair_k(const PairConvParams& p, hipStream_t s) {
    constexpr int tile = 64 * JTB;
    const unsigned grid = (unsigned)((p.total + tile - 1) / tile);
    const size_t lds = conv3x3_pair_lds_bytes(48, p.Ws, tile);
    auto k = conv3x3_pair_kernel<6, 3, F16, JTB, WGS>;
    static DeviceOnce attr_once;
    if (attr_once.first()) {
        hipError_t e = allow_big_lds_at_base_zero(reinterpret_cast<const void*>(k));
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, s, p);
    return hipGetLastError();
}

hipError_t launch_conv3x3_pair(const PairConvParams& p, int C, hipStream_t s) {
    if (p.total <= 0) return hipSuccess;
    const int tile = conv3x3_pair_tile(C, p.Ws);
    if (!tile || (long long)p.total + 384 + 4 * p.Ws + 4 >= (1 << 24) || (long long)p.total * 96 >= (1LL << 31)) return hipErrorInvalidValue;
    static const int wgs3_env = experiment_int("KWS_T3_PAIR_WGS3", 1);
    // three workgroups per CU on 192-position tiles when three images fit (53 KB each) and y_i's tile fits four tiles per wave
    if (wgs3_env && 192 + 2 * (p.Ws + 1) <= 256 && conv3x3_pair_lds_bytes(48, p.Ws, 192) <= 53 * 1024)
        return p.f16 ? launch_pair_k<true, 3, 3>(p, s) : launch_pair_k<false, 3, 3>(p, s);
    if (tile == 256) return p.f16 ? launch_pair_k<true, 4, 2>(p, s) : launch_pair_k<false, 4, 2>(p, s);
    return p.f16 ? launch_pair_k<true, 3, 2>(p, s) : launch_pair_k<false, 3, 2>(p, s);
}

// ------------------------------------------------------------------------------------------------ three layers in one kernel
// A run of THREE layers of equal dilation (res15: (4,5,6) d = 2, (7,8,9) d = 4, (10,11,12) d = 8; every run of hey_snips) on 16-bit
// tensors, reference model/resnet.py:20-26, 46-55.  The pair kernel above, one layer deeper: the input tile is staged with a halo of
// THREE rows, the first layer is computed on the output tile + two halo rows into a second LDS region, the second on the tile + one
// row into a third, the last on the tile.  Every intermediate value is rounded to the tensor type exactly as the store it replaces,
// so the result is bit-identical to one kernel per layer.  What the residual stream needs, by the parity of the first layer a:
//   a even:  x_a = relu(conv_a(in)) + x_{a-2}: the residual comes from MEMORY (tile + two halo rows, in the layout it was written in,
//            cells through the table); x_a is conv_{a+1}'s input and conv_{a+2}'s residual and never leaves the CU; the map of
//            y_{a+1} reuses the input tile's LDS (dead after the first k-loop).
//   a odd:   x_{a+1} = relu(conv_{a+1}(y_a)) + in: the residual is the staged tile; x_{a+1} feeds conv_{a+2} from LDS AND is stored
//            (tile positions only, in the input's own layout: cell = flattened position) as the next even layer's residual.
// Per run of res15: one staged read of 1.2 - 1.7 x the tile (+ 1.1 - 1.4 x of residual, a even) and one or two stored tensors, where a
// single layer + a pair move 5 - 6 tensor passes; the first / second layer's halo rows are computed three / two times (the matrix pipe is
// a quarter busy in these layers).
template <int JT>
__device__ __forceinline__ void triple_decode(int P_first, const TripleConvParams& p, float inv_cpc, int (&q)[JT], int (&pb)[JT]) {
    // cell-in-clip and clip of positions P_first, P_first + 16, ...: one division, then steps of 16 (cpc_in >= 16: launcher).  Halo
    // positions in front of the tensor are negative: whole clips are added first (their entries are never used, only in range)
    int r;
    int b = fdiv(P_first + p.wrap_clips * p.cpc_in, p.cpc_in, inv_cpc, r) - p.wrap_clips;
#pragma unroll
    for (int j = 0; j < JT; ++j) {
        q[j] = r;
        pb[j] = b;
        r += 16;
        if (r >= p.cpc_in) {
            r -= p.cpc_in;
            ++b;
        }
    }
}
template <bool F16>
__device__ __forceinline__ f32x4 cl_unpack4(u32x2 rw) {
    f32x4 rv;
    if (F16) {   // (scalar conversions on purpose: see conv3x3_tile_kernel)
        rv[0] = (float)__builtin_bit_cast(_Float16, (unsigned short)(rw[0] & 0xffffu));
        rv[1] = (float)__builtin_bit_cast(_Float16, (unsigned short)(rw[0] >> 16));
        rv[2] = (float)__builtin_bit_cast(_Float16, (unsigned short)(rw[1] & 0xffffu));
        rv[3] = (float)__builtin_bit_cast(_Float16, (unsigned short)(rw[1] >> 16));
    } else {
        rv = (f32x4){lo_f(rw[0]), hi_f(rw[0]), lo_f(rw[1]), hi_f(rw[1])};
    }
    return rv;
}

// JT1 / JT2 / JTB: position tiles per wave in the three phases (64 JT1 >= TILE + 4 halo, 64 JT2 >= TILE + 2 halo, TILE = 64 JTB)
template <bool F16, int JT1, int JT2, int JTB, bool EVEN, int WGS>
__global__ __launch_bounds__(256, WGS) void conv3x3_triple_kernel(TripleConvParams p) {
    constexpr int NB = 6, MT = 3, CELL = NB * 16, STEPS = (9 * NB + 3) / 4, NT = 256;
    constexpr int TILE_P = 64 * JTB;
    constexpr int NQ = NB, NGRP = NT / NQ;
    extern __shared__ __align__(16) char lds[];
    if (range_gate_closed(p.rg)) return;
    t3_require_lds_base_zero(lds);
#ifdef T3_TIMING   // 100 MHz wall-clock stamps of this workgroup's phases (tools/t3_phases.py --triple)
    unsigned long long t3ts[12];
#define T3X_TS(i) t3ts[i] = __builtin_amdgcn_s_memrealtime();
#else
#define T3X_TS(i)
#endif
    T3X_TS(0)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, pcol = lane & 15;
    const int Ws = p.Ws, halo = Ws + 1;
    int tile_id = (int)blockIdx.x;
    {   // blocks that share an XCD take a contiguous run of tiles (halo re-reads hit that L2)
        const int nwg = (int)gridDim.x, xcd = tile_id & 7, q8 = nwg >> 3, r8 = nwg & 7;
        tile_id = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (tile_id >> 3);
    }
    const int P0 = tile_id * TILE_P;
    // cells of the input tile [P0 - 3 halo, ..), of the first layer's map [P0 - 2 halo, ..), of the second layer's [P0 - halo, ..)
    const int n_in = TILE_P + 6 * halo, n_m1 = TILE_P + 4 * halo, n_m2 = TILE_P + 2 * halo;
    // LDS: [zero cell][input tile][map 1][map 2 -- a even: over the input tile][k-step table 512 B][3 border tables]
    const int mid1_off = (n_in + 1) * CELL;
    const int mid2_off = EVEN ? CELL : mid1_off + n_m1 * CELL;
    const int tab_off = EVEN ? mid1_off + n_m1 * CELL : mid2_off + n_m2 * CELL;
    const int border_off = tab_off + 512;           // [3][16 classes][NB*8] floats
    const float inv_cpc = 1.0f / (float)p.cpc_in;

    // ---- table entries of this lane's positions in the three phases (requested first, consumed after the staging loads are out)
    // (the entries of the second and third phase are requested behind the k-loop in front of them: they would only occupy registers until then)
    int q1[JT1], b1[JT1];
    triple_decode<JT1>(P0 - 2 * halo + w * JT1 * 16 + pcol, p, inv_cpc, q1, b1);
    int pe1[JT1], pr1[EVEN ? JT1 : 1], pe2[JT2];
    int pe3m[JTB], pe3o[JTB], b3[JTB];
#pragma unroll
    for (int j = 0; j < JT1; ++j) {
        if (EVEN) {
            const i32x4 e = *reinterpret_cast<const i32x4*>(p.postab + 4 * q1[j]);
            pe1[j] = e[0];
            pr1[EVEN ? j : 0] = e[2];
        } else {
            pe1[j] = p.postab[4 * q1[j]];
        }
    }

    // ---- stage cells [P0 - 3 halo, P0 + TILE_P + 3 halo), tables
    {
        const int qd = tid % NQ, grp = tid / NQ;
        if (T3_DMA_STAGE && !(KWS_DBG(p.debug & 2))) stage_cells_dma<CELL>(reinterpret_cast<const char*>(p.in), P0 - 3 * halo, n_in, p.total, lds + CELL, w, lane);
        if (tid < CELL / 16) *reinterpret_cast<u32x4*>(lds + tid * 16) = (u32x4){0u, 0u, 0u, 0u};
        if (tid < 4 * (STEPS + 2)) {
            const int bi = tid, tap = bi / NB, cblk = bi - tap * NB, ty = tap / 3, tx = tap - 3 * ty;
            reinterpret_cast<int2_*>(lds + tab_off)[tid] = (int2_){((ty - 1) * Ws + (tx - 1)) * CELL + cblk * 16, tap < 9 ? tap : 31};
        }
        for (int t = tid; t < 3 * 32 * NB; t += NT) {
            const int which = t / (32 * NB), r = t - which * 32 * NB;
            const float* src = p.border[which];
            f32x4 bv = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (src) bv = *reinterpret_cast<const f32x4*>(src + 4 * r);
            *reinterpret_cast<f32x4*>(lds + border_off + 16 * t) = bv;
        }
        if (T3_DMA_STAGE && !(KWS_DBG(p.debug & 2))) {
            // (requested at the top of this block)
        } else if (grp < NGRP) {
            const char* src = reinterpret_cast<const char*>(p.in);
            constexpr int UNR = 10;   // all of a thread's loads in flight together
            for (int i0 = grp; i0 < n_in; i0 += UNR * NGRP) {
                f32x4 v[UNR];
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    const int q = min(max(P0 - 3 * halo + i0 + u * NGRP, 0), p.total - 1);   // (cells that are never tapped: clamped, not tested)
                    v[u] = (KWS_DBG(p.debug & 2)) ? (f32x4){1.f, 2.f, 3.f, 4.f} : *reinterpret_cast<const f32x4*>(src + (size_t)q * CELL + qd * 16);
                }
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    const int i = i0 + u * NGRP;
                    if (i < n_in) *reinterpret_cast<f32x4*>(lds + (i + 1) * CELL + qd * 16) = v[u];
                }
            }
        }
    }
    const int abytes = STEPS * MT * (F16 ? 2 : 3) * 1024;
    const __amdgpu_buffer_rsrc_t ars_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.apk[0]), 0, abytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ars_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.apk[1]), 0, abytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ars_c = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.apk[2]), 0, abytes, 0x00020000);
    u32x4 afirst[MT];
    pair_first_frags<MT, F16>(ars_a, lane * 16, afirst);   // in flight across the barrier

    // ---- a even: the first layer's residual x_{a-2} at this lane's cells of map 1 (requested now, consumed after the first k-loop)
    int tmask1[JT1];
#pragma unroll
    for (int j = 0; j < JT1; ++j) {
        const int lm = (w * JT1 + j) * 16 + pcol;
        const int Pm = P0 - 2 * halo + lm;
        tmask1[j] = (lm < n_m1 && Pm >= 0 && Pm < p.total) ? pe1[j] : 0;   // outside: no live tap, no store, never read by the next layer either
    }
    u32x2 resh[EVEN ? JT1 : 1][MT];
    if (EVEN) {
        const char* const resp = reinterpret_cast<const char*>(p.res);
#pragma unroll
        for (int j = 0; j < JT1; ++j) {
            const size_t rcell = (size_t)(b1[j] * p.cpc_res + pr1[EVEN ? j : 0]) * CELL;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                resh[EVEN ? j : 0][m] = (u32x2){0u, 0u};
                if ((tmask1[j] >> 13) & 1) resh[EVEN ? j : 0][m] = *reinterpret_cast<const u32x2*>(resp + rcell + (m * 16 + 4 * g) * 2);
            }
        }
    }
    if (T3_DMA_STAGE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's part of the tile has landed in LDS
    T3X_TS(1)
    __syncthreads();
    T3X_TS(2)

    const int2_* const ktab = reinterpret_cast<const int2_*>(lds + tab_off) + g;
    float amax = 0.f;
    // ---------------------------------------------------------------- layer a on map 1's cells -> LDS
    {
        int lbase[JT1];
#pragma unroll
        for (int j = 0; j < JT1; ++j) lbase[j] = ((w * JT1 + j) * 16 + pcol + halo + 1) * CELL;   // its cell in the input tile (behind the zero cell)
        f32x4 acc[JT1][MT];
        if (KWS_DBG(p.debug & 1)) {
#pragma unroll
            for (int j = 0; j < JT1; ++j)
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[j][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
        } else
            pair_kloop<NB, MT, F16, JT1>(lds, ktab, ars_a, lane * 16, afirst, lbase, tmask1, acc);
        T3X_TS(3)
        pair_first_frags<MT, F16>(ars_b, lane * 16, afirst);   // the next layer's first fragments: in flight across the epilogue and the barrier
        {
            int q2[JT2], b2[JT2];
            triple_decode<JT2>(P0 - halo + w * JT2 * 16 + pcol, p, inv_cpc, q2, b2);
#pragma unroll
            for (int j = 0; j < JT2; ++j) pe2[j] = p.postab[4 * q2[j]];
        }
#pragma unroll
        for (int j = 0; j < JT1; ++j) {
            if (!((tmask1[j] >> 13) & 1)) continue;     // (outside the tensor / padding of a partial sub-map: every tap that would land here is dead in the consumer's mask)
            const int lm = (w * JT1 + j) * 16 + pcol;
            const int bmask = (tmask1[j] >> 9) & 15;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int co0 = m * 16 + 4 * g;
                const f32x4 bb = *reinterpret_cast<const f32x4*>(lds + border_off + (bmask * (NB * 8) + co0) * 4);
                f32x4 rv = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (EVEN) rv = cl_unpack4<F16>(resh[EVEN ? j : 0][m]);
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] = relu1(fmaf(acc[j][m][r], p.inv_scale[0], bb[r])) + rv[r];   // (x + 0 = x exactly for x >= +0: the same bits as the store without a residual)
                    amax = fmaxf(amax, fabsf(v[r]));
                }
                *reinterpret_cast<u32x2*>(lds + mid1_off + lm * CELL + co0 * 2) = cl_pack4<F16>(v);
            }
        }
    }
    T3X_TS(4)
    __syncthreads();
    T3X_TS(5)
    // ---------------------------------------------------------------- layer a + 1 on map 2's cells -> LDS (a odd: + the staged input, and -> out2)
    {
        int lbase[JT2], tmask[JT2];
#pragma unroll
        for (int j = 0; j < JT2; ++j) {
            const int lm = (w * JT2 + j) * 16 + pcol;
            const int Pm = P0 - halo + lm;
            lbase[j] = mid1_off + (lm + halo) * CELL;
            tmask[j] = (lm < n_m2 && Pm >= 0 && Pm < p.total) ? pe2[j] : 0;
        }
        f32x4 acc[JT2][MT];
        if (KWS_DBG(p.debug & 64)) {
#pragma unroll
            for (int j = 0; j < JT2; ++j)
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[j][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
        } else
            pair_kloop<NB, MT, F16, JT2>(lds, ktab, ars_b, lane * 16, afirst, lbase, tmask, acc);
        T3X_TS(6)
        pair_first_frags<MT, F16>(ars_c, lane * 16, afirst);
        {
            int q3[JTB];
            triple_decode<JTB>(P0 + w * JTB * 16 + pcol, p, inv_cpc, q3, b3);
#pragma unroll
            for (int j = 0; j < JTB; ++j) {
                const int2_ e = *reinterpret_cast<const int2_*>(p.postab + 4 * q3[j]);
                pe3m[j] = e[0];
                pe3o[j] = e[1];
            }
        }
        char* const out2p = reinterpret_cast<char*>(p.out2);
#pragma unroll
        for (int j = 0; j < JT2; ++j) {
            if (!((tmask[j] >> 13) & 1)) continue;
            const int lm = (w * JT2 + j) * 16 + pcol;
            const int bmask = (tmask[j] >> 9) & 15;
            const bool own = lm >= halo && lm < halo + TILE_P;        // a position of this workgroup's output tile
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int co0 = m * 16 + 4 * g;
                const f32x4 bb = *reinterpret_cast<const f32x4*>(lds + border_off + ((16 + bmask) * (NB * 8) + co0) * 4);
                f32x4 rv = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (!EVEN) rv = cl_unpack4<F16>(*reinterpret_cast<const u32x2*>(lds + (lm + 2 * halo + 1) * CELL + co0 * 2));   // x_{a-1} at this position
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] = relu1(fmaf(acc[j][m][r], p.inv_scale[1], bb[r])) + rv[r];
                    amax = fmaxf(amax, fabsf(v[r]));
                }
                const u32x2 pk = cl_pack4<F16>(v);
                *reinterpret_cast<u32x2*>(lds + mid2_off + lm * CELL + co0 * 2) = pk;
                if (!EVEN && own && !(KWS_DBG(p.debug & 4))) *reinterpret_cast<u32x2*>(out2p + (size_t)(P0 - halo + lm) * CELL + co0 * 2) = pk;
            }
        }
    }
    T3X_TS(7)
    __syncthreads();
    T3X_TS(8)
    // ---------------------------------------------------------------- layer a + 2 from map 2 (a even: + x_a from map 1) -> memory
    {
        int lbase[JTB], tmask[JTB];
#pragma unroll
        for (int j = 0; j < JTB; ++j) {
            const int local = (w * JTB + j) * 16 + pcol;
            lbase[j] = mid2_off + (local + halo) * CELL;
            tmask[j] = P0 + local < p.total ? pe3m[j] : 0;
        }
        f32x4 acc[JTB][MT];
        if (KWS_DBG(p.debug & 128)) {
#pragma unroll
            for (int j = 0; j < JTB; ++j)
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[j][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
        } else
            pair_kloop<NB, MT, F16, JTB>(lds, ktab, ars_c, lane * 16, afirst, lbase, tmask, acc);
        T3X_TS(9)
        char* const outp = reinterpret_cast<char*>(p.out);
#pragma unroll
        for (int j = 0; j < JTB; ++j) {
            if (!((tmask[j] >> 13) & 1)) continue;
            const int local = (w * JTB + j) * 16 + pcol;
            const int bmask = (tmask[j] >> 9) & 15;
            const size_t ocell = (size_t)(b3[j] * p.cpc_out + pe3o[j]) * CELL;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int co0 = m * 16 + 4 * g;
                const f32x4 bb = *reinterpret_cast<const f32x4*>(lds + border_off + ((32 + bmask) * (NB * 8) + co0) * 4);
                f32x4 rv = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (EVEN) rv = cl_unpack4<F16>(*reinterpret_cast<const u32x2*>(lds + mid1_off + (local + 2 * halo) * CELL + co0 * 2));   // x_a at this position
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] = relu1(fmaf(acc[j][m][r], p.inv_scale[2], bb[r])) + rv[r];
                    amax = fmaxf(amax, fabsf(v[r]));
                }
                if (!(KWS_DBG(p.debug & 4))) *reinterpret_cast<u32x2*>(outp + ocell + co0 * 2) = cl_pack4<F16>(v);
            }
        }
    }
    range_note(p.rg, amax);
#ifdef T3_TIMING
    T3X_TS(10)
    if (p.dbg_ts && lane == 0 && blockIdx.x < 8192) {     // 4 waves x 12 words per workgroup
        unsigned long long* o = p.dbg_ts + ((size_t)blockIdx.x * 4 + w) * 12;
        for (int i = 0; i < 11; ++i) o[i] = t3ts[i];
        o[11] = ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) << 32) | (unsigned)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));
    }
#endif
}

// LDS bytes of the triple kernel for sub-maps Ws cells wide and TILE output positions per workgroup
static size_t conv3x3_triple_lds_bytes(int Ws, int tile, bool even) {
    const int halo = Ws + 1;
    const int cells = 1 + (tile + 6 * halo) + (tile + 4 * halo) + (even ? 0 : tile + 2 * halo);
    return (size_t)cells * 96 + 512 + 3 * 16 * 48 * 4;
}
// configurations built: 0 = none; 1: TILE 192, 4 + 4 tiles per wave in the first two phases, three workgroups per CU; 2: TILE 256, 5 + 5, two;
// 3: TILE 192, 4 + 4, two; 4: TILE 192, 5 + 4, two (halo rows up to 32 cells)
static int conv3x3_triple_config(int Ws, bool even) {
    static const int forced = experiment_int("KWS_T3_TRIPLE_CFG", 0);
    const int halo = Ws + 1;
    auto fits = [&](int cfg) {
        switch (cfg) {
            case 1: return 192 + 4 * halo <= 256 && conv3x3_triple_lds_bytes(Ws, 192, even) <= 53 * 1024;
            case 2: return 256 + 4 * halo <= 320 && conv3x3_triple_lds_bytes(Ws, 256, even) <= 80 * 1024 - 256;
            case 3: return 192 + 4 * halo <= 256 && conv3x3_triple_lds_bytes(Ws, 192, even) <= 80 * 1024 - 256;
            case 4: return 192 + 4 * halo <= 320 && 192 + 2 * halo <= 256 && conv3x3_triple_lds_bytes(Ws, 192, even) <= 80 * 1024 - 256;
        }
        return false;
    };
    if (forced && fits(forced)) return forced;
    for (int cfg = 1; cfg <= 4; ++cfg)
        if (fits(cfg)) return cfg;
    return 0;
}
bool conv3x3_triple_supported(int C, int Ws, bool first_even) { return (C + 7) / 8 * 8 == 48 && conv3x3_triple_config(Ws, first_even) != 0; }

template <bool F16, int JT1, int JT2, int JTB, bool EVEN, int WGS>
static hipError_t launch_triple_k(const TripleConvParams& p, hipStream_t s) {
    constexpr int tile = 64 * JTB;
    const unsigned grid = (unsigned)((p.total + tile - 1) / tile);
    const size_t lds = conv3x3_triple_lds_bytes(p.Ws, tile, EVEN);
    auto k = conv3x3_triple_kernel<F16, JT1, JT2, JTB, EVEN, WGS>;
    static DeviceOnce attr_once;
    if (attr_once.first()) {
        hipError_t e = allow_big_lds_at_base_zero(reinterpret_cast<const void*>(k));
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, s, p);
    return hipGetLastError();
}
template <bool F16, bool EVEN>
static hipError_t launch_triple_cfg(const TripleConvParams& p, int cfg, hipStream_t s) {
    switch (cfg) {
        case 1: return launch_triple_k<F16, 4, 4, 3, EVEN, 3>(p, s);
        case 2: return launch_triple_k<F16, 5, 5, 4, EVEN, 2>(p, s);
        case 3: return launch_triple_k<F16, 4, 4, 3, EVEN, 2>(p, s);
        case 4: return launch_triple_k<F16, 5, 4, 3, EVEN, 2>(p, s);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_conv3x3_triple(const TripleConvParams& p_in, int C, hipStream_t s) {
    if (p_in.total <= 0) return hipSuccess;
    TripleConvParams p = p_in;
    const bool even = p.first_even != 0;
    const int cfg = (C + 7) / 8 * 8 == 48 ? conv3x3_triple_config(p.Ws, even) : 0;
    const int halo = p.Ws + 1;
    p.wrap_clips = (3 * halo + p.cpc_in - 1) / std::max(p.cpc_in, 1);
    // positions are decoded with fp32 reciprocals (exact below 2^24), byte offsets are 32-bit, a wave steps through the table 16 cells at a time
    if (!cfg || p.cpc_in < 16 || (long long)p.total + (long long)(p.wrap_clips + 1) * p.cpc_in + 384 + 6 * halo >= (1 << 24) ||
        (long long)p.total * 96 >= (1LL << 31) || (long long)p.B * std::max(p.cpc_out, p.cpc_res) * 96 >= (1LL << 31) ||
        (even ? (p.res == nullptr || p.out2 != nullptr) : (p.out2 == nullptr || p.res != nullptr)))
        return hipErrorInvalidValue;
    if (p.f16) return even ? launch_triple_cfg<true, true>(p, cfg, s) : launch_triple_cfg<true, false>(p, cfg, s);
    return even ? launch_triple_cfg<false, true>(p, cfg, s) : launch_triple_cfg<false, false>(p, cfg, s);
}

// ------------------------------------------------------------------------------------------------ fp32 NCHW -> CL
// (B, C, H, W) fp32 -> pooled (window kh x kw, stride = window, floor; mode 0 average, 1 max; 1 x 1 = plain transpose)
// channels-last (B, H/kh, W/kw, cp) fp32, i.e. layout(1).  One thread: one output position x four channels.
__global__ __launch_bounds__(256) void nchw_to_cl_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                         long long total, int C, int H, int W, int Hp, int Wp, int kh,
                                                         int kw, int is_max, int cp, RangeGate rg) {
    if (range_gate_closed(rg)) return;
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
    float amax = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) amax = fmaxf(amax, fabsf(v[r]));
    range_note(rg, amax);
}

hipError_t launch_nchw_to_cl(const float* in, float* out, int B, int C, int H, int W, int kh, int kw, int is_max, int cp,
                             hipStream_t s, RangeGate rg) {
    const int Hp = H / kh, Wp = W / kw;
    const long long total = (long long)B * (cp / 4) * Hp * Wp;
    if (total <= 0) return hipSuccess;
    hipLaunchKernelGGL(nchw_to_cl_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, in, out, total, C, H, W,
                       Hp, Wp, kh, kw, is_max, cp, rg);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ conv_0 straight to CL
// conv_0 (1 -> C, 3x3, pad 1, no bias) + ReLU [+ AvgPool(kh, kw), stride = window, floor] of a ResNet (reference
// model/resnet.py:40-44), written directly as the channels-last fp32 tensor the tiled layers read: one thread = one (pooled) output
// position x four channels, plain fp32 FMAs over the 3 x 3 taps of every window member (1.6 MFLOP per clip: the pass is bound by
// its 0.8 MB per clip of output, not by arithmetic).  Replaces conv_igemm_kernel (NCHW out, 0.41 ms per 1 024 res15 clips at 1.6
// TB/s) + nchw_to_cl_kernel (0.29 - 0.56 ms).  Lanes run quad-fastest, so the 12 lanes of a position read the same input words.
// CLT: element type of the output tensor (CL_F32, or the 16-bit operand type of the `bf16` / `fp16` dtypes).  A thread owns one 16-byte
// chunk of the output cell (4 fp32 / 8 16-bit channels) of C0_PX consecutive output positions, its 9 x (4 | 8) weights in registers:
// with one position per thread the nine 16-byte weight loads of every thread kept the CU's vector-memory path busy for 144 of 180
// clocks per wave (0.28 ms per 1 024 res15 clips, three times what the output write needs).
// (Pooled outputs already cost kh*kw positions each: one per thread.)
template <int CLT, int C0_PX>
__global__ __launch_bounds__(256) void conv0_cl_kernel(const float* __restrict__ feat, const float* __restrict__ w9 /*[9][cp]*/,
                                                       void* __restrict__ out, long long total /* threads */, long long npos, int T, int F,
                                                       int Hp, int Wp, int kh, int kw, int cp, int fast, RangeGate rg) {
    if (range_gate_closed(rg)) return;
    constexpr int EPT = CLT == CL_F32 ? 4 : 8;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int nq = cp / EPT;
    const int q = (int)(i % nq);
    const long long pos0 = (i / nq) * C0_PX;
    float w[9][EPT];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int e = 0; e < EPT; e += 4) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(w9 + k * cp + EPT * q + e);
            w[k][e] = t[0]; w[k][e + 1] = t[1]; w[k][e + 2] = t[2]; w[k][e + 3] = t[3];
        }
    int ox = (int)(pos0 % Wp);
    long long t = pos0 / Wp;
    int oy = (int)(t % Hp);
    long long b = t / Hp;
    float amax = 0.f;
    if (C0_PX == 4 && fast) {
        // (r3) Row-aligned form (no pooling, F a multiple of four, 16-byte aligned rows): the thread's four positions share one 3 x 6 window of
        // the feature map -- three 16-byte loads + six single words instead of 36 bounds-checked loads with their index arithmetic (the generic
        // loop below spent ~220 vector instructions per position on 72 FMAs).  Same FMA order per output: bit-identical.
        const float* src = feat + b * (long long)T * F;
        float win[3][6];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int yy = oy + ky - 1;
            const bool rok = yy >= 0 && yy < T;
            const float* row = src + (long long)min(max(yy, 0), T - 1) * F + ox;
            const f32x4 mid = *reinterpret_cast<const f32x4*>(row);
            const float lft = ox > 0 ? row[-1] : 0.f, rgt = ox + 4 < F ? row[4] : 0.f;
            win[ky][0] = rok ? lft : 0.f;
            win[ky][1] = rok ? mid[0] : 0.f;
            win[ky][2] = rok ? mid[1] : 0.f;
            win[ky][3] = rok ? mid[2] : 0.f;
            win[ky][4] = rok ? mid[3] : 0.f;
            win[ky][5] = rok ? rgt : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float c[EPT];
#pragma unroll
            for (int e = 0; e < EPT; ++e) c[e] = 0.f;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int e = 0; e < EPT; ++e) c[e] = fmaf(w[3 * ky + kx][e], win[ky][u + kx], c[e]);
            float sum[EPT];
#pragma unroll
            for (int e = 0; e < EPT; ++e) sum[e] = 0.f + fmaxf(c[e], 0.f);
            const long long o = (pos0 + u) * cp + EPT * q;
            if (CLT == CL_F32) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(out) + o) = (f32x4){sum[0], sum[1], sum[2], sum[3]};
            else {
                const u32x2 lo = cl_pack4<CLT == CL_F16>((f32x4){sum[0], sum[1], sum[2], sum[3]});
                const u32x2 hi = cl_pack4<CLT == CL_F16>((f32x4){sum[EPT - 4], sum[EPT - 3], sum[EPT - 2], sum[EPT - 1]});
                *reinterpret_cast<u32x4*>(reinterpret_cast<unsigned short*>(out) + o) = (u32x4){lo[0], lo[1], hi[0], hi[1]};
            }
#pragma unroll
            for (int e = 0; e < EPT; ++e) amax = fmaxf(amax, fabsf(sum[e]));
        }
        range_note(rg, amax);
        return;
    }
    for (int u = 0; u < C0_PX && pos0 + u < npos; ++u) {
        const float* src = feat + b * (long long)T * F;
        float sum[EPT];
#pragma unroll
        for (int e = 0; e < EPT; ++e) sum[e] = 0.f;
        for (int my = 0; my < kh; ++my)
            for (int mx = 0; mx < kw; ++mx) {
                const int y = oy * kh + my, x = ox * kw + mx;
                float c[EPT];
#pragma unroll
                for (int e = 0; e < EPT; ++e) c[e] = 0.f;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const int yy = y + ky - 1, xx = x + kx - 1;
                        const float v = (yy >= 0 && yy < T && xx >= 0 && xx < F) ? src[yy * F + xx] : 0.f;
#pragma unroll
                        for (int e = 0; e < EPT; ++e) c[e] = fmaf(w[3 * ky + kx][e], v, c[e]);
                    }
#pragma unroll
                for (int e = 0; e < EPT; ++e) sum[e] += fmaxf(c[e], 0.f);
            }
        if (kh * kw > 1) {
#pragma unroll
            for (int e = 0; e < EPT; ++e) sum[e] = sum[e] / (float)(kh * kw);   // a true division, as nn.AvgPool2d's sum / count
        }
        const long long o = (pos0 + u) * cp + EPT * q;
        if (CLT == CL_F32) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(out) + o) = (f32x4){sum[0], sum[1], sum[2], sum[3]};
        else {
            const u32x2 lo = cl_pack4<CLT == CL_F16>((f32x4){sum[0], sum[1], sum[2], sum[3]});
            const u32x2 hi = cl_pack4<CLT == CL_F16>((f32x4){sum[EPT - 4], sum[EPT - 3], sum[EPT - 2], sum[EPT - 1]});
            *reinterpret_cast<u32x4*>(reinterpret_cast<unsigned short*>(out) + o) = (u32x4){lo[0], lo[1], hi[0], hi[1]};
        }
#pragma unroll
        for (int e = 0; e < EPT; ++e) amax = fmaxf(amax, fabsf(sum[e]));
        if (++ox == Wp) {
            ox = 0;
            if (++oy == Hp) {
                oy = 0;
                ++b;
            }
        }
    }
    range_note(rg, amax);
}

// (r3) The pooled form for the shipped pooling windows (PKH x PKW = 2 x 2: res26; 4 x 3: the res8 family on the tiled plan): one thread = one pooled
// position x one 16-byte chunk of channels; the (PKH + 2) x (PKW + 2) window of the feature map is loaded once (16 / 30 words instead of 36 / 108
// bounds-checked loads -- the generic loop spends three times as many vector instructions on indices and bounds as on FMAs).  Same FMA, ReLU-sum and
// division order as conv0_cl_kernel: bit-identical.
template <int CLT, int PKH, int PKW>
__global__ __launch_bounds__(256) void conv0_cl_pool_kernel(const float* __restrict__ feat, const float* __restrict__ w9 /*[9][cp]*/, void* __restrict__ out,
                                                            long long total /* threads */, int T, int F, int Hp, int Wp, int cp, RangeGate rg) {
    if (range_gate_closed(rg)) return;
    constexpr int EPT = CLT == CL_F32 ? 4 : 8;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int nq = cp / EPT;
    const int q = (int)(i % nq);
    const long long pos = i / nq;
    float w[9][EPT];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int e = 0; e < EPT; e += 4) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(w9 + k * cp + EPT * q + e);
            w[k][e] = t[0]; w[k][e + 1] = t[1]; w[k][e + 2] = t[2]; w[k][e + 3] = t[3];
        }
    const int ox = (int)(pos % Wp);
    const long long t = pos / Wp;
    const int oy = (int)(t % Hp);
    const long long b = t / Hp;
    const float* src = feat + b * (long long)T * F;
    const int y0 = oy * PKH - 1, x0 = ox * PKW - 1;       // window origin; rows y0+1 .. y0+PKH and columns x0+1 .. x0+PKW are always inside the map
    float win[PKH + 2][PKW + 2];
#pragma unroll
    for (int r = 0; r < PKH + 2; ++r) {
        const int yy = y0 + r;
        const bool rok = (r >= 1 && r <= PKH) || (yy >= 0 && yy < T);
        const float* row = src + (long long)min(max(yy, 0), T - 1) * F;
#pragma unroll
        for (int c = 0; c < PKW + 2; ++c) {
            const int xx = x0 + c;
            const bool ok = rok && ((c >= 1 && c <= PKW) || (xx >= 0 && xx < F));
            const float v = row[min(max(xx, 0), F - 1)];
            win[r][c] = ok ? v : 0.f;
        }
    }
    float sum[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) sum[e] = 0.f;
#pragma unroll
    for (int my = 0; my < PKH; ++my)
#pragma unroll
        for (int mx = 0; mx < PKW; ++mx) {
            float c[EPT];
#pragma unroll
            for (int e = 0; e < EPT; ++e) c[e] = 0.f;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int e = 0; e < EPT; ++e) c[e] = fmaf(w[3 * ky + kx][e], win[my + ky][mx + kx], c[e]);
#pragma unroll
            for (int e = 0; e < EPT; ++e) sum[e] += fmaxf(c[e], 0.f);
        }
#pragma unroll
    for (int e = 0; e < EPT; ++e) sum[e] = sum[e] / (float)(PKH * PKW);   // a true division, as nn.AvgPool2d's sum / count
    const long long o = pos * cp + EPT * q;
    if (CLT == CL_F32) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(out) + o) = (f32x4){sum[0], sum[1], sum[2], sum[3]};
    else {
        const u32x2 lo = cl_pack4<CLT == CL_F16>((f32x4){sum[0], sum[1], sum[2], sum[3]});
        const u32x2 hi = cl_pack4<CLT == CL_F16>((f32x4){sum[EPT - 4], sum[EPT - 3], sum[EPT - 2], sum[EPT - 1]});
        *reinterpret_cast<u32x4*>(reinterpret_cast<unsigned short*>(out) + o) = (u32x4){lo[0], lo[1], hi[0], hi[1]};
    }
    float amax = 0.f;
#pragma unroll
    for (int e = 0; e < EPT; ++e) amax = fmaxf(amax, fabsf(sum[e]));
    range_note(rg, amax);
}

template <int PKH, int PKW>
static hipError_t launch_conv0_cl_pool(const float* feat, const float* w9, void* out, int cl_type, long long total, int T, int F, int Hp, int Wp, int cp,
                                       hipStream_t s, RangeGate rg) {
    auto k = cl_type == CL_BF16 ? conv0_cl_pool_kernel<CL_BF16, PKH, PKW> : cl_type == CL_F16 ? conv0_cl_pool_kernel<CL_F16, PKH, PKW> : conv0_cl_pool_kernel<CL_F32, PKH, PKW>;
    hipLaunchKernelGGL(k, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, feat, w9, out, total, T, F, Hp, Wp, cp, rg);
    return hipGetLastError();
}

hipError_t launch_conv0_cl(const float* feat, const float* w9, void* out, int cl_type, int B, int T, int F, int kh, int kw, int cp,
                           hipStream_t s, RangeGate rg) {
    const int Hp = T / kh, Wp = F / kw;
    const long long npos = (long long)B * Hp * Wp;
    const int px = kh * kw > 1 ? 1 : 4;
    const long long total = (npos + px - 1) / px * (cp / (cl_type == CL_F32 ? 4 : 8));
    if (total <= 0) return hipSuccess;
    if (kh == 2 && kw == 2) return launch_conv0_cl_pool<2, 2>(feat, w9, out, cl_type, total, T, F, Hp, Wp, cp, s, rg);
    if (kh == 4 && kw == 3) return launch_conv0_cl_pool<4, 3>(feat, w9, out, cl_type, total, T, F, Hp, Wp, cp, s, rg);
    auto k = px == 1 ? (cl_type == CL_BF16 ? conv0_cl_kernel<CL_BF16, 1> : cl_type == CL_F16 ? conv0_cl_kernel<CL_F16, 1> : conv0_cl_kernel<CL_F32, 1>)
                     : (cl_type == CL_BF16 ? conv0_cl_kernel<CL_BF16, 4> : cl_type == CL_F16 ? conv0_cl_kernel<CL_F16, 4> : conv0_cl_kernel<CL_F32, 4>);
    // row-aligned fast path: a thread's four positions lie in one row of one clip and its rows are 16-byte aligned
    const int fast = px == 4 && F % 4 == 0 && Wp == F && Hp == T && (reinterpret_cast<uintptr_t>(feat) & 15) == 0 && npos % 4 == 0;
    hipLaunchKernelGGL(k, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, feat, w9, out, total, npos, T, F, Hp, Wp, kh, kw, cp, fast, rg);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ mean + linear on CL
// ResNet tail (reference model/resnet.py:57-59), last BatchNorm folded in: mean(BN(x)) == BN(mean(x)).
// One workgroup per clip; x is (B, HW, cp) in layout(1), element type CLT.  A thread owns one 16-byte chunk of the cell (4 fp32 or 8
// 16-bit channels) and every nsl-th cell, four loads in flight (element-wise 2- / 4-byte reads ran at 1.2 TB/s).
template <int CLT>
__global__ __launch_bounds__(256) void mean_linear_cl_kernel(const void* __restrict__ x, float* __restrict__ logits,
                                                             int C, int cp, int HW, const float* mean, const float* rstd,
                                                             const float* __restrict__ wt, const float* __restrict__ bias,
                                                             int n_out, RangeGate rg) {
    extern __shared__ float sm[];   // [nsl][cp] partial sums, then [cp] means
    if (range_gate_closed(rg)) return;
    constexpr int EPC = CLT == CL_F32 ? 4 : 8;   // elements per 16-byte chunk
    const int b = blockIdx.x;
    const int nq = cp / EPC, nsl = 256 / nq;     // chunks per cell, cell slices summed in parallel
    const int qd = threadIdx.x % nq, sl = threadIdx.x / nq;
    const char* base = reinterpret_cast<const char*>(x) + ((size_t)b * HW * cp + (size_t)qd * EPC) * (CLT == CL_F32 ? 4 : 2);
    const size_t cell_b = (size_t)cp * (CLT == CL_F32 ? 4 : 2);
    if (sl < nsl) {
        float acc[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
        auto add = [&](u32x4 v) {
            if constexpr (CLT == CL_F32) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const unsigned word = v[e];   // (never __builtin_bit_cast a vector ELEMENT: hipcc 7.2 then reads element 0)
                    acc[e] += __builtin_bit_cast(float, word);
                }
            } else if constexpr (CLT == CL_F16) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const unsigned word = v[e];
                    acc[2 * e] += (float)__builtin_bit_cast(_Float16, (unsigned short)(word & 0xffffu));
                    acc[2 * e + 1] += (float)__builtin_bit_cast(_Float16, (unsigned short)(word >> 16));
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const unsigned word = v[e];
                    acc[2 * e] += lo_f(word);
                    acc[2 * e + 1] += hi_f(word);
                }
            }
        };
        int i = sl;
        for (; i + 3 * nsl < HW; i += 4 * nsl) {
            u32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const u32x4*>(base + (size_t)(i + u * nsl) * cell_b);
#pragma unroll
            for (int u = 0; u < 4; ++u) add(v[u]);
        }
        for (; i < HW; i += nsl) add(*reinterpret_cast<const u32x4*>(base + (size_t)i * cell_b));
#pragma unroll
        for (int e = 0; e < EPC; ++e) sm[sl * cp + qd * EPC + e] = acc[e];
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

hipError_t launch_mean_linear_cl(const void* x, int cl_type, float* logits, int B, int C, int cp, int HW, const float* mean,
                                 const float* rstd, const float* w, const float* bias, int n_out, hipStream_t s, RangeGate rg) {
    if (B <= 0) return hipSuccess;
    if (cp > 256 || cp % 8) return hipErrorInvalidValue;
    const int nsl = 256 / (cp / (cl_type == CL_F32 ? 4 : 8));
    auto k = cl_type == CL_BF16 ? mean_linear_cl_kernel<CL_BF16> : cl_type == CL_F16 ? mean_linear_cl_kernel<CL_F16> : mean_linear_cl_kernel<CL_F32>;
    hipLaunchKernelGGL(k, dim3((unsigned)B), dim3(256), (size_t)(nsl + 1) * cp * sizeof(float), s, x,
                       logits, C, cp, HW, mean, rstd, w, bias, n_out, rg);
    return hipGetLastError();
}

}  // namespace kws
