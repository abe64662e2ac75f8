// Weight-stationary, streaming 3x3 "same" convolutions over 16-bit channels-last tensors (the `bf16` / `fp16` dtypes): runs of THREE
// consecutive ResNet layers of equal dilation, or a single layer, in one persistent kernel.  Reference: model/resnet.py:20-26, 44-56
// (Conv2d(C, C, 3, padding=d, dilation=d, bias=False) -> ReLU -> (+ prev_x on even i) -> BatchNorm(affine=False)); res15 / res26 /
// hey_snips with 41-48 channels.
//
// Why (round 5).  The tile kernels of conv3x3_tile.hip re-read a layer's weight fragments from L1 / L2 for every 64-position strip (51 of
// the CU's 64 B/clk at full matrix rate), pay a memory round trip, three barriers with empty pipelines and ~900 set-up / epilogue vector
// instructions per 192-320 positions, and compute a fifth of their MFMAs on halo rows: res15 `bf16` sat at 0.40 matrix-pipe busy and
// 1.9 TB/s -- under both roofs (profiles/r04/final_res15_bf16_summary.json).  Here
//   * a layer's 42 weight fragments (14 k-steps x 3 channel tiles, 41 KB) live in ONE wave's registers for the whole launch (168 of a 512-
//     register budget: one workgroup of four waves per CU, one wave per SIMD): the k-loop reads nothing but its B fragments from LDS;
//   * the three layers of a run are three WAVES of the workgroup working on the same stream of positions, `LAG` = 112 positions apart: layer l
//     reads its input from an LDS ring that layer l - 1 (or the DMA loader, for the first layer) fills, and writes its output into the next
//     ring -- intermediate maps never leave the CU, there is no halo recomputation inside a workgroup's span (only 2 x (Ws + 1) positions per
//     layer at its two ends, of ~16 000), and one barrier per 64 positions is the only synchronisation;
//   * the fourth wave is the LOADER: it copies the input tensor into ring 0 with global_load_lds_dwordx4 (memory -> LDS, no registers) two
//     steps ahead of the first layer, and turns the per-position table of the host (tap mask, border class, output / residual cells) into
//     16-byte records in an LDS ring, so that the compute waves never divide, never gather and -- except for a residual that comes from
//     memory and the final stores -- never touch global memory.
// A single layer (res15's 13th; leftovers of other depths) runs as three compute waves on thirds of a 192-position step + the loader.
//
// Arithmetic is the tile kernels' bit for bit: same fragments (pack_conv3x3_tile_weights_f16 / pack_conv_weights_bf16x6), same K order
// (k-step s, lane group g -> block 4 s + g = (tap, 8-channel block)), one accumulation chain per (position tile, channel tile), same epilogue
// (relu(fma(acc, 2^-S, border bias)) + residual, round to nearest even), same rounding of the intermediate maps to the tensor type: the
// results equal the pair / triple / one-kernel-per-layer forms bit for bit (tests/test_gpu_parity.py).
//
// Positions are the flattened cells of layout(d) (conv3x3_tile.hip); a tap that leaves its sub-map reads the shared zero cell at LDS offset 0
// (per-position tap mask from the table).  Rings: slot(p) = (p - X0) mod R with guard copies of the first / last cells behind / in front of
// the ring, so a B-fragment address is `ring base + slot * 96 + tap offset` with no wrap arithmetic.
#include "kws_internal.h"

#include <algorithm>
#include <type_traits>

namespace kws {

namespace {
typedef __bf16 s_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 s_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 s_f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 s_bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned su32x4 __attribute__((ext_vector_type(4)));
typedef unsigned su32x2 __attribute__((ext_vector_type(2)));
typedef int si32x4 __attribute__((ext_vector_type(4)));

typedef const su32x4 __attribute__((address_space(3))) * s_lds_u32x4_cptr;
typedef su32x4 __attribute__((address_space(3))) * s_lds_u32x4_ptr;
typedef const su32x2 __attribute__((address_space(3))) * s_lds_u32x2_cptr;
typedef su32x2 __attribute__((address_space(3))) * s_lds_u32x2_ptr;
typedef const si32x4 __attribute__((address_space(3))) * s_lds_i32x4_cptr;
typedef si32x4 __attribute__((address_space(3))) * s_lds_i32x4_ptr;
typedef const f32x4 __attribute__((address_space(3))) * s_lds_f32x4_cptr;
__device__ __forceinline__ su32x2 s_read8(int addr) { return *reinterpret_cast<s_lds_u32x2_cptr>((unsigned)addr); }
__device__ __forceinline__ void s_write8(int addr, su32x2 v) { *reinterpret_cast<s_lds_u32x2_ptr>((unsigned)addr) = v; }

__device__ __forceinline__ float s_relu1(float x) {
    const int b = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, b > 0 ? b : 0);
}
__device__ __forceinline__ unsigned s_pack2(float a, float b) {
    const s_bf16x2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, v);
}
template <bool F16>
__device__ __forceinline__ su32x2 s_pack4(f32x4 v) {
    if (F16) return (su32x2){__builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){v[0], v[1]}, s_f16x2)),
                             __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){v[2], v[3]}, s_f16x2))};
    return (su32x2){s_pack2(v[0], v[1]), s_pack2(v[2], v[3])};
}
template <bool F16>
__device__ __forceinline__ f32x4 s_unpack4(su32x2 rw) {
    f32x4 rv;
    if (F16) {   // (scalar conversions on purpose: __builtin_bit_cast of a vector ELEMENT reads element 0 with hipcc 7.2)
        rv[0] = (float)__builtin_bit_cast(_Float16, (unsigned short)(rw[0] & 0xffffu));
        rv[1] = (float)__builtin_bit_cast(_Float16, (unsigned short)(rw[0] >> 16));
        rv[2] = (float)__builtin_bit_cast(_Float16, (unsigned short)(rw[1] & 0xffffu));
        rv[3] = (float)__builtin_bit_cast(_Float16, (unsigned short)(rw[1] >> 16));
    } else {
        rv = (f32x4){__builtin_bit_cast(float, rw[0] << 16), __builtin_bit_cast(float, rw[0] & 0xffff0000u),
                     __builtin_bit_cast(float, rw[1] << 16), __builtin_bit_cast(float, rw[1] & 0xffff0000u)};
    }
    return rv;
}
// q / dv for 0 <= q < 2^24 with a precomputed reciprocal (exact after one correction step); rem receives q % dv
__device__ __forceinline__ int s_fdiv(int q, int dv, float inv, int& rem) {
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
__device__ __forceinline__ int s_mod(int x, int r) {   // x mod r for any sign of x (uniform values: scalar arithmetic)
    int m = x % r;
    return m < 0 ? m + r : m;
}

constexpr int SC = 96;            // bytes per cell: 48 channels x 2 B
constexpr int S_STEPS = 14;       // k-steps: 9 taps x 6 blocks of 8 channels, four blocks per MFMA
constexpr int S_MT = 3;           // 16-channel output tiles

// geometry of a run of L layers
template <int L>
struct StreamCfg;
template <>
struct StreamCfg<3> {
    static constexpr int S = 64;          // positions per step and layer (four tiles of the layer's one wave)
    static constexpr int LAG = 128;       // layer l + 1 works two steps behind layer l: the step layer l is writing never meets the (Ws + 1 <= 48)-cell reach of its consumer, and every layer's steps stay 64-aligned in every ring
    static constexpr int BACK = 96;       // the first layer starts 2 x 48 positions in front of the span (the last layer needs it from S0 - 2 (Ws + 1))
    static constexpr int R0 = 384, G0 = 64;   // ring 0 (the staged input): live window = 128 (residual of layer 1) + 256 (three steps of look-ahead + the segment in flight)
    static constexpr int RREC = 512;      // record ring (power of two)
};
template <>
struct StreamCfg<1> {
    static constexpr int S = 192;         // three compute waves, four tiles each
    static constexpr int LAG = 0;
    static constexpr int BACK = 0;
    static constexpr int R0 = 960, G0 = 64;   // five steps: live window = 41 (reach) + 4 S (the step being read, two of look-ahead, the segment in flight); guards of 64 cells (>= the 48-cell reach)
    static constexpr int RREC = 1024;
};
// (the loader's DMA stream runs THREE steps in front of the first layer: segments stay aligned to the ring, the copy issued in iteration t - 2 is
// waited for at the top of iteration t - 1 and read from step t on)
constexpr int S_T_START = -4;             // loader iterations in front of the first compute step

template <int L, bool EVEN>
struct StreamLds {
    using C = StreamCfg<L>;
    // rings 1 and 2 hold 256 cells (live window: the 64 cells being written + two steps of lag + the consumer's 41-cell reach) unless the map is ALSO
    // the residual of the layer after next (x_a of an even-first run: ring 1), which reads it two more steps behind: 320
    static constexpr int R1 = L == 3 ? (EVEN ? 320 : 256) : 0, R2 = L == 3 ? 256 : 0, G1 = L == 3 ? 48 : 0;
    // ring 3 (odd-first runs): the LAST layer's output, two steps, copied to memory by the storer wave one step behind (the compute waves then issue no
    // memory operation at all; an even-first run has no LDS left for it and stores from its compute waves)
    static constexpr int R3 = (L == 3 && !EVEN) ? 128 : 0;
    static constexpr bool STORER = R3 > 0;
    static constexpr int ZERO = 0;                                   // zeros: what a dead tap reads (address 0 + the tile's immediate offset 16 k cells + 16 bytes)
    static constexpr int ZERO_BYTES = 5120;
    static constexpr int BORDER = ZERO_BYTES;                        // [L][16 classes][48] floats
    static constexpr int KOFF = BORDER + L * 16 * 48 * 4;            // [4 lane groups][16] ints: byte offset of (tap, channel block) of k-step s relative to the centre cell
    static constexpr int REC = KOFF + 256;                           // [RREC] x {tap mask | class << 9 | valid << 13, output cell, residual cell, 0}
    static constexpr int MAP0 = REC + C::RREC * 16 + C::G0 * SC;     // byte address of slot 0 of ring 0 (its front guard lies below)
    static constexpr int MAP0_END = MAP0 + (C::R0 + C::G0) * SC;
    static constexpr int MAP1 = MAP0_END + G1 * SC;
    static constexpr int MAP1_END = MAP1 + (R1 + G1) * SC;
    static constexpr int MAP2 = MAP1_END + G1 * SC;
    static constexpr int MAP2_END = MAP2 + (R2 + G1) * SC;
    static constexpr int MAP3 = MAP2_END;
    static constexpr int MAP3_END = MAP3 + R3 * SC;
    // an even-first run's first layer takes its residual x_{a-2} from memory: the LOADER fetches those cells (through the records' residual cell) and
    // parks them here, two steps of 64 cells by step parity, so that the compute waves read it like any other ring
    static constexpr int RSTAGE = MAP3_END;
    static constexpr int RSTAGE_STEP = 64 * SC;
    static constexpr int BYTES = RSTAGE + ((L == 3 && EVEN) ? 2 * RSTAGE_STEP : 0);
};
static_assert(StreamLds<3, false>::BYTES <= 160 * 1024 - 256 && StreamLds<3, true>::BYTES <= 160 * 1024 - 256 && StreamLds<1, false>::BYTES <= 160 * 1024 - 256, "one workgroup per CU");
static_assert(StreamCfg<3>::R0 % StreamCfg<3>::S == 0 && StreamCfg<1>::R0 % StreamCfg<1>::S == 0 && StreamCfg<3>::R0 % 64 == 0 && StreamCfg<1>::R0 % 64 == 0 &&
                  StreamCfg<3>::LAG % 64 == 0 && StreamCfg<3>::BACK % 16 == 0 && 3 * 16 * SC + 16 <= StreamLds<3, false>::ZERO_BYTES,
              "segments never wrap inside ring 0, and the four tiles of a step are consecutive slots of every ring");

#define SMFH(A_, B_, C_) C_ = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(s_f16x8, A_), __builtin_bit_cast(s_f16x8, B_), C_, 0, 0, 0)
#define SMFB(A_, B_, C_) C_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(s_bf16x8, A_), __builtin_bit_cast(s_bf16x8, B_), C_, 0, 0, 0)
}  // namespace

// ELL: which layer of the run this wave computes (0 .. L - 1).  Everything that depends on it is a compile-time constant of the instantiation.
template <bool F16, int L, bool EVEN, int ELL>
struct StreamRole {
    using C = StreamCfg<L>;
    using M = StreamLds<L, EVEN>;
    static constexpr bool LAST = ELL == L - 1;
    static constexpr bool LAYER_EVEN = EVEN ? (ELL % 2 == 0) : (ELL % 2 == 1);     // parity of the reference's layer index a + ELL
    static constexpr int RES = !LAYER_EVEN ? 0 : (ELL == 0 ? 1 : 2);                 // residual: none / from memory (table cell) / from LDS ring ELL - 1
    static constexpr bool TO_RING = !LAST || M::STORER;                            // the output goes to an LDS ring (the next layer's input, or the storer's ring 3)
    static constexpr bool OUT2 = L == 3 && !EVEN && ELL == 1 && !M::STORER;        // x_{a+1} also goes to memory (the next run's residual) -- from the compute waves only where there is no storer
    static constexpr int IN_ADDR = ELL == 0 ? M::MAP0 : (ELL == 1 ? M::MAP1 : M::MAP2);
    static constexpr int IN_R = ELL == 0 ? C::R0 : (ELL == 1 ? M::R1 : M::R2);
    static constexpr int OUT_ADDR = ELL == 0 ? M::MAP1 : (ELL == 1 ? M::MAP2 : M::MAP3);
    static constexpr int OUT_R = ELL == 0 ? M::R1 : (ELL == 1 ? M::R2 : M::R3);
    static constexpr int OUT_G = LAST ? 0 : M::G1;                                  // (ring 3 has no guards: nobody taps it)
    static constexpr int RES_ADDR = ELL == 1 ? M::MAP0 : M::MAP1;                   // (RES == 2: x_{i-2} is the input of layer ELL - 1)
    static constexpr int RES_R = ELL == 1 ? C::R0 : M::R1;
};

// compile-time unrolled loop: f(std::integral_constant<int, i>) for i in [B, E)
template <int B, int E, class F>
__device__ __forceinline__ void stream_unroll(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        stream_unroll<B + 1, E>(f);
    }
}

// A layer is TWO waves on one SIMD (waves w and w + 4 of the workgroup): half 0 owns channel tiles 0 and 1, half 1 channel tile 2.  One wave per
// SIMD could not feed the matrix pipe: a lone wave issues one instruction per ~4 clocks, and a position tile is 42 MFMAs (672 clocks) beside ~260
// other instructions (measured: 1 750 - 2 100 clocks per tile).  Two waves double the issue slots, fill each other's MFMA bubbles, and split the
// weight registers (112 / 56 of 256 each); neither ever waits for the other -- every (position tile, channel tile) keeps its own accumulation chain.
template <int HALF>
struct StreamHalf {
    static constexpr int M0 = HALF == 0 ? 0 : 2;
    static constexpr int NM = HALF == 0 ? 2 : 1;
};

template <int NM>
struct StreamWaveState {     // what a compute wave keeps across steps (registers)
    su32x4 A[S_STEPS][NM];
    int ktap[5];                // taps of the five k-steps whose lane groups straddle two taps (s = 1, 4, 7, 10, 13)
    int base[S_STEPS];          // LDS address of tile 0's B fragment of k-step s in the CURRENT step (advanced by 64 cells per step)
    int tmk[4], ocl[4];         // the current step's records: tap mask | border class << 9 | valid << 13, output cell (read at the end of the step before)
    unsigned amax;
};

// records of the step that starts at p_step into the wave state (tap-mask word, output cell) -- read one step ahead, at the END of the step before
template <bool F16, int L, bool EVEN, int ELL, int HALF>
__device__ __forceinline__ void stream_fetch_records(const StreamConvParams& p, StreamWaveState<StreamHalf<HALF>::NM>& st, const int p_step, const int X0,
                                                     const int need_lo, const int need_hi, const int pcol) {
    using R = StreamRole<F16, L, EVEN, ELL>;
    using C = StreamCfg<L>;
    using M = StreamLds<L, EVEN>;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int p0 = p_step + 16 * k;
        const int ra = M::REC + ((((p0 - X0) & (C::RREC - 1)) + pcol) << 4);
        if (R::LAST && !R::TO_RING) {
            const su32x2 e = s_read8(ra);
            st.tmk[k] = (int)e[0];
            st.ocl[k] = (int)e[1];
        } else {
            st.tmk[k] = *reinterpret_cast<const int __attribute__((address_space(3)))*>((unsigned)ra);
            st.ocl[k] = 0;
        }
        const int pl = p0 + pcol;
        // outside the tensor, or a position nobody needs from this layer (only at the two ends of a span): no live tap, no epilogue -- the tile still
        // runs its MFMAs on zeros, so the k-loop has no run-time branch.  Per POSITION, not per tile: in a tile that straddles the end of what is needed
        // the positions beyond it would be computed from ring cells nobody wrote -- harmless for the results, but their garbage reached the fp16
        // range guard's maximum (found on res26 `fp16` at 4 096 clips: a spurious second pass, 18.5 ms instead of 5.6)
        if (!((unsigned)pl < (unsigned)p.total) || pl < need_lo || pl >= need_hi) st.tmk[k] = 0;
    }
}

// one step of one compute wave: four position tiles starting at the (wave-uniform, 64-aligned) position p_step, this wave's channel tiles.
// Software-pipelined by hand, one scheduling fence per k-step: [MFMAs of (tile k, k-step s)] [B fragment eight k-steps ahead] [a slice of tile k - 1's
// epilogue]; a tile's border bias / LDS residual are requested at its k-step 5 and consumed with the next tile's first k-steps, so the wave neither
// drains the matrix pipe for an epilogue nor waits on an LDS round trip inside one.
template <bool F16, int L, bool EVEN, int ELL, int HALF>
__device__ __forceinline__ void stream_compute_step(const StreamConvParams& p, StreamWaveState<StreamHalf<HALF>::NM>& st, const int p_step, const int X0, const int S0,
                                                    const int S1, const int need_lo, const int need_hi, const int g, const int pcol, const int tstep) {
    using R = StreamRole<F16, L, EVEN, ELL>;
    using C = StreamCfg<L>;
    using M = StreamLds<L, EVEN>;
    using HF = StreamHalf<HALF>;
    constexpr int NM = HF::NM;
    // slots of the step's first tile in the rings: steps are 64-aligned and every ring is a multiple of 64 long, so tile k sits 16 k slots on
    const int s_out = R::TO_RING ? s_mod(p_step - X0, R::OUT_R) : 0;
    const int s_res = R::RES == 2 ? s_mod(p_step - X0, R::RES_R) : 0;
    const float inv_scale = p.inv_scale[ELL];
    char* const outp = reinterpret_cast<char*>(p.out);
    char* const out2p = reinterpret_cast<char*>(p.out2);

    // B fragments: st.base[s] is the address of tile 0's fragment of k-step s; tile k adds 16 k cells as the read's immediate offset.  A dead tap's
    // address is 0 -- the zero region at the bottom of LDS is long enough for the four immediates.  A ring of NBUF registers runs over the step's
    // 4 x 14 (tile, k-step) sequence: fragment i + NBUF is requested right behind the MFMAs that consume fragment i.
    constexpr int NBUF = 8, NSEQ = 4 * S_STEPS;
    su32x4 b[NBUF];
    // tap of (k-step s, lane group g): block 4 s + g of 6 per tap.  In nine k-steps all four lane groups share a tap (an immediate); in the other five
    // (s = 1, 4, 7, 10, 13) groups 2, 3 are one tap further (a per-lane register; 31 = the zero-weight padding blocks of k-step 13, bit 31 is never set)
#ifndef STREAM_ABLATE   // (experiments, results wrong) 1: no B-fragment reads after the first eight; 2: no MFMAs; 4: no epilogues; 8: B reads without the tap mask
#define STREAM_ABLATE 0
#endif
#define S_LOADB(I_)                                                                                             \
    if (!((STREAM_ABLATE & 1) && (I_) >= NBUF)) {                                                               \
        constexpr int k_ = (I_) / S_STEPS, s_ = (I_) % S_STEPS;                                                 \
        int msk_; /* 0 or -1 */                                                                                 \
        if (s_ % 3 == 1) msk_ = __builtin_amdgcn_sbfe(st.tmk[k_], st.ktap[s_ / 3], 1);                          \
        else asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(msk_) : "v"(st.tmk[k_]), "n"((4 * s_) / 6));                  \
        b[(I_) % NBUF] = *reinterpret_cast<s_lds_u32x4_cptr>((unsigned)(st.base[s_] & msk_) + k_ * 16 * SC);    \
    }
    stream_unroll<0, NBUF>([&](auto ic) { S_LOADB(decltype(ic)::value) });
    f32x4 acc[2][NM];          // by tile parity: the next tile's chain starts while the last one's epilogue runs
    f32x4 ebb[NM];             // border bias of the tile whose epilogue comes next
    su32x2 erv[NM];            // ... and its residual (from the LDS ring or the wave's parked block)
    // epilogue of (tile k, channel-tile slot m): relu(fma(acc, 2^-S, bias)) + residual -> round -> ring / memory
    auto epilogue = [&](auto kc, auto mc) {
        constexpr int k = decltype(kc)::value, m = decltype(mc)::value;
        const int tm = st.tmk[k];
        if ((tm >> 13) & 1) {
            const int pl = p_step + 16 * k + pcol;
            const bool own = pl >= S0 && pl < S1;
            const int so = s_out + 16 * k;
            const int co0 = (HF::M0 + m) * 16 + 4 * g;
            f32x4 rv = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (R::RES != 0) rv = s_unpack4<F16>(erv[m]);
            f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = s_relu1(fmaf(acc[k & 1][m][r], inv_scale, ebb[m][r])) + rv[r];   // (x + 0 = x exactly for x >= +0: the same bits as without a residual)
                if (F16) st.amax = max(st.amax, __builtin_bit_cast(unsigned, v[r]));      // bit patterns: every stored value is >= +0; a NaN or a sign bit reads as huge
            }
            const su32x2 pk = s_pack4<F16>(v);
            if (STREAM_ABLATE & 48) asm volatile("" ::"v"(pk));     // (experiments) 16: no stores to memory; 32: no ring writes
            if (R::TO_RING && !(STREAM_ABLATE & 32)) {
                const int oa = R::OUT_ADDR + (so + pcol) * SC + co0 * 2;
                s_write8(oa, pk);
                if (R::OUT_G > 0 && so < R::OUT_G) s_write8(oa + R::OUT_R * SC, pk);               // (uniform) the guard copy behind the ring
                if (R::OUT_G > 0 && so >= R::OUT_R - R::OUT_G) s_write8(oa - R::OUT_R * SC, pk);   // (uniform) ... in front of it
            } else if (!R::TO_RING && own && !(STREAM_ABLATE & 16)) {
                *reinterpret_cast<su32x2*>(outp + (size_t)st.ocl[k] * SC + co0 * 2) = pk;
            }
            if (R::OUT2 && own && !(STREAM_ABLATE & 16)) *reinterpret_cast<su32x2*>(out2p + (size_t)pl * SC + co0 * 2) = pk;
        }
    };
    // the operands of tile k's epilogue: requested while its k-loop still runs
    auto epilogue_operands = [&](auto kc) {
        constexpr int k = decltype(kc)::value;
        const int bmask = (st.tmk[k] >> 9) & 15;
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            const int co0 = (HF::M0 + m) * 16 + 4 * g;
            ebb[m] = *reinterpret_cast<s_lds_f32x4_cptr>((unsigned)(M::BORDER + ((ELL * 16 + bmask) * 48 + co0) * 4));
            if (R::RES == 1) erv[m] = s_read8(M::RSTAGE + (tstep & 1) * M::RSTAGE_STEP + (16 * k + pcol) * SC + co0 * 2);   // (parked by the loader, by step parity)
            if (R::RES == 2) erv[m] = s_read8(R::RES_ADDR + (s_res + 16 * k + pcol) * SC + co0 * 2);
        }
    };
    stream_unroll<0, NSEQ>([&](auto ic) {
        constexpr int i = decltype(ic)::value, k = i / S_STEPS, s = i % S_STEPS;
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            if (s == 0) acc[k & 1][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (STREAM_ABLATE & 2) acc[k & 1][m][0] += __builtin_bit_cast(float, b[i % NBUF][0]);
            else if (F16) SMFH(st.A[s][m], b[i % NBUF], acc[k & 1][m]);
            else SMFB(st.A[s][m], b[i % NBUF], acc[k & 1][m]);
        }
        if (i + NBUF < NSEQ) S_LOADB(i + NBUF < NSEQ ? i + NBUF : 0)
        if (!(STREAM_ABLATE & 4) && k >= 1 && s >= 1 && s <= NM) epilogue(std::integral_constant<int, (k >= 1 ? k - 1 : 0)>{}, std::integral_constant<int, (s >= 1 && s <= NM ? s - 1 : 0)>{});
        if (s == 5) epilogue_operands(std::integral_constant<int, k>{});   // (behind the previous tile's epilogue, which has consumed ebb / erv by k-step NM)
        __builtin_amdgcn_sched_barrier(0);
    });
    if (!(STREAM_ABLATE & 4)) stream_unroll<0, NM>([&](auto mc) { epilogue(std::integral_constant<int, 3>{}, mc); });
    else {
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int m = 0; m < NM; ++m) asm volatile("" ::"v"(acc[q][m]));
    }
#undef S_LOADB
    // the next step's records, and its fragment addresses: one step on in the input ring (or back to its start)
    stream_fetch_records<F16, L, EVEN, ELL, HALF>(p, st, p_step + C::S, X0, need_lo, need_hi, pcol);
    {
        const int s_in = s_mod(p_step - X0, R::IN_R);
        const int delta = (s_in + C::S >= R::IN_R ? C::S - R::IN_R : C::S) * SC;      // (uniform; a step never straddles the ring's end)
#pragma unroll
        for (int s = 0; s < S_STEPS; ++s) st.base[s] += delta;
    }
}

// before a wave's first step: its records and fragment addresses
template <bool F16, int L, bool EVEN, int ELL, int HALF>
__device__ __forceinline__ void stream_prime(const StreamConvParams& p, StreamWaveState<StreamHalf<HALF>::NM>& st, const int p_step, const int X0, const int need_lo,
                                             const int need_hi, const int g, const int pcol) {
    using R = StreamRole<F16, L, EVEN, ELL>;
    using M = StreamLds<L, EVEN>;
    stream_fetch_records<F16, L, EVEN, ELL, HALF>(p, st, p_step, X0, need_lo, need_hi, pcol);
    const int lb = R::IN_ADDR + (s_mod(p_step - X0, R::IN_R) + pcol) * SC;
    si32x4 ko[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) ko[q] = *reinterpret_cast<s_lds_i32x4_cptr>((unsigned)(M::KOFF + g * 64 + q * 16));
#pragma unroll
    for (int s = 0; s < S_STEPS; ++s) st.base[s] = lb + ko[s >> 2][s & 3];
}

// everything a compute wave does, for one half of one layer
template <bool F16, int L, bool EVEN, int HALF>
__device__ __forceinline__ void stream_compute_wave(const StreamConvParams& p, const int lw, const int X0, const int S0, const int S1, const int NT, const int lane) {
    using C = StreamCfg<L>;
    using HF = StreamHalf<HALF>;
    const int g = lane >> 4, pcol = lane & 15;
    const int H = p.Ws + 1;
    const int ell = L == 3 ? lw : 0;
    StreamWaveState<HF::NM> st;
    st.amax = 0u;
    {   // this wave's share of its layer's weight fragments, for the whole launch
        const int wp = F16 ? 2 : 3;                                  // parts per fragment group as packed on the host (part 0 is the operand)
        const su32x4* Ag = reinterpret_cast<const su32x4*>(p.apk[ell]) + lane;
#pragma unroll
        for (int s = 0; s < S_STEPS; ++s)
#pragma unroll
            for (int m = 0; m < HF::NM; ++m) st.A[s][m] = Ag[((s * S_MT + HF::M0 + m) * wp) * 64];
        // The fragments have ARRIVED before the step loop starts: an empty asm that reads them makes hipcc wait here, once.  Left pending into the
        // loop, its wait-count pass guards every MFMA of every tile with `s_waitcnt vmcnt(N)` -- and vmcnt counts the wave's stores and its
        // residual requests too.
#pragma unroll
        for (int s = 0; s < S_STEPS; ++s)
#pragma unroll
            for (int m = 0; m < HF::NM; ++m) asm volatile("" : "+v"(st.A[s][m]));
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            const int bi = 4 * (3 * q + 1) + g, tap = bi / 6;
            st.ktap[q] = tap < 9 ? tap : 31;                         // (zero-weight padding blocks test bit 31 of the mask word, never set)
        }
    }
    const int need_margin = (L - 1 - ell) * H;
    const int need_lo = max(S0 - need_margin, 0), need_hi = min(S1 + need_margin, p.total);
    const int x_wave = L == 3 ? X0 - ell * C::LAG : X0 + lw * 64;    // where this wave's stream starts (a single layer: three thirds of a 192-position step)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // (the workgroup's set-up barrier)
#ifdef STREAM_TIMING   // (experiments) where a wave's time goes: cycles in its own work / at the step barrier, printed by two workgroups
    unsigned long long tw_work = 0, tw_bar = 0, tw0 = __builtin_readcyclecounter();
#define ST_T(x) x
#else
#define ST_T(x)
#endif
    for (int t = S_T_START; t < NT; ++t) {
        ST_T(const unsigned long long ta = __builtin_readcyclecounter();)
        if (t == -1) {
            // step 0's records and fragment addresses
            if (L == 3) {
                if (ell == 0) stream_prime<F16, L, EVEN, 0, HALF>(p, st, x_wave, X0, need_lo, need_hi, g, pcol);
                else if (ell == 1) stream_prime<F16, L, EVEN, L == 3 ? 1 : 0, HALF>(p, st, x_wave, X0, need_lo, need_hi, g, pcol);
                else stream_prime<F16, L, EVEN, L == 3 ? 2 : 0, HALF>(p, st, x_wave, X0, need_lo, need_hi, g, pcol);
            } else {
                stream_prime<F16, L, EVEN, 0, HALF>(p, st, x_wave, X0, need_lo, need_hi, g, pcol);
            }
        } else if (t >= 0) {
            const int p_step = x_wave + t * C::S;
            if (L == 3) {
                if (ell == 0) stream_compute_step<F16, L, EVEN, 0, HALF>(p, st, p_step, X0, S0, S1, need_lo, need_hi, g, pcol, t);
                else if (ell == 1) stream_compute_step<F16, L, EVEN, L == 3 ? 1 : 0, HALF>(p, st, p_step, X0, S0, S1, need_lo, need_hi, g, pcol, t);
                else stream_compute_step<F16, L, EVEN, L == 3 ? 2 : 0, HALF>(p, st, p_step, X0, S0, S1, need_lo, need_hi, g, pcol, t);
            } else {
                stream_compute_step<F16, L, EVEN, 0, HALF>(p, st, p_step, X0, S0, S1, need_lo, need_hi, g, pcol, t);
            }
        }
        ST_T(const unsigned long long tb = __builtin_readcyclecounter(); if (t >= 0) tw_work += tb - ta;)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // (see the loader's barrier)
        ST_T(if (t >= 0) tw_bar += __builtin_readcyclecounter() - tb;)
    }
#ifdef STREAM_TIMING
    if (lane == 0 && (blockIdx.x == 3 || blockIdx.x == 100))
        printf("stream L=%d even=%d wg %d layer %d half %d: steps %d total %llu work %llu barrier %llu cycles\n", L, (int)EVEN, (int)blockIdx.x, lw, HALF, NT,
               __builtin_readcyclecounter() - tw0, tw_work, tw_bar);
#endif
    if (F16 && !p.rg.gated && p.rg.flag && st.amax >= 0x47000000u) *p.rg.flag = 1u;   // a stored magnitude >= 32768 (or a NaN): the fp16 range guard
}

template <bool F16, int L, bool EVEN>
__global__ __launch_bounds__(512, 2) void conv3x3_stream_kernel(StreamConvParams p) {
    using C = StreamCfg<L>;
    using M = StreamLds<L, EVEN>;
    extern __shared__ __align__(16) char lds[];
    if (range_gate_closed(p.rg)) return;
    if ((unsigned)reinterpret_cast<uintptr_t>(lds) != 0u) __builtin_trap();   // LDS is addressed through absolute 32-bit integers
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);          // 0 .. 7: waves w and w + 4 share a SIMD (the dispatcher deals waves to SIMDs cyclically)
    const int lw = w & 3, half = w >> 2;
    const int S0 = (int)blockIdx.x * p.span;
    if (S0 >= p.total) return;                                       // (whole workgroup)
    const int S1 = min(S0 + p.span, p.total);
    const int H = p.Ws + 1;
    const int X0 = S0 - C::BACK;
    const int NT = (S1 - (X0 - (L - 1) * C::LAG) + C::S - 1) / C::S;   // steps until the last layer has covered the span

    // ---- once per workgroup: the zero region, border tables
    for (int t = tid; t < M::ZERO_BYTES / 16; t += 512) *reinterpret_cast<su32x4*>(lds + M::ZERO + t * 16) = (su32x4){0u, 0u, 0u, 0u};
    for (int t = tid; t < L * 16 * 12; t += 512) {
        const int which = t / (16 * 12), r = t - which * 16 * 12;
        f32x4 bv = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (p.border[which]) bv = *reinterpret_cast<const f32x4*>(p.border[which] + 4 * r);
        *reinterpret_cast<f32x4*>(lds + M::BORDER + 16 * t) = bv;
    }
    if (tid < 64) {   // k-step table: entry [g][s] (s < 14; two spare)
        const int gg = tid >> 4, ss = tid & 15, bi = 4 * ss + gg, tap = bi / 6, cblk = bi - tap * 6, ty = tap / 3, tx = tap - 3 * ty;
        reinterpret_cast<int*>(lds + M::KOFF)[tid] = ss < S_STEPS ? ((ty - 1) * p.Ws + (tx - 1)) * SC + cblk * 16 : 0;
    }
    if (lw < 3) {
        if (half == 0) stream_compute_wave<F16, L, EVEN, 0>(p, lw, X0, S0, S1, NT, lane);
        else stream_compute_wave<F16, L, EVEN, 1>(p, lw, X0, S0, S1, NT, lane);
        return;
    }
    // ---------------------------------------------------------------- the loader (wave 3; wave 7 only keeps the barriers company)
    const float inv_cpc = 1.0f / (float)p.cpc_in;
    constexpr int NREC = C::S / 64;                                  // records per loader lane and step
    si32x4 recv[NREC];
    int recpos[NREC];
#pragma unroll
    for (int j = 0; j < NREC; ++j) {
        recv[j] = (si32x4){0, 0, 0, 0};
        recpos[j] = 0;
    }
    bool have_rec = false, have_res = false;
    su32x4 resv[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) resv[i] = (su32x4){0u, 0u, 0u, 0u};
    // the storer (wave 7, odd-first runs of three): one step behind the layers, it copies what they finished to memory as whole 16-byte pieces of
    // whole 96-byte cells -- x_{a+1} out of ring 2 (the next run's residual, in the input's own layout) and the last layer's output out of ring 3
    // (cells of the consumer's layout, from the records).  The compute waves' 8-byte stores were store-ISSUE bound (-15 % without them).
    auto store_step = [&](int tprev) {
        if (!M::STORER || tprev < 0) return;
#pragma unroll
        for (int which = 0; which < 2; ++which) {
            const int P = X0 - (which == 0 ? 1 : 2) * C::LAG + tprev * C::S;       // what layer 1 / layer 2 covered in step tprev
            if (P + C::S <= S0 || P >= S1) continue;                                 // (uniform) nothing of this workgroup's own span
            const int slot = s_mod(P - X0, which == 0 ? M::R2 : (M::R3 > 0 ? M::R3 : 1));
            const int ring = which == 0 ? M::MAP2 : M::MAP3;
            char* const dstp = reinterpret_cast<char*>(which == 0 ? p.out2 : p.out);
#pragma unroll
            for (int i = 0; i < C::S * 6 / 64; ++i) {
                const int c = i * 64 + lane;
                const int cell = c / 6, qd = c - cell * 6;
                const int pos = P + cell;
                const su32x2 e = s_read8(M::REC + (((pos - X0) & (C::RREC - 1)) << 4));      // {tap mask word, output cell}
                const su32x4 v = *reinterpret_cast<s_lds_u32x4_cptr>((unsigned)(ring + (slot + cell) * SC + qd * 16));
                const bool live = pos >= S0 && pos < S1 && ((e[0] >> 13) & 1);
                const size_t dcell = which == 0 ? (size_t)pos : (size_t)e[1];
                if (live) *reinterpret_cast<su32x4*>(dstp + dcell * SC + qd * 16) = v;
            }
        }
    };
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // (the workgroup's set-up barrier)
    for (int t = S_T_START; t < NT; ++t) {
        if (half == 1) store_step(t - 1);
        if (half == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // what the previous iteration requested has landed (LDS ring 0) / arrived (records)
            if (have_rec) {
#pragma unroll
                for (int j = 0; j < NREC; ++j)
                    *reinterpret_cast<s_lds_i32x4_ptr>((unsigned)(M::REC + (((recpos[j] - X0) & (C::RREC - 1)) << 4))) = recv[j];
            }
            if (L == 3 && EVEN) {
                // the first layer's residual x_{a-2}: what the previous iteration fetched is parked for step t + 1 (by step parity); then the cells of step
                // t + 2 -- whose records this wave has just written -- are fetched through the records' residual cell: whole 96-byte cells, 16 bytes per lane
                if (have_res) {
#pragma unroll
                    for (int i = 0; i < 6; ++i)
                        *reinterpret_cast<s_lds_u32x4_ptr>((unsigned)(M::RSTAGE + ((t + 1) & 1) * M::RSTAGE_STEP + (i * 64 + lane) * 16)) = resv[i];
                }
                have_res = t + 2 >= 0;
                if (have_res) {
                    const char* const resp = reinterpret_cast<const char*>(p.res);
                    const int PR = X0 + (t + 2) * C::S;
#pragma unroll
                    for (int i = 0; i < 6; ++i) {
                        const int c = i * 64 + lane;
                        const int cell = c / 6, qd = c - cell * 6;
                        const int pos = PR + cell;
                        const si32x4 e = *reinterpret_cast<s_lds_i32x4_cptr>((unsigned)(M::REC + (((pos - X0) & (C::RREC - 1)) << 4)));
                        resv[i] = (su32x4){0u, 0u, 0u, 0u};
                        if ((unsigned)pos < (unsigned)p.total && ((e[0] >> 13) & 1)) resv[i] = *reinterpret_cast<const su32x4*>(resp + (size_t)e[2] * SC + qd * 16);
                    }
                }
            }
            // ring 0: cells [PD, PD + S) of the input tensor, a flat copy (cells outside the tensor are clamped: they are never tapped)
            const int PD = X0 + (t + 3) * C::S;
            if (PD < S1 + (L * H) + 64 && PD + C::S > 0) {
                const int slot = s_mod(PD - X0, C::R0);              // (a multiple of S, which divides R0: a segment never wraps)
                const char* src = reinterpret_cast<const char*>(p.in);
#pragma unroll
                for (int i = 0; i < C::S * 6 / 64; ++i) {
                    const int c = i * 64 + lane;
                    const int cell = c / 6, qd = c - cell * 6;
                    const int q = min(max(PD + cell, 0), p.total - 1);
                    const __attribute__((address_space(1))) void* gsrc = (const __attribute__((address_space(1))) void*)(src + (size_t)q * SC + qd * 16);
                    const int dst = M::MAP0 + slot * SC + i * 1024;
                    __builtin_amdgcn_global_load_lds(gsrc, (__attribute__((address_space(3))) void*)(lds + dst), 16, 0, 0);
                    // guard copies (64 cells = six 1 KB pieces each): the ring's first 64 cells again behind it, its last 64 again in front of it
                    if (slot == 0 && i < 6) __builtin_amdgcn_global_load_lds(gsrc, (__attribute__((address_space(3))) void*)(lds + dst + C::R0 * SC), 16, 0, 0);
                    if (slot + C::S == C::R0 && i >= C::S * 6 / 64 - 6) __builtin_amdgcn_global_load_lds(gsrc, (__attribute__((address_space(3))) void*)(lds + dst - C::R0 * SC), 16, 0, 0);
                }
            }
            // records of positions [X0 + (t + 3) S, + S): requested now, written to the ring in the next iteration, readable from step t + 2 on -- one
            // step before the first layer gets there (it requests a residual that comes from memory a step ahead)
            have_rec = t + 3 >= 0;
            if (have_rec) {
#pragma unroll
                for (int j = 0; j < NREC; ++j) {
                    const int pr = X0 + (t + 3) * C::S + j * 64 + lane;
                    recpos[j] = pr;
                    int q;
                    const int bclip = s_fdiv(min(max(pr, 0), p.total - 1), p.cpc_in, inv_cpc, q);
                    const si32x4 e = *reinterpret_cast<const si32x4*>(p.postab + 4 * q);
                    recv[j] = (si32x4){e[0], bclip * p.cpc_out + e[1], bclip * p.cpc_res + e[2], 0};
                }
            }
        }
        // The step barrier orders LDS only: this wave's ring / record writes are complete (lgkmcnt), then everybody meets.  NOT __syncthreads():
        // that also waits for the wave's outstanding GLOBAL operations (vmcnt(0)) -- the last layer's stores, a residual requested for the next
        // step, the loader's copies for the step after next.  The loader waits for its own copies itself, one iteration after issuing them.
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    if (half == 1) store_step(NT - 1);          // what the layers wrote in the last step (visible behind the last barrier)
}

bool conv3x3_stream_supported(int C, int Ws) { return (C + 7) / 8 * 8 == 48 && Ws >= 1 && Ws + 1 <= 48; }

template <bool F16, int L, bool EVEN>
static hipError_t launch_stream_k(const StreamConvParams& p, unsigned grid, hipStream_t s) {
    auto k = conv3x3_stream_kernel<F16, L, EVEN>;
    static DeviceOnce attr_once;
    if (attr_once.first()) {
        hipError_t e = allow_big_lds_at_base_zero(reinterpret_cast<const void*>(k));
        if (e != hipSuccess) return e;
    }
    constexpr size_t lds_bytes = StreamLds<L, EVEN>::BYTES;
    hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds_bytes, s, p);
    return hipGetLastError();
}

hipError_t launch_conv3x3_stream(const StreamConvParams& p_in, int C, int n_cu, hipStream_t s) {
    if (p_in.total <= 0) return hipSuccess;
    StreamConvParams p = p_in;
    const bool even = p.first_even != 0;
    const int L = p.n_layers;
    if (!conv3x3_stream_supported(C, p.Ws) || (L != 1 && L != 3) || (long long)p.total + 4096 >= (1 << 24) || (long long)p.total * 96 >= (1LL << 31) ||
        (long long)p.B * std::max(p.cpc_out, p.cpc_res) * 96 >= (1LL << 31) || p.cpc_in < 1)
        return hipErrorInvalidValue;
    // which tensors a run of this shape must bring: the first layer's residual comes from memory when it is even; x_{a+1} of an odd-first triple goes out
    if ((even ? p.res == nullptr : p.res != nullptr) || ((L == 3 && !even) ? p.out2 == nullptr : p.out2 != nullptr)) return hipErrorInvalidValue;
    const int step = L == 3 ? StreamCfg<3>::S : StreamCfg<1>::S;
    const int min_span = 4 * step;                                    // shorter spans are all pipeline fill
    int grid = std::max(1, std::min(std::max(n_cu, 1), (p.total + min_span - 1) / min_span));
    p.span = ((p.total + grid - 1) / grid + 15) / 16 * 16;
    grid = (p.total + p.span - 1) / p.span;
    if (L == 3) {
        if (p.f16) return even ? launch_stream_k<true, 3, true>(p, grid, s) : launch_stream_k<true, 3, false>(p, grid, s);
        return even ? launch_stream_k<false, 3, true>(p, grid, s) : launch_stream_k<false, 3, false>(p, grid, s);
    }
    if (even) return hipErrorInvalidValue;     // (an even single layer takes its residual from memory: the tile kernel's job)
    return p.f16 ? launch_stream_k<true, 1, false>(p, grid, s) : launch_stream_k<false, 1, false>(p, grid, s);
}

}  // namespace kws
