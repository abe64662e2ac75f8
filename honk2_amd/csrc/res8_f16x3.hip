// res8 forward, fully fused, with fp32-accurate products formed from THREE fp16 x fp16 terms ("f16x3").
//
// Same function, dataflow and work split as res8_bf16x6.hip (reference model/resnet.py:38-60 for config/resnet/res8.json),
// with a cheaper exact-product scheme for conv_1..conv_6:
//   * fp16 has an 11-bit significand, so an fp32 value splits into TWO fp16 parts x = x1 + x2 with 22 bits kept (bf16: three
//     parts for 24).  The product is accumulated as a2 b1 + a1 b2 + a1 b1 on v_mfma_f32_16x16x32_f16 (same rate as the bf16
//     MFMA); each fp16 x fp16 product is exact in the fp32 accumulator and what is dropped is <= 3 * 2^-22 |ab| -- CPU
//     emulation of a res8 layer's GEMM: rms error 8e-8 relative, against 3.6e-7 for an fp32 FMA chain and 2.6e-8 for the
//     six-term bf16 form.  Three MFMAs instead of six per 16x16x32 MACs: the roof for fp32-accurate matrix work becomes
//     2516 / 3 = 839 TFLOP/s.
//   * fp16's narrow exponent range is the catch.  The MFMA keeps fp16 subnormals (tools/f16_denorm_probe.cpp), so a second
//     part below 2^-14 is still exact to 2^-25 absolute: harmless for activations (O(1), their error stays far below the
//     fp32 rounding of the sum) but not for weights (|w| ~ 0.05).  Weights are therefore scaled by a power of two per layer
//     (max |w| * 2^S in [128, 256)) before the split and the accumulator is scaled back by 2^-S -- exact, and free: the
//     factor is folded into the BatchNorm scale (odd layers) or into the FMA that adds the residual (even layers).
//     fp16 overflows at 65 504.  Trained models stay at O(10), but nothing in the reference forbids more (a channel with a tiny
//     running variance multiplies by 300): before a map is written the workgroup takes its maximum magnitude and, above 2^14,
//     stores it scaled down by a power of two that the consumer multiplies back (range guard below; a test drives logits to
//     4e8 through it).  Only the input features themselves must stay below 65 504 (log-mel features are <= ~30);
//     KWS_RES8_IMPL=bf16x6 selects the six-term bf16 kernel, which has fp32's range everywhere.
//   * LDS holds the map as [384 cells][2 parts][48 channels] fp16: 192 B per cell, 72 KB instead of 108; B fragments are
//     two ds_read_b128 per position tile; the fp32 -> 2 x fp16 split happens once per output element in the epilogue.
//   conv_0 (K = 9) keeps the fp32-input MFMA path.
#include "kws_internal.h"

#include <cmath>
#include <cstring>

namespace kws {

namespace {
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// Channel-block-major map: [2 parts][6 blocks of 8 channels][384 cells] x 16 B; cell(y, x) = 14 (y + 1) + x + 1 (one shared
// zero halo column between rows).  A ds_read_b128 of a B fragment is served in four 16-lane groups, each of which holds
// all 16 positions of the tile (8 lanes of one k-group + the complementary 8 of the next k-group, whose plane starts a
// multiple of 256 B further on), so it is conflict-free iff the tile's 16 cells are distinct mod 16.  Consecutive
// positions never are (a 13-wide row wraps inside every tile: the cell-per-position layouts before this one cost
// 2.0x - 2.8x the conflict-free LDS cycles on reads, 4.6x - 8x on the epilogue's 8-byte stores, and
// SQ_LDS_BANK_CONFLICT was 54 % of the LDS pipe's active cycles), but the MFMA does not care which positions share a
// tile: tiles 0..19 take ONE position from each residue class of cell index mod 16 (18..22 cells per class; a class that
// has run out leaves a pad lane, which clones another lane of its tile: same address = broadcast) and tile 20 takes
// the 15 positions left over.  Modelled LDS cycles per layer: reads 2 572 (ideal 2 352; 4 664 before), stores 1 056
// (504; 2 328).  The table comes from tools/r8_tiles.py, which also holds the bank model.
constexpr int PLANE_B = 384 * 16;                     // one (part, channel block) plane: 6 144 B
constexpr int PART_B = 6 * PLANE_B;                   // 36 864
constexpr int MAP_BYTES = 2 * PART_B;                 // 73 728
// position (0..324, row-major over the 25 x 13 map) of lane pcol of tile t at [16 t + pcol]; 0x8000 marks a pad lane
__device__ const unsigned short R8H_POS[21 * 16] = {
    0x0001, 0x0002, 0x0003, 0x0004, 0x0005, 0x0006, 0x0007, 0x0008, 0x0009, 0x000a, 0x000b, 0x000c, 0x001b, 0x000d, 0x000e, 0x0000,
    0x0010, 0x0011, 0x0012, 0x0013, 0x0014, 0x0015, 0x0016, 0x0017, 0x0018, 0x0019, 0x0028, 0x001a, 0x002a, 0x001c, 0x001d, 0x000f,
    0x001f, 0x0020, 0x0021, 0x0022, 0x0023, 0x0024, 0x0025, 0x0026, 0x0035, 0x0027, 0x0037, 0x0029, 0x0039, 0x002b, 0x002c, 0x001e,
    0x002e, 0x002f, 0x0030, 0x0031, 0x0032, 0x0033, 0x0042, 0x0034, 0x0044, 0x0036, 0x0046, 0x0038, 0x0048, 0x003a, 0x003b, 0x002d,
    0x003d, 0x003e, 0x003f, 0x0040, 0x004f, 0x0041, 0x0051, 0x0043, 0x0053, 0x0045, 0x0055, 0x0047, 0x0057, 0x0049, 0x004a, 0x003c,
    0x004c, 0x004d, 0x005c, 0x004e, 0x005e, 0x0050, 0x0060, 0x0052, 0x0062, 0x0054, 0x0064, 0x0056, 0x0066, 0x0058, 0x0059, 0x004b,
    0x0069, 0x005b, 0x006b, 0x005d, 0x006d, 0x005f, 0x006f, 0x0061, 0x0071, 0x0063, 0x0073, 0x0065, 0x0083, 0x0067, 0x0076, 0x005a,
    0x0078, 0x006a, 0x007a, 0x006c, 0x007c, 0x006e, 0x007e, 0x0070, 0x0080, 0x0072, 0x0090, 0x0074, 0x0092, 0x0075, 0x0085, 0x0068,
    0x0087, 0x0079, 0x0089, 0x007b, 0x008b, 0x007d, 0x008d, 0x007f, 0x009d, 0x0081, 0x009f, 0x0082, 0x00a1, 0x0084, 0x0094, 0x0077,
    0x0096, 0x0088, 0x0098, 0x008a, 0x009a, 0x008c, 0x00aa, 0x008e, 0x00ac, 0x008f, 0x00ae, 0x0091, 0x00b0, 0x0093, 0x00a3, 0x0086,
    0x00a5, 0x0097, 0x00a7, 0x0099, 0x00b7, 0x009b, 0x00b9, 0x009c, 0x00bb, 0x009e, 0x00bd, 0x00a0, 0x00bf, 0x00a2, 0x00b2, 0x0095,
    0x00b4, 0x00a6, 0x00c4, 0x00a8, 0x00c6, 0x00a9, 0x00c8, 0x00ab, 0x00ca, 0x00ad, 0x00cc, 0x00af, 0x00ce, 0x00b1, 0x00c1, 0x00a4,
    0x00d1, 0x00b5, 0x00d3, 0x00b6, 0x00d5, 0x00b8, 0x00d7, 0x00ba, 0x00d9, 0x00bc, 0x00db, 0x00be, 0x00eb, 0x00c0, 0x00de, 0x00b3,
    0x00e0, 0x00c3, 0x00e2, 0x00c5, 0x00e4, 0x00c7, 0x00e6, 0x00c9, 0x00e8, 0x00cb, 0x00f8, 0x00cd, 0x00fa, 0x00cf, 0x00ed, 0x00c2,
    0x00ef, 0x00d2, 0x00f1, 0x00d4, 0x00f3, 0x00d6, 0x00f5, 0x00d8, 0x0105, 0x00da, 0x0107, 0x00dc, 0x0109, 0x00dd, 0x00fc, 0x00d0,
    0x00fe, 0x00e1, 0x0100, 0x00e3, 0x0102, 0x00e5, 0x0112, 0x00e7, 0x0114, 0x00e9, 0x0116, 0x00ea, 0x0118, 0x00ec, 0x010b, 0x00df,
    0x010d, 0x00f0, 0x010f, 0x00f2, 0x011f, 0x00f4, 0x0121, 0x00f6, 0x0123, 0x00f7, 0x0125, 0x00f9, 0x0127, 0x00fb, 0x011a, 0x00ee,
    0x011c, 0x00ff, 0x012c, 0x0101, 0x012e, 0x0103, 0x0130, 0x0104, 0x0132, 0x0106, 0x0134, 0x0108, 0x0136, 0x010a, 0x0129, 0x00fd,
    0x0139, 0x010e, 0x013b, 0x0110, 0x013d, 0x0111, 0x013f, 0x0113, 0x0141, 0x0115, 0x0143, 0x0117, 0x8139, 0x0119, 0x8139, 0x010c,
    0x811d, 0x011d, 0x811d, 0x011e, 0x811d, 0x0120, 0x811d, 0x0122, 0x811d, 0x0124, 0x811d, 0x0126, 0x811d, 0x0128, 0x811d, 0x011b,
    0x012b, 0x013a, 0x012d, 0x013c, 0x012f, 0x013e, 0x0131, 0x0140, 0x0133, 0x0142, 0x0135, 0x0144, 0x0137, 0x012a, 0x0138, 0x812b,
};

constexpr int FS = 41;                                // staged feature row stride (fp32 words)
constexpr int FEAT_BYTES = ((102 * FS * 4 + 15) / 16) * 16;
static_assert(FEAT_BYTES <= MAP_BYTES, "the feature map is staged inside the (idle) activation map");
constexpr int RED_OFF = MAP_BYTES;                    // fp32 words from here on
constexpr int BNT_WORDS = 96;                       // the last BatchNorm's scale[48], shift[48] (the others are folded into the weights)
constexpr int NEXT_OFF = RED_OFF + (4 * 48 + 48 + BNT_WORDS) * 4;   // one word: the clip this workgroup takes next
constexpr int POS_OFF = NEXT_OFF + 16;                // the position table (336 x 2 B)
constexpr int X_LDS_BYTES = POS_OFF + 21 * 16 * 2;

__device__ __forceinline__ float relu1(float x) {
    const int b = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, b > 0 ? b : 0);
}
__device__ __forceinline__ unsigned pack2(float a, float b) {
    const f16x2 v = {(_Float16)a, (_Float16)b};
    return __builtin_bit_cast(unsigned, v);
}

// second fp16 parts of (a, b) given their packed first parts h: fp16(a - float(h.lo)), fp16(b - float(h.hi)).  The
// difference is exact in fp32, so one mixed-precision FMA per value (f16 source, f32 addend, f16 result) gives the same
// bits as convert-back, subtract, convert (which costs three instructions per value).
__device__ __forceinline__ unsigned resid2(unsigned h, float a, float b) {
    unsigned l;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(h), "v"(a));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(h), "v"(b));
    return l;
}
// split 4 consecutive channels into two fp16 parts and store them at LDS address `addr` (+ PART_B for part 2)
typedef u32x2 __attribute__((address_space(3))) * lds_u32x2_ptr;
__device__ __forceinline__ void store_split(int addr, f32x4 v) {
    u32x2 h, m;
    h[0] = pack2(v[0], v[1]);
    h[1] = pack2(v[2], v[3]);
    m[0] = resid2(h[0], v[0], v[1]);
    m[1] = resid2(h[1], v[2], v[3]);
    *reinterpret_cast<lds_u32x2_ptr>((unsigned)addr) = h;
    *reinterpret_cast<lds_u32x2_ptr>((unsigned)(addr + PART_B)) = m;
}

struct XCtx {
    char* lds;
    float* red;
    float* mvec;
    const float* bnt;
    int tid, lane, w, g, pcol;
    int cm[3];   // channel tile behind the wave's slot m (wave-uniform): slot 0 is the tile the wave also owns on position tile 20
    int qa[6];   // byte address of this lane's k-slot base in each of the wave's 6 position tiles (see "K order" below)
    int es[3];   // store address of slot m's four channels, relative to qa[j]
    int dB, dC, dD;   // what kinds B, C, D add to qa[j]
    int padmask; // bit j: this lane of tile j is a pad lane (a clone of another lane: stored again, but counted once)
    bool xvalid;
};

// fp16 range guard.  Activations go to LDS as fp16 pairs, so a value beyond 65504 would overflow.  Before a layer's output is
// written, the workgroup takes the maximum magnitude of the whole map (one value per wave through LDS, riding on the barrier
// that is there anyway) and, only if it exceeds 2^14, stores the map scaled down by the power of two that brings it back;
// the consumer multiplies its accumulators by the same power of two (exact both ways).  Trained models never get there.
// magnitudes are compared as bit patterns (monotone for non-negative floats); six DPP steps leave the wave's maximum in lane
// 63 (quad swaps, half-row and row mirrors, then the two row broadcasts of GFX9)
__device__ __forceinline__ unsigned wave_umax(unsigned v) {
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, true));   // row_half_mirror
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, true));   // row_mirror
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, true));   // row_bcast15 -> rows 1, 3
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, true));   // row_bcast31 -> rows 2, 3
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
// one slot per wave; consecutive guards alternate between two groups of four so that a group is never written while the other
// workgroup members may still be reading it (a full barrier lies between a write and the next write of the same group)
__device__ __forceinline__ void guard_push(unsigned* group, int w, int lane, float amax) {
    const unsigned m = wave_umax(__builtin_bit_cast(unsigned, amax));
    if (lane == 0) group[w] = m;
}
__device__ __forceinline__ float guard_read(const unsigned* group) {
    return __builtin_bit_cast(float, max(max(group[0], group[1]), max(group[2], group[3])));
}
__device__ __forceinline__ int range_shift(float mx) {   // 0 for mx <= 2^14 (and for NaN), else ceil(log2(mx)) - 14
    int e = 0;
    if (mx > 16384.f) {
        (void)frexpf(mx, &e);        // mx = f * 2^e, f in [0.5, 1)
        e -= 14;
        e = e > 100 ? 100 : e;       // inf: give up gracefully (the values were not finite anyway)
    }
    return e;
}

struct AFrags {
    u32x4 a[3][2];   // [slot][part]
};
struct BFrag {
    u32x4 p[2];      // the two fragments of one position tile and k-step (regular steps: the two fp16 parts of the same cells)
};

// ---- K order.  K = 9 taps x 6 blocks of 8 input channels = 54 blocks, four per k-step (one per lane group g).  The blocks
// are dealt to the steps so that a lane's LDS address is  qa[j] + (per-kind lane constant) + (compile-time immediate):
//   kind A, steps 0..8   taps (3 i, 3 i + 1) x channel-block pair q:  g -> block 2 q + (g & 1) of tap 3 i + (g >> 1)
//   kind B, steps 9..11  taps (2, 5) x pair q:                        g -> block 2 q + (g & 1) of tap 2 + 3 (g >> 1)
//   kind C, step 12      tap 8, blocks 0..3:                          g -> block g
//   kind D, step 13      tap 8, blocks 4, 5 -- half a step, so its a1 b1 and a1 b2 terms share ONE MFMA: groups 0, 1 read
//                        part 1 of the cells, groups 2, 3 part 2 of the same two blocks, against a fragment that carries a1
//                        twice; a2 b1 is a second MFMA on a regular part-1 read (2 MFMAs per tile instead of 3).
// qa[j] already holds kind A's lane constant (g & 1) PLANE_B + (g >> 1) 16 and a bias of -QBIAS (tap 0 reaches back 15
// cells; ds_read offsets are unsigned); kinds B, C, D add one per-lane register each (dB, dC, dD).  With the step loop
// fully unrolled the k-loop carries no address arithmetic beyond those adds (4.5 of 13.5 steps): 9 vector instructions per
// k-step before (two integer divisions per step pair), 2 now.
constexpr int QBIAS = (R8_RS + 1) * 16;
constexpr int tap_off(int t) { return ((t / 3 - 1) * R8_RS + (t % 3 - 1)) * 16; }
constexpr int step_kind(int s) { return s < 9 ? 0 : (s < 12 ? 1 : (s == 12 ? 2 : 3)); }
constexpr int step_imm(int s) {
    return s < 9    ? 2 * (s % 3) * PLANE_B + tap_off(3 * (s / 3)) + QBIAS
           : s < 12 ? 2 * (s - 9) * PLANE_B + tap_off(2) + QBIAS
           : s == 12 ? tap_off(8) + QBIAS
                     : 4 * PLANE_B + tap_off(8) + QBIAS;
}
static_assert(step_imm(8) + PART_B < 65536 && step_imm(0) >= 0, "ds_read offsets are 16 bits, unsigned");
constexpr int KSTEPS = R8X_KSTEPS;                    // 14
constexpr int A_FRAG_B = 64 * 16;                     // one A fragment: 1 KB
constexpr int A_STEP_B = 3 * 2 * A_FRAG_B;            // a k-step's fragments: [channel tile][part][lane] x 16 B
constexpr int A_LAYER_B = R8H_ASTEPS * A_STEP_B;

// (c.qa[] are full 32-bit LDS addresses: one VGPR + a 16-bit immediate per ds_read_b128, nothing for the compiler to re-derive)
typedef const u32x4 __attribute__((address_space(3))) * lds_u32x4_ptr;
__device__ __forceinline__ u32x4 lds_read16(int addr) { return *reinterpret_cast<lds_u32x4_ptr>((unsigned)addr); }
__device__ __forceinline__ void load_b(BFrag& b, const int (&qa)[6], int dB, int dC, int dD, int s, int j) {
    const int k = step_kind(s), imm = step_imm(s);
    if (k == 3) {
        b.p[0] = lds_read16((qa[j] + dD) + imm);   // (part 1 | part 2) of blocks 4, 5
        b.p[1] = lds_read16(qa[j] + imm);          // part 1 (groups 2, 3 meet zero weights)
    } else {
        const int base = k == 0 ? qa[j] : qa[j] + (k == 1 ? dB : dC);
        b.p[0] = lds_read16(base + imm);
        b.p[1] = lds_read16(base + imm + PART_B);
    }
}
#ifndef R8H_WAITS
#define R8H_WAITS 0
#endif
// the six fragments of k-step `sidx` (0 .. 6 * 14 - 1 over the layers): scalar base + per-slot scalar offset + lane offset
__device__ __forceinline__ void load_a(AFrags& f, __amdgpu_buffer_rsrc_t rs, int voff, int sbase, const int (&om)[3]) {
#if R8H_WAITS   // requested so that the FIRST fragment a k-step uses (slot 0, part 1: see mf_tile) is the LAST to arrive: one vmcnt wait per k-step, not three
#pragma unroll
    for (int m = 2; m >= 0; --m)
#pragma unroll
        for (int pt = 1; pt >= 0; --pt)
            f.a[m][pt] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff + pt * A_FRAG_B, sbase + om[m], 0));
#else
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int pt = 0; pt < 2; ++pt)
            f.a[m][pt] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff + pt * A_FRAG_B, sbase + om[m], 0));
#endif
}

#define MF(A_, B_, C_) C_ = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, A_), __builtin_bit_cast(f16x8, B_), C_, 0, 0, 0)

#ifndef R8H_ORDER    // MFMA order inside a (k-step, position tile): 0 chain-major, 1 term-major
#define R8H_ORDER 0
#endif
// the MFMAs of one (k-step, position tile, slot): three-term product, small terms first
template <int TERMS>
__device__ __forceinline__ void mf_tile(int s, const u32x4 (&a)[2], const BFrag& b, f32x4& acc, bool zig = false) {
    if (step_kind(s) == 3) {
        if (TERMS >= 3) MF(a[1], b.p[1], acc);   // a2 b1
        MF(a[0], b.p[0], acc);                   // a1 b1 + a1 b2
    } else {
        if (TERMS >= 3) {
#if R8H_ORDER == 2      // consecutive MFMAs share an operand: (a2 b1) (a1 b1) (a1 b2), and back again for the next slot
            if (zig) {
                MF(a[0], b.p[1], acc);
                MF(a[0], b.p[0], acc);
                MF(a[1], b.p[0], acc);
            } else {
                MF(a[1], b.p[0], acc);
                MF(a[0], b.p[0], acc);
                MF(a[0], b.p[1], acc);
            }
            return;
#endif
#if R8H_WAITS          // the chain starts on the fragment parts that arrive last (a part 1, b part 2): one LDS wait per tile, not two
            MF(a[0], b.p[1], acc);
            MF(a[1], b.p[0], acc);
#else
            MF(a[1], b.p[0], acc);
            MF(a[0], b.p[1], acc);
#endif
        }
        MF(a[0], b.p[0], acc);
    }
}

// A/B knobs of this file (defaults = what measured best):
//   R8H_PRIO      wave priority inside the k-loops (s_setprio)
#ifndef R8H_PRIO
#define R8H_PRIO 0
#endif
#ifndef R8H_EPRIO   // wave priority OUTSIDE the k-loops (epilogues, conv_0, staging, tail)
#define R8H_EPRIO 0
#endif
#ifndef R8H_FENCE    // 1: a scheduling fence after every accumulator's chain of MFMAs -- hipcc's post-RA scheduler otherwise deals the nine MFMAs of a
                     // tile out term by term; a dependent MFMA issued right behind its predecessor takes C from the pipe, not from the register
                     // file, and the kernel runs at its power cap: 11.10 -> 10.73 ms (r3; term-major ON PURPOSE: 11.27)
#define R8H_FENCE 1
#endif
#ifndef R8H_BDEPTH   // position tiles of look-ahead of the k-loop's LDS reads
#define R8H_BDEPTH 1
#endif
#ifndef R8H_ABLATE      // 1: KWS_R8_DEBUG bits 4 / 8 drop the k-loops' LDS operand reads / weight loads (timing experiments; results are wrong)
#define R8H_ABLATE 0
#endif
// (r5) what fewer MFMAs would buy at the power cap, measured instead of argued (tools/r8_levers.sh; results wrong by construction):
//   R8H_ABLATE_K = 13 runs k-steps 0..12 only (no kind-D half step: 39 instead of 41 MFMAs per tile, layer and slot -- what a dense K of 13 steps
//   would execute), 12 drops step 12 too (36); R8H_ABLATE_NOX = 1 skips position tile 20's MFMAs (N 336 -> 320: what a pad-free tiling of the
//   325 positions could save at most)
#ifndef R8H_ABLATE_K
#define R8H_ABLATE_K R8X_KSTEPS
#endif
#ifndef R8H_ABLATE_NOX
#define R8H_ABLATE_NOX 0
#endif

// phase timestamps for tools/r8_phases.py: build with -DR8H_TIMING (they overwrite the consumed feature rows)
#ifdef R8H_TIMING
#define R8H_TS_DECL unsigned long long ts[16] = {}, rt0 = __builtin_amdgcn_s_memrealtime();
#define R8H_TS(i) ts[i] = __builtin_readcyclecounter();
#else
#define R8H_TS_DECL
#define R8H_TS(i)
#endif
#ifdef R8H_TIMING
#define R8H_LTS(i) if (layer == 2) ts[i] = __builtin_readcyclecounter();
#define R8H_TSARG , unsigned long long (&ts)[16]
#define R8H_TSPASS , ts
#else
#define R8H_LTS(i)
#define R8H_TSARG
#define R8H_TSPASS
#endif

// One conv_i + epilogue (i = layer + 1).  `fa0` arrives holding k-step 0's weight fragments (requested before the previous
// epilogue and its barriers) and leaves holding the next layer's.
template <int TERMS, bool EVEN>
__device__ __forceinline__ void x_layer(const Res8hParams& p, XCtx& c, __amdgpu_buffer_rsrc_t ars, int& avoff, const int (&om)[3],
                                        const int layer, f32x4 (&acc)[5][3], f32x4& accx, f32x4 (&prev)[5][3], f32x4& prevx,
                                        int& shift, AFrags (&fa)[2] R8H_TSARG) {
    constexpr bool even = EVEN;                  // reference layer i = layer + 1: residual on even i (odd `layer`)
    const bool last = layer == R8_LAYERS - 1;
    const int g = c.g;
    const f32x4 zero = (f32x4){0.f, 0.f, 0.f, 0.f};
    constexpr int NB = R8H_BDEPTH + 1;
    BFrag bb[NB];   // ring over the (k-step, position tile) sequence
    if (R8H_ABLATE)
        for (int i = 0; i < NB; ++i) bb[i].p[0] = bb[i].p[1] = (u32x4){0x3c003c00u, 0x38003800u, 0x3a003a00u, 0x34003400u};
    if (R8H_PRIO | R8H_EPRIO) __builtin_amdgcn_s_setprio(R8H_PRIO);
    R8H_LTS(8)
    const int sb = layer * A_LAYER_B;
    // per-lane offsets are re-materialised per layer from opaque copies: left alone, LICM hoists every sum built from them
    // (store addresses, k-slot bases, table addresses: ~40 registers) out of the layer loop and the allocator spills them
    // (in place: a separate copy would keep both values alive across the loop)
    asm volatile("" : "+v"(c.dB), "+v"(c.dC), "+v"(c.dD), "+v"(c.es[0]), "+v"(c.es[1]), "+v"(c.es[2]), "+v"(avoff));
    const int dB = c.dB, dC = c.dC, dD = c.dD, es[3] = {c.es[0], c.es[1], c.es[2]};
    if (!(KWS_DBG(p.debug & 2))) {
        // B fragments are fetched R8H_BDEPTH position tiles ahead (2 ds_read_b128 per tile; the LDS counter is 4 bits, a
        // whole k-step's reads cannot be outstanding); A fragments one k-step ahead, the last step requests the next layer's
        // first.
        if (!(R8H_ABLATE && (KWS_DBG(p.debug & 4))))
#pragma unroll
            for (int t = 0; t < R8H_BDEPTH; ++t) load_b(bb[t % NB], c.qa, dB, dC, dD, t / 6, t % 6);
        constexpr int KRUN = R8H_ABLATE_K;    // = KSTEPS outside the lever experiments
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            const AFrags& fc = fa[s & 1];
            if (s >= KRUN) {   // (lever experiments only) a dropped step: no operands, no MFMAs; the next layer's first fragments still land in fa[0]
                if (s == KSTEPS - 1) load_a(fa[0], ars, avoff, last ? sb : sb + A_LAYER_B, om);
                continue;
            }
            if (!(R8H_ABLATE && (KWS_DBG(p.debug & 8)))) {
                if (s + 1 < KRUN) load_a(fa[(s + 1) & 1], ars, avoff, sb + (s + 1) * A_STEP_B, om);
                else if (KRUN == KSTEPS) load_a(fa[(s + 1) & 1], ars, avoff, last ? sb : sb + A_LAYER_B, om);   // (last layer: a harmless re-read)
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const int t = 6 * s + j, tn = t + R8H_BDEPTH;
                const BFrag& bcur = bb[t % NB];
                if (!(R8H_ABLATE && (KWS_DBG(p.debug & 4))) && tn < 6 * KRUN) load_b(bb[tn % NB], c.qa, dB, dC, dD, tn / 6, tn % 6);
                __builtin_amdgcn_sched_barrier(0);
                if (j < 5) {
                    if (s == 0) acc[j][0] = acc[j][1] = acc[j][2] = zero;
                    if (R8H_ORDER != 1 || TERMS < 3) {   // chain-major: the three terms of an accumulator back to back
#pragma unroll
                        for (int m = 0; m < 3; ++m) {
                            mf_tile<TERMS>(s, fc.a[m], bcur, acc[j][m], m & 1);
                            if (R8H_FENCE && m < 2) __builtin_amdgcn_sched_barrier(0);
                        }
                    } else {                             // term-major: one B fragment meets the three slots' A fragments in turn
                        const bool kd = step_kind(s) == 3;
#pragma unroll
                        for (int m = 0; m < 3; ++m) { MF(fc.a[m][1], bcur.p[kd ? 1 : 0], acc[j][m]); }
                        if (!kd) {
#pragma unroll
                            for (int m = 0; m < 3; ++m) { MF(fc.a[m][0], bcur.p[1], acc[j][m]); }
                        }
#pragma unroll
                        for (int m = 0; m < 3; ++m) { MF(fc.a[m][0], bcur.p[0], acc[j][m]); }
                    }
                } else if (c.w < 3) {   // wave 3 owns no extra tile
                    if (s == 0) accx = zero;
                    if (!R8H_ABLATE_NOX) mf_tile<TERMS>(s, fc.a[0], bcur, accx);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (s == 0) { R8H_LTS(9) }
            if (s == 6) { R8H_LTS(10) }
        }
    } else {
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
            for (int m = 0; m < 3; ++m) acc[j][m] = zero;
        accx = zero;
    }
    if (R8H_PRIO | R8H_EPRIO) __builtin_amdgcn_s_setprio(R8H_EPRIO);
    R8H_LTS(11)

    // ---- epilogue in fp32: ReLU and, on even i, the residual (reference model/resnet.py:46-55).  BatchNorm i is not applied here: its
    //      scale is folded into conv_{i+1}'s weights and its shift rides on the constant channel (slot 45) of the map, which is set
    //      below (kws_api.cpp, finalize).  Odd i: the map takes relu(acc) as it comes -- the layer's weight scale 2^S stays on it and
    //      1 / 2^S is in the next layer's weights -- so an odd epilogue is ONE instruction per value, and half of what it stores are
    //      exact zeros (operands that do not toggle the matrix pipe, on a kernel that runs at its power cap).  Even i: x = relu(acc)
    //      2^-S + prev_x, stored as it is.
    const float up = shift > 0 ? ldexpf(1.f, shift) : 1.f;   // undo the range guard of the map this layer read (uniform)
    const float inv = p.inv_scale[layer] * up;
    const float kap = p.kappa[layer];
    float amax = 0.f;
    // What the map takes: odd i -- relu(acc), rewritten in place; even i -- the updated residual registers themselves (no copy: a copy
    // cost 60 v_mov_b64 per layer).  The constant channel: odd i selects kappa into slot 45's row; even i needs nothing -- its kappa is 1,
    // conv's row 45 is all zero weights and prev_x's row 45 was set to 1 behind conv_0, so x[45] = 0 * 2^-S + 1.
    f32x4 (&outv)[5][3] = even ? prev : acc;
    f32x4& outx = even ? prevx : accx;
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        if (even) {
#pragma unroll
            for (int j = 0; j < 5; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) prev[j][m][r] = fmaf(relu1(acc[j][m][r]), inv, prev[j][m][r]);
        } else {
#pragma unroll
            for (int j = 0; j < 5; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[j][m][r] = relu1(acc[j][m][r]);
            if (shift > 0) {   // (wave-uniform, never taken for trained models: the input map was stored scaled down)
#pragma unroll
                for (int j = 0; j < 5; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[j][m][r] *= up;
            }
            if (c.cm[m] == 2) {   // (wave-uniform) this slot holds channels 32..47: lane group 3, second row = the constant channel 45
#pragma unroll
                for (int j = 0; j < 5; ++j) acc[j][m][1] = g == 3 ? kap : acc[j][m][1];
            }
        }
        if (m == 0) {   // the extra tile holds slot 0's channels
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (even) {
                    prevx[r] = fmaf(relu1(accx[r]), inv, prevx[r]);
                } else {
                    float v = relu1(accx[r]);
                    if (shift > 0) v *= up;
                    accx[r] = v;
                }
            }
            if (!even && c.cm[0] == 2) accx[1] = g == 3 ? kap : accx[1];
        }
    }
    {   // two values per v_max3_f32, three independent chains (one per slot) instead of one of 32 dependent instructions; every value is >= 0
        float am[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                am[m] = fmaxf(fmaxf(am[m], outv[j][m][0]), outv[j][m][1]);
                am[m] = fmaxf(fmaxf(am[m], outv[j][m][2]), outv[j][m][3]);
            }
        amax = fmaxf(fmaxf(am[0], am[1]), am[2]);
    }
    if (c.xvalid) {
        amax = fmaxf(fmaxf(amax, outx[0]), outx[1]);
        amax = fmaxf(fmaxf(amax, outx[2]), outx[3]);
    }
    unsigned* const ggrp = reinterpret_cast<unsigned*>(c.red) + 4 * ((layer + 1) & 1);   // the reduction buffer is idle until the tail
    if (!last) guard_push(ggrp, c.w, c.lane, amax);
    R8H_LTS(12)

    __syncthreads();  // every wave has finished reading this layer's input map
    R8H_LTS(13)
    if (!last) {
        shift = __builtin_amdgcn_readfirstlane(range_shift(guard_read(ggrp)));   // uniform
        if (shift > 0) {   // never taken for trained models: keep it a (wave-uniform) branch, not selects on every value
            asm volatile("; range guard: scale the map down" ::: "memory");
            const float down = ldexpf(1.f, -shift);   // (scaled copies: the residual registers keep their values)
#pragma unroll
            for (int j = 0; j < 5; ++j)
#pragma unroll
                for (int m = 0; m < 3; ++m) store_split(c.qa[j] + es[m], outv[j][m] * down);
            if (c.xvalid) store_split(c.qa[5] + es[0], outx * down);
        } else {
#pragma unroll
            for (int j = 0; j < 5; ++j)
#pragma unroll
                for (int m = 0; m < 3; ++m) store_split(c.qa[j] + es[m], outv[j][m]);
            if (c.xvalid) store_split(c.qa[5] + es[0], outx);
        }
        R8H_LTS(14)
        __syncthreads();
        R8H_LTS(15)
    }
}
// spatial mean of the last layer's (BatchNorm'ed) output + Linear(45, n_labels): 16-lane shuffles, one LDS combine over the
// four waves, n_labels threads.  Its own function, called after the layer loop: inside the loop its lane-derived addresses
// were hoisted to the loop's front and spilled.
__device__ __forceinline__ void x_tail(const Res8hParams& p, const XCtx& c0, const unsigned short* pos_tab, const int clip,
                                       const f32x4 (&acc)[5][3], const f32x4& accx) {
    // (lane-derived values rebuilt from the thread id: kept alive across the layer loop they were spilled)
    struct { int tid, pcol, w, padmask; bool xvalid; float* red; float* mvec; int cm[3]; } c;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    c.tid = tid;
    c.pcol = tid & 15;
    c.w = c0.w;
    c.red = c0.red;
    c.mvec = c0.mvec;
    c.padmask = 0;
#pragma unroll
    for (int j = 0; j < 6; ++j) c.padmask |= (pos_tab[16 * (j < 5 ? 5 * c.w + j : 20) + c.pcol] >> 15) << j;
    c.xvalid = c.w < 3 && !((c.padmask >> 5) & 1);
#pragma unroll
    for (int m = 0; m < 3; ++m) c.cm[m] = c0.cm[m];
    const int g = (tid & 63) >> 4;
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = 0.f;
#pragma unroll
            for (int j = 0; j < 5; ++j) v += (c.padmask >> j) & 1 ? 0.f : acc[j][m][r];   // a pad lane's position is counted by the lane it clones
            v += __shfl_xor(v, 8);
            v += __shfl_xor(v, 4);
            v += __shfl_xor(v, 2);
            v += __shfl_xor(v, 1);
            if (c.pcol == 0) c.red[c.w * 48 + 16 * c.cm[m] + 4 * g + r] = v;
        }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float v = c.xvalid ? accx[r] : 0.f;
        v += __shfl_xor(v, 8);
        v += __shfl_xor(v, 4);
        v += __shfl_xor(v, 2);
        v += __shfl_xor(v, 1);
        if (c.pcol == 0 && c.w < 3) c.red[c.w * 48 + 16 * c.cm[0] + 4 * g + r] += v;
    }
    __syncthreads();
    if (c.tid < 48)   // mean over the positions, then the last BatchNorm (mean(BN(x)) == BN(mean(x)))
        c.mvec[c.tid] = fmaf((c.red[c.tid] + c.red[48 + c.tid] + c.red[96 + c.tid] + c.red[144 + c.tid]) / (float)R8_NPOS, c0.bnt[c.tid],
                             c0.bnt[48 + c.tid]);
    __syncthreads();
    if (c.tid < p.n_labels) {
        const float* wr = p.out_w + c.tid * R8_C;
        float o = 0.f;
        for (int ch = 0; ch < R8_C; ++ch) o = fmaf(wr[ch], c.mvec[ch], o);
        p.logits[(size_t)clip * p.n_labels + c.tid] = o + p.out_b[c.tid];
    }
}
}  // namespace

size_t res8h_lds_bytes() { return (size_t)X_LDS_BYTES; }

// Two workgroups (two clips) per CU: 77 KB of LDS and <= 256 registers each.  While one is in an epilogue, in conv_0 or
// waiting for its features, the other's waves use the matrix pipe -- the overlap a single workgroup with one wave per SIMD
// cannot have.  Measured (KWS_R8_WGS_PER_CU=1|2 with this very kernel): 21.9 -> 16.9 ms per 65 536 clips; the one-wave-
// per-SIMD, 502-register build of the same code took 18.0 ms.  The two workgroups of a CU do not run at the same speed (84
// against 106 us per clip: the arbiters favour the older waves), hence the clip queue below.  A start-up stagger of the second
// workgroup changes nothing (v8: second workgroup identified per physical CU -- HW_ID / XCC_ID arrival counters -- and
// delays of 30 / 60 / 90 k ticks: 13.62 - 13.76 ms, noise).

template <int TERMS>   // 3: fp32-accurate; 1: plain fp16 operands (KWS_DTYPE_F16)
__global__ __launch_bounds__(256, 2) void res8h_kernel(Res8hParams p) {
    extern __shared__ __attribute__((aligned(16))) char ldsb[];
    XCtx c;
    c.lds = ldsb;
    unsigned* feat_w = reinterpret_cast<unsigned*>(ldsb);   // conv_0's input is staged where the activation map will be
    c.red = reinterpret_cast<float*>(ldsb + RED_OFF);
    c.mvec = c.red + 4 * 48;
    float* bnt = c.mvec + 48;
    c.bnt = bnt;

    c.w = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int w = c.w;
    // slot m of wave w holds channel tile (w + m) mod 3: slot 0 is then the tile the wave owns on position tile 20 (waves
    // 0..2), and no instruction of the k-loop depends on which wave runs it
    int om[3];
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        c.cm[m] = (w + m) % 3;
        om[m] = c.cm[m] * 2 * A_FRAG_B;
    }
    const __amdgpu_buffer_rsrc_t ars =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.apk2), 0, R8_LAYERS * A_LAYER_B, 0x00020000);

    for (int i = threadIdx.x; i < BNT_WORDS; i += 256) bnt[i] = p.bn_tab[i];
    unsigned short* const pos_tab = reinterpret_cast<unsigned short*>(ldsb + POS_OFF);
    for (int i = threadIdx.x; i < 21 * 16; i += 256) pos_tab[i] = R8H_POS[i];

    // Clips are handed out by a device-wide counter, not by a fixed stride: the two workgroups of a CU do NOT progress at
    // the same rate (the SIMDs' arbiters favour the older waves: measured 84 against 106 us per clip, so with a static
    // split the favoured workgroup of every CU sat idle for the last fifth of the launch), and the XCDs differ by a few
    // per cent as well.  The first clip is blockIdx.x, ticket t is clip gridDim.x + t; the counter resets itself when the last
    // workgroup retires (kws_internal.h, queue_retire: no memset node in front of the launch).  It is read one clip ahead
    // (the atomic is in flight behind the feature loads) and published to the other waves through one LDS word.
    int* const next_clip = reinterpret_cast<int*>(ldsb + NEXT_OFF);
    if (R8H_EPRIO) __builtin_amdgcn_s_setprio(R8H_EPRIO);
    for (int clip = blockIdx.x; clip < p.B;) {
        __syncthreads();  // previous clip's tail has consumed red/mvec and the map
        R8H_TS_DECL
        R8H_TS(0)
        int taken = 0;
        if (threadIdx.x == 0) taken = (int)(gridDim.x + atomicAdd(p.queue, 1u));

        // Everything derived from the lane id is recomputed per clip from an opaque copy of it.  Otherwise the compiler
        // hoists some 80 per-lane addresses and selectors out of this loop, keeps them alive across it and -- at 256
        // registers -- spills them (7.9 GB of scratch reloads per 65 536 clips in the counters).
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        c.tid = tid;
        c.lane = tid & 63;
        c.g = c.lane >> 4;
        c.pcol = c.lane & 15;
        const int g = c.g, pcol = c.pcol, lane = c.lane;
        c.padmask = 0;
        const int ga = (g & 1) * PLANE_B + (g >> 1) * 16 - QBIAS;
        const int lds0 = (int)(unsigned)reinterpret_cast<uintptr_t>(ldsb);   // low word of the flat address = the LDS address
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int nt = j < 5 ? 5 * w + j : 20;
            const int e = pos_tab[16 * nt + pcol];
            const int nn = e & 0x7fff;
            const int y = nn / W8_W;
            const int x = nn - y * W8_W;
            c.qa[j] = ((y + 1) * R8_RS + x + 1) * 16 + ga + lds0;
            c.padmask |= (e >> 15) << j;
        }
        c.dB = (g >> 1) * (R8_RS - 1) * 16;               // second tap of the pair: one row down instead of one cell right
        c.dC = (g >> 1) * (2 * PLANE_B - 16);             // four channel blocks of one tap
        c.dD = (g >> 1) * (PART_B - 16);                  // groups 2, 3: part 2 of the same cell
#pragma unroll
        for (int m = 0; m < 3; ++m) c.es[m] = (2 * c.cm[m] + (g >> 1)) * PLANE_B + 8 * (g & 1) - ga;
        c.xvalid = w < 3 && !((c.padmask >> 5) & 1);

        // ---- stage the (101, 40) feature map as fp16 pairs with a zero top row / left column, inside the idle map region.
        //      Caller-provided features may exceed fp16's range (the reference takes any fp32 value, model/resnet.py:39-41):
        //      p.feat_shift then holds, per clip, the power of two its features are scaled down by here; conv_0 + ReLU +
        //      AvgPool commute with a positive scale, which the pool's multiplier undoes (exact both ways).
        const int fshift = p.feat_shift ? __builtin_amdgcn_readfirstlane(p.feat_shift[clip]) : 0;
        {
            const f32x4* f4 = reinterpret_cast<const f32x4*>(p.feat + (size_t)clip * p.T * p.F);
            f32x4 v[4];
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int q4 = it * 256 + tid;
                v[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (q4 < 1010) v[it] = f4[q4];
            }
            if (fshift > 0) {   // (wave-uniform; never taken for log-mel features)
                const float down = ldexpf(1.f, -fshift);
#pragma unroll
                for (int it = 0; it < 4; ++it) v[it] *= down;
            }
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int q4 = it * 256 + tid;
                if (q4 < 1010) {
                    const int idx = 4 * q4;
                    const int cell = idx + idx / 40 + FS + 1;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {   // one word per feature: fp16 part 1 in the low half, part 2 in the high half
                        unsigned hl;
                        asm("v_cvt_f16_f32 %0, %1" : "=v"(hl) : "v"(v[it][e]));
                        asm("v_fma_mixhi_f16 %0, %0, -1.0, %1 op_sel_hi:[1,0,0]" : "+v"(hl) : "v"(v[it][e]));
                        feat_w[cell + e] = hl;
                    }
                }
            }
            if (tid < FS) feat_w[tid] = 0u;
            if (tid < 101) feat_w[(tid + 1) * FS] = 0u;
            if (tid == 0) *next_clip = taken;
        }
        __syncthreads();
        R8H_TS(1)

        // ---- conv_0 + ReLU + AvgPool(4,3) on the fp16 matrix cores, result in accumulator layout = prev_x.
        //      K = 9 is so short that all THREE terms of the fp32-accurate product fit one 16x16x32 step side by side:
        //      lane group 0 holds (w1, x1) for taps 0..7, group 1 (w1, x2), group 2 (w2, x1), group 3 the three
        //      products of tap 8 -- one MFMA per (position tile, window member, channel tile) instead of three (648 ->
        //      216 per wave and clip).  Every lane reads eight staged words (both fp16 parts of a feature) at
        //      compile-time offsets from its window base (group 3's base is shifted onto tap 8) and v_perm_b32 gathers the
        //      halves its group needs.  TERMS == 1 (plain fp16 operands) keeps the one-term layout: taps 0..7 on group 0,
        //      tap 8 on group 1.  2^-S and the 1/12 of the pool ride on one multiply.
        f32x4 prev[5][3], prevx;
        {
            int lbw[6];   // byte address of the window base of this lane's position in each tile (+ tap 8 for lane group 1)
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const int nt = j < 5 ? 5 * w + j : 20;
                const int nn = pos_tab[16 * nt + pcol] & 0x7fff;
                const int y = nn / W8_W;
                const int x = nn - y * W8_W;
                lbw[j] = (4 * y * FS + 3 * x + (g == (TERMS >= 3 ? 3 : 1) ? 2 * FS + 2 : 0)) * 4;
            }
            // selectors: low halves = part 1, high halves = part 2 of (even word, odd word); 0x0c bytes give zero
            const unsigned sel_h0 = g == 0 ? 0x05040100u : (g == 1 ? 0x0c0c0100u : 0x0c0c0c0cu);
            const unsigned sel_l0 = g == 0 ? 0x07060302u : (g == 1 ? 0x0c0c0302u : 0x0c0c0c0cu);
            const unsigned sel_h = g == 0 ? 0x05040100u : 0x0c0c0c0cu;
            const unsigned sel_l = g == 0 ? 0x07060302u : 0x0c0c0c0cu;
            // K-packed form: groups 0 and 2 take part 1 of words (2i, 2i+1), group 1 part 2; group 3 takes both halves of
            // word 0 (x1, x2 of tap 8), then x1 of tap 8 again (word 0 is substituted for word 2 on that group) and zeros
            const unsigned selk_a = g == 1 ? 0x07060302u : (g == 3 ? 0x03020100u : 0x05040100u);
            const unsigned selk_b = g == 1 ? 0x07060302u : (g == 3 ? 0x0c0c0100u : 0x05040100u);
            const unsigned selk_c = g == 1 ? 0x07060302u : (g == 3 ? 0x0c0c0c0cu : 0x05040100u);
            u32x4 a0[3][2];
            const u32x4* A0 = reinterpret_cast<const u32x4*>(p.w0h) + lane;
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                if (TERMS >= 3) {
                    a0[m][0] = A0[(6 + c.cm[m]) * 64];      // the K-packed fragments follow the two-part ones
                    a0[m][1] = a0[m][0];
                } else {
#pragma unroll
                    for (int pt = 0; pt < 2; ++pt) a0[m][pt] = A0[(c.cm[m] * 2 + pt) * 64];
                }
            }
            const f32x4 zero = (f32x4){0.f, 0.f, 0.f, 0.f};
            float is0 = p.inv_scale0;
            asm volatile("" : "+s"(is0));   // (computed here, per clip: hoisted to the kernel's front the product sat in a spilled VGPR)
            const float post = is0 * (1.0f / 12.0f) * (fshift > 0 ? ldexpf(1.f, fshift) : 1.f);
            const char* fb = reinterpret_cast<const char*>(feat_w);
            // B fragments of pooling-window member (OY, OX) of the position whose window base is at byte address LB
#define C0_FRAG(LB, OY, OX, BH, BL)                                                                            \
    {                                                                                                          \
        unsigned w_[8];                                                                                        \
        _Pragma("unroll") for (int e = 0; e < 8; ++e)                                                          \
            w_[e] = *reinterpret_cast<const unsigned*>(fb + (LB) + (((OY) + e / 3) * FS + (OX) + e % 3) * 4);  \
        BH[0] = __builtin_amdgcn_perm(w_[1], w_[0], sel_h0);                                                   \
        BL[0] = __builtin_amdgcn_perm(w_[1], w_[0], sel_l0);                                                   \
        _Pragma("unroll") for (int i = 1; i < 4; ++i) {                                                        \
            BH[i] = __builtin_amdgcn_perm(w_[2 * i + 1], w_[2 * i], sel_h);                                    \
            BL[i] = __builtin_amdgcn_perm(w_[2 * i + 1], w_[2 * i], sel_l);                                    \
        }                                                                                                      \
    }
#define C0_FRAGK(LB, OY, OX, BK)                                                                               \
    {                                                                                                          \
        unsigned w_[8];                                                                                        \
        _Pragma("unroll") for (int e = 0; e < 8; ++e)                                                          \
            w_[e] = *reinterpret_cast<const unsigned*>(fb + (LB) + (((OY) + e / 3) * FS + (OX) + e % 3) * 4);  \
        BK[0] = __builtin_amdgcn_perm(w_[1], w_[0], selk_a);                                                   \
        BK[1] = __builtin_amdgcn_perm(w_[3], g == 3 ? w_[0] : w_[2], selk_b);                                  \
        BK[2] = __builtin_amdgcn_perm(w_[5], w_[4], selk_c);                                                   \
        BK[3] = __builtin_amdgcn_perm(w_[7], w_[6], selk_c);                                                   \
    }
            int tile_gate = 1;
            asm volatile("" : "+s"(tile_gate));
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                f32x4 s0 = zero, s1 = zero, s2 = zero;
                // Each position tile's 12 window members stay in a basic block of their own behind a uniform branch hipcc cannot fold (`tile_gate`
                // is 1, out of an empty asm): as ONE block of 60 members the pre-RA scheduler pulls the window reads of all five tiles to the
                // front and spills 3 KB per lane (`__builtin_amdgcn_sched_barrier(0)` and a memory-clobber asm per tile do not stop it: measured,
                // r5).  Rounds 3-4 had the experiment branch `p.debug & 1` standing here, with the same effect and the same cost (one scalar
                // compare + branch per tile); the default build has no experiment bits any more.
                if (tile_gate && !(KWS_DBG(p.debug & 1))) {
#pragma unroll
                    for (int wi = 0; wi < 12; ++wi) {
                        const int oy = wi / 3, ox = wi - 3 * oy;
                        u32x4 bh, bl;
                        f32x4 cc[3];
                        if (TERMS >= 3) {
                            C0_FRAGK(lbw[j], oy, ox, bh)
#pragma unroll
                            for (int m = 0; m < 3; ++m) {
                                cc[m] = zero;
                                MF(a0[m][0], bh, cc[m]);
                            }
                        } else {
                            C0_FRAG(lbw[j], oy, ox, bh, bl)
#pragma unroll
                            for (int m = 0; m < 3; ++m) {
                                cc[m] = zero;
                                MF(a0[m][1], bh, cc[m]);
                            }
#pragma unroll
                            for (int m = 0; m < 3; ++m) { MF(a0[m][0], bl, cc[m]); }
#pragma unroll
                            for (int m = 0; m < 3; ++m) { MF(a0[m][0], bh, cc[m]); }
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r) {   // scalar adds: on f32x4 values these become v_pk_add_f32, which issues slower than two plain adds beside MFMAs
                            s0[r] += relu1(cc[0][r]);
                            s1[r] += relu1(cc[1][r]);
                            s2[r] += relu1(cc[2][r]);
                        }
                    }
                }
                prev[j][0] = s0 * post;
                prev[j][1] = s1 * post;
                prev[j][2] = s2 * post;
            }
            {
                f32x4 sx = zero;
                if (tile_gate && !(KWS_DBG(p.debug & 1)) && w < 3) {   // wave 3 owns no extra tile
#pragma unroll
                    for (int wi = 0; wi < 12; ++wi) {
                        const int oy = wi / 3, ox = wi - 3 * oy;
                        u32x4 bh, bl;
                        f32x4 cx = zero;
                        if (TERMS >= 3) {
                            C0_FRAGK(lbw[5], oy, ox, bh)
                            MF(a0[0][0], bh, cx);
                        } else {
                            C0_FRAG(lbw[5], oy, ox, bh, bl)
                            MF(a0[0][1], bh, cx);
                            MF(a0[0][0], bl, cx);
                            MF(a0[0][0], bh, cx);
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r) sx[r] += relu1(cx[r]);
                    }
                }
                prevx = sx * post;
            }
#undef C0_FRAG
#undef C0_FRAGK
        }
        R8H_TS(2)
        // every wave is done with the staged features: turn the region back into a map -- zero halo, then the pooled conv_0
        // output in its interior
        int shift;   // range guard of the map the next layer reads
        int avoff = lane * 16;
        AFrags fa[2];
        load_a(fa[0], ars, avoff, 0, om);   // conv_1's first weight fragments
        {
            float am[3] = {0.f, 0.f, 0.f};   // pooled ReLU outputs: >= 0.  Three independent chains of v_max3_f32, not one of 64 dependent maxima
#pragma unroll
            for (int j = 0; j < 5; ++j)
#pragma unroll
                for (int m = 0; m < 3; ++m) {
                    am[m] = fmaxf(fmaxf(am[m], prev[j][m][0]), prev[j][m][1]);
                    am[m] = fmaxf(fmaxf(am[m], prev[j][m][2]), prev[j][m][3]);
                }
            float amax = fmaxf(fmaxf(am[0], am[1]), am[2]);
            if (c.xvalid) amax = fmaxf(fmaxf(amax, prevx[0]), fmaxf(fmaxf(prevx[1], prevx[2]), prevx[3]));
            guard_push(reinterpret_cast<unsigned*>(c.red), w, lane, amax);   // group 0; layer l uses group (l + 1) & 1
        }
        __syncthreads();
        // Only the 59 cells outside the 25 x 13 interior need clearing (map row 0, the shared halo cell in front of every
        // row, everything from map row 26 on): the interior is overwritten right below, all 48 channel slots of both parts.
        for (int t = tid; t < 59 * 12; t += 256) {
            const int hc = t / 12, sub = t - 12 * hc;
            const int cell = hc < 14 ? hc : (hc < 39 ? R8_RS * (hc - 13) : R8_RS * 26 + (hc - 39));
            *reinterpret_cast<u32x4*>(ldsb + (sub / 6) * PART_B + (sub % 6) * PLANE_B + cell * 16) = (u32x4){0u, 0u, 0u, 0u};
        }
        shift = __builtin_amdgcn_readfirstlane(range_shift(guard_read(reinterpret_cast<const unsigned*>(c.red))));
#pragma unroll
        for (int m = 0; m < 3; ++m)   // the constant channel (slot 45) of x_0's map: 1 (it carries conv_2's share of BatchNorm 1; conv_1's weights ignore it)
            if (c.cm[m] == 2) {
#pragma unroll
                for (int j = 0; j < 5; ++j) prev[j][m][1] = g == 3 ? 1.0f : prev[j][m][1];
                if (m == 0) prevx[1] = g == 3 ? 1.0f : prevx[1];
            }
        if (shift > 0) {   // (wave-uniform, never taken for trained models)
            asm volatile("; range guard: scale the map down" ::: "memory");
            const float down = ldexpf(1.f, -shift);
#pragma unroll
            for (int j = 0; j < 5; ++j)
#pragma unroll
                for (int m = 0; m < 3; ++m) store_split(c.qa[j] + c.es[m], prev[j][m] * down);
            if (c.xvalid) store_split(c.qa[5] + c.es[0], prevx * down);
        } else {
#pragma unroll
            for (int j = 0; j < 5; ++j)
#pragma unroll
                for (int m = 0; m < 3; ++m) store_split(c.qa[j] + c.es[m], prev[j][m]);
            if (c.xvalid) store_split(c.qa[5] + c.es[0], prevx);
        }
        __syncthreads();

        R8H_TS(3)
        f32x4 acc[5][3], accx;
        // Two copies of the layer's code (odd / even i: the residual path is compile-time, a runtime flag cost selects and register
        // copies on every value), walked three times: the six inlined copies of round 2 made the kernel 66 KB, more than the
        // instruction cache.
#pragma unroll 1
        for (int lp = 0; lp < R8_LAYERS / 2; ++lp) {
            x_layer<TERMS, false>(p, c, ars, avoff, om, 2 * lp, acc, accx, prev, prevx, shift, fa R8H_TSPASS);
#ifdef R8H_TIMING
            if (lp == 0) { R8H_TS(4) }
#endif
            x_layer<TERMS, true>(p, c, ars, avoff, om, 2 * lp + 1, acc, accx, prev, prevx, shift, fa R8H_TSPASS);
#ifdef R8H_TIMING
            if (lp == 0) { R8H_TS(5) }
            else if (lp == 1) { R8H_TS(6) }
#endif
        }
        x_tail(p, c, pos_tab, clip, prev, prevx);   // the last layer is even: its x is in the residual registers
        R8H_TS(7)
#ifdef R8H_TIMING
        if ((threadIdx.x & 63) == 0) {      // the clip's features are dead: park the timestamps there (tools/r8_phases.py)
            unsigned long long* o = reinterpret_cast<unsigned long long*>(const_cast<float*>(p.feat) + (size_t)clip * p.T * p.F) + 16 * w;
            for (int i = 0; i < 16; ++i) o[i] = ts[i];
            if (w == 0) {   // 100 MHz wall clock over the same span: shader clock = d ticks / d realtime x 100 MHz
                o[64] = rt0;
                o[65] = __builtin_amdgcn_s_memrealtime();
                o[66] = ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) << 32) |   // XCC_ID
                        (unsigned)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));                        // HW_ID
            }
        }
#endif
        clip = *next_clip;   // written before the staging barrier of this clip; every wave reads it before the loop's top barrier
    }
    if (threadIdx.x == 0) queue_retire(p.queue);
}

// per clip: the power of two that brings max |feature| under 2^14 (0 for anything a front end produces); one wave per clip
__global__ __launch_bounds__(256) void feat_shift_kernel(const float* __restrict__ feat, int B, int n4, int* __restrict__ shift) {
    const int clip = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (clip >= B) return;
    const f32x4* f4 = reinterpret_cast<const f32x4*>(feat) + (size_t)clip * n4;
    float amax = 0.f;
    for (int i = lane; i < n4; i += 64) {
        const f32x4 v = f4[i];
        amax = fmaxf(fmaxf(amax, fabsf(v[0])), fmaxf(fmaxf(fabsf(v[1]), fabsf(v[2])), fabsf(v[3])));
    }
    const unsigned m = wave_umax(__builtin_bit_cast(unsigned, amax));
    if (lane == 0) shift[clip] = range_shift(__builtin_bit_cast(float, m));
}

hipError_t launch_feat_shift(const float* feat, int B, int n_per_clip, int* shift, hipStream_t s) {
    if (B <= 0) return hipSuccess;
    hipLaunchKernelGGL(feat_shift_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, s, feat, B, n_per_clip / 4, shift);
    return hipGetLastError();
}

hipError_t launch_res8h(const Res8hParams& p, int grid, hipStream_t s) {
    static DeviceOnce attr_once;
    if (attr_once.first()) {
        hipError_t e = hipFuncSetAttribute((const void*)res8h_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)res8h_lds_bytes());
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute((const void*)res8h_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)res8h_lds_bytes());
        if (e != hipSuccess) return e;
    }
    if (p.B <= 0) return hipSuccess;
    if (p.terms == 1)
        hipLaunchKernelGGL(res8h_kernel<1>, dim3((unsigned)grid), dim3(256), res8h_lds_bytes(), s, p);
    else
        hipLaunchKernelGGL(res8h_kernel<3>, dim3((unsigned)grid), dim3(256), res8h_lds_bytes(), s, p);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------- host packing
// conv_0 weight (45, 9) times `scale` -> [channel tile 3][part 2][lane 64][8 fp16]: cout = 16 m + (lane & 15), k-slot
// 8 (lane >> 4) + j holds tap k for k < 9, zeros otherwise; then [channel tile 3][lane 64][8 fp16] K-packed (below)
void pack_res8h_conv0(const float* wt, float scale, unsigned short* dst) {
    for (int m = 0; m < 3; ++m)
        for (int lane = 0; lane < 64; ++lane) {
            const int co = 16 * m + (lane & 15), g = lane >> 4;
            for (int j = 0; j < 8; ++j) {
                const int k = 8 * g + j;
                const float v = (k < 9 && co < R8_C) ? wt[co * 9 + k] * scale : 0.f;
                const unsigned short h = f16_rne_host(v);
                dst[((m * 2 + 0) * 64 + lane) * 8 + j] = h;
                dst[((m * 2 + 1) * 64 + lane) * 8 + j] = f16_rne_host(v - f16_to_f_host(h));
            }
            // K-packed fragment (fragments 6..8): the three terms side by side -- lane group 0: w1 of taps 0..7 (meets x1),
            // group 1: w1 again (meets x2), group 2: w2 (meets x1), group 3: tap 8 as (w1, w1, w2) against (x1, x2, x1)
            for (int j = 0; j < 8; ++j) {
                const int tap = g < 3 ? j : (j < 3 ? 8 : -1);
                const bool low = g == 2 || (g == 3 && j == 2);
                unsigned short out = 0;
                if (tap >= 0 && co < R8_C) {
                    const float v = wt[co * 9 + tap] * scale;
                    const unsigned short h = f16_rne_host(v);
                    out = low ? f16_rne_host(v - f16_to_f_host(h)) : h;
                }
                dst[((6 + m) * 64 + lane) * 8 + j] = out;
            }
        }
}

// conv_i weight (45, 46, 3, 3) -- input channel 45 = the previous BatchNorm's shift, met by the map's constant channel -- times `scale` ->
// [k-step 14][channel tile 3][part 2][lane 64][8 fp16] in the kernel's K order
// (kinds A - D above): cout = 16 m + (lane & 15), lane group g = lane >> 4 holds 8 input channels of one tap.  The last
// step's part-1 fragment carries a1 of blocks 4, 5 of tap 8 on groups 0, 1 AND on groups 2, 3 (it meets part 1 | part 2 of
// the activations in one MFMA), its part-2 fragment a2 on groups 0, 1 and zeros on groups 2, 3.
void pack_res8h_layer(const float* wt, float scale, unsigned short* dst) {
    auto weight = [&](int co, int tap, int cblk, int j, bool low) -> unsigned short {
        float v = 0.f;
        const int ci = 8 * cblk + j;
        if (co < R8_C && ci < R8_C + 1) v = wt[((size_t)co * (R8_C + 1) + ci) * 9 + tap] * scale;   // ci == 45: the folded BatchNorm shift
        const unsigned short h = f16_rne_host(v);
        return low ? f16_rne_host(v - f16_to_f_host(h)) : h;
    };
    for (int s = 0; s < KSTEPS; ++s)
        for (int m = 0; m < 3; ++m)
            for (int lane = 0; lane < 64; ++lane) {
                const int co = 16 * m + (lane & 15), g = lane >> 4;
                int tap, cblk;
                bool zero_low = false;
                if (s < 9) {
                    tap = 3 * (s / 3) + (g >> 1);
                    cblk = 2 * (s % 3) + (g & 1);
                } else if (s < 12) {
                    tap = 2 + 3 * (g >> 1);
                    cblk = 2 * (s - 9) + (g & 1);
                } else if (s == 12) {
                    tap = 8;
                    cblk = g;
                } else {
                    tap = 8;
                    cblk = 4 + (g & 1);
                    zero_low = g >= 2;
                }
                for (int j = 0; j < 8; ++j) {
                    dst[((((size_t)s * 3 + m) * 2 + 0) * 64 + lane) * 8 + j] = weight(co, tap, cblk, j, false);
                    dst[((((size_t)s * 3 + m) * 2 + 1) * 64 + lane) * 8 + j] = zero_low ? 0 : weight(co, tap, cblk, j, true);
                }
            }
}

}  // namespace kws
