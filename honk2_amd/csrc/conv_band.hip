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
//   * LDS image: [part][8-channel block][cell] x 16 B, cells = band rows x Wl (row stride padded), block planes a multiple of 16
//     cells apart.  A ds_read_b128 of a B fragment is served in four 16-lane groups, each holding all 16 positions of the tile (8
//     lanes of one k-group + the complementary 8 of the next one, whose plane starts 0 mod 256 B further on): it is conflict-free
//     iff the tile's 16 cells are distinct mod 16 (res8_f16x3.hip, tools/r8_tiles.py).  Position tiles are therefore built on the
//     host by residue class of the cell index (conv_band_plan picks the row stride Wl that balances the classes); the first
//     version's cell-major image (tiles of 16 consecutive positions) spent 54 % of its LDS cycles in bank conflicts and left the
//     matrix pipe 49 % busy.  Staging walks cells lane-fastest so that its 16-byte LDS writes are conflict-free too.
//   * K order (tap, 8-channel block), four blocks per v_mfma_f32_16x16x32_f16 (one per 16-lane group); a k-step's (tap, block) of
//     each lane group is a byte offset from a table in LDS.  Weights: fp16 parts scaled by 2^S in fragment order from L2 (shared by
//     every workgroup), one k-step ahead.
//   * Waves 2 x 2: wave (wm, wn) owns half of the channel tiles (MH = 2 or 3) and half of the band's position tiles (<= 4; <= 8 with fp16 tensors): per
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

constexpr int BAND_NT = 4;    // position tiles per wave (half a band) of the narrow instantiation
constexpr int BAND_NT1 = 8;   // ... of the wide one, for bands of more than eight tiles: every weight fragment a wave loads meets up to eight position tiles
                              // instead of four (r3: the weight stream from L2 was the kernel's largest cost, see DESIGN); fp16 tensors hold one operand part, half the
                              // LDS per row, and get there with twice the rows per band
#ifndef BAND_APF
#define BAND_APF 3            // k-steps of weight-fragment look-ahead (A/B knob)
#endif
#ifndef BAND_ABLATE
#define BAND_ABLATE 0        // timing experiments (results wrong): 1 no LDS fragment reads in the k-loop, 2 no weight-fragment loads
#endif

namespace {
// 32-bit LDS addresses for the fragment reads (res8_f16x3.hip, conv3x3_tile.hip): the kernel holds only dynamic LDS, which starts at address 0
typedef const u32x4 __attribute__((address_space(3))) * band_lds_u32x4_ptr;
typedef const int __attribute__((address_space(3))) * band_lds_i32_ptr;
__device__ __forceinline__ u32x4 band_lds_read16(int addr) { return *reinterpret_cast<band_lds_u32x4_ptr>((unsigned)addr); }
__device__ __forceinline__ int band_lds_read4(int addr) { return *reinterpret_cast<band_lds_i32_ptr>((unsigned)addr); }
}  // namespace

// (r3) The k-loop for a wave that owns NTL position tiles, NTL a compile-time constant.  The loop it replaces took the tile count at run time:
// uniform branches around every fragment read and MFMA group, register copies where the paths met again, 64-bit vector adds for the weight
// addresses and a vector add of the `lds` symbol per read -- 4.5 vector + as many scalar instructions per MFMA in the single-term form, where a
// k-step is only 2 NTL MFMAs long: the loop was bound by its own bookkeeping (profiles/r03: vector pipe 47 % busy, matrix pipe 42 %).  Here a k-step
// is MH x NP buffer loads at scalar offsets (BAND_APF steps ahead), NTL x NP LDS reads of the NEXT step's fragments at lbase[j] + koff
// (one add each), one table word, and the MFMAs: chain-major for three-term products (res8_f16x3.hip, R8H_FENCE).
template <int MH, int TERMS, int NTL, int NT>
__device__ __forceinline__ void band_kloop(const __amdgpu_buffer_rsrc_t ars, const int avoff, const int ktab_addr, const int (&lbase)[NT],
                                           const int partb, const int ksteps, f32x4 (&acc)[MH][NT]) {
    constexpr int NP = TERMS >= 3 ? 2 : 1;
    constexpr int ASTEP_B = 2 * MH * 2 * 1024;           // bytes of weight fragments per k-step: [2 MH channel tiles][2 parts][64 lanes] x 16 B
    constexpr int NA = BAND_APF + 1;
    // B fragments: single-term products finish a tile in MH MFMAs -- less than an LDS round trip -- so the whole NEXT k-step's fragments are in
    // flight (two sets of NTL); three-term products take 3 MH MFMAs per tile and look one tile ahead (two buffers: the registers buy 2 x 8 tiles)
    constexpr bool STEP_AHEAD = TERMS < 3;
    constexpr int NBUF = STEP_AHEAD ? 2 * NTL : 2;
    static_assert((NA * NTL) % 2 == 0, "the two-buffer ring keeps its phase from block to block");
    static_assert(!STEP_AHEAD || NA % 2 == 0, "single-term products pick the buffer set by (u & 1): an odd BAND_APF + 1 flips it between unrolled blocks");
    u32x4 a[NA][MH][NP], b[NBUF][NP];
    auto load_a = [&](u32x4 (&ar)[MH][NP], int st) {
#pragma unroll
        for (int m = 0; m < MH; ++m)
#pragma unroll
            for (int pt = 0; pt < NP; ++pt)
                ar[m][pt] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(ars, avoff + (m * 2 + pt) * 1024, st * ASTEP_B, 0));
    };
    auto load_b = [&](u32x4 (&br)[NP], int j, int koff) {
        const int ad = lbase[j] + koff;
#pragma unroll
        for (int pt = 0; pt < NP; ++pt) br[pt] = band_lds_read16(ad + pt * partb);
    };
    int kc = band_lds_read4(ktab_addr);
#pragma unroll
    for (int u = 0; u < BAND_APF; ++u) load_a(a[u], min(u, ksteps - 1));
    if (STEP_AHEAD) {
#pragma unroll
        for (int j = 0; j < NTL; ++j) load_b(b[j], j, kc);
    } else
        load_b(b[0], 0, kc);
    for (int s = 0; s < ksteps; s += NA) {
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            if (s + u >= ksteps) break;
            const int kn = band_lds_read4(ktab_addr + 16 * (s + u + 1));     // (the table carries two spare steps)
            if (!(BAND_ABLATE & 2)) load_a(a[(u + BAND_APF) % NA], min(s + u + BAND_APF, ksteps - 1));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NTL; ++j) {
                const int t = u * NTL + j;                    // position in the unrolled block
                const int cur = STEP_AHEAD ? (u & 1) * NTL + j : (t & 1);
                const int nxt = STEP_AHEAD ? ((u + 1) & 1) * NTL + j : ((t + 1) & 1);
                if (!(BAND_ABLATE & 1)) {
                    if (STEP_AHEAD) load_b(b[nxt], j, kn);      // the next step's fragment of this tile (past the end: a harmless re-read)
                    else if (j + 1 < NTL) load_b(b[nxt], j + 1, kc);
                    else load_b(b[nxt], 0, kn);
                } else
                    b[nxt][0] = b[cur][0], b[nxt][NP - 1] = b[cur][NP - 1];
                __builtin_amdgcn_sched_barrier(0);
                const u32x4 (&bc)[NP] = b[cur];
#pragma unroll
                for (int m = 0; m < MH; ++m) {
                    if (TERMS >= 3) {
                        BMF(a[u][m][NP - 1], bc[0], acc[m][j]);
                        BMF(a[u][m][0], bc[NP - 1], acc[m][j]);
                    }
                    BMF(a[u][m][0], bc[0], acc[m][j]);
                    if (TERMS >= 3) __builtin_amdgcn_sched_barrier(0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            kc = kn;
        }
    }
}

#ifndef BAND_STAGE_UNR
#define BAND_STAGE_UNR 13
#endif
// MH: channel tiles per wave (the layer has up to 2 MH); TERMS: 3 (two-part operands, fp32-accurate) or 1 (fp16 tensor in, one part)
template <int MH, int TERMS, int NT>
__global__ __launch_bounds__(256, 2) void conv_band_kernel(BandConvParams p) {
    constexpr int NP = TERMS >= 3 ? 2 : 1;
    constexpr bool S16 = TERMS == 1;
    extern __shared__ __align__(16) char lds[];
    if (range_gate_closed(p.rg)) return;
#ifdef BAND_TIMING   // 100 MHz wall-clock stamps of this workgroup's phases (tools/band_phases.py)
    unsigned long long bts[6];
#define BAND_TS(i) bts[i] = __builtin_amdgcn_s_memrealtime();
#else
#define BAND_TS(i)
#endif
    BAND_TS(0)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w & 1, wn = w >> 1;
    const int g = lane >> 4, pcol = lane & 15;
    // the bands of a clip share kh - 1 of their input rows: blocks that share an XCD (blockIdx mod 8) take a contiguous run of
    // (clip, band) units, so the rows a neighbour already fetched are L2 hits (bijective for any grid size)
    int unit = (int)blockIdx.x;
    {
        const int nwg = (int)gridDim.x, xcd = unit & 7, q8 = nwg >> 3, r8 = nwg & 7;
        unit = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (unit >> 3);
    }
    const int band = unit % p.nbands, b = unit / p.nbands;
    const int r0 = band * p.R;                       // first output row = first input row of the band
    const int rows_out = min(p.R, p.Ho - r0);
    const int rows_in = p.R + p.kh - 1;
    const int nbi = p.Cpi / 8;
    const int planeb = p.PS * 16;                    // bytes per 8-channel block plane
    const int partb = nbi * planeb;                  // bytes per part
    const int ktab_off = NP * partb;                 // int[(ksteps + 2) * 4]

    // ---------------------------------------------------------------- k-step table + staging
    for (int i = tid; i < (p.ksteps + 2) * 4; i += 256) {
        const int tap = i / nbi, cb = i - tap * nbi;
        const int ky = tap / p.kw, kx = tap - ky * p.kw;
        reinterpret_cast<int*>(lds + ktab_off)[i] = tap < p.kh * p.kw ? cb * planeb + (ky * p.Wl + kx) * 16 : 0;   // (padding blocks carry zero weights)
    }
    {
        // One wave-level load = 16 consecutive source cells (lanes 0-15) x four consecutive 16-byte chunks of each (lane >> 4): 64
        // contiguous bytes per cell for the vector-memory path, and every 16-lane group writes 16 consecutive cells of one block
        // plane -- conflict-free LDS stores (chunk-fastest lanes put the 16 chunks of a cell on four banks; cell-fastest lanes
        // over a single chunk touch 64 cache lines per load).
        const int gcell = S16 ? p.Cpi * 2 : p.Cpi * 4;
        const char* src = reinterpret_cast<const char*>(p.in) + ((size_t)b * p.H + r0) * p.W * gcell;
        const int nsrc = rows_in * p.W;
        const int last = (p.H - r0) * p.W - 1;                   // rows past the input feed only outputs that are never stored: clamp
        const int nch = S16 ? p.Cpi / 8 : p.Cpi / 4;             // 16-byte chunks per global cell
        const int nqq = (nch + 3) / 4;                           // chunk quads per cell (the last one may be partial with fp16 tensors)
        const int ncg = (nsrc + 15) / 16;
        const int nit = ncg * nqq;                               // wave-level items
        const int lc = lane & 15, lq = lane >> 4;
        // (r4) fp16 tensors: all of a wave's loads in flight together (13 items per wave for a 25-row band): with four per pass the staging was three to
        // four memory round trips in a row, 5.3 of the workgroup's 31.7 us (tools/band_phases.py; cnn-trad-pool2 fp16 1.83 -> 1.79 ms).  The fp32 form
        // splits what it loads and is no faster with more in flight (4.34 -> 4.43 ms): it keeps four.
        constexpr int UNR = S16 ? BAND_STAGE_UNR : 4;
        for (int i0 = w; i0 < nit; i0 += UNR * 4) {
            f32x4 v[UNR];
            int dst[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int i = min(i0 + u * 4, nit - 1);
                const int cg = i / nqq, qq = i - cg * nqq;
                const int sc = cg * 16 + lc, q = qq * 4 + lq;
                const int row = sc / p.W, x = sc - row * p.W;
                // fp32: chunk q = channels 4q..4q+3 -> block q / 2, half q & 1;  fp16: chunk q = block q
                dst[u] = (sc < nsrc && q < nch) ? (S16 ? q * planeb : (q >> 1) * planeb + (q & 1) * 8) + (row * p.Wl + x) * 16 : -1;
                v[u] = *reinterpret_cast<const f32x4*>(src + (size_t)min(sc, last) * gcell + min(q, nch - 1) * 16);
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                if (i0 + u * 4 >= nit || dst[u] < 0) continue;
                if (S16) {
                    *reinterpret_cast<f32x4*>(lds + dst[u]) = v[u];
                } else {
                    u32x2 pr[2];
                    band_split4(v[u], pr);
                    *reinterpret_cast<u32x2*>(lds + dst[u]) = pr[0];
                    *reinterpret_cast<u32x2*>(lds + partb + dst[u]) = pr[1];
                }
            }
        }
    }

    // this lane's output positions: tile t = wn * nth + j of the layer's position table (conv_band_plan), entry {cell, oy << 16 | ox}
    const int nth = (p.ntiles + 1) >> 1;
    int lbase[NT], opos[NT];
    const int ntile = min(nth, max(0, p.ntiles - wn * nth));     // tiles of this wave (uniform)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int t = min(wn * nth + j, p.ntiles - 1);
        const int cellv = p.postab[(t * 16 + pcol) * 2], yx = p.postab[(t * 16 + pcol) * 2 + 1];
        const int oy = yx >> 16, ox = yx & 0xffff;
        lbase[j] = cellv * 16;
        opos[j] = (j < ntile && yx >= 0 && oy < rows_out) ? (r0 + oy) * p.Wo + ox : -1;
    }
    BAND_TS(1)
    __syncthreads();
    BAND_TS(2)

    // ---------------------------------------------------------------- k-loop
    f32x4 acc[MH][NT];
#pragma unroll
    for (int m = 0; m < MH; ++m)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[m][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    {
        if ((unsigned)reinterpret_cast<uintptr_t>(lds) != 0u) __builtin_trap();     // the integer LDS addresses below assume dynamic LDS at 0
        const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(reinterpret_cast<const unsigned short*>(p.apk)), 0,
                                                                             p.ksteps * (2 * MH * 2 * 1024), 0x00020000);
        const int avoff = lane * 16 + (wm * MH) * 2 * 1024;      // this wave's first channel tile: [k-step][2 MH tiles][2 parts][64 lanes] x 16 B
        const int ktab_addr = ktab_off + 4 * g;
#define BAND_CASE(N) case N: if (N <= NT) band_kloop<MH, TERMS, (N <= NT ? N : 1), NT>(ars, avoff, ktab_addr, lbase, partb, p.ksteps, acc); break;
        switch (ntile) {      // the wave's tile count becomes a compile-time constant of its k-loop
            BAND_CASE(8) BAND_CASE(7) BAND_CASE(6) BAND_CASE(5) BAND_CASE(4) BAND_CASE(3) BAND_CASE(2) BAND_CASE(1)
            default: break;
        }
#undef BAND_CASE
    }

    BAND_TS(3)
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
        for (int j = 0; j < NT; ++j) {
            if (j >= ntile || opos[j] < 0) continue;
            f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float x = fmaf(acc[m][j][r], p.inv_scale, bv[r]);
                if (p.relu) x = fmaxf(x, 0.f);
                v[r] = co0 + r < p.Cout ? x : 0.f;
                amax = fmaxf(amax, fabsf(v[r]));
            }
            if (p.out_f16) {
                typedef _Float16 f16x2_ __attribute__((ext_vector_type(2)));
                const u32x2 pk = {__builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){v[0], v[1]}, f16x2_)),
                                  __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){v[2], v[3]}, f16x2_))};
                *reinterpret_cast<u32x2*>(reinterpret_cast<unsigned short*>(p.out) + (size_t)b * p.Ho * p.Wo * p.Cpo + (size_t)opos[j] * p.Cpo + co0) = pk;
            } else {
                *reinterpret_cast<f32x4*>(outb + (size_t)opos[j] * p.Cpo + co0) = v;
            }
        }
    }
    range_note(p.rg, amax);
#ifdef BAND_TIMING
    BAND_TS(4)
    if (p.dbg_ts && lane == 0 && blockIdx.x < 8192) {     // 4 waves x 8 words per workgroup
        unsigned long long* o = p.dbg_ts + ((size_t)blockIdx.x * 4 + w) * 8;
        for (int i = 0; i < 5; ++i) o[i] = bts[i];
        o[5] = ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) << 32) | (unsigned)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));
    }
#endif
}

size_t conv_band_lds_bytes(int Cpi, int kh, int kw, int PS, int parts) {
    const int nbi = Cpi / 8, ksteps = (kh * kw * nbi + 3) / 4;
    return (size_t)parts * nbi * PS * 16 + (size_t)(ksteps + 2) * 16;
}

// Position tiles of an R-row band whose LDS rows are Wl cells apart: tile t takes the t-th position of every residue class of
// cell = oy Wl + ox mod 16 (lane = class); a class that has run out leaves a pad lane, which clones another lane of its tile (same
// address: a broadcast) and stores nothing.  Returns the number of tiles.
static int band_tiles(int R, int Wo, int Wl, std::vector<int>* tab) {
    std::vector<int> cls[16];
    for (int oy = 0; oy < R; ++oy)
        for (int ox = 0; ox < Wo; ++ox) cls[(oy * Wl + ox) & 15].push_back((oy << 16) | ox);
    size_t nt = 0;
    for (const auto& c : cls) nt = std::max(nt, c.size());
    if (tab) {
        tab->assign(nt * 16 * 2, 0);
        for (size_t t = 0; t < nt; ++t) {
            int clone = -1;
            for (int r = 0; r < 16 && clone < 0; ++r)
                if (t < cls[r].size()) clone = cls[r][t];
            for (int r = 0; r < 16; ++r) {
                const bool have = t < cls[r].size();
                const int yx = have ? cls[r][t] : clone;
                (*tab)[(t * 16 + r) * 2] = (yx >> 16) * Wl + (yx & 0xffff);
                (*tab)[(t * 16 + r) * 2 + 1] = have ? yx : -1;
            }
        }
    }
    return (int)nt;
}

// Rows per band R and LDS row stride Wl: the band must fit half a CU's LDS with `parts` operand parts (2: fp32-accurate products, 1: fp16 tensors)
// and 2 x BAND_NT (2 x BAND_NT1 for one part) position tiles;
// among those, the best share of useful MFMA slots (tile fill x rows covered, halo rows re-staged by every band counted against
// small R).  false: the layer does not fit this kernel.
bool conv_band_plan(int Cin, int Cout, int H, int W, int kh, int kw, int parts, BandPlan& out) {
    const int max_nt = 2 * BAND_NT1;
    const int Cpi = (Cin + 15) / 16 * 16, Ho = H - kh + 1, Wo = W - kw + 1, mh = conv_band_mh(Cout);
    out = BandPlan{};
    if (Cin < 16 || (mh != 2 && mh != 3) || Ho < 1 || Wo < 1 || Wo > 0xffff) return false;
    double best_eff = 0.0;
    for (int R = 1; R <= Ho; ++R) {
        bool any = false;
        for (int Wl = W; Wl < W + 16; ++Wl) {
            const int PS = ((R + kh - 1) * Wl + 15) / 16 * 16;
            if (conv_band_lds_bytes(Cpi, kh, kw, PS, parts) > 80 * 1024 - 256) continue;
            const int nt = band_tiles(R, Wo, Wl, nullptr);
            if (nt > max_nt) continue;
            any = true;
            const int nb = (Ho + R - 1) / R, slots = 2 * ((nt + 1) / 2);
            const double eff = (double)(Ho * Wo) / ((double)nb * slots * 16) * ((double)R / (R + 0.25 * (kh - 1)));
            if (eff > best_eff + 1e-9) {
                best_eff = eff;
                out.R = R; out.Wl = Wl; out.PS = PS; out.ntiles = nt;
            }
        }
        if (!any && R * Wo > 16 * max_nt) break;
    }
    if (out.R == 0) return false;
    band_tiles(out.R, Wo, out.Wl, &out.tab);
    return true;
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

template <int MH, int TERMS, int NT>
static hipError_t launch_band_k(const BandConvParams& p, hipStream_t s) {
    const size_t lds = conv_band_lds_bytes(p.Cpi, p.kh, p.kw, p.PS, TERMS >= 3 ? 2 : 1);
    auto k = conv_band_kernel<MH, TERMS, NT>;
    static DeviceOnce attr_once;   // per instantiation: allow > 64 KB of dynamic LDS
    if (attr_once.first()) {
        hipError_t e = allow_big_lds_at_base_zero(reinterpret_cast<const void*>(k));
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k, dim3((unsigned)(p.B * p.nbands)), dim3(256), lds, s, p);
    return hipGetLastError();
}

hipError_t launch_conv_band(const BandConvParams& p, hipStream_t s) {
    if (p.B <= 0) return hipSuccess;
    const int mh = conv_band_mh(p.Cout);
    const int parts = p.terms == 3 ? 2 : 1;
    if (p.R < 1 || p.Cpi % 16 || p.Cpo % 16 || p.Cpo < p.Cout || p.Cpo > 32 * mh || p.ntiles < 1 || p.ntiles > 2 * BAND_NT1 ||
        p.Wl < p.W || p.PS % 16 || p.PS < (p.R + p.kh - 1) * p.Wl || !p.postab || (p.terms != 3 && p.terms != 1) ||
        conv_band_lds_bytes(p.Cpi, p.kh, p.kw, p.PS, parts) > 160 * 1024 - 512)
        return hipErrorInvalidValue;
    const bool wide = p.ntiles > 2 * BAND_NT;      // more than four tiles per wave
    if (p.terms == 3) {
        if (!wide) return mh == 2 ? launch_band_k<2, 3, BAND_NT>(p, s) : launch_band_k<3, 3, BAND_NT>(p, s);
        return mh == 2 ? launch_band_k<2, 3, BAND_NT1>(p, s) : launch_band_k<3, 3, BAND_NT1>(p, s);
    }
    if (!wide) return mh == 2 ? launch_band_k<2, 1, BAND_NT>(p, s) : launch_band_k<3, 1, BAND_NT>(p, s);
    return mh == 2 ? launch_band_k<2, 1, BAND_NT1>(p, s) : launch_band_k<3, 1, BAND_NT1>(p, s);
}

}  // namespace kws
