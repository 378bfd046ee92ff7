// mbn_f32_stem.hip — fused "stem" of MobileNet-V1 for gfx950, fp32: layers 1-3 of the reference's sequence in ONE
// kernel (MobileNet.c:240-498: convolute 3x3x3 s2 -> depthwise 3x3 s1 -> pointwise 1x1), each with its folded-BN
// scale/shift and ReLU6. SURVEY.md §8f ranks the fused depthwise->pointwise block first among the "next" items; the
// stem is where it pays most: the three layers are all HBM-bound (AI 9.8 / 2.2 / 10.7 flop/B) and unfused they move
// 154+411, 411+411 and 411+822 MB per 256 images; fused, the 112x112x32 intermediates never leave the CU:
// read 154 MB, write 822 MB.
//
// One workgroup = one tile of 8 x 16 output pixels (= 128 rows of the pointwise GEMM), persistent over tiles:
//   A. input patch 21 x 37 x 3 floats (stride-2 footprint of the 10 x 18 conv1 halo region) -> LDS, coalesced rows,
//      zero outside the image (TF-SAME bottom/right pad of conv1);
//   B. conv1 + BN + ReLU6 for the 10 x 18 region -> LDS [pixel][32]; pixels outside the 112 x 112 map are written as
//      ZERO (they are the depthwise layer's zero padding, not conv outputs); a lane computes 4 channels of 2 adjacent
//      pixels so each LDS weight float4 feeds 8 FMAs;
//   C. depthwise 3x3 + BN + ReLU6 for the 8 x 16 tile -> LDS as the pointwise GEMM's A tile [128][32], 16-byte chunks
//      XOR-swizzled exactly like mbn_f32_pw.hip (conflict-free ds_read_b128 for the MFMA fragments);
//   D. pointwise 32 -> 64 on v_mfma_f32_32x32x2_f32 (wave w owns rows 32w..32w+31, all 64 columns) + BN + ReLU6,
//      stored as whole 128-byte lines.
// The three filters and their scale/shift vectors stay in LDS / registers for the workgroup's lifetime.
// Shapes: Cin 3; conv1 -> C1, pointwise -> C3 with (C1, C3) = (32, 64) (alpha = 1) or (16, 32) (alpha = 0.5: BASELINE
// config 5's 0.5x160, where the three layers were 40 % of the step as separate launches); input side a multiple of 32.
// The kernel is a template over (C1, C3): with 16 channels a lane still owns 4 of them, so there are 4 channel quads per
// pixel instead of 8 and every phase hands a lane half as many pixels (3 instead of 6 in conv1, 2 instead of 4 in the
// depthwise) to keep all 256 lanes busy; A/B tile rows are 64 instead of 128 bytes (four 16-byte slots, their own swizzle).
#include "mbn_internal.h"
#include "mbn_epilogue.h"

namespace {

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));

constexpr int TH = 8, TW = 16;                 // output tile (pixels of the 112x112 map)
constexpr int CR = TH + 2, CC = TW + 2;        // conv1 region incl. the depthwise halo: 10 x 18
constexpr int PR = 2 * CR + 1;                     // input patch: 21 rows x 37 pixels
constexpr int PROW = 112;                      // floats per patch row in LDS (37*3 = 111, padded)
constexpr int PROWPAD = 4;                     // tail pad of the patch: the last lane's 16-byte reads run 2 floats past a row

struct StemArgs {

    float *out;
    const float *in, *w1, *s1, *b1, *wd, *s2, *b2, *wp, *s3, *b3;
    const uint8_t *in8;         // raw uint8 HWC image instead of `in`: normalised at load, x/127.5 - 1 (MBN_IO_IN_U8)
    int batch, res, h;          // input side, conv1/dw/pw side (res/2)
    int tiles_y, tiles_x;
    unsigned ntiles;        // < 2^31 (launcher checks): tile indices stay 32-bit, the per-tile index math is scalar and cheap
};

// fp32 A/B tiles [rows][C1]: 16-byte slots XOR-swizzled so the ds_read_b128 of 16 consecutive rows hit 16 distinct bank
// quads. C1 = 32: 128-byte rows, 8 slots, slot ^ (row>>1); C1 = 16: 64-byte rows, 4 slots, slot ^ (row>>2).
template <int C1>
__device__ __forceinline__ int swz(int row, int chunk)
{
    if (C1 == 32) return (row << 5) + (((chunk ^ (row >> 1)) & 7) << 2);
    return (row << 4) + (((chunk ^ (row >> 2)) & 3) << 2);
}
// bf16 mode: A and B tiles hold bf16, 32 channels = 64-byte rows = four 16-byte slots; slot XOR (row>>2)&3 makes the
// ds_read_b128 of 16 consecutive rows hit 16 distinct bank quads ((row&3)*4 + slot'). Returns the 4-byte-word offset.
// (C1 = 16: 32-byte rows, two slots, slot ^ (row>>3)&1: rows r and r+8 share a bank quad pair and take opposite halves.)
template <int C1>
__device__ __forceinline__ int swzb(int row, int slot)
{
    if (C1 == 32) return (row << 4) + (((slot ^ (row >> 2)) & 3) << 2);
    return (row << 3) + (((slot ^ (row >> 3)) & 1) << 2);
}
__device__ __forceinline__ float relu6(float v) { return fminf(fmaxf(v, 0.f), 6.f); }
__device__ __forceinline__ f4 bn_relu6(f4 a, f4 s, f4 b)
{
    return f4{ relu6(fmaf(a.x, s.x, b.x)), relu6(fmaf(a.y, s.y, b.y)), relu6(fmaf(a.z, s.z, b.z)), relu6(fmaf(a.w, s.w, b.w)) };
}
// acc += x * w on four channels: one fused multiply-add per channel (same rounding as the unfused kernels' fmaf), written
// on vectors so the compiler can pair channels into v_pk_fma_f32
__device__ __forceinline__ f4 fma4(float x, f4 w, f4 acc) { return __builtin_elementwise_fma(f4{ x, x, x, x }, w, acc); }
__device__ __forceinline__ f4 fma4v(f4 x, f4 w, f4 acc) { return __builtin_elementwise_fma(x, w, acc); }

constexpr int PAIRS = PROW / 2;                // float2 per patch row (56)
constexpr int NPF = (PR * PAIRS + 255) / 256;  // float2 prefetch registers per lane (5)

// Phase A, first half: this lane's float2 pieces of tile t's input patch (rows 16ty-2 .. 16ty+18, floats 96tx-6 ..
// 96tx+105 of each row), zero outside the image. Pair boundaries never straddle the image edge (96tx-6 and 3*res even).
// Branch-free: buffer loads against this image's descriptor; a piece outside the image (or past the patch) carries an out-of-range
// offset and the hardware returns zeros (the exec-masked form compiled to two branches per load and a vmcnt(0) drain behind them).
__device__ __forceinline__ void patch_load(const StemArgs &a, unsigned t, int tid, f2 (&pf)[NPF])
{
    const int tx = (int)(t % (unsigned)a.tiles_x);
    const unsigned q0 = t / (unsigned)a.tiles_x;
    const int ty = (int)(q0 % (unsigned)a.tiles_y);
    const long n = q0 / (unsigned)a.tiles_y;
    const long img = n * a.res * a.res * 3;
    const int iy0 = 2 * (TH * ty - 1), fx0 = 6 * (TW * tx - 1), rowf = a.res * 3;
    constexpr unsigned OOB = 0xF0000000u;
    const unsigned img_elems = (unsigned)(a.res * a.res * 3);
    if (a.in8) {
        const __amdgpu_buffer_rsrc_t rsrc = mbn_make_rsrc(a.in8 + img, img_elems);
#pragma unroll
        for (int k = 0; k < NPF; k++) {
            const int i = tid + k * 256, r = i / PAIRS, j = i % PAIRS;
            const int iy = iy0 + r, fx = fx0 + 2 * j;
            const bool ok = i < PR * PAIRS && iy >= 0 && iy < a.res && fx >= 0 && fx < rowf;
            const unsigned u = (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rsrc, ok ? (unsigned)(iy * rowf + fx) : OOB, 0, 0);
            // same fmaf as normalize_u8_f32: bit-identical; the zero padding is zero AFTER normalisation
            pf[k] = ok ? f2{ fmaf((float)(u & 0xff), 1.0f / 127.5f, -1.0f), fmaf((float)(u >> 8), 1.0f / 127.5f, -1.0f) } : f2{ 0.f, 0.f };
        }
    } else {
        const __amdgpu_buffer_rsrc_t rsrc = mbn_make_rsrc(a.in + img, img_elems * 4u);
#pragma unroll
        for (int k = 0; k < NPF; k++) {
            const int i = tid + k * 256, r = i / PAIRS, j = i % PAIRS;
            const int iy = iy0 + r, fx = fx0 + 2 * j;
            const bool ok = i < PR * PAIRS && iy >= 0 && iy < a.res && fx >= 0 && fx < rowf;
            pf[k] = __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, ok ? (unsigned)(iy * rowf + fx) * 4u : OOB, 0, 0));
        }
    }
}
__device__ __forceinline__ void patch_store(float *in_s, int tid, const f2 (&pf)[NPF])
{
#pragma unroll
    for (int k = 0; k < NPF; k++) {
        const int i = tid + k * 256;
        if (i < PR * PAIRS) *reinterpret_cast<f2 *>(in_s + 2 * i) = pf[k];      // row r, pair j -> r*PROW + 2j = 2i
    }
}

// BF = bf16 mode of the network (BASELINE config 5): storage bf16, arithmetic fp32. Every layer OUTPUT is rounded to bf16
// (RNE) — here that means the two on-chip intermediates are rounded before they are written to LDS, exactly where the
// separate launches would store them — the pointwise filter is the bf16 copy, and the result is stored as bf16.
__device__ __forceinline__ float rbf(float v) { return (float)(__bf16)v; }
__device__ __forceinline__ f4 rbf4(f4 v) { return f4{ rbf(v.x), rbf(v.y), rbf(v.z), rbf(v.w) }; }
// exact three-way bf16 split of fp32 values (mbn_f32_pw_x6.hip: h = bf16(x), m = bf16(x - h), l = x - h - m), packed two per word
__device__ __forceinline__ void x6_split2(float x0, float x1, unsigned &h, unsigned &m, unsigned &l)
{
    typedef float f2e __attribute__((ext_vector_type(2)));
    typedef __bf16 b2e __attribute__((ext_vector_type(2)));
    const f2e v = f2e{ x0, x1 };
    const b2e hh = __builtin_convertvector(v, b2e);
    const f2e r = v - __builtin_convertvector(hh, f2e);
    const b2e mm = __builtin_convertvector(r, b2e);
    const f2e lo = r - __builtin_convertvector(mm, f2e);
    h = __builtin_bit_cast(unsigned, hh);
    m = __builtin_bit_cast(unsigned, mm);
    l = __builtin_bit_cast(unsigned, __builtin_convertvector(lo, b2e));
}
__device__ __forceinline__ void x6_split4(f4 v, unsigned (&h)[2], unsigned (&m)[2], unsigned (&l)[2])
{
    x6_split2(v.x, v.y, h[0], m[0], l[0]);
    x6_split2(v.z, v.w, h[1], m[1], l[1]);
}
__device__ __forceinline__ void x6_split8(f4 v0, f4 v1, unsigned (&h)[4], unsigned (&m)[4], unsigned (&l)[4])
{
    x6_split2(v0.x, v0.y, h[0], m[0], l[0]);
    x6_split2(v0.z, v0.w, h[1], m[1], l[1]);
    x6_split2(v1.x, v1.y, h[2], m[2], l[2]);
    x6_split2(v1.z, v1.w, h[3], m[3], l[3]);
}

// WPE = workgroups per CU (= waves per SIMD). The alpha = 1 forms need 59.6 KB (fp32) / 48.7 KB (bf16: A/B tiles in bf16) of LDS
// and 246 / 232 VGPRs: two. With the nine depthwise tap vectors re-read from LDS per tile instead of living in 36 VGPRs for the
// whole kernel (WDL, WPE >= 3) the bf16 form fits three (150 VGPRs) and the alpha = 0.5 forms four (118 / 126 VGPRs, 28.6 /
// 33.7 KB): more workgroups put one's conv1 phase (VALU) under the others' barriers and stores. Measured (batch 512,
// profiles/r02/j_stem_occupancy.txt): alpha = 0.5 bf16 0.196 -> 0.153 ms, fp32 0.221 -> 0.173 ms with four; alpha = 1 bf16:
// see the launcher.
// X6 = 6 | 9 (fp32 mode only, opt-in: mbn_tune_set("pw_emul", ...)): phase D forms its fp32 products from the exact three-way bf16
// split of both operands (mbn_f32_pw_x6.hip): the depthwise output is split on its way into LDS (three bf16 planes), the filter once
// per workgroup, and C1/16 x 6 (9) v_mfma_f32_32x32x16_bf16 per column block replace C1/2 fp32 MFMAs of twice the duration.
template <int C1, int C3, bool BF, int WPE, bool MC = false, int X6 = 0, bool PADC = true>
__global__ __launch_bounds__(256, WPE) void stem_fused_f32(StemArgs a)
{
    static_assert(!MC || BF || C1 == 32, "conv1 on the MFMA: bf16 mode (16x16x32 bf16, split operands), or fp32 alpha = 1 (16x16x4 fp32)");
    constexpr bool MCF = MC && !BF;                // round 4: conv1 as a K = 27 -> 28 GEMM on v_mfma_f32_16x16x4_f32 (fp32 products, fp32 sums)
    static_assert(X6 == 0 || (!BF && (X6 == 6 || X6 == 9)), "split products: fp32 mode, 6 or 9 of them");
    constexpr int APL = TH * TW * C1 / 2, BPL = C3 * C1 / 2;       // X6: floats per bf16 plane of the A / B tile
    constexpr int MH = C1 / 16;                    // MC: 16-channel halves (2 / 1)
    constexpr bool WDL = WPE >= 3;
    constexpr int Q1 = C1 / 4;                     // channel quads per pixel (8 / 4)
    constexpr int PB = 3 * Q1 / 4;                 // conv1 pixels per lane (6 / 3): CR * (CC / PB) * Q1 = 240 busy lanes
    static_assert(CC % PB == 0 && CR * (CC / PB) * Q1 <= 256, "conv1 lane mapping");
    constexpr int NXV = (3 * (2 * PB + 1) + 3) / 4;    // float4 reads covering the (2 PB + 1) input pixels x 3 channels of a row (10 / 6)
    constexpr int PC = 128 * Q1 / 256;             // depthwise pixels per lane (4 / 2)
    constexpr int NI = C3 / 32;                    // 32-column blocks of the pointwise output (2 / 1)
    constexpr int KG = C1 / 8;                     // fp32 MFMA k-groups of 8 (4 / 2)
    // fp32 alpha = 1 at three workgroups per CU: the pointwise filter's MFMA fragments (the same 8 float4 per lane for every tile) live in
    // 32 VGPRs instead of an 8 KB LDS tile — with the taps in LDS (WDL) that is 52.7 KB and <= 168 VGPRs: a third workgroup fits
    constexpr bool BREG = !BF && X6 == 0 && WPE >= 3 && C1 == 32;
    __shared__ __attribute__((aligned(16))) float in_s[PR * PROW + PROWPAD]; //  9.4 KB
    __shared__ __attribute__((aligned(16))) float w1_s[27 * C1];          //  3.4 KB
    // conv1 output [pixel][C1P]: with conv1 on the fp32 matrix cores (MCF) a lane stores 16 bytes of ONE pixel and the 8 lanes of a ds_write_b128
    // group are 8 consecutive pixels: at the natural 128-byte pixel stride all on one bank group (8-way: round 5 counted SQ_LDS_BANK_CONFLICT at
    // 0.61 of SQ_LDS_IDX_ACTIVE for this kernel), and the depthwise phase's ds_read_b128 (lanes of a 16-lane group = 4 pixels 4 columns apart x 4
    // channel quads) 2-way. At 40 floats per pixel (10 sixteen-byte units) the 16 lanes of every read group fall on 16 different units and the writes
    // are 2-way (r6; profiles/r06/j_*). The other forms keep the dense rows their lane maps were built for.
    constexpr int C1P = (MCF && C1 == 32 && PADC) ? 40 : C1;    // PADC = false: lab A/B (conv_variant = 6), the dense rows of round 4-5
    __shared__ __attribute__((aligned(16))) float c1_s[CR * CC * C1P];    // 22.5 KB (28.1 KB padded)
    __shared__ __attribute__((aligned(16))) float a_s[X6 ? 3 * APL : TH * TW * C1 / (BF ? 2 : 1)];   // 16 KB (fp32, C1 = 32) ... 4 KB (bf16, C1 = 16); X6: three bf16 planes, 24 KB
    __shared__ __attribute__((aligned(16))) float b_s[BREG ? 4 : X6 ? 3 * BPL : C3 * C1 / (BF ? 2 : 1)];        //  8 KB ... 1 KB; X6: 12 KB; BREG: none
    __shared__ __attribute__((aligned(16))) float sb_s[4 * C1];           // s1 | b1 | s2 | b2
    __shared__ __attribute__((aligned(16))) float wd_s[WDL ? 9 * C1 : 4]; // depthwise taps [ky][kx][C1] (WDL)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int c4 = tid % Q1;                                              // this lane's channel quad in phases B, C

    // ---- per-workgroup constants
    for (int i = tid * 4; i < 27 * C1; i += 1024) *reinterpret_cast<f4 *>(w1_s + i) = *reinterpret_cast<const f4 *>(a.w1 + i);
    if (BF) {
        // bf16 B tile [C3][C1] bf16; with two column blocks (C3 = 64) LDS row rho holds output channel 2*(rho&31) + (rho>>5)
        // (channel-paired column blocks: lane l's two accumulators are the adjacent channels 2l, 2l+1 -> packed 4-byte stores)
        constexpr int SL = C1 / 8;                                        // 16-byte slots per row (4 / 2)
        if (tid < C3 * SL) {
            const int row = tid / SL, slot = tid % SL, ch = NI == 2 ? 2 * (row & 31) + (row >> 5) : row;
            *reinterpret_cast<f4 *>(b_s + swzb<C1>(row, slot)) =
                *reinterpret_cast<const f4 *>(reinterpret_cast<const __bf16 *>(a.wp) + ch * C1 + slot * 8);
        }
    } else if constexpr (X6 != 0) {
        constexpr int SL = C1 / 8;                                        // 16-byte slots per plane row (4 / 2)
        if (tid < C3 * SL) {                                              // 8 consecutive k of one filter row -> the three planes
            const int row = tid / SL, slot = tid % SL;
            const float *src = a.wp + row * C1 + slot * 8;
            unsigned hw[4], mw[4], lw[4];
            x6_split8(*reinterpret_cast<const f4 *>(src), *reinterpret_cast<const f4 *>(src + 4), hw, mw, lw);
            typedef unsigned u4e __attribute__((ext_vector_type(4)));
            *reinterpret_cast<u4e *>(b_s + swzb<C1>(row, slot)) = u4e{ hw[0], hw[1], hw[2], hw[3] };
            *reinterpret_cast<u4e *>(b_s + BPL + swzb<C1>(row, slot)) = u4e{ mw[0], mw[1], mw[2], mw[3] };
            *reinterpret_cast<u4e *>(b_s + 2 * BPL + swzb<C1>(row, slot)) = u4e{ lw[0], lw[1], lw[2], lw[3] };
        }
    } else if constexpr (!BREG)
    for (int i = tid; i < C3 * Q1; i += 256) {                            // pointwise filter [C3][C1] -> swizzled B tile
        const int row = i / Q1, ch = i % Q1;
        *reinterpret_cast<f4 *>(b_s + swz<C1>(row, ch)) = *reinterpret_cast<const f4 *>(a.wp + row * C1 + ch * 4);
    }
    [[maybe_unused]] f4 bfr[BREG ? KG : 1][NI];                           // BREG: filter rows ni*32 + li, k = 8g + 4lh .. +3
    if constexpr (BREG) {
#pragma unroll
        for (int g = 0; g < KG; g++)
#pragma unroll
            for (int ni = 0; ni < NI; ni++) bfr[g][ni] = *reinterpret_cast<const f4 *>(a.wp + (ni * 32 + (tid & 31)) * C1 + (2 * g + ((tid & 63) >> 5)) * 4);
    }
    if (tid < 4 * C1) {
        const float *src = tid < C1 ? a.s1 : tid < 2 * C1 ? a.b1 : tid < 3 * C1 ? a.s2 : a.b2;
        sb_s[tid] = src[tid & (C1 - 1)];
    }
    f4 wd[WDL ? 1 : 9];
    if constexpr (WDL) {
        for (int i = tid * 4; i < 9 * C1; i += 1024) *reinterpret_cast<f4 *>(wd_s + i) = *reinterpret_cast<const f4 *>(a.wd + i);
    } else {
#pragma unroll
        for (int k = 0; k < 9; k++) wd[k] = *reinterpret_cast<const f4 *>(a.wd + k * C1 + c4 * 4);
    }
    float s3[NI], b3[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ni++) {                                     // bf16, two blocks: accumulator ni of lane li is channel 2*li + ni
        const int ch = (BF && NI == 2) ? 2 * li + ni : ni * 32 + li;
        s3[ni] = a.s3[ch]; b3[ni] = a.b3[ch];
    }

    f2 pf[NPF];
    if (blockIdx.x < a.ntiles) {
        patch_load(a, blockIdx.x, tid, pf);
        patch_store(in_s, tid, pf);
    }
    __syncthreads();

    // ---- MC (bf16 mode, C1 = 32): conv1 as a GEMM on v_mfma_f32_16x16x32_bf16. The 10 x 18 region is 12 blocks of 16 pixels (three
    // per wave), K = 27 taps padded to 32 = one instruction deep, the 32 channels two instructions wide. A lane gathers its
    // pixel's 8 taps k = 8 kg .. 8 kg + 7 from the fp32 patch (k = 9 ky + 3 kx + ci sits at patch row 2r + ky, float 6c + k % 9),
    // splits them exactly into bf16 hi + bf16 lo, and the products are hi*Whi + lo*Whi + hi*Wlo (the dropped lo*Wlo is 2^-16 of
    // a product; the result is rounded to bf16 = 2^-9 right after). The weight operands are gathered and split once per kernel.
    // In fp32 mode this form costs the same SIMD cycles as the packed VALU form (fp32 MFMA = VALU rate); in bf16 it is
    // ~1100 against ~2500 cycles per wave and tile.
    [[maybe_unused]] int mc_dl[8];
    [[maybe_unused]] unsigned mc_kok = 0;                                     // bit j: tap k = 8 kg + j exists (k < 27)
    [[maybe_unused]] bf8 mc_wh[MH], mc_wl[MH];
    [[maybe_unused]] f4 mc_s1[MH], mc_b1[MH];                                    // BN of the lane's 4 output channels 16h + 4kg .. +3
    // ---- MCF (fp32 mode, C1 = 32): the same GEMM view on v_mfma_f32_16x16x4_f32, K = 27 taps padded to 28 = 7 instructions deep. Weights are the
    // A operand (rows = 16 channels of half h, k = 4 t + kg), the 16 pixels of a block the B operand: C/D puts channels 16 h + 4 kg .. + 3 of pixel pc
    // into this lane — one float4 of BN, one 16-byte LDS store, as in the bf16 form. 42 MFMAs of 32 cycles per wave and tile replace 324 v_pk_fma_f32 +
    // 57 ds_read_b128: on a SIMD whose fp32 MFMA and VALU issue times add (profiles/r04/b_pmc_fp32_step_all_passes.txt: the stem is issue-bound, 763
    // vector instructions per wave and tile) the matrix form is the cheaper instruction stream. Its sums are NOT those of the fmaf chain of the VALU
    // form (profiles/r04/d_mfma_vs_fmaf_chain.txt), so the stand-alone conv1 kernel uses the same instruction and k order when this form ships.
    [[maybe_unused]] int mcf_dl[7];
    [[maybe_unused]] float mcf_w[2][7];
    [[maybe_unused]] f4 mcf_s1[2], mcf_b1[2];
    if constexpr (MCF) {
        const int kg = lane >> 4, pc = lane & 15;
#pragma unroll
        for (int t = 0; t < 7; t++) {
            const int k = 4 * t + kg;
            mcf_dl[t] = k < 27 ? (k / 9) * PROW + (k % 9) : 0;
#pragma unroll
            for (int h = 0; h < 2; h++) mcf_w[h][t] = k < 27 ? w1_s[k * C1 + 16 * h + pc] : 0.f;
        }
#pragma unroll
        for (int h = 0; h < 2; h++) {
            mcf_s1[h] = *reinterpret_cast<const f4 *>(sb_s + 16 * h + 4 * kg);
            mcf_b1[h] = *reinterpret_cast<const f4 *>(sb_s + C1 + 16 * h + 4 * kg);
        }
    }
    if constexpr (MC && BF) {
        const int kg = lane >> 4, pc = lane & 15;
        unsigned whb[MH][8], wlb[MH][8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int k = 8 * kg + j;
            mc_dl[j] = (k / 9) * PROW + (k % 9);
            if (k < 27) mc_kok |= 1u << j;
#pragma unroll
            for (int h = 0; h < MH; h++) {
                const float w = k < 27 ? w1_s[k * C1 + 16 * h + pc] : 0.f;
                const unsigned wb = __builtin_bit_cast(unsigned, w) & 0xffff0000u;
                whb[h][j] = wb;
                wlb[h][j] = __builtin_bit_cast(unsigned, w - __builtin_bit_cast(float, wb));
            }
        }
#pragma unroll
        for (int h = 0; h < MH; h++) {
            typedef unsigned u4 __attribute__((ext_vector_type(4)));
            u4 ph, pl;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                ph[i] = (whb[h][2 * i] >> 16) | (whb[h][2 * i + 1] & 0xffff0000u);
                pl[i] = (wlb[h][2 * i] >> 16) | (wlb[h][2 * i + 1] & 0xffff0000u);
            }
            mc_wh[h] = __builtin_bit_cast(bf8, ph);
            mc_wl[h] = __builtin_bit_cast(bf8, pl);
            mc_s1[h] = *reinterpret_cast<const f4 *>(sb_s + 16 * h + 4 * kg);
            mc_b1[h] = *reinterpret_cast<const f4 *>(sb_s + C1 + 16 * h + 4 * kg);
        }
    }

    // phase B item of this lane: conv1 row br, pixels bc .. bc+PB-1 (lanes 240..255 have none)
    const int bpg = tid / Q1, br = bpg / (CC / PB), bc = (bpg % (CC / PB)) * PB;
    // phase C item: tile row cy, pixels cx .. cx+PC-1
    const int cy = (tid / Q1) / (TW / PC), cx = ((tid / Q1) % (TW / PC)) * PC;

    for (unsigned t = blockIdx.x; t < a.ntiles; t += gridDim.x) {
        const int tx = (int)(t % (unsigned)a.tiles_x);
        const unsigned q0 = t / (unsigned)a.tiles_x;
        const int ty = (int)(q0 % (unsigned)a.tiles_y);
        const long n = q0 / (unsigned)a.tiles_y;
        const unsigned tnext = t + gridDim.x;
        if (tnext < a.ntiles) patch_load(a, tnext, tid, pf);              // in flight while phase B computes

        // ---- B. conv1 (3x3x3, stride 2, pad 0 top/left) + BN + ReLU6 over the 10 x 18 region, 6 pixels x 4 ch per lane
        __builtin_amdgcn_s_setprio(2);                              // VALU phases ahead of the co-resident workgroups' MFMA phase (-3 % / -5 %: profiles/r02/j_stem_occupancy.txt)
        if constexpr (MCF) {
            // one 16-pixel block at a time (measured: the three blocks interleaved — 21 gathers first, six accumulator chains — is 2 % SLOWER, 0.282-0.284
            // against 0.277-0.279 ms: it holds 20 more VGPRs live through the phase, and the co-resident workgroup fills the chain's gaps anyway)
            const int kg = lane >> 4, pc = lane & 15;
#pragma unroll 1
            for (int bi = 0; bi < 3; bi++) {
                const int blk = wave * 3 + bi;                            // 16-pixel block 0..11 of the region (pixels 180..191 do not exist)
                const int q = 16 * blk + pc, p = min(q, CR * CC - 1);
                const int r = (p * 3641) >> 16, c = p - r * CC;           // p / 18 for p < 192
                const float *src = in_s + (2 * r) * PROW + 6 * c;
                float xv[7];
#pragma unroll
                for (int t = 0; t < 7; t++) xv[t] = src[mcf_dl[t]];
                if (kg == 3) xv[6] = 0.f;                                 // k = 27 does not exist (its weight is 0 too)
                const int oy = TH * ty - 1 + r, ox = TW * tx - 1 + c;     // position of THIS lane's pixel in the conv1 map
                const bool inside = oy >= 0 && oy < a.h && ox >= 0 && ox < a.h;
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    f4 acc = f4{ 0.f, 0.f, 0.f, 0.f };                    // same instruction, same k order as conv1_mfma_f32 (mbn_f32_misc.hip): same bits
#pragma unroll
                    for (int t = 0; t < 7; t++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(mcf_w[h][t], xv[t], acc, 0, 0, 0);
                    const f4 v = inside ? bn_relu6(acc, mcf_s1[h], mcf_b1[h]) : f4{ 0.f, 0.f, 0.f, 0.f };   // outside: the depthwise zero padding
                    if (q < CR * CC) *reinterpret_cast<f4 *>(c1_s + q * C1P + 16 * h + 4 * kg) = v;
                }
            }
        } else
        if constexpr (MC) {
            typedef unsigned u4 __attribute__((ext_vector_type(4)));
            const int kg = lane >> 4, pc = lane & 15;
#pragma unroll 1
            for (int bi = 0; bi < 3; bi++) {
                const int blk = wave * 3 + bi;                            // 16-pixel block 0..11 of the region (pixels 180..191 do not exist)
                const int q = 16 * blk + pc, p = min(q, CR * CC - 1);
                const int r = (p * 3641) >> 16, c = p - r * CC;           // p / 18 for p < 192
                const float *src = in_s + (2 * r) * PROW + 6 * c;
                unsigned xb[8], lb[8];
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const float x = ((mc_kok >> j) & 1u) ? src[mc_dl[j]] : 0.f;
                    xb[j] = __builtin_bit_cast(unsigned, x) & 0xffff0000u;
                    lb[j] = __builtin_bit_cast(unsigned, x - __builtin_bit_cast(float, xb[j]));
                }
                u4 ph, pl;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    ph[i] = (xb[2 * i] >> 16) | (xb[2 * i + 1] & 0xffff0000u);
                    pl[i] = (lb[2 * i] >> 16) | (lb[2 * i + 1] & 0xffff0000u);
                }
                const bf8 xh = __builtin_bit_cast(bf8, ph), xl = __builtin_bit_cast(bf8, pl);
                const int oy = TH * ty - 1 + r, ox = TW * tx - 1 + c;     // position of THIS lane's pixel in the conv1 map
                const bool inside = oy >= 0 && oy < a.h && ox >= 0 && ox < a.h;
#pragma unroll
                for (int h = 0; h < MH; h++) {
                    // weights as the A operand (rows = channels), pixels as B (columns): C/D puts channels 16h + 4kg .. +3 of
                    // pixel pc into this lane: one float4 of BN, one 16-byte LDS store, the gather's (r, c) reused
                    f4 acc = f4{ 0.f, 0.f, 0.f, 0.f };
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(mc_wl[h], xh, acc, 0, 0, 0);      // smallest terms first
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(mc_wh[h], xl, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(mc_wh[h], xh, acc, 0, 0, 0);
                    const f4 v = inside ? rbf4(bn_relu6(acc, mc_s1[h], mc_b1[h])) : f4{ 0.f, 0.f, 0.f, 0.f };   // outside: the depthwise zero padding
                    if (q < CR * CC) *reinterpret_cast<f4 *>(c1_s + q * C1 + 16 * h + 4 * kg) = v;
                }
            }
        } else
        if (bpg < CR * (CC / PB)) {
            f4 acc[PB];
#pragma unroll
            for (int p = 0; p < PB; p++) acc[p] = f4{ 0.f, 0.f, 0.f, 0.f };
#pragma unroll 1
            for (int ky = 0; ky < 3; ky++) {                              // not unrolled: keeps the live set small
                const float *row = in_s + (2 * br + ky) * PROW + (2 * bc) * 3;    // 2 PB + 1 pixels x 3 channels (+ pad floats)
                // the row segment starts at float 6*bc: a multiple of 16 bytes for PB = 6 (bc = 0, 6, 12), only of 8 bytes for
                // PB = 3 (bc = 3, 9, 15): 16-byte LDS reads there, 8-byte reads here
                float x[4 * NXV];
                if constexpr (PB == 6) {
#pragma unroll
                    for (int j = 0; j < NXV; j++) {
                        const f4 t = *reinterpret_cast<const f4 *>(row + 4 * j);
                        x[4 * j] = t.x; x[4 * j + 1] = t.y; x[4 * j + 2] = t.z; x[4 * j + 3] = t.w;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 2 * NXV - 1; j++) {
                        const f2 t = *reinterpret_cast<const f2 *>(row + 2 * j);
                        x[2 * j] = t.x; x[2 * j + 1] = t.y;
                    }
                }
#pragma unroll
                for (int kx = 0; kx < 3; kx++)
#pragma unroll
                    for (int ci = 0; ci < 3; ci++) {
                        const f4 w = *reinterpret_cast<const f4 *>(w1_s + ((ky * 3 + kx) * 3 + ci) * C1 + c4 * 4);
#pragma unroll
                        for (int p = 0; p < PB; p++) acc[p] = fma4(x[(2 * p + kx) * 3 + ci], w, acc[p]);
                    }
            }
            const int oy = TH * ty - 1 + br, ox = TW * tx - 1 + bc;      // position in the 112x112 conv1 map
            const bool rowok = oy >= 0 && oy < a.h;
            const f4 s1 = *reinterpret_cast<const f4 *>(sb_s + c4 * 4), b1 = *reinterpret_cast<const f4 *>(sb_s + C1 + c4 * 4);
#pragma unroll
            for (int p = 0; p < PB; p++) {                                // outside the map: the depthwise zero padding
                f4 v = (rowok && ox + p >= 0 && ox + p < a.h) ? bn_relu6(acc[p], s1, b1) : f4{ 0.f, 0.f, 0.f, 0.f };
                if (BF) v = rbf4(v);
                *reinterpret_cast<f4 *>(c1_s + (br * CC + bc + p) * C1 + c4 * 4) = v;
            }
        }
        __syncthreads();
        if (tnext < a.ntiles) patch_store(in_s, tid, pf);                 // in_s was last read in B; next read after 2 barriers

        // ---- C. depthwise 3x3 (stride 1, pad 1) + BN + ReLU6: 4 adjacent pixels x 4 ch per lane -> swizzled A tile
        {
            f4 acc[PC];
#pragma unroll
            for (int p = 0; p < PC; p++) acc[p] = f4{ 0.f, 0.f, 0.f, 0.f };
#pragma unroll
            for (int dy = 0; dy < 3; dy++) {
                f4 v[PC + 2];
#pragma unroll
                for (int j = 0; j < PC + 2; j++) v[j] = *reinterpret_cast<const f4 *>(c1_s + ((cy + dy) * CC + cx + j) * C1P + c4 * 4);
#pragma unroll
                for (int dx = 0; dx < 3; dx++) {
                    f4 wt;
                    if constexpr (WDL) wt = *reinterpret_cast<const f4 *>(wd_s + (dy * 3 + dx) * C1 + c4 * 4);
                    else wt = wd[dy * 3 + dx];
#pragma unroll
                    for (int p = 0; p < PC; p++) acc[p] = fma4v(v[p + dx], wt, acc[p]);
                }
            }
            const f4 s2 = *reinterpret_cast<const f4 *>(sb_s + 2 * C1 + c4 * 4), b2 = *reinterpret_cast<const f4 *>(sb_s + 3 * C1 + c4 * 4);
#pragma unroll
            for (int p = 0; p < PC; p++) {
                const f4 v = bn_relu6(acc[p], s2, b2);
                if (BF) {                                                 // the layer output, rounded to bf16 (RNE): 8 bytes per lane
                    const int row = cy * TW + cx + p;
                    *reinterpret_cast<bf4 *>(a_s + swzb<C1>(row, c4 >> 1) + 2 * (c4 & 1)) = bf4{ (__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w };
                } else if constexpr (X6 != 0) {                           // exact split of the fp32 value: 8 bytes per lane and plane
                    const int o = swzb<C1>(cy * TW + cx + p, c4 >> 1) + 2 * (c4 & 1);
                    unsigned hw[2], mw[2], lw[2];
                    x6_split4(v, hw, mw, lw);
                    typedef unsigned u2e __attribute__((ext_vector_type(2)));
                    *reinterpret_cast<u2e *>(a_s + o) = u2e{ hw[0], hw[1] };
                    *reinterpret_cast<u2e *>(a_s + APL + o) = u2e{ mw[0], mw[1] };
                    *reinterpret_cast<u2e *>(a_s + 2 * APL + o) = u2e{ lw[0], lw[1] };
                } else *reinterpret_cast<f4 *>(a_s + swz<C1>(cy * TW + cx + p, c4)) = v;
            }
        }
        __syncthreads();

        // ---- D. pointwise C1 -> C3: wave w computes rows 32w .. 32w+31 x all C3 columns
        __builtin_amdgcn_s_setprio(0);
        f16v acc[NI];
#pragma unroll
        for (int ni = 0; ni < NI; ni++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[ni][r] = 0.f;
        if constexpr (BF) {
            // bf16 mode: both operands ARE bf16 (the depthwise output was rounded above, the filter is the bf16 copy), so the
            // C1-deep product is C1/16 v_mfma_f32_32x32x16_bf16 per column block instead of C1/2 fp32 MFMAs — the fp32 form made
            // the bf16 stem SLOWER per image than the fp32 stem
#pragma unroll
            for (int ks = 0; ks < C1 / 16; ks++) {
                const f4 av = *reinterpret_cast<const f4 *>(a_s + swzb<C1>(wave * 32 + li, 2 * ks + lh));
#pragma unroll
                for (int ni = 0; ni < NI; ni++) {
                    const f4 bv = *reinterpret_cast<const f4 *>(b_s + swzb<C1>(ni * 32 + li, 2 * ks + lh));
                    acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, av), __builtin_bit_cast(bf8, bv), acc[ni], 0, 0, 0);
                }
            }
            // Buffer stores as in mbn_epilogue.h: descriptor = this image's output map (< 4 GiB whatever the batch), the per-lane
            // byte offset is ONE VGPR for all 16 stores (the lane's half lh shifts the pixel by 4 columns: row q = 32 wave + (r & 3) +
            // 8 (r >> 2) + 4 lh sits at y = 2 wave + (r >> 3), x = (r & 3) + 8 ((r >> 2) & 1) + 4 lh of the 8 x 16 tile), the
            // wave-uniform rest is the scalar offset: no 64-bit address arithmetic per store.
            const __amdgpu_buffer_rsrc_t orsrc = mbn_make_rsrc(reinterpret_cast<__bf16 *>(a.out) + n * a.h * a.h * C3, (unsigned)(a.h * a.h * C3 * 2));
            const unsigned lane_off = (unsigned)(4 * lh * C3) * 2u + (unsigned)li * (NI == 2 ? 4u : 2u);
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int y = TH * ty + 2 * __builtin_amdgcn_readfirstlane(wave) + (r >> 3), x = TW * tx + (r & 3) + 8 * ((r >> 2) & 1);
                const unsigned soff = (unsigned)((y * a.h + x) * C3) * 2u;
                if constexpr (NI == 2) {
                    // channel-paired store: lane li holds channels 2li (acc[0]) and 2li+1 (acc[1]) of pixel row q: one dword per
                    // row, 32 lanes = the pixel's whole 128-byte line
                    const float v0 = relu6(fmaf(acc[0][r], s3[0], b3[0])), v1 = relu6(fmaf(acc[1][r], s3[1], b3[1]));
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, bf2{ (__bf16)v0, (__bf16)v1 }), orsrc, lane_off, soff, 0);
                } else {
                    __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, (__bf16)relu6(fmaf(acc[0][r], s3[0], b3[0]))), orsrc, lane_off, soff, 0);
                }
            }
        } else {
        if constexpr (X6 != 0) {
            // product list of mbn_f32_pw_x6.hip (plane of A, plane of B; 0 = h, 1 = m, 2 = l), smallest terms first
            constexpr int pa9[9] = { 2, 2, 1, 2, 0, 1, 1, 0, 0 }, pb9[9] = { 2, 1, 2, 0, 2, 1, 0, 1, 0 };
            constexpr int pa6[6] = { 2, 0, 1, 1, 0, 0 }, pb6[6] = { 0, 2, 1, 0, 1, 0 };
#pragma unroll
            for (int ks = 0; ks < C1 / 16; ks++) {
                f4 av[3], bv[3][NI];
#pragma unroll
                for (int pl = 0; pl < 3; pl++) {
                    av[pl] = *reinterpret_cast<const f4 *>(a_s + pl * APL + swzb<C1>(wave * 32 + li, 2 * ks + lh));
#pragma unroll
                    for (int ni = 0; ni < NI; ni++) bv[pl][ni] = *reinterpret_cast<const f4 *>(b_s + pl * BPL + swzb<C1>(ni * 32 + li, 2 * ks + lh));
                }
#pragma unroll
                for (int q = 0; q < X6; q++)
#pragma unroll
                    for (int ni = 0; ni < NI; ni++)
                        acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, av[X6 == 6 ? pa6[q % 6] : pa9[q]]),
                                                                          __builtin_bit_cast(bf8, bv[X6 == 6 ? pb6[q % 6] : pb9[q]][ni]), acc[ni], 0, 0, 0);
            }
        } else
#pragma unroll
        for (int g = 0; g < KG; g++) {
            const int chunk = 2 * g + lh;
            const f4 av = *reinterpret_cast<const f4 *>(a_s + swz<C1>(wave * 32 + li, chunk));
            f4 bv[NI];
#pragma unroll
            for (int ni = 0; ni < NI; ni++) {
                if constexpr (BREG) bv[ni] = bfr[g][ni];
                else bv[ni] = *reinterpret_cast<const f4 *>(b_s + swz<C1>(ni * 32 + li, chunk));
            }
#pragma unroll
            for (int s = 0; s < 4; s++)
#pragma unroll
                for (int ni = 0; ni < NI; ni++) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bv[ni][s], acc[ni], 0, 0, 0);
        }
        if constexpr (C1 == 32 && !BREG) {
            // alpha = 1 in fp32 keeps plain global stores: the buffer-store form below measured 1.2 % SLOWER here (0.3077 against 0.3040 ms,
            // same call, three repetitions) although it drops ~80 address instructions and 32 VGPRs; it wins at alpha = 0.5 (-6 %) and in bf16
            float *obase = a.out + ((n * a.h + TH * ty) * a.h + TW * tx) * C3;
#pragma unroll
            for (int ni = 0; ni < NI; ni++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int q = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    const int y = q >> 4, x = q & 15;
                    obase[((long)y * a.h + x) * C3 + ni * 32 + li] = relu6(fmaf(acc[ni][r], s3[ni], b3[ni]));
                }
        } else {
        const __amdgpu_buffer_rsrc_t orsrc = mbn_make_rsrc(a.out + n * a.h * a.h * C3, (unsigned)(a.h * a.h * C3 * 4));   // see the bf16 branch
        const unsigned lane_off = (unsigned)(4 * lh * C3 + li) * 4u;
#pragma unroll
        for (int ni = 0; ni < NI; ni++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int y = TH * ty + 2 * __builtin_amdgcn_readfirstlane(wave) + (r >> 3), x = TW * tx + (r & 3) + 8 * ((r >> 2) & 1);
                const unsigned soff = (unsigned)((y * a.h + x) * C3 + ni * 32) * 4u;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, relu6(fmaf(acc[ni][r], s3[ni], b3[ni]))), orsrc, lane_off, soff, 0);
            }
        }
        }
        // No barrier here: the next tile's B writes c1_s (last read in C, one barrier ago) and its C writes a_s only after
        // the barrier that follows B, which every wave reaches after finishing the a_s reads above.
    }
}

}   // namespace

// Fused layers 1-3. Returns MBN_EUNSUPPORTED when the shapes are not the alpha = 1 stem (the caller then runs the three
// layers separately).
int mbn_launch_f32_stem(mbn_context *ctx, hipStream_t stream, float *out, const float *in, const float *w1,
                        const float *s1, const float *b1, const float *wd, const float *s2, const float *b2,
                        const float *wp, const float *s3, const float *b3, int batch, int res, int c1, int c3, int in_u8, int bf16)
{
    if (!((c1 == 32 && c3 == 64) || (c1 == 16 && c3 == 32)) || res < 32 || (res % 32) != 0 || batch <= 0) return MBN_EUNSUPPORTED;
    const float *ptrs[] = { w1, s1, b1, wd, s2, b2, wp, s3, b3 };
    for (const float *p : ptrs)
        if (!p || ((uintptr_t)p % 16) != 0) return MBN_EUNSUPPORTED;
    if (!out || !in || ((uintptr_t)out % 16) != 0 || ((uintptr_t)in % (in_u8 ? 2 : 8)) != 0) return MBN_EINVAL;
    StemArgs a;
    a.in8 = in_u8 ? (const uint8_t *)in : nullptr;
    a.out = out; a.in = in; a.w1 = w1; a.s1 = s1; a.b1 = b1; a.wd = wd; a.s2 = s2; a.b2 = b2; a.wp = wp; a.s3 = s3; a.b3 = b3;
    a.batch = batch; a.res = res; a.h = res / 2;
    a.tiles_y = a.h / TH; a.tiles_x = a.h / TW;
    if ((long)batch * a.tiles_y * a.tiles_x >= 0x7fffffffL) return MBN_EUNSUPPORTED;
    a.ntiles = (unsigned)((long)batch * a.tiles_y * a.tiles_x);
    // alpha = 1 bf16: three workgroups per CU (taps from LDS) 0.448-0.455 ms against 0.472-0.480 with two (taps in registers), same run
    const int wpe_bf = g_mbn_tune.misc == 2 ? 2 : 3;                                  // A/B hook: misc = 2
    const int per_cu = c1 == 16 ? 4 : (bf16 ? wpe_bf : 2);
    long grid = (long)ctx->num_cus * per_cu;
    if (grid > (long)a.ntiles) grid = (long)a.ntiles;
    const dim3 g((unsigned)grid), b(256);
    if (c1 == 32) {
#ifdef MBN_LAB
        if (bf16 && wpe_bf == 3 && g_mbn_tune.conv_variant == 2) hipLaunchKernelGGL((stem_fused_f32<32, 64, true, 3>), g, b, 0, stream, a);     // conv1 on the VALU (A/B)
        else if (bf16 && wpe_bf != 3) hipLaunchKernelGGL((stem_fused_f32<32, 64, true, 2>), g, b, 0, stream, a);                                // two workgroups per CU (A/B)
        else
#endif
        if (bf16) hipLaunchKernelGGL((stem_fused_f32<32, 64, true, 3, true>), g, b, 0, stream, a);   // conv1 on the bf16 MFMA, three workgroups per CU
        else if (g_mbn_tune.pw_emul == 6) hipLaunchKernelGGL((stem_fused_f32<32, 64, false, 2, true, 6>), g, b, 0, stream, a);   // opt-in split products in phase D
        else if (g_mbn_tune.pw_emul == 9) hipLaunchKernelGGL((stem_fused_f32<32, 64, false, 2, true, 9>), g, b, 0, stream, a);
#ifdef MBN_LAB
        // r3 A/B (profiles/r03/k_stem_three_workgroups.txt): the fp32 alpha = 1 stem at three workgroups per CU — filter fragments in 32 VGPRs instead of
        // the 8 KB LDS tile, taps in LDS, buffer stores: 52.7 KB, 156 VGPRs. SLOWER: 0.370 ms against 0.307 (that kernel on a 2-per-CU grid,
        // conv_variant = 4: 0.338): the leaner code costs 10 % and the third workgroup another 10 %; not shipped
        else if (g_mbn_tune.conv_variant == 3 || g_mbn_tune.conv_variant == 4) {
            long g3 = (long)ctx->num_cus * (g_mbn_tune.conv_variant == 3 ? 3 : 2);
            if (g3 > (long)a.ntiles) g3 = (long)a.ntiles;
            hipLaunchKernelGGL((stem_fused_f32<32, 64, false, 3>), dim3((unsigned)g3), b, 0, stream, a);
        }
#endif
#ifdef MBN_LAB
        else if (g_mbn_tune.conv_variant == 6) hipLaunchKernelGGL((stem_fused_f32<32, 64, false, 2, true, 0, false>), g, b, 0, stream, a);   // r6 A/B: conv1 output rows unpadded (rounds 4-5)
        else if (g_mbn_tune.conv_variant == 5) hipLaunchKernelGGL((stem_fused_f32<32, 64, false, 2>), g, b, 0, stream, a);   // r4 A/B: conv1 on the VALU (round 3's form; timing only: its bits are the fmaf chain's, not conv1_mfma_f32's)
#endif
        // round 4: conv1 on v_mfma_f32_16x16x4_f32 (profiles/r04/f_stem_conv1_mfma.txt: 0.291-0.297 -> 0.277-0.279 ms at batch 256)
        else hipLaunchKernelGGL((stem_fused_f32<32, 64, false, 2, true>), g, b, 0, stream, a);
    } else {
#ifdef MBN_LAB
        if (bf16 && g_mbn_tune.conv_variant == 2) hipLaunchKernelGGL((stem_fused_f32<16, 32, true, 4>), g, b, 0, stream, a);     // conv1 on the VALU (A/B)
        else
#endif
        if (bf16) {            // MFMA conv1 + buffer-store epilogue: 92 VGPRs, 28.6 KB of LDS: five workgroups per CU (0.1163 -> 0.1108 ms)
            long g5 = (long)ctx->num_cus * 5;
            if (g5 > (long)a.ntiles) g5 = (long)a.ntiles;
            hipLaunchKernelGGL((stem_fused_f32<16, 32, true, 5, true>), dim3((unsigned)g5), b, 0, stream, a);
        }
        else if (g_mbn_tune.pw_emul == 6) hipLaunchKernelGGL((stem_fused_f32<16, 32, false, 4, false, 6>), g, b, 0, stream, a);
        else if (g_mbn_tune.pw_emul == 9) hipLaunchKernelGGL((stem_fused_f32<16, 32, false, 4, false, 9>), g, b, 0, stream, a);
        else hipLaunchKernelGGL((stem_fused_f32<16, 32, false, 4>), g, b, 0, stream, a);
    }
    return MBN_OK;
}
