// mbn_f32_dwpw3.hip — fused depthwise 3x3 -> pointwise 1x1 block for gfx950, fp32, WAVE-PRIVATE form (round 6).
// Same contract as mbn_f32_dwpw2.hip (one launch replaces a `depthwise` + `pointwise` pair of the reference's sequence,
// kernel.cl:62-92 + 94-114, pairs L4-5 ... of MobileNet.c:322-2599; bit-identical to the two separate launches):
//
//   out[m][n] = relu6( s3[n] * sum_c relu6( s2[c] * sum_{dy,dx} in[pix(m)+(dy,dx)][c] * wd[dy][dx][c] + b2[c] ) * wp[n][c] + b3[n] )
//
// Why a third form. The unified-wave kernel (dwpw2) hands the depthwise output of a 32-channel chunk from the 8 waves that
// produced it to the 8 waves that multiply it through a workgroup barrier — one barrier per 32 MFMAs per wave, with ONE
// workgroup per CU, so at every barrier the CU's matrix pipes drain and each SIMD's two waves wait for each other
// (round 5 stamps, profiles/r05/g_*: 1100-3500 cycles of wait + barrier in a 6700-cycle step whose MFMAs take 4096; the
// MFMA-only skeleton of that kernel runs at 1.33-1.48x its MFMA time, the stand-alone GEMM with four independent
// 4-wave workgroups per CU at 1.10x). Here no wave ever waits for another one:
//   * a wave owns 32 output pixels x 128 output channels (64 accumulators): it computes the depthwise output of ITS 32
//     pixels, 16 channels at a time (lane = 2 adjacent pixels x 4 channels, as in dwpw2), writes them into its private
//     2 KB A tile in LDS and reads them back as MFMA fragments — LDS operations of one wave execute in order, so the
//     hand-over needs no barrier and no wait beyond the read's own lgkmcnt;
//   * the pointwise filter slice [128 output channels][Cin] stays RESIDENT in LDS for the whole launch (<= 130 KB: Cin
//     <= 256), loaded once per workgroup: no filter DMA in the loop, no counted vmcnt, no barrier for it either. A
//     block with more than 128 output channels runs as Cout/128 slices in different workgroups (the depthwise part is
//     recomputed per slice: +9/128 of the slice's FMAs);
//   * the 8 waves of a workgroup (2 per SIMD, <= 256 VGPRs) are independent persistent pipelines over their own tile
//     sequences; the SIMD's issue logic interleaves one wave's MFMAs with the other's loads, LDS traffic and waits.
// Per 16-channel step and wave: 3 x (S+3) buffer_load_dwordx4 (one step ahead of their use), 36 v_pk_fma_f32 + 4 + 8
// (BN, clamp), 2 ds_write_b128, 2 + 8 fragment reads, 11 tap reads, 32 MFMAs — the k loop of a tile is fully unrolled
// (Cin is a template parameter), so every LDS address is a lane constant + an immediate and every x load a lane offset +
// an immediate: no address arithmetic inside a tile.
// Arithmetic order = mbn_f32_dw.hip (dy-major fma chain, BN, clamp) and mbn_f32_pw.hip (k pairs (8g+s, 8g+4+s) in
// increasing g, s on v_mfma_f32_32x32x2_f32): bit-identical to the two launches and to dwpw2.
// LDS images: filter rows padded to Cin + 4 floats (fragment reads of 16 lanes fall on 16 different 16-byte bank
// groups); A tile rows of 16 floats with the 16-byte unit XORed by (row >> 2) & 3; tile row = 16 * (pixel & 1) +
// (pixel >> 1), so both ds_write_b128 of a lane and the fragment ds_read_b128 are conflict-free (MI355X_MICROARCH.md
// §LDS lane groups). Epilogue: LDS filter row 32 t + l holds output channel 4 l + t, so lane l's four accumulator
// blocks are 4 adjacent channels: one buffer_store_dwordx4 per lane and row pair = 512 contiguous bytes per pixel.
#include "mbn_internal.h"
#include "mbn_epilogue.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef mbn_f16v f16v;

constexpr int BN3 = 128;                       // output channels per workgroup slice
constexpr int WT = 32;                         // output pixels per wave tile
constexpr int KS = 16;                         // channels per step

struct DwPw3Args {
    float *out;
    const float *in, *wd, *s2, *b2, *wp, *s3, *b3;
    long m;                 // output pixels = batch * ho * wo
    int h, w, ho, wo;       // input / output map sides
    int cout;
    int pad_top, pad_left;
    int nh;                 // 128-channel slices = cout / 128
    int tiles;              // wave tiles = ceil(m / 32)
    unsigned in_bytes;
    unsigned wo_m, wo_s, ho_m, ho_s;   // floor(v / wo) = umulhi(v, wo_m) >> wo_s for v < 2^31 (m == 0: the divisor is 1)
    float inv_wo, inv_ho;   // 1 / wo, 1 / ho
    int dbg;                // lab ablations (dwpw_variant = 300 + bits): 1 no x loads after the prologue, 2 no depthwise math, 4 no stores, 16 no MFMA,
                            // 32 no tap reads, 64 no BN / clamp / A-tile writes, 128 no fragment reads, 256 no per-tile zeroing / window offsets (timing only)
};

__device__ __forceinline__ float relu6(float v) { return fminf(fmaxf(v, 0.f), 6.f); }
__device__ __forceinline__ f4 bn_relu6(f4 a, f4 s, f4 b)
{
    const f4 v = __builtin_elementwise_fma(a, s, b);          // two v_pk_fma_f32: the same single-rounding fma per component
    return f4{ relu6(v.x), relu6(v.y), relu6(v.z), relu6(v.w) };
}

// S = depthwise stride (1, 2), CIN = input channels (64, 128, 256). DBG: the lab instantiation with the ablation switches.
template <int S, int CIN, bool DBG>
__global__ __launch_bounds__(512) void dwpw3_f32(DwPw3Args a)
{
    constexpr int XC = S + 3;                          // input columns feeding 2 adjacent output pixels
    constexpr int NK = CIN / KS;                       // steps per tile
    constexpr int LDB = CIN + 4;                       // padded filter row (floats)
    static_assert(NK >= 4 && (CIN % 64) == 0, "Cin in multiples of 64");
    __shared__ __attribute__((aligned(16))) float lds[BN3 * LDB + 8 * WT * KS + 11 * CIN + 2 * BN3];
    float *const wp_s = lds, *const a_s = wp_s + BN3 * LDB, *const wd_s = a_s + 8 * WT * KS, *const sb_s = wd_s + 9 * CIN;
    float *const sc3_s = sb_s + 2 * CIN, *const sh3_s = sc3_s + BN3;
    const int dbg = DBG ? a.dbg : 0;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned mtot = (unsigned)a.m;

    // ---- which slice and which tiles. Workgroup b sits on XCD b & 7 (dispatch order; used for locality only). The XCD's
    // workgroups split into nh slice groups; the tile range of the XCD is walked by the JM workgroups of a slice group,
    // 8 adjacent tiles per workgroup and round. The remainder round spreads its tiles one per CU first (wave-major).
    const int xcd = (int)blockIdx.x & 7, j = (int)blockIdx.x >> 3, g8 = (int)gridDim.x >> 3;
    const int slice = j % a.nh, jm = j / a.nh, JM = g8 / a.nh;
    const int n0 = slice * BN3;
    const int r0 = (int)(((long)a.tiles * xcd) >> 3), r1 = (int)(((long)a.tiles * (xcd + 1)) >> 3);
    const int per_round = JM * 8;
    const int full = (r1 - r0) / per_round, rem = (r1 - r0) - full * per_round;
    const int slot = jm * 8 + wave_u, eslot = wave_u * JM + jm;
    const int ntile = jm < JM ? full + (eslot < rem ? 1 : 0) : 0;          // (workgroups beyond nh * JM per XCD: none launched)
    auto tile_at = [&](int i) __attribute__((always_inline)) { return i < full ? r0 + i * per_round + slot : r0 + full * per_round + eslot; };

    // ---- resident images: filter slice (row 32 t + l <- channel n0 + 4 l + t), depthwise taps, scale / shift
    // (all loads of a thread in flight before its first LDS write: one memory round trip instead of one per piece)
    {
        constexpr int NP = BN3 * (CIN / 4) / 512;          // 16-byte pieces per thread: 4 / 8 / 16
        f4 pc[NP];
#pragma unroll
        for (int i = 0; i < NP; i++) {
            const int p = tid + i * 512, r = p / (CIN / 4), u = p % (CIN / 4);
            const int ch = n0 + 4 * (r & 31) + (r >> 5);
            pc[i] = *reinterpret_cast<const f4 *>(a.wp + (size_t)ch * CIN + 4 * u);
        }
        f4 tp[3];
        const int ti = tid * 4;
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const int idx = ti + i * 2048;                  // 9 * CIN floats of taps: <= 2304 -> 2 rounds of 2048 (+ a third for Cin 256)
            tp[i] = idx < 9 * CIN ? *reinterpret_cast<const f4 *>(a.wd + idx) : f4{ 0.f, 0.f, 0.f, 0.f };
        }
        f4 sb[2] = { f4{ 0.f, 0.f, 0.f, 0.f }, f4{ 0.f, 0.f, 0.f, 0.f } };
        if (ti < CIN) {
            sb[0] = *reinterpret_cast<const f4 *>(a.s2 + ti);
            sb[1] = *reinterpret_cast<const f4 *>(a.b2 + ti);
        }
#pragma unroll
        for (int i = 0; i < NP; i++) {
            const int p = tid + i * 512, r = p / (CIN / 4), u = p % (CIN / 4);
            *reinterpret_cast<f4 *>(wp_s + r * LDB + 4 * u) = pc[i];
        }
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const int idx = ti + i * 2048;
            if (idx < 9 * CIN) *reinterpret_cast<f4 *>(wd_s + idx) = tp[i];
        }
        if (ti < CIN) {
            *reinterpret_cast<f4 *>(sb_s + ti) = sb[0];
            *reinterpret_cast<f4 *>(sb_s + CIN + ti) = sb[1];
        }
    }
    if (tid < BN3) { sc3_s[tid] = a.s3[n0 + tid]; sh3_s[tid] = a.b3[n0 + tid]; }
    __syncthreads();
    if (ntile == 0) return;

    // ---- roles of this lane
    const int q = lane >> 2, c4 = lane & 3;                         // depthwise: pixels 2q, 2q+1 of the tile, channels 4*c4..+3 of the step
    const int li = lane & 31, lh = lane >> 5;                       // MFMA: row / column li, k half lh
    const __amdgpu_buffer_rsrc_t irsrc = mbn_make_rsrc(a.in, a.in_bytes);
    const __amdgpu_buffer_rsrc_t orsrc = mbn_make_rsrc(a.out, (unsigned)(a.m * a.cout * 4));
    float *const a_w = a_s + wave_u * (WT * KS);
    const int aw0 = q * KS + ((c4 ^ ((q >> 2) & 3)) << 2);          // tile row q      <- pixel 2q
    const int aw1 = aw0 + 16 * KS;                                  // tile row 16 + q <- pixel 2q + 1 (same XOR: (16 + q) >> 2 & 3 == q >> 2 & 3)
    const int fra0 = li * KS + (((0 + lh) ^ ((li >> 2) & 3)) << 2);  // A fragments: row li, 16-byte unit 2 g + lh
    const int fra1 = li * KS + (((2 + lh) ^ ((li >> 2) & 3)) << 2);
    const float *const bl01 = wp_s + li * LDB + lh * 4;             // B fragments of blocks 0, 1: + t*32*LDB + k*16 + g*8 (immediates)
    const float *const bl23 = bl01 + 64 * LDB;                      // ... of blocks 2, 3 (second base: the immediate is 16 bits)
    const float *const wk = wd_s + c4 * 4;                          // taps of this lane's channels: + k*16 + tap*CIN
    const float *const sk = sb_s + c4 * 4;

    unsigned off[3][XC];
    // Window offsets of the tile at m0 (the full-rate form of dwpw2's set_offsets_fast; the launcher admits only inputs in its range):
    // tile's first pixel on the scalar unit, the lane's pixel by two float-reciprocal divisions of small numbers (exact), one 32-bit
    // multiply; validity separable by row and column: an invalid row is 0x80000000, an invalid column 0x70000000, so any sum with an
    // invalid term lies beyond the descriptor's num_records without wrapping and the buffer unit returns zeros (= the zero padding).
    auto set_offsets = [&](unsigned m0) __attribute__((always_inline)) {
        const unsigned q0 = a.wo_m ? __umulhi(m0, a.wo_m) >> a.wo_s : m0;
        const unsigned x0 = m0 - q0 * (unsigned)a.wo;
        const unsigned nn0 = a.ho_m ? __umulhi(q0, a.ho_m) >> a.ho_s : q0;
        const unsigned y0 = q0 - nn0 * (unsigned)a.ho;
        const unsigned r = x0 + 2u * (unsigned)q;
        const unsigned q1 = (unsigned)__builtin_fmaf((float)r, a.inv_wo, 0.5f * a.inv_wo);
        const unsigned x = r - q1 * (unsigned)a.wo;
        const unsigned yy = y0 + q1;
        const unsigned q2 = (unsigned)__builtin_fmaf((float)yy, a.inv_ho, 0.5f * a.inv_ho);
        const unsigned y = yy - q2 * (unsigned)a.ho;
        const unsigned n = nn0 + q2;
        const bool mok = m0 + 2u * (unsigned)q < mtot;
        const int iy0 = (int)y * S - a.pad_top, ix0 = (int)x * S - a.pad_left;
        const unsigned cs = (unsigned)CIN * 4u, rs = (unsigned)a.w * cs;
        const int pix = __mul24((int)(n * (unsigned)a.h) + iy0, a.w) + ix0;
        const unsigned base = (unsigned)pix * cs + (unsigned)(c4 * 16);
        unsigned rowv[3], colv[XC];
#pragma unroll
        for (int dy = 0; dy < 3; dy++) rowv[dy] = (mok && (unsigned)(iy0 + dy) < (unsigned)a.h) ? base + dy * rs : 0x80000000u;
#pragma unroll
        for (int jj = 0; jj < XC; jj++) colv[jj] = ((unsigned)(ix0 + jj) < (unsigned)a.w) ? jj * cs : 0x70000000u;
#pragma unroll
        for (int dy = 0; dy < 3; dy++)
#pragma unroll
            for (int jj = 0; jj < XC; jj++) off[dy][jj] = rowv[dy] + colv[jj];
    };

    f4 xr[3][XC];
    auto ldx_row = [&](const int k, const int dy) __attribute__((always_inline)) {
#pragma unroll
        for (int jj = 0; jj < XC; jj++)
            xr[dy][jj] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(irsrc, off[dy][jj] + (unsigned)(k * KS * 4), 0, 0));
    };
    f4 wrow[3], wss[2];
    auto ldw_row = [&](const int k, const int dy) __attribute__((always_inline)) {
        if (dbg & 32) return;
#pragma unroll
        for (int dx = 0; dx < 3; dx++) wrow[dx] = *reinterpret_cast<const f4 *>(wk + k * KS + (dy * 3 + dx) * CIN);
    };
    auto ldw_ss = [&](const int k) __attribute__((always_inline)) {
        if (dbg & 32) return;
        wss[0] = *reinterpret_cast<const f4 *>(sk + k * KS);
        wss[1] = *reinterpret_cast<const f4 *>(sk + CIN + k * KS);
    };
    f4 dacc0 = f4{ 0.f, 0.f, 0.f, 0.f }, dacc1 = dacc0;
    // filter row dy of step k into the two running sums (dy = 0 starts them), then the taps the next piece needs
    auto dw_row = [&](const int k, const int dy) __attribute__((always_inline)) {
        if (dy == 0) {
            dacc0 = f4{ 0.f, 0.f, 0.f, 0.f };
            dacc1 = dacc0;
        }
        if (!(dbg & 2)) {
#pragma unroll
            for (int dx = 0; dx < 3; dx++) {
                dacc0 = __builtin_elementwise_fma(xr[dy][dx], wrow[dx], dacc0);
                dacc1 = __builtin_elementwise_fma(xr[dy][dx + S], wrow[dx], dacc1);
            }
        }
        if (dy < 2) ldw_row(k, dy + 1);
        else ldw_ss(k);
    };
    auto dw_fin = [&]() __attribute__((always_inline)) {
        if (dbg & 64) return;
        *reinterpret_cast<f4 *>(a_w + aw0) = bn_relu6(dacc0, wss[0], wss[1]);
        *reinterpret_cast<f4 *>(a_w + aw1) = bn_relu6(dacc1, wss[0], wss[1]);
    };

    f16v acc[4];
    f4 fa[2], fb[2][4];                                // A fragments [g] (g = 0 of the NEXT step is read behind dw_fin, g = 1 inside the step), B fragments [g][block]
    auto ldfrag_a = [&](const int g) __attribute__((always_inline)) {
        if (dbg & 128) return;
        fa[g] = *reinterpret_cast<const f4 *>(a_w + (g ? fra1 : fra0));
    };
    auto ldfrag_b = [&](const int k, const int g) __attribute__((always_inline)) {
        if (dbg & 128) return;
        fb[g][0] = *reinterpret_cast<const f4 *>(bl01 + k * KS + g * 8);
        fb[g][1] = *reinterpret_cast<const f4 *>(bl01 + 32 * LDB + k * KS + g * 8);
        fb[g][2] = *reinterpret_cast<const f4 *>(bl23 + k * KS + g * 8);
        fb[g][3] = *reinterpret_cast<const f4 *>(bl23 + 32 * LDB + k * KS + g * 8);
    };
    // first = the tile's first k pair: the accumulators start from the inline constant 0 (no zeroing between tiles)
    auto mfma8 = [&](const int g, const int s0, const bool first) __attribute__((always_inline)) {
        if (dbg & 16) return;
        const f16v zero = { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f };
#pragma unroll
        for (int s = s0; s < s0 + 2; s++)
#pragma unroll
            for (int t = 0; t < 4; t++)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[g][s], fb[g][t][s], (first && s == s0) ? zero : acc[t], 0, 0, 0);
    };
    // Accumulator register r of block t: channel n0 + 4 li + t, tile row i = 8 (r >> 2) + 4 lh + (r & 3) = pixel m0 + 2 (i & 15) + (i >> 4).
    // BN of rows r, r + 1 of a channel: one v_pk_fma_f32 (adjacent registers); the clamps write each value where its 16-byte store wants it.
    auto epilogue_mode = [&](unsigned m0, const bool inside) __attribute__((always_inline)) {
        const f4 sc = *reinterpret_cast<const f4 *>(sc3_s + 4 * li), sh = *reinterpret_cast<const f4 *>(sh3_s + 4 * li);
        const unsigned lane_off = ((unsigned)(8 * lh) * (unsigned)a.cout + (unsigned)(n0 + 4 * li)) * 4u;
        const unsigned rowb = (unsigned)a.cout * 4u;
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            f2 p[4];
#pragma unroll
            for (int t = 0; t < 4; t++) p[t] = __builtin_elementwise_fma(f2{ acc[t][r], acc[t][r + 1] }, f2{ sc[t], sc[t] }, f2{ sh[t], sh[t] });
            f4 o[2];
#pragma unroll
            for (int hh = 0; hh < 2; hh++)
                o[hh] = f4{ relu6(hh ? p[0].y : p[0].x), relu6(hh ? p[1].y : p[1].x), relu6(hh ? p[2].y : p[2].x), relu6(hh ? p[3].y : p[3].x) };
            // The two stores back to back, then two wait states before any VALU instruction may write their data registers. gfx950, measured here
            // (tools/dwpw3_debug.py, profiles/r06/a_*): a buffer_store_dwordx4 with an SGPR soffset followed directly by a VALU write of its first data
            // register stores the NEW value in lanes 12-15 of every 16 — the ">64-bit store data" hazard, which the compiler pads only in the
            // immediate-soffset form. Pinned with sched_barriers so nothing is scheduled in between.
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int hh = 0; hh < 2; hh++) {
                const int rr = r + hh;
                const unsigned pix = m0 + 16 * ((rr >> 2) & 1) + 2 * (rr & 3) + (rr >> 3);         // + 8 lh per lane
                const unsigned soff = pix * rowb;
                // rows past m (the last tile): the whole offset through the VGPR, so the descriptor's range check drops them
                if (inside) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, o[hh]), orsrc, lane_off, soff, 0);
                else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, o[hh]), orsrc, lane_off + soff, 0, 0);
            }
            asm volatile("s_nop 1" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto epilogue = [&](unsigned m0) __attribute__((always_inline)) {
        if (dbg & 4) return;
        if (m0 + WT <= mtot) epilogue_mode(m0, true);
        else epilogue_mode(m0, false);
    };

    // ---- prologue: step 0 of the first tile computed in full, the window of step 1 in flight
    int it = 0;
    unsigned m0M = (unsigned)tile_at(0) * WT;
    set_offsets(m0M);
#pragma unroll
    for (int dy = 0; dy < 3; dy++) ldx_row(0, dy);
    ldw_row(0, 0);
#pragma unroll
    for (int dy = 0; dy < 3; dy++) {
        dw_row(0, dy);
        ldx_row(1, dy);
    }
    dw_fin();
    ldfrag_a(0);
    ldfrag_b(0, 0);
    ldw_row(1, 0);
    bool pendE = false;
    unsigned m0E = 0;

    // ---- main loop: one iteration = one tile of this wave = NK steps, fully unrolled. In step k: the MFMAs of step k (A fragments of parity
    // k & 1, read at the end of step k - 1), the depthwise part of step k + 1 (of the next tile's step 0 in the last step) cut by filter row
    // in front of the MFMA groups, and the window loads of step k + 2 row by row behind the FMAs that consumed the row.
    for (;;) {
        const bool have_next = it + 1 < ntile;
        const unsigned m0N = have_next ? (unsigned)tile_at(it + 1) * WT : m0M;
        if (pendE) {                                   // the previous tile's 16 stores, ahead of this tile's first MFMA
            epilogue(m0E);
            pendE = false;
        }
#pragma unroll
        for (int k = 0; k < NK; k++) {
            const int kd = (k + 1) % NK, kl = (k + 2) % NK;
            if (k == NK - 2 && have_next && !(dbg & 256)) set_offsets(m0N);                // the L cursor enters the next tile (no next tile: stale offsets, unused results)
            __builtin_amdgcn_sched_barrier(0);
            // group 0
            ldfrag_a(1);
            ldfrag_b(k, 1);
            dw_row(kd, 0);
            if (!(dbg & 1)) ldx_row(kl, 0);
            __builtin_amdgcn_sched_barrier(0);
            mfma8(0, 0, k == 0);
            __builtin_amdgcn_sched_barrier(0);
            // group 1
            dw_row(kd, 1);
            if (!(dbg & 1)) ldx_row(kl, 1);
            __builtin_amdgcn_sched_barrier(0);
            mfma8(0, 2, false);
            __builtin_amdgcn_sched_barrier(0);
            // group 2
            ldfrag_b(kd, 0);                           // B fragments g = 0 of the next step (fb[0] has been consumed)
            dw_row(kd, 2);
            if (!(dbg & 1)) ldx_row(kl, 2);
            __builtin_amdgcn_sched_barrier(0);
            mfma8(1, 0, false);
            __builtin_amdgcn_sched_barrier(0);
            // group 3
            dw_fin();
            ldfrag_a(0);
            ldw_row(kl, 0);
            __builtin_amdgcn_sched_barrier(0);
            mfma8(1, 2, false);
            __builtin_amdgcn_sched_barrier(0);
        }
        pendE = true;
        m0E = m0M;
        if (!have_next) break;
        m0M = m0N;
        it++;
    }
    epilogue(m0E);
}

template <int S, int CIN>
void launch3(const DwPw3Args &a, hipStream_t s, int grid)
{
#ifdef MBN_LAB
    if (a.dbg) { hipLaunchKernelGGL((dwpw3_f32<S, CIN, true>), dim3((unsigned)grid), dim3(512), 0, s, a); return; }
#endif
    hipLaunchKernelGGL((dwpw3_f32<S, CIN, false>), dim3((unsigned)grid), dim3(512), 0, s, a);
}

}   // namespace

// 1 when the wave-private form takes this block (the caller has passed mbn_f32_dwpw_check): Cin 64 / 128 / 256 (filter slice resident in LDS,
// k loop unrolled), the full-rate window offsets' range (every input byte offset + a left-pad column below the invalid-column constant; n h + iy0
// and the row / column quotients in exact float / mul24 range), at least one workgroup per XCD and slice.
int mbn_f32_dwpw3_eligible(const mbn_context *ctx, int batch, int in_rows, int in_cols, int out_rows, int out_cols, int cin, int cout, int stride,
                           int pad_top, int pad_left)
{
    if (cin != 64 && cin != 128 && cin != 256) return 0;
    if ((cout % BN3) != 0 || cout > 1024 || (out_cols & 1) || (stride != 1 && stride != 2)) return 0;
    if (4.0 * batch * in_rows * in_cols * cin + 4.0 * (pad_left + 1) * cin > (double)0x70000000u) return 0;
    if ((double)batch * in_rows >= 8388000.0 || in_cols >= 32768 || out_cols >= 32768 || out_rows >= 32768 || pad_left > 1 || pad_top > 1) return 0;
    const int nh = cout / BN3;
    if (ctx->num_cus / (8 * nh) < 1) return 0;
    return 1;
}

int mbn_launch_f32_dwpw3(mbn_context *ctx, hipStream_t stream, float *out, const float *in, const float *wd,
                         const float *s2, const float *b2, const float *wp, const float *s3, const float *b3, int batch,
                         int in_rows, int in_cols, int out_rows, int out_cols, int cin, int cout, int stride, int pad_top,
                         int pad_left)
{
    if (!mbn_f32_dwpw3_eligible(ctx, batch, in_rows, in_cols, out_rows, out_cols, cin, cout, stride, pad_top, pad_left)) return MBN_EUNSUPPORTED;
    DwPw3Args a;
    a.out = out; a.in = in; a.wd = wd; a.s2 = s2; a.b2 = b2; a.wp = wp; a.s3 = s3; a.b3 = b3;
    a.m = (long)batch * out_rows * out_cols;
    a.h = in_rows; a.w = in_cols; a.ho = out_rows; a.wo = out_cols;
    a.cout = cout; a.pad_top = pad_top; a.pad_left = pad_left;
    a.nh = cout / BN3;
    a.tiles = (int)((a.m + WT - 1) / WT);
    mbn_udiv_magic((unsigned)out_cols, &a.wo_m, &a.wo_s);
    mbn_udiv_magic((unsigned)out_rows, &a.ho_m, &a.ho_s);
    a.in_bytes = (unsigned)(4.0 * batch * in_rows * in_cols * cin);
    a.inv_wo = 1.0f / (float)out_cols;
    a.inv_ho = 1.0f / (float)out_rows;
    const int variant = g_mbn_tune.dwpw_variant;
    a.dbg = variant >= 300 ? variant - 300 : 0;
    // whole slice groups per XCD; no more workgroups than 8-tile rounds exist (small problems)
    int per_xcd = ctx->num_cus / 8;
    per_xcd -= per_xcd % a.nh;
    const long rounds = ((a.tiles + 7) / 8 + 7) / 8;                          // 8-tile groups per XCD, rounded up
    if ((long)per_xcd > rounds * a.nh) per_xcd = (int)(rounds * a.nh);
    const int grid = per_xcd * 8;
    if (stride == 1) {
        if (cin == 64) launch3<1, 64>(a, stream, grid);
        else if (cin == 128) launch3<1, 128>(a, stream, grid);
        else launch3<1, 256>(a, stream, grid);
    } else {
        if (cin == 64) launch3<2, 64>(a, stream, grid);
        else if (cin == 128) launch3<2, 128>(a, stream, grid);
        else launch3<2, 256>(a, stream, grid);
    }
    return MBN_OK;
}
