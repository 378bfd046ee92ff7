// mbn_f32_dwpw3.hip — fused depthwise 3x3 -> pointwise 1x1 block for gfx950, fp32, WAVE-PRIVATE form (round 6).
// Same contract as mbn_f32_dwpw2.hip (one launch replaces a `depthwise` + `pointwise` pair of the reference's sequence,
// kernel.cl:62-92 + 94-114, pairs L4-5 ... of MobileNet.c:322-2599; bit-identical to the two separate launches):
//
//   out[m][n] = relu6( s3[n] * sum_c relu6( s2[c] * sum_{dy,dx} in[pix(m)+(dy,dx)][c] * wd[dy][dx][c] + b2[c] ) * wp[n][c] + b3[n] )
//
// Why a third form. The unified-wave kernel (dwpw2) hands the depthwise output of a 32-channel chunk from the 8 waves that
// produced it to the 8 waves that multiply it through a workgroup barrier, one barrier per 32 MFMAs per wave with ONE workgroup
// per CU, and streams the pointwise filter through LDS once per tile (as many bytes as the activations it multiplies). Here:
//   * a wave owns 32 output pixels x 128 output channels (64 accumulators): it computes the depthwise output of ITS pixels
//     (lane = 2 adjacent pixels x 4 channels, as in dwpw2), writes them into its private A tile in LDS and reads them back as
//     MFMA fragments — LDS operations of one wave execute in order, so the hand-over needs no barrier;
//   * the pointwise filter slice [128 output channels][Cin] stays RESIDENT in LDS for the whole launch (<= 130 KB: Cin <= 256),
//     loaded once per workgroup. A block with more than 128 output channels runs as Cout/128 slices in different workgroups
//     (the depthwise part is recomputed per slice: +9/128 of the slice's FMAs);
//   * the 8 waves of a workgroup (2 per SIMD, <= 256 VGPRs) are independent persistent pipelines over their own tile sequences.
// What the first build of this form measured (profiles/r06/b_*, d_*): the barrier was NOT what the block kernels lose their time to —
// their x-window loads are: every input byte is requested six times (3 rows x 2 column overlaps), and what the L1 does not absorb goes
// to the L2 (TCP_TCC_READ_REQ: dwpw2 2.0x the unique bytes + the filter, the strip-tiled first build of this kernel 5.2x with no L1 reuse
// between its free-running waves). So the tile and lane maps are built for the L1:
//   * a tile is a 2-D PATCH, not a strip: the 16 pixel pairs of a tile are consecutive in a band-major zigzag order (bands of R = 4 | 2 | 1
//     output rows; inside a band column pair by column pair, the band's R rows innermost), i.e. R rows x 16/R column pairs. The three
//     window rows of vertically adjacent pixels are then requested by the same wave within one step (L1 hits), and a patch touches
//     (R + 2) x (32/R + 2) input pixels instead of 3 x 34;
//   * KS = 32 (Cin <= 128): a pixel's 32 channels of a chunk = one 128-byte line per load instruction (8 lanes x 16 B), the tile's 16 pairs
//     in two half-rounds of 8; KS = 16 (Cin = 256, where LDS has no room for the 32-channel A tiles): 4 lanes x 16 B = half a line.
// A substep = 32 MFMAs (k groups 2 s, 2 s + 1 of the tile) + one depthwise half-round (KS / 16 substeps ahead of the MFMAs that consume it)
// + the window loads of the half-round after that, cut by filter row in front of the four MFMA groups. The substeps of a tile are fully
// unrolled (Cin is a template parameter): every LDS address is a lane constant + an immediate, every x load a lane offset + an immediate.
// The zigzag map lives in set_offsets alone: it leaves the pair's output pixel offset in LDS, from where the epilogue reads the 8 it needs.
// Arithmetic order = mbn_f32_dw.hip (dy-major fma chain, BN, clamp) and mbn_f32_pw.hip (k pairs (8g+s, 8g+4+s) in increasing g, s on
// v_mfma_f32_32x32x2_f32): bit-identical to the two launches and to dwpw2.
// LDS images: filter rows padded to Cin + 4 floats; A tile row = 16 * (pixel & 1) + pair, 16-byte units XORed by (row >> 2) & 3 (KS 16) or
// (row >> 1) & 7 (KS 32): the ds_write_b128 of the depthwise lanes and the fragment ds_read_b128 are conflict-free (MI355X_MICROARCH.md
// §LDS lane groups). Epilogue: LDS filter row 32 t + l holds output channel 4 l + t, so lane l's four accumulator blocks are 4 adjacent
// channels: one buffer_store_dwordx4 per lane and row pair = 512 contiguous bytes per pixel.
// gfx950 hazard found here (profiles/r06/a_*): a buffer_store_dwordx4 followed directly by a VALU write of its first data register stores
// the NEW value in lanes 12-15 of every 16 (the ">64-bit store data" hazard; the compiler pads only the immediate-soffset form): the
// epilogue issues its stores in pairs and waits two states behind each pair, pinned by sched_barriers.
#include "mbn_internal.h"
#include "mbn_epilogue.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef mbn_f16v f16v;

constexpr int BN3 = 128;                       // output channels per workgroup slice
constexpr int WT = 32;                         // output pixels per wave tile (16 pixel pairs)
constexpr unsigned PO_INVALID = 0x80000000u;   // output offset of a pixel pair past the end: beyond any descriptor, the store is dropped

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
    // zigzag order of the pixel pairs: bands of R = 1 << rsh output rows; pb = R * wo / 2 pairs per band, bi = ho / R bands per image
    int rsh;
    unsigned pb, bi;
    unsigned pb_m, pb_s, bi_m, bi_s;   // floor(v / d) = umulhi(v, d_m) >> d_s for v < 2^31 (m == 0: the divisor is 1)
    float inv_pb, inv_bi;
    int dbg;                // lab ablations (dwpw_variant = 300 + bits): 1 no x loads after the prologue, 2 no depthwise math, 4 no stores, 16 no MFMA,
                            // 32 no tap reads, 64 no BN / clamp / A-tile writes, 128 no fragment reads, 256 no per-tile window offsets (timing only)
};

__device__ __forceinline__ float relu6(float v) { return fminf(fmaxf(v, 0.f), 6.f); }
__device__ __forceinline__ f4 bn_relu6(f4 a, f4 s, f4 b)
{
    const f4 v = __builtin_elementwise_fma(a, s, b);          // two v_pk_fma_f32: the same single-rounding fma per component
    return f4{ relu6(v.x), relu6(v.y), relu6(v.z), relu6(v.w) };
}

// S = depthwise stride (1, 2), CIN = input channels (64, 128, 256), KS = channels per depthwise half-round and lane group (16: 4 lanes per
// pixel pair, all 16 pairs at once; 32: 8 lanes per pair, 8 pairs per half-round). ABL: ablation mask (lab; see DwPw3Args::dbg).
template <int S, int CIN, int KS, int ABL, int SCHED = 0, bool B64 = false>
__global__ __launch_bounds__(512) void dwpw3_f32(DwPw3Args a)
{
    constexpr int XC = S + 3;                          // input columns feeding 2 adjacent output pixels
    constexpr int NS = CIN / 16;                       // substeps per tile (32 MFMAs each)
    constexpr int AH = KS / 16;                        // half-rounds per chunk = substeps between a half-round and the MFMAs of its chunk
    constexpr int LPP = KS / 4;                        // lanes per pixel pair
    // B64 (lab A/B): B fragments as 8-byte reads, one MFMA group (2 k pairs x 4 blocks) at a time: 16 live registers instead of 32; rows padded to Cin + 2
    // floats (the 32 lanes of a ds_read_b64 group then fall on 32 different bank pairs)
    constexpr int LDB = B64 ? CIN + 2 : CIN + 4;       // padded filter row (floats)
    constexpr int ABUF = WT * KS;                      // floats per A buffer; AH buffers per wave (KS 32: by chunk parity)
    static_assert(NS >= 4 && (CIN % 64) == 0 && (KS == 16 || KS == 32), "Cin in multiples of 64");
    __shared__ __attribute__((aligned(16))) float lds[BN3 * LDB + 8 * AH * ABUF + 11 * CIN + 2 * BN3 + 8 * 32];
    float *const wp_s = lds, *const a_s = wp_s + BN3 * LDB, *const wd_s = a_s + 8 * AH * ABUF, *const sb_s = wd_s + 9 * CIN;
    float *const sc3_s = sb_s + 2 * CIN, *const sh3_s = sc3_s + BN3;
    unsigned *const po_s = reinterpret_cast<unsigned *>(sh3_s + BN3);      // [wave][tile parity][16 pairs]: output byte offset of each pair's first pixel
    const int dbg = ABL < 0 ? a.dbg : ABL;         // ABL >= 0: the ablation mask is a compile-time constant (0 = the shipped kernel); -1: runtime (a.dbg)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned ptot = (unsigned)(a.m >> 1);        // pixel pairs

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
    const int ntile = full + (eslot < rem ? 1 : 0);
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
            if constexpr (B64) {
                *reinterpret_cast<f2 *>(wp_s + r * LDB + 4 * u) = f2{ pc[i].x, pc[i].y };
                *reinterpret_cast<f2 *>(wp_s + r * LDB + 4 * u + 2) = f2{ pc[i].z, pc[i].w };
            } else *reinterpret_cast<f4 *>(wp_s + r * LDB + 4 * u) = pc[i];
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
    const int qh = lane / LPP, cl = lane % LPP;                     // depthwise: pair qh (+ 8 in the second half-round of KS 32), channels 4*cl..+3 of the chunk
    const int li = lane & 31, lh = lane >> 5;                       // MFMA: row / column li, k half lh
    const __amdgpu_buffer_rsrc_t irsrc = mbn_make_rsrc(a.in, a.in_bytes);
    const __amdgpu_buffer_rsrc_t orsrc = mbn_make_rsrc(a.out, (unsigned)(a.m * a.cout * 4));
    float *const a_w = a_s + wave_u * (AH * ABUF);
    unsigned *const po_w = po_s + wave_u * 32;
    // A-tile slot of (half-round hh, pixel e of the pair): tile row 16 e + pair, 16-byte unit cl XORed by the row's swizzle term
    auto aw_at = [&](const int hh) __attribute__((always_inline)) {
        const int row = 8 * hh * (AH - 1) + qh;                     // KS 32: pair = 8 hh + qh; KS 16: pair = qh
        return KS == 16 ? row * KS + ((cl ^ ((row >> 2) & 3)) << 2) : row * KS + ((cl ^ ((row >> 1) & 7)) << 2);
    };
    const int aw[2] = { aw_at(0), aw_at(1) };                       // (+ 16 * KS floats for the pair's second pixel: the XOR term is the same)
    // A fragment of 16-byte unit u: row li
    auto fra_at = [&](const int u) __attribute__((always_inline)) {
        return KS == 16 ? li * KS + (((u + lh) ^ ((li >> 2) & 3)) << 2) : li * KS + (((u + lh) ^ ((li >> 1) & 7)) << 2);
    };
    const int fra[4] = { fra_at(0), fra_at(2), fra_at(KS == 32 ? 4 : 0), fra_at(KS == 32 ? 6 : 2) };
    const float *const bl01 = wp_s + li * LDB + lh * 4;             // B fragments of blocks 0, 1: + t*32*LDB + s*16 + g*8 (immediates)
    const float *const bl23 = bl01 + 64 * LDB;                      // ... of blocks 2, 3 (second base: the immediate is 16 bits)
    const float *const wk = wd_s + cl * 4;                          // taps of this lane's channels: + chunk*KS + tap*CIN

    // SEP (stride 2 with two half-rounds: 2 x 15 offsets do not fit beside the 15-register-wider window): the offsets stay in their separable
    // form rowv + colv (16 registers) and every load adds its pair (one full-rate v_add_u32 per load)
    constexpr bool SEP = S == 2 && AH == 2;
    unsigned off[SEP ? 1 : AH][SEP ? 1 : 3][SEP ? 1 : XC];
    unsigned rowv_k[SEP ? AH : 1][3], colv_k[SEP ? AH : 1][XC];
    // Window offsets of half-round hh of the tile whose first pair is p0 (wave-uniform), and the pair's output offset into po_w[slot].
    // Tile-uniform part on the scalar unit (magic division); the lane's pair by float-reciprocal divisions of small numbers (exact: the
    // quotients' numerators stay below 2^21, see the launcher's range checks); validity separable by row and column: an invalid row is
    // 0x80000000, an invalid column 0x70000000, so any sum with an invalid term lies beyond the descriptor's num_records without wrapping and
    // the buffer unit returns zeros (= the zero padding).
    auto set_offsets = [&](unsigned p0, const int hh, int pslot) __attribute__((always_inline)) {
        const unsigned bg0 = a.pb_m ? __umulhi(p0, a.pb_m) >> a.pb_s : p0;          // band (over the whole batch) of the tile's first pair
        const unsigned i0 = p0 - bg0 * a.pb;
        const unsigned im0 = a.bi_m ? __umulhi(bg0, a.bi_m) >> a.bi_s : bg0;        // its image
        const unsigned b0 = bg0 - im0 * a.bi;
        const unsigned qq = (unsigned)(8 * hh * (AH - 1) + qh);
        const unsigned i1 = i0 + qq;
        const unsigned dq = (unsigned)__builtin_fmaf((float)i1, a.inv_pb, 0.5f * a.inv_pb);
        const unsigned i = i1 - (unsigned)__mul24((int)dq, (int)a.pb);
        const unsigned b1 = b0 + dq;
        const unsigned dn = (unsigned)__builtin_fmaf((float)b1, a.inv_bi, 0.5f * a.inv_bi);
        const unsigned band = b1 - (unsigned)__mul24((int)dn, (int)a.bi);
        const unsigned n = im0 + dn;
        const unsigned y = (band << a.rsh) + (i & ((1u << a.rsh) - 1u)), x = (i >> a.rsh) << 1;
        const bool mok = p0 + qq < ptot;
        const int iy0 = (int)y * S - a.pad_top, ix0 = (int)x * S - a.pad_left;
        const unsigned cs = (unsigned)CIN * 4u, rs = (unsigned)a.w * cs;
        const int pix = __mul24((int)(n * (unsigned)a.h) + iy0, a.w) + ix0;
        const unsigned base = (unsigned)pix * cs + (unsigned)(cl * 16);
        unsigned rowv[3], colv[XC];
#pragma unroll
        for (int dy = 0; dy < 3; dy++) rowv[dy] = (mok && (unsigned)(iy0 + dy) < (unsigned)a.h) ? base + dy * rs : 0x80000000u;
#pragma unroll
        for (int jj = 0; jj < XC; jj++) colv[jj] = ((unsigned)(ix0 + jj) < (unsigned)a.w) ? jj * cs : 0x70000000u;
        if constexpr (SEP) {
#pragma unroll
            for (int dy = 0; dy < 3; dy++) rowv_k[hh][dy] = rowv[dy];
#pragma unroll
            for (int jj = 0; jj < XC; jj++) colv_k[hh][jj] = colv[jj];
        } else {
#pragma unroll
            for (int dy = 0; dy < 3; dy++)
#pragma unroll
                for (int jj = 0; jj < XC; jj++) off[hh][dy][jj] = rowv[dy] + colv[jj];
        }
        const unsigned opix = (unsigned)__mul24((int)(n * (unsigned)a.ho + y), a.wo) + x;
        const unsigned po = mok ? opix * ((unsigned)a.cout * 4u) : PO_INVALID;
        // the slot address from a freshly read lane id (two full-rate instructions per tile) instead of a register that lives through the whole
        // loop: at 256 VGPRs (Cin 256) that register was spilled and its reload's vmcnt(0) drained the window loads once per tile
        unsigned lid;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lid));
        if ((lid % LPP) == 0) po_w[pslot * 16 + 8 * hh * (AH - 1) + (int)(lid / LPP)] = po;
    };
    auto set_offsets_tile = [&](unsigned p0, int pslot) __attribute__((always_inline)) {
#pragma unroll
        for (int hh = 0; hh < AH; hh++) set_offsets(p0, hh, pslot);
    };

    // Substep index u (of a tile) -> the half-round's offset set, A buffer and channel offsets
    //   KS 16: half-round u = all pairs, channels 16 u;  KS 32: chunk u >> 1, half u & 1 (pairs 8 (u & 1) + qh), channels 32 (u >> 1)
    f4 xr[3][XC];
    auto ldx_row = [&](const int u, const int dy) __attribute__((always_inline)) {
        const int hh = AH == 2 ? (u & 1) : 0, cb = AH == 2 ? (u >> 1) * 128 : u * 64;
#pragma unroll
        for (int jj = 0; jj < XC; jj++)
            xr[dy][jj] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(irsrc, (SEP ? rowv_k[hh][dy] + colv_k[hh][jj] : off[SEP ? 0 : hh][SEP ? 0 : dy][SEP ? 0 : jj]) + (unsigned)cb, 0, 0));
    };
    f4 wrow[3], wss[2];
    auto ldw_row = [&](const int u, const int dy) __attribute__((always_inline)) {
        if (dbg & 32) return;
        const int cf = AH == 2 ? (u >> 1) * 32 : u * 16;
#pragma unroll
        for (int dx = 0; dx < 3; dx++) wrow[dx] = *reinterpret_cast<const f4 *>(wk + cf + (dy * 3 + dx) * CIN);
    };
    auto ldw_ss = [&](const int u) __attribute__((always_inline)) {
        if (dbg & 32) return;
        const int cf = AH == 2 ? (u >> 1) * 32 : u * 16;
        wss[0] = *reinterpret_cast<const f4 *>(wk + 9 * CIN + cf);          // scale | shift follow the nine tap planes (sb_s = wd_s + 9 CIN): one base register
        wss[1] = *reinterpret_cast<const f4 *>(wk + 10 * CIN + cf);
    };
    f4 dacc0 = f4{ 0.f, 0.f, 0.f, 0.f }, dacc1 = dacc0;
    // filter row dy of half-round u into the two running sums (dy = 0 starts them), then the taps the next piece needs
    auto dw_row = [&](const int u, const int dy) __attribute__((always_inline)) {
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
        if (dy < 2) ldw_row(u, dy + 1);
        else ldw_ss(u);
    };
    auto dw_fin = [&](const int u) __attribute__((always_inline)) {
        if (dbg & 64) return;
        const int hh = AH == 2 ? (u & 1) : 0, buf = AH == 2 ? ((u >> 1) & 1) * ABUF : 0;
        *reinterpret_cast<f4 *>(a_w + buf + aw[hh]) = bn_relu6(dacc0, wss[0], wss[1]);
        *reinterpret_cast<f4 *>(a_w + buf + aw[hh] + 16 * KS) = bn_relu6(dacc1, wss[0], wss[1]);
    };

    f16v acc[4];
    f4 fa[2], fb[B64 ? 1 : 2][B64 ? 1 : 4];            // A fragments [g] (g = 0 of the NEXT substep is read behind dw_fin, g = 1 inside the substep), B fragments [g][block]
    f2 fbh[B64 ? 2 : 1][B64 ? 4 : 1];                  // B64: B half fragments [group parity][block]
    auto ldfrag_a = [&](const int u, const int g) __attribute__((always_inline)) {
        if (dbg & 128) return;
        const int buf = AH == 2 ? ((u >> 1) & 1) * ABUF : 0, idx = AH == 2 ? 2 * (u & 1) + g : g;
        fa[g] = *reinterpret_cast<const f4 *>(a_w + buf + fra[idx]);
    };
    // B64: the half fragments of MFMA group G (g = G >> 1, k pairs 2 (G & 1), 2 (G & 1) + 1) of substep u
    auto ldfrag_bh = [&](const int u, const int G) __attribute__((always_inline)) {
        if (dbg & 128) return;
        const int o = u * 16 + (G >> 1) * 8 + (G & 1) * 2;
        fbh[B64 ? (G & 1) : 0][0] = *reinterpret_cast<const f2 *>(bl01 + o);
        fbh[B64 ? (G & 1) : 0][B64 ? 1 : 0] = *reinterpret_cast<const f2 *>(bl01 + 32 * LDB + o);
        fbh[B64 ? (G & 1) : 0][B64 ? 2 : 0] = *reinterpret_cast<const f2 *>(bl23 + o);
        fbh[B64 ? (G & 1) : 0][B64 ? 3 : 0] = *reinterpret_cast<const f2 *>(bl23 + 32 * LDB + o);
    };
    auto ldfrag_b = [&](const int u, const int g) __attribute__((always_inline)) {
        if (dbg & 128) return;
        if constexpr (B64) return;
        fb[g][0] = *reinterpret_cast<const f4 *>(bl01 + u * 16 + g * 8);
        fb[g][B64 ? 0 : 1] = *reinterpret_cast<const f4 *>(bl01 + 32 * LDB + u * 16 + g * 8);
        fb[g][B64 ? 0 : 2] = *reinterpret_cast<const f4 *>(bl23 + u * 16 + g * 8);
        fb[g][B64 ? 0 : 3] = *reinterpret_cast<const f4 *>(bl23 + 32 * LDB + u * 16 + g * 8);
    };
    // first = the tile's first k pair: the accumulators start from the inline constant 0 (no zeroing between tiles)
    auto mfma8 = [&](const int g, const int s0, const bool first) __attribute__((always_inline)) {
        if (dbg & 16) return;
        const f16v zero = { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f };
#pragma unroll
        for (int s = s0; s < s0 + 2; s++)
#pragma unroll
            for (int t = 0; t < 4; t++)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[g][s], B64 ? fbh[B64 ? ((2 * g + (s0 >> 1)) & 1) : 0][B64 ? t : 0][s & 1] : fb[B64 ? 0 : g][B64 ? 0 : t][s], (first && s == s0) ? zero : acc[t], 0, 0, 0);
    };
    // Accumulator register r of block t: channel n0 + 4 li + t, tile row 8 (r >> 2) + 4 lh + (r & 3) = pixel (r >> 3) of pair 8 ((r >> 2) & 1) + 4 lh + (r & 3).
    // The lane's 8 pairs' output offsets come from po_w (two ds_read_b128); BN of rows r, r + 1 of a channel is one v_pk_fma_f32 (adjacent
    // registers); the clamps write each value where its 16-byte store wants it. A pair past the end carries PO_INVALID: dropped by the range check.
    auto epilogue = [&](int pslot) __attribute__((always_inline)) {
        if (dbg & 4) return;
        const f4 sc = *reinterpret_cast<const f4 *>(sc3_s + 4 * li), sh = *reinterpret_cast<const f4 *>(sh3_s + 4 * li);
        const u4 pa = *reinterpret_cast<const u4 *>(po_w + pslot * 16 + 4 * lh), pb4 = *reinterpret_cast<const u4 *>(po_w + pslot * 16 + 8 + 4 * lh);
        const unsigned lane_col = (unsigned)(n0 + 4 * li) * 4u;
        unsigned vo[2][4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            vo[0][c] = pa[c] + lane_col;
            vo[1][c] = pb4[c] + lane_col;
        }
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
            // the two stores back to back, then two wait states before any VALU instruction may write their data registers (see the header)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int hh = 0; hh < 2; hh++) {
                const int rr = r + hh;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, o[hh]), orsrc, vo[(rr >> 2) & 1][rr & 3], (rr >> 3) ? rowb : 0u, 0);
            }
            asm volatile("s_nop 1" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- prologue: the first AH half-rounds of the first tile computed in full, the window of the next one in flight
    int it = 0;
    unsigned p0M = (unsigned)tile_at(0) * 16u;
    set_offsets_tile(p0M, 0);
#pragma unroll
    for (int u = 0; u < AH; u++) {
#pragma unroll
        for (int dy = 0; dy < 3; dy++) ldx_row(u, dy);
        ldw_row(u, 0);
#pragma unroll
        for (int dy = 0; dy < 3; dy++) dw_row(u, dy);
        dw_fin(u);
    }
#pragma unroll
    for (int dy = 0; dy < 3; dy++) ldx_row(AH, dy);
    ldfrag_a(0, 0);
    ldfrag_b(0, 0);
    if constexpr (B64) ldfrag_bh(0, 0);
    ldw_row(AH, 0);
    bool pendE = false;

    // ---- main loop: one iteration = one tile of this wave = NS substeps, fully unrolled. In substep u: the MFMAs of k groups 2 u, 2 u + 1; the
    // depthwise half-round u + AH (of the next tile when past the end) cut by filter row in front of the MFMA groups; the window loads of
    // half-round u + AH + 1 row by row behind the FMAs that consumed the row.
    for (;;) {
        const bool have_next = it + 1 < ntile;
        const unsigned p0N = have_next ? (unsigned)tile_at(it + 1) * 16u : p0M;
        if (SCHED != 3 && pendE) {                     // the previous tile's 16 stores, ahead of this tile's first MFMA
            epilogue((it - 1) & 1);
            pendE = false;
        }

#pragma unroll
        for (int u = 0; u < NS; u++) {
            const int ud = (u + AH) % NS, ul = (u + AH + 1) % NS;
            // the load cursor enters the next tile (no next tile: stale offsets, unused results)
            if (u == NS - AH - 1 && have_next && !(dbg & 256)) set_offsets_tile(p0N, (it + 1) & 1);
            if constexpr (SCHED == 3) {
                // SCHED 3 (lab A/B): in a tile's FIRST substep the whole depthwise part and its window loads are issued AHEAD of the previous tile's stores, the 32
                // MFMAs behind them. vmcnt retires in order: a load issued behind the 16 stores is not seen complete before they are, and in the pinned form the
                // first substep's loads (consumed one substep later) sit right behind them.
                if (u == 0) {
                    __builtin_amdgcn_sched_barrier(0);
                    ldfrag_a(u, 1);
                    ldfrag_b(u, 1);
#pragma unroll
                    for (int dy = 0; dy < 3; dy++) {
                        dw_row(ud, dy);
                        if (!(dbg & 1)) ldx_row(ul, dy);
                    }
                    dw_fin(ud);                        // (this substep's fragments are in registers: the A tile may be overwritten; the sums and scale / shift die here, ahead of the epilogue's temporaries)
                    __builtin_amdgcn_sched_barrier(0);
                    if (pendE) {
                        epilogue((it - 1) & 1);
                        pendE = false;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (B64) ldfrag_bh(u, 1);
                    mfma8(0, 0, true);
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (B64) ldfrag_bh(u, 2);
                    mfma8(0, 2, false);
                    __builtin_amdgcn_sched_barrier(0);
                    ldfrag_b((u + 1) % NS, 0);
                    if constexpr (B64) ldfrag_bh(u, 3);
                    __builtin_amdgcn_sched_barrier(0);
                    mfma8(1, 0, false);
                    __builtin_amdgcn_sched_barrier(0);
                    ldfrag_a((u + 1) % NS, 0);
                    if constexpr (B64) ldfrag_bh((u + 1) % NS, 0);
                    ldw_row(ul, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    mfma8(1, 2, false);
                    __builtin_amdgcn_sched_barrier(0);
                    continue;
                }
            }
            // SCHED (lab A/B, exp2): 0 = the four pinned groups below; 1 = one scheduling region per substep, the compiler's own order; 2 = one region with
            // the interleave requested through sched_group_barrier: after every MFMA two VALU and one memory / LDS instruction
#define SB() do { if constexpr (SCHED == 0) __builtin_amdgcn_sched_barrier(0); } while (0)
            __builtin_amdgcn_sched_barrier(0);
            // group 0
            ldfrag_a(u, 1);
            ldfrag_b(u, 1);
            if constexpr (B64) ldfrag_bh(u, 1);
            dw_row(ud, 0);
            if (!(dbg & 1)) ldx_row(ul, 0);
            SB();
            mfma8(0, 0, u == 0);
            SB();
            // group 1
            if constexpr (B64) ldfrag_bh(u, 2);
            dw_row(ud, 1);
            if (!(dbg & 1)) ldx_row(ul, 1);
            SB();
            mfma8(0, 2, false);
            SB();
            // group 2
            ldfrag_b((u + 1) % NS, 0);                 // B fragments g = 0 of the next substep (fb[0] has been consumed)
            if constexpr (B64) ldfrag_bh(u, 3);
            dw_row(ud, 2);
            if (!(dbg & 1)) ldx_row(ul, 2);
            SB();
            mfma8(1, 0, false);
            SB();
            // group 3
            dw_fin(ud);
            ldfrag_a((u + 1) % NS, 0);
            if constexpr (B64) ldfrag_bh((u + 1) % NS, 0);
            ldw_row(ul, 0);
            SB();
            mfma8(1, 2, false);
            if constexpr (SCHED == 2) {
#pragma unroll
                for (int i = 0; i < 32; i++) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
                    __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);      // two VALU
                    __builtin_amdgcn_sched_group_barrier(0x320, 1, 0);      // one DS read / DS write / VMEM read
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#undef SB
        }
        pendE = true;
        if (!have_next) break;
        p0M = p0N;
        it++;
    }
    epilogue(it & 1);
}

template <int S, int CIN, int KS>
void launch3(const DwPw3Args &a, hipStream_t s, int grid)
{
#ifdef MBN_LAB
    // compile-time ablations of the two stride-1 default shapes (no runtime branches in the substep: the timings are those of the real instruction
    // stream minus the part switched off); any other mask / shape: the runtime-switch build (its ~30 scalar branches per substep cost time of their own)
    if constexpr (S == 1 && ((CIN == 128 && KS == 32) || (CIN == 256 && KS == 16))) {
        switch (a.dbg) {
        case 1: hipLaunchKernelGGL((dwpw3_f32<S, CIN, KS, 1>), dim3((unsigned)grid), dim3(512), 0, s, a); return;
        case 2: hipLaunchKernelGGL((dwpw3_f32<S, CIN, KS, 2>), dim3((unsigned)grid), dim3(512), 0, s, a); return;
        case 4: hipLaunchKernelGGL((dwpw3_f32<S, CIN, KS, 4>), dim3((unsigned)grid), dim3(512), 0, s, a); return;
        case 7: hipLaunchKernelGGL((dwpw3_f32<S, CIN, KS, 7>), dim3((unsigned)grid), dim3(512), 0, s, a); return;
        case 16: hipLaunchKernelGGL((dwpw3_f32<S, CIN, KS, 16>), dim3((unsigned)grid), dim3(512), 0, s, a); return;
        case 23: hipLaunchKernelGGL((dwpw3_f32<S, CIN, KS, 23>), dim3((unsigned)grid), dim3(512), 0, s, a); return;
        case 39: hipLaunchKernelGGL((dwpw3_f32<S, CIN, KS, 39>), dim3((unsigned)grid), dim3(512), 0, s, a); return;
        case 103: hipLaunchKernelGGL((dwpw3_f32<S, CIN, KS, 103>), dim3((unsigned)grid), dim3(512), 0, s, a); return;
        case 231: hipLaunchKernelGGL((dwpw3_f32<S, CIN, KS, 231>), dim3((unsigned)grid), dim3(512), 0, s, a); return;
        case 487: hipLaunchKernelGGL((dwpw3_f32<S, CIN, KS, 487>), dim3((unsigned)grid), dim3(512), 0, s, a); return;
        case 503: hipLaunchKernelGGL((dwpw3_f32<S, CIN, KS, 503>), dim3((unsigned)grid), dim3(512), 0, s, a); return;
        default: break;
        }
    }
    if (a.dbg) { hipLaunchKernelGGL((dwpw3_f32<S, CIN, KS, -1>), dim3((unsigned)grid), dim3(512), 0, s, a); return; }
#endif
    hipLaunchKernelGGL((dwpw3_f32<S, CIN, KS, 0>), dim3((unsigned)grid), dim3(512), 0, s, a);
}

}   // namespace

// 1 when the wave-private form takes this block (the caller has passed mbn_f32_dwpw_check): Cin 64 / 128 / 256 (filter slice resident in LDS,
// substeps unrolled), the full-rate window offsets' range (every input byte offset + a left-pad column below the invalid-column constant; n h + iy0
// in mul24 range; the band / image quotients' numerators below 2^21: exact float-reciprocal division), at least one workgroup per XCD and slice.
int mbn_f32_dwpw3_eligible(const mbn_context *ctx, int batch, int in_rows, int in_cols, int out_rows, int out_cols, int cin, int cout, int stride,
                           int pad_top, int pad_left)
{
    if (cin != 64 && cin != 128 && cin != 256) return 0;
#ifndef MBN_LAB
    if (!MBN_DWPW3_DEFAULT(stride, cin)) return 0;      // the shipped library holds only the instantiations its dispatch rule reaches (stride 1, Cin 128 / 256)
#endif
    if ((cout % BN3) != 0 || cout > 1024 || (out_cols & 1) || (stride != 1 && stride != 2)) return 0;
    if (4.0 * batch * in_rows * in_cols * cin + 4.0 * (pad_left + 1) * cin > (double)0x70000000u) return 0;
    if ((double)batch * in_rows >= 8388000.0 || in_cols >= 32768 || out_cols >= 16384 || out_rows >= 32768 || pad_left > 1 || pad_top > 1) return 0;
    if ((double)batch * out_rows >= 2000000.0 || (double)batch * out_rows * out_cols >= 2147483000.0) return 0;
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
    // bands of 4 output rows where the map's height allows (2, then 1 otherwise); lab: exp1 = 1 | 2 | 3 forces R = 1 | 2 | 4 where it divides
    a.rsh = (out_rows % 4) == 0 ? 2 : (out_rows % 2) == 0 ? 1 : 0;
#ifdef MBN_LAB
    if (g_mbn_tune.exp1 >= 1 && g_mbn_tune.exp1 <= 3 && (out_rows % (1 << (g_mbn_tune.exp1 - 1))) == 0) a.rsh = g_mbn_tune.exp1 - 1;
#endif
    a.pb = (unsigned)((out_cols / 2) << a.rsh);
    a.bi = (unsigned)(out_rows >> a.rsh);
    mbn_udiv_magic(a.pb, &a.pb_m, &a.pb_s);
    mbn_udiv_magic(a.bi, &a.bi_m, &a.bi_s);
    a.inv_pb = 1.0f / (float)a.pb;
    a.inv_bi = 1.0f / (float)a.bi;
    a.in_bytes = (unsigned)(4.0 * batch * in_rows * in_cols * cin);
    const int variant = g_mbn_tune.dwpw_variant;
    a.dbg = variant >= 300 ? variant - 300 : 0;
    // whole slice groups per XCD; no more workgroups per slice than the XCD has tiles (small problems: the remainder round hands out its tiles
    // one per CU first, so a few tiles spread over many CUs instead of filling the 8 waves of a few)
    int per_xcd = ctx->num_cus / 8;
    per_xcd -= per_xcd % a.nh;
    const long tiles_xcd = (a.tiles + 7) / 8;
    if ((long)per_xcd > tiles_xcd * a.nh) per_xcd = (int)(tiles_xcd * a.nh);
    const int grid = per_xcd * 8;
    // 32-channel half-rounds (a full 128-byte line per pixel and load) where the A tiles fit beside the filter slice (Cin <= 128).
    // Lab: exp0 = 16 forces the 16-channel form.
    [[maybe_unused]] bool ks32 = cin <= 128;
#ifdef MBN_LAB
    if (g_mbn_tune.exp0 == 16) ks32 = false;
    if (!ks32 && cin <= 128) {
        if (stride == 1) { if (cin == 64) launch3<1, 64, 16>(a, stream, grid); else launch3<1, 128, 16>(a, stream, grid); }
        else { if (cin == 64) launch3<2, 64, 16>(a, stream, grid); else launch3<2, 128, 16>(a, stream, grid); }
        return MBN_OK;
    }
#endif
#ifdef MBN_LAB
    if (stride == 1 && g_mbn_tune.exp2 == 5 && !a.dbg && cin >= 128) {             // lab A/B: 8-byte B fragment reads + the first substep's loads ahead of the stores
        if (cin == 128) hipLaunchKernelGGL((dwpw3_f32<1, 128, 32, 0, 3, true>), dim3((unsigned)grid), dim3(512), 0, stream, a);
        else hipLaunchKernelGGL((dwpw3_f32<1, 256, 16, 0, 3, true>), dim3((unsigned)grid), dim3(512), 0, stream, a);
        return MBN_OK;
    }
    if (stride == 1 && g_mbn_tune.exp2 == 4 && !a.dbg && cin >= 128) {             // lab A/B: 8-byte B fragment reads (16 VGPRs fewer)
        if (cin == 128) hipLaunchKernelGGL((dwpw3_f32<1, 128, 32, 0, 0, true>), dim3((unsigned)grid), dim3(512), 0, stream, a);
        else hipLaunchKernelGGL((dwpw3_f32<1, 256, 16, 0, 0, true>), dim3((unsigned)grid), dim3(512), 0, stream, a);
        return MBN_OK;
    }
    if (stride == 1 && g_mbn_tune.exp2 >= 1 && g_mbn_tune.exp2 <= 3 && !a.dbg && cin >= 128) {
        const int sc = g_mbn_tune.exp2;
        if (cin == 128) {
            if (sc == 1) hipLaunchKernelGGL((dwpw3_f32<1, 128, 32, 0, 1>), dim3((unsigned)grid), dim3(512), 0, stream, a);
            else if (sc == 2) hipLaunchKernelGGL((dwpw3_f32<1, 128, 32, 0, 2>), dim3((unsigned)grid), dim3(512), 0, stream, a);
            else hipLaunchKernelGGL((dwpw3_f32<1, 128, 32, 0, 3>), dim3((unsigned)grid), dim3(512), 0, stream, a);
        } else {
            if (sc == 1) hipLaunchKernelGGL((dwpw3_f32<1, 256, 16, 0, 1>), dim3((unsigned)grid), dim3(512), 0, stream, a);
            else if (sc == 2) hipLaunchKernelGGL((dwpw3_f32<1, 256, 16, 0, 2>), dim3((unsigned)grid), dim3(512), 0, stream, a);
            else hipLaunchKernelGGL((dwpw3_f32<1, 256, 16, 0, 3>), dim3((unsigned)grid), dim3(512), 0, stream, a);
        }
        return MBN_OK;
    }
#endif
#ifdef MBN_LAB
    if (stride == 1) {
        if (cin == 64) launch3<1, 64, 32>(a, stream, grid);
        else if (cin == 128) launch3<1, 128, 32>(a, stream, grid);
        else launch3<1, 256, 16>(a, stream, grid);
    } else {
        if (cin == 64) launch3<2, 64, 32>(a, stream, grid);
        else if (cin == 128) launch3<2, 128, 32>(a, stream, grid);
        else launch3<2, 256, 16>(a, stream, grid);
    }
#else
    if (cin == 128) launch3<1, 128, 32>(a, stream, grid);
    else launch3<1, 256, 16>(a, stream, grid);
#endif
    return MBN_OK;
}
