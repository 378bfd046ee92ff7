// mbn_f32_pw3.hip — pointwise 1x1 conv (+ BN + ReLU6) for gfx950, fp32, SHORT K (Cin 64 / 128 / 256): the wave-private GEMM of
// mbn_f32_dwpw3.hip without its depthwise part (round 6). kernel.cl:94-114 `pointwise`; out[m][n] = relu6(s[n] * sum_k in[m][k] * w[n][k] + b[n]).
//
// Why. pw_gemm's 128x128 tile does 2 ... 8 k-tiles per output tile at K = 64 ... 256: its epilogue (64 stores per wave, the accumulator
// hand-over, the tile change) is exposed once per 64 ... 256 MFMAs and the stand-alone short-K layers sit at 0.55-0.72 of the fp32 MFMA
// peak against 0.81-0.86 for K >= 512. Here the filter slice [128 output channels][K] is resident in LDS for the
// whole launch (no filter traffic per tile, no barrier in the loop), a wave owns 32 pixels x 128 channels (64 accumulators), stages ITS
// 32 x 16/32 activation block through registers into its private LDS tile (the lanes that load a pixel's 64 / 128 contiguous bytes are
// not the lanes that feed it to the MFMA) one to four half-rounds ahead, and its 16 x 16-byte stores per tile are the only other
// vector-memory traffic. No VALU in the substep besides one address add per load pair.
// Same instruction (v_mfma_f32_32x32x2_f32), same k pairs (8g + s, 8g + 4 + s) in the same order, same epilogue arithmetic as pw_gemm:
// bit-identical to it (tests: test_f32_pointwise with pw_tile = 9), so the dispatch may depend on M.
// LDS images as in mbn_f32_dwpw3.hip: filter rows padded to K + 4 floats, row 32 t + l = output channel 4 l + t of the slice (a lane's four
// accumulator blocks are 4 adjacent channels: 16-byte stores, 512 contiguous bytes per pixel); A tile row = 16 (pixel & 1) + (pixel >> 1),
// 16-byte units XORed by (row >> 2) & 3 (KS 16) / (row >> 1) & 7 (KS 32). The store-data hazard of 16-byte buffer stores found there is
// padded here the same way (two stores, two wait states, pinned).
#include "mbn_internal.h"
#include "mbn_epilogue.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef mbn_f16v f16v;

constexpr int WT = 32;

struct Pw3Args {
    float *out;
    const float *in, *wp, *s3, *b3;
    long m;
    int n;                  // output channels (multiple of the slice width)
    int nh;                 // slices = n / (32 NB)
    int tiles;              // ceil(m / 32)
    unsigned in_bytes;
};

__device__ __forceinline__ float relu6(float v) { return fminf(fmaxf(v, 0.f), 6.f); }

// K = input channels; KS = channels per half-round (16: 4 lanes per pixel pair, 32: 8 lanes = a whole 128-byte line per pixel and load);
// PD = half-rounds the loads run ahead of the half-round that writes them into the A tile (up to two tiles ahead: the short-K layers are the
// HBM-bound ones, and two 1 KB loads per half-round and wave need several half-rounds in flight to cover the memory latency)
// NB = 32-channel accumulator blocks per wave (4: 128-channel slices, 16-byte stores; 2: 64-channel slices — the slice of a K = 512 filter that fits in LDS —
// 8-byte stores); NW = waves per workgroup (8 = 2 per SIMD at <= 256 VGPRs; 12 = 3 per SIMD at <= 168: the 32 accumulators of NB = 2 leave room)
template <int K, int KS, int PD, int NB = 4, int NW = 8>
__global__ __launch_bounds__(64 * NW) void pw3_f32(Pw3Args a)
{
    constexpr int NS = K / 16, AH = KS / 16, LPP = KS / 4, LDB = K + 4, ABUF = WT * KS, BN3 = 32 * NB, NT = 64 * NW;
    static_assert(NB == 4 || NB == 2, "slice width");
    static_assert(AH + PD <= 2 * NS + 1 && ((NS % PD) == 0 || (PD % NS) == 0) && AH <= PD && (K % 64) == 0, "prefetch depth: at most two tiles ahead");
    __shared__ __attribute__((aligned(16))) float lds[BN3 * LDB + NW * AH * ABUF + 2 * BN3];
    float *const wp_s = lds, *const a_s = wp_s + BN3 * LDB, *const sc3_s = a_s + NW * AH * ABUF, *const sh3_s = sc3_s + BN3;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned mtot = (unsigned)a.m;

    // slice and tiles of this wave: as in dwpw3 (XCD = blockIdx & 7 for locality; 8 adjacent tiles per workgroup and round; remainder spread)
    const int xcd = (int)blockIdx.x & 7, j = (int)blockIdx.x >> 3, g8 = (int)gridDim.x >> 3;
    const int slice = j % a.nh, jm = j / a.nh, JM = g8 / a.nh;
    const int n0 = slice * BN3;
    const int r0 = (int)(((long)a.tiles * xcd) >> 3), r1 = (int)(((long)a.tiles * (xcd + 1)) >> 3);
    const int per_round = JM * NW;
    const int full = (r1 - r0) / per_round, rem = (r1 - r0) - full * per_round;
    const int slot = jm * NW + wave_u, eslot = wave_u * JM + jm;
    const int ntile = full + (eslot < rem ? 1 : 0);
    auto tile_at = [&](int i) __attribute__((always_inline)) { return i < full ? r0 + i * per_round + slot : r0 + full * per_round + eslot; };

    {   // resident filter slice (row 32 t + l <- channel n0 + NB l + t): every load of a thread in flight before its first LDS write
        constexpr int NPT = BN3 * (K / 4), NP = (NPT + NT - 1) / NT;
        f4 pc[NP];
#pragma unroll
        for (int i = 0; i < NP; i++) {
            const int p = tid + i * NT, r = p / (K / 4), u = p % (K / 4);
            pc[i] = p < NPT ? *reinterpret_cast<const f4 *>(a.wp + (size_t)(n0 + NB * (r & 31) + (r >> 5)) * K + 4 * u) : f4{ 0.f, 0.f, 0.f, 0.f };
        }
#pragma unroll
        for (int i = 0; i < NP; i++) {
            const int p = tid + i * NT, r = p / (K / 4), u = p % (K / 4);
            if (p < NPT) *reinterpret_cast<f4 *>(wp_s + r * LDB + 4 * u) = pc[i];
        }
    }
    if (tid < BN3) { sc3_s[tid] = a.s3[n0 + tid]; sh3_s[tid] = a.b3[n0 + tid]; }
    __syncthreads();
    if (ntile == 0) return;

    const int qh = lane / LPP, cl = lane % LPP;
    const int li = lane & 31, lh = lane >> 5;
    const __amdgpu_buffer_rsrc_t irsrc = mbn_make_rsrc(a.in, a.in_bytes);
    const __amdgpu_buffer_rsrc_t orsrc = mbn_make_rsrc(a.out, (unsigned)(a.m * a.n * 4));
    float *const a_w = a_s + wave_u * (AH * ABUF);
    auto aw_at = [&](const int hh) __attribute__((always_inline)) {
        const int row = 8 * hh * (AH - 1) + qh;
        return KS == 16 ? row * KS + ((cl ^ ((row >> 2) & 3)) << 2) : row * KS + ((cl ^ ((row >> 1) & 7)) << 2);
    };
    const int aw[2] = { aw_at(0), aw_at(1) };
    auto fra_at = [&](const int u) __attribute__((always_inline)) {
        return KS == 16 ? li * KS + (((u + lh) ^ ((li >> 2) & 3)) << 2) : li * KS + (((u + lh) ^ ((li >> 1) & 7)) << 2);
    };
    const int fra[4] = { fra_at(0), fra_at(2), fra_at(KS == 32 ? 4 : 0), fra_at(KS == 32 ? 6 : 2) };
    const float *const bl01 = wp_s + li * LDB + lh * 4;
    const float *const bl23 = bl01 + (NB == 4 ? 64 : 0) * LDB;                 // (blocks 2, 3 of the 128-channel slice: a second base, the immediate is 16 bits)
    // byte offset of this lane's first pixel (pixel 2 * pair) inside a tile, per half-round; the second pixel is + K * 4 (an immediate). A pixel
    // past the end has an offset >= m * K * 4 = num_records: the buffer unit returns zeros, no check needed.
    unsigned lo[AH];
#pragma unroll
    for (int hh = 0; hh < AH; hh++) lo[hh] = (unsigned)(2 * (8 * hh * (AH - 1) + qh)) * (unsigned)(K * 4) + (unsigned)(cl * 16);

    f4 xr[PD][2];
    // loads of half-round u (of the tile at byte offset tb) into register slot sl
    auto ldx = [&](const int u, unsigned tb, const int sl) __attribute__((always_inline)) {
        const int hh = AH == 2 ? (u & 1) : 0, cb = AH == 2 ? (u >> 1) * 128 : u * 64;
        const unsigned vo = lo[hh] + tb;
        xr[sl][0] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(irsrc, vo + (unsigned)cb, 0, 0));
        xr[sl][1] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(irsrc, vo + (unsigned)(cb + K * 4), 0, 0));
    };
    auto a_write = [&](const int u, const int sl) __attribute__((always_inline)) {
        const int hh = AH == 2 ? (u & 1) : 0, buf = AH == 2 ? ((u >> 1) & 1) * ABUF : 0;
        *reinterpret_cast<f4 *>(a_w + buf + aw[hh]) = xr[sl][0];
        *reinterpret_cast<f4 *>(a_w + buf + aw[hh] + 16 * KS) = xr[sl][1];
    };
    f16v acc[NB];
    f4 fa[2], fb[2][NB];
    auto ldfrag_a = [&](const int u, const int g) __attribute__((always_inline)) {
        const int buf = AH == 2 ? ((u >> 1) & 1) * ABUF : 0, idx = AH == 2 ? 2 * (u & 1) + g : g;
        fa[g] = *reinterpret_cast<const f4 *>(a_w + buf + fra[idx]);
    };
    auto ldfrag_b = [&](const int u, const int g) __attribute__((always_inline)) {
        fb[g][0] = *reinterpret_cast<const f4 *>(bl01 + u * 16 + g * 8);
        fb[g][1] = *reinterpret_cast<const f4 *>(bl01 + 32 * LDB + u * 16 + g * 8);
        if constexpr (NB == 4) {
            fb[g][NB - 2] = *reinterpret_cast<const f4 *>(bl23 + u * 16 + g * 8);
            fb[g][NB - 1] = *reinterpret_cast<const f4 *>(bl23 + 32 * LDB + u * 16 + g * 8);
        }
    };
    auto mfma8 = [&](const int g, const int s0, const bool first) __attribute__((always_inline)) {
        const f16v zero = { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f };
#pragma unroll
        for (int s = s0; s < s0 + 2; s++)
#pragma unroll
            for (int t = 0; t < NB; t++)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[g][s], fb[g][t][s], (first && s == s0) ? zero : acc[t], 0, 0, 0);
    };
    // accumulator register r of block t: channel n0 + 4 li + t, tile row 8 (r >> 2) + 4 lh + (r & 3) = pixel m0 + 16 ((r >> 2) & 1) + 8 lh + 2 (r & 3) + (r >> 3)
    auto epilogue_mode = [&](unsigned m0, const bool inside) __attribute__((always_inline)) {
        float sc[NB], sh[NB];
#pragma unroll
        for (int t = 0; t < NB; t++) { sc[t] = sc3_s[NB * li + t]; sh[t] = sh3_s[NB * li + t]; }
        const unsigned rowb = (unsigned)a.n * 4u;
        const unsigned lane_off = (unsigned)(8 * lh) * rowb + (unsigned)(n0 + NB * li) * 4u;
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            f2 p[NB];
#pragma unroll
            for (int t = 0; t < NB; t++) p[t] = __builtin_elementwise_fma(f2{ acc[t][r], acc[t][r + 1] }, f2{ sc[t], sc[t] }, f2{ sh[t], sh[t] });
            float o[2][NB];
#pragma unroll
            for (int hh = 0; hh < 2; hh++)
#pragma unroll
                for (int t = 0; t < NB; t++) o[hh][t] = relu6(hh ? p[t].y : p[t].x);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int hh = 0; hh < 2; hh++) {
                const int rr = r + hh;
                const unsigned soff = (m0 + 16 * ((rr >> 2) & 1) + 2 * (rr & 3) + (rr >> 3)) * rowb;
                // rows past m (the last tile): the whole offset through the VGPR, so the descriptor's range check drops them
                if constexpr (NB == 4) {
                    const u4 v = __builtin_bit_cast(u4, f4{ o[hh][0], o[hh][1], o[hh][NB - 2], o[hh][NB - 1] });
                    if (inside) __builtin_amdgcn_raw_buffer_store_b128(v, orsrc, lane_off, soff, 0);
                    else __builtin_amdgcn_raw_buffer_store_b128(v, orsrc, lane_off + soff, 0, 0);
                } else {
                    typedef unsigned u2e __attribute__((ext_vector_type(2)));
                    const u2e v = __builtin_bit_cast(u2e, f2{ o[hh][0], o[hh][1] });
                    if (inside) __builtin_amdgcn_raw_buffer_store_b64(v, orsrc, lane_off, soff, 0);
                    else __builtin_amdgcn_raw_buffer_store_b64(v, orsrc, lane_off + soff, 0, 0);
                }
            }
            if constexpr (NB == 4) asm volatile("s_nop 1" ::: "memory");      // store-data hazard of 16-byte buffer stores (mbn_f32_dwpw3.hip)
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto epilogue = [&](unsigned m0) __attribute__((always_inline)) {
        if (m0 + WT <= mtot) epilogue_mode(m0, true);
        else epilogue_mode(m0, false);
    };

    // ---- prologue: half-rounds 0 .. AH + PD - 1 of the wave's flattened (tile, half-round) sequence requested, the first AH written into the A tile.
    // A half-round's register slot is its flattened index mod PD. When PD > NS the parity of the tile enters the slot: the loop body is unrolled over
    // TP = PD / NS (>= 1) tiles so every slot index stays a literal.
    constexpr int TP = PD > NS ? PD / NS : 1;
    const unsigned KB = (unsigned)(K * 4);
    auto m0_at = [&](int i) __attribute__((always_inline)) { return (unsigned)tile_at(i < ntile ? i : ntile - 1) * WT; };   // past the end: the last tile again (valid addresses, unused results)
    int it = 0;
#pragma unroll
    for (int h = 0; h < AH; h++) {
        ldx(h % NS, m0_at(h / NS) * KB, h % PD);
        a_write(h % NS, h % PD);
    }
#pragma unroll
    for (int h = AH; h < AH + PD; h++) ldx(h % NS, m0_at(h / NS) * KB, h % PD);
    ldfrag_a(0, 0);
    ldfrag_b(0, 0);
    bool pendE = false;
    unsigned m0E = 0;

    // ---- main loop: one unrolled pass = TP tiles of NS substeps. Substep u of tile it + tp (flattened index f = tp * NS + u): MFMAs of k groups 2 u, 2 u + 1;
    // half-round f + AH goes from its registers into the A tile; its slot is refilled with half-round f + AH + PD (one or two tiles ahead when past the end)
    bool done = false;
    while (!done) {
#pragma unroll
        for (int tp = 0; tp < TP; tp++) {
            const unsigned m0M = m0_at(it);
            const unsigned tb[3] = { m0M * KB, m0_at(it + 1) * KB, m0_at(it + 2) * KB };
            if (pendE) {
                epilogue(m0E);
                pendE = false;
            }
#pragma unroll
            for (int u = 0; u < NS; u++) {
                const int f = tp * NS + u, hd = u + AH, hl = u + AH + PD;
                __builtin_amdgcn_sched_barrier(0);
                ldfrag_a(u, 1);
                ldfrag_b(u, 1);
                __builtin_amdgcn_sched_barrier(0);
                mfma8(0, 0, u == 0);
                mfma8(0, 2, false);
                __builtin_amdgcn_sched_barrier(0);
                ldfrag_b((u + 1) % NS, 0);
                a_write(hd % NS, (f + AH) % PD);              // half-round f + AH: registers -> A tile (its loads were issued PD substeps ago)
                ldx(hl % NS, tb[hl / NS], (f + AH) % PD);     // ... and its slot refilled ((f + AH + PD) % PD is the same slot)
                ldfrag_a((u + 1) % NS, 0);
                __builtin_amdgcn_sched_barrier(0);
                mfma8(1, 0, false);
                mfma8(1, 2, false);
                __builtin_amdgcn_sched_barrier(0);
            }
            pendE = true;
            m0E = m0M;
            it++;
            if (it >= ntile) { done = true; break; }
        }
    }
    epilogue(m0E);
}

}   // namespace

// 1 when the resident-filter form can take this call: fp32, BN + ReLU6, Cin 64 / 128 / 256 with Cout a multiple of 128 (128-channel slices), or Cin 512 with Cout
// a multiple of 64 (64-channel slices: what of a K = 512 filter fits in LDS); 16-byte aligned operands, 32-bit byte offsets, at least one workgroup per XCD and slice
int mbn_f32_pw3_eligible(const mbn_call &c, const float *out, const float *in, const float *filt, long m, int cin, int op_size)
{
    if (c.dtype != MBN_DT_F32 || c.act != MBN_ACT_RELU6 || !c.scale || !c.shift) return 0;
    if (cin != 64 && cin != 128 && cin != 256 && cin != 512) return 0;
    const int bn = cin == 512 ? 64 : 128;
    if ((op_size % bn) != 0 || op_size > 1024 || m <= 0) return 0;
    if ((double)(m + 64) * cin * 4.0 >= 4294967296.0 || (double)(m + 64) * op_size * 4.0 >= 4294967296.0) return 0;
    if (((uintptr_t)out % 16) || ((uintptr_t)in % 16) || ((uintptr_t)filt % 16) || ((uintptr_t)c.scale % 16) || ((uintptr_t)c.shift % 16)) return 0;
    if (c.ctx->num_cus / (8 * (op_size / bn)) < 1) return 0;
    return 1;
}

int mbn_launch_f32_pw3(const mbn_call &c, float *out, const float *in, const float *filt, long m, int cin, int op_size)
{
    if (!mbn_f32_pw3_eligible(c, out, in, filt, m, cin, op_size)) return MBN_EUNSUPPORTED;
    const int bn = cin == 512 ? 64 : 128, nw = cin == 512 ? 12 : 8;
    Pw3Args a;
    a.out = out; a.in = in; a.wp = filt; a.s3 = c.scale; a.b3 = c.shift;
    a.m = m; a.n = op_size; a.nh = op_size / bn;
    a.tiles = (int)((m + WT - 1) / WT);
    a.in_bytes = (unsigned)((double)m * cin * 4.0);
    int per_xcd = c.ctx->num_cus / 8;
    per_xcd -= per_xcd % a.nh;
    const long tiles_xcd = (a.tiles + 7) / 8;
    if ((long)per_xcd > tiles_xcd * a.nh) per_xcd = (int)(tiles_xcd * a.nh);
    const dim3 g((unsigned)(per_xcd * 8)), b((unsigned)(64 * nw));
    // prefetch depth (half-rounds ahead): 2 / 4 / 4. Deeper (4 / 8 / 8: up to two tiles ahead; lab exp1 = 3) measured equal on every layer
    // (profiles/r06/m_*): the short-K layers are bound by their output stream and the matrix pipe, not by load latency.
#ifdef MBN_LAB
    if (g_mbn_tune.exp1 == 3 && cin <= 256) {
        if (cin == 64) hipLaunchKernelGGL((pw3_f32<64, 32, 4>), g, b, 0, c.stream, a);
        else if (cin == 128) hipLaunchKernelGGL((pw3_f32<128, 32, 8>), g, b, 0, c.stream, a);
        else hipLaunchKernelGGL((pw3_f32<256, 16, 8>), g, b, 0, c.stream, a);
        return MBN_OK;
    }
#endif
    if (cin == 64) hipLaunchKernelGGL((pw3_f32<64, 32, 2>), g, b, 0, c.stream, a);
    else if (cin == 128) hipLaunchKernelGGL((pw3_f32<128, 32, 4>), g, b, 0, c.stream, a);
    else if (cin == 256) hipLaunchKernelGGL((pw3_f32<256, 16, 4>), g, b, 0, c.stream, a);
    else hipLaunchKernelGGL((pw3_f32<512, 16, 4, 2, 12>), g, b, 0, c.stream, a);
    return MBN_OK;
}
