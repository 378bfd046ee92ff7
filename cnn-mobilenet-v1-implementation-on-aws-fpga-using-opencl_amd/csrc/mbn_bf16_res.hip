// mbn_bf16_res.hip — a RUN of depthwise 3x3 (stride 1) -> pointwise 1x1 blocks on a small map in ONE launch, bf16 mode, the activations
// RESIDENT in LDS from the first block's input to the last block's output (round 6).
// Replaces the launch pairs `depthwise` + `pointwise` (kernel.cl:62-92 + 94-114) of consecutive blocks with equal shapes — the five 256 -> 256
// blocks on the 10 x 10 map of the 0.5x160 network (layers 14-23 of the sequence MobileNet.c:322-2599) — which as five fused launches took
// 5 x 25 us for work whose HBM floor is 5 x 9 us: at 100 pixels per image a launch is mostly its prologue, its tail round and the gap to the next
// launch. An image's whole map is 100 x 256 bf16 = 51 KB: a workgroup keeps it in LDS through all the blocks, and only the filters (128 KB per
// block, bf16) come from the L2, as MFMA operands straight into registers.
//
// One workgroup = 8 waves = one image at a time (persistent over images). Two LDS images of the map, both [pixel][C + 8] bf16 (528-byte rows):
//   X: the block's input with a one-pixel ZERO border ((H + 2) x (W + 2) pixels): the depthwise window reads need no edge logic;
//   Y: the depthwise output = the pointwise GEMM's activation operand (H x W pixels in 104 rows; the GEMM's four 32-pixel column blocks read on into the constants behind
//      it — inside the allocation, those columns are never stored).
// Per block: (1) depthwise as a COLUMN MARCH: a lane owns one column x 8 channels (its 9 taps + scale / shift in registers for the whole block, read from an LDS
//   copy that the workgroup fetched once while the previous block's GEMM ran) and walks down the rows; every input row is read and widened ONCE (3 ds_read_b128) and feeds the three output
//   rows it belongs to (three running sums, same dy-major fma chain as the separate kernels): 24 widening instructions per output pixel group instead of
//   72 — the first form of this kernel (9 reads + 72 widenings per output) was exactly as fast as five separate launches, both bound by that VALU work.
//   Waves 0-3 take two columns each over the whole height, waves 4-7 (the partners on the same SIMDs) a quarter of the rows of the remaining columns (the phase is
//   bound by the SIMD's VALU issue, so what counts is the SUM of row steps of its two waves — a job of n rows costs n + 2 — not their balance: giving waves 4-7 the
//   lower rows of the first eight columns as well, 10 + 9 steps instead of 12 + 5, measured 17 % slower in this phase);
//   BN + ReLU6, round to bf16, one ds_write_b128 into Y per output; barrier;
//   (2) pointwise, transposed: wave w owns output channels 32 w .. 32 w + 31 as the ROWS of v_mfma_f32_32x32x16_bf16 (its filter rows are the
//   A operand: 16 registers-quads loaded from global memory per block, prefetched under the depthwise phase), the pixels are the columns (B operand:
//   ds_read_b128 of Y, one per MFMA, conflict-free on the 528-byte rows); C/D puts 4 consecutive output channels of ONE pixel into a lane:
//   BN + ReLU6, round, one ds_write_b64 into X's interior (the next block's input); barrier.
// Same arithmetic as the separate bf16 launches (fp32 products of bf16 operands, fp32 sums, every layer output rounded to bf16 RNE) in another
// summation order for the pointwise part: within the bf16 tolerance of the parity tests, like mbn_bf16_dwpw2.hip.
// Envelope: C = 256 in and out, stride 1, TF-SAME padding (pad 1), H * W <= 104, (H + 2) * (W + 2) <= 144, 1 ... 8 blocks.
#include "mbn_internal.h"
#include "mbn_epilogue.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f8 __attribute__((ext_vector_type(8)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));
typedef unsigned u2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
typedef mbn_f16v f16v;

constexpr int RES_MAXBLK = 8;
constexpr int XPIX = 144, YPIX = 128, YROWS = 104;   // pixel rows of the two LDS images (Y holds 104 — a map inside the envelope has at most 100 pixels; the GEMM's reads of
                                                     // pixel rows up to 127 run on into the constants behind it: inside the allocation, results unused)

struct ResArgs {
    __bf16 *out;
    const __bf16 *in;
    int batch, h, w, nblk;
    int dbg;                // lab ablations (exp0 = 900 + bits): 1 no depthwise march, 2 no MFMA loop, 4 no epilogue, 8 no tap / filter prefetch loads, 16 no image load / store (timing only)
    const float *wd[RES_MAXBLK], *s2[RES_MAXBLK], *b2[RES_MAXBLK], *s3[RES_MAXBLK], *b3[RES_MAXBLK];
    const __bf16 *wp[RES_MAXBLK];
};

__device__ __forceinline__ f8 widen8(u4v p)
{
    f8 r;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        r[2 * i] = __builtin_bit_cast(float, p[i] << 16);
        r[2 * i + 1] = __builtin_bit_cast(float, p[i] & 0xffff0000u);
    }
    return r;
}
__device__ __forceinline__ f8 ld8g(const float *p)
{
    const f4 a = *reinterpret_cast<const f4 *>(p), b = *reinterpret_cast<const f4 *>(p + 4);
    return f8{ a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w };
}
__device__ __forceinline__ float relu6(float v) { return fminf(fmaxf(v, 0.f), 6.f); }

template <int C>
__global__ __launch_bounds__(512) void res_blocks_bf16(ResArgs a)
{
    static_assert(C == 256, "8 waves x 32 output channels");
    constexpr int RS = C + 8;                          // LDS row stride in bf16 (528 bytes: 16-byte aligned, 33 sixteen-byte units: conflict-free operand reads)
    constexpr int RSB = RS * 2;                        // ... in bytes
    constexpr int G = C / 8;                           // 8-channel groups per pixel (32)
    constexpr int PSTEP = 512 / G;                     // pixels covered by the workgroup per depthwise round (16)
    constexpr int NIT = YPIX / PSTEP;                  // depthwise rounds (8: pixels prow, prow + 16, ... < H * W)
    constexpr int KG = C / 16;                         // MFMA k steps per block (16)
    // one allocation, explicit order: X | Y | pointwise scale, shift of every block | depthwise taps + scale + shift of the block at hand (158,592 bytes)
    __shared__ __attribute__((aligned(16))) char lds_raw[XPIX * RSB + YROWS * RSB + RES_MAXBLK * 2 * C * 4 + 11 * C * 4];
    __bf16 *const x_s = reinterpret_cast<__bf16 *>(lds_raw), *const y_s = reinterpret_cast<__bf16 *>(lds_raw + XPIX * RSB);
    float *const sb3_s = reinterpret_cast<float *>(lds_raw + XPIX * RSB + YROWS * RSB);      // the epilogue reads them per block: LDS, not a memory round trip
    float *const tp_s = sb3_s + RES_MAXBLK * 2 * C;                                          // [9 taps | s2 | b2][C]: loaded ONCE per block and workgroup (as register prefetches
                                                                                             // every wave fetched them for itself: 90 KB of L2 traffic per block beside the 128 KB filter)

#ifdef MBN_LAB
    const int dbg = a.dbg;                             // ablation bits: run-time in the lab build only
#else
    constexpr int dbg = 0;
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int P = a.h * a.w, WB = a.w + 2;
    char *const xb = reinterpret_cast<char *>(x_s), *const yb = reinterpret_cast<char *>(y_s);

    // zero X once: the border is never written again
    for (int i = tid; i < XPIX * RS / 8; i += 512) reinterpret_cast<u4v *>(x_s)[i] = u4v{ 0u, 0u, 0u, 0u };
    for (int i = tid; i < a.nblk * C; i += 512) {
        const int blk = i / C, ch = i % C;
        sb3_s[blk * 2 * C + ch] = a.s3[blk][ch];
        sb3_s[blk * 2 * C + C + ch] = a.b3[blk][ch];
    }

    // ---- image load / store role: channel group cg, pixels prow + 16 k (offsets recomputed where used: nothing of this lives through the blocks)
    const int cg = tid % G, prow = tid / G;
    const float inv_w = 1.0f / (float)a.w;
    auto xint = [&](int q) __attribute__((always_inline)) {              // X interior byte offset of pixel q (exact quotient for q < 128)
        const int y = (int)__builtin_fmaf((float)q, inv_w, 0.5f * inv_w), x = q - y * a.w;
        return (unsigned)(((y + 1) * WB + x + 1) * RSB);
    };
    // ---- pointwise role: output channels 32 wave + li (A operand rows), pixels 32 b + li (B operand columns)
    const unsigned yfrag = (unsigned)(li * RSB + lh * 16);              // + b * 32 * RSB + g * 32

    // ---- depthwise march role: column pair jobs (see the header); lane = (column of the pair, channel group)
    const int cgm = lane & 31;
    const int CP = (a.w + 1) / 2, RQ = (a.h + 3) / 4;
    // the 704 sixteen-byte pieces of a block's depthwise constants, two per thread: piece tid is taps; piece 512 + tid is taps in wave 0, scale in wave 1, shift in
    // wave 2 and absent above — the choice is per wave, so the base stays on the scalar unit and the lane part is one register for every block
    const bool tp2 = wave_u < 3;
    auto tap_piece = [&](int blk, int second) __attribute__((always_inline)) {
        const float *base = !second ? a.wd[blk] : wave_u == 0 ? a.wd[blk] + 8 * C : wave_u == 1 ? a.s2[blk] : a.b2[blk];
        return *reinterpret_cast<const f4 *>(base + 4 * (second ? lane : tid));
    };
    {
        const f4 p0 = tap_piece(0, 0), p1 = tp2 ? tap_piece(0, 1) : f4{ 0.f, 0.f, 0.f, 0.f };
        *reinterpret_cast<f4 *>(tp_s + 4 * tid) = p0;
        if (tp2) *reinterpret_cast<f4 *>(tp_s + 4 * (tid + 512)) = p1;
    }

    u4v wfr[KG];                                                          // this wave's filter rows of the coming block: k = 16 g + 8 lh .. + 7 of output channel 32 wave + li
    {
        const __bf16 *wrow = a.wp[0] + (size_t)(32 * wave_u + li) * C + 8 * lh;
#pragma unroll
        for (int g = 0; g < KG / 2; g++) wfr[g] = *reinterpret_cast<const u4v *>(wrow + 16 * g);
#pragma unroll
        for (int g = KG / 2; g < KG; g++) wfr[g] = u4v{ 0u, 0u, 0u, 0u };
    }

    for (int n = blockIdx.x; n < a.batch; n += gridDim.x) {
        __syncthreads();                                                  // (the previous image's output has left X; first pass: the zeroing is done)
        // ---- image -> X interior: 16-byte pieces
        if (!(dbg & 16)) {
            const __bf16 *src = a.in + (size_t)n * P * C;
            int prow_n = prow;                                            // laundered: the eight X offsets are recomputed per image instead of living (spilled) across the block loop
            asm volatile("" : "+v"(prow_n));
            u4v pc[NIT];
#pragma unroll
            for (int k = 0; k < NIT; k++) {
                const int q = prow_n + PSTEP * k;
                pc[k] = q < P ? *reinterpret_cast<const u4v *>(reinterpret_cast<const char *>(src) + (unsigned)((q * C + cg * 8) * 2)) : u4v{ 0u, 0u, 0u, 0u };   // (uniform base + 32-bit lane offset)
            }
#pragma unroll
            for (int k = 0; k < NIT; k++) {
                const int q = prow_n + PSTEP * k;
                if (q < P) *reinterpret_cast<u4v *>(xb + xint(q) + cg * 16) = pc[k];
            }
        }
        __syncthreads();

        for (int blk = 0; blk < a.nblk; blk++) {
            // ---- (1) depthwise 3x3 + BN + ReLU6: X -> Y, column march; this lane's taps / scale / shift from LDS into registers first
            f8 tap[9], sc, sh;
            {
                const float *tl = tp_s + cgm * 8;
#pragma unroll
                for (int t = 0; t < 9; t++) {
                    const f4 lo = *reinterpret_cast<const f4 *>(tl + t * C), hi = *reinterpret_cast<const f4 *>(tl + t * C + 4);
                    tap[t] = f8{ lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w };
                }
                const f4 a0 = *reinterpret_cast<const f4 *>(tl + 9 * C), a1 = *reinterpret_cast<const f4 *>(tl + 9 * C + 4);
                const f4 b0 = *reinterpret_cast<const f4 *>(tl + 10 * C), b1 = *reinterpret_cast<const f4 *>(tl + 10 * C + 4);
                sc = f8{ a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w };
                sh = f8{ b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w };
            }
            if (!(dbg & 1))
            for (int cpj = (wave_u < 4 ? wave_u : 4); cpj < CP; cpj += (wave_u < 4 ? CP : 1)) {
                const int col = 2 * cpj + (lane >> 5);
                const int r0 = wave_u < 4 ? 0 : (wave_u - 4) * RQ;
                const int nr = wave_u < 4 ? a.h : min(RQ, a.h - r0);
                if (nr <= 0) break;
                const bool cok = col < a.w;
                const unsigned colx = (unsigned)(min(col, a.w - 1));                      // (a column past the map reads the last one: valid LDS, never written out)
                unsigned xa = (unsigned)((r0 * WB + colx) * RSB + cgm * 16);              // bordered row r0 + i, bordered columns colx .. colx + 2
                unsigned ya = (unsigned)(((r0 - 2) * a.w + (int)colx) * RSB + cgm * 16);  // output row r0 + i - 2
                f8 s0, s1, s2;
#pragma unroll
                for (int i = 0; i < 8; i++) s0[i] = s1[i] = s2[i] = 0.f;
                // input row i of the job: dy = 0 for output i (starts its sum), dy = 1 for output i - 1, dy = 2 for output i - 2 (completes it)
#define RES_DW_ROW(I, SA, SB, SC)                                                                                             \
                if ((I) < nr + 2) {                                                                                           \
                    const f8 x0 = widen8(*reinterpret_cast<const u4v *>(xb + xa)), x1 = widen8(*reinterpret_cast<const u4v *>(xb + xa + RSB)), \
                             x2 = widen8(*reinterpret_cast<const u4v *>(xb + xa + 2 * RSB));                                    \
                    f8 z;                                                                                                     \
                    _Pragma("unroll") for (int i_ = 0; i_ < 8; i_++) z[i_] = 0.f;                                             \
                    SA = __builtin_elementwise_fma(x0, tap[0], z);                                                            \
                    SA = __builtin_elementwise_fma(x1, tap[1], SA);                                                           \
                    SA = __builtin_elementwise_fma(x2, tap[2], SA);                                                           \
                    SB = __builtin_elementwise_fma(x0, tap[3], SB);                                                           \
                    SB = __builtin_elementwise_fma(x1, tap[4], SB);                                                           \
                    SB = __builtin_elementwise_fma(x2, tap[5], SB);                                                           \
                    SC = __builtin_elementwise_fma(x0, tap[6], SC);                                                           \
                    SC = __builtin_elementwise_fma(x1, tap[7], SC);                                                           \
                    SC = __builtin_elementwise_fma(x2, tap[8], SC);                                                           \
                    if ((I) >= 2 && cok) {                                                                                    \
                        bf8 o;                                                                                                \
                        const f8 v_ = __builtin_elementwise_fma(SC, sc, sh);                                                  \
                        _Pragma("unroll") for (int i_ = 0; i_ < 8; i_++) o[i_] = (__bf16)relu6(v_[i_]);                       \
                        *reinterpret_cast<bf8 *>(yb + ya) = o;                                                                \
                    }                                                                                                         \
                    xa += (unsigned)(WB * RSB);                                                                               \
                    ya += (unsigned)(a.w * RSB);                                                                              \
                }
                for (int i0 = 0; i0 < nr + 2; i0 += 3) {
                    RES_DW_ROW(i0, s0, s2, s1)
                    RES_DW_ROW(i0 + 1, s1, s0, s2)
                    RES_DW_ROW(i0 + 2, s2, s1, s0)
                }
#undef RES_DW_ROW
            }
            // the second half of THIS block's filter rows: requested now (the taps' registers are free until the prefetch behind the MFMAs), used from k step 8 on —
            // eight k steps of MFMAs on both waves of the SIMD cover the L2 round trip
            if (!(dbg & 8)) {
                const __bf16 *wrow = a.wp[blk] + (size_t)(32 * wave_u + li) * C + 8 * lh;
#pragma unroll
                for (int g = KG / 2; g < KG; g++) wfr[g] = *reinterpret_cast<const u4v *>(wrow + 16 * g);
            }
            __syncthreads();
            f4 tq0 = f4{ 0.f, 0.f, 0.f, 0.f }, tq1 = tq0;                 // the next block's depthwise constants, on their way from memory to LDS
            // ---- (2) pointwise 1x1 + BN + ReLU6: Y x filter -> X interior. D[channel][pixel] = sum_k W[channel][k] * Y[pixel][k]
            {
                f16v acc[4];
#pragma unroll
                for (int b = 0; b < 4; b++)
#pragma unroll
                    for (int r = 0; r < 16; r++) acc[b][r] = 0.f;
                if (!(dbg & 2))
#pragma unroll
                for (int g = 0; g < KG; g++) {
                    u4v yf[4];
#pragma unroll
                    for (int b = 0; b < 4; b++) yf[b] = *reinterpret_cast<const u4v *>(yb + yfrag + (unsigned)(b * 32 * RSB + g * 32));
#pragma unroll
                    for (int b = 0; b < 4; b++)
                        acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, wfr[g]), __builtin_bit_cast(bf8, yf[b]), acc[b], 0, 0, 0);
                }
                // the next block's filter rows (first half; the next image's first block after the last one): requested now, they fly under the epilogue and the barrier
                if (!(dbg & 8)) {
                    const int nb = blk + 1 < a.nblk ? blk + 1 : 0;
                    const __bf16 *wrow = a.wp[nb] + (size_t)(32 * wave_u + li) * C + 8 * lh;
#pragma unroll
                    for (int g = 0; g < KG / 2; g++) wfr[g] = *reinterpret_cast<const u4v *>(wrow + 16 * g);       // (the first half; the second behind the next depthwise phase: registers)
                    tq0 = tap_piece(nb, 0);                                // and this thread's two pieces of its depthwise constants, written into LDS behind the epilogue
                    if (tp2) tq1 = tap_piece(nb, 1);
                }
                // C/D: register r of block b = output channel 32 wave + 8 (r >> 2) + 4 lh + (r & 3) of pixel 32 b + li
                unsigned xi[4];
#pragma unroll
                for (int b = 0; b < 4; b++) xi[b] = xint(min(32 * b + li, P - 1)) + (unsigned)((32 * wave_u + 4 * lh) * 2);
                const float *s3p = sb3_s + blk * 2 * C + 32 * wave_u + 4 * lh, *b3p = s3p + C;
                if (!(dbg & 4))
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const f4 sc = *reinterpret_cast<const f4 *>(s3p + 8 * j), sh = *reinterpret_cast<const f4 *>(b3p + 8 * j);
#pragma unroll
                    for (int b = 0; b < 4; b++) {
                        const bf4 o = bf4{ (__bf16)relu6(fmaf(acc[b][4 * j], sc.x, sh.x)), (__bf16)relu6(fmaf(acc[b][4 * j + 1], sc.y, sh.y)),
                                           (__bf16)relu6(fmaf(acc[b][4 * j + 2], sc.z, sh.z)), (__bf16)relu6(fmaf(acc[b][4 * j + 3], sc.w, sh.w)) };
                        if (32 * b + li < P) *reinterpret_cast<bf4 *>(xb + xi[b] + (unsigned)(16 * j)) = o;
                    }
                }
            }
            *reinterpret_cast<f4 *>(tp_s + 4 * tid) = tq0;                 // (every lane read this block's constants before the barrier in front of the GEMM)
            if (tp2) *reinterpret_cast<f4 *>(tp_s + 4 * (tid + 512)) = tq1;
            __syncthreads();
        }
        // ---- X interior -> output
        if (!(dbg & 16)) {
            __bf16 *dst = a.out + (size_t)n * P * C;
            int prow_n = prow;
            asm volatile("" : "+v"(prow_n));
#pragma unroll
            for (int k = 0; k < NIT; k++) {
                const int q = prow_n + PSTEP * k;
                if (q < P) *reinterpret_cast<u4v *>(reinterpret_cast<char *>(dst) + (unsigned)((q * C + cg * 8) * 2)) = *reinterpret_cast<const u4v *>(xb + xint(q) + cg * 16);
            }
        }
    }
}

}   // namespace

// 1 when a run of blocks can stay resident: C = 256, stride 1 with pad 1 (TF-SAME), a map of at most 128 pixels whose bordered form fits 144
int mbn_bf16_res_eligible(int rows, int cols, int channels, int nblocks)
{
    return channels == 256 && rows >= 1 && cols >= 1 && rows * cols <= YROWS && (rows + 2) * (cols + 2) <= XPIX && nblocks >= 1 && nblocks <= RES_MAXBLK;
}

int mbn_launch_bf16_res_blocks(mbn_context *ctx, hipStream_t stream, void *out, const void *in, const mbn_block_params *blocks, int nblocks, int batch,
                               int rows, int cols, int channels)
{
    if (!mbn_bf16_res_eligible(rows, cols, channels, nblocks)) return MBN_EUNSUPPORTED;
    if (!out || !in || !blocks || batch <= 0) return MBN_EINVAL;
    if (((uintptr_t)out % 16) || ((uintptr_t)in % 16)) return MBN_EUNSUPPORTED;
    ResArgs a;
    a.out = (__bf16 *)out; a.in = (const __bf16 *)in;
    a.batch = batch; a.h = rows; a.w = cols; a.nblk = nblocks;
    a.dbg = g_mbn_tune.exp0 >= 900 ? g_mbn_tune.exp0 - 900 : 0;
    for (int i = 0; i < nblocks; i++) {
        const mbn_block_params &b = blocks[i];
        const void *ptrs[] = { b.wd, b.s2, b.b2, b.wp_bf16, b.s3, b.b3 };
        for (const void *p : ptrs)
            if (!p || ((uintptr_t)p % 16)) return p ? MBN_EUNSUPPORTED : MBN_EINVAL;
        a.wd[i] = (const float *)b.wd; a.s2[i] = (const float *)b.s2; a.b2[i] = (const float *)b.b2;
        a.wp[i] = (const __bf16 *)b.wp_bf16; a.s3[i] = (const float *)b.s3; a.b3[i] = (const float *)b.b3;
    }
    for (int i = nblocks; i < RES_MAXBLK; i++) { a.wd[i] = a.s2[i] = a.b2[i] = a.s3[i] = a.b3[i] = nullptr; a.wp[i] = nullptr; }
    long grid = ctx->num_cus;
    if (grid > batch) grid = batch;
    hipLaunchKernelGGL((res_blocks_bf16<256>), dim3((unsigned)grid), dim3(512), 0, stream, a);
    return MBN_OK;
}
