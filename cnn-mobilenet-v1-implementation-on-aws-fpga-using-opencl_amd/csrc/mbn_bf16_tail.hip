// mbn_bf16_tail.hip — the last two blocks of the network and the global pool in ONE launch, bf16 mode, an image's maps RESIDENT in LDS (round 6):
//   depthwise 3x3 stride 2 (C0) -> pointwise C0 -> C1 -> depthwise 3x3 stride 1 (C1) -> pointwise C1 -> C1 -> global average pool
// = layers 24-28 of the sequence MobileNet.c:322-2599 + :2601-2679 (kernel.cl:62-92, 94-114, 116-132) at the 0.5x160 network's sizes: 10 x 10 x 256 in,
// 5 x 5 x 512 inside, 512 pooled values out per image. As five launches these layers took 8.7 + 9.7 + 11.9 + 13.1 + 7.1 us at batch 512 plus four launch
// gaps — 13 % of the 0.5 ms step — for 26 MB of input: a 5 x 5 map gives a launch 12,800 pixels to spread over 256 CUs, and the pool walks 25 dependent
// loads. Here one workgroup (8 waves) takes TWO images per pass through all five layers (the filters — 768 KB per pass, read by every workgroup — are what the
// launch moves most of: two images per pass halve them per image):
//   depthwise 1 reads the images straight from memory (every input pixel feeds at most four outputs) -> Y [64][C1 + 8] bf16: a depthwise output = the next
//   GEMM's B operand, rows img * P1 + p; X1 [50][C1 + 8]: the first pointwise output, later the second one's (Z); both depthwise layers' taps + scale + shift
//   and both pointwise layers' scale + shift stay in LDS for the whole launch (160,544 bytes in all).
//   The pointwise GEMMs are transposed as in mbn_bf16_res.hip: a wave's output channels are the rows of the matrix instruction (filter rows from the L2 straight
//   into registers as the A operand), the 50 pixels the columns (B operand from Y: four 16-pixel blocks). v_mfma_f32_16x16x32_bf16: its A layout lets four lanes
//   read 64 contiguous bytes of a filter row (32x32x16: two lanes, 32 bytes) — the filter fetch is bound by lines handled per instruction. 512 output channels =
//   two 32-channel blocks per wave: both resident for K = 256 (128 VGPRs), one after the other for K = 512, the second one's rows requested in halves as the
//   first one's k steps release their registers.
// Arithmetic: the depthwise sums are the dy-major fma chains of the stand-alone kernels; the pointwise sums are 16x16x32 products in k order, the instruction and
// order of the stand-alone K >= 256 kernel (pw_stream_bf16's M16 form); every layer output is rounded to bf16; the pool is pool_f32_nhwc's left-to-right fp32 sum
// and division: the same bits as the five launches (tests compare at the bf16 tolerance and report the difference, 0 on the shapes tried).
// Measured (batch 512, profiles/r06/x_*): 0.046 ms against 0.067 ms for the five launches in isolation (+ four launch gaps in the network); compile-time
// ablation of this code: depthwise phases 13.6 us, filter loads 17.5, matrix instructions + fragment reads 6, pool 2.3, the rest 6 — they add: one workgroup
// per CU walks its phases one after the other.
// Envelope: C0 = 256, C1 = 512, input map H0 x W0 with even sides <= 10 (stride 2 pads bottom / right only: TF-SAME on an even side), pad 1 for stride 1.
#include "mbn_internal.h"
#include "mbn_epilogue.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f8 __attribute__((ext_vector_type(8)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
typedef mbn_f16v f16v;

struct TailArgs {
    __bf16 *out;                 // [batch][C1]
    const __bf16 *in;            // [batch][h0][w0][C0]
    int batch, h0, w0;
    const float *wd0, *s0, *b0;  // depthwise 1 (stride 2): taps [3][3][C0], scale, shift
    const __bf16 *wp0;           // pointwise 1: [C1][C0]
    const float *s1, *b1;
    const float *wd1, *s2, *b2;  // depthwise 2 (stride 1): [3][3][C1]
    const __bf16 *wp1;           // pointwise 2: [C1][C1]
    const float *s3, *b3;
    int dbg;                     // lab ablations (exp0 = 800 + bits): 1 no depthwise phases, 2 no MFMAs, 4 no filter loads, 8 no image load, 16 no pool (timing only)
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
__device__ __forceinline__ f8 ld8s(const float *p)
{
    const f4 a = *reinterpret_cast<const f4 *>(p), b = *reinterpret_cast<const f4 *>(p + 4);
    return f8{ a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w };
}
__device__ __forceinline__ float relu6(float v) { return fminf(fmaxf(v, 0.f), 6.f); }

template <int C0, int C1>
__global__ __launch_bounds__(512) void tail_bf16(TailArgs a)
{
    static_assert(C0 == 256 && C1 == 512, "8 waves x two 32-channel row blocks; one pooled channel per thread");
    constexpr int RS0 = C0 + 8, RS1 = C1 + 8;          // LDS rows in bf16: 528 / 1040 bytes = 4 banks past a multiple of 64
    constexpr int RSB0 = RS0 * 2, RSB1 = RS1 * 2;
    constexpr int G0 = C0 / 8, G1 = C1 / 8;            // 8-channel groups per pixel
    constexpr int KS0 = C0 / 32, KS1 = C1 / 32;        // k steps of v_mfma_f32_16x16x32_bf16 (8, 16)
    constexpr int NIMG = 2, QMAX = NIMG * 25;          // images per pass; their pixels are rows img * P1 + p of Y and X1
    constexpr int YB = 64 * RSB1, X1B = QMAX * RSB1;   // Y: 64 rows (four 16-pixel column blocks are read; rows past the pass's pixels hold garbage that only reaches unused columns)
    __shared__ __attribute__((aligned(16))) char lds[YB + X1B + (11 * C0 + 11 * C1 + 4 * C1) * 4];      // 160,544 bytes
    char *const yb = lds, *const x1 = lds + YB;
    float *const tp0 = reinterpret_cast<float *>(lds + YB + X1B);              // [9 taps | scale | shift][C0]
    float *const tp1 = tp0 + 11 * C0;                                          // [9 taps | scale | shift][C1]
    float *const sb1 = tp1 + 11 * C1;                                          // pointwise 1: scale | shift [C1]
    float *const sb3 = sb1 + 2 * C1;                                           // pointwise 2

#ifdef MBN_LAB
    const int dbg = a.dbg;
#elif defined(MBN_TAIL_ABL)
    constexpr int dbg = MBN_TAIL_ABL;                  // tools: a compile-time ablation of the shipped code generation (timing only)
#else
    constexpr int dbg = 0;
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j16 = lane & 15, q4 = lane >> 4;
    const int h1 = a.h0 >> 1, w1 = a.w0 >> 1;
    const int P0 = a.h0 * a.w0, P1 = h1 * w1;

    // the wave's filter rows as A operands of v_mfma_f32_16x16x32_bf16: lane (j16, q4) holds k = 32 g + 8 q4 .. + 7 of row 16 blk16 + j16 — four lanes read 64 contiguous
    // bytes of a row (the 32x32x16 layout: two lanes, 32 bytes; the filter fetch is bound by the lines the L1 handles per instruction, not by bytes).
    // pointwise 1 (K = 256): both 32-channel blocks of the wave, index (blk * KS0 + g) * 2 + half; pointwise 2 (K = 512): one block at a time, index g * 2 + half
    u4v wf[2 * KS1];                                   // 128 VGPRs
    auto load_pw0 = [&]() __attribute__((always_inline)) {
        if (dbg & 4) return;
#pragma unroll
        for (int blk = 0; blk < 2; blk++)
#pragma unroll
            for (int hf = 0; hf < 2; hf++) {
                const __bf16 *wrow = a.wp0 + (size_t)(32 * (wave_u + 8 * blk) + 16 * hf + j16) * C0 + 8 * q4;
#pragma unroll
                for (int g = 0; g < KS0; g++) wf[(blk * KS0 + g) * 2 + hf] = *reinterpret_cast<const u4v *>(wrow + 32 * g);
            }
    };
    auto load_pw1 = [&](int blk, int g0, int g1) __attribute__((always_inline)) {       // k steps g0 ... g1 - 1 of the block's rows
        if (dbg & 4) return;
#pragma unroll
        for (int hf = 0; hf < 2; hf++) {
            const __bf16 *wrow = a.wp1 + (size_t)(32 * (wave_u + 8 * blk) + 16 * hf + j16) * C1 + 8 * q4;
#pragma unroll
            for (int g = 0; g < KS1; g++)
                if (g >= g0 && g < g1) wf[g * 2 + hf] = *reinterpret_cast<const u4v *>(wrow + 32 * g);
        }
    };
    load_pw0();

    // ---- constants of all four layers: once per workgroup, as 2624 sixteen-byte pieces in the LDS order [wd0 | s0 | b0 | wd1 | s2 | b2 | s1 | b1 | s3 | b3]: the six loads of a
    // thread are issued back to back (one loop per array took a memory round trip per iteration: 30 us of a 73 us kernel)
    {
        constexpr int E0 = 9 * C0 / 4, E1 = E0 + C0 / 4, E2 = E1 + C0 / 4, E3 = E2 + 9 * C1 / 4, E4 = E3 + C1 / 4, E5 = E4 + C1 / 4, E6 = E5 + C1 / 4, E7 = E6 + C1 / 4,
                      E8 = E7 + C1 / 4, E9 = E8 + C1 / 4;
        constexpr int NR = (E9 + 511) / 512;
        f4 cv[NR];
#pragma unroll
        for (int r = 0; r < NR; r++) {
            const int i = tid + 512 * r;
            const float *src = i < E0 ? a.wd0 + 4 * i : i < E1 ? a.s0 + 4 * (i - E0) : i < E2 ? a.b0 + 4 * (i - E1) : i < E3 ? a.wd1 + 4 * (i - E2) : i < E4 ? a.s2 + 4 * (i - E3) :
                               i < E5 ? a.b2 + 4 * (i - E4) : i < E6 ? a.s1 + 4 * (i - E5) : i < E7 ? a.b1 + 4 * (i - E6) : i < E8 ? a.s3 + 4 * (i - E7) : a.b3 + 4 * (i - E8);
            cv[r] = i < E9 ? *reinterpret_cast<const f4 *>(src) : f4{ 0.f, 0.f, 0.f, 0.f };
        }
#pragma unroll
        for (int r = 0; r < NR; r++) {
            const int i = tid + 512 * r;
            if (i < E9) *reinterpret_cast<f4 *>(tp0 + 4 * i) = cv[r];
        }
    }
    __syncthreads();

    // C/D of a 16 x 16 block (hf, b): lane (j16, q4), register r = channel 16 hf + 4 q4 + r of pixel row 16 b + j16: BN + ReLU6, round, 8 bytes into X1
    auto epilogue = [&](const f4 (&acc)[2][4], const float *sb, int cbase, int Q, int j16, int q4) __attribute__((always_inline)) {
#pragma unroll
        for (int hf = 0; hf < 2; hf++) {
            const int cb = cbase + 16 * hf + 4 * q4;
            const f4 sc = *reinterpret_cast<const f4 *>(sb + cb), sh = *reinterpret_cast<const f4 *>(sb + C1 + cb);
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const f4 v = acc[hf][b];
                const bf4 o = bf4{ (__bf16)relu6(fmaf(v.x, sc.x, sh.x)), (__bf16)relu6(fmaf(v.y, sc.y, sh.y)),
                                   (__bf16)relu6(fmaf(v.z, sc.z, sh.z)), (__bf16)relu6(fmaf(v.w, sc.w, sh.w)) };
                if (16 * b + j16 < Q) *reinterpret_cast<bf4 *>(x1 + (16 * b + j16) * RSB1 + cb * 2) = o;
            }
        }
    };

    for (int n0 = NIMG * blockIdx.x; n0 < a.batch; n0 += NIMG * gridDim.x) {
        const int nimg = min(NIMG, a.batch - n0), Q = nimg * P1;
        // ---- (b) depthwise 1, stride 2, pad bottom / right: the images (straight from memory: every input pixel feeds at most four outputs) -> Y [Q][C0]
        {
            int tb_ = tid;                                 // laundered per phase: the item offsets are recomputed where used instead of living (spilled) beside the filter rows
            asm volatile("" : "+v"(tb_));
#pragma unroll 1
            for (int it = tb_; it < ((dbg & 1) ? 0 : Q * G0); it += 512) {
                const int q = it / G0, cg = it % G0;
                const int img = q >= P1 ? 1 : 0, p = q - img * P1;
                const int oy = p / w1, ox = p - oy * w1;
                const char *src = reinterpret_cast<const char *>(a.in + (size_t)(n0 + img) * P0 * C0) + cg * 16;
                const float *tl = tp0 + cg * 8;
                u4v xr[9];
#pragma unroll
                for (int dy = 0; dy < 3; dy++)
#pragma unroll
                    for (int dx = 0; dx < 3; dx++) {
                        const int iy = 2 * oy + dy, ix = 2 * ox + dx;
                        xr[dy * 3 + dx] = (iy < a.h0 && ix < a.w0) ? *reinterpret_cast<const u4v *>(src + (unsigned)((iy * a.w0 + ix) * (C0 * 2))) : u4v{ 0u, 0u, 0u, 0u };
                    }
                f8 acc;
#pragma unroll
                for (int i = 0; i < 8; i++) acc[i] = 0.f;
#pragma unroll
                for (int t = 0; t < 9; t++) acc = __builtin_elementwise_fma(widen8(xr[t]), ld8s(tl + t * C0), acc);      // (a tap outside the map multiplies zero: the same sum)
                const f8 v = __builtin_elementwise_fma(acc, ld8s(tl + 9 * C0), ld8s(tl + 10 * C0));
                bf8 o;
#pragma unroll
                for (int i = 0; i < 8; i++) o[i] = (__bf16)relu6(v[i]);
                *reinterpret_cast<bf8 *>(yb + q * RSB0 + cg * 16) = o;
            }
        }
        __syncthreads();
        // ---- (c) pointwise 1: D[channel][pixel] = sum_k W[channel][k] Y[pixel][k], K = C0, the wave's two 32-channel blocks -> X1 [Q][C1]
        int jc_ = j16, qc_ = q4;                           // laundered: fragment and epilogue offsets are not carried (spilled) across the phases
        asm volatile("" : "+v"(jc_), "+v"(qc_));
#pragma unroll
        for (int blk = 0; blk < 2; blk++) {
            f4 acc[2][4];
#pragma unroll
            for (int hf = 0; hf < 2; hf++)
#pragma unroll
                for (int b = 0; b < 4; b++) acc[hf][b] = f4{ 0.f, 0.f, 0.f, 0.f };
            if (!(dbg & 2))
#pragma unroll
            for (int g = 0; g < KS0; g++) {
                u4v yf[4];
#pragma unroll
                for (int b = 0; b < 4; b++) yf[b] = *reinterpret_cast<const u4v *>(yb + (16 * b + jc_) * RSB0 + g * 64 + qc_ * 16);
#pragma unroll
                for (int hf = 0; hf < 2; hf++)
#pragma unroll
                    for (int b = 0; b < 4; b++)
                        acc[hf][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, wf[(blk * KS0 + g) * 2 + hf]), __builtin_bit_cast(bf8, yf[b]), acc[hf][b], 0, 0, 0);
            }
            if (blk == 1) load_pw1(0, 0, KS1);             // pointwise 2's first block: under this epilogue and the second depthwise layer
            epilogue(acc, sb1, 32 * (wave_u + 8 * blk), Q, jc_, qc_);
        }
        __syncthreads();
        // ---- (d) depthwise 2, stride 1, pad 1: X1 -> Y [Q][C1]
        {
            int td_ = tid;
            asm volatile("" : "+v"(td_));
#pragma unroll 1
            for (int it = td_; it < ((dbg & 1) ? 0 : Q * G1); it += 512) {
                const int q = it / G1, cg = it % G1;
                const int img = q >= P1 ? 1 : 0, p = q - img * P1;
                const int oy = p / w1, ox = p - oy * w1;
                const char *xi = x1 + (img * P1) * RSB1 + cg * 16;
                const float *tl = tp1 + cg * 8;
                f8 acc;
#pragma unroll
                for (int i = 0; i < 8; i++) acc[i] = 0.f;
#pragma unroll
                for (int dy = 0; dy < 3; dy++)
#pragma unroll
                    for (int dx = 0; dx < 3; dx++) {
                        const int iy = oy + dy - 1, ix = ox + dx - 1;
                        // (a wave's 64 lanes are the 64 channel groups of ONE pixel: the test is uniform, a tap outside the map is skipped by a scalar branch)
                        if (iy >= 0 && iy < h1 && ix >= 0 && ix < w1)
                            acc = __builtin_elementwise_fma(widen8(*reinterpret_cast<const u4v *>(xi + (iy * w1 + ix) * RSB1)), ld8s(tl + (dy * 3 + dx) * C1), acc);
                    }
                const f8 v = __builtin_elementwise_fma(acc, ld8s(tl + 9 * C1), ld8s(tl + 10 * C1));
                bf8 o;
#pragma unroll
                for (int i = 0; i < 8; i++) o[i] = (__bf16)relu6(v[i]);
                *reinterpret_cast<bf8 *>(yb + q * RSB1 + cg * 16) = o;
            }
        }
        __syncthreads();
        // ---- (e) pointwise 2, K = C1: the wave's two 32-channel blocks one after the other -> Z (= X1's place) [Q][C1]. The second block's filter rows are
        // requested in halves as the first block's k steps release their registers
        int je_ = j16, qe_ = q4;
        asm volatile("" : "+v"(je_), "+v"(qe_));
#pragma unroll
        for (int blk = 0; blk < 2; blk++) {
            f4 acc[2][4];
#pragma unroll
            for (int hf = 0; hf < 2; hf++)
#pragma unroll
                for (int b = 0; b < 4; b++) acc[hf][b] = f4{ 0.f, 0.f, 0.f, 0.f };
#pragma unroll
            for (int g = 0; g < KS1; g++) {
                if (!(dbg & 2)) {
                    u4v yf[4];
#pragma unroll
                    for (int b = 0; b < 4; b++) yf[b] = *reinterpret_cast<const u4v *>(yb + (16 * b + je_) * RSB1 + g * 64 + qe_ * 16);
#pragma unroll
                    for (int hf = 0; hf < 2; hf++)
#pragma unroll
                        for (int b = 0; b < 4; b++)
                            acc[hf][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, wf[g * 2 + hf]), __builtin_bit_cast(bf8, yf[b]), acc[hf][b], 0, 0, 0);
                }
                if (blk == 0 && g == KS1 / 2 - 1) load_pw1(1, 0, KS1 / 2);
            }
            if (blk == 0) load_pw1(1, KS1 / 2, KS1);
            else if (n0 + NIMG * (int)gridDim.x < a.batch) load_pw0();      // another pass: its pointwise 1, under this epilogue, the pool and its first depthwise layer
            epilogue(acc, sb3, 32 * (wave_u + 8 * blk), Q, je_, qe_);
        }
        __syncthreads();
        // ---- (f) global average pool: channel tid, pixels left to right, fp32 (pool_f32_nhwc's sum and division)
        if (!(dbg & 16)) {
            int tf_ = tid;
            asm volatile("" : "+v"(tf_));
            for (int img = 0; img < nimg; img++) {
                float s = 0.f;
                for (int p = 0; p < P1; p++) s += (float)*reinterpret_cast<const __bf16 *>(x1 + (img * P1 + p) * RSB1 + tf_ * 2);
                a.out[(size_t)(n0 + img) * C1 + tf_] = (__bf16)(s / (float)P1);
            }
        }
        __syncthreads();                                   // (Z is X1: the next pass's first pointwise layer writes it two barriers later, but its depthwise phase writes Y: keep the order simple)
    }
}

}   // namespace

int mbn_bf16_tail_eligible(int rows, int cols, int c0, int c1)
{
    return c0 == 256 && c1 == 512 && rows >= 2 && cols >= 2 && rows <= 10 && cols <= 10 && !(rows & 1) && !(cols & 1);
}

int mbn_launch_bf16_tail(mbn_context *ctx, hipStream_t stream, void *out, const void *in, const mbn_block_params *blocks, int batch, int rows, int cols, int c0, int c1)
{
    if (!mbn_bf16_tail_eligible(rows, cols, c0, c1)) return MBN_EUNSUPPORTED;
    if (!out || !in || !blocks || batch <= 0) return MBN_EINVAL;
    if (((uintptr_t)out % 2) || ((uintptr_t)in % 16) || (double)batch * rows * cols * c0 * 2 >= 4294967296.0) return MBN_EUNSUPPORTED;
    for (int i = 0; i < 2; i++) {
        const void *ptrs[] = { blocks[i].wd, blocks[i].s2, blocks[i].b2, blocks[i].wp_bf16, blocks[i].s3, blocks[i].b3 };
        for (const void *p : ptrs)
            if (!p || ((uintptr_t)p % 16)) return p ? MBN_EUNSUPPORTED : MBN_EINVAL;
    }
    TailArgs a;
    a.out = (__bf16 *)out; a.in = (const __bf16 *)in; a.batch = batch; a.h0 = rows; a.w0 = cols;
    a.wd0 = (const float *)blocks[0].wd; a.s0 = (const float *)blocks[0].s2; a.b0 = (const float *)blocks[0].b2;
    a.wp0 = (const __bf16 *)blocks[0].wp_bf16; a.s1 = (const float *)blocks[0].s3; a.b1 = (const float *)blocks[0].b3;
    a.wd1 = (const float *)blocks[1].wd; a.s2 = (const float *)blocks[1].s2; a.b2 = (const float *)blocks[1].b2;
    a.wp1 = (const __bf16 *)blocks[1].wp_bf16; a.s3 = (const float *)blocks[1].s3; a.b3 = (const float *)blocks[1].b3;
    a.dbg = g_mbn_tune.exp0 >= 800 && g_mbn_tune.exp0 < 832 ? g_mbn_tune.exp0 - 800 : 0;
    long grid = ctx->num_cus;
    if (grid > (batch + 1) / 2) grid = (batch + 1) / 2;                      // two images per pass
    hipLaunchKernelGGL((tail_bf16<256, 512>), dim3((unsigned)grid), dim3(512), 0, stream, a);
    return MBN_OK;
}
