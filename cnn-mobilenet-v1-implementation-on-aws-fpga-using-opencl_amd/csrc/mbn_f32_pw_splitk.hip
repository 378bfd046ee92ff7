// mbn_f32_pw_splitk.hip — pointwise 1x1 / FC for the FEW-TILE regime (batch 1..4: M = 49..784 output pixels), fp32.
// Replaces the OpenCL `pointwise` kernel (kernel.cl:94-114; MobileNet.c:1218-2576 call sites) and the FC use of it at
// the batch sizes of BASELINE configs[1] (one image).
//
// Why a second GEMM: at batch 1 the 512 -> 512 layers are 4 x 8 = 32 tiles of 64x64 on a 256-CU chip, and each wave of
// pw_gemm walks the WHOLE K as one dependent chain of K/2 v_mfma_f32_32x32x2 (64 cycles each): K = 512 is 16 k cycles =
// 7.5 us of pure issue latency with 7/8 of the chip idle (measured 14-22 us per layer). No load pipeline fixes
// that; only more independent accumulation chains do. Here:
//   * one workgroup = one 16x16 output tile (v_mfma_f32_16x16x4_f32, 32 cycles) — 13 x 32 = 416 workgroups on the
//     14x14x512 layers, 4 x 64 = 256 on the 7x7x1024 ones;
//   * the S waves of a workgroup (4, 8 or 16) each take K/S of the reduction: their operand loads are all issued before
//     the first MFMA (16-byte loads straight from global memory into MFMA operand registers — at these sizes A and the
//     filter live in L2/Infinity Cache and there is nothing to stage), so a wave's chain is K/(4S) MFMAs = 1-2 k cycles;
//   * the partial tiles meet in LDS and are summed in FIXED order ((p0 + p1) + p2) + ... by the first 256 threads, which
//     then apply BN/bias + activation and store 64-byte row segments. No atomics, no workspace, deterministic.
// The k assignment inside a 16-wide k-step is permuted (lane group q of a 16-byte load holds k = 4q..4q+3, MFMA i of the
// step consumes element i of every group), identically for A and B, so it is a reordering of the sum only.
//
// Summation order differs from pw_gemm's (sequential over K), so results differ from the large-batch kernel in the last
// bits (both are within 2e-6 of the oracle on the network's layers); the regime is "at most 4 images per call AND
// a layer whose 4-image form has fewer 64x64 tiles than half the CUs" and the split S depends on K alone, so a layer takes
// the same kernel and the same summation order for every batch of 1..4 images, and pw_gemm for every batch of 5 or more.
// `forward(n)[:k] == forward(k)` bit-identity therefore holds for n, k <= 4 and for n, k >= 5 (and the shard property of the
// multi-GPU path for shards of 5 images or more), not between a batch of 1-4 images and a larger one; the net runner does
// not fork sub-batches of fewer than 5 images (mbn_net_set_streams).
// A/B hook: mbn_tune_set("pw_splitk", 1) disables this kernel, 2 forces it wherever the shape allows.
#include "mbn_internal.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));

struct SkArgs {
    float *out;
    const float *in, *filt, *scale, *shift;
    int m, k, n, act;
    int nt;        // 16-column tiles
    int kq;        // k per wave (multiple of 16)
};

// TB x TB blocks of 16x16 per workgroup tile (1: 16x16, 2: 32x32 — same sums, half the operand re-reads; for the larger M of the
// regime); CH = 16-wide k-steps held in registers at once (CH * 2 * TB float4 per lane); the launcher picks CH | kq / 16.
// Per-element summation order is the same for every TB and CH (it is fixed by S and the k map above).
template <int S, int CH, int TB>
__global__ __launch_bounds__(64 * S) void pw_splitk_f32(SkArgs a)
{
    constexpr int TW = 16 * TB, TE = TW * TW;          // tile width, elements per tile
    __shared__ float red[S][TE];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int tile = blockIdx.x;
    const int row0 = (tile / a.nt) * TW, col0 = (tile % a.nt) * TW;
    const int li = lane & 15, q = lane >> 4;
    // rows/columns past the matrix read the last valid one (finite data, result discarded at the store)
    const float *ap[TB], *bp[TB];
#pragma unroll
    for (int t = 0; t < TB; t++) {
        ap[t] = a.in + (size_t)min(row0 + 16 * t + li, a.m - 1) * a.k + w * a.kq + 4 * q;
        bp[t] = a.filt + (size_t)min(col0 + 16 * t + li, a.n - 1) * a.k + w * a.kq + 4 * q;
    }
    f4 acc[TB][TB];
#pragma unroll
    for (int i = 0; i < TB; i++)
#pragma unroll
        for (int j = 0; j < TB; j++) acc[i][j] = f4{ 0.f, 0.f, 0.f, 0.f };
    const int steps = a.kq / 16;
    for (int s0 = 0; s0 < steps; s0 += CH) {
        f4 av[CH][TB], bv[CH][TB];
#pragma unroll
        for (int j = 0; j < CH; j++)
#pragma unroll
            for (int t = 0; t < TB; t++) {
                av[j][t] = *reinterpret_cast<const f4 *>(ap[t] + (s0 + j) * 16);
                bv[j][t] = *reinterpret_cast<const f4 *>(bp[t] + (s0 + j) * 16);
            }
        __builtin_amdgcn_sched_barrier(0);      // all loads in flight before the first MFMA (the scheduler sinks them otherwise)
#pragma unroll
        for (int j = 0; j < CH; j++)
#pragma unroll
            for (int e = 0; e < 4; e++)
#pragma unroll
                for (int ti = 0; ti < TB; ti++)
#pragma unroll
                    for (int tj = 0; tj < TB; tj++)
                        acc[ti][tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j][ti][e], bv[j][tj][e], acc[ti][tj], 0, 0, 0);
    }
    // C/D layout of 16x16x4: lane holds column li, rows 4q + r. LDS tile in [row][col] order: conflict-free writes.
#pragma unroll
    for (int ti = 0; ti < TB; ti++)
#pragma unroll
        for (int tj = 0; tj < TB; tj++)
#pragma unroll
            for (int r = 0; r < 4; r++) red[w][(16 * ti + 4 * q + r) * TW + 16 * tj + li] = acc[ti][tj][r];
    __syncthreads();
    for (int e = tid; e < TE; e += 64 * S) {
        float v = red[0][e];
#pragma unroll
        for (int i = 1; i < S; i++) v += red[i][e];
        const int row = row0 + e / TW, col = col0 + e % TW;
        if (row < a.m && col < a.n) {
            v = fmaf(v, a.scale ? a.scale[col] : 1.f, a.shift ? a.shift[col] : 0.f);
            if (a.act == MBN_ACT_RELU6) v = fminf(fmaxf(v, 0.f), 6.f);
            else if (a.act == MBN_ACT_RELU) v = fmaxf(v, 0.f);
            a.out[(size_t)row * a.n + col] = v;
        }
    }
}

template <int S, int TB>
void launch_s(const SkArgs &a, unsigned grid, hipStream_t st)
{
    const int steps = a.kq / 16;
    constexpr int CMAX = 8 / TB;                 // at most 16 float4 loads (64 VGPRs) in flight per lane
    if (steps % CMAX == 0) hipLaunchKernelGGL((pw_splitk_f32<S, CMAX, TB>), dim3(grid), dim3(64 * S), 0, st, a);
    else if (steps % 4 == 0) hipLaunchKernelGGL((pw_splitk_f32<S, 4, TB>), dim3(grid), dim3(64 * S), 0, st, a);
    else if (steps % 2 == 0) hipLaunchKernelGGL((pw_splitk_f32<S, 2, TB>), dim3(grid), dim3(64 * S), 0, st, a);
    else hipLaunchKernelGGL((pw_splitk_f32<S, 1, TB>), dim3(grid), dim3(64 * S), 0, st, a);
}

}   // namespace

// MBN_OK if launched; MBN_EUNSUPPORTED if the shape is outside this kernel's regime (the caller then takes pw_gemm).
int mbn_launch_f32_pw_splitk(const mbn_call &c, float *out, const float *in, const float *filt, long m, int cin, int op_size)
{
    const int mode = g_mbn_tune.pw_splitk;
    if (mode == 1 || c.dtype != MBN_DT_F32) return MBN_EUNSUPPORTED;
    if (cin < 128 || (cin % 64) != 0 || m <= 0 || m > 65536) return MBN_EUNSUPPORTED;
    if (((uintptr_t)in % 16) || ((uintptr_t)filt % 16)) return MBN_EUNSUPPORTED;
    // The regime is a function of the LAYER SHAPE and of "at most 4 images per call" only — evaluated as if the call held 4
    // images — so every batch of 1..4 images takes the same kernel for a given layer, and S below depends on K alone:
    // forward(n)[:k] == forward(k) stays bit-exact within 1..4 images as it is within 5 and more.
    // Rule: pw_gemm's 64x64 tiles of a 4-image call would cover less than half of the CUs.
    // One pixel per image (the FC layer, MobileNet.c:2681-2763; pointwise layers on 1 x 1 maps): this kernel at EVERY batch. Measured
    // (profiles/r03/v_fc_splitk.txt): 22 -> 7-13 us for 8 ... 256 images (pw_gemm's 64 x 64 tiles are 16 ... 64 workgroups walking all of K),
    // equal at 512, slower from 1024 up (+15 us on a step of > 10 ms) — and the logits' summation order no longer depends on the batch at all.
    const bool one_px = c.batch >= 1 && m == (long)c.batch;
    if (mode != 2 && mode != 16 && mode != 32 && !one_px) {            // 2 / 16 / 32: wherever the shape allows (16, 32: with that workgroup tile forced — tests)
        if (c.batch < 1 || c.batch > 4 || m % c.batch) return MBN_EUNSUPPORTED;
        const long m4 = m / c.batch * 4;
        if (((m4 + 63) / 64) * ((op_size + 63) / 64) * 2 > c.ctx->num_cus) return MBN_EUNSUPPORTED;
    }
    // waves per workgroup = K slices: 64 k per wave where K allows (K/S must be a multiple of 16)
    const int S = (cin >= 1024 && cin % 256 == 0) ? 16 : (cin >= 512 && cin % 128 == 0) ? 8 : 4;
    // tile: 16x16 while that gives at most ~2 workgroups per CU, 32x32 beyond (same sums; halves the L2 traffic of the
    // operand re-reads, which is what bounds the 3-4 image calls: 512 -> 512 at 4 images 18 -> 10 us)
    const int tb = mode == 16 ? 1 : mode == 32 ? 2 : (((m + 15) / 16) * ((op_size + 15) / 16) > 2L * c.ctx->num_cus ? 2 : 1);
    const long mt = (m + 16 * tb - 1) / (16 * tb), nt = (op_size + 16 * tb - 1) / (16 * tb);
    if (mt * nt > 1L << 20) return MBN_EUNSUPPORTED;
    SkArgs a;
    a.out = out; a.in = in; a.filt = filt; a.scale = c.scale; a.shift = c.shift;
    a.m = (int)m; a.k = cin; a.n = op_size; a.act = c.act; a.nt = (int)nt; a.kq = cin / S;
    const unsigned grid = (unsigned)(mt * nt);
    if (tb == 1) {
        if (S == 4) launch_s<4, 1>(a, grid, c.stream);
        else if (S == 8) launch_s<8, 1>(a, grid, c.stream);
        else launch_s<16, 1>(a, grid, c.stream);
    } else {
        if (S == 4) launch_s<4, 2>(a, grid, c.stream);
        else if (S == 8) launch_s<8, 2>(a, grid, c.stream);
        else launch_s<16, 2>(a, grid, c.stream);
    }
    return MBN_OK;
}
