// mbn_f32_pw.hip — fp32 1x1 pointwise conv (and the FC layer) as an MFMA GEMM for gfx950, with the folded-BN
// scale/shift + ReLU/ReLU6 epilogue fused. Replaces the arithmetic of the reference's `pointwise` kernel
// (kernel.cl:94-114; reused as FC at MobileNet.c:2681-2763) in the fp32 mode the metric measures.
//
//   out[m][n] = act( scale[n] * sum_k in[m][k] * filt[n][k] + shift[n] ),   m = pixel (N*H*W), n = out channel
//
// NHWC makes `in` a row-major [M][K] matrix and `out` a row-major [M][Cout] matrix with no data movement;
// `filt` keeps kernel.cl's own order [oc][ic] = [Cout][K]. Both operands are therefore K-contiguous ("NT" GEMM).
//
// MFMA: v_mfma_f32_32x32x2_f32 — exact fp32 (bit-identical to an fmaf chain over k), 64 FLOP/clk/SIMD, the fp32
// matrix peak of 157.3 TFLOP/s. Lane l feeds A[i=l&31][k=l>>5] and B[k=l>>5][j=l&31], one float each.
// K order inside an 8-wide k-group is permuted so that ONE ds_read_b128 per lane feeds four MFMAs: lane half
// h=l>>5 owns k = 8g+4h .. 8g+4h+3 and MFMA step s consumes element s of both operands' float4 (the same k on
// both sides, so the sum over k is unchanged up to fp32 summation order).
//
// LDS: tiles [rows][32 floats] (128-B rows) staged through registers with 16-B global loads; the 16-B chunk index
// is XOR-swizzled with (row>>1)&7, which makes the ds_read_b128 of 16 consecutive rows hit 16 distinct 16-B
// slots of the 256-B bank row (conflict-free for the b128 lane groups) and keeps ds_write_b128 conflict-free.
// Double-buffered over K with one barrier per 32-deep k-tile; next tile's global loads are issued before the
// MFMAs of the current one and written to LDS after them.
//
// Grid: 1-D, remapped so that the workgroups sharing one A row-panel (all n-tiles of an m-tile) are consecutive
// on ONE XCD (blocks b and b+8 share an XCD): the A panel is fetched from HBM once and re-read from that XCD's L2.
#include "mbn_internal.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

struct PwArgs {
    float *out;
    const float *in, *filt, *scale, *shift;
    long m;
    int k, n, act;
    int mt, nt;   // tile counts
};

constexpr int BK = 32;

__device__ __forceinline__ int swz(int row, int chunk) { return (row << 5) + (((chunk ^ (row >> 1)) & 7) << 2); }

template <int BM, int BN, int WM, int WN, int NBUF, bool KFULL>
__global__ __launch_bounds__(64 * (BM / WM) * (BN / WN)) void pw_gemm_f32(PwArgs a)
{
    constexpr int WAVES_N = BN / WN;
    constexpr int NT = 64 * (BM / WM) * WAVES_N;              // threads per workgroup
    constexpr int MI = WM / 32, NI = WN / 32;
    constexpr int A_LD = BM * 8 / NT, B_LD = BN * 8 / NT;     // float4 loads per thread per k-tile
    static_assert(A_LD >= 1 && B_LD >= 1 && A_LD * NT == BM * 8 && B_LD * NT == BN * 8, "tile/threads mismatch");
    __shared__ __attribute__((aligned(16))) float lds[NBUF * (BM + BN) * BK];

    // XCD-aware bijective remap (cdna guide T1): same-XCD blocks get consecutive logical ids.
    const int nwg = a.mt * a.nt;
    const int bid = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int tn = lid % a.nt, tm = lid / a.nt;
    const long m0 = (long)tm * BM;
    const int n0 = tn * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave / WAVES_N) * WM, wn = (wave % WAVES_N) * WN;
    const int li = lane & 31, lh = lane >> 5;

    f16v acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; mi++)
#pragma unroll
        for (int ni = 0; ni < NI; ni++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[mi][ni][r] = 0.f;

    // per-thread staging coordinates (row, 16-B chunk) — fixed for the whole K loop
    const float *a_src[A_LD];
    const float *b_src[B_LD];
    int a_dst[A_LD], b_dst[B_LD], a_kc[A_LD], b_kc[B_LD];
#pragma unroll
    for (int p = 0; p < A_LD; p++) {
        int idx = p * NT + tid, row = idx >> 3, ch = idx & 7;
        long gm = m0 + row;
        if (gm >= a.m) gm = a.m - 1;                 // clamp: rows past M are computed but never stored
        a_src[p] = a.in + gm * a.k + ch * 4;
        a_dst[p] = swz(row, ch);
        a_kc[p] = ch * 4;
    }
#pragma unroll
    for (int p = 0; p < B_LD; p++) {
        int idx = p * NT + tid, row = idx >> 3, ch = idx & 7;
        int gn = n0 + row;
        if (gn >= a.n) gn = a.n - 1;
        b_src[p] = a.filt + (long)gn * a.k + ch * 4;
        b_dst[p] = BM * BK + swz(row, ch);
        b_kc[p] = ch * 4;
    }

    f4 a_reg[A_LD], b_reg[B_LD];
    const f4 zero4 = f4{ 0.f, 0.f, 0.f, 0.f };
    auto stage_load = [&](int k0) {
#pragma unroll
        for (int p = 0; p < A_LD; p++)
            a_reg[p] = (KFULL || k0 + a_kc[p] < a.k) ? *reinterpret_cast<const f4 *>(a_src[p] + k0) : zero4;
#pragma unroll
        for (int p = 0; p < B_LD; p++)
            b_reg[p] = (KFULL || k0 + b_kc[p] < a.k) ? *reinterpret_cast<const f4 *>(b_src[p] + k0) : zero4;
    };
    auto stage_store = [&](int buf) {
        float *base = lds + buf * (BM + BN) * BK;
#pragma unroll
        for (int p = 0; p < A_LD; p++) *reinterpret_cast<f4 *>(base + a_dst[p]) = a_reg[p];
#pragma unroll
        for (int p = 0; p < B_LD; p++) *reinterpret_cast<f4 *>(base + b_dst[p]) = b_reg[p];
    };
    auto compute = [&](int buf) {
        const float *As = lds + buf * (BM + BN) * BK;
        const float *Bs = As + BM * BK;
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const int chunk = 2 * g + lh;
            f4 av[MI], bv[NI];
#pragma unroll
            for (int mi = 0; mi < MI; mi++) av[mi] = *reinterpret_cast<const f4 *>(As + swz(wm + mi * 32 + li, chunk));
#pragma unroll
            for (int ni = 0; ni < NI; ni++) bv[ni] = *reinterpret_cast<const f4 *>(Bs + swz(wn + ni * 32 + li, chunk));
#pragma unroll
            for (int s = 0; s < 4; s++)
#pragma unroll
                for (int mi = 0; mi < MI; mi++)
#pragma unroll
                    for (int ni = 0; ni < NI; ni++)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi][s], bv[ni][s], acc[mi][ni], 0, 0, 0);
        }
    };

    const int nk = (a.k + BK - 1) / BK;
    stage_load(0);
    stage_store(0);
    __syncthreads();
    if (NBUF == 1) {
        for (int kt = 0; kt < nk; kt++) {
            if (kt + 1 < nk) stage_load((kt + 1) * BK);
            compute(0);
            if (kt + 1 < nk) {
                __syncthreads();
                stage_store(0);
                __syncthreads();
            }
        }
    } else {
        for (int kt = 0; kt < nk; kt++) {
            const int cur = kt & (NBUF - 1);
            if (kt + 1 < nk) stage_load((kt + 1) * BK);
            compute(cur);
            if (kt + 1 < nk) stage_store(cur ^ (NBUF - 1));
            __syncthreads();
        }
    }

    // epilogue: C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
    // One store instruction writes two 128-B row segments (full cache lines).
#pragma unroll
    for (int ni = 0; ni < NI; ni++) {
        const int col = n0 + wn + ni * 32 + li;
        const bool cok = col < a.n;
        const int cc = cok ? col : a.n - 1;
        const float sc = a.scale ? a.scale[cc] : 1.f;
        const float sh = a.shift ? a.shift[cc] : 0.f;
#pragma unroll
        for (int mi = 0; mi < MI; mi++) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const long row = m0 + wm + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                float v = fmaf(acc[mi][ni][r], sc, sh);
                if (a.act == MBN_ACT_RELU6) v = fminf(fmaxf(v, 0.f), 6.f);
                else if (a.act == MBN_ACT_RELU) v = fmaxf(v, 0.f);
                if (cok && row < a.m) a.out[row * a.n + col] = v;
            }
        }
    }
}

// Fallback for K not a multiple of 4 or unaligned pointers: one lane per output element.
__global__ __launch_bounds__(256) void pw_generic_f32(PwArgs a)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.m * a.n) return;
    const long row = t / a.n;
    const int col = (int)(t % a.n);
    const float *ip = a.in + row * a.k, *fp = a.filt + (long)col * a.k;
    float acc = 0.f;
    for (int i = 0; i < a.k; i++) acc = fmaf(ip[i], fp[i], acc);
    float v = fmaf(acc, a.scale ? a.scale[col] : 1.f, a.shift ? a.shift[col] : 0.f);
    if (a.act == MBN_ACT_RELU6) v = fminf(fmaxf(v, 0.f), 6.f);
    else if (a.act == MBN_ACT_RELU) v = fmaxf(v, 0.f);
    a.out[t] = v;
}

template <int BM, int BN, int WM, int WN>
void launch_cfg(PwArgs &a, hipStream_t s)
{
    constexpr int NT = 64 * (BM / WM) * (BN / WN);
    a.mt = (int)((a.m + BM - 1) / BM);
    a.nt = (a.n + BN - 1) / BN;
    const dim3 grid((unsigned)(a.mt * a.nt)), block(NT);
    const bool kfull = (a.k % BK) == 0;
    if (a.k <= BK) {
        if (kfull) hipLaunchKernelGGL((pw_gemm_f32<BM, BN, WM, WN, 1, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((pw_gemm_f32<BM, BN, WM, WN, 1, false>), grid, block, 0, s, a);
    } else {
        if (kfull) hipLaunchKernelGGL((pw_gemm_f32<BM, BN, WM, WN, 2, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((pw_gemm_f32<BM, BN, WM, WN, 2, false>), grid, block, 0, s, a);
    }
}

}   // namespace

int mbn_launch_f32_pointwise(const mbn_call &c, float *out, const float *in, const float *filt, long m, int cin,
                             int op_size)
{
    PwArgs a;
    a.out = out; a.in = in; a.filt = filt; a.scale = c.scale; a.shift = c.shift;
    a.m = m; a.k = cin; a.n = op_size; a.act = c.act; a.mt = a.nt = 0;
    if (m <= 0 || (long)((m + 31) / 32) * ((op_size + 31) / 32) > 0x7fffffffL) return MBN_EINVAL;
    const bool fast = (cin % 4) == 0 && ((uintptr_t)in % 16) == 0 && ((uintptr_t)filt % 16) == 0;
    if (!fast) {
        long total = m * op_size;
        hipLaunchKernelGGL(pw_generic_f32, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c.stream, a);
        return MBN_OK;
    }
    // Tile choice measured per layer on MI355X (tools/layer_bench.py --tune pw_tile=1..10, profiles/r01): every
    // shape lands within a few % of each other (the loop is matrix-pipe bound), <128,64> with 3 workgroups per CU is
    // best or tied from K = 256 up, 8 waves of 32x64 win slightly for K <= 256, 64x64 for narrow outputs / small grids.
    int tile = g_mbn_tune.pw_tile;
    if (tile == 0) {
        const long big_tiles = ((m + 127) / 128) * ((op_size + 63) / 64);
        if (big_tiles < 2L * c.ctx->num_cus || op_size <= 64) tile = 3;
        else if (cin <= 256 && op_size >= 128) tile = 5;
        else tile = 2;
    }
    switch (tile) {
    case 1: launch_cfg<128, 128, 64, 64>(a, c.stream); break;   // 4 waves, 64 KB LDS, 2 WG/CU
    case 2: launch_cfg<128, 64, 64, 32>(a, c.stream); break;    // 4 waves
    case 3: launch_cfg<64, 64, 32, 32>(a, c.stream); break;     // 4 waves, small problems
    case 4: launch_cfg<256, 128, 64, 64>(a, c.stream); break;   // 8 waves, 96 KB LDS, 1 WG/CU
    case 5: launch_cfg<128, 128, 32, 64>(a, c.stream); break;   // 8 waves of 32x64, 2 WG/CU
    case 6: launch_cfg<128, 256, 64, 64>(a, c.stream); break;   // 8 waves, 96 KB LDS
    case 7: launch_cfg<64, 128, 32, 64>(a, c.stream); break;    // 4 waves, 48 KB LDS, 3 WG/CU
    case 8: launch_cfg<128, 64, 32, 64>(a, c.stream); break;    // 4 waves of 32x64
    case 9: launch_cfg<256, 64, 64, 64>(a, c.stream); break;    // 4 waves of 64x64, tall
    case 10: launch_cfg<64, 256, 64, 64>(a, c.stream); break;   // 4 waves, wide
    default: return MBN_EINVAL;
    }
    return MBN_OK;
}
