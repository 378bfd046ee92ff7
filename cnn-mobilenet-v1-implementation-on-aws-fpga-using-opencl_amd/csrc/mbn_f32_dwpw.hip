// mbn_f32_dwpw.hip — fused depthwise 3x3 -> pointwise 1x1 block for gfx950, fp32: one launch
// replaces a `depthwise` + `pointwise` kernel pair of the reference's sequence (kernel.cl:62-92 + 94-114, the pairs
// L4-5 ... L26-27 of MobileNet.c:322-2599), each with folded-BN scale/shift + ReLU6. The depthwise output is never
// written to HBM: it is produced 32 channels at a time directly into the LDS A tile of the pointwise GEMM.
//
//   out[m][n] = relu6( s3[n] * sum_c relu6( s2[c] * sum_{dy,dx} in[pix(m)+(dy,dx)][c] * wd[dy][dx][c] + b2[c] ) * wp[n][c] + b3[n] )
//
// Workgroup = 16 waves, tile = 128 consecutive output pixels (flattened n,y,x) x BN output channels, persistent over
// tiles, K loop over 32-channel chunks. The waves are SPECIALISED (wave w runs on SIMD w % 4: every SIMD hosts two
// consumers and two producers, 128 VGPRs each):
//   * waves 8-15, PRODUCERS (VALU + memory): a lane (pair = t>>3, c4 = t&7) owns 2 horizontally adjacent output
//     pixels x 4 channels. Its 3 x (S+3) input float4s per chunk come from buffer_load_dwordx4 with
//     per-tile precomputed byte offsets; taps outside the image carry an out-of-range offset, for which the buffer
//     unit returns 0 — the zero padding costs no VALU and no branches. They compute depthwise + BN + ReLU6 for chunk
//     g+1 into the swizzled A tile of the other LDS buffer and bring the pointwise filter chunk [BN][32] in with
//     global_load_lds_dwordx4 while
//   * waves 0-7, CONSUMERS (MFMA only): multiply chunk g — v_mfma_f32_32x32x2_f32 on a 64 x 64 (BN = 256) or
//     32 x 64 (BN = 128) wave tile, LDS image and swizzle identical to mbn_f32_pw.hip.
// One s_barrier per chunk hands buffers over; the flattened (tile, chunk) sequence is pipelined across tile
// boundaries, and a tile's epilogue stores are issued after the barrier so the producers already work on the next
// chunk. The producers' load latency and VALU time hide under the consumers' MFMA time (a chunk is 8192 MFMA cycles per
// SIMD at BN = 256 against ~1000 producer VALU cycles). The arithmetic order of both stages equals the unfused
// kernels', so the result is bit-identical to depthwise-then-pointwise.
// The depthwise filter [9][Cin] and its scale/shift stay in LDS for the workgroup's lifetime (Cin <= 1024).
// When Cout > BN the depthwise work is recomputed per column tile (n-tile index fastest, so the re-reads hit L2).
#include "mbn_internal.h"
#include "mbn_epilogue.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef mbn_f16v f16v;
typedef unsigned u4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BKF = 32;
constexpr int NCW = 8, NPW = 8;                // consumer / producer waves: waves land on SIMD (wave % 4), so every SIMD
constexpr int NT = 64 * (NCW + NPW);           // hosts two MFMA-issuing waves and two VALU/memory waves (128 VGPRs each)
constexpr int CMAX = 1024;                     // largest Cin (depthwise constants resident in LDS: 44 KB)
constexpr unsigned OOB = 0xF0000000u;          // byte offset beyond any supported tensor: the load returns zeros

struct DwPwArgs {
    float *out;
    const float *in, *wd, *s2, *b2, *wp, *s3, *b3;
    long m;                 // output pixels = batch * ho * wo
    int h, w, ho, wo;       // input / output map sides
    int cin, cout;
    int pad_top, pad_left;
    int mt, nt;
    unsigned in_bytes;
    unsigned wo_m, wo_s, ho_m, ho_s;   // floor(v / wo) = umulhi(v, wo_m) >> wo_s for v < 2^31 (m == 0: the divisor is 1)
};

__device__ __forceinline__ int swz(int row, int chunk) { return (row << 5) + (((chunk ^ (row >> 1)) & 7) << 2); }
__device__ __forceinline__ float relu6(float v) { return fminf(fmaxf(v, 0.f), 6.f); }
__device__ __forceinline__ f4 bn_relu6(f4 a, f4 s, f4 b)
{
    return f4{ relu6(fmaf(a.x, s.x, b.x)), relu6(fmaf(a.y, s.y, b.y)), relu6(fmaf(a.z, s.z, b.z)), relu6(fmaf(a.w, s.w, b.w)) };
}
__device__ __forceinline__ int xcd_remap(int vb, int nwg)
{
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = vb & 7;
    return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (vb >> 3);
}

template <int S, int BN>
__global__ __launch_bounds__(NT) void dwpw_f32(DwPwArgs a)
{
    constexpr int WN = 64, WM = BN == 256 ? 64 : 32;   // consumer wave tile: 8 waves as 2 x 4 (BN 256) or 4 x 2 (BN 128)
    constexpr int WAVES_N = BN / WN;
    static_assert((BM / WM) * WAVES_N == NCW, "8 consumer waves");
    constexpr int MI = WM / 32, NI = WN / 32;
    constexpr int NP = 64 * NPW;                       // producer lanes
    constexpr int B_LD = BN * 8 / NP;                  // 16-B filter pieces per producer lane per chunk
    constexpr int XC = S + 3;                          // input columns feeding 2 adjacent output pixels
    constexpr int NX = 3 * XC;                         // buffer loads per producer lane per chunk
    // one LDS object, carved by hand (as in mbn_f32_pw.hip; with separate arrays hipcc puts s_waitcnt vmcnt(0) in front
    // of the first MFMA operand read after every direct-to-LDS load)
    __shared__ __attribute__((aligned(16))) float lds[2 * BM * BKF + 2 * BN * BKF + 11 * CMAX];
    float *const a_s0 = lds, *const b_s0 = lds + 2 * BM * BKF, *const wd_s = b_s0 + 2 * BN * BKF, *const sb_s = wd_s + 9 * CMAX;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nk = a.cin / 32, nwg = a.mt * a.nt;
    const unsigned mtot = (unsigned)a.m;

    for (int i = tid * 4; i < 9 * a.cin; i += NT * 4) *reinterpret_cast<f4 *>(wd_s + i) = *reinterpret_cast<const f4 *>(a.wd + i);
    for (int i = tid * 4; i < a.cin; i += NT * 4) {
        *reinterpret_cast<f4 *>(sb_s + i) = *reinterpret_cast<const f4 *>(a.s2 + i);
        *reinterpret_cast<f4 *>(sb_s + a.cin + i) = *reinterpret_cast<const f4 *>(a.b2 + i);
    }
    __syncthreads();
    if ((int)blockIdx.x >= nwg) return;

    if (wave_u < NPW) {          // producers are the OLDER waves: VALU issue between co-resident waves is arbitrated by age
        // =============================================================== PRODUCERS
        const int pw_u = wave_u;                                         // producer wave 0..7
        const int t = pw_u * 64 + lane, c4 = t & 7, pair = t >> 3;              // pair 0..63: tile rows 2*pair, 2*pair+1
        const __amdgpu_buffer_rsrc_t rsrc = mbn_make_rsrc(a.in, a.in_bytes);
        unsigned off[3][XC];
        const float *b_src[B_LD];
        auto set_tile = [&](int v) __attribute__((always_inline)) {
            const int lid = xcd_remap(v, nwg);
            const int n0 = (lid % a.nt) * BN;
            const unsigned m = (unsigned)(lid / a.nt) * BM + 2 * pair;
            const bool mok = m < mtot;
            // (n, y, x) of the pixel by multiply-high division (host-computed magic numbers), then every tap offset as
            // base + dy * row stride + j * column stride: this runs once per tile per lane and used to cost ~400 VALU
            // instructions (two 32-bit divisions + 12-15 independent offset computations)
            const unsigned q = a.wo_m ? __umulhi(m, a.wo_m) >> a.wo_s : m;
            const unsigned x = m - q * (unsigned)a.wo;
            const unsigned n = a.ho_m ? __umulhi(q, a.ho_m) >> a.ho_s : q;
            const unsigned y = q - n * (unsigned)a.ho;
            const int iy0 = (int)y * S - a.pad_top, ix0 = (int)x * S - a.pad_left;
            const unsigned cs = (unsigned)a.cin * 4u, rs = (unsigned)a.w * cs;                    // column / row stride in bytes
            const unsigned base = ((n * a.h + iy0) * a.w + ix0) * cs + (unsigned)(c4 * 4) * 4u;         // wraps for taps that are masked out below
#pragma unroll
            for (int dy = 0; dy < 3; dy++) {
                const bool rok = mok && (unsigned)(iy0 + dy) < (unsigned)a.h;
#pragma unroll
                for (int j = 0; j < XC; j++) {
                    const bool ok = rok && (unsigned)(ix0 + j) < (unsigned)a.w;
                    off[dy][j] = ok ? base + dy * rs + j * cs : OOB;
                }
            }
#pragma unroll
            for (int p = 0; p < B_LD; p++) {
                const int row = (p * NP + t) >> 3;
                b_src[p] = a.wp + (long)(n0 + row) * a.cin + ((c4 ^ (row >> 1)) & 7) * 4;
            }
        };
        f4 xr[3][XC];
        auto ldx = [&](int kc) __attribute__((always_inline)) {
#pragma unroll
            for (int dy = 0; dy < 3; dy++)
#pragma unroll
                for (int j = 0; j < XC; j++)
                    xr[dy][j] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off[dy][j], kc * 128, 0));
        };
        auto glds_b = [&](int kc, int buf) __attribute__((always_inline)) {
#pragma unroll
            for (int p = 0; p < B_LD; p++)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(b_src[p] + kc * 32),
                                                 (__attribute__((address_space(3))) void *)(b_s0 + buf * BN * BKF + (p * (NP / 8) + pw_u * 8) * BKF),
                                                 16, 0, 0);
        };
        auto dw = [&](int kc, int buf) __attribute__((always_inline)) {
            const float *wk = wd_s + kc * 32 + c4 * 4;
            f4 acc0 = f4{ 0.f, 0.f, 0.f, 0.f }, acc1 = acc0;
#pragma unroll
            for (int dy = 0; dy < 3; dy++)
#pragma unroll
                for (int dx = 0; dx < 3; dx++) {
                    const f4 w = *reinterpret_cast<const f4 *>(wk + (dy * 3 + dx) * a.cin);
                    acc0 = __builtin_elementwise_fma(xr[dy][dx], w, acc0);
                    acc1 = __builtin_elementwise_fma(xr[dy][dx + S], w, acc1);
                }
            const f4 s = *reinterpret_cast<const f4 *>(sb_s + kc * 32 + c4 * 4);
            const f4 b = *reinterpret_cast<const f4 *>(sb_s + a.cin + kc * 32 + c4 * 4);
            *reinterpret_cast<f4 *>(a_s0 + buf * BM * BKF + swz(2 * pair, c4)) = bn_relu6(acc0, s, b);
            *reinterpret_cast<f4 *>(a_s0 + buf * BM * BKF + swz(2 * pair + 1, c4)) = bn_relu6(acc1, s, b);
        };

        // cursor = the chunk whose input loads are in flight: (tile cvb, chunk ckc)
        int cvb = blockIdx.x, ckc = 0;
        auto advance = [&]() __attribute__((always_inline)) -> bool {                 // next chunk of the flattened sequence; false at the end
            if (++ckc < nk) return true;
            ckc = 0;
            cvb += gridDim.x;
            if (cvb >= nwg) return false;
            set_tile(cvb);
            return true;
        };
        set_tile(cvb);
        ldx(0);
        glds_b(0, 0);
        dw(0, 0);
        bool have = advance();
        if (have) {
            ldx(ckc);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NX) : "memory");     // filter chunk 0 landed; the NX newer loads may fly
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();                                                  // chunk 0 handed to the consumers
        int p = 0;
        while (have) {
            // consumers multiply the chunk in buffer p; produce the cursor's chunk into buffer p^1 (free since the barrier)
            const int kc = ckc;
            glds_b(kc, p ^ 1);
            dw(kc, p ^ 1);                                                // waits for this chunk's buffer loads only
            have = advance();
            if (have) {
                ldx(ckc);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NX) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();
            p ^= 1;
        }
        __syncthreads();                                                  // the consumers' last chunk
        return;
    }

    // =================================================================== CONSUMERS
    const int cw_u = wave_u - NPW;                                        // consumer wave 0..7
    const int wm = (cw_u / WAVES_N) * WM, wn = (cw_u % WAVES_N) * WN;
    const int li = lane & 31, lh = lane >> 5;
    const __amdgpu_buffer_rsrc_t orsrc = mbn_make_rsrc(a.out, (unsigned)(a.m * a.cout * 4));
    __syncthreads();                                                      // chunk 0 is in buffer 0
    int p = 0;
    for (int vb = blockIdx.x; vb < nwg; vb += gridDim.x) {
        const int lid = xcd_remap(vb, nwg);
        const int n0 = (lid % a.nt) * BN;
        const unsigned m0 = (unsigned)(lid / a.nt) * BM;
        f16v acc[MI][NI];
#pragma unroll
        for (int mi = 0; mi < MI; mi++)
#pragma unroll
            for (int ni = 0; ni < NI; ni++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[mi][ni][r] = 0.f;
        for (int kc = 0; kc < nk; kc++) {
            const float *As = a_s0 + p * BM * BKF, *Bs = b_s0 + p * BN * BKF;
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
            __syncthreads();                                              // buffer p may be refilled; buffer p^1 is ready
            p ^= 1;
        }
        // ---- epilogue after the hand-over barrier: BN + ReLU6 (mbn_epilogue.h); only the last row tile can be ragged
        if (m0 + BM <= mtot) mbn_store_relu6_f32<MI, NI, 0>(orsrc, (unsigned)a.cout, m0 + wm, n0 + wn, lane, acc, a.s3, a.b3, mtot, a.cout);
        else mbn_store_relu6_f32<MI, NI, 1>(orsrc, (unsigned)a.cout, m0 + wm, n0 + wn, lane, acc, a.s3, a.b3, mtot, a.cout);
    }
}

template <int S, int BN>
void launch(DwPwArgs &a, hipStream_t s, int num_cus)
{
    a.mt = (int)((a.m + BM - 1) / BM);
    a.nt = a.cout / BN;
    const long nwg = (long)a.mt * a.nt;
    const int per_cu = g_mbn_tune.misc;
    long grid = (long)num_cus * (per_cu > 0 ? per_cu : 1);
    if (grid > nwg) grid = nwg;
    hipLaunchKernelGGL((dwpw_f32<S, BN>), dim3((unsigned)grid), dim3(NT), 0, s, a);
}

}   // namespace

// Shape/alignment envelope of the fused block kernel: MBN_OK, MBN_EUNSUPPORTED (run the two layers separately) or
// MBN_EINVAL (null pointer).
int mbn_f32_dwpw_check(const float *out, const float *in, const float *wd, const float *s2, const float *b2,
                       const float *wp, const float *s3, const float *b3, int batch, int in_rows, int in_cols,
                       int out_rows, int out_cols, int cin, int cout, int stride, int pad_top, int pad_left)
{
    const float *ptrs[] = { in, wd, s2, b2, wp, s3, b3, out };
    for (const float *p : ptrs)
        if (!p) return MBN_EINVAL;
    if (batch <= 0 || (stride != 1 && stride != 2) || cin < 32 || (cin % 32) != 0 || cin > CMAX || cout < 128 || cout > 1024 ||
        (cout % 128) != 0 || (out_cols & 1) || out_rows <= 0 || out_cols <= 0 || in_rows <= 0 || in_cols <= 0 ||
        pad_top < 0 || pad_left < 0)
        return MBN_EUNSUPPORTED;
    if (4.0 * batch * in_rows * in_cols * cin >= (double)OOB) return MBN_EUNSUPPORTED;
    if ((long)batch * out_rows * out_cols > 0x7fffff00L) return MBN_EUNSUPPORTED;                 // 32-bit pixel index
    if (4.0 * ((double)batch * out_rows * out_cols + 256.0) * cout >= 4294967296.0) return MBN_EUNSUPPORTED;          // buffer stores; + a row tile of head room: ragged rows must not wrap (32-bit offsets)
    for (const float *p : ptrs)
        if (((uintptr_t)p % 16) != 0) return MBN_EUNSUPPORTED;
    return MBN_OK;
}

// Fused depthwise 3x3 (stride 1 or 2, zero padding pad_top/pad_left, none needed explicitly on the high side) ->
// pointwise 1x1, both with BN scale/shift + ReLU6.
int mbn_launch_f32_dwpw(mbn_context *ctx, hipStream_t stream, float *out, const float *in, const float *wd,
                        const float *s2, const float *b2, const float *wp, const float *s3, const float *b3, int batch,
                        int in_rows, int in_cols, int out_rows, int out_cols, int cin, int cout, int stride, int pad_top,
                        int pad_left)
{
    const int rc = mbn_f32_dwpw_check(out, in, wd, s2, b2, wp, s3, b3, batch, in_rows, in_cols, out_rows, out_cols, cin,
                                      cout, stride, pad_top, pad_left);
    if (rc != MBN_OK) return rc;
    DwPwArgs a;
    a.out = out; a.in = in; a.wd = wd; a.s2 = s2; a.b2 = b2; a.wp = wp; a.s3 = s3; a.b3 = b3;
    a.m = (long)batch * out_rows * out_cols;
    a.h = in_rows; a.w = in_cols; a.ho = out_rows; a.wo = out_cols;
    a.cin = cin; a.cout = cout; a.pad_top = pad_top; a.pad_left = pad_left;
    mbn_udiv_magic((unsigned)out_cols, &a.wo_m, &a.wo_s);
    mbn_udiv_magic((unsigned)out_rows, &a.ho_m, &a.ho_s);
    a.in_bytes = (unsigned)(4.0 * batch * in_rows * in_cols * cin);
    const bool wide = (cout % 256) == 0 && g_mbn_tune.pw_tile != 1;      // pw_tile=1: force the 128-column tile (A/B hook)
    if (stride == 1) {
        if (wide) launch<1, 256>(a, stream, ctx->num_cus);
        else launch<1, 128>(a, stream, ctx->num_cus);
    } else {
        if (wide) launch<2, 256>(a, stream, ctx->num_cus);
        else launch<2, 128>(a, stream, ctx->num_cus);
    }
    return MBN_OK;
}
