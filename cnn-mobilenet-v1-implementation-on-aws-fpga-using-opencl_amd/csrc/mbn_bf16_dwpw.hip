// mbn_bf16_dwpw.hip — the fused depthwise 3x3 -> pointwise 1x1 block (mbn_f32_dwpw.hip) in the
// network's bf16 mode (BASELINE config 5): activations and the pointwise filter are stored as bf16, all arithmetic is
// fp32, every layer output is rounded to bf16 (RNE) — including the depthwise output, which here only ever exists in
// LDS. Replaces a `depthwise` + `pointwise` launch pair of the reference's sequence (kernel.cl:62-92 + 94-114).
//
// Same structure as the fp32 kernel with three changes that follow from the 2-byte elements:
//   * a K chunk is 64 channels, so A and B rows in LDS are again 128 bytes and the LDS image, swizzle and the
//     direct-to-LDS filter staging are byte-for-byte those of pw_gemm<__bf16>;
//   * 8 PRODUCER waves (the older ones: VALU issue between co-resident waves goes by age): a lane owns 2 adjacent output
//     pixels x 8 channels, fetches its 3 x (S+3) input vectors with 16-byte buffer loads (hardware zero padding), keeps them
//     packed and widens one row at a time (each vector once), does the depthwise + BN + ReLU6 math in fp32 and writes 8 rounded bf16 as one 16-byte LDS store;
//   * 4 CONSUMER waves on v_mfma_f32_32x32x16_bf16 (one MFMA per 16-byte chunk per 32x32 block): with a sixteenth of the
//     fp32 kernel's matrix time per element the block is bound by the producers' VALU work and by HBM, so 12 waves per
//     workgroup (170 VGPRs each) are enough.
// Result vs the two separate bf16 launches: same values up to the summation order of the pointwise (both accumulate the
// same bf16 x bf16 products in fp32), i.e. within the bf16 tolerance of the parity tests, not bit-identical.
#include "mbn_internal.h"
#include "mbn_epilogue.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f8 __attribute__((ext_vector_type(8)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef mbn_f16v f16v;

constexpr int BM = 128, BKF = 32;              // LDS rows are 128 bytes = 32 words = 64 bf16
constexpr int NCW = 4, NPW = 8;                // consumer / producer waves
constexpr int NT = 64 * (NCW + NPW);
constexpr int CMAX = 1024;
constexpr unsigned OOB = 0xF0000000u;

struct BArgs {
    __bf16 *out;
    const __bf16 *in, *wp;
    const float *wd, *s2, *b2, *s3, *b3;
    long m;
    int h, w, ho, wo, cin, cout, pad_top, pad_left, mt, nt;
    unsigned in_bytes;
    unsigned wo_m, wo_s, ho_m, ho_s;   // floor(v / wo) = umulhi(v, wo_m) >> wo_s for v < 2^31 (m == 0: the divisor is 1)
};

__device__ __forceinline__ int swz(int row, int chunk) { return (row << 5) + (((chunk ^ (row >> 1)) & 7) << 2); }
__device__ __forceinline__ int xcd_remap(int vb, int nwg)
{
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = vb & 7;
    return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (vb >> 3);
}
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
__device__ __forceinline__ f8 ld8(const float *p)
{
    const f4 a = *reinterpret_cast<const f4 *>(p), b = *reinterpret_cast<const f4 *>(p + 4);
    return f8{ a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w };
}

template <int S, int BN>
__global__ __launch_bounds__(NT) void dwpw_bf16(BArgs a)
{
    constexpr int WN = 64, WM = BN == 256 ? 128 : 64;  // consumer wave tile: 4 waves as 1 x 4 (BN 256) or 2 x 2 (BN 128)
    constexpr int WAVES_N = BN / WN;
    static_assert((BM / WM) * WAVES_N == NCW, "4 consumer waves");
    constexpr int MI = WM / 32, NI = WN / 32;
    constexpr int NP = 64 * NPW;
    constexpr int B_LD = BN * 8 / NP;                  // 16-B filter pieces per producer lane per chunk
    constexpr int XC = S + 3;
    __shared__ __attribute__((aligned(16))) float lds[2 * BM * BKF + 2 * BN * BKF + 11 * CMAX];
    float *const a_s0 = lds, *const b_s0 = lds + 2 * BM * BKF, *const wd_s = b_s0 + 2 * BN * BKF, *const sb_s = wd_s + 9 * CMAX;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nk = a.cin / 64, nwg = a.mt * a.nt;
    const unsigned mtot = (unsigned)a.m;

    for (int i = tid * 4; i < 9 * a.cin; i += NT * 4) *reinterpret_cast<f4 *>(wd_s + i) = *reinterpret_cast<const f4 *>(a.wd + i);
    for (int i = tid * 4; i < a.cin; i += NT * 4) {
        *reinterpret_cast<f4 *>(sb_s + i) = *reinterpret_cast<const f4 *>(a.s2 + i);
        *reinterpret_cast<f4 *>(sb_s + a.cin + i) = *reinterpret_cast<const f4 *>(a.b2 + i);
    }
    __syncthreads();
    if ((int)blockIdx.x >= nwg) return;

    if (wave_u < NPW) {
        // =============================================================== PRODUCERS (older waves)
        const int t = wave_u * 64 + lane, c8 = t & 7, pair = t >> 3;     // pair 0..63: tile rows 2*pair, 2*pair+1; channels 8*c8..+7
        const __amdgpu_buffer_rsrc_t rsrc = mbn_make_rsrc(a.in, a.in_bytes);
        unsigned off[3][XC];
        const __bf16 *b_src[B_LD];
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
            const unsigned cs = (unsigned)a.cin * 2u, rs = (unsigned)a.w * cs;                    // column / row stride in bytes
            const unsigned base = ((n * a.h + iy0) * a.w + ix0) * cs + (unsigned)(c8 * 8) * 2u;         // wraps for taps that are masked out below
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
                b_src[p] = a.wp + (long)(n0 + row) * a.cin + ((c8 ^ (row >> 1)) & 7) * 8;
            }
        };
        u4v xr[3][XC];                                                    // the window stays packed (4 VGPRs per vector) ...
        auto ldx = [&](int kc) __attribute__((always_inline)) {
#pragma unroll
            for (int dy = 0; dy < 3; dy++)
#pragma unroll
                for (int j = 0; j < XC; j++) xr[dy][j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off[dy][j], kc * 128, 0);
        };
        auto glds_b = [&](int kc, int buf) __attribute__((always_inline)) {
#pragma unroll
            for (int p = 0; p < B_LD; p++)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(b_src[p] + kc * 64),
                                                 (__attribute__((address_space(3))) void *)(b_s0 + buf * BN * BKF + (p * (NP / 8) + wave_u * 8) * BKF),
                                                 16, 0, 0);
        };
        auto dw = [&](int kc, int buf) __attribute__((always_inline)) {
            const float *wk = wd_s + kc * 64 + c8 * 8;
            f8 acc0, acc1;
#pragma unroll
            for (int i = 0; i < 8; i++) acc0[i] = acc1[i] = 0.f;
#pragma unroll
            for (int dy = 0; dy < 3; dy++) {
                f8 row[XC];                                               // ... and is widened one row at a time, each vector once
#pragma unroll
                for (int j = 0; j < XC; j++) row[j] = widen8(xr[dy][j]);
#pragma unroll
                for (int dx = 0; dx < 3; dx++) {
                    const f8 w = ld8(wk + (dy * 3 + dx) * a.cin);
                    acc0 = __builtin_elementwise_fma(row[dx], w, acc0);
                    acc1 = __builtin_elementwise_fma(row[dx + S], w, acc1);
                }
            }
            const f8 s = ld8(sb_s + kc * 64 + c8 * 8), b = ld8(sb_s + a.cin + kc * 64 + c8 * 8);
            bf8 o0, o1;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                o0[i] = (__bf16)fminf(fmaxf(fmaf(acc0[i], s[i], b[i]), 0.f), 6.f);      // the layer output is rounded to bf16 here
                o1[i] = (__bf16)fminf(fmaxf(fmaf(acc1[i], s[i], b[i]), 0.f), 6.f);
            }
            *reinterpret_cast<bf8 *>(a_s0 + buf * BM * BKF + swz(2 * pair, c8)) = o0;
            *reinterpret_cast<bf8 *>(a_s0 + buf * BM * BKF + swz(2 * pair + 1, c8)) = o1;
        };

        int cvb = blockIdx.x, ckc = 0;                 // cursor = the chunk whose input loads are in flight
        auto advance = [&]() __attribute__((always_inline)) -> bool {
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
        if (have) ldx(ckc);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // filter chunk 0 landed (and the window: widened above)
        __syncthreads();                                                  // chunk 0 handed to the consumers
        int p = 0;
        while (have) {
            const int kc = ckc;
            glds_b(kc, p ^ 1);
            dw(kc, p ^ 1);
            have = advance();
            if (have) ldx(ckc);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // our filter pieces landed
            __syncthreads();
            p ^= 1;
        }
        __syncthreads();                                                  // the consumers' last chunk
        return;
    }

    // =================================================================== CONSUMERS (younger waves)
    const int cw_u = wave_u - NPW;
    const int wm = (cw_u / WAVES_N) * WM, wn = (cw_u % WAVES_N) * WN;
    const int li = lane & 31, lh = lane >> 5;
    const __amdgpu_buffer_rsrc_t orsrc = mbn_make_rsrc(a.out, (unsigned)(a.m * a.cout * 2));
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
                const int chunk = 2 * g + lh;                             // lane half h holds k = 16g + 8h .. +7 of the 64-wide chunk
                f4 av[MI], bv[NI];
#pragma unroll
                for (int mi = 0; mi < MI; mi++) av[mi] = *reinterpret_cast<const f4 *>(As + swz(wm + mi * 32 + li, chunk));
#pragma unroll
                for (int ni = 0; ni < NI; ni++) bv[ni] = *reinterpret_cast<const f4 *>(Bs + swz(wn + ni * 32 + li, chunk));
#pragma unroll
                for (int mi = 0; mi < MI; mi++)
#pragma unroll
                    for (int ni = 0; ni < NI; ni++)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, av[mi]), __builtin_bit_cast(bf8, bv[ni]),
                                                                              acc[mi][ni], 0, 0, 0);
            }
            __syncthreads();
            p ^= 1;
        }
        if (m0 + BM <= mtot) mbn_store_relu6_f32<MI, NI, 0, __bf16>(orsrc, (unsigned)a.cout, m0 + wm, n0 + wn, lane, acc, a.s3, a.b3, mtot, a.cout);
        else mbn_store_relu6_f32<MI, NI, 1, __bf16>(orsrc, (unsigned)a.cout, m0 + wm, n0 + wn, lane, acc, a.s3, a.b3, mtot, a.cout);
    }
}

template <int S, int BN>
void launch(BArgs &a, hipStream_t s, int num_cus)
{
    a.mt = (int)((a.m + BM - 1) / BM);
    a.nt = a.cout / BN;
    const long nwg = (long)a.mt * a.nt;
    long grid = num_cus;
    if (grid > nwg) grid = nwg;
    hipLaunchKernelGGL((dwpw_bf16<S, BN>), dim3((unsigned)grid), dim3(NT), 0, s, a);
}

}   // namespace

// Envelope: as the fp32 kernel's, with Cin a multiple of 64 (one K chunk) and bf16 element sizes.
int mbn_bf16_dwpw_check(const void *out, const void *in, const float *wd, const float *s2, const float *b2, const void *wp,
                        const float *s3, const float *b3, int batch, int in_rows, int in_cols, int out_rows, int out_cols,
                        int cin, int cout, int stride, int pad_top, int pad_left)
{
    const void *ptrs[] = { in, wd, s2, b2, wp, s3, b3, out };
    for (const void *p : ptrs)
        if (!p) return MBN_EINVAL;
    if (batch <= 0 || (stride != 1 && stride != 2) || (cin != 32 && (cin < 64 || (cin % 64) != 0)) || cin > CMAX || cout < 64 || cout > 1024 ||
        (cout % 64) != 0 || (out_cols & 1) || out_rows <= 0 || out_cols <= 0 || in_rows <= 0 || in_cols <= 0 ||
        pad_top < 0 || pad_left < 0)
        return MBN_EUNSUPPORTED;
    if (2.0 * batch * in_rows * in_cols * cin >= (double)OOB) return MBN_EUNSUPPORTED;
    if ((long)batch * out_rows * out_cols > 0x7fffff00L) return MBN_EUNSUPPORTED;
    if (2.0 * ((double)batch * out_rows * out_cols + 256.0) * cout >= 4294967296.0) return MBN_EUNSUPPORTED;   // + a row tile of head room: ragged rows must not wrap (32-bit offsets)
    for (const void *p : ptrs)
        if (((uintptr_t)p % 16) != 0) return MBN_EUNSUPPORTED;
    return MBN_OK;
}

int mbn_launch_bf16_dwpw(mbn_context *ctx, hipStream_t stream, void *out, const void *in, const float *wd, const float *s2,
                         const float *b2, const void *wp, const float *s3, const float *b3, int batch, int in_rows,
                         int in_cols, int out_rows, int out_cols, int cin, int cout, int stride, int pad_top, int pad_left)
{
    const int rc = mbn_bf16_dwpw_check(out, in, wd, s2, b2, wp, s3, b3, batch, in_rows, in_cols, out_rows, out_cols, cin, cout,
                                       stride, pad_top, pad_left);
    if (rc != MBN_OK) return rc;
    BArgs a;
    a.out = (__bf16 *)out; a.in = (const __bf16 *)in; a.wp = (const __bf16 *)wp;
    a.wd = wd; a.s2 = s2; a.b2 = b2; a.s3 = s3; a.b3 = b3;
    a.m = (long)batch * out_rows * out_cols;
    a.h = in_rows; a.w = in_cols; a.ho = out_rows; a.wo = out_cols;
    a.cin = cin; a.cout = cout; a.pad_top = pad_top; a.pad_left = pad_left;
    mbn_udiv_magic((unsigned)out_cols, &a.wo_m, &a.wo_s);
    mbn_udiv_magic((unsigned)out_rows, &a.ho_m, &a.ho_s);
    a.in_bytes = (unsigned)(2.0 * batch * in_rows * in_cols * cin);
    const bool wide = (cout % 256) == 0 && g_mbn_tune.pw_tile != 1;
    if (stride == 1) {
        if (wide) launch<1, 256>(a, stream, ctx->num_cus);
        else launch<1, 128>(a, stream, ctx->num_cus);
    } else {
        if (wide) launch<2, 256>(a, stream, ctx->num_cus);
        else launch<2, 128>(a, stream, ctx->num_cus);
    }
    return MBN_OK;
}
