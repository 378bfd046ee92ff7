// mbn_f32_dwpw2_x6.hip — the unified-wave fused depthwise 3x3 -> pointwise 1x1 block (mbn_f32_dwpw2.hip; kernel.cl:62-92 +
// 94-114, pairs L4-5 ... of MobileNet.c:322-2599) with the pointwise products formed on the bf16 matrix cores from EXACT
// three-way operand splits. OPT-IN, as mbn_f32_pw_x6.hip: only under mbn_tune_set("pw_emul", 6 | 9); fp32 in, fp32 out.
//
// Same step structure as dwpw2_f32 — every wave computes its 1/8 of the depthwise chunk (2 pixels x 4 channels per lane,
// same fma order as mbn_f32_dw.hip: the depthwise values are the same bits), then its 32x64 / 64x64 MFMA tile; x window
// loaded a chunk ahead with out-of-range offsets for the padding, filter chunk by LDS-DMA, one barrier per chunk with
// explicit counters. What changes:
//   * D writes the chunk's A tile as three bf16 planes [128 px][32 ch] (h = bf16(v), m = bf16(v - h), l = v - h - m: all 24 bits
//     of v, mbn_f32_pw_x6.hip) — 40 more VALU instructions per lane and chunk, 6 ds_write_b64 instead of 2 ds_write_b128;
//   * the pointwise filter comes pre-split (split_filter, once per launch, channel-paired rows for the 8-byte epilogue
//     stores) and its three planes of the chunk are one contiguous image: 3 or 6 one-KiB DMA pieces per wave;
//   * M is 2 k16-steps of 6 (or 9) v_mfma_f32_32x32x16_bf16 per 32x32 block instead of 16 v_mfma_f32_32x32x2_f32 of twice the
//     duration: 768 / 1536 matrix-pipe cycles per chunk and wave instead of 2048 / 4096.
// The product order per k16-step and the chunk order are those of pw_gemm_x(b), so this kernel returns the same bits as the
// stand-alone depthwise launch followed by the pointwise launch under the same pw_emul (tested).
// LDS: A 2 x 24 KB, filter 2 x 24 KB (128 columns) or 2 x 48 KB (256 columns), depthwise constants 11 x CM floats, pointwise
// scale/shift. 256-column tiles fit only with CM = 256 (157 KB): blocks with Cin <= 256 and Cout = 256; wider blocks take the
// 128-column tile (Cin <= 512) or fall back to the fp32-MFMA kernels.
#include "mbn_internal.h"
#include "mbn_epilogue.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef mbn_f16v f16v;
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));

constexpr int BM = 128;
constexpr int PW = 16;                         // words per plane row (32 bf16)
constexpr int NW = 8, NT = 64 * NW;
constexpr unsigned OOB = 0xF0000000u;

struct XbArgs {
    float *out;
    const float *in, *wd, *s2, *b2, *s3, *b3;
    const unsigned *wimg;   // pre-split pointwise filter: [n-tile][chunk][plane][BN rows][64 B], rows channel-paired
    long m;
    int h, w, ho, wo;
    int cin, cout;
    int pad_top, pad_left;
    int mt, nt;
    unsigned in_bytes, wimg_bytes;
    unsigned wo_m, wo_s, ho_m, ho_s;
};

__device__ __forceinline__ int pswz(int row, int c) { return row * PW + (((c ^ (row >> 2)) & 3) << 2); }
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
// 4 consecutive channels of one pixel -> the three bf16 planes (exact: h + m + l == v)
__device__ __forceinline__ void split4(const f4 &v, u2 &H, u2 &M, u2 &L)
{
    const f2 p[2] = { f2{ v.x, v.y }, f2{ v.z, v.w } };
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const bf2 h = __builtin_convertvector(p[j], bf2);
        const f2 r = p[j] - __builtin_convertvector(h, f2);
        const bf2 m = __builtin_convertvector(r, bf2);
        const f2 l = r - __builtin_convertvector(m, f2);
        const bf2 lo = __builtin_convertvector(l, bf2);
        H[j] = __builtin_bit_cast(unsigned, h);
        M[j] = __builtin_bit_cast(unsigned, m);
        L[j] = __builtin_bit_cast(unsigned, lo);
    }
}

template <int VM_LEFT>
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(VM_LEFT) : "memory");
}

template <int NPC>
__device__ __forceinline__ void dma_image(__amdgpu_buffer_rsrc_t rsrc, unsigned *lds_b, const unsigned *voff, unsigned soff, int wave_u)
{
#pragma unroll
    for (int p = 0; p < NPC; p++)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)(lds_b + (p * NW + wave_u) * 256), 16, voff[p], soff, 0, 0);
}

// product list of mbn_f32_pw_x6.hip (plane of A, plane of B; 0 = h, 1 = m, 2 = l), smallest terms first
template <int NP> struct Prod;
template <> struct Prod<9> { static constexpr int pa[9] = { 2, 2, 1, 2, 0, 1, 1, 0, 0 }, pb[9] = { 2, 1, 2, 0, 2, 1, 0, 1, 0 }; };
template <> struct Prod<6> { static constexpr int pa[6] = { 2, 0, 1, 1, 0, 0 }, pb[6] = { 0, 2, 1, 0, 1, 0 }; };

// CM = largest Cin (depthwise constants in LDS), NO = largest Cout (pointwise scale/shift in LDS)
template <int S, int BN, int NP, int CM, int NO, bool PRE>
__global__ __launch_bounds__(NT) void dwpw2_x6(XbArgs a)
{
    constexpr int WN = 64, WM = BN == 256 ? 64 : 32;
    constexpr int WAVES_N = BN / WN;
    static_assert((BM / WM) * WAVES_N == NW, "8 waves");
    constexpr int MI = WM / 32, NI = WN / 32;
    constexpr int XC = S + 3, NX = 3 * XC;
    constexpr int PLANE_A = BM * PW, ABUF = 3 * PLANE_A;          // words
    constexpr int PLANE_B = BN * PW, BBUF = 3 * PLANE_B;
    constexpr int NPC = BBUF / 256 / NW;                          // 1-KiB DMA pieces per wave and chunk (3 / 6)
    static_assert(NPC * NW * 256 == BBUF, "filter image / waves mismatch");
    constexpr int NPH = NP / 2;                                   // products in the first half of a k16-step
    __shared__ __attribute__((aligned(16))) unsigned lds[2 * ABUF + 2 * BBUF + 11 * CM + 2 * NO];
    unsigned *const a_s0 = lds, *const b_s0 = lds + 2 * ABUF;
    float *const wd_s = reinterpret_cast<float *>(b_s0 + 2 * BBUF), *const sb_s = wd_s + 9 * CM;
    float *const sc3_s = sb_s + 2 * CM, *const sh3_s = sc3_s + NO;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nk = a.cin / 32, nwg = a.mt * a.nt;
    const unsigned mtot = (unsigned)a.m;

    for (int i = tid * 4; i < 9 * a.cin; i += NT * 4) *reinterpret_cast<f4 *>(wd_s + i) = *reinterpret_cast<const f4 *>(a.wd + i);
    for (int i = tid * 4; i < a.cin; i += NT * 4) {
        *reinterpret_cast<f4 *>(sb_s + i) = *reinterpret_cast<const f4 *>(a.s2 + i);
        *reinterpret_cast<f4 *>(sb_s + a.cin + i) = *reinterpret_cast<const f4 *>(a.b2 + i);
    }
    for (int i = tid; i < a.cout; i += NT) { sc3_s[i] = a.s3[i]; sh3_s[i] = a.b3[i]; }
    __syncthreads();
    if ((int)blockIdx.x >= nwg) return;

    const int c4 = tid & 7, pair = tid >> 3;                        // depthwise: tile rows 2*pair, 2*pair+1, channels 4*c4..+3 of the chunk
    const int wm = (wave_u / WAVES_N) * WM, wn = (wave_u % WAVES_N) * WN;
    const int li = lane & 31, lh = lane >> 5;
    const __amdgpu_buffer_rsrc_t irsrc = mbn_make_rsrc(a.in, a.in_bytes);
    const __amdgpu_buffer_rsrc_t wrsrc = mbn_make_rsrc(a.wimg, a.wimg_bytes);
    const __amdgpu_buffer_rsrc_t orsrc = mbn_make_rsrc(a.out, (unsigned)(a.m * a.cout * 4));
    // plane slots this lane writes: 8 bytes (4 channels) at chunk c4 >> 1, half c4 & 1 of rows 2*pair and 2*pair + 1
    const int aw0 = pswz(2 * pair, c4 >> 1) + (c4 & 1) * 2, aw1 = pswz(2 * pair + 1, c4 >> 1) + (c4 & 1) * 2;
    int fr_a[2], fr_b[2];
#pragma unroll
    for (int s = 0; s < 2; s++) {
        fr_a[s] = pswz(wm + li, 2 * s + lh);
        fr_b[s] = pswz(wn + li, 2 * s + lh);
    }
    unsigned b_vo[NPC];
#pragma unroll
    for (int p = 0; p < NPC; p++) b_vo[p] = (unsigned)((p * NW + wave_u) * 1024 + lane * 16);
    const float *wk = wd_s + c4 * 4;
    const float *sk = sb_s + c4 * 4;

    unsigned off[3][XC];
    auto set_offsets = [&](unsigned m0) __attribute__((always_inline)) {
        const unsigned m = m0 + 2 * pair;
        const bool mok = m < mtot;
        const unsigned q = a.wo_m ? __umulhi(m, a.wo_m) >> a.wo_s : m;
        const unsigned x = m - q * (unsigned)a.wo;
        const unsigned n = a.ho_m ? __umulhi(q, a.ho_m) >> a.ho_s : q;
        const unsigned y = q - n * (unsigned)a.ho;
        const int iy0 = (int)y * S - a.pad_top, ix0 = (int)x * S - a.pad_left;
        const unsigned cs = (unsigned)a.cin * 4u, rs = (unsigned)a.w * cs;
        const unsigned base = ((n * a.h + iy0) * a.w + ix0) * cs + (unsigned)(c4 * 4) * 4u;
#pragma unroll
        for (int dy = 0; dy < 3; dy++) {
            const bool rok = mok && (unsigned)(iy0 + dy) < (unsigned)a.h;
#pragma unroll
            for (int j = 0; j < XC; j++) {
                const bool ok = rok && (unsigned)(ix0 + j) < (unsigned)a.w;
                off[dy][j] = ok ? base + dy * rs + j * cs : OOB;
            }
        }
    };
    f4 xr[3][XC];
    auto ldx = [&](int kc) __attribute__((always_inline)) {
#pragma unroll
        for (int dy = 0; dy < 3; dy++)
#pragma unroll
            for (int j = 0; j < XC; j++)
                xr[dy][j] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(irsrc, off[dy][j], kc * 128, 0));
    };
    auto ldx_part = [&](int kc, const int part) __attribute__((always_inline)) {
        constexpr int PER = (NX + 3) / 4;
#pragma unroll
        for (int i = part * PER; i < (part + 1) * PER && i < NX; i++)
            xr[i / XC][i % XC] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(irsrc, off[i / XC][i % XC], kc * 128, 0));
    };
    f4 wreg[11];
    auto ldw = [&](int kc) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 9; t++) wreg[t] = *reinterpret_cast<const f4 *>(wk + kc * 32 + t * a.cin);
        wreg[9] = *reinterpret_cast<const f4 *>(sk + kc * 32);
        wreg[10] = *reinterpret_cast<const f4 *>(sk + a.cin + kc * 32);
    };
    auto dw = [&](int kc, const int buf) __attribute__((always_inline)) {
        if (!PRE) ldw(kc);
        f4 acc0 = f4{ 0.f, 0.f, 0.f, 0.f }, acc1 = acc0;
#pragma unroll
        for (int dy = 0; dy < 3; dy++)
#pragma unroll
            for (int dx = 0; dx < 3; dx++) {
                acc0 = __builtin_elementwise_fma(xr[dy][dx], wreg[dy * 3 + dx], acc0);
                acc1 = __builtin_elementwise_fma(xr[dy][dx + S], wreg[dy * 3 + dx], acc1);
            }
        u2 H, M, L;
        unsigned *const ab = a_s0 + buf * ABUF;
        split4(bn_relu6(acc0, wreg[9], wreg[10]), H, M, L);
        *reinterpret_cast<u2 *>(ab + aw0) = H;
        *reinterpret_cast<u2 *>(ab + PLANE_A + aw0) = M;
        *reinterpret_cast<u2 *>(ab + 2 * PLANE_A + aw0) = L;
        split4(bn_relu6(acc1, wreg[9], wreg[10]), H, M, L);
        *reinterpret_cast<u2 *>(ab + aw1) = H;
        *reinterpret_cast<u2 *>(ab + PLANE_A + aw1) = M;
        *reinterpret_cast<u2 *>(ab + 2 * PLANE_A + aw1) = L;
    };

    f16v acc[MI][NI];
    u4 fa[3][MI], fb[3][NI];
    auto ldfrag = [&](const int buf, const int s) __attribute__((always_inline)) {
#pragma unroll
        for (int pl = 0; pl < 3; pl++) {
#pragma unroll
            for (int mi = 0; mi < MI; mi++) fa[pl][mi] = *reinterpret_cast<const u4 *>(a_s0 + buf * ABUF + pl * PLANE_A + fr_a[s] + mi * 32 * PW);
#pragma unroll
            for (int ni = 0; ni < NI; ni++) fb[pl][ni] = *reinterpret_cast<const u4 *>(b_s0 + buf * BBUF + pl * PLANE_B + fr_b[s] + ni * 32 * PW);
        }
    };
    auto mfma_part = [&](const int q0, const int q1) __attribute__((always_inline)) {
#pragma unroll
        for (int q = q0; q < q1; q++)
#pragma unroll
            for (int mi = 0; mi < MI; mi++)
#pragma unroll
                for (int ni = 0; ni < NI; ni++)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, fa[Prod<NP>::pa[q]][mi]),
                                                                          __builtin_bit_cast(bf8, fb[Prod<NP>::pb[q]][ni]), acc[mi][ni], 0, 0, 0);
    };
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int mi = 0; mi < MI; mi++)
#pragma unroll
            for (int ni = 0; ni < NI; ni++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[mi][ni][r] = 0.f;
    };

    // three cursors over the flattened (tile, chunk) sequence, as in dwpw2_f32: L one chunk ahead of D one chunk ahead of M
    int vbM, kM, n0M; unsigned m0M;
    int vbD, kD, n0D; unsigned m0D;
    bool validD;
    auto origin = [&](int vb, unsigned &m0, int &n0) __attribute__((always_inline)) {
        const int lid = xcd_remap(vb, nwg);
        n0 = (lid % a.nt) * BN;
        m0 = (unsigned)(lid / a.nt) * BM;
    };
    auto img_off = [&](int n0, int kc) __attribute__((always_inline)) { return (unsigned)((n0 / BN) * nk + kc) * (unsigned)(BBUF * 4); };

    vbM = blockIdx.x; kM = 0;
    origin(vbM, m0M, n0M);
    set_offsets(m0M);
    ldx(0);
    if (PRE) ldw(0);
    dma_image<NPC>(wrsrc, b_s0, b_vo, img_off(n0M, 0), wave_u);
    dw(0, 0);
    vbD = vbM; kD = 1; m0D = m0M; n0D = n0M; validD = true;
    if (kD >= nk) {
        kD = 0; vbD += gridDim.x; validD = vbD < nwg;
        if (validD) { origin(vbD, m0D, n0D); set_offsets(m0D); }
    }
    if (validD) {
        ldx(kD);
        if (PRE) ldw(kD);
    }
    zero_acc();
    if (validD) lds_barrier<NX>();
    else lds_barrier<0>();

#define MBN_X6_STEP(P)                                                                                                  \
    {                                                                                                                   \
        __builtin_amdgcn_s_setprio(3);                                                                                  \
        ldfrag(P, 0);                                                                                                   \
        bool validL = false;                                                                                            \
        int vbL = vbD, kL = kD + 1, n0L = n0D;                                                                          \
        unsigned m0L = m0D;                                                                                             \
        if (validD) {                                                                                                   \
            dw(kD, P ^ 1);                                                                                              \
            dma_image<NPC>(wrsrc, b_s0 + (P ^ 1) * BBUF, b_vo, img_off(n0D, kD), wave_u);   /* behind the depthwise part, as in dwpw2_f32 */ \
            validL = true;                                                                                              \
            if (kL >= nk) {                                                                                             \
                kL = 0; vbL += gridDim.x; validL = vbL < nwg;                                                           \
                if (validL) { origin(vbL, m0L, n0L); set_offsets(m0L); }                                                \
            }                                                                                                           \
        }                                                                                                               \
        __builtin_amdgcn_s_setprio(0);                                                                                  \
        if (validL) ldx_part(kL, 0);                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
        mfma_part(0, NPH);                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
        if (validL) ldx_part(kL, 1);                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
        mfma_part(NPH, NP);                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
        ldfrag(P, 1);                                                                                                   \
        if (validL) ldx_part(kL, 2);                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
        mfma_part(0, NPH);                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
        if (validL) ldx_part(kL, 3);                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
        mfma_part(NPH, NP);                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
        if (PRE && validL) ldw(kL);                                                                                     \
        if (validL) lds_barrier<NX>();                                                                                  \
        else lds_barrier<0>();                                                                                          \
        if (kM == nk - 1) {                                                                                             \
            if (m0M + BM <= mtot) mbn_store_relu6_f32_pair<MI, NI, 0>(orsrc, (unsigned)a.cout, m0M + wm, n0M + wn, lane, acc, sc3_s, sh3_s); \
            else mbn_store_relu6_f32_pair<MI, NI, 1>(orsrc, (unsigned)a.cout, m0M + wm, n0M + wn, lane, acc, sc3_s, sh3_s); \
            zero_acc();                                                                                                 \
        }                                                                                                               \
        if (!validD) break;                                                                                             \
        vbM = vbD; kM = kD; m0M = m0D; n0M = n0D;                                                                       \
        vbD = vbL; kD = kL; m0D = m0L; n0D = n0L; validD = validL;                                                      \
    }

    for (;;) {
        MBN_X6_STEP(0)
        MBN_X6_STEP(1)
    }
#undef MBN_X6_STEP
}

template <int S, int BN, int NP, int CM, int NO, bool PRE>
void launch_x6(XbArgs &a, hipStream_t s, int num_cus)
{
    a.mt = (int)((a.m + BM - 1) / BM);
    a.nt = a.cout / BN;
    const long nwg = (long)a.mt * a.nt;
    long grid = num_cus;
    if (grid > nwg) grid = nwg;
    hipLaunchKernelGGL((dwpw2_x6<S, BN, NP, CM, NO, PRE>), dim3((unsigned)grid), dim3(NT), 0, s, a);
}

}   // namespace

// MBN_OK if launched, MBN_EUNSUPPORTED if pw_emul is off or the block is outside this kernel's envelope (the caller then takes the
// fp32-MFMA kernels). The caller has passed mbn_f32_dwpw_check.
int mbn_launch_f32_dwpw2_x6(mbn_context *ctx, hipStream_t stream, float *out, const float *in, const float *wd, const float *s2,
                            const float *b2, const float *wp, const float *s3, const float *b3, int batch, int in_rows, int in_cols,
                            int out_rows, int out_cols, int cin, int cout, int stride, int pad_top, int pad_left)
{
    const int np = g_mbn_tune.pw_emul;
    if (np != 6 && np != 9) return MBN_EUNSUPPORTED;
    if (cin > 512 || cout > 1024) return MBN_EUNSUPPORTED;
    XbArgs a;
    a.out = out; a.in = in; a.wd = wd; a.s2 = s2; a.b2 = b2; a.s3 = s3; a.b3 = b3;
    a.m = (long)batch * out_rows * out_cols;
    a.h = in_rows; a.w = in_cols; a.ho = out_rows; a.wo = out_cols;
    a.cin = cin; a.cout = cout; a.pad_top = pad_top; a.pad_left = pad_left;
    mbn_udiv_magic((unsigned)out_cols, &a.wo_m, &a.wo_s);
    mbn_udiv_magic((unsigned)out_rows, &a.ho_m, &a.ho_s);
    a.in_bytes = (unsigned)(4.0 * batch * in_rows * in_cols * cin);
    // 256-column tile: Cin <= 256 and Cout == 256 (LDS), and only when those tiles alone fill the chip (as dwpw2_f32)
    const bool wide = cout == 256 && cin <= 256 && g_mbn_tune.pw_tile != 1 && ((a.m + BM - 1) / BM) >= ctx->num_cus;
    const int bn = wide ? 256 : 128;
    const unsigned *img = nullptr;
    unsigned img_bytes = 0;
    const int rc = mbn_pw_emul_filter_image(ctx, stream, wp, cout, cin, bn, 1, &img, &img_bytes);
    if (rc != MBN_OK) return rc;
    a.wimg = img; a.wimg_bytes = img_bytes;
    const int cus = ctx->num_cus;
    if (np == 6) {
        if (stride == 1) { if (wide) launch_x6<1, 256, 6, 256, 256, false>(a, stream, cus); else launch_x6<1, 128, 6, 512, 1024, true>(a, stream, cus); }
        else { if (wide) launch_x6<2, 256, 6, 256, 256, false>(a, stream, cus); else launch_x6<2, 128, 6, 512, 1024, true>(a, stream, cus); }
    } else {
        if (stride == 1) { if (wide) launch_x6<1, 256, 9, 256, 256, false>(a, stream, cus); else launch_x6<1, 128, 9, 512, 1024, true>(a, stream, cus); }
        else { if (wide) launch_x6<2, 256, 9, 256, 256, false>(a, stream, cus); else launch_x6<2, 128, 9, 512, 1024, true>(a, stream, cus); }
    }
    return MBN_OK;
}
