// mbn_f32_pw_x6.hip — OPT-IN form of the fp32 1x1 pointwise GEMM (kernel.cl:94-114) that computes the fp32 products on the
// bf16 matrix cores from an EXACT three-way split of every operand. Not the default: mbn_tune_set("pw_emul", 6 | 9).
//
// Every fp32 x is written as x = h + m + l with h = bf16(x), m = bf16(x - h), l = x - h - m: both subtractions are exact in
// fp32 and l has at most 8 significant bits, so the three bf16 values carry all 24 bits of x (checked bit for bit by the
// parity tests on random data: h + m + l == x). a*b is then the sum of nine bf16 x bf16 products, each of them EXACT in the
// fp32 accumulator of v_mfma_f32_32x32x16_bf16; pw_emul = 9 issues all nine, pw_emul = 6 drops m*l, l*m and l*l, whose sum is
// below 2^-24 |a*b| — less than the rounding of the one fp32 product they belong to. What differs from pw_gemm<float> is
// therefore the ORDER of the fp32 additions (as between pw_gemm and the split-K kernel), not the precision of the terms:
// tests/test_parity_gpu.py holds both forms to the same per-layer bound against the double-precision oracle as the fp32
// MFMA kernel, and profiles/r02/m_pw_emul.txt lists the three error figures side by side.
//
// Why: v_mfma_f32_32x32x2_f32 retires 64 FLOP/clk/SIMD, v_mfma_f32_32x32x16_bf16 1024: six bf16 MFMAs per 16 k replace
// eight fp32 MFMAs of twice the duration each — 2.7x fewer matrix-pipe cycles (1.8x with nine). The split itself is VALU
// work (about 5 instructions per element: 2 conversions, 2 subtractions, shifts), and VALU time ADDS to MFMA time on a SIMD
// (DESIGN §3.6), so it is done ONCE per workgroup tile on the way into LDS — global -> registers -> three bf16 planes in LDS —
// not per wave in the fragment path; a 128x128 tile splits (128 + 128) x 32 values per 32 k for 4 x 48 MFMAs.
//
// LDS: three planes [rows][64 B] for A and three for B (32 k of bf16 per row); the 16-byte chunk index is XORed with
// (row >> 2) & 3, which makes both the staging ds_write_b128 (4 lanes per row) and the fragment ds_read_b128 (16 consecutive
// rows, one chunk) hit 16 distinct 16-byte slots of the 256-byte bank row (SQ_LDS_BANK_CONFLICT = 0 measured). Persistent
// workgroups with the XCD-contiguous tile order of pw_gemm; scale/shift of the tile's columns staged in LDS; the same
// buffer-store epilogue (mbn_epilogue.h).
//
// Two kernels. pw_gemm_xb (shipped for outputs of 128 columns and wider): the FILTER is split once per launch by a small
// pre-kernel (split_filter, ~4 us) into a per-filter workspace laid out as the LDS tile images, and streamed from there by
// LDS-DMA one k-tile ahead (double-buffered); only the activations are split in the GEMM (single-buffered planes, loads
// issued two k-tiles ahead, explicit s_waitcnt counters at the barriers). pw_gemm_x: both operands split on the way in,
// no workspace (narrow outputs; first use inside a hipGraph capture, where nothing may be allocated).
//
// Measured (profiles/r02/m_pw_emul.txt, batch 256): 512 -> 512 at 14x14 0.195 ms (pw_gemm<float>) -> 0.151-0.160; the step
// 87-89 k -> 110-114 k images/s with the fused blocks and the stem on the same form. The matrix pipe is busy 47 % of the kernel (SQ_VALU_MFMA_BUSY_CYCLES), the clock holds
// 2.05 GHz (GRBM_GUI_ACTIVE; 2.09 under pw_gemm<float>). Three 2-byte planes per operand make the 128x128 tile 74 KB of LDS: two
// workgroups (two waves per SIMD) per CU, and 1568 tiles on 512 slots end in a 6 %-full fourth round (7 tile times per CU against
// 6.125). Neither the in-GEMM split nor the fragment-read latency is the gap: an all-DMA, single-barrier, software-pipelined form
// of the loop measured 10 % faster as a kernel and slower with its separate split pass (same file, (13)); 64-row tiles and three
// workgroups per CU do not fit the register file or lose on the 7x7 layers ((11), (14)).
// The split is exact for 2^-110 <= |x| < 2^127 (bf16 has fp32's exponent range; h can round to infinity in the top binade, the
// matrix cores flush the bf16 denormals of the low planes below 2^-110): (19).
#include "mbn_internal.h"
#include "mbn_epilogue.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef mbn_f16v f16v;
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

struct XArgs {
    float *out;
    const float *in, *filt, *scale, *shift;
    long m;
    int k, n, act;
    int mt, nt;
    int fast_epi;
    const unsigned *bws;          // pw_gemm_xb: the filter's three bf16 planes, pre-split, in LDS tile-image order (split_filter)
    unsigned bws_bytes;
    int prio;
};

constexpr int KT = 32;            // k per k-tile
constexpr int PW = 16;            // 4-byte words per plane row (32 bf16)

__device__ __forceinline__ int pswz(int row, int c) { return row * PW + (((c ^ (row >> 2)) & 3) << 2); }

__device__ __forceinline__ int x_remap(int vb, int nwg)
{
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = vb & 7;
    return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (vb >> 3);
}

// 8 consecutive k of one row -> the three bf16 planes (exact: h + m + l == x)
__device__ __forceinline__ void split8(const f4 &x0, const f4 &x1, u4 &H, u4 &M, u4 &L)
{
    const f2 v[4] = { f2{ x0.x, x0.y }, f2{ x0.z, x0.w }, f2{ x1.x, x1.y }, f2{ x1.z, x1.w } };
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const bf2 h = __builtin_convertvector(v[j], bf2);
        const f2 r = v[j] - __builtin_convertvector(h, f2);
        const bf2 m = __builtin_convertvector(r, bf2);
        const f2 l = r - __builtin_convertvector(m, f2);
        const bf2 lo = __builtin_convertvector(l, bf2);
        H[j] = __builtin_bit_cast(unsigned, h);
        M[j] = __builtin_bit_cast(unsigned, m);
        L[j] = __builtin_bit_cast(unsigned, lo);
    }
}

// product list: plane of A, plane of B (0 = h, 1 = m, 2 = l); smallest terms first
template <int NP> struct Prod;
template <> struct Prod<9> { static constexpr int pa[9] = { 2, 2, 1, 2, 0, 1, 1, 0, 0 }, pb[9] = { 2, 1, 2, 0, 2, 1, 0, 1, 0 }; };
template <> struct Prod<6> { static constexpr int pa[6] = { 2, 0, 1, 1, 0, 0 }, pb[6] = { 0, 2, 1, 0, 1, 0 }; };
#ifdef MBN_LAB
template <> struct Prod<3> { static constexpr int pa[3] = { 1, 0, 0 }, pb[3] = { 0, 1, 0 }; };      // measurement only: 2^-16
template <> struct Prod<1> { static constexpr int pa[1] = { 0 }, pb[1] = { 0 }; };                  // measurement only: bf16 operands
#endif

template <int BM, int BN, int WM, int WN, int NP, int NBUF, int OCC>
__global__ __launch_bounds__(64 * (BM / WM) * (BN / WN)) __attribute__((amdgpu_waves_per_eu(OCC, OCC))) void pw_gemm_x(XArgs a)
{
    constexpr int WAVES_N = BN / WN, NT = 64 * (BM / WM) * WAVES_N;
    constexpr int MI = WM / 32, NI = WN / 32;
    constexpr int RP = NT / 4;                               // rows per staging pass (4 lanes x 32 B per row)
    constexpr int A_LD = BM / RP, B_LD = BN / RP;
    static_assert(A_LD >= 1 && B_LD >= 1 && A_LD * RP == BM && B_LD * RP == BN, "tile/threads mismatch");
    constexpr int PLANE_A = BM * PW, PLANE_B = BN * PW, BUF = 3 * (PLANE_A + PLANE_B);
    constexpr int NPL = NP == 1 ? 1 : NP == 3 ? 2 : 3;       // planes actually used
    // behind the planes: scale | shift of this tile's BN columns, two slots by tile parity (written at the top of a tile, read in
    // its epilogue; the slot is rewritten two tiles later, with that tile's barriers in between)
    __shared__ __attribute__((aligned(16))) unsigned lds[NBUF * BUF + 4 * BN];
    float *const ss_s = reinterpret_cast<float *>(lds + NBUF * BUF);
    const bool ss_lds = a.scale && a.shift;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nwg = a.mt * a.nt;
    const int wm = (wave / WAVES_N) * WM, wn = (wave % WAVES_N) * WN;
    const int li = lane & 31, lh = lane >> 5;
    const int nk = a.k / KT;
    const int st_c = tid & 3, st_r = tid >> 2;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int wm_u = (wave_u / WAVES_N) * WM, wn_u = (wave_u % WAVES_N) * WN;
    const __amdgpu_buffer_rsrc_t orsrc = mbn_make_rsrc(a.out, (unsigned)(a.m * a.n * 4L));

    const float *a_src[A_LD], *b_src[B_LD];
    f4 a_reg[A_LD][2], b_reg[B_LD][2];
    int st_off[A_LD > B_LD ? A_LD : B_LD];
#pragma unroll
    for (int p = 0; p < (A_LD > B_LD ? A_LD : B_LD); p++) st_off[p] = pswz(p * RP + st_r, st_c);

    auto set_tile = [&](int vb, long &m0, int &n0) {
        const int lid = x_remap(vb, nwg);
        m0 = (long)(lid / a.nt) * BM;
        n0 = (lid % a.nt) * BN;
#pragma unroll
        for (int p = 0; p < A_LD; p++) {
            long gm = m0 + p * RP + st_r;
            if (gm >= a.m) gm = a.m - 1;                     // rows past M are computed but never stored
            a_src[p] = a.in + gm * a.k + st_c * 8;
        }
#pragma unroll
        for (int p = 0; p < B_LD; p++) {
            int gn = n0 + p * RP + st_r;
            if (gn >= a.n) gn = a.n - 1;
            b_src[p] = a.filt + (long)gn * a.k + st_c * 8;
        }
    };
    auto stage_load = [&](int k0) {
#pragma unroll
        for (int p = 0; p < A_LD; p++) {
            a_reg[p][0] = *reinterpret_cast<const f4 *>(a_src[p] + k0);
            a_reg[p][1] = *reinterpret_cast<const f4 *>(a_src[p] + k0 + 4);
        }
#pragma unroll
        for (int p = 0; p < B_LD; p++) {
            b_reg[p][0] = *reinterpret_cast<const f4 *>(b_src[p] + k0);
            b_reg[p][1] = *reinterpret_cast<const f4 *>(b_src[p] + k0 + 4);
        }
    };
    auto stage_store = [&](int buf) {
        unsigned *base = lds + buf * BUF;
#pragma unroll
        for (int p = 0; p < A_LD; p++) {
            u4 H, M, L;
            split8(a_reg[p][0], a_reg[p][1], H, M, L);
            *reinterpret_cast<u4 *>(base + st_off[p]) = H;
            if (NPL > 1) *reinterpret_cast<u4 *>(base + PLANE_A + st_off[p]) = M;
            if (NPL > 2) *reinterpret_cast<u4 *>(base + 2 * PLANE_A + st_off[p]) = L;
        }
#pragma unroll
        for (int p = 0; p < B_LD; p++) {
            u4 H, M, L;
            split8(b_reg[p][0], b_reg[p][1], H, M, L);
            *reinterpret_cast<u4 *>(base + 3 * PLANE_A + st_off[p]) = H;
            if (NPL > 1) *reinterpret_cast<u4 *>(base + 3 * PLANE_A + PLANE_B + st_off[p]) = M;
            if (NPL > 2) *reinterpret_cast<u4 *>(base + 3 * PLANE_A + 2 * PLANE_B + st_off[p]) = L;
        }
    };

    // fragment offsets: k16-step s of a k-tile reads chunk 2s + lh of rows wm + 32 mi + li (the swizzle depends on the row
    // only through (row >> 2) & 3 and 32 mi does not change it: one offset per step)
    int fr_a[2], fr_b[2];
#pragma unroll
    for (int s = 0; s < 2; s++) {
        fr_a[s] = pswz(wm + li, 2 * s + lh);
        fr_b[s] = pswz(wn + li, 2 * s + lh);
    }

    long m0;
    int n0;
    int vb = blockIdx.x, tpar = 0;
    if (vb >= nwg) return;
    set_tile(vb, m0, n0);
    stage_load(0);

    for (;;) {
        f16v acc[MI][NI];
#pragma unroll
        for (int mi = 0; mi < MI; mi++)
#pragma unroll
            for (int ni = 0; ni < NI; ni++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[mi][ni][r] = 0.f;

        float *const sc_s = ss_s + (tpar ? 2 * BN : 0), *const sh_s = sc_s + BN;
        if (ss_lds && tid < BN) {
            const int col = n0 + tid < a.n ? n0 + tid : a.n - 1;
            sc_s[tid] = a.scale[col];
            sh_s[tid] = a.shift[col];
        }
        tpar ^= 1;
        stage_store(0);
        __syncthreads();
        for (int kt = 0; kt < nk; kt++) {
            const int cur = NBUF == 2 ? (kt & 1) : 0;
            if (kt + 1 < nk) stage_load((kt + 1) * KT);
            const unsigned *As = lds + cur * BUF, *Bs = As + 3 * PLANE_A;
#pragma unroll
            for (int s = 0; s < 2; s++) {
                u4 fa[NPL][MI], fb[NPL][NI];
#pragma unroll
                for (int pl = 0; pl < NPL; pl++) {
#pragma unroll
                    for (int mi = 0; mi < MI; mi++) fa[pl][mi] = *reinterpret_cast<const u4 *>(As + pl * PLANE_A + fr_a[s] + mi * 32 * PW);
#pragma unroll
                    for (int ni = 0; ni < NI; ni++) fb[pl][ni] = *reinterpret_cast<const u4 *>(Bs + pl * PLANE_B + fr_b[s] + ni * 32 * PW);
                }
#pragma unroll
                for (int q = 0; q < NP; q++)
#pragma unroll
                    for (int mi = 0; mi < MI; mi++)
#pragma unroll
                        for (int ni = 0; ni < NI; ni++)
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, fa[Prod<NP>::pa[q]][mi]),
                                                                                  __builtin_bit_cast(bf8, fb[Prod<NP>::pb[q]][ni]), acc[mi][ni], 0, 0, 0);
            }
            if (NBUF == 1) __syncthreads();                   // single buffer: everyone has read tile kt before it is overwritten
            if (kt + 1 < nk) stage_store(NBUF == 2 ? cur ^ 1 : 0);
            __syncthreads();
        }

        // next tile: first global loads before this tile's epilogue stores
        const long cm0 = m0;
        const int cn0 = n0;
        const int nvb = vb + gridDim.x;
        const bool more = nvb < nwg;
        if (more) {
            set_tile(nvb, m0, n0);
            stage_load(0);
        }

        if (a.act == MBN_ACT_RELU6 && ss_lds && cm0 + BM <= a.m && cn0 + BN <= a.n && a.fast_epi) {
            // the staged constants are indexed by column inside the tile: pass pointers shifted by the tile's first column
            mbn_store_relu6_f32<MI, NI, 0, float>(orsrc, (unsigned)a.n, (unsigned)cm0 + wm_u, cn0 + wn_u, lane, acc, sc_s - cn0, sh_s - cn0, (unsigned)a.m, a.n);
        } else {
#pragma unroll
            for (int ni = 0; ni < NI; ni++) {
                const int col = cn0 + wn + ni * 32 + li;
                const bool cok = col < a.n;
                const int cc = cok ? col : a.n - 1;
                const float sc = a.scale ? a.scale[cc] : 1.f;
                const float sh = a.shift ? a.shift[cc] : 0.f;
#pragma unroll
                for (int mi = 0; mi < MI; mi++)
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const unsigned row = (unsigned)cm0 + wm + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        float v = fmaf(acc[mi][ni][r], sc, sh);
                        if (a.act == MBN_ACT_RELU6) v = fminf(fmaxf(v, 0.f), 6.f);
                        else if (a.act == MBN_ACT_RELU) v = fmaxf(v, 0.f);
                        // 32-bit offsets through the output's descriptor (the launcher keeps M * N * 4 below 4 GiB): no 64-bit
                        // per-lane addresses live across the tile loop (they were the kernel's only spills = its scratch segment)
                        if (cok && row < (unsigned)a.m) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), orsrc, (row * (unsigned)a.n + (unsigned)col) * 4u, 0, 0);
                    }
            }
        }
        if (!more) break;
        vb = nvb;
    }
}

// ---- filter pre-split: [N][K] fp32 -> per (n-tile, k-tile) the three planes of BN rows x 64 B in the LDS image order of
// pw_gemm_xb (rows past N repeat row N-1). One thread per (row, 16-byte chunk): 4 * BN threads per block.
template <int BN>
__global__ __launch_bounds__(4 * BN) void split_filter(unsigned *ws, const float *filt, int n, int k, int paired)
{
    const int ntile = blockIdx.x, kt = blockIdx.y, nk = gridDim.y;
    const int row = threadIdx.x >> 2, c = threadIdx.x & 3;
    // paired: image row `row` holds output channel mbn_pair_channel(row) (mbn_epilogue.h: 8-byte epilogue stores of the fused blocks)
    const int gn = min(ntile * BN + (paired ? mbn_pair_channel(row) : row), n - 1);
    const float *src = filt + (long)gn * k + kt * KT + c * 8;
    u4 H, M, L;
    split8(*reinterpret_cast<const f4 *>(src), *reinterpret_cast<const f4 *>(src + 4), H, M, L);
    unsigned *dst = ws + (size_t)(ntile * nk + kt) * 3 * BN * PW + pswz(row, c);
    *reinterpret_cast<u4 *>(dst) = H;
    *reinterpret_cast<u4 *>(dst + BN * PW) = M;
    *reinterpret_cast<u4 *>(dst + 2 * BN * PW) = L;
}

// s_barrier with explicit counters (as in mbn_bf16_pw_ring.hip): __syncthreads would drain vmcnt(0) — the A loads issued two
// k-tiles ahead included — at every barrier while an LDS-DMA is outstanding. VM_LEFT < 0: vector memory is not waited for.
template <int VM_LEFT>
__device__ __forceinline__ void xb_barrier()
{
    if constexpr (VM_LEFT < 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(VM_LEFT) : "memory");
}

__device__ __forceinline__ void dma_pieces(const __amdgpu_buffer_rsrc_t rsrc, unsigned *lds_dst, const unsigned *voff, unsigned soff, int wave_u,
                                           const int NPC, const int NW)
{
#pragma unroll
    for (int p = 0; p < NPC; p++)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)(lds_dst + (p * NW + wave_u) * 256), 16, voff[p], soff, 0, 0);
}

// ---- the GEMM with the filter planes streamed by LDS-DMA from the pre-split image (double-buffered: a k-tile ahead) and only
// the activations split on the way in (single-buffered planes, two barriers per k-tile): half the VALU work and LDS writes of
// pw_gemm_x, 16 staging VGPRs less.
template <int BM, int BN, int WM, int WN, int NP, int OCC, int NBB = 2, int ASETS = 2>
__global__ __launch_bounds__(64 * (BM / WM) * (BN / WN)) __attribute__((amdgpu_waves_per_eu(OCC, OCC))) void pw_gemm_xb(XArgs a)
{
    constexpr int WAVES_N = BN / WN, NW = (BM / WM) * WAVES_N, NT = 64 * NW;
    constexpr int MI = WM / 32, NI = WN / 32;
    constexpr int RP = NT / 4;
    constexpr int A_LD = BM / RP;
    static_assert(A_LD >= 1 && A_LD * RP == BM, "tile/threads mismatch");
    constexpr int PLANE_A = BM * PW, PLANE_B = BN * PW, IMG_B = 3 * PLANE_B;        // words
    constexpr int NPC = IMG_B / 256 / NW;                                          // 1-KB DMA pieces per wave per k-tile
    static_assert(NPC * NW * 256 == IMG_B, "filter image / waves mismatch");
    // NBB = 1: one filter buffer (50 KB per workgroup: three per CU); its DMA is then issued behind the first barrier of the k-tile
    __shared__ __attribute__((aligned(16))) unsigned lds[3 * PLANE_A + NBB * IMG_B + 4 * BN];
    unsigned *const Bbuf = lds + 3 * PLANE_A;
    float *const ss_s = reinterpret_cast<float *>(lds + 3 * PLANE_A + NBB * IMG_B);
    const bool ss_lds = a.scale && a.shift;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nwg = a.mt * a.nt;
    const int wm = (wave / WAVES_N) * WM, wn = (wave % WAVES_N) * WN;
    const int li = lane & 31, lh = lane >> 5;
    const int nk = a.k / KT;
    const int st_c = tid & 3, st_r = tid >> 2;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int wm_u = (wave_u / WAVES_N) * WM, wn_u = (wave_u % WAVES_N) * WN;
    const __amdgpu_buffer_rsrc_t orsrc = mbn_make_rsrc(a.out, (unsigned)(a.m * a.n * 4L));
    const __amdgpu_buffer_rsrc_t brsrc = mbn_make_rsrc(a.bws, a.bws_bytes);

    const float *a_src[A_LD];
    f4 a_r0[A_LD][2], a_r1[A_LD][2];          // two register sets: the loads of a k-tile are issued TWO k-tiles before their split
    int st_off[A_LD];
    unsigned b_vo[NPC];
#pragma unroll
    for (int p = 0; p < A_LD; p++) st_off[p] = pswz(p * RP + st_r, st_c);
#pragma unroll
    for (int p = 0; p < NPC; p++) b_vo[p] = (unsigned)((p * NW + wave) * 1024 + lane * 16);

    int ntile_cur = 0;
    auto set_tile = [&](int vb, long &m0, int &n0) {
        const int lid = x_remap(vb, nwg);
        m0 = (long)(lid / a.nt) * BM;
        ntile_cur = lid % a.nt;
        n0 = ntile_cur * BN;
#pragma unroll
        for (int p = 0; p < A_LD; p++) {
            long gm = m0 + p * RP + st_r;
            if (gm >= a.m) gm = a.m - 1;
            a_src[p] = a.in + gm * a.k + st_c * 8;
        }
    };
    auto stage_load = [&](f4 (&r)[A_LD][2], int k0) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < A_LD; p++) {
            r[p][0] = *reinterpret_cast<const f4 *>(a_src[p] + k0);
            r[p][1] = *reinterpret_cast<const f4 *>(a_src[p] + k0 + 4);
        }
    };
    auto stage_store = [&](const f4 (&r)[A_LD][2]) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < A_LD; p++) {
            u4 H, M, L;
            split8(r[p][0], r[p][1], H, M, L);
            *reinterpret_cast<u4 *>(lds + st_off[p]) = H;
            *reinterpret_cast<u4 *>(lds + PLANE_A + st_off[p]) = M;
            *reinterpret_cast<u4 *>(lds + 2 * PLANE_A + st_off[p]) = L;
        }
    };
    auto dma_b = [&](int kt, int buf) __attribute__((always_inline)) {
        dma_pieces(brsrc, Bbuf + buf * IMG_B, b_vo, (unsigned)(ntile_cur * nk + kt) * (unsigned)(IMG_B * 4), wave_u, NPC, NW);
    };

    int fr_a[2], fr_b[2];
#pragma unroll
    for (int s = 0; s < 2; s++) {
        fr_a[s] = pswz(wm + li, 2 * s + lh);
        fr_b[s] = pswz(wn + li, 2 * s + lh);
    }

    long m0;
    int n0;
    int vb = blockIdx.x, tpar = 0;
    if (vb >= nwg) return;
    set_tile(vb, m0, n0);
    stage_load(a_r0, 0);
    if (ASETS == 2 && nk > 1) stage_load(a_r1, KT);
    dma_b(0, 0);

    for (;;) {
        f16v acc[MI][NI];
#pragma unroll
        for (int mi = 0; mi < MI; mi++)
#pragma unroll
            for (int ni = 0; ni < NI; ni++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[mi][ni][r] = 0.f;

        auto compute = [&](const unsigned *Bs) __attribute__((always_inline)) {
            const unsigned *As = lds;
#pragma unroll
            for (int s = 0; s < 2; s++) {
                u4 fa[3][MI], fb[3][NI];
#pragma unroll
                for (int pl = 0; pl < 3; pl++) {
#pragma unroll
                    for (int mi = 0; mi < MI; mi++) fa[pl][mi] = *reinterpret_cast<const u4 *>(As + pl * PLANE_A + fr_a[s] + mi * 32 * PW);
#pragma unroll
                    for (int ni = 0; ni < NI; ni++) fb[pl][ni] = *reinterpret_cast<const u4 *>(Bs + pl * PLANE_B + fr_b[s] + ni * 32 * PW);
                }
#pragma unroll
                for (int q = 0; q < NP; q++)
#pragma unroll
                    for (int mi = 0; mi < MI; mi++)
#pragma unroll
                        for (int ni = 0; ni < NI; ni++)
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, fa[Prod<NP>::pa[q]][mi]),
                                                                                  __builtin_bit_cast(bf8, fb[Prod<NP>::pb[q]][ni]), acc[mi][ni], 0, 0, 0);
                if (OCC >= 3) __builtin_amdgcn_sched_barrier(0);   // one step's fragments live at a time (168 VGPRs)
            }
        };
        float *const sc_s = ss_s + (tpar ? 2 * BN : 0), *const sh_s = sc_s + BN;
        if (ss_lds && tid < BN) {
            const int col = n0 + tid < a.n ? n0 + tid : a.n - 1;
            sc_s[tid] = a.scale[col];
            sh_s[tid] = a.shift[col];
        }
        tpar ^= 1;
        stage_store(a_r0);
        if (ASETS == 1 && nk > 1) stage_load(a_r0, KT);       // one register set: the next k-tile's loads go out right behind the split
        __syncthreads();                                      // A planes of k-tile 0 written, filter image 0 landed (vmcnt(0) ahead of the barrier)
        // one k-tile: PEND holds the loads of k-tile kt + 1 (issued a k-tile ago), FRESH receives those of kt + 2. HAS1 / HAS2 =
        // "k-tile kt + 1 / kt + 2 exists": literal constants in the steady state and in the even tail, so that the loop body is
        // straight-line code (with branches in it the compiler's counter model turns conservative and drains vmcnt(0) right
        // behind the DMA issue)
#define XB_KTILE(KTV, PEND, FRESH, HAS1, HAS2)                                                                              \
        {                                                                                                                  \
            const int kt_ = (KTV);                                                                                         \
            if (NBB == 2 && HAS1) dma_b(kt_ + 1, (kt_ + 1) & 1);      /* older than the A loads below: vmcnt(their count) covers it */ \
            if (ASETS == 2 && NBB == 2 && HAS2) stage_load(FRESH, (kt_ + 2) * KT);                                         \
            __builtin_amdgcn_sched_barrier(0);                /* loads in flight before the first MFMA */                  \
            compute(Bbuf + (NBB == 2 ? (kt_ & 1) : 0) * IMG_B);                                                            \
            xb_barrier<-1>();                                 /* everyone has read the A planes (and the filter image) of k-tile kt */ \
            __builtin_amdgcn_sched_barrier(0);                /* keep the split (and the wait for its loads) behind the MFMAs */ \
            if (NBB == 1 && HAS1) dma_b(kt_ + 1, 0);                                                                       \
            if (ASETS == 2 && NBB == 1 && HAS2) stage_load(FRESH, (kt_ + 2) * KT);                                         \
            if (a.prio) __builtin_amdgcn_s_setprio(3);        /* the split is VALU work: not behind the other workgroup's MFMA stream */ \
            if (HAS1) stage_store(PEND);                                                                                   \
            if (a.prio) __builtin_amdgcn_s_setprio(0);                                                                     \
            if (ASETS == 1 && HAS2) stage_load(PEND, (kt_ + 2) * KT);                                                      \
            if (HAS2) xb_barrier<2 * A_LD>();                 /* A planes of kt + 1 written, filter image kt + 1 landed */ \
            else xb_barrier<0>();                                                                                          \
        }
        int kt = 0;
#define XB_R1 (ASETS == 2 ? a_r1 : a_r0)
        for (; kt + 3 < nk; kt += 2) {
            XB_KTILE(kt, XB_R1, a_r0, true, true)
            XB_KTILE(kt + 1, a_r0, XB_R1, true, true)
        }
        if (nk - kt == 2) {
            XB_KTILE(kt, XB_R1, a_r0, true, false)
            XB_KTILE(kt + 1, a_r0, XB_R1, false, false)
        } else {
            for (; kt < nk; kt += 2) {                        // odd K / 32 (not in the network): at most three k-tiles, run-time conditions
                XB_KTILE(kt, XB_R1, a_r0, kt + 1 < nk, kt + 2 < nk)
                if (kt + 1 < nk) XB_KTILE(kt + 1, a_r0, XB_R1, kt + 2 < nk, kt + 3 < nk)
            }
        }
#undef XB_R1
#undef XB_KTILE

        const long cm0 = m0;
        const int cn0 = n0;
        const int nvb = vb + gridDim.x;
        const bool more = nvb < nwg;
        if (more) {
            set_tile(nvb, m0, n0);
            stage_load(a_r0, 0);
            if (ASETS == 2 && nk > 1) stage_load(a_r1, KT);
            dma_b(0, 0);                                      // both filter buffers are free after the loop's last barrier
        }

        if (a.act == MBN_ACT_RELU6 && ss_lds && cm0 + BM <= a.m && cn0 + BN <= a.n && a.fast_epi) {
            mbn_store_relu6_f32<MI, NI, 0, float>(orsrc, (unsigned)a.n, (unsigned)cm0 + wm_u, cn0 + wn_u, lane, acc, sc_s - cn0, sh_s - cn0, (unsigned)a.m, a.n);
        } else {
#pragma unroll
            for (int ni = 0; ni < NI; ni++) {
                const int col = cn0 + wn + ni * 32 + li;
                const bool cok = col < a.n;
                const int cc = cok ? col : a.n - 1;
                const float sc = a.scale ? a.scale[cc] : 1.f;
                const float sh = a.shift ? a.shift[cc] : 0.f;
#pragma unroll
                for (int mi = 0; mi < MI; mi++)
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const unsigned row = (unsigned)cm0 + wm + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        float v = fmaf(acc[mi][ni][r], sc, sh);
                        if (a.act == MBN_ACT_RELU6) v = fminf(fmaxf(v, 0.f), 6.f);
                        else if (a.act == MBN_ACT_RELU) v = fmaxf(v, 0.f);
                        // 32-bit offsets through the output's descriptor (the launcher keeps M * N * 4 below 4 GiB): no 64-bit
                        // per-lane addresses live across the tile loop (they were the kernel's only spills = its scratch segment)
                        if (cok && row < (unsigned)a.m) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), orsrc, (row * (unsigned)a.n + (unsigned)col) * 4u, 0, 0);
                    }
            }
        }
        if (!more) break;
        vb = nvb;
    }
}

template <int BM, int BN, int WM, int WN, int NP, int OCC, int NBB = 2, int ASETS = 2>
int launch_xb(XArgs &a, const mbn_call &c)
{
    constexpr int NT = 64 * (BM / WM) * (BN / WN);
    a.mt = (int)((a.m + BM - 1) / BM);
    a.nt = (a.n + BN - 1) / BN;
    const int nk = a.k / KT;
    (void)nk;
    const unsigned *img = nullptr;
    unsigned img_bytes = 0;
    const int rc = mbn_pw_emul_filter_image(c.ctx, c.stream, a.filt, a.n, a.k, BN, 0, &img, &img_bytes);
    if (rc != MBN_OK) return rc;
    a.bws = img;
    a.bws_bytes = img_bytes;
    const int lds_bytes = (3 * BM * PW + NBB * 3 * BN * PW + 4 * BN) * 4;
    int per_cu = 160 * 1024 / lds_bytes;
    if (per_cu > OCC * 4 / (NT / 64)) per_cu = OCC * 4 / (NT / 64);
    if (per_cu < 1) per_cu = 1;
    if (g_mbn_tune.misc > 0) per_cu = g_mbn_tune.misc;
    const long nwg = (long)a.mt * a.nt;
    long grid = (long)c.ctx->num_cus * per_cu;
    if (grid > nwg) grid = nwg;
    hipLaunchKernelGGL((pw_gemm_xb<BM, BN, WM, WN, NP, OCC, NBB, ASETS>), dim3((unsigned)grid), dim3(NT), 0, c.stream, a);
    return MBN_OK;
}

template <int NP>
int launch_np_b(XArgs &a, const mbn_call &c, int tile)
{
    switch (tile) {
    case 11: return launch_xb<128, 128, 64, 64, NP, 2>(a, c);     // 4 waves of 64x64, 74 KB: 2 workgroups per CU — the one shape the dispatch takes
#ifdef MBN_LAB                                                     // the shapes of profiles/r02/m_pw_emul.txt (10)-(18): slower, 12 of them spill
    case 12: return launch_xb<128, 64, 64, 32, NP, 3>(a, c);      // 4 waves of 64x32, 49 KB: 3 workgroups per CU
    case 13: return launch_xb<256, 128, 64, 64, NP, 2>(a, c);     // 8 waves of 64x64, 98 KB
    case 14: return launch_xb<128, 128, 32, 64, NP, 2>(a, c);     // 8 waves of 32x64, 74 KB, 1 workgroup per CU
    case 15: return launch_xb<128, 128, 32, 64, NP, 4>(a, c);     // 8 waves of 32x64, 2 workgroups per CU (128 VGPRs)
    case 16: return launch_xb<128, 128, 64, 64, NP, 3, 1, 1>(a, c);  // one filter buffer, one A register set, 50 KB: 3 workgroups per CU (168 VGPRs)
    case 17: return launch_xb<128, 128, 64, 64, NP, 2, 1>(a, c);  // one filter buffer at 2 workgroups per CU (A/B of the buffer alone)
    case 18: return launch_xb<64, 128, 32, 64, NP, 3, 1, 1>(a, c);   // 64-row tiles (finer tail), 4 waves of 32x64, 38 KB: 3 workgroups per CU
    case 19: return launch_xb<64, 128, 32, 64, NP, 4, 1, 1>(a, c);   // the same at 4 workgroups per CU (128 VGPRs)
    case 20: return launch_xb<64, 128, 32, 64, NP, 2, 2, 2>(a, c);   // the same with two filter buffers and two A register sets, 2 per CU
#endif
    default: return MBN_EUNSUPPORTED;
    }
}

template <int BM, int BN, int WM, int WN, int NP, int NBUF, int OCC>
void launch_x(XArgs &a, hipStream_t s, int num_cus)
{
    constexpr int NT = 64 * (BM / WM) * (BN / WN);
    a.mt = (int)((a.m + BM - 1) / BM);
    a.nt = (a.n + BN - 1) / BN;
    const int lds_bytes = NBUF * 3 * (BM + BN) * PW * 4 + 16 * BN;
    int per_cu = 160 * 1024 / lds_bytes;
    if (per_cu > OCC * 4 / (NT / 64)) per_cu = OCC * 4 / (NT / 64);
    if (per_cu > 4) per_cu = 4;
    if (g_mbn_tune.misc > 0) per_cu = g_mbn_tune.misc;
    const long nwg = (long)a.mt * a.nt;
    long grid = (long)num_cus * per_cu;
    if (grid > nwg) grid = nwg;
    hipLaunchKernelGGL((pw_gemm_x<BM, BN, WM, WN, NP, NBUF, OCC>), dim3((unsigned)grid), dim3(NT), 0, s, a);
}

template <int NP>
int launch_np(XArgs &a, hipStream_t s, int num_cus, int tile)
{
    switch (tile) {
    case 6: launch_x<128, 128, 64, 64, NP, 1, 2>(a, s, num_cus); break;     // single buffer, 50 KB: 3 workgroups per CU — the dispatch's form without a filter image
    case 7: launch_x<128, 64, 64, 32, NP, 1, 3>(a, s, num_cus); break;      // single buffer, 37 KB: 4 workgroups per CU — narrow outputs
#ifdef MBN_LAB
    case 1: launch_x<128, 128, 64, 64, NP, 2, 1>(a, s, num_cus); break;     // 4 waves of 64x64, 98 KB
    case 2: launch_x<128, 64, 64, 32, NP, 2, 2>(a, s, num_cus); break;      // 4 waves of 64x32, 73 KB: 2 workgroups per CU
    case 3: launch_x<64, 64, 32, 32, NP, 2, 3>(a, s, num_cus); break;       // 4 waves of 32x32, 49 KB: 3 workgroups per CU
    case 4: launch_x<256, 128, 64, 64, NP, 2, 2>(a, s, num_cus); break;     // 8 waves of 64x64, 146 KB
    case 5: launch_x<128, 128, 32, 64, NP, 2, 2>(a, s, num_cus); break;     // 8 waves of 32x64, 98 KB
    case 8: launch_x<256, 128, 64, 64, NP, 1, 2>(a, s, num_cus); break;     // 8 waves, single buffer, 74 KB
    case 9: launch_x<128, 128, 64, 64, NP, 1, 3>(a, s, num_cus); break;     // single buffer, 3 workgroups per CU (168 VGPRs)
#endif
    default: return MBN_EUNSUPPORTED;
    }
    return MBN_OK;
}

}   // namespace

// The pre-split filter image of pw_gemm_xb and dwpw2_x6: [n-tile][k-tile][plane][bn rows][64 B] (rows past n repeat row n-1;
// paired: rows in mbn_pair_channel order), written on `stream` by split_filter into a workspace owned by the context. One
// workspace per (filter pointer, layout): two sub-batch streams run different layers at the same time, and the same layer's
// image is rewritten with identical bytes by every launch — the filter may change between calls like any other argument.
// MBN_EUNSUPPORTED when the workspace would have to be allocated inside a stream capture.
// mbn_tune_set("pw_emul_static", 1): the caller's promise that filters change only through mbn_upload / mbn_memset / mbn_free —
// an image is then split once and reused (13 launches of ~5 us less per forward of the network) until one of those calls
// touches its filter (mbn_pw_emul_invalidate).
void mbn_pw_emul_invalidate(mbn_context *ctx, const void *dst, size_t bytes)
{
    std::lock_guard<std::mutex> g(ctx->mu);
    const uintptr_t lo = (uintptr_t)dst, hi = lo + bytes;
    for (auto &kv : ctx->emul_ws)
        if (kv.first.first < hi && kv.first.first + kv.second.src_bytes > lo) kv.second.built = false;
}

int mbn_pw_emul_filter_image(mbn_context *ctx, hipStream_t stream, const float *filt, int n, int k, int bn, int paired,
                             const unsigned **img, unsigned *bytes)
{
    if ((bn != 64 && bn != 128 && bn != 256) || k < KT || (k % KT) != 0 || n <= 0) return MBN_EUNSUPPORTED;
    const int nt = (n + bn - 1) / bn, nk = k / KT;
    const size_t need = (size_t)nt * nk * 3 * bn * PW * 4;
    if (need >= 0xFFFFFFFFull) return MBN_EUNSUPPORTED;
    void *ws = nullptr;
    const bool is_static = g_mbn_tune.pw_emul_static != 0;
    bool first_build = false;
    {
        std::lock_guard<std::mutex> g(ctx->mu);
        const std::pair<uintptr_t, int> key((uintptr_t)filt, bn * 2 + (paired ? 1 : 0));
        auto it = ctx->emul_ws.find(key);
        if (it != ctx->emul_ws.end() && it->second.bytes >= need) {
            ws = it->second.p;
            // Inside a stream capture `built` is ignored and the split is always recorded as a node of the graph: a graph that baked in
            // "already split" would keep replaying GEMMs on the old image after mbn_upload rewrote the filter (the graph's key
            // does not change with the weights). Costs the 13 split launches per replay, in graph mode only.
            hipStreamCaptureStatus cs0 = hipStreamCaptureStatusNone;
            if (is_static) (void)hipStreamIsCapturing(stream, &cs0);
            if (is_static && cs0 == hipStreamCaptureStatusNone && it->second.built && it->second.src_bytes == (size_t)n * k * 4) {      // split once, reused (pw_emul_static)
                *img = (const unsigned *)ws;
                *bytes = (unsigned)need;
                return MBN_OK;
            }
        } else {
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            (void)hipStreamIsCapturing(stream, &cs);
            if (cs != hipStreamCaptureStatusNone) return MBN_EUNSUPPORTED;
            if (it != ctx->emul_ws.end()) { (void)hipDeviceSynchronize(); (void)hipFree(it->second.p); ctx->emul_ws.erase(it); }
            if (hipMalloc(&ws, need) != hipSuccess) return MBN_EUNSUPPORTED;
            mbn_emul_img e;
            e.p = ws; e.bytes = need;
            ctx->emul_ws[key] = e;
        }
        mbn_emul_img &e = ctx->emul_ws[key];
        e.src_bytes = (size_t)n * k * 4;
        if (is_static) {
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            (void)hipStreamIsCapturing(stream, &cs);
            first_build = cs == hipStreamCaptureStatusNone;   // inside a capture the split stays a node of the graph and nothing is marked
        }
        e.built = false;
    }
    if (bn == 64) hipLaunchKernelGGL((split_filter<64>), dim3(nt, nk), dim3(256), 0, stream, (unsigned *)ws, filt, n, k, paired);
    else if (bn == 128) hipLaunchKernelGGL((split_filter<128>), dim3(nt, nk), dim3(512), 0, stream, (unsigned *)ws, filt, n, k, paired);
    else hipLaunchKernelGGL((split_filter<256>), dim3(nt, nk), dim3(1024), 0, stream, (unsigned *)ws, filt, n, k, paired);
    if (first_build) {
        // another stream may take the image next without a dependency on this one: complete it before it is marked (once per filter)
        if (hipStreamSynchronize(stream) == hipSuccess) {
            std::lock_guard<std::mutex> g(ctx->mu);
            auto it = ctx->emul_ws.find(std::pair<uintptr_t, int>((uintptr_t)filt, bn * 2 + (paired ? 1 : 0)));
            if (it != ctx->emul_ws.end() && it->second.p == ws) it->second.built = true;
        }
    }
    *img = (const unsigned *)ws;
    *bytes = (unsigned)need;
    return MBN_OK;
}

// MBN_OK if launched; MBN_EUNSUPPORTED if the mode is off or the shape is outside this kernel's envelope.
int mbn_launch_f32_pw_emul(const mbn_call &c, float *out, const float *in, const float *filt, long m, int cin, int op_size)
{
    const int np = g_mbn_tune.pw_emul;
    if (np == 0 || c.dtype != MBN_DT_F32) return MBN_EUNSUPPORTED;
#ifdef MBN_LAB
    if (np != 1 && np != 3 && np != 6 && np != 9) return MBN_EUNSUPPORTED;      // 3 / 1 products: only for the error table of tools/pw_emul_check.py
#else
    if (np != 6 && np != 9) return MBN_EUNSUPPORTED;
#endif
    if (cin < KT || (cin % KT) != 0 || m <= 0) return MBN_EUNSUPPORTED;
    if (((uintptr_t)in % 16) || ((uintptr_t)filt % 16)) return MBN_EUNSUPPORTED;
    if ((long)((m + 63) / 64) * ((op_size + 63) / 64) > 0x7fffffffL) return MBN_EUNSUPPORTED;
    XArgs a;
    a.out = out; a.in = in; a.filt = filt; a.scale = c.scale; a.shift = c.shift;
    a.m = m; a.k = cin; a.n = op_size; a.act = c.act; a.mt = a.nt = 0;
    if ((double)m * op_size * 4.0 >= 4294967296.0) return MBN_EUNSUPPORTED;      // buffer stores with 32-bit offsets: the default kernels take larger outputs
    a.fast_epi = 1;
    a.bws = nullptr; a.bws_bytes = 0;
    a.prio = (g_mbn_tune.conv_variant & 16) ? 0 : 1;          // A/B hook: conv_variant bit 4 = no raised priority over the split
    const int cus = c.ctx->num_cus;
    int tile = g_mbn_tune.pw_tile;
    if (tile == 0) {
        // fewer 128x128 tiles than CUs (the FC layer, small batches): pw_gemm's 64x64 tiles fill the chip better
        if (((m + 127) / 128) * ((op_size + 127) / 128) < cus) return MBN_EUNSUPPORTED;
        // 128-column outputs and wider: the filter pre-split once per launch and streamed by LDS-DMA (3-8 % faster than
        // splitting it per tile, profiles/r02/m_pw_emul.txt); without a workspace (first use inside a graph capture) and for
        // narrow outputs: both operands split on the way in
        if (op_size >= 128 && (np == 6 || np == 9)) {
            const int t = (g_mbn_tune.conv_variant & 32) ? 17 : 11;          // A/B hook: conv_variant bit 5 = one filter buffer
            const int rc = np == 9 ? launch_np_b<9>(a, c, t) : launch_np_b<6>(a, c, t);
            if (rc == MBN_OK) return rc;
        }
        tile = op_size >= 128 ? 6 : 7;
    }
    if (tile >= 11) {
        if (np == 9) return launch_np_b<9>(a, c, tile);
        if (np == 6) return launch_np_b<6>(a, c, tile);
        return MBN_EUNSUPPORTED;
    }
    switch (np) {
    case 9: return launch_np<9>(a, c.stream, cus, tile);
    case 6: return launch_np<6>(a, c.stream, cus, tile);
#ifdef MBN_LAB
    case 3: return launch_np<3>(a, c.stream, cus, tile);
    case 1: return launch_np<1>(a, c.stream, cus, tile);
#endif
    default: return MBN_EUNSUPPORTED;
    }
}
