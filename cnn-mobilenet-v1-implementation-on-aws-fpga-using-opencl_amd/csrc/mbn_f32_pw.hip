// mbn_f32_pw.hip — 1x1 pointwise conv (and the FC layer) as an MFMA GEMM for gfx950, with the folded-BN scale/shift +
// ReLU/ReLU6 epilogue fused; fp32 (the metric's mode) or bf16 storage, fp32 accumulate. Replaces the arithmetic of the
// reference's `pointwise` kernel (kernel.cl:94-114; reused as FC at MobileNet.c:2681-2763).
//
//   out[m][n] = act( scale[n] * sum_k in[m][k] * filt[n][k] + shift[n] ),   m = pixel (N*H*W), n = out channel
//
// NHWC makes `in` a row-major [M][K] matrix and `out` a row-major [M][Cout] matrix with no data movement;
// `filt` keeps kernel.cl's own order [oc][ic] = [Cout][K]. Both operands are therefore K-contiguous ("NT" GEMM).
//
// MFMA, fp32: v_mfma_f32_32x32x2_f32 — fp32 products, fp32 accumulate, 64 FLOP/clk/SIMD, the fp32 matrix peak of 157.3 TFLOP/s. NOT
// bit-identical to a sequential fmaf chain over k (round 4, tools/mfma_vs_fma_chain.py, profiles/r04/d_mfma_vs_fmaf_chain.txt: against pw_generic's
// chain 26-37 % of the elements differ in the last bits, both equally far from float64): the two products of an instruction are not chained through
// single-rounding fmas. What IS fixed is the k order and the grouping, so every kernel built on this instruction with the same k order agrees bit
// for bit (pw_gemm tiles, the fused block kernels, the stem's pointwise phase). Lane l feeds A[i=l&31][k=l>>5] and B[k=l>>5][j=l&31], one float each. K order inside
// an 8-wide k-group is permuted so that ONE ds_read_b128 per lane feeds four MFMAs: lane half h=l>>5 owns
// k = 8g+4h .. 8g+4h+3 and MFMA step s consumes element s of both operands' float4 (the same k on both sides).
// MFMA, bf16: v_mfma_f32_32x32x16_bf16 — lane (r=l&31, h=l>>5) holds A[r][k=8h+j], j=0..7 = the same 16-byte chunk
// of the K-contiguous row, so the SAME LDS image (128-byte rows = 64 bf16) and the same ds_read_b128 feed one MFMA.
//
// LDS: tiles [rows][128 bytes] staged with 16-B accesses; the 16-B chunk index is XOR-swizzled with (row>>1)&7, which
// makes the ds_read_b128 of 16 consecutive rows hit 16 distinct 16-B slots of the 256-B bank row
// (SQ_LDS_BANK_CONFLICT = 0 measured). Staging is direct-to-LDS (global_load_lds_dwordx4) when K is a multiple of the
// k-tile, through registers otherwise; double-buffered over K with one barrier per k-tile.
//
// Scheduling: PERSISTENT workgroups. The grid is (workgroups that fit per CU) x (CUs); each workgroup walks the
// tile list with stride gridDim, and issues the first global loads of its NEXT tile before the epilogue stores of
// the current one, so a tile's load latency and a workgroup's launch cost are not paid once per tile.
// Tile order is XCD-aware: the virtual block id vb = blockIdx + i*gridDim keeps vb%8 = blockIdx%8 (blocks b and b+8
// share an XCD), and same-XCD ids are mapped to consecutive tiles with the n-tile index fastest, so the workgroups
// that share an A row-panel run on ONE XCD at the same time: the panel is fetched once and re-read from that L2.
//
// Measured on MI355X, fp32 (profiles/r01/gemm_ablation.txt): the bare ds_read+MFMA loop runs at 81 % of the fp32
// matrix peak (89 % at the 2.18 GHz the chip holds under this load); barriers cost nothing; register staging cost
// 11 % (-> direct-to-LDS), epilogue stores 7 %. A B-stationary "streaming" form for the K = 32/64 layers (filter
// resident in LDS, each wave loading 32-row A slices straight into MFMA operand registers, no barriers) was built,
// parity-checked and measured 25-38 % SLOWER than this tiled kernel on layers 3 and 5, so it is not shipped.
#include "mbn_internal.h"
#include "mbn_epilogue.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef mbn_f16v f16v;
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

#ifdef MBN_LAB
// lab: workgroups 0..7 (one per XCD) of the last pw_gemm launch: { s_memtime at start, at end (core clock cycles), s_memrealtime at start, at end (100 MHz) }
__device__ unsigned long long g_pw_clk[8][4];
#endif

// tune key pw_clock: core-clock cycles (s_memtime) and 100 MHz reference ticks (s_memrealtime) that workgroups 0..7 (one per XCD: the
// tiles are persistent, so each lives as long as the launch) of every pw_gemm launch were alive, accumulated; [8] = launches
__device__ unsigned long long g_pw_clk_acc[9][2];

struct PwArgs {
    void *out;
    const void *in, *filt;
    const float *scale, *shift;
    long m;
    int k, n, act;
    int mt, nt;     // tile counts
    int out_f32;    // bf16 mode: write fp32 (FC logits)
    int fast_epi;   // 0 = always the general epilogue (A/B hook: tune conv_variant=9)
    int loop2;      // 1 = software-pipelined k-loop (default); 0 = plain loop (A/B hook: tune conv_variant=8)
    int no_ss;      // 1 = epilogue reads scale/shift from global memory as in round 1 (A/B hook: tune conv_variant=7)
    int no_cw;      // 1 = __syncthreads() (vmcnt(0)) also at the first barrier behind a fast epilogue (lab A/B: exp0 = 77)
    int xn;         // XCD groups along n (1, 2 or 4): > 1 when the filter does not fit an XCD's L2 next to the streamed A panels
    int clk;        // 1 = accumulate this launch's held clock into g_pw_clk_acc (tune key pw_clock; bench.py's untimed profiled steps)
};

constexpr int BKB = 128;            // k-tile in BYTES per row (32 fp32 / 64 bf16)
constexpr int SSMAX = 1024;         // widest output whose scale/shift are staged in LDS for the epilogue (8 KB)
constexpr int BKF = BKB / 4;        // ... in 4-byte LDS words

__device__ __forceinline__ int swz(int row, int chunk) { return (row << 5) + (((chunk ^ (row >> 1)) & 7) << 2); }

// virtual block id -> logical tile id: ids that share vb%8 (one XCD) get a contiguous range of tiles (bijective for
// any tile count, cdna guide T1)
__device__ __forceinline__ int xcd_remap(int vb, int nwg)
{
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = vb & 7;
    return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (vb >> 3);
}

// LD pieces of 8 rows x 128 B per wave: buffer_load_dwordx4 ... lds writes 1 KiB linearly at the wave-uniform LDS
// address in M0; piece p of wave w covers tile rows p*(NT/8) + 8w .. +7. (A __device__ function, not a lambda: the host pass
// of hipcc silently drops the kernel's launch stub when this builtin appears inside a lambda of the kernel.)
__device__ __forceinline__ void lds_dma_rows(const void *gbase, unsigned gbytes, float *lds_tile, const unsigned *voff, int soff, int wave_u,
                                             const int LD, const int NT)
{
    const __amdgpu_buffer_rsrc_t rsrc = mbn_make_rsrc(gbase, gbytes);      // 4 SGPRs, loop-invariant
#pragma unroll
    for (int p = 0; p < LD; p++)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)(lds_tile + (p * (NT / 8) + wave_u * 8) * BKF),
                                                 16, voff[p], soff, 0, 0);
}

// ABL (lab, timing only — results are wrong with any bit set): 1 = no LDS-DMA inside the k-loop, 2 = no barriers inside the k-loop, 4 = no epilogue stores,
// 8 = no LDS fragment reads inside the k-loop
template <typename T, int BM, int BN, int WM, int WN, int NBUF, bool KFULL, bool GLDS, int ABL = 0>
__global__ __launch_bounds__(64 * (BM / WM) * (BN / WN)) void pw_gemm(PwArgs a)
{
    constexpr bool BF = sizeof(T) == 2;
    constexpr int EPC = 16 / sizeof(T);                       // elements per 16-byte chunk
    constexpr int BKE = 8 * EPC;                              // k-tile in elements (32 / 64)
    constexpr int WAVES_N = BN / WN;
    constexpr int NT = 64 * (BM / WM) * WAVES_N;              // threads per workgroup
    constexpr int MI = WM / 32, NI = WN / 32;
    constexpr bool PAIRN = BF && (NI % 2) == 0;               // bf16: channel-paired column blocks (mbn_epilogue.h): LDS filter row rho <- channel mbn_pair_channel(rho)
    constexpr int A_LD = BM * 8 / NT, B_LD = BN * 8 / NT;     // 16-B loads per thread per k-tile
    constexpr int ST = A_LD > B_LD ? A_LD : B_LD;
    static_assert(A_LD >= 1 && B_LD >= 1 && A_LD * NT == BM * 8 && B_LD * NT == BN * 8, "tile/threads mismatch");
    __shared__ __attribute__((aligned(16))) float lds[NBUF * (BM + BN) * BKF + (GLDS ? 2 * SSMAX : 0)];
    // GLDS kernels (every pointwise layer of the network): scale | shift of all N channels sit behind the tiles and the fast
    // epilogue reads them with ds_read. As global loads they were the wave's youngest vector-memory operations — younger than
    // the NEXT tile's first LDS-DMA, issued just before the epilogue — so waiting for them (in-order vmcnt) drained that DMA
    // once per tile (the same stall the stamps of the fused block kernel showed: profiles/r02/g_dwpw2_stamps.txt)
    float *const sc_s = lds + NBUF * (BM + BN) * BKF, *const sh_s = sc_s + SSMAX;
    const bool ss_lds = GLDS && a.scale && a.shift && a.n <= SSMAX && !a.no_ss;
    if (ss_lds) {
        for (int i = threadIdx.x; i < a.n; i += 64 * (BM / WM) * (BN / WN)) { sc_s[i] = a.scale[i]; sh_s[i] = a.shift[i]; }
        __syncthreads();
    }

    const T *gin = reinterpret_cast<const T *>(a.in);
    const T *gfilt = reinterpret_cast<const T *>(a.filt);
    const int nwg = a.mt * a.nt;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave / WAVES_N) * WM, wn = (wave % WAVES_N) * WN;
    const int li = lane & 31, lh = lane >> 5;
    const int nk = (a.k + BKE - 1) / BKE;

    // per-thread staging coordinates inside a tile (row, 16-B chunk): fixed for the whole kernel
    const int st_ch = tid & 7;
    int st_row[ST];
#pragma unroll
    for (int p = 0; p < ST; p++) st_row[p] = (p * NT + tid) >> 3;

    // per-lane LDS word offsets of the four fragment groups (the swizzle is an XOR: one base per group; MFMA block and
    // buffer offsets are then constants)
    int fr_a[4], fr_b[4];
#pragma unroll
    for (int g = 0; g < 4; g++) {
        fr_a[g] = swz(wm + li, 2 * g + lh);
        fr_b[g] = swz(wn + li, 2 * g + lh);
    }

    const T *a_src[A_LD];
    const T *b_src[B_LD];
    unsigned a_vo[A_LD], b_vo[B_LD];              // byte offsets for the scalar-base direct-to-LDS loads
    f4 a_reg[A_LD], b_reg[B_LD];
    const f4 zero4 = f4{ 0.f, 0.f, 0.f, 0.f };

    // Tile order. xn == 1: same-XCD workgroups (vb % 8) walk a contiguous range of tiles with the n-tile index fastest, so an
    // A row-panel is fetched once per XCD while the WHOLE filter is swept per panel — fine while the filter (N x K) fits the
    // XCD's 4 MB L2 beside the panels. xn > 1 (2 MB filters and up: the 7x7 layers): the 8 XCDs form (8 / xn) x xn groups; a
    // group owns an m-range AND an n-range, i.e. a filter slice of N/xn rows that stays L2-resident for the group's whole
    // share, at the price of each A panel being fetched by xn XCDs. Round 1 measured 4.5x the algorithmic bytes on layer 27
    // (reads 7.8x) with the single ordering: the 4 MB filter was re-fetched per sweep. Which XCD a block lands on is a
    // speed assumption only (round-robin dispatch); the mapping is a bijection for any placement.
    const int xm = 8 / a.xn;
    auto set_tile = [&](int vb, long &m0, int &n0) {
        int mtile, ntile;
        if (a.xn > 1) {
            const int x = vb & 7, j = vb >> 3, gm = x / a.xn, gn = x % a.xn;
            const int mlo = (int)((long)a.mt * gm / xm), nlo = a.nt * gn / a.xn, nn = a.nt * (gn + 1) / a.xn - nlo;
            mtile = mlo + j / nn;
            ntile = nlo + j % nn;
        } else {
            const int lid = xcd_remap(vb, nwg);
            mtile = lid / a.nt;
            ntile = lid % a.nt;
        }
        n0 = ntile * BN;
        m0 = (long)mtile * BM;
#pragma unroll
        for (int p = 0; p < A_LD; p++) {
            long gm = m0 + st_row[p];
            if (gm >= a.m) gm = a.m - 1;             // clamp: rows past M are computed but never stored
            // the pipelined loop addresses by descriptor + 32-bit offset, the other loops by pointer: set up only one of them
            if (GLDS && a.loop2) a_vo[p] = ((unsigned)gm * (unsigned)a.k + ((st_ch ^ (st_row[p] >> 1)) & 7) * EPC) * (unsigned)sizeof(T);
            else a_src[p] = gin + gm * a.k + (GLDS ? ((st_ch ^ (st_row[p] >> 1)) & 7) : st_ch) * EPC;
        }
#pragma unroll
        for (int p = 0; p < B_LD; p++) {
            int gn = n0 + (PAIRN ? mbn_pair_channel(st_row[p]) : st_row[p]);
            if (gn >= a.n) gn = a.n - 1;
            if (GLDS && a.loop2) b_vo[p] = ((unsigned)gn * (unsigned)a.k + ((st_ch ^ (st_row[p] >> 1)) & 7) * EPC) * (unsigned)sizeof(T);
            else b_src[p] = gfilt + (long)gn * a.k + (GLDS ? ((st_ch ^ (st_row[p] >> 1)) & 7) : st_ch) * EPC;
        }
    };
    auto stage_load = [&](int k0) {
        const bool ok = KFULL || (k0 + st_ch * EPC < a.k);
#pragma unroll
        for (int p = 0; p < A_LD; p++) a_reg[p] = ok ? *reinterpret_cast<const f4 *>(a_src[p] + k0) : zero4;
#pragma unroll
        for (int p = 0; p < B_LD; p++) b_reg[p] = ok ? *reinterpret_cast<const f4 *>(b_src[p] + k0) : zero4;
    };
    auto stage_store = [&](int buf) {
        float *base = lds + buf * (BM + BN) * BKF;
#pragma unroll
        for (int p = 0; p < A_LD; p++) *reinterpret_cast<f4 *>(base + swz(st_row[p], st_ch)) = a_reg[p];
#pragma unroll
        for (int p = 0; p < B_LD; p++) *reinterpret_cast<f4 *>(base + BM * BKF + swz(st_row[p], st_ch)) = b_reg[p];
    };

    // Direct-to-LDS staging (global_load_lds_dwordx4): one wave-instruction writes 1 KiB = 8 rows x 128 B linearly
    // at a wave-uniform LDS base + lane*16, so the image stays [row][slot] and the XOR swizzle is applied to the
    // per-lane SOURCE chunk instead (set_tile). No staging VGPRs, no ds_write pass.
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int wm_u = (wave_u / WAVES_N) * WM, wn_u = (wave_u % WAVES_N) * WN;      // wave-uniform copies for the epilogue
    const bool g_fast_epilogue = a.fast_epi != 0;
    const __amdgpu_buffer_rsrc_t orsrc = mbn_make_rsrc(a.out, g_fast_epilogue ? (unsigned)(a.m * a.n * (long)sizeof(T)) : 0u);
    auto stage_glds = [&](int k0, int buf) {
        float *base = lds + buf * (BM + BN) * BKF;
#pragma unroll
        for (int p = 0; p < A_LD; p++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(a_src[p] + k0),
                                             (__attribute__((address_space(3))) void *)(base + (p * (NT / 8) + wave_u * 8) * BKF),
                                             16, 0, 0);
#pragma unroll
        for (int p = 0; p < B_LD; p++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(b_src[p] + k0),
                                             (__attribute__((address_space(3))) void *)(base + BM * BKF + (p * (NT / 8) + wave_u * 8) * BKF),
                                             16, 0, 0);
    };

    // Direct-to-LDS staging, buffer form (buffer_load_dwordx4 ... lds): descriptor + per-lane 32-bit byte offset fixed for
    // the tile + scalar offset along K — no vector address arithmetic in the k-loop, and (unlike global_load_lds, which
    // LLVM tracks as a "flat" access) an outstanding load does not turn every later LDS wait into lgkmcnt(0).
    // Needs M*K*sizeof(T) and N*K*sizeof(T) below 4 GiB (mbn_launch_f32_pointwise checks).
    const unsigned a_bytes = a.loop2 ? (unsigned)(a.m * a.k * (long)sizeof(T)) : 0u;
    const unsigned b_bytes = a.loop2 ? (unsigned)((long)a.n * a.k * (long)sizeof(T)) : 0u;
    auto stage_glds2 = [&](int k0, int buf) __attribute__((always_inline)) {
        float *base = lds + buf * (BM + BN) * BKF;
        const int kb = k0 * (int)sizeof(T);                      // scalar byte offset along K
        lds_dma_rows(a.in, a_bytes, base, a_vo, kb, wave_u, A_LD, NT);
        lds_dma_rows(a.filt, b_bytes, base + BM * BKF, b_vo, kb, wave_u, B_LD, NT);
    };

    constexpr int NSTF_ = PAIRN ? 8 * MI * NI : 16 * MI * NI;      // store instructions per lane of the fast epilogue
    constexpr int NSTF = NSTF_ > 63 ? 63 : NSTF_;                  // (vmcnt is a 6-bit counter)
    bool prev_fast = false;                                        // the previous tile of this workgroup left through the fast epilogue
    long m0;
    int n0;
    int vb = blockIdx.x;
    // xn > 1: this block's XCD group has (its m-tiles) x (its n-tiles) tiles, walked by j = vb >> 3 (gridDim is a multiple of 8)
    int vb_end = nwg;
    if (a.xn > 1) {
        const int x = vb & 7, gm = x / a.xn, gn = x % a.xn;
        const int cnt = (int)((long)a.mt * (gm + 1) / xm - (long)a.mt * gm / xm) * (a.nt * (gn + 1) / a.xn - a.nt * gn / a.xn);
        vb_end = (cnt << 3) + x;                 // vb = 8 j + x < 8 cnt + x  <=>  j < cnt
    }
    if (vb >= vb_end) return;
    const bool hclk = a.clk && blockIdx.x < 8 && threadIdx.x == 0;
    unsigned long long hc0 = 0, hr0 = 0;
    if (hclk) { hc0 = __builtin_amdgcn_s_memtime(); hr0 = __builtin_amdgcn_s_memrealtime(); }
#ifdef MBN_LAB
    const bool clk = blockIdx.x < 8 && threadIdx.x == 0;
    if (clk) { g_pw_clk[blockIdx.x][0] = __builtin_amdgcn_s_memtime(); g_pw_clk[blockIdx.x][2] = __builtin_amdgcn_s_memrealtime(); }
#endif
    set_tile(vb, m0, n0);
    if (GLDS) { if (a.loop2) stage_glds2(0, 0); else stage_glds(0, 0); }
    else stage_load(0);

    for (;;) {
        f16v acc[MI][NI];
#pragma unroll
        for (int mi = 0; mi < MI; mi++)
#pragma unroll
            for (int ni = 0; ni < NI; ni++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[mi][ni][r] = 0.f;

        auto compute = [&](int buf) {
            const float *As = lds + buf * (BM + BN) * BKF;
            const float *Bs = As + BM * BKF;
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int chunk = 2 * g + lh;
                f4 av[MI], bv[NI];
#pragma unroll
                for (int mi = 0; mi < MI; mi++) av[mi] = *reinterpret_cast<const f4 *>(As + swz(wm + mi * 32 + li, chunk));
#pragma unroll
                for (int ni = 0; ni < NI; ni++) bv[ni] = *reinterpret_cast<const f4 *>(Bs + swz(wn + ni * 32 + li, chunk));
                if constexpr (BF) {
#pragma unroll
                    for (int mi = 0; mi < MI; mi++)
#pragma unroll
                        for (int ni = 0; ni < NI; ni++)
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                                __builtin_bit_cast(bf8, av[mi]), __builtin_bit_cast(bf8, bv[ni]), acc[mi][ni], 0, 0, 0);
                } else {
#pragma unroll
                    for (int s = 0; s < 4; s++)
#pragma unroll
                        for (int mi = 0; mi < MI; mi++)
#pragma unroll
                            for (int ni = 0; ni < NI; ni++)
                                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi][s], bv[ni][s], acc[mi][ni], 0, 0, 0);
                }
            }
        };

        // ---- K loop of this tile (registers — or, with GLDS, LDS buffer 0 in flight — hold its k-tile 0 on entry)
        if (!GLDS) stage_store(0);
        // with a glds outstanding hipcc emits s_waitcnt vmcnt(0) ahead of __syncthreads(): behind a tile's epilogue that also waits for every
        // store of it to be acknowledged. The fast epilogue issues exactly NSTF stores, all YOUNGER than this tile's first LDS-DMA (issued ahead of
        // them, below): a counted wait leaves them in flight for one more k-tile (vmcnt retires in order)
        if (GLDS && prev_fast) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NSTF) : "memory");
        else __syncthreads();
        if (GLDS && a.loop2) {
            // Software-pipelined form (default): the LDS fragments of group g+1 are requested before the MFMAs of group g
            // — for the last group of a k-tile that is group 0 of the next buffer, right behind the hand-over barrier — so
            // no MFMA waits for its own ds_read. The buffer index is a compile-time constant (two k-tiles per trip), which
            // turns every fragment address into per-lane base + immediate: no VALU in the loop (VALU time does not
            // overlap fp32 MFMA time on a SIMD, tools/micro/).
            f4 fa[2][MI], fb[2][NI];
            auto ldfrag_do = [&](const float *base, int g, int slot) __attribute__((always_inline)) {
#pragma unroll
                for (int mi = 0; mi < MI; mi++) fa[slot][mi] = *reinterpret_cast<const f4 *>(base + fr_a[g] + mi * 32 * BKF);
#pragma unroll
                for (int ni = 0; ni < NI; ni++) fb[slot][ni] = *reinterpret_cast<const f4 *>(base + BM * BKF + fr_b[g] + ni * 32 * BKF);
            };
            auto ldfrag = [&](const float *base, int g, int slot) __attribute__((always_inline)) {
                if (!(ABL & 8)) ldfrag_do(base, g, slot);          // ABL & 8: the fragment registers keep what they were given once, below
            };
            if (ABL & 8) { ldfrag_do(lds, 0, 0); ldfrag_do(lds, 1, 1); }
            auto mfma_group = [&](int slot) __attribute__((always_inline)) {
                if constexpr (BF) {
#pragma unroll
                    for (int mi = 0; mi < MI; mi++)
#pragma unroll
                        for (int ni = 0; ni < NI; ni++)
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                                __builtin_bit_cast(bf8, fa[slot][mi]), __builtin_bit_cast(bf8, fb[slot][ni]), acc[mi][ni], 0, 0, 0);
                } else {
#pragma unroll
                    for (int s = 0; s < 4; s++)
#pragma unroll
                        for (int mi = 0; mi < MI; mi++)
#pragma unroll
                            for (int ni = 0; ni < NI; ni++)
                                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[slot][mi][s], fb[slot][ni][s], acc[mi][ni], 0, 0, 0);
                }
            };
            // one k-tile from buffer CUR; NEXT: stage the following k-tile into the other buffer and hand over to it.
            // (CUR and NEXT are literal constants at every call site; the always_inline lambdas fold them. The barrier sits
            // between the two halves in the kernel body itself: a lambda may not call __syncthreads in the host pass.)
            auto ktile_head = [&](const int CUR, const bool NEXT, int kt) __attribute__((always_inline)) {
                const float *base = lds + CUR * (BM + BN) * BKF;
                if (NEXT && !(ABL & 1)) stage_glds2((kt + 1) * BKE, CUR ^ 1);
#pragma unroll
                for (int g = 0; g < 3; g++) {
                    ldfrag(base, g + 1, (g + 1) & 1);
                    __builtin_amdgcn_sched_barrier(0);      // keep the fragment requests ahead of the MFMAs they overlap
                    mfma_group(g & 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            auto ktile_tail = [&](const int CUR, const bool NEXT) __attribute__((always_inline)) {
                if (NEXT) ldfrag(lds + (CUR ^ 1) * (BM + BN) * BKF, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                mfma_group(1);
                __builtin_amdgcn_sched_barrier(0);
            };
            // nk is even here (mbn_launch_f32_pointwise): steady state = two k-tiles per trip, straight-line; last pair peeled.
            // Each barrier: the next buffer has landed (vmcnt(0) ahead of it) and this one is fully read.
            ldfrag(lds, 0, 0);
            int kt = 0;
            for (; kt + 2 < nk; kt += 2) {
                ktile_head(0, true, kt);
                if (!(ABL & 2)) __syncthreads();
                ktile_tail(0, true);
                ktile_head(1, true, kt + 1);
                if (!(ABL & 2)) __syncthreads();
                ktile_tail(1, true);
            }
            ktile_head(0, true, kt);
            if (!(ABL & 2)) __syncthreads();
            ktile_tail(0, true);
            ktile_head(1, false, kt + 1);
            if (!(ABL & 2)) __syncthreads();
            ktile_tail(1, false);
        } else if (GLDS) {
            for (int kt = 0; kt < nk; kt++) {
                const int cur = kt & 1;
                if (kt + 1 < nk) stage_glds((kt + 1) * BKE, cur ^ 1);
                compute(cur);
                __syncthreads();
            }
        } else if (NBUF == 1) {
            for (int kt = 0; kt < nk; kt++) {
                if (kt + 1 < nk) stage_load((kt + 1) * BKE);
                compute(0);
                if (kt + 1 < nk) {
                    __syncthreads();
                    stage_store(0);
                    __syncthreads();
                }
            }
            __syncthreads();          // WAR: the next tile's stage_store(0) overwrites what slower waves still read
        } else {
            for (int kt = 0; kt < nk; kt++) {
                const int cur = kt & 1;
                if (kt + 1 < nk) stage_load((kt + 1) * BKE);
                compute(cur);
                if (kt + 1 < nk) stage_store(cur ^ 1);
                __syncthreads();
            }
        }

        // ---- next tile: pointers + first global loads BEFORE this tile's epilogue stores
        const long cm0 = m0;
        const int cn0 = n0;
        const int nvb = vb + gridDim.x;
        const bool more = nvb < vb_end;
        if (more) {
            set_tile(nvb, m0, n0);
            if (GLDS) { if (a.loop2) stage_glds2(0, 0); else stage_glds(0, 0); }      // both buffers are free after the K loop's last barrier
            else stage_load(0);
        }
        // the counted wait at the top of the next tile (vmcnt(NSTF)) is right only if this DMA is OLDER than every epilogue store in the ISA:
        // pin the order (tests/test_host_cpu.py checks the instruction stream of the built kernel)
        __builtin_amdgcn_sched_barrier(0);

        // ---- epilogue: C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
        // fp32: one store instruction writes two 128-B row segments (full cache lines). bf16 with an even number of column
        // blocks per wave (PAIRN): the filter rows were staged channel-paired, so blocks 2t and 2t+1 hold adjacent channels
        // and mbn_store_relu6_bf16_pair writes the same two 128-B segments per instruction with packed pairs.
        prev_fast = false;
        if (!(BF && a.out_f32) && a.act == MBN_ACT_RELU6 && a.scale && a.shift && cm0 + BM <= a.m && cn0 + BN <= a.n &&
            g_fast_epilogue && (!PAIRN || (a.n & 1) == 0)) {
            // interior tile of a BN + ReLU6 layer (every pointwise layer of the network): lean stores, mbn_epilogue.h
            prev_fast = ss_lds && a.loop2 && !a.no_cw;
            const float *scp = ss_lds ? sc_s : a.scale, *shp = ss_lds ? sh_s : a.shift;
            if constexpr (PAIRN) {
                if (ss_lds) mbn_store_relu6_bf16_pair<MI, NI, 0>(orsrc, (unsigned)a.n, (unsigned)cm0 + wm_u, cn0 + wn_u, lane, acc, sc_s, sh_s);
                else mbn_store_relu6_bf16_pair<MI, NI, 0>(orsrc, (unsigned)a.n, (unsigned)cm0 + wm_u, cn0 + wn_u, lane, acc, a.scale, a.shift);
            } else {
                if (ABL & 4) { float t_ = 0.f; for (int mi = 0; mi < MI; mi++) for (int ni = 0; ni < NI; ni++) for (int r = 0; r < 16; r++) t_ += acc[mi][ni][r];
                               if (t_ == 123.456f) reinterpret_cast<float *>(a.out)[lane] = t_; }
                else if (ss_lds) mbn_store_relu6_f32<MI, NI, 0, T>(orsrc, (unsigned)a.n, (unsigned)cm0 + wm_u, cn0 + wn_u, lane, acc, sc_s, sh_s, (unsigned)a.m, a.n);
                else mbn_store_relu6_f32<MI, NI, 0, T>(orsrc, (unsigned)a.n, (unsigned)cm0 + wm_u, cn0 + wn_u, lane, acc, a.scale,
                                                       a.shift, (unsigned)a.m, a.n);
            }
            (void)scp; (void)shp;
        } else if constexpr (PAIRN) {
            // element-wise path on the channel-paired layout (ragged tiles, the FC layer's fp32 logits, no BN): rare and small
#pragma unroll
            for (int ni = 0; ni < NI; ni++) {
                const int col = cn0 + wn + (ni >> 1) * 64 + 2 * li + (ni & 1);
                const bool cok = col < a.n;
                const int cc = cok ? col : a.n - 1;
                const float sc = a.scale ? a.scale[cc] : 1.f;
                const float sh = a.shift ? a.shift[cc] : 0.f;
#pragma unroll
                for (int mi = 0; mi < MI; mi++)
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const long row = cm0 + wm + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        float v = fmaf(acc[mi][ni][r], sc, sh);
                        if (a.act == MBN_ACT_RELU6) v = fminf(fmaxf(v, 0.f), 6.f);
                        else if (a.act == MBN_ACT_RELU) v = fmaxf(v, 0.f);
                        if (cok && row < a.m) {
                            if (a.out_f32) reinterpret_cast<float *>(a.out)[row * a.n + col] = v;
                            else reinterpret_cast<__bf16 *>(a.out)[row * a.n + col] = (__bf16)v;
                        }
                    }
            }
        } else
#pragma unroll
        for (int ni = 0; ni < NI; ni++) {
            const int col = cn0 + wn + ni * 32 + li;
            const bool cok = col < a.n;
            const int cc = cok ? col : a.n - 1;
            const float sc = a.scale ? a.scale[cc] : 1.f;
            const float sh = a.shift ? a.shift[cc] : 0.f;
#pragma unroll
            for (int mi = 0; mi < MI; mi++) {
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const long row = cm0 + wm + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    float v = fmaf(acc[mi][ni][r], sc, sh);
                    if (a.act == MBN_ACT_RELU6) v = fminf(fmaxf(v, 0.f), 6.f);
                    else if (a.act == MBN_ACT_RELU) v = fmaxf(v, 0.f);
                    if (cok && row < a.m) {
                        if (!BF || a.out_f32) reinterpret_cast<float *>(a.out)[row * a.n + col] = v;
                        else reinterpret_cast<__bf16 *>(a.out)[row * a.n + col] = (__bf16)v;
                    }
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!more) break;
        vb = nvb;
    }
#ifdef MBN_LAB
    if (clk) { g_pw_clk[blockIdx.x][1] = __builtin_amdgcn_s_memtime(); g_pw_clk[blockIdx.x][3] = __builtin_amdgcn_s_memrealtime(); }
#endif
    if (hclk) {
        atomicAdd(&g_pw_clk_acc[blockIdx.x][0], __builtin_amdgcn_s_memtime() - hc0);
        atomicAdd(&g_pw_clk_acc[blockIdx.x][1], __builtin_amdgcn_s_memrealtime() - hr0);
        if (blockIdx.x == 0) atomicAdd(&g_pw_clk_acc[8][0], 1ull);
    }
}

// Fallback for K not a multiple of the 16-byte chunk or unaligned pointers: one lane per output element.
template <typename T>
__global__ __launch_bounds__(256) void pw_generic(PwArgs a)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.m * a.n) return;
    const long row = t / a.n;
    const int col = (int)(t % a.n);
    const T *ip = reinterpret_cast<const T *>(a.in) + row * a.k, *fp = reinterpret_cast<const T *>(a.filt) + (long)col * a.k;
    float acc = 0.f;
    for (int i = 0; i < a.k; i++) acc = fmaf((float)ip[i], (float)fp[i], acc);
    float v = fmaf(acc, a.scale ? a.scale[col] : 1.f, a.shift ? a.shift[col] : 0.f);
    if (a.act == MBN_ACT_RELU6) v = fminf(fmaxf(v, 0.f), 6.f);
    else if (a.act == MBN_ACT_RELU) v = fmaxf(v, 0.f);
    if (sizeof(T) == 4 || a.out_f32) reinterpret_cast<float *>(a.out)[t] = v;
    else reinterpret_cast<__bf16 *>(a.out)[t] = (__bf16)v;
}

template <typename T, int BM, int BN, int WM, int WN>
void launch_cfg(PwArgs &a, hipStream_t s, int num_cus)
{
    constexpr int NT = 64 * (BM / WM) * (BN / WN);
    constexpr int BKE = 8 * (16 / (int)sizeof(T));
    a.mt = (int)((a.m + BM - 1) / BM);
    a.nt = (a.n + BN - 1) / BN;
    const int nbuf = a.k <= BKE ? 1 : 2;
    const int lds_bytes = nbuf * (BM + BN) * BKB + (nbuf == 2 ? 2 * SSMAX * 4 : 0);   // + the staged scale/shift of the GLDS kernels
    // persistent grid: as many workgroups as are resident at once (LDS- and wave-limited)
    int per_cu = 160 * 1024 / lds_bytes;
    const int wave_cap = 32 / (NT / 64);                 // 32 waves per CU
    if (per_cu > wave_cap) per_cu = wave_cap;
    if (per_cu > 4) per_cu = 4;
    if (g_mbn_tune.misc > 0) per_cu = g_mbn_tune.misc;   // A/B hook: workgroups per CU (1000 = one tile per workgroup)
    const long nwg = (long)a.mt * a.nt;
    long grid_l = (long)num_cus * per_cu;
    // a single-k-tile problem (layer 3 in fp32) is a pure streaming kernel: one tile per workgroup measured faster
    if (grid_l > nwg || g_mbn_tune.misc >= 1000 || nbuf == 1) grid_l = nwg;
    // n-split across XCD groups (see set_tile): only with a persistent grid that is a multiple of 8 and enough tiles per group
    a.xn = 1;
    {
        const double filt_bytes = (double)a.n * a.k * sizeof(T);
        // measured (profiles/r02/c_gemm_xcd_groups.txt): layer 27 fp32 (4 MB filter) FETCH_SIZE 207 -> 105 MiB raw with 2 groups,
        // same time (MFMA-bound); layer 25 fp32 and both layers in bf16 (filters of 2 MB and less) fetch MORE with 2 groups
        // (25 -> 29, 52 -> 63 MiB) and bf16 runs 3 % slower: the n-split starts where the filter alone fills the L2
        int xn = filt_bytes >= 16.0 * 1048576 ? 4 : filt_bytes >= 4.0 * 1048576 ? 2 : 1;
        if (g_mbn_tune.pw_xn > 0) xn = g_mbn_tune.pw_xn;                  // A/B hook: 1, 2, 4 (1 = the single ordering)
        if ((xn == 2 || xn == 4) && grid_l < nwg && (grid_l % 8) == 0 && a.nt >= xn && a.mt >= 8 / xn) a.xn = xn;
    }
    const dim3 grid((unsigned)grid_l), block(NT);
    const bool kfull = (a.k % BKE) == 0;
    const bool glds = kfull && nbuf == 2 && g_mbn_tune.pw_stage != 1;   // pw_stage=1: register staging (A/B hook)
    if (nbuf == 1) {
        if (kfull) hipLaunchKernelGGL((pw_gemm<T, BM, BN, WM, WN, 1, true, false>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((pw_gemm<T, BM, BN, WM, WN, 1, false, false>), grid, block, 0, s, a);
    } else {
#ifdef MBN_LAB
        if (glds && sizeof(T) == 4 && g_mbn_tune.exp1 > 0 && ((BM == 64 && BN == 64) || (BM == 128 && BN == 128))) {
            switch (g_mbn_tune.exp1) {                                 // ablations of the fp32 GEMM (timing only): bits of ABL
#define MBN_ABL_CASE(v) case v: hipLaunchKernelGGL((pw_gemm<T, BM, BN, WM, WN, 2, true, true, v>), grid, block, 0, s, a); return;
            MBN_ABL_CASE(1) MBN_ABL_CASE(2) MBN_ABL_CASE(3) MBN_ABL_CASE(4) MBN_ABL_CASE(7) MBN_ABL_CASE(8) MBN_ABL_CASE(9) MBN_ABL_CASE(10) MBN_ABL_CASE(12) MBN_ABL_CASE(15)
#undef MBN_ABL_CASE
            default: break;
            }
        }
#endif
        if (glds) hipLaunchKernelGGL((pw_gemm<T, BM, BN, WM, WN, 2, true, true>), grid, block, 0, s, a);
#ifdef MBN_LAB
        else if (kfull) hipLaunchKernelGGL((pw_gemm<T, BM, BN, WM, WN, 2, true, false>), grid, block, 0, s, a);      // pw_stage = 1 only
#endif
        else hipLaunchKernelGGL((pw_gemm<T, BM, BN, WM, WN, 2, false, false>), grid, block, 0, s, a);
    }
}

}   // namespace

// `out`/`in`/`filt` are fp32 or bf16 according to c.dtype; scale/shift fp32. bf16 + MBN_IO_OUT_F32 writes fp32.
int mbn_launch_f32_pointwise(const mbn_call &c, void *out, const void *in, const void *filt, long m, int cin,
                             int op_size)
{
    const bool bf = c.dtype == MBN_DT_BF16;
    PwArgs a;
    a.out = out; a.in = in; a.filt = filt; a.scale = c.scale; a.shift = c.shift;
    a.m = m; a.k = cin; a.n = op_size; a.act = c.act; a.mt = a.nt = 0;
    a.out_f32 = bf && (c.io_flags & MBN_IO_OUT_F32) ? 1 : 0;
    a.loop2 = (g_mbn_tune.conv_variant == 8 || ((cin / (bf ? 64 : 32)) & 1) || (cin % (bf ? 64 : 32)) ||
               (double)m * cin * (bf ? 2 : 4) >= 4294967296.0 || (double)op_size * cin * (bf ? 2 : 4) >= 4294967296.0) ? 0 : 1;
    a.no_ss = g_mbn_tune.conv_variant == 7 ? 1 : 0;
    a.no_cw = g_mbn_tune.exp0 == 77 ? 1 : 0;
    a.clk = g_mbn_tune.pw_clock.load(std::memory_order_relaxed) ? 1 : 0;
    a.fast_epi = (g_mbn_tune.conv_variant == 9 || (double)m * op_size * 4.0 >= 4294967296.0) ? 0 : 1;   // buffer stores: < 4 GiB
    if (m <= 0 || (long)((m + 31) / 32) * ((op_size + 31) / 32) > 0x7fffffffL) return MBN_EINVAL;
    const int epc = bf ? 8 : 4;
    const bool fast = (cin % epc) == 0 && ((uintptr_t)in % 16) == 0 && ((uintptr_t)filt % 16) == 0;
    if (!fast) {
        long total = m * op_size;
        dim3 grid((unsigned)((total + 255) / 256));
        if (bf) hipLaunchKernelGGL(pw_generic<__bf16>, grid, dim3(256), 0, c.stream, a);
        else hipLaunchKernelGGL(pw_generic<float>, grid, dim3(256), 0, c.stream, a);
        return MBN_OK;
    }
    // bf16: the streaming kernel (mbn_bf16_pw_stream.hip: 3 activation + 2 filter LDS slots per workgroup, two workgroups per CU,
    // counted vmcnt) wherever the shape is inside its envelope. Lab knob pw_ring: 1 = never (the tiled pw_gemm<bf16>), 2 = round 2's
    // one-workgroup-per-CU ring kernel wherever eligible, 3 = round 2's dispatch (that ring kernel for K = 64 only), 4 = this kernel wherever eligible.
    const int ring_mode = g_mbn_tune.pw_ring;
    // opt-in: fp32 products on the bf16 matrix cores from exact operand splits (mbn_f32_pw_x6.hip; pw_emul = 6 | 9)
    if (!bf && g_mbn_tune.pw_emul != 0 && mbn_launch_f32_pw_emul(c, (float *)out, (const float *)in, (const float *)filt, m, cin, op_size) == MBN_OK)
        return MBN_OK;
    // fp32, few tiles (batch 1..4): K split over the waves of a 16x16-tile workgroup (mbn_f32_pw_splitk.hip)
    if (!bf && g_mbn_tune.pw_tile == 0 && mbn_launch_f32_pw_splitk(c, (float *)out, (const float *)in, (const float *)filt, m, cin, op_size) == MBN_OK)
        return MBN_OK;
#ifdef MBN_LAB
    // LAB ONLY (measured slower than pw_gemm<bf16>, profiles/r03/e_bf16_wide_gemm.txt): wide layers with a packed filter image behind
    // the plain filter (MBN_IO_FILT_PACKED): 196 x 256 tiles, filter straight into registers, activations through a 3-slot LDS ring,
    // deferred epilogue (mbn_bf16_pw_wide.hip). pw_ring = 6 selects it.
    if (bf && (c.io_flags & MBN_IO_FILT_PACKED) && ring_mode == 6 && g_mbn_tune.pw_tile == 0 &&
        mbn_launch_bf16_pw_wide(c, out, in, (const char *)filt + mbn_packed_filter_offset(op_size, cin), m, cin, op_size) == MBN_OK)
        return MBN_OK;
#endif
#ifdef MBN_LAB
    // big-tile form of the streaming kernel (256 x 256, one 16-wave workgroup per CU) for whole rounds of the grid; the rows it leaves
    // (less than one round) go through this function again and land on pw_gemm. Lab knob pw_ring = 7: wherever eligible.
    if (bf && ring_mode == 7 && g_mbn_tune.pw_tile == 0) {
        long done = 0;
        if (mbn_launch_bf16_pw_big(c, out, in, filt, m, cin, op_size, &done) == MBN_OK) {
            if (done < m && g_mbn_tune.exp2 < 98)                                    // lab exp2 >= 98: the big-tile launch alone (timing)
                return mbn_launch_f32_pointwise(c, (char *)out + done * op_size * 2, (const char *)in + done * cin * 2, filt, m - done, cin, op_size);
            return MBN_OK;
        }
    }
#endif
#ifdef MBN_LAB
    // LAB ONLY (round 6; measured equal to slower than the M16 streaming form below: layer 15 0.060-0.065 against 0.063 ms, layer 25 0.039 against 0.035,
    // profiles/r06/u_*): K = 512 with the filter in registers for the whole launch (mbn_bf16_pw_rf.hip), only the activations pass the LDS. Same bits as the
    // M16 streaming form. pw_ring = 8 selects it wherever eligible.
    if (bf && ring_mode == 8 && g_mbn_tune.pw_tile == 0 && mbn_launch_bf16_pw_rf(c, out, in, filt, m, cin, op_size) == MBN_OK)
        return MBN_OK;
#endif
    // K >= 256: the streaming kernel with its products on v_mfma_f32_16x16x32_bf16 — the shape the chip holds a 16 % higher clock under: layers 15 / 25 / 27 at
    // batch 512 -4.6 % / -3.5 % / -10.7 % against pw_gemm<bf16> (profiles/r03/x_bf16_mfma_shape.txt). It sums 32 products per instruction, so its bits are
    // not those of the 32x32x16 kernels: it is taken for these layers at EVERY M (the choice depends on K and N alone), which keeps an image's result
    // independent of the batch. Round 4 (profiles/r04/c_bf16_mfma_shape_pointwise.txt, batch 512): K = 256 joins — 1.0x224 layer 13 0.0368 (pw_gemm) -> 0.0363 ms,
    // 0.5x160 layers 15-23 (256 -> 256) 0.0191 (32x32x16 streaming form) -> 0.0173, layer 25 (256 -> 512) 0.0122 (pw_gemm) -> 0.0108; for K <= 128 the 16x16x32
    // form is 4-16 % SLOWER than the 32x32x16 streaming form (two k-groups of 32 leave a 64-deep k-tile nothing to overlap) and stays off. Lab: pw_ring = 1 never.
    if (bf && (ring_mode == 0 || ring_mode == 5 || ring_mode == 8) && cin >= 256 && g_mbn_tune.pw_tile == 0 && mbn_launch_bf16_pw_stream(c, out, in, filt, m, cin, op_size, true) == MBN_OK)
        return MBN_OK;
    // measured per layer at batch 512 (profiles/r03/b_bf16_stream_gemm.txt, same call, against pw_gemm<bf16>): K = 64 0.179 -> 0.116 ms,
    // K = 128 0.080 -> 0.076, K = 256 with N = 256 0.114 -> 0.100; K = 256 with N = 512 and every K >= 512 layer 0-5 % SLOWER (there the
    // L2 -> LDS operand stream of a 128 x 128 tile, not the look-ahead, is the limit: ablation in the same file) -> pw_gemm keeps those.
    // pw_ring = 4 (lab): the streaming kernel wherever eligible.
    if (bf && (ring_mode == 4 || ((ring_mode == 0 || ring_mode == 6) && (cin <= 128 || (cin <= 256 && op_size <= 256)))) && g_mbn_tune.pw_tile == 0 &&
        mbn_launch_bf16_pw_stream(c, out, in, filt, m, cin, op_size) == MBN_OK)
        return MBN_OK;
#ifdef MBN_LAB
    if (bf && ring_mode >= 2 && (ring_mode == 2 || cin == 64) && g_mbn_tune.pw_tile == 0 &&
        mbn_launch_bf16_pw_ring(c, out, in, filt, m, cin, op_size) == MBN_OK)
        return MBN_OK;
#endif
    // round 6: short K (Cin 64 / 128 / 256) with BN + ReLU6 on the wave-private GEMM with the filter slice resident in LDS (mbn_f32_pw3.hip; same
    // bits as pw_gemm, so the choice may depend on M). pw_tile = 9: wherever eligible; 10: never; 0: where it measured faster (MBN_PW3_DEFAULT).
    if (!bf && (g_mbn_tune.pw_tile == 9 || (g_mbn_tune.pw_tile == 0 && MBN_PW3_DEFAULT(m, cin, op_size, c.ctx->num_cus))) &&
        mbn_launch_f32_pw3(c, (float *)out, (const float *)in, (const float *)filt, m, cin, op_size) == MBN_OK)
        return MBN_OK;
    // Tile choice measured per layer on MI355X in fp32 with the software-pipelined loop (tools/layer_bench.py --tune
    // pw_tile=1..8, profiles/r01/e_gemm_tile_sweep_pipelined.txt): 64x64 tiles at 4 workgroups per CU are best from
    // K = 512 up and for K = 256 with wide outputs (133 TFLOP/s = 85 % of the fp32 matrix peak on the 512 -> 512 layers),
    // 8 waves of 32x64 on a 128x128 tile win slightly for K <= 256, <128,64> in between; small grids take 64x64 too.
    int tile = g_mbn_tune.pw_tile;
    if (tile == 9 || tile == 10) tile = 0;                                 // (pw3 switches: this kernel's own rule)
    // a forced shape this kernel does not have in this build (the value may be meant for the pw_emul kernels, which fell through to
    // here because the layer is outside their envelope): the per-layer rule decides
    if (!MBN_LAB_BUILD && tile != 2 && tile != 3 && tile != 5) tile = 0;
    if (tile == 0) {
        const long big_tiles = ((m + 127) / 128) * ((op_size + 63) / 64);
        if (big_tiles < 2L * c.ctx->num_cus || op_size <= 64 || m <= 16384) tile = 3;   // 7x7 layers: finer tiles balance better
        else if (!bf && (cin >= 512 || (cin >= 256 && op_size >= 512))) tile = 3;
        else if ((bf || cin <= 256) && op_size >= 128) tile = 5;   // bf16: 128x128 wins at every K (HBM-bound, fewest B re-reads)
        else tile = 2;
    }
    const int cus = c.ctx->num_cus;
    if (bf) {                                                              // bf16: the shipped shapes (3, 5) + A/B shapes (profiles/r02/d_bf16_ring_gemm.txt: 5 wins everywhere)
        switch (tile) {
        case 3:                                                                  // small problems
            if (op_size >= 128) launch_cfg<__bf16, 64, 128, 32, 64>(a, c.stream, cus);   // 4 waves of 32x64: even NI, channel-paired 4-byte stores
            else launch_cfg<__bf16, 64, 64, 32, 32>(a, c.stream, cus);                  // narrow outputs (alpha < 1 early layers): a 128-column tile would idle
            break;
        case 5: launch_cfg<__bf16, 128, 128, 32, 64>(a, c.stream, cus); break;
        case 2: launch_cfg<__bf16, 128, 64, 64, 32>(a, c.stream, cus); break;
#ifdef MBN_LAB                                                                   // shapes of the tile sweep (profiles/r02/d_bf16_ring_gemm.txt: 5 wins everywhere)
        case 1: launch_cfg<__bf16, 128, 128, 64, 64>(a, c.stream, cus); break;   // 4 waves of 64x64
        case 4: launch_cfg<__bf16, 256, 128, 64, 64>(a, c.stream, cus); break;   // 8 waves of 64x64, 96 KB
        case 7: launch_cfg<__bf16, 128, 64, 32, 32>(a, c.stream, cus); break;    // 8 waves of 32x32, 48 KB LDS: 3 WG = 24 waves per CU
        case 8: launch_cfg<__bf16, 64, 128, 32, 32>(a, c.stream, cus); break;
        case 9: launch_cfg<__bf16, 256, 256, 64, 64>(a, c.stream, cus); break;   // r3: 16 waves of 64x64, 136 KB LDS: half the L2 -> LDS bytes per flop of 128x128
        case 10: launch_cfg<__bf16, 256, 256, 128, 64>(a, c.stream, cus); break; // r3: 8 waves of 128x64
#endif
        default: return MBN_EUNSUPPORTED;
        }
        return MBN_OK;
    }
    switch (tile) {
    case 2: launch_cfg<float, 128, 64, 64, 32>(a, c.stream, cus); break;    // 4 waves, 48 KB LDS, 3 WG/CU
    case 3: launch_cfg<float, 64, 64, 32, 32>(a, c.stream, cus); break;     // 4 waves, small problems
    case 5: launch_cfg<float, 128, 128, 32, 64>(a, c.stream, cus); break;   // 8 waves of 32x64, 2 WG/CU
#ifdef MBN_LAB                                                              // shapes of the tile sweep (profiles/r01/e_gemm_tile_sweep_pipelined.txt)
    case 1: launch_cfg<float, 128, 128, 64, 64>(a, c.stream, cus); break;   // 4 waves, 64 KB LDS, 2 WG/CU
    case 4: launch_cfg<float, 256, 128, 64, 64>(a, c.stream, cus); break;   // 8 waves, 96 KB LDS, 1 WG/CU
    case 6: launch_cfg<float, 128, 256, 64, 64>(a, c.stream, cus); break;   // 8 waves, 96 KB LDS
    case 7: launch_cfg<float, 64, 128, 32, 64>(a, c.stream, cus); break;    // 4 waves, 48 KB LDS, 3 WG/CU
    case 8: launch_cfg<float, 128, 64, 32, 64>(a, c.stream, cus); break;    // 4 waves of 32x64
    case 9: launch_cfg<float, 64, 128, 32, 32>(a, c.stream, cus); break;    // r4: 8 waves of 32x32 on 64x128: the shipped wave tile, 25 % fewer staged bytes per flop, 2 WG/CU
    case 10: launch_cfg<float, 128, 64, 32, 32>(a, c.stream, cus); break;   // r4: the same on 128x64
#endif
    default: return MBN_EUNSUPPORTED;
    }
    return MBN_OK;
}

extern "C" int mbn_pw_clock_read(mbn_context *ctx, int reset, double *ghz, long long *launches)
{
    if (!ctx || !ghz) return MBN_EINVAL;
    unsigned long long h[9][2];
    MBN_HIP_TRY(ctx, hipSetDevice(ctx->device));
    MBN_HIP_TRY(ctx, hipDeviceSynchronize());
    MBN_HIP_TRY(ctx, hipMemcpyFromSymbol(h, HIP_SYMBOL(g_pw_clk_acc), sizeof(h)));
    double core = 0, real = 0;
    for (int i = 0; i < 8; i++) { core += (double)h[i][0]; real += (double)h[i][1]; }
    *ghz = real > 0 ? core / (real / 100.0e6) / 1.0e9 : 0.0;       // s_memrealtime counts at 100 MHz on gfx950
    if (launches) *launches = (long long)h[8][0];
    if (reset) {
        memset(h, 0, sizeof(h));
        MBN_HIP_TRY(ctx, hipMemcpyToSymbol(HIP_SYMBOL(g_pw_clk_acc), h, sizeof(h)));
    }
    return MBN_OK;
}

#ifdef MBN_LAB
// lab diagnostic: the clock stamps of the last pw_gemm launch (see g_pw_clk)
extern "C" int mbn_debug_pw_clock(unsigned long long *host32)
{
    unsigned long long *host4 = host32;
    if (!host4) return MBN_EINVAL;
    if (hipDeviceSynchronize() != hipSuccess) return MBN_EDEVICE;
    return hipMemcpyFromSymbol(host4, HIP_SYMBOL(g_pw_clk), sizeof(g_pw_clk)) == hipSuccess ? MBN_OK : MBN_EDEVICE;
}
#endif
