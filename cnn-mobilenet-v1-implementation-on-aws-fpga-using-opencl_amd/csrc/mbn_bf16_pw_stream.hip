// mbn_bf16_pw_stream.hip — the 1x1 pointwise conv of the network's bf16 mode (kernel.cl:94-114 `pointwise`;
// out = relu6(scale * (in . filt^T) + shift), in [M][K] bf16, filt [N][K] bf16, out [M][N] bf16, fp32 accumulate on
// v_mfma_f32_32x32x16_bf16) as a GEMM whose operand stream never drains: TWO 8-wave workgroups per CU, each with a ring
// of THREE activation slots and TWO filter slots in LDS fed by buffer_load ... lds, counted s_waitcnt vmcnt(N) in front of
// a raw s_barrier, and the (tile, k-tile) sequence flattened across the persistent workgroup's tiles.
//
// Why (round 3; counters in profiles/r03/a_pmc_bf16_gemm.txt). pw_gemm<bf16> — two LDS buffers, `issue k+1 -> MFMA k ->
// vmcnt(0) -> barrier` — left the waves parked in s_waitcnt / s_barrier for 53 % of their cycles on the 512 -> 512 layers with
// the matrix pipe 42 % busy, the LDS 32 % busy, no bank conflicts and TA back-pressure at 2 %: every k-tile waited out a
// full HBM round trip with one k-tile (16 KB of activations per workgroup) in flight — 0.38 of the HBM rate two rounds
// running. Round 2's ring kernel (mbn_bf16_pw_ring.hip) had three k-tiles in flight but needed 128 KB of LDS, i.e. ONE
// workgroup per CU: two waves per SIMD cannot cover the LDS-DMA issue, the fragment latency and the barrier, and it lost
// from K = 256 up. This kernel keeps the tiled kernel's occupancy (2 x 8 waves = 4 waves per SIMD) AND has two activation
// k-tiles in flight per workgroup: the activations are the operand that comes from HBM, so they get the third slot; the
// filter (<= 2 MB, L2-resident, re-read by every m-tile) needs one k-tile of look-ahead only.
//   LDS: 3 x 16 KB (A: 128 rows x 128 B) + 2 x 16 KB (B: 128 rows x 128 B) = 80 KB = exactly half a CU's LDS; scale/shift
//   do not fit beside it, so a lane loads its four values per tile into registers at the head of the tile's last k-step,
//   AHEAD of that step's LDS-DMA (waiting for them then leaves the younger DMA in flight: in-order vmcnt).
// Step i of the flattened sequence (k-tile i lives in A slot i % 3, B slot i % 2):
//   s_waitcnt vmcnt(N) ; s_barrier      k-tile i has landed for every wave; every wave is done reading k-tile i-1
//   LDS-DMA  B(i+1) -> B slot (i+1)%2   (vacated by k-tile i-1), then A(i+2) -> A slot (i+2)%3 (vacated by k-tile i-1)
//   8 MFMAs per wave on k-tile i (fragments of group g+1 requested before the MFMAs of group g)
//   [last k-step of a tile: epilogue, 16 channel-paired 4-byte stores per lane]
// N = what was issued after B(i): the 2 pieces of A(i+1), plus the 16 stores if step i-1 ended a tile.
// Tile 128 x 128, waves 4 (m) x 2 (n) of 32 x 64; LDS image, source-side XOR swizzle, fragment reads, XCD-aware tile order
// and the channel-paired epilogue are those of pw_gemm<__bf16> (mbn_f32_pw.hip), so the two kernels return the same bits.
// Envelope: K a multiple of 64, N a multiple of 128, BN + ReLU6 epilogue, tensors < 4 GiB; everything else stays on pw_gemm.
#include "mbn_internal.h"
#include "mbn_epilogue.h"

#include <type_traits>

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2e __attribute__((ext_vector_type(2)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf2e __attribute__((ext_vector_type(2)));
typedef mbn_f16v f16v;

constexpr int BKE = 64, BKF = 32;                          // k-tile: 64 bf16 = 128-byte rows = 32 LDS words
constexpr int LDP = 2;                                     // 16-byte pieces per lane per operand per k-tile (both tile shapes)
constexpr unsigned OOB = 0xF0000000u;                      // a buffer offset past every tensor in the envelope (< 3.75 GiB): the load is dropped

// Two shapes of the same kernel:
//   <128,128,32,64,3,4>  the form described above: 8 waves, 80 KB, two workgroups per CU
//   <256,256,64,64,2,4>  "big tile" (round 3): 16 waves of 64 x 64 on a 256 x 256 tile, two slots per operand (128 KB, one workgroup per
//                        CU, still 4 waves per SIMD). A 128 x 128 tile moves (128+128) x 128 B through the CU's 64 B/clk L1 path per
//                        1024 MFMA cycles per SIMD — the path is as busy as the matrix pipe, and the K >= 512 layers sit on BOTH limits
//                        (profiles/r03/d_l2_to_cu_fill_rate.txt, b_bf16_stream_gemm.txt); 256 x 256 halves the bytes per flop.
template <int BM_, int BN_, int WM_, int WN_, int ASLOTS_>
struct Shape {
    static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_, ASLOTS = ASLOTS_, BSLOTS = 2;
    static constexpr int MI = WM / 32, NI = WN / 32, WAVES_N = BN / WN;
    static constexpr int NT = 64 * (BM / WM) * WAVES_N;
    static constexpr int AF = BM * BKF, BF_ = BN * BKF;       // floats per A / B slot
    static constexpr int NST = 16 * MI;                       // store instructions per lane per epilogue
    static constexpr int AHEAD = ASLOTS - 1;                  // k-tiles of activations in flight ahead of the compute cursor
    static_assert(NI == 2, "channel-paired epilogue: two column blocks per wave");
    static_assert(BM * 8 == LDP * NT && BN * 8 == LDP * NT, "two pieces per lane per operand");
};

#ifdef MBN_LAB
// lab: workgroup 0 / wave 0 of the last launch: { s_memtime at start, at end (core clock cycles), s_memrealtime at start, at end (100 MHz) }
__device__ unsigned long long g_stream_clk[4];
#endif

struct StreamArgs {
    __bf16 *out;
    const __bf16 *in, *filt;
    const float *scale, *shift;
    long m;
    int k, n, mt, nt;
    int delay;      // lab (exp0 >= 100): start delay of a workgroup, (exp0 - 100) x 1024 cycles x its phase ((blockIdx / 8) & 3): de-phases the tiles' store bursts
    int stag;       // lab (exp2): 1 = odd waves issue their LDS-DMA behind the second MFMA group instead of ahead of the first, 2 = all waves
};

__device__ __forceinline__ int swz(int row, int chunk) { return (row << 5) + (((chunk ^ (row >> 1)) & 7) << 2); }
__device__ __forceinline__ int xcd_remap(int vb, int nwg)
{
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = vb & 7;
    return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (vb >> 3);
}

// all but the VM_LEFT youngest vector-memory operations of this wave are done, its LDS reads are done, then s_barrier
// (asm: nothing is moved across it; __syncthreads() would drain vmcnt(0) while an LDS-DMA is pending)
template <int VM_LEFT, bool BAR = true>
__device__ __forceinline__ void stream_barrier()
{
    if (BAR) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(VM_LEFT) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(VM_LEFT) : "memory");
}

template <int NT>
__device__ __forceinline__ void dma2(__amdgpu_buffer_rsrc_t rsrc, float *slot, const unsigned (&voff)[LDP], int soff, int wave_u)
{
#pragma unroll
    for (int p = 0; p < LDP; p++)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)(slot + (p * (NT / 8) + wave_u * 8) * BKF),
                                                 16, voff[p], soff, 0, 0);
}

// ABL (lab build only; 0 in the shipped kernel): ablation bits for timing — results are wrong with any of them set.
//   1 = no LDS-DMA in the steps (the ring keeps what the prologue loaded), 2 = no fragment reads (one set read before the loop),
//   4 = no MFMAs, 8 = no barriers (the counted waits stay), 16 = no epilogue stores
// M16 (lab): the products on v_mfma_f32_16x16x32_bf16 instead of 32x32x16 — same LDS image, same bytes read per k-tile, the shape the chip holds a
// higher clock under (tools/micro/mfma_shape_clock.hip); sums 32 products per instruction, so NOT bit-identical to the 32x32x16 kernels
template <typename SH, int ABL, bool M16 = false>
__global__ __launch_bounds__(SH::NT, 4) void pw_stream_bf16(StreamArgs a)
{
    constexpr int BM = SH::BM, BN = SH::BN, WM = SH::WM, WN = SH::WN, MI = SH::MI, NI = SH::NI, WAVES_N = SH::WAVES_N, NT = SH::NT;
    constexpr int ASLOTS = SH::ASLOTS, BSLOTS = SH::BSLOTS, AF = SH::AF, BF_ = SH::BF_, NST = SH::NST, AHEAD = SH::AHEAD;
    __shared__ __attribute__((aligned(16))) float lds[ASLOTS * AF + BSLOTS * BF_];      // 81 920 bytes: two workgroups per CU (131 072: one)
    float *const Bring = lds + ASLOTS * AF;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = (wave_u / WAVES_N) * WM, wn = (wave_u % WAVES_N) * WN;
    const int li = lane & 31, lh = lane >> 5;
    const int nk = a.k / BKE, nwg = a.mt * a.nt;
    if ((int)blockIdx.x >= nwg) return;
    const int ntile = (nwg - 1 - (int)blockIdx.x) / (int)gridDim.x + 1;      // tiles of this workgroup
#ifdef MBN_LAB
    const bool clk = blockIdx.x == 0 && tid == 0;
    if (clk) { g_stream_clk[0] = __builtin_amdgcn_s_memtime(); g_stream_clk[2] = __builtin_amdgcn_s_memrealtime(); }
    if (a.delay > 0) {
        const long long t0 = __builtin_readcyclecounter(), wait = (long long)(((int)blockIdx.x >> 3) & 3) * a.delay * 1024;
        while (__builtin_readcyclecounter() - t0 < wait) __builtin_amdgcn_s_sleep(32);
    }
#endif

    const __amdgpu_buffer_rsrc_t arsrc = mbn_make_rsrc(a.in, (unsigned)(a.m * a.k * 2));
    const __amdgpu_buffer_rsrc_t brsrc = mbn_make_rsrc(a.filt, (unsigned)((long)a.n * a.k * 2));
    const __amdgpu_buffer_rsrc_t orsrc = mbn_make_rsrc(a.out, (unsigned)(a.m * a.n * 2));
    const __amdgpu_buffer_rsrc_t scrsrc = mbn_make_rsrc(a.scale, (unsigned)a.n * 4u);
    const __amdgpu_buffer_rsrc_t shrsrc = mbn_make_rsrc(a.shift, (unsigned)a.n * 4u);

    const int st_ch = tid & 7;
    int fr_a[4], fr_b[4];
#pragma unroll
    for (int g = 0; g < 4; g++) {
        fr_a[g] = swz(wm + li, 2 * g + lh);
        fr_b[g] = swz(wn + li, 2 * g + lh);
    }
    // M16: lane (r = l & 15, q = l >> 4) holds k = 32 kg + 8 q .. + 7 of row r of a 16-row block: chunk 4 kg + q of the 128-byte row
    [[maybe_unused]] int fr16_a[2], fr16_b[2];
#pragma unroll
    for (int kg = 0; kg < 2; kg++) {
        fr16_a[kg] = swz(wm + (lane & 15), 4 * kg + (lane >> 4));
        fr16_b[kg] = swz(wn + (lane & 15), 4 * kg + (lane >> 4));
    }

    // ---- issue cursors: the filter runs one k-tile ahead of the compute cursor, the activations two
    unsigned a_vo[LDP], b_vo[LDP];
    int a_vb = blockIdx.x, a_kt = 0, a_slot = 0;
    int b_vb = blockIdx.x, b_kt = 0, b_slot = 0;
    // Past the end of the workgroup's sequence the cursors keep issuing — with out-of-range offsets, which the buffer unit drops —
    // so that EVERY step issues exactly 2 + 2 LDS-DMA operations: the counted waits are then compile-time constants, for the
    // barriers here and for the compiler's own wait on the scale/shift loads (with conditional issue it fell back to vmcnt(0) there,
    // i.e. drained the ring once per tile).
    auto set_a_tile = [&](int vb) __attribute__((always_inline)) {
        if (vb >= nwg) {
#pragma unroll
            for (int p = 0; p < LDP; p++) a_vo[p] = OOB;
            return;
        }
        const long m0 = (long)(xcd_remap(vb, nwg) / a.nt) * BM;
#pragma unroll
        for (int p = 0; p < LDP; p++) {
            const int row = (p * NT + tid) >> 3;
            long gm = m0 + row;
            if (gm >= a.m) gm = a.m - 1;                                 // rows past M are computed but never stored
            a_vo[p] = ((unsigned)gm * (unsigned)a.k + (unsigned)(((st_ch ^ (row >> 1)) & 7) * 8)) * 2u;
        }
    };
    auto set_b_tile = [&](int vb) __attribute__((always_inline)) {
        if (vb >= nwg) {
#pragma unroll
            for (int p = 0; p < LDP; p++) b_vo[p] = OOB;
            return;
        }
        const int n0 = (xcd_remap(vb, nwg) % a.nt) * BN;
#pragma unroll
        for (int p = 0; p < LDP; p++) {
            const int row = (p * NT + tid) >> 3;
            const int gn = n0 + mbn_pair_channel(row);                   // channel-paired column blocks (mbn_epilogue.h)
            b_vo[p] = ((unsigned)gn * (unsigned)a.k + (unsigned)(((st_ch ^ (row >> 1)) & 7) * 8)) * 2u;
        }
    };
    auto issue_a = [&]() __attribute__((always_inline)) {
        dma2<NT>(arsrc, lds + a_slot * AF, a_vo, a_kt * BKE * 2, wave_u);
        if (++a_slot == ASLOTS) a_slot = 0;
        if (++a_kt == nk) {
            a_kt = 0;
            a_vb += gridDim.x;
            set_a_tile(a_vb);
        }
    };
    auto issue_b = [&]() __attribute__((always_inline)) {
        dma2<NT>(brsrc, Bring + b_slot * BF_, b_vo, b_kt * BKE * 2, wave_u);
        b_slot ^= 1;
        if (++b_kt == nk) {
            b_kt = 0;
            b_vb += gridDim.x;
            set_b_tile(b_vb);
        }
    };
    set_a_tile(a_vb);
    set_b_tile(b_vb);
    // prologue: B(0), A(0), A(1) — in this order, so that "everything up to B(i)" is one counted wait from the first step on
    issue_b();
#pragma unroll
    for (int q = 0; q < AHEAD; q++) issue_a();

    f16v acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; mi++)
#pragma unroll
    for (int ni = 0; ni < NI; ni++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[mi][ni][r] = 0.f;
    constexpr int MI16 = WM / 16, NI16 = WN / 16;
    [[maybe_unused]] f4 acc16[M16 ? MI16 : 1][M16 ? NI16 : 1];
    if constexpr (M16) {
#pragma unroll
        for (int i = 0; i < MI16; i++)
#pragma unroll
            for (int j = 0; j < NI16; j++) acc16[i][j] = f4{ 0.f, 0.f, 0.f, 0.f };
    }
    int cvb = blockIdx.x, cas = 0, cbs = 0;
    f4 fa_fix = f4{ 0.f, 0.f, 0.f, 0.f };
    if (ABL & 2) { stream_barrier<0>(); fa_fix = *reinterpret_cast<const f4 *>(lds + fr_a[0]); }
    bool prev_end = false;                         // step i-1 ended a tile: its 16 stores were issued after A(i+1)
    // One step. LAST (compile time) = the tile's last k-step: scale/shift loads ahead of the DMA, epilogue behind the MFMAs. The
    // sequence is written as tiles x (nk-1 plain steps + one LAST step) so that the scale/shift registers are defined and
    // consumed inside one straight-line region: as loop-carried values of a single flattened loop the compiler's counter model
    // put s_waitcnt vmcnt(0) in front of every redefinition of them, i.e. drained the ring in every step (seen in the ISA).
    auto step = [&](auto last_tag, auto late_tag) __attribute__((always_inline)) {
        constexpr bool LAST = decltype(last_tag)::value;
        constexpr bool LATE = decltype(late_tag)::value;     // this wave issues the step's LDS-DMA behind MFMA group 1 (see the launcher)
        // younger than B(i) at this point: A(i+1) (issued behind B(i) in step i-1 / the prologue), then step i-1's stores
        if (ABL & 1) stream_barrier<0, !(ABL & 8)>();
        else if (prev_end && !(ABL & 16)) stream_barrier<LDP * (AHEAD - 1) + NST, !(ABL & 8)>();
        else stream_barrier<LDP * (AHEAD - 1), !(ABL & 8)>();
        f2e sc, sh;
        [[maybe_unused]] f2e sc16[2], sh16[2];
        int n0 = 0;
        unsigned m0 = 0;
        if constexpr (LAST) {                                            // ahead of this step's DMA: see the header
            const int lid = xcd_remap(cvb, nwg);
            n0 = (lid % a.nt) * BN;
            m0 = (unsigned)(lid / a.nt) * BM;
            if constexpr (M16) {
#pragma unroll
                for (int jp = 0; jp < 2; jp++) {
                    const unsigned co16 = (unsigned)(n0 + wn + 32 * jp + 2 * (lane & 15)) * 4u;
                    sc16[jp] = __builtin_bit_cast(f2e, __builtin_amdgcn_raw_buffer_load_b64(scrsrc, co16, 0, 0));
                    sh16[jp] = __builtin_bit_cast(f2e, __builtin_amdgcn_raw_buffer_load_b64(shrsrc, co16, 0, 0));
                }
            } else {
            const unsigned co = (unsigned)(n0 + wn + 2 * li) * 4u;
            sc = __builtin_bit_cast(f2e, __builtin_amdgcn_raw_buffer_load_b64(scrsrc, co, 0, 0));
            sh = __builtin_bit_cast(f2e, __builtin_amdgcn_raw_buffer_load_b64(shrsrc, co, 0, 0));
            }
        }
        if (!(ABL & 1) && !LATE) {
            issue_b();                                                   // B(i+1)
            issue_a();                                                   // A(i+2)
        }
        const float *As = lds + cas * AF, *Bs = Bring + cbs * BF_;
        if (++cas == ASLOTS) cas = 0;
        cbs ^= 1;
        if constexpr (M16) {
            // two k-groups of 32: 2 + 4 fragment reads and 8 MFMAs (16 cycles each) per group, the reads of group 1 ahead of the MFMAs of group 0
            f4 a16[2][MI16], b16[2][NI16];
#pragma unroll
            for (int kg = 0; kg < 2; kg++) {
#pragma unroll
                for (int i = 0; i < MI16; i++) a16[kg][i] = *reinterpret_cast<const f4 *>(As + fr16_a[kg] + i * 16 * BKF);
#pragma unroll
                for (int j = 0; j < NI16; j++) b16[kg][j] = *reinterpret_cast<const f4 *>(Bs + fr16_b[kg] + j * 16 * BKF);
            }
#pragma unroll
            for (int kg = 0; kg < 2; kg++)
#pragma unroll
                for (int i = 0; i < MI16; i++)
#pragma unroll
                    for (int j = 0; j < NI16; j++)
                        acc16[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, a16[kg][i]), __builtin_bit_cast(bf8, b16[kg][j]), acc16[i][j], 0, 0, 0);
        } else {
        f4 fa[2][MI], fb[2][NI];
        if (ABL & 2) {
#pragma unroll
            for (int q = 0; q < 2; q++) {
#pragma unroll
                for (int mi = 0; mi < MI; mi++) fa[q][mi] = fa_fix;
#pragma unroll
                for (int ni = 0; ni < NI; ni++) fb[q][ni] = fa_fix;
            }
        } else {
#pragma unroll
        for (int mi = 0; mi < MI; mi++) fa[0][mi] = *reinterpret_cast<const f4 *>(As + fr_a[0] + mi * 32 * BKF);
#pragma unroll
        for (int ni = 0; ni < NI; ni++) fb[0][ni] = *reinterpret_cast<const f4 *>(Bs + fr_b[0] + ni * 32 * BKF);
        }
#pragma unroll
        for (int g = 0; g < 4; g++) {
            if (g < 3 && !(ABL & 2)) {
#pragma unroll
                for (int mi = 0; mi < MI; mi++) fa[(g + 1) & 1][mi] = *reinterpret_cast<const f4 *>(As + fr_a[g + 1] + mi * 32 * BKF);
#pragma unroll
                for (int ni = 0; ni < NI; ni++) fb[(g + 1) & 1][ni] = *reinterpret_cast<const f4 *>(Bs + fr_b[g + 1] + ni * 32 * BKF);
            }
            if (ABL & 4) {
                asm volatile("" ::"v"(fa[g & 1][0]), "v"(fb[g & 1][0]), "v"(fb[g & 1][1]));      // keep the reads alive
            } else {
#pragma unroll
            for (int mi = 0; mi < MI; mi++)
#pragma unroll
            for (int ni = 0; ni < NI; ni++)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, fa[g & 1][mi]), __builtin_bit_cast(bf8, fb[g & 1][ni]),
                                                                      acc[mi][ni], 0, 0, 0);
            }
            if (LATE && g == 1 && !(ABL & 1)) {
                __builtin_amdgcn_sched_barrier(0);
                issue_b();
                issue_a();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        }
        prev_end = LAST;
        if constexpr (LAST) {
            // channel-paired epilogue of mbn_store_relu6_bf16_pair with the lane's scale/shift in registers; rows past M are
            // dropped by the descriptor's range check (the whole offset goes through the VGPR then)
            const bool inside = (long)m0 + BM <= a.m;
            if constexpr (M16) {
                // C/D of the 16 x 16 blocks: lane (c = l & 15, q = l >> 4) holds column c, rows 4 q .. 4 q + 3. With the channel-paired filter rows, blocks j
                // and j + NI16 / 2 of a 64-column group hold the adjacent channels 2 (16 j' + c) and + 1: packed 4-byte stores, 64 bytes per row and block
                const int c16 = lane & 15, q16 = lane >> 4;
#pragma unroll
                for (int jp = 0; jp < NI16 / 2; jp++) {
                    const f2e s2 = sc16[jp], h2 = sh16[jp];
#pragma unroll
                    for (int i = 0; i < MI16; i++)
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const unsigned ro = m0 + wm + 16 * i + 4 * q16 + r;
                            const float v0 = fminf(fmaxf(fmaf(acc16[i][jp][r], s2.x, h2.x), 0.f), 6.f);
                            const float v1 = fminf(fmaxf(fmaf(acc16[i][jp + NI16 / 2][r], s2.y, h2.y), 0.f), 6.f);
                            const unsigned v = __builtin_bit_cast(unsigned, bf2e{ (__bf16)v0, (__bf16)v1 });
                            const unsigned off = (ro * (unsigned)a.n + (unsigned)(n0 + wn + 32 * jp + 2 * c16)) * 2u;
                            __builtin_amdgcn_raw_buffer_store_b32(v, orsrc, off, 0, 0);
                        }
                }
#pragma unroll
                for (int i = 0; i < MI16; i++)
#pragma unroll
                    for (int j = 0; j < NI16; j++) acc16[i][j] = f4{ 0.f, 0.f, 0.f, 0.f };
                cvb += gridDim.x;
                (void)inside;
                return;
            }
            const unsigned lane_off = ((unsigned)(4 * lh) * (unsigned)a.n + (unsigned)(2 * li)) * 2u;
#pragma unroll
            for (int mi = 0; mi < MI; mi++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const unsigned ro = m0 + wm + mi * 32 + (r & 3) + 8 * (r >> 2);
                const float v0 = fminf(fmaxf(fmaf(acc[mi][0][r], sc.x, sh.x), 0.f), 6.f);
                const float v1 = fminf(fmaxf(fmaf(acc[mi][1][r], sc.y, sh.y), 0.f), 6.f);
                const unsigned v = __builtin_bit_cast(unsigned, bf2e{ (__bf16)v0, (__bf16)v1 });
                const unsigned soff = (ro * (unsigned)a.n + (unsigned)(n0 + wn)) * 2u;
                if (ABL & 16) asm volatile("" ::"v"(v));
                else if (inside) __builtin_amdgcn_raw_buffer_store_b32(v, orsrc, lane_off, soff, 0);
                else __builtin_amdgcn_raw_buffer_store_b32(v, orsrc, lane_off + soff, 0, 0);
            }
#pragma unroll
            for (int mi = 0; mi < MI; mi++)
#pragma unroll
            for (int ni = 0; ni < NI; ni++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[mi][ni][r] = 0.f;
            cvb += gridDim.x;
        }
    };
#ifdef MBN_LAB
    if (a.stag == 2 || (a.stag == 1 && (wave_u & 1))) {
        for (int t = 0; t < ntile; t++) {
            for (int kt = 0; kt + 1 < nk; kt++) step(std::false_type{}, std::true_type{});
            step(std::true_type{}, std::true_type{});
        }
        if (clk) { g_stream_clk[1] = __builtin_amdgcn_s_memtime(); g_stream_clk[3] = __builtin_amdgcn_s_memrealtime(); }
        return;
    }
#endif
    for (int t = 0; t < ntile; t++) {
        for (int kt = 0; kt + 1 < nk; kt++) step(std::false_type{}, std::false_type{});
        step(std::true_type{}, std::false_type{});
    }
#ifdef MBN_LAB
    if (clk) { g_stream_clk[1] = __builtin_amdgcn_s_memtime(); g_stream_clk[3] = __builtin_amdgcn_s_memrealtime(); }
#endif
}

}   // namespace

typedef Shape<128, 128, 32, 64, 3> ShapeStd;
#ifdef MBN_LAB
typedef Shape<256, 256, 64, 64, 2> ShapeBig;
#endif

static bool stream_common_ok(const mbn_call &c, const void *out, const void *in, const void *filt, long m, int cin, int op_size)
{
    if (c.dtype != MBN_DT_BF16 || (c.io_flags & (MBN_IO_OUT_F32 | MBN_IO_IN_F32)) || c.act != MBN_ACT_RELU6 || !c.scale || !c.shift) return false;
    if (((uintptr_t)in % 16) != 0 || ((uintptr_t)filt % 16) != 0 || ((uintptr_t)out % 4) != 0 || ((uintptr_t)c.scale % 8) != 0 ||
        ((uintptr_t)c.shift % 8) != 0)
        return false;
    // the output bound leaves a whole 256-row tile of head room: a ragged last tile's rows past M reach the range check as 32-bit byte offsets
    // (row * N + col) * 2, which must not wrap into the first rows of the output
    if ((double)m * cin * 2 >= (double)OOB || ((double)m + 256.0) * op_size * 2 >= 4294967296.0 || (double)op_size * cin * 2 >= (double)OOB) return false;
    return true;
}

// MBN_OK when launched; MBN_EUNSUPPORTED when the shape is outside this kernel's envelope (the caller uses pw_gemm).
// m16: the products on v_mfma_f32_16x16x32_bf16 (taken for K >= 512 at EVERY M, so that a layer's kernel — and with it the bits of an image's result —
// does not depend on the batch; profiles/r03/x_bf16_mfma_shape.txt)
int mbn_launch_bf16_pw_stream(const mbn_call &c, void *out, const void *in, const void *filt, long m, int cin, int op_size, bool m16)
{
    constexpr int BM = ShapeStd::BM, BN = ShapeStd::BN, NT = ShapeStd::NT;
    if (!stream_common_ok(c, out, in, filt, m, cin, op_size)) return MBN_EUNSUPPORTED;
    if (cin < BKE || (cin % BKE) != 0 || op_size < BN || (op_size % BN) != 0 || (!m16 && m < 4 * BM) || m < 1) return MBN_EUNSUPPORTED;
    StreamArgs a;
    a.out = (__bf16 *)out; a.in = (const __bf16 *)in; a.filt = (const __bf16 *)filt; a.scale = c.scale; a.shift = c.shift;
    a.m = m; a.k = cin; a.n = op_size; { const int e2 = g_mbn_tune.exp2; a.stag = e2 >= 98 ? 0 : e2; }
    a.mt = (int)((m + BM - 1) / BM);
    a.nt = op_size / BN;
    const long nwg = (long)a.mt * a.nt;
    if (nwg > 0x7fffffffL) return MBN_EUNSUPPORTED;
    long grid = 2L * c.ctx->num_cus;                       // 80 KB of LDS: two workgroups per CU
    { const int e0 = g_mbn_tune.exp0; a.delay = e0 >= 100 ? e0 - 100 : 0; if (e0 > 0 && e0 < 100) grid = (long)e0 * c.ctx->num_cus; }      // lab: workgroups per CU of the persistent grid / start delay
    if (grid > nwg) grid = nwg;
#ifdef MBN_LAB
    switch (g_mbn_tune.exp1) {                                 // ablations (timing only)
    case 1: hipLaunchKernelGGL((pw_stream_bf16<ShapeStd, 1>), dim3((unsigned)grid), dim3(NT), 0, c.stream, a); return MBN_OK;
    case 2: hipLaunchKernelGGL((pw_stream_bf16<ShapeStd, 2>), dim3((unsigned)grid), dim3(NT), 0, c.stream, a); return MBN_OK;
    case 3: hipLaunchKernelGGL((pw_stream_bf16<ShapeStd, 3>), dim3((unsigned)grid), dim3(NT), 0, c.stream, a); return MBN_OK;
    case 4: hipLaunchKernelGGL((pw_stream_bf16<ShapeStd, 4>), dim3((unsigned)grid), dim3(NT), 0, c.stream, a); return MBN_OK;
    case 8: hipLaunchKernelGGL((pw_stream_bf16<ShapeStd, 8>), dim3((unsigned)grid), dim3(NT), 0, c.stream, a); return MBN_OK;
    case 16: hipLaunchKernelGGL((pw_stream_bf16<ShapeStd, 16>), dim3((unsigned)grid), dim3(NT), 0, c.stream, a); return MBN_OK;
    case 6: hipLaunchKernelGGL((pw_stream_bf16<ShapeStd, 6>), dim3((unsigned)grid), dim3(NT), 0, c.stream, a); return MBN_OK;
    case 7: hipLaunchKernelGGL((pw_stream_bf16<ShapeStd, 7>), dim3((unsigned)grid), dim3(NT), 0, c.stream, a); return MBN_OK;
    case 5: hipLaunchKernelGGL((pw_stream_bf16<ShapeStd, 5>), dim3((unsigned)grid), dim3(NT), 0, c.stream, a); return MBN_OK;
    default: break;
    }
#endif
    if (m16 || g_mbn_tune.misc == 16) {                          // (lab: misc = 16 forces this form on every eligible shape)
        hipLaunchKernelGGL((pw_stream_bf16<ShapeStd, 0, true>), dim3((unsigned)grid), dim3(NT), 0, c.stream, a);
        return MBN_OK;
    }
    hipLaunchKernelGGL((pw_stream_bf16<ShapeStd, 0>), dim3((unsigned)grid), dim3(NT), 0, c.stream, a);
    return MBN_OK;
}

#ifdef MBN_LAB
// LAB ONLY (profiles/r03/l_bf16_big_tile_gemm.txt: no gain where the network would use it).
// Big-tile form: 256 x 256 tiles, one 16-wave workgroup per CU, for the first `rounds` x CUs tiles of the problem (whole rounds of the
// persistent grid: a 4th round on 16 of 256 CUs would cost a whole tile time). *rows_done = the rows those tiles cover; the caller runs
// the remaining rows [*rows_done, m) through pw_gemm. MBN_EUNSUPPORTED (and *rows_done = 0) outside the envelope.
int mbn_launch_bf16_pw_big(const mbn_call &c, void *out, const void *in, const void *filt, long m, int cin, int op_size, long *rows_done)
{
    constexpr int BM = ShapeBig::BM, BN = ShapeBig::BN, NT = ShapeBig::NT;
    *rows_done = 0;
    if (!stream_common_ok(c, out, in, filt, m, cin, op_size)) return MBN_EUNSUPPORTED;
    if (cin < 2 * BKE || (cin % BKE) != 0 || (op_size % BN) != 0) return MBN_EUNSUPPORTED;
    const int nt = op_size / BN;
    const long mt = m / BM;                                 // whole row tiles only
    const long cus = c.ctx->num_cus;
    long tiles = (mt * nt / cus) * cus;                     // whole rounds
    tiles -= tiles % nt;                                    // ... and whole row tiles
    if (g_mbn_tune.exp0 == 99) tiles = mt * nt;             // lab: every whole tile, ragged last round included
    if (tiles < cus) return MBN_EUNSUPPORTED;
    StreamArgs a;
    a.out = (__bf16 *)out; a.in = (const __bf16 *)in; a.filt = (const __bf16 *)filt; a.scale = c.scale; a.shift = c.shift;
    a.delay = 0;
    a.m = (tiles / nt) * BM; a.k = cin; a.n = op_size; { const int e2 = g_mbn_tune.exp2; a.stag = e2 >= 98 ? e2 - 98 : e2; }   /* 98, 99, 100: stag 0, 1, 2 without the remainder launch */
    a.mt = (int)(tiles / nt);
    a.nt = nt;
    *rows_done = a.m;
#ifdef MBN_LAB
    switch (g_mbn_tune.exp1) {                                 // ablations (timing only)
    case 1: hipLaunchKernelGGL((pw_stream_bf16<ShapeBig, 1>), dim3((unsigned)cus), dim3(NT), 0, c.stream, a); return MBN_OK;
    case 2: hipLaunchKernelGGL((pw_stream_bf16<ShapeBig, 2>), dim3((unsigned)cus), dim3(NT), 0, c.stream, a); return MBN_OK;
    case 3: hipLaunchKernelGGL((pw_stream_bf16<ShapeBig, 3>), dim3((unsigned)cus), dim3(NT), 0, c.stream, a); return MBN_OK;
    case 4: hipLaunchKernelGGL((pw_stream_bf16<ShapeBig, 4>), dim3((unsigned)cus), dim3(NT), 0, c.stream, a); return MBN_OK;
    case 8: hipLaunchKernelGGL((pw_stream_bf16<ShapeBig, 8>), dim3((unsigned)cus), dim3(NT), 0, c.stream, a); return MBN_OK;
    case 16: hipLaunchKernelGGL((pw_stream_bf16<ShapeBig, 16>), dim3((unsigned)cus), dim3(NT), 0, c.stream, a); return MBN_OK;
    case 6: hipLaunchKernelGGL((pw_stream_bf16<ShapeBig, 6>), dim3((unsigned)cus), dim3(NT), 0, c.stream, a); return MBN_OK;
    case 7: hipLaunchKernelGGL((pw_stream_bf16<ShapeBig, 7>), dim3((unsigned)cus), dim3(NT), 0, c.stream, a); return MBN_OK;
    case 5: hipLaunchKernelGGL((pw_stream_bf16<ShapeBig, 5>), dim3((unsigned)cus), dim3(NT), 0, c.stream, a); return MBN_OK;
    default: break;
    }
#endif
    hipLaunchKernelGGL((pw_stream_bf16<ShapeBig, 0>), dim3((unsigned)cus), dim3(NT), 0, c.stream, a);
    return MBN_OK;
}

// lab diagnostic: the clock stamps of the last streaming-GEMM launch (see g_stream_clk)
extern "C" int mbn_debug_stream_clock(unsigned long long *host4)
{
    if (!host4) return MBN_EINVAL;
    if (hipDeviceSynchronize() != hipSuccess) return MBN_EDEVICE;
    return hipMemcpyFromSymbol(host4, HIP_SYMBOL(g_stream_clk), sizeof(g_stream_clk)) == hipSuccess ? MBN_OK : MBN_EDEVICE;
}
#endif
