// mbn_bf16_pw_ring.hip — the 1x1 pointwise conv in the network's bf16 mode as a STREAMING GEMM for gfx950: a 4-slot LDS ring
// fed by buffer_load ... lds with three k-tiles in flight, instead of the double-buffered tile of mbn_f32_pw.hip.
// Same contract as mbn_pointwise in bf16 mode (kernel.cl:94-114 `pointwise`; out = relu6(scale * (in . filt^T) + shift),
// in [M][K] bf16, filt [N][K] bf16, out [M][N] bf16, fp32 accumulate on v_mfma_f32_32x32x16_bf16).
//
// Why. In bf16 every pointwise layer of the network is HBM-bound by two orders of magnitude in arithmetic intensity: a
// 128x128x64 k-tile is 256 MFMA cycles per SIMD against 32 KB of operands. With two LDS buffers the loop is
//   issue k+1 -> 256 cycles of MFMA on k -> wait for k+1 -> barrier,
// so every k-tile waits out a full L2/HBM round trip with nothing else in flight: 0.36-0.42 of HBM (round 1 and the first
// r02 runs). Here a workgroup keeps THREE k-tiles (96 KB) in flight at all times: the wait in front of a barrier is a
// counted s_waitcnt vmcnt(N) that only asks for the OLDEST of them (N = the vector-memory operations issued after it: 4
// per younger k-tile, plus the 16 stores of an epilogue that was issued in between), and the LDS-DMA for k-tile i+3 is
// issued right behind the barrier of k-tile i, into the slot k-tile i-1 just vacated. The (tile, k) sequence is flattened
// across the persistent workgroup's tiles, so the ring never drains at a tile boundary.
// One 512-thread workgroup per CU (4 x 32 KB ring + 16 KB scale/shift = 144 KB LDS), tile 128 x 128, 8 waves of 32 x 64, XCD-aware tile order
// as in mbn_f32_pw.hip; LDS image, source-side XOR swizzle, fragment reads and the channel-paired 4-byte-store epilogue are
// those of pw_gemm<__bf16>. Envelope: K a multiple of 64, N a multiple of 128, BN + ReLU6 epilogue, tensors < 4 GiB —
// everything else stays on pw_gemm.
#include "mbn_internal.h"
#include "mbn_epilogue.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef mbn_f16v f16v;

constexpr int BN = 128, BKE = 64, BKF = 32;                // k-tile: 64 bf16 = 128-byte rows = 32 LDS words
constexpr int NT = 512;
constexpr int WN = 64, NI = 2, WAVES_N = BN / WN;
// Two shapes of the same kernel (template parameters BM = tile rows, AHEAD = k-tiles in flight, NSLOT = AHEAD + 1 ring slots):
//   <128, 3>: 8 waves of 32 x 64, four 32 KB slots  — shipped for K = 64 (one k-tile per tile: the ring is what pipelines
//             anything at all: layer 5 at batch 512 0.174 -> 0.115 ms);
//   <256, 2>: 8 waves of 64 x 64, three 48 KB slots — built to test whether the 32 x 64 wave tile's LDS fragment traffic is
//             what holds pw_gemm<bf16> at 0.88 PFLOP/s on the 512 -> 512 layers. It is not: 0.74 PFLOP/s there (0.0711 vs
//             0.0603 ms), 0.0458 vs 0.0401 ms on 256 -> 512, slower on the 7x7 layers; only 256 -> 256 at M = 401 k gains
//             (0.118 -> 0.0995 ms), a layer the bf16 net runs fused. Reading all fragments of a k-tile before its MFMAs instead
//             of one group ahead changed nothing (0.0711 vs 0.0686). Reachable only with tune pw_ring = 2
//             (profiles/r02/d_bf16_ring_gemm.txt). What the numbers say: at K = 512 the layer is 52.6 GFLOP against 206 MB —
//             0.040 ms at the 1.25-1.5 PFLOP/s a tuned dense bf16 GEMM sustains on this chip under DVFS and 0.041 ms at
//             5 TB/s — so both bounds sit at ~0.04 ms and pw_gemm is at 68 % of either, not at 39 % of one.
constexpr int NMAX = 1024;                                  // widest output the LDS copy of scale/shift holds (8 KB)

struct RingArgs {
    __bf16 *out;
    const __bf16 *in, *filt;
    const float *scale, *shift;
    long m;
    int k, n, mt, nt;
};

__device__ __forceinline__ int swz(int row, int chunk) { return (row << 5) + (((chunk ^ (row >> 1)) & 7) << 2); }
__device__ __forceinline__ int xcd_remap(int vb, int nwg)
{
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = vb & 7;
    return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (vb >> 3);
}

// wait until all but the VM_LEFT youngest vector-memory operations of this wave are done and its LDS traffic is done, then
// s_barrier (asm: no LDS or global access is moved across it; __syncthreads would drain vmcnt(0), see mbn_f32_dwpw2.hip)
template <int VM_LEFT>
__device__ __forceinline__ void ring_barrier()
{
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(VM_LEFT) : "memory");
}
// nd = k-tiles issued after the awaited one (0..2), ne = epilogues issued after it (0..3)
template <int NDMA, int NST>
__device__ __forceinline__ void ring_barrier_dyn(int nd, int ne)
{
    if constexpr (3 * NST + 2 * NDMA > 63) {               // the 64x64 shape: nk >= AHEAD = 2, so nd <= 1 and ne <= 1
        switch (ne * 2 + nd) {
        case 0: ring_barrier<0>(); break;
        case 1: ring_barrier<NDMA>(); break;
        case 2: ring_barrier<NST>(); break;
        default: ring_barrier<NST + NDMA>(); break;
        }
        return;
    } else
    switch (ne * 3 + nd) {
    case 0: ring_barrier<0>(); break;
    case 1: ring_barrier<NDMA>(); break;
    case 2: ring_barrier<2 * NDMA>(); break;
    case 3: ring_barrier<NST>(); break;
    case 4: ring_barrier<NST + NDMA>(); break;
    case 5: ring_barrier<NST + 2 * NDMA>(); break;
    case 6: ring_barrier<2 * NST>(); break;
    case 7: ring_barrier<2 * NST + NDMA>(); break;
    case 8: ring_barrier<2 * NST + 2 * NDMA>(); break;
    case 9: ring_barrier<3 * NST>(); break;
    case 10: ring_barrier<3 * NST + NDMA>(); break;
    default: ring_barrier<3 * NST + 2 * NDMA>(); break;
    }
}

template <int LDP>
__device__ __forceinline__ void dma_rows(__amdgpu_buffer_rsrc_t rsrc, float *lds_tile, const unsigned *voff, int soff, int wave_u)
{
#pragma unroll
    for (int p = 0; p < LDP; p++)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)(lds_tile + (p * (NT / 8) + wave_u * 8) * BKF),
                                                 16, voff[p], soff, 0, 0);
}

template <int BM, int AHEAD>
__global__ __launch_bounds__(NT) void pw_ring_bf16(RingArgs a)
{
    constexpr int NSLOT = AHEAD + 1;
    constexpr int WM = BM / 4, MI = WM / 32;                    // 4 x 2 waves
    constexpr int SLOTF = (BM + BN) * BKF;                      // floats per ring slot (32 / 48 KB)
    constexpr int LDA = BM * 8 / NT, LDB = BN * 8 / NT;         // 16-byte pieces per lane per k-tile: A (2 / 4), B (2)
    constexpr int NDMA = LDA + LDB;                             // LDS-DMA instructions per lane per k-tile
    constexpr int NST = 8 * MI * NI;                            // store instructions per lane per epilogue (channel-paired)
    static_assert(AHEAD * NST + (AHEAD - 1) * NDMA <= 63 || NST + NDMA <= 63, "vmcnt is a 6-bit field");
    __shared__ __attribute__((aligned(16))) float lds[NSLOT * SLOTF + 2 * NMAX];
    // scale | shift of all N channels, read by the epilogue with ds_read: a global load there would be the wave's youngest
    // vector-memory operation and its wait (vmcnt(0)) would drain the three k-tiles in flight once per tile
    float *const sc_s = lds + NSLOT * SLOTF, *const sh_s = sc_s + NMAX;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = (wave_u / WAVES_N) * WM, wn = (wave_u % WAVES_N) * WN;
    const int li = lane & 31, lh = lane >> 5;
    const int nk = a.k / BKE, nwg = a.mt * a.nt;
    if ((int)blockIdx.x >= nwg) return;
    for (int i = tid; i < a.n; i += NT) { sc_s[i] = a.scale[i]; sh_s[i] = a.shift[i]; }
    __syncthreads();
    const int ntile = (nwg - 1 - (int)blockIdx.x) / (int)gridDim.x + 1;      // tiles of this workgroup
    const int total = ntile * nk;                                            // flattened k-tiles

    const __amdgpu_buffer_rsrc_t arsrc = mbn_make_rsrc(a.in, (unsigned)(a.m * a.k * 2));
    const __amdgpu_buffer_rsrc_t brsrc = mbn_make_rsrc(a.filt, (unsigned)((long)a.n * a.k * 2));
    const __amdgpu_buffer_rsrc_t orsrc = mbn_make_rsrc(a.out, (unsigned)(a.m * a.n * 2));

    // staging coordinates of this lane inside a tile: piece p covers row p*64 + tid/8, 16-byte slot tid%8 (swizzle on the source)
    const int st_ch = tid & 7;
    int fr_a[4], fr_b[4];
#pragma unroll
    for (int g = 0; g < 4; g++) {
        fr_a[g] = swz(wm + li, 2 * g + lh);
        fr_b[g] = swz(wn + li, 2 * g + lh);
    }

    // ---- issue cursor (runs AHEAD k-tiles in front of the compute cursor)
    int ivb = blockIdx.x, ikt = 0, issued = 0;
    int islot = 0;                                                       // ring slot of the next k-tile to issue
    unsigned a_vo[LDA], b_vo[LDB];
    auto set_issue_tile = [&](int vb) __attribute__((always_inline)) {
        const int lid = xcd_remap(vb, nwg);
        const int n0 = (lid % a.nt) * BN;
        const long m0 = (long)(lid / a.nt) * BM;
#pragma unroll
        for (int p = 0; p < LDA; p++) {
            const int row = (p * NT + tid) >> 3;
            long gm = m0 + row;
            if (gm >= a.m) gm = a.m - 1;                                 // rows past M are computed but never stored
            a_vo[p] = ((unsigned)gm * (unsigned)a.k + (unsigned)(((st_ch ^ (row >> 1)) & 7) * 8)) * 2u;
        }
#pragma unroll
        for (int p = 0; p < LDB; p++) {
            const int row = (p * NT + tid) >> 3;
            const int gn = n0 + mbn_pair_channel(row);                   // channel-paired column blocks (mbn_epilogue.h)
            b_vo[p] = ((unsigned)gn * (unsigned)a.k + (unsigned)(((st_ch ^ (row >> 1)) & 7) * 8)) * 2u;
        }
    };
    auto issue = [&]() __attribute__((always_inline)) {                   // k-tile `issued` of the flattened sequence -> the next ring slot
        float *slot = lds + islot * SLOTF;
        if (++islot == NSLOT) islot = 0;
        dma_rows<LDA>(arsrc, slot, a_vo, ikt * BKE * 2, wave_u);
        dma_rows<LDB>(brsrc, slot + BM * BKF, b_vo, ikt * BKE * 2, wave_u);
        issued++;
        if (++ikt == nk) {
            ikt = 0;
            ivb += gridDim.x;
            if (ivb < nwg) set_issue_tile(ivb);
        }
    };
    set_issue_tile(ivb);
#pragma unroll 1
    for (int f = 0; f < AHEAD && f < total; f++) issue();

    // ---- compute cursor
    f16v acc[MI][NI];
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int mi = 0; mi < MI; mi++)
#pragma unroll
            for (int ni = 0; ni < NI; ni++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[mi][ni][r] = 0.f;
    };
    zero_acc();
    int cvb = blockIdx.x, ckt = 0, cslot = 0;
    int epi_age = 100;                     // iterations since the last epilogue was issued (only 1..3 matter)
    for (int i = 0; i < total; i++) {
        // k-tile i was issued at iteration i-AHEAD (or in the prologue); behind it: the k-tiles issued since, and the stores of the
        // epilogues of iterations i-3 .. i-1 (an iteration issues its DMA before its epilogue)
        const int nd = issued - 1 - i;
        int ne = 0;
        if (nk >= AHEAD) ne = epi_age <= AHEAD ? 1 : 0;                  // at most one tile ended in the last three iterations
        else {                                                           // short K: count tile ends among iterations i-3 .. i-1
#pragma unroll
            for (int b = 1; b <= AHEAD; b++) {
                const int j = i - b;
                if (j >= 0 && (j % nk) == nk - 1) ne++;
            }
        }
        ring_barrier_dyn<NDMA, NST>(nd, ne);
        if (issued < total) issue();                                     // into the slot k-tile i-1 vacated (everybody is past it)
        const float *As = lds + cslot * SLOTF, *Bs = As + BM * BKF;
        if (++cslot == NSLOT) cslot = 0;
        f4 fa[2][MI], fb[2][NI];
#pragma unroll
        for (int mi = 0; mi < MI; mi++) fa[0][mi] = *reinterpret_cast<const f4 *>(As + fr_a[0] + mi * 32 * BKF);
#pragma unroll
        for (int ni = 0; ni < NI; ni++) fb[0][ni] = *reinterpret_cast<const f4 *>(Bs + fr_b[0] + ni * 32 * BKF);
#pragma unroll
        for (int g = 0; g < 4; g++) {
            if (g < 3) {
#pragma unroll
                for (int mi = 0; mi < MI; mi++) fa[(g + 1) & 1][mi] = *reinterpret_cast<const f4 *>(As + fr_a[g + 1] + mi * 32 * BKF);
#pragma unroll
                for (int ni = 0; ni < NI; ni++) fb[(g + 1) & 1][ni] = *reinterpret_cast<const f4 *>(Bs + fr_b[g + 1] + ni * 32 * BKF);
            }
#pragma unroll
            for (int mi = 0; mi < MI; mi++)
#pragma unroll
                for (int ni = 0; ni < NI; ni++)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, fa[g & 1][mi]),
                                                                          __builtin_bit_cast(bf8, fb[g & 1][ni]), acc[mi][ni], 0, 0, 0);
        }
        epi_age++;
        if (++ckt == nk) {
            const int lid = xcd_remap(cvb, nwg);
            const int n0 = (lid % a.nt) * BN;
            const unsigned m0 = (unsigned)(lid / a.nt) * BM;
            if ((long)m0 + BM <= a.m) mbn_store_relu6_bf16_pair<MI, NI, 0>(orsrc, (unsigned)a.n, m0 + wm, n0 + wn, lane, acc, sc_s, sh_s);
            else mbn_store_relu6_bf16_pair<MI, NI, 1>(orsrc, (unsigned)a.n, m0 + wm, n0 + wn, lane, acc, sc_s, sh_s);
            zero_acc();
            ckt = 0;
            cvb += gridDim.x;
            epi_age = 1;
        }
    }
}

}   // namespace

// MBN_OK when launched; MBN_EUNSUPPORTED when the shape is outside the ring kernel's envelope (the caller uses pw_gemm).
int mbn_launch_bf16_pw_ring(const mbn_call &c, void *out, const void *in, const void *filt, long m, int cin, int op_size)
{
    if (c.dtype != MBN_DT_BF16 || (c.io_flags & (MBN_IO_OUT_F32 | MBN_IO_IN_F32)) || c.act != MBN_ACT_RELU6 || !c.scale || !c.shift)
        return MBN_EUNSUPPORTED;
    const bool big = cin >= 256 && g_mbn_tune.misc != 128;     // 256 x 128 tiles, 64 x 64 wave tiles (misc = 128: A/B hook, the 128-row shape)
    const int bm = big ? 256 : 128;
    if (cin < BKE || (cin % BKE) != 0 || op_size < BN || (op_size % BN) != 0 || op_size > NMAX || m < 4 * bm) return MBN_EUNSUPPORTED;
    if (((uintptr_t)in % 16) != 0 || ((uintptr_t)filt % 16) != 0 || ((uintptr_t)out % 4) != 0 || ((uintptr_t)c.scale % 8) != 0 ||
        ((uintptr_t)c.shift % 8) != 0)
        return MBN_EUNSUPPORTED;
    if ((double)m * cin * 2 >= 4294967296.0 || (double)m * op_size * 2 >= 4294967296.0 || (double)op_size * cin * 2 >= 4294967296.0)
        return MBN_EUNSUPPORTED;
    RingArgs a;
    a.out = (__bf16 *)out; a.in = (const __bf16 *)in; a.filt = (const __bf16 *)filt; a.scale = c.scale; a.shift = c.shift;
    a.m = m; a.k = cin; a.n = op_size;
    a.mt = (int)((m + bm - 1) / bm);
    a.nt = op_size / BN;
    const long nwg = (long)a.mt * a.nt;
    if (nwg > 0x7fffffffL) return MBN_EUNSUPPORTED;
    long grid = c.ctx->num_cus;                            // 136 / 152 KB of LDS: one workgroup per CU
    if (grid > nwg) grid = nwg;
#ifdef MBN_LAB
    if (big) { hipLaunchKernelGGL((pw_ring_bf16<256, 2>), dim3((unsigned)grid), dim3(NT), 0, c.stream, a); return MBN_OK; }      // pw_ring = 2 only
#else
    if (big) return MBN_EUNSUPPORTED;
#endif
    hipLaunchKernelGGL((pw_ring_bf16<128, 3>), dim3((unsigned)grid), dim3(NT), 0, c.stream, a);
    return MBN_OK;
}
