// mbn_bf16_pw_wide.hip — the 1x1 pointwise conv of the network's bf16 mode (kernel.cl:94-114 `pointwise`;
// out = relu6(scale * (in . filt^T) + shift), in [M][K] bf16, out [M][N] bf16, fp32 accumulate on v_mfma_f32_32x32x16_bf16)
// for the wide layers (N a multiple of 256, K a multiple of 128): 196 x 256 output tiles, the filter read straight into MFMA
// operand registers from a pre-packed image, the activations through a 3-slot LDS ring.
//
// Why this shape (round 3; profiles/r03/b_bf16_stream_gemm.txt, d_l2_to_cu_fill_rate.txt). With 128 x 128 tiles both operands
// go through LDS and the kernel moves 925 MB from L2 into the CUs for a 512 -> 512 layer at batch 512 (each activation row
// panel is staged by 4 workgroups, the filter by 784): the ablation of the streaming kernel runs 47 us with NO arithmetic
// at all, every step waiting out the L2 latency of a filter k-tile that only one LDS slot can prefetch. What bounds the
// operand stream is bytes in flight per CU = LDS capacity. So:
//   * the FILTER never touches LDS. A workgroup is 8 waves side by side along N (32 columns each); a wave needs, per
//     k-tile of 64, exactly 32 columns x 64 k = 4 KB of filter, which nobody else in the workgroup needs. mbn_pack_filter_bf16
//     lays the filter out in MFMA B-operand order [n-tile][k-tile][wave][k16 step][lane][8 bf16], so those 4 KB are four
//     fully coalesced 1-KB wave loads into 16 VGPRs, one k-tile ahead — registers, not LDS, hold the filter bytes in flight.
//   * the ACTIVATIONS get all of the LDS: a ring of three 32-KB slots (224 rows x 128 B per k-tile), two k-tiles in flight.
//     Every wave reads all 7 row blocks of a slot (1 KB of LDS per MFMA, half the LDS rate at full matrix rate).
//   * 196 rows per tile: M = images x 49 x 4^j for every MobileNet map at 224 input, so 196-row tiles (one 14x14 image, four
//     7x7 images) cut M without a remainder and the tile count is a multiple of the CU count at batch 256 / 512 — a 256-row
//     tile leaves 784 tiles for 256 CUs (3.06 rounds: the fourth round is 23 % of the time). The price is 7 MFMA row blocks
//     for 6.125 blocks of rows (rows 196..223 of a tile are computed and dropped): 12.5 % matrix work, on layers whose bound
//     is the operand stream and HBM, not the matrix pipe.
// L2 -> CU bytes of the 512 -> 512 layer: 231 MB activations (2 n-tiles x 224/196) + 268 MB filter = 499 MB instead of 925.
// One workgroup per CU (96 KB LDS, ~220 VGPRs: two waves per SIMD). Step i of the flattened (tile, k-tile) sequence:
//   s_waitcnt vmcnt(4 [+56 after an epilogue]) ; s_barrier      A(i) landed for every wave, everybody done with A(i-1)
//   7 fragment reads of k16 step 0                               (their latency is covered by the issue of:)
//   filter k-tile i+1 -> the other register set (4 loads) ; LDS-DMA A(i+2) -> slot (i+2) % 3 (4 pieces per wave)
//   4 x { 7 fragment reads of the next k16 step ; 7 MFMAs }
//   [last k-step of the tile: epilogue]
// The tile is unrolled over its K/64 steps (template NK = 4, 8, 16) so that every counted wait of the compiler's own model
// (filter loads, scale/shift) is exact: with the epilogue as a branch of a flattened loop it waited for the 56 stores.
// Epilogue: a lane holds ONE output channel (col = lane & 31) of 16 rows per block. Adjacent lanes exchange half their
// rows by DPP (quad_perm 1,0,3,2) so that every lane packs two adjacent channels of 8 rows into 4-byte stores: 8 stores per
// block, 64 contiguous bytes per row and half-wave.
// Same sums in the same order as pw_gemm<bf16> / pw_stream_bf16 (k ascending in 16-groups per accumulator): same bits.
#include "mbn_internal.h"
#include "mbn_epilogue.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf2e __attribute__((ext_vector_type(2)));
typedef mbn_f16v f16v;

constexpr int TM = 196;                 // rows of a tile that are stored
constexpr int MI = 7;                   // MFMA row blocks computed (224 rows)
constexpr int BN = 256;                 // 8 waves x 32 columns
constexpr int BKE = 64, BKF = 32;       // k-tile: 64 bf16 = 128-byte rows = 32 LDS words
constexpr int NT = 512;
constexpr int SLOT_ROWS = 256;          // 32 pieces of 8 rows per slot; pieces of rows >= 224 are issued out of range (dropped)
constexpr int AF = SLOT_ROWS * BKF;     // floats per slot (32 KB)
constexpr int ASLOTS = 3;                 // ring slots: the activations run 2 k-tiles ahead of the compute cursor, the filter 1 (two register sets)
constexpr unsigned OOB = 0xF0000000u;   // past every tensor in the envelope (< 3.75 GiB): the buffer unit drops the access

struct WideArgs {
    __bf16 *out;
    const __bf16 *in;
    const void *fpk;                    // packed filter image (mbn_pack_filter_bf16)
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

template <int VM_LEFT, bool BAR = true>
__device__ __forceinline__ void wide_barrier()
{
    if (BAR) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(VM_LEFT) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(VM_LEFT) : "memory");
}

// ---- filter packing: [N][K] bf16 -> [N/256][K/64][wave 8][k16 step 4][lane 64][8 bf16]; lane (c = l & 31, h = l >> 5) holds
// filt[n0 + 32 w + c][k0 + 16 s + 8 h + j], j = 0..7: the B operand of v_mfma_f32_32x32x16_bf16 for the wave's 32 columns.
__global__ __launch_bounds__(256) void pack_filter_bf16(u4v *__restrict__ dst, const __bf16 *__restrict__ src, int n, int k)
{
    const long t = (long)blockIdx.x * 256 + threadIdx.x;          // one 16-byte chunk per thread
    const long total = (long)n * k / 8;
    if (t >= total) return;
    const int lane = (int)(t & 63);
    long q = t >> 6;
    const int s = (int)(q & 3);
    q >>= 2;
    const int w = (int)(q & 7);
    q >>= 3;
    const int nk = k / BKE;
    const int kt = (int)(q % nk), nt = (int)(q / nk);
    const int col = nt * BN + w * 32 + (lane & 31), kk = kt * BKE + s * 16 + (lane >> 5) * 8;
    dst[t] = *reinterpret_cast<const u4v *>(src + (long)col * k + kk);
}

// ABL (lab build; 0 in the shipped kernel): timing ablations, results are wrong with any bit set —
//   1 no LDS-DMA in the steps, 2 no filter loads in the steps, 4 no MFMAs, 8 no fragment reads, 16 no epilogue stores, 32 no barrier
template <int NK, int ABL>
__global__ __launch_bounds__(NT, 2) void pw_wide_bf16(WideArgs a)
{
    __shared__ __attribute__((aligned(16))) float lds[ASLOTS * AF];      // 98 304 bytes

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int nwg = a.mt * a.nt;
    if ((int)blockIdx.x >= nwg) return;
    const int ntile_cnt = (nwg - 1 - (int)blockIdx.x) / (int)gridDim.x + 1;      // tiles of this workgroup

    const __amdgpu_buffer_rsrc_t arsrc = mbn_make_rsrc(a.in, (unsigned)(a.m * a.k * 2));
    const __amdgpu_buffer_rsrc_t brsrc = mbn_make_rsrc(a.fpk, (unsigned)((long)a.n * a.k * 2));
    const __amdgpu_buffer_rsrc_t orsrc = mbn_make_rsrc(a.out, (unsigned)(a.m * a.n * 2));
    const __amdgpu_buffer_rsrc_t scrsrc = mbn_make_rsrc(a.scale, (unsigned)a.n * 4u);
    const __amdgpu_buffer_rsrc_t shrsrc = mbn_make_rsrc(a.shift, (unsigned)a.n * 4u);

    // fragment addresses: row = mi * 32 + li, chunk = 2 s + lh; (row >> 1) & 7 does not depend on mi
    int fr[4];
#pragma unroll
    for (int s = 0; s < 4; s++) fr[s] = swz(li, 2 * s + lh);

    // ---- tile descriptors of the issue cursors. Piece q of wave w covers slot rows 8 (w + 8 q) .. + 7: lane l loads row
    // 8 p + (l >> 3), source chunk (l & 7) ^ ((row >> 1) & 7) (the swizzle goes on the source: the LDS image is lane-linear).
    // Past the end of the sequence, and for slot rows >= 224, the offset is out of range: the piece is dropped, every step
    // still issues exactly 4 + 4 operations, and every counted wait is a constant.
    unsigned a_vo[4];                       // activation offsets of the tile the A cursor is in
    int a_vb = blockIdx.x, a_kt = 0, a_slot = 0;
    int b_vb = blockIdx.x, b_kt = 0;
    unsigned b_so = 0;                      // byte offset of (n-tile, k-tile 0, this wave) in the packed image
    auto set_a_tile = [&](int vb) __attribute__((always_inline)) {
        const long m0 = vb < nwg ? (long)(xcd_remap(vb, nwg) / a.nt) * TM : 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int row = 8 * (wave_u + 8 * q) + (lane >> 3);
            const long gm = m0 + row;
            a_vo[q] = (vb < nwg && row < 32 * MI && gm < a.m) ? ((unsigned)gm * (unsigned)a.k + (unsigned)((((lane & 7) ^ (row >> 1)) & 7) * 8)) * 2u : OOB;
        }
    };
    auto set_b_tile = [&](int vb) __attribute__((always_inline)) {
        b_so = vb < nwg ? (unsigned)(((xcd_remap(vb, nwg) % a.nt) * NK * 8 + wave_u) * 4096) : OOB;
    };
    auto issue_a = [&]() __attribute__((always_inline)) {
        float *slot = lds + a_slot * AF;
#pragma unroll
        for (int q = 0; q < 4; q++)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(arsrc, (__attribute__((address_space(3))) void *)(slot + 8 * (wave_u + 8 * q) * BKF), 16, a_vo[q],
                                                     a_kt * BKE * 2, 0, 0);
        if (++a_slot == ASLOTS) a_slot = 0;
        if (++a_kt == NK) {
            a_kt = 0;
            a_vb += gridDim.x;
            set_a_tile(a_vb);
        }
    };
    auto load_b = [&](u4v (&b)[4]) __attribute__((always_inline)) {
        // the range check sees the VGPR offset only: a dropped load carries OOB there
        const unsigned so = b_so == OOB ? 0u : b_so + (unsigned)b_kt * 8u * 4096u;
        const unsigned vo = b_so == OOB ? OOB : (unsigned)lane * 16u;
#pragma unroll
        for (int s = 0; s < 4; s++) b[s] = __builtin_bit_cast(u4v, __builtin_amdgcn_raw_buffer_load_b128(brsrc, vo + s * 1024u, so, 0));
        if (++b_kt == NK) {
            b_kt = 0;
            b_vb += gridDim.x;
            set_b_tile(b_vb);
        }
    };
    set_a_tile(a_vb);
    set_b_tile(b_vb);
    u4v b0[4], b1[4];
    load_b(b0);                                  // B(0), then A(0), A(1): "everything up to B(i)" is one counted wait from step 0 on
    issue_a();
    issue_a();

    f16v acc[MI];
#pragma unroll
    for (int mi = 0; mi < MI; mi++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[mi][r] = 0.f;

    // ---- deferred epilogue. A tile's 52 store instructions are NOT issued behind its last k-step: all 256 workgroups run in
    // phase (same tile count, same work), so that burst was 26 MB of stores at once, 5.8 us per tile during which — vmcnt
    // retires in order — no load issued behind them could be seen complete: loads and stores ran one after the other (ablation,
    // profiles/r03: loads alone 27 us, stores alone 23 us, both 47 us). Instead the tile's last step turns the accumulators into
    // 52 packed registers (BN + ReLU6 + bf16 pairs) and the NEXT tile's k-steps store them one row block per step: a steady
    // 7-8 stores per step beside the loads. The first tile's steps store a dummy (out-of-range offset, dropped), so that every
    // step issues the same operations and every counted wait stays a constant.
    unsigned outp[MI][8];
#pragma unroll
    for (int mi = 0; mi < MI; mi++)
#pragma unroll
        for (int jj = 0; jj < 8; jj++) outp[mi][jj] = 0u;
    const bool odd = li & 1;
    const unsigned ldc2 = (unsigned)a.n * 2u;                                                // bytes per output row
    const unsigned row_l = (unsigned)(4 * lh + (odd ? 16 : 0));                              // the lane's row inside (block, j): see finalize
    // per-lane byte offset of a store, the same for every tile: (lane's row, lane's channel pair inside the 256-column tile); tile,
    // block and j go through the scalar offset. Block 6 holds tile rows 192..223 of which 192..195 exist: lanes with row_l == 0 only.
    const unsigned lane_c = row_l * ldc2 + (unsigned)(wave_u * 32 + (li & ~1)) * 2u;
    unsigned o_lc = OOB, o_lc6 = OOB;            // lane offsets of the pending tile (OOB: nothing pending)
    unsigned o_m0 = 0, o_nb = 0;                 // its first row, byte offset of its first column
    bool o_ragged = false;
    auto store_block = [&](int mi) __attribute__((always_inline)) {
#pragma unroll
        for (int jj = 0; jj < 8; jj++) {
            if (mi * 32 + 8 * (jj >> 2) + (jj & 3) >= TM) continue;                          // block 6: j >= 4 lies past the tile's 196 rows
            const unsigned rt = o_m0 + (unsigned)(mi * 32 + (jj & 3) + 8 * (jj >> 2));       // wave-uniform row of (block, j)
            const unsigned lb = mi == MI - 1 ? o_lc6 : o_lc;
            if (ABL & 16) asm volatile("" ::"v"(outp[mi][jj]), "v"(lb));
            else if (!o_ragged) __builtin_amdgcn_raw_buffer_store_b32(outp[mi][jj], orsrc, lb, rt * ldc2 + o_nb, 0);
            else {                                                                           // rows past M: whole offset through the VGPR, range-checked
                const bool keep = lb != OOB && (long)rt + row_l < a.m;
                __builtin_amdgcn_raw_buffer_store_b32(outp[mi][jj], orsrc, keep ? lb + rt * ldc2 + o_nb : OOB, 0, 0);
            }
        }
    };
    // row blocks stored in step KT of a tile, and how many store instructions that is
    auto blocks_of = [](int kt, int &lo, int &hi) {
        if (NK == 4) { lo = 2 * kt; hi = kt == 3 ? 7 : 2 * kt + 2; }
        else { lo = kt < MI ? kt : 0; hi = kt < MI ? kt + 1 : 0; }
    };

    int cvb = blockIdx.x, cas = 0;
    f4 fa_fix = f4{ 0.f, 0.f, 0.f, 0.f };
    if (ABL & 8) { wide_barrier<0>(); fa_fix = *reinterpret_cast<const f4 *>(lds + fr[0]); }
    // one k-step on filter set bc, loading the next k-tile's filter into bn; KT = the step's index inside its tile
    auto step = [&](auto kt_tag, u4v (&bc)[4], u4v (&bn)[4]) __attribute__((always_inline)) {
        constexpr int KT = decltype(kt_tag)::value;
        constexpr bool LAST = KT == NK - 1;
        // younger than B(i) at this point: the 4 pieces of A(i+1) and the deferred stores of step i-1
        constexpr int PKT = (KT + NK - 1) % NK;
        constexpr int PST = NK == 4 ? (PKT == 3 ? 4 : 16) : (PKT < MI - 1 ? 8 : PKT == MI - 1 ? 4 : 0);
        if (ABL & 3) wide_barrier<0, !(ABL & 32)>();
        else wide_barrier<4 + ((ABL & 16) ? 0 : PST), !(ABL & 32)>();
        const float *As = lds + cas * AF;
        if (++cas == ASLOTS) cas = 0;
        // fragment f = (k16 step s, row block mi) = 7 s + mi, read FD fragments ahead of its MFMA through a 4-deep register
        // window (a full second fragment set, 28 more VGPRs, spilled 62-70 registers at 256)
        constexpr int FD = 3;
        f4 win[4];
        auto rd = [&](int f) __attribute__((always_inline)) {
            if (ABL & 8) return fa_fix;
            return *reinterpret_cast<const f4 *>(As + fr[f / MI] + (f % MI) * 32 * BKF);
        };
#pragma unroll
        for (int f = 0; f < FD; f++) win[f] = rd(f);
        float sc = 1.f, sh = 0.f;
        int n0 = 0;
        unsigned m0 = 0;
        if constexpr (LAST) {                            // ahead of this step's loads, so that waiting for them drains nothing younger
            const int lid = xcd_remap(cvb, nwg);
            n0 = (lid % a.nt) * BN;
            m0 = (unsigned)(lid / a.nt) * TM;
            const unsigned co = (unsigned)(n0 + wave_u * 32 + li) * 4u;
            sc = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(scrsrc, co, 0, 0));
            sh = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(shrsrc, co, 0, 0));
        }
        if (!(ABL & 2)) load_b(bn);                      // B(i+1)
        if (!(ABL & 1)) issue_a();                       // A(i+2)
        {
            int lo, hi;
            blocks_of(KT, lo, hi);
#pragma unroll
            for (int mi = 0; mi < MI; mi++)
                if (mi >= lo && mi < hi) store_block(mi);    // the previous tile's row block(s) of this step
        }
        // pinned: left to itself the scheduler sinks the filter loads to the end of the step (right in front of the barrier behind which
        // they are needed) and turns the fragment window into read-one-use-one (seen in the ISA: lgkmcnt(1) in front of every MFMA)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int f = 0; f < 4 * MI; f++) {
            if (f + FD < 4 * MI) win[(f + FD) & 3] = rd(f + FD);
            if (ABL & 4) asm volatile("" ::"v"(win[f & 3]), "v"(bc[f / MI]));
            else acc[f % MI] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, win[f & 3]), __builtin_bit_cast(bf8, bc[f / MI]), acc[f % MI], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (LAST) {
            // finalize: lane l holds channel c = n0 + 32 w + (l & 31) of rows R(r) = (r & 3) + 8 (r >> 2) + 4 lh of each block. Adjacent
            // lanes exchange half their rows (DPP quad_perm 1,0,3,2): an even lane keeps channels (c, c+1) of rows R(0..7), its odd
            // neighbour channels (c-1, c) of rows R(8..15) = R(0..7) + 16 — two adjacent channels per 4-byte store.
#pragma unroll
            for (int mi = 0; mi < MI; mi++) {
#pragma unroll
                for (int jj = 0; jj < 8; jj++) {
                    if (mi * 32 + 8 * (jj >> 2) + (jj & 3) >= TM) continue;
                    const float x = fminf(fmaxf(fmaf(acc[mi][jj], sc, sh), 0.f), 6.f);
                    const float y = fminf(fmaxf(fmaf(acc[mi][jj + 8], sc, sh), 0.f), 6.f);
                    const float tx = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));
                    const float ty = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, y), 0xB1, 0xF, 0xF, true));
                    outp[mi][jj] = __builtin_bit_cast(unsigned, odd ? bf2e{ (__bf16)ty, (__bf16)y } : bf2e{ (__bf16)x, (__bf16)tx });
                }
            }
#pragma unroll
            for (int mi = 0; mi < MI; mi++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[mi][r] = 0.f;
            o_lc = lane_c;
            o_lc6 = row_l == 0 ? lane_c : OOB;
            o_m0 = m0;
            o_nb = (unsigned)n0 * 2u;
            o_ragged = (long)m0 + TM > a.m;                                                  // the last row tile when M % 196 != 0
            cvb += gridDim.x;
        }
    };
#define WS(KT, BC, BNX) step(std::integral_constant<int, KT>{}, BC, BNX);
    for (int t = 0; t < ntile_cnt; t++) {
        if constexpr (NK == 4) { WS(0, b0, b1) WS(1, b1, b0) WS(2, b0, b1) WS(3, b1, b0) }
        else if constexpr (NK == 8) { WS(0, b0, b1) WS(1, b1, b0) WS(2, b0, b1) WS(3, b1, b0) WS(4, b0, b1) WS(5, b1, b0) WS(6, b0, b1) WS(7, b1, b0) }
        else {
            WS(0, b0, b1) WS(1, b1, b0) WS(2, b0, b1) WS(3, b1, b0) WS(4, b0, b1) WS(5, b1, b0) WS(6, b0, b1) WS(7, b1, b0)
            WS(8, b0, b1) WS(9, b1, b0) WS(10, b0, b1) WS(11, b1, b0) WS(12, b0, b1) WS(13, b1, b0) WS(14, b0, b1) WS(15, b1, b0)
        }
    }
#undef WS
    // the last tile's stores
#pragma unroll
    for (int mi = 0; mi < MI; mi++) store_block(mi);
}

}   // namespace

// Envelope of the wide kernel (the caller falls back to the other bf16 GEMMs outside it).
static bool wide_shape_ok(long m, int cin, int op_size)
{
    return (cin == 256 || cin == 512 || cin == 1024) && op_size >= 256 && (op_size % BN) == 0 && m >= 4 * TM &&
           (double)m * cin * 2 < (double)OOB && (double)m * op_size * 2 < (double)OOB && (double)op_size * cin * 2 < (double)OOB;
}

int mbn_launch_pack_filter_bf16(mbn_context *ctx, hipStream_t s, void *dst, const void *src, int n, int k)
{
    (void)ctx;
    if (n <= 0 || k <= 0 || (n % BN) != 0 || (k % BKE) != 0 || ((uintptr_t)dst % 16) != 0 || ((uintptr_t)src % 16) != 0) return MBN_EUNSUPPORTED;
    const long total = (long)n * k / 8;
    hipLaunchKernelGGL(pack_filter_bf16, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (u4v *)dst, (const __bf16 *)src, n, k);
    return MBN_OK;
}

// MBN_OK when launched; MBN_EUNSUPPORTED when the shape is outside this kernel's envelope. `fpk` = the packed image of the filter.
int mbn_launch_bf16_pw_wide(const mbn_call &c, void *out, const void *in, const void *fpk, long m, int cin, int op_size)
{
    if (c.dtype != MBN_DT_BF16 || (c.io_flags & (MBN_IO_OUT_F32 | MBN_IO_IN_F32)) || c.act != MBN_ACT_RELU6 || !c.scale || !c.shift || !fpk)
        return MBN_EUNSUPPORTED;
    if (!wide_shape_ok(m, cin, op_size)) return MBN_EUNSUPPORTED;
    if (((uintptr_t)in % 16) != 0 || ((uintptr_t)fpk % 16) != 0 || ((uintptr_t)out % 4) != 0 || ((uintptr_t)c.scale % 4) != 0 || ((uintptr_t)c.shift % 4) != 0)
        return MBN_EUNSUPPORTED;
    WideArgs a;
    a.out = (__bf16 *)out; a.in = (const __bf16 *)in; a.fpk = fpk; a.scale = c.scale; a.shift = c.shift;
    a.m = m; a.k = cin; a.n = op_size;
    a.mt = (int)((m + TM - 1) / TM);
    a.nt = op_size / BN;
    const long nwg = (long)a.mt * a.nt;
    if (nwg > 0x7fffffffL) return MBN_EUNSUPPORTED;
    long grid = c.ctx->num_cus;                                // 96 KB of LDS, ~220 VGPRs: one workgroup per CU
    if (grid > nwg) grid = nwg;
    const dim3 g((unsigned)grid), b(NT);
#ifdef MBN_LAB
    if (cin == 512) switch (g_mbn_tune.exp1) {                 // ablations (timing only, K = 512)
#define WIDE_ABL(X) case X: hipLaunchKernelGGL((pw_wide_bf16<8, X>), g, b, 0, c.stream, a); return MBN_OK;
        WIDE_ABL(4) WIDE_ABL(8) WIDE_ABL(12) WIDE_ABL(15) WIDE_ABL(16) WIDE_ABL(28) WIDE_ABL(32)
#undef WIDE_ABL
    default: break;
    }
#endif
    if (cin == 256) hipLaunchKernelGGL((pw_wide_bf16<4, 0>), g, b, 0, c.stream, a);
    else if (cin == 512) hipLaunchKernelGGL((pw_wide_bf16<8, 0>), g, b, 0, c.stream, a);
    else hipLaunchKernelGGL((pw_wide_bf16<16, 0>), g, b, 0, c.stream, a);
    return MBN_OK;
}

int mbn_bf16_pw_wide_eligible(long m, int cin, int op_size) { return wide_shape_ok(m, cin, op_size) ? 1 : 0; }
