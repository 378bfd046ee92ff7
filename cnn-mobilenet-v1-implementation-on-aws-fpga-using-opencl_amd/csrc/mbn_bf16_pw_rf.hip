// mbn_bf16_pw_rf.hip — the 1x1 pointwise conv of the bf16 mode at K = 512 (kernel.cl:94-114 `pointwise`; layers 15-25 of the sequence
// MobileNet.c:322-2599: out = relu6(scale * (in . filt^T) + shift), in [M][512] bf16, filt [N][512] bf16, out [M][N] bf16, fp32 accumulate) with the
// FILTER IN REGISTERS for the whole launch (round 6). LAB BUILD ONLY: bit for bit the M16 streaming kernel, measured equal to slower (the end of this comment).
//
// Why. On these layers the tiled kernels (pw_gemm<bf16>, pw_stream_bf16) move 804 MB through the L2 -> LDS path per launch for 206 MB of HBM
// traffic — 128 x 128 tiles pass the activations N / 128 = 4 times and the 512 KB filter once per 128 rows — and read 1.5 LDS fragments per matrix
// instruction: 55 us where the HBM bytes cost 34 (profiles/LOG.md R3.2, R6.9). Here a workgroup (8 waves) owns 256 output channels for the WHOLE
// launch: wave w keeps the 32 filter rows of its channels in 128 VGPRs (the matrix instruction's A operand: 16 k-steps x two 16-row blocks), loaded
// once. Only the activations move: a pixel row is 1024 bytes = ONE buffer_load_dwordx4 ... lds per wave, into a ring of four 32-pixel stages
// (rows padded to 1040 bytes: conflict-free 16-byte fragment reads), three stages in flight ahead of the one being multiplied. Per stage a wave reads
// 32 B fragments and issues 64 v_mfma_f32_16x16x32_bf16 (one LDS read per two instructions; nothing but pixels passes the LDS); the BN / ReLU6 /
// rounding and the two 16-byte stores per lane of the PREVIOUS stage are issued between them. One raw barrier with a counted vmcnt per stage. The N / 256 workgroups that share a pixel stream sit on the same XCD
// and walk it in step: the second reader is served by that XCD's L2.
// Arithmetic: the products of v_mfma_f32_16x16x32_bf16 over k = 0 ... 511 in 16 steps of 32, in this order, into fp32 — the instruction, k grouping
// and order of pw_stream_bf16's M16 form, which these layers ran on before (operands swapped: D^T = B^T A^T, same products and sums).
// Measured (profiles/r06/u_*; batch 512): layer 15 0.060-0.065 ms against 0.063 for pw_stream_bf16's M16 form, layer 25 0.039 against 0.035. Ablation: the
// loop without DMA and stores still takes 1.7 us per stage (3600 cycles where its 128 matrix instructions per SIMD take 2048): the epilogue's ~100 VALU
// instructions per wave and stage ADD to the matrix time instead of hiding under it, and 1024 matrix cycles per wave between two barriers leave the barrier
// skew and the fragment latency exposed with two waves per SIMD. Fewer bytes through the LDS did not buy time: these layers are not bound by that path.
// Envelope: K = 512, N a multiple of 256, BN + ReLU6 epilogue, bf16 in / out, tensors < 3.75 GiB, 16-byte aligned operands.
#include "mbn_internal.h"
#include "mbn_epilogue.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));
typedef unsigned u2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf4 __attribute__((ext_vector_type(4)));

constexpr int RF_TP = 32;                      // pixels per stage
constexpr int RF_ST = 4;                       // stages in the ring
constexpr int RF_CW = 256;                     // output channels per workgroup (8 waves x 32)
constexpr unsigned OOB = 0xF0000000u;          // a buffer offset past every tensor in the envelope: the load writes zeros into LDS

struct RfArgs {
    __bf16 *out;
    const __bf16 *in, *filt;
    const float *scale, *shift;
    long m;
    int n, mt, nh, gq;      // output channels; 32-pixel tiles; channel slices (N / 256); pixel streams per XCD
    int dbg;                // lab ablations (exp0 = 700 + bits): 1 no LDS-DMA after the prologue, 2 no fragment reads / MFMAs, 4 no stores (timing only)
};

__device__ __forceinline__ float relu6(float v) { return fminf(fmaxf(v, 0.f), 6.f); }

// all but the VM_LEFT youngest vector-memory operations of this wave are done, its LDS reads are done, then s_barrier (asm: nothing moves across it)
template <int VM_LEFT>
__device__ __forceinline__ void rf_barrier()
{
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(VM_LEFT) : "memory");
}

template <int K>
__global__ __launch_bounds__(512) void pw_rf_bf16(RfArgs a)
{
    static_assert(K == 512, "a pixel row is one 64-lane x 16-byte LDS-DMA");
    constexpr int ROWB = K * 2 + 16;               // LDS row: 1040 bytes = 260 words = 4 banks past a multiple of 64: sixteen rows' 16-byte reads cover all banks once
    constexpr int STB = RF_TP * ROWB;              // 33,280 bytes per stage
    constexpr int KG = K / 32;                     // k steps of v_mfma_f32_16x16x32_bf16
    constexpr int RPW = RF_TP / 8;                 // pixel rows a wave brings in per stage (4)
    constexpr int NSTORE = 2;                      // 16-byte stores per lane per stage
    __shared__ __attribute__((aligned(16))) char lds[RF_ST * STB];          // 133,120 bytes: one workgroup per CU

#ifdef MBN_LAB
    const int dbg = a.dbg;
#else
    constexpr int dbg = 0;
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j16 = lane & 15, q = lane >> 4;

    // this workgroup: XCD x (round-robin dispatch), channel slice `half`, pixel stream `grp` of the XCD: tiles x + 8 (grp + gq * step)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int half = slot % a.nh, grp = slot / a.nh;
    const int t0 = xcd + 8 * grp, tstep = 8 * a.gq;
    if (t0 >= a.mt) return;
    const int nt = (a.mt - 1 - t0) / tstep + 1;

    const __amdgpu_buffer_rsrc_t irsrc = mbn_make_rsrc(a.in, (unsigned)(a.m * K * 2));
    const __amdgpu_buffer_rsrc_t orsrc = mbn_make_rsrc(a.out, (unsigned)(a.m * a.n * 2));

    // activations: row r = 4 wave + p of tile ti -> stage ti % 4. Past the workgroup's last tile (and past M) the offset is out of range: zeros, and
    // every step issues exactly RPW operations, so the counted waits are constants
    auto dma = [&](int ti) __attribute__((always_inline)) {
        char *st = lds + (ti & (RF_ST - 1)) * STB;
        const long p0 = (long)(t0 + ti * tstep) * RF_TP + wave_u * RPW;
#pragma unroll
        for (int p = 0; p < RPW; p++) {
            const unsigned vo = (ti < nt && p0 + p < a.m) ? (unsigned)(p0 + p) * (unsigned)(K * 2) + (unsigned)lane * 16u : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(irsrc, (__attribute__((address_space(3))) void *)(st + (wave_u * RPW + p) * ROWB), 16, vo, 0, 0, 0);
        }
    };

    // the ring's first three stages are requested before anything else: the filter rows below arrive under them
    dma(0);
    dma(1);
    dma(2);

    // ---- the wave's filter rows: channels c0 + 16 blk + j16, k = 32 g + 8 q .. + 7 (the A operand of step g, block blk)
    const int c0 = half * RF_CW + 32 * wave_u;
    u4v wfr[KG][2];
    {
        const __bf16 *wrow = a.filt + (size_t)(c0 + j16) * K + 8 * q;
#pragma unroll
        for (int g = 0; g < KG; g++)
#pragma unroll
            for (int b = 0; b < 2; b++) wfr[g][b] = *reinterpret_cast<const u4v *>(wrow + (size_t)(16 * b) * K + 32 * g);
    }
    // C/D: register r of block (blk, pb) = channel c0 + 16 blk + 4 q + r of pixel 16 pb + j16
    f4 sc[2], sh[2];
#pragma unroll
    for (int b = 0; b < 2; b++) {
        sc[b] = *reinterpret_cast<const f4 *>(a.scale + c0 + 16 * b + 4 * q);
        sh[b] = *reinterpret_cast<const f4 *>(a.shift + c0 + 16 * b + 4 * q);
    }

    const unsigned bfrag = (unsigned)(j16 * ROWB + q * 16);              // + pb * 16 * ROWB + g * 64
    // stores: v_permlane16_swap pairs the lanes q, q + 1 of a pixel: the even one ends up with channels 4 q ... 4 q + 7 of block 0, the odd one with
    // 4 (q - 1) ... + 7 of block 1 — 16 bytes per lane, 64 contiguous bytes per pixel and instruction instead of 32
    const unsigned obase = (unsigned)(c0 + ((q & 1) ? 16 + 4 * (q - 1) : 4 * q)) * 2u;

    // BN + ReLU6 + rounding of one 16-pixel block of the PREVIOUS tile and its store: issued between the matrix instructions of the current one
    auto epilogue = [&](const f4 (&pacc)[2][2], int pb, long pix0) __attribute__((always_inline)) {
        u2v d[2];
#pragma unroll
        for (int b = 0; b < 2; b++) {
            const f4 v = pacc[b][pb];
            d[b] = __builtin_bit_cast(u2v, bf4{ (__bf16)relu6(fmaf(v.x, sc[b].x, sh[b].x)), (__bf16)relu6(fmaf(v.y, sc[b].y, sh[b].y)),
                                                 (__bf16)relu6(fmaf(v.z, sc[b].z, sh[b].z)), (__bf16)relu6(fmaf(v.w, sc[b].w, sh[b].w)) });
        }
        const auto lo = __builtin_amdgcn_permlane16_swap(d[0].x, d[1].x, false, false);
        const auto hi = __builtin_amdgcn_permlane16_swap(d[0].y, d[1].y, false, false);
        const unsigned po = (unsigned)(pix0 + 16 * pb) * (unsigned)(a.n * 2) + obase;          // past M: beyond the buffer's range, the store is dropped
        if (!(dbg & 4)) __builtin_amdgcn_raw_buffer_store_b128(u4v{ lo[0], hi[0], lo[1], hi[1] }, orsrc, po, 0, 0);
    };

    f4 pacc[2][2];                                                       // the previous tile's sums, waiting for their epilogue
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
        for (int pb = 0; pb < 2; pb++) pacc[b][pb] = f4{ 0.f, 0.f, 0.f, 0.f };
    long ppix = 0;

    for (int ti = 0; ti < nt; ti++) {
        // tile ti has landed for every wave, and every wave is done reading tile ti - 1 (whose stage the DMA below refills).
        // Issued behind DMA(ti), steady state: stores(ti-4), DMA(ti+1), stores(ti-3), DMA(ti+2), stores(ti-2) = 3 x NSTORE + 2 x RPW; fewer in the first four
        // steps (the first waits for everything: the filter rows were requested behind the first three stages)
        if (ti == 0) rf_barrier<0>();
        else if (ti < 4) rf_barrier<2 * RPW>();
        else rf_barrier<3 * NSTORE + 2 * RPW>();
        if (!(dbg & 1)) dma(ti + 3);

        const char *st = lds + (ti & (RF_ST - 1)) * STB;
        f4 acc[2][2];
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int pb = 0; pb < 2; pb++) acc[b][pb] = f4{ 0.f, 0.f, 0.f, 0.f };
#pragma unroll
        for (int g = 0; g < KG; g++) {
            if (!(dbg & 2)) {
                u4v yf[2];
#pragma unroll
                for (int pb = 0; pb < 2; pb++) yf[pb] = *reinterpret_cast<const u4v *>(st + bfrag + (unsigned)(pb * 16 * ROWB + g * 64));
#pragma unroll
                for (int b = 0; b < 2; b++)
#pragma unroll
                    for (int pb = 0; pb < 2; pb++)
                        acc[b][pb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, wfr[g][b]), __builtin_bit_cast(bf8, yf[pb]), acc[b][pb], 0, 0, 0);
            }
            // the previous tile's epilogue in the shadow of this tile's matrix instructions (one wave per SIMD slot pair runs them: VALU and stores issue while they execute)
            if (ti > 0 && g == KG / 4) epilogue(pacc, 0, ppix);
            if (ti > 0 && g == (3 * KG) / 4) epilogue(pacc, 1, ppix);
        }
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int pb = 0; pb < 2; pb++) pacc[b][pb] = acc[b][pb];
        ppix = (long)(t0 + ti * tstep) * RF_TP + j16;
    }
    epilogue(pacc, 0, ppix);
    epilogue(pacc, 1, ppix);
    // the ring still holds DMA in flight (zeros past the last tile): drain before the LDS is handed back
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}   // namespace

// 1 when the layer is inside the kernel's envelope
static int mbn_bf16_pw_rf_eligible(const mbn_call &c, const void *out, const void *in, const void *filt, long m, int cin, int op_size)
{
    if (c.dtype != MBN_DT_BF16 || (c.io_flags & (MBN_IO_OUT_F32 | MBN_IO_IN_F32)) || c.act != MBN_ACT_RELU6 || !c.scale || !c.shift) return 0;
    if (cin != 512 || op_size < RF_CW || (op_size % RF_CW) != 0 || m < 1) return 0;
    if (((uintptr_t)in % 16) || ((uintptr_t)filt % 16) || ((uintptr_t)out % 8) || ((uintptr_t)c.scale % 16) || ((uintptr_t)c.shift % 16)) return 0;
    // 32-bit byte offsets: the input below the out-of-range marker, the output with a tile of head room below 4 GiB (a ragged last tile's rows past M must not wrap)
    if ((double)m * cin * 2 >= (double)OOB || ((double)m + 64.0) * op_size * 2 >= 4294967296.0) return 0;
    const int nh = op_size / RF_CW;
    if (c.ctx->num_cus < 8 * nh) return 0;
    return 1;
}

int mbn_launch_bf16_pw_rf(const mbn_call &c, void *out, const void *in, const void *filt, long m, int cin, int op_size)
{
    if (!mbn_bf16_pw_rf_eligible(c, out, in, filt, m, cin, op_size)) return MBN_EUNSUPPORTED;
    RfArgs a;
    a.out = (__bf16 *)out; a.in = (const __bf16 *)in; a.filt = (const __bf16 *)filt;
    a.scale = c.scale; a.shift = c.shift;
    a.m = m; a.n = op_size;
    a.mt = (int)((m + RF_TP - 1) / RF_TP);
    a.nh = op_size / RF_CW;
    a.gq = c.ctx->num_cus / 8 / a.nh;                      // pixel streams per XCD: one workgroup per CU
    // fewer tiles than streams: the grid shrinks to the streams that have a tile (tile t belongs to XCD t % 8, stream (t / 8) % gq)
    const int streams_used = a.mt >= 8 * a.gq ? a.gq : (a.mt + 7) / 8;
    a.gq = streams_used;
    a.dbg = 0;
#ifdef MBN_LAB
    a.dbg = g_mbn_tune.exp0 >= 700 && g_mbn_tune.exp0 < 732 ? g_mbn_tune.exp0 - 700 : 0;
#endif
    hipLaunchKernelGGL(pw_rf_bf16<512>, dim3((unsigned)(8 * a.nh * a.gq)), dim3(512), 0, c.stream, a);
    return MBN_OK;
}
