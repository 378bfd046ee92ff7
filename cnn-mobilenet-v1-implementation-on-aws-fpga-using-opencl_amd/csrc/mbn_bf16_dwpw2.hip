// mbn_bf16_dwpw2.hip — fused depthwise 3x3 -> pointwise 1x1 block in the network's bf16 mode (BASELINE config 5),
// UNIFIED-WAVE form (round 2): the bf16 counterpart of mbn_f32_dwpw2.hip and the replacement of the producer/consumer
// kernel of mbn_bf16_dwpw.hip. Activations and the pointwise filter are bf16, all arithmetic is fp32, every layer output
// is rounded to bf16 (RNE) — including the depthwise output, which only ever exists in LDS. Replaces a `depthwise` +
// `pointwise` launch pair of the reference's sequence (kernel.cl:62-92 + 94-114; pairs L4-5 ... L26-27, MobileNet.c:322-2599).
//
// In bf16 the matrix work is a sixteenth of the fp32 kernel's (v_mfma_f32_32x32x16_bf16), so a block is bound by HBM and
// by the depthwise VALU work (widening 12-15 packed vectors, 72 packed FMAs, BN + ReLU6, rounding: ~230 instructions per
// lane per 64-channel chunk). What the round-1 kernel lost is latency: its producers drained vmcnt(0) in front of every
// barrier, i.e. the x-window prefetch for the next chunk never overlapped anything (0.15 ms for block 14-15 against
// 0.12 ms for the two separate launches and a 0.034 ms HBM floor). Here, as in the fp32 unified kernel: 8 waves, every
// wave produces its 1/8 of the chunk (2 pixels x 8 channels per lane) AND multiplies its 64x64 / 32x64 tile; x-window
// loads are issued one chunk ahead and stay in flight across the barrier (counted s_waitcnt vmcnt(NX): only the filter
// chunk's LDS-DMA has to have landed); one barrier per chunk. A K chunk is 64 channels, so LDS rows are 128 bytes and the
// LDS image, swizzle and fragment reads are byte-for-byte those of pw_gemm<__bf16>.
// Result vs the two separate bf16 launches: the same bf16 x bf16 products accumulated in fp32 in a different order, i.e.
// within the bf16 tolerance of the parity tests, not bit-identical.
#include "mbn_internal.h"
#include "mbn_epilogue.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f8 __attribute__((ext_vector_type(8)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef mbn_f16v f16v;

constexpr int BKF = 32;                        // LDS rows are 128 bytes = 32 words = 64 bf16
constexpr int BM8 = 128;                       // rows of the 8-wave tile (2 waves per SIMD, 256 VGPRs each); the 4-wave form (two workgroups per CU): 64
constexpr int BM_H32 = 256;                    // rows of the Cin = 32 tile (H32: four lanes per pixel pair): shared by the kernel and its launcher
constexpr int CMAX4 = 256, NOUT4 = 256;        // LDS-resident constants of the 4-wave form (61 KB per workgroup)
constexpr int NOUT_G = 1024;                   // widest pointwise output whose scale/shift the LDS copy holds
constexpr int CMAX_G = 1024;                   // largest Cin (depthwise constants resident in LDS: 44 KB)
constexpr unsigned OOB = 0xF0000000u;          // byte offset beyond any supported tensor: the load returns zeros

struct DwPw2Args {
    __bf16 *out;
    const __bf16 *in, *wp;
    const float *wd, *s2, *b2, *s3, *b3;
    long m;                 // output pixels = batch * ho * wo
    int h, w, ho, wo;       // input / output map sides
    int cin, cout;
    int pad_top, pad_left;
    int mt, nt;
    unsigned in_bytes, wp_bytes;
    int dbg;                // experiments (tune misc): 1 = no x loads after the first, 2 = no depthwise math, 4 = no output stores, 8 = no filter DMA, 16 = no MFMA
    unsigned wo_m, wo_s, ho_m, ho_s;   // floor(v / wo) = umulhi(v, wo_m) >> wo_s for v < 2^31 (m == 0: the divisor is 1)
    int use4;               // launcher: the 4-wave / two-workgroups-per-CU form is allowed (tune exp2 == 44 in the lab build until measured)
    int fast_off;           // launcher: 1 = the FO instantiation (tile offsets in their full-rate form, as in mbn_f32_dwpw2.hip: input < 0x70000000 bytes)
    float inv_wo, inv_ho;   // 1 / wo, 1 / ho
};

__device__ __forceinline__ int swz(int row, int chunk) { return (row << 5) + (((chunk ^ (row >> 1)) & 7) << 2); }
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
__device__ __forceinline__ int xcd_remap(int vb, int nwg)
{
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = vb & 7;
    return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (vb >> 3);
}

// Workgroup barrier with the waits spelled out. __syncthreads() is a workgroup-scope fence over every address space: with
// global loads in flight for the NEXT chunk the waitcnt pass drains them (s_waitcnt vmcnt(0)) in front of every barrier
// (also with the "local"-only fence form), which serialises the prefetch with the hand-over. Here: wait until all but
// the VM_LEFT youngest vector-memory operations are done (= the LDS-DMA of the filter chunk has landed, the x-window
// loads issued after it may still fly), until this wave's own LDS writes are done (lgkmcnt(0)), then s_barrier. The asm
// is volatile with a memory clobber, so the compiler moves no LDS or global access across it.
template <int VM_LEFT>
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(VM_LEFT) : "memory");
}

// pointwise filter chunk -> LDS, buffer form (a __device__ function: see mbn_f32_pw.hip lds_dma_rows)
template <int B_LD, int NT>
__device__ __forceinline__ void dma_filter(__amdgpu_buffer_rsrc_t rsrc, float *lds_b, const unsigned *voff, int soff, int wave_u)
{
#pragma unroll
    for (int p = 0; p < B_LD; p++)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)(lds_b + (p * (NT / 8) + wave_u * 8) * BKF),
                                                 16, voff[p], soff, 0, 0);
}

// DBG = false: the shipped kernel, the experiment switches (a.dbg) fold away; DBG = true only for tune dwpw_variant >= 100
// M16 = true (LAB, round 4, tune misc = 32): the pointwise products on v_mfma_f32_16x16x32_bf16 — the same LDS image and the same bytes read per chunk
// (a lane's 16-byte fragment is k = 32 kg + 8 q .. + 7 of row r: chunk 4 kg + q of the 128-byte row), two k-groups of 32 per 64-channel chunk; the
// shape the chip holds a higher clock under in MFMA-dense loops (profiles/r03/x_bf16_mfma_shape.txt). Measured here (profiles/r04/c_bf16_mfma_shape_blocks.txt,
// bf16 1.0x224 batch 512, alternating runs): blocks 4-11 0.746-0.754 ms against 0.739-0.749 with 32x32x16 — equal within the run-to-run spread, 0.8 % worse
// on the means: a block's matrix work is a sixteenth of the fp32 kernel's, too thin to pull the clock down, and the 16 x 16 form holds 24-32 more
// fragment registers. Parity-tested (oracle + exact integers, tests/test_parity_gpu.py::test_bf16_dwpw_fused under MBN_LAB=1), not shipped.
// FO (round 5): set_offsets from full-rate instructions — see mbn_f32_dwpw2.hip (set_offsets_fast). In bf16 a block with Cin <= 64 is ONE chunk per tile, so the
// offsets are computed in every step: the general form's two v_mul_hi_u32, six v_mul_lo_u32 and 12-15 compare/select pairs under exec-mask branches were ~20 % of it.
// NW (round 5): 8 = one workgroup of 8 waves per CU on 128-row tiles (rounds 2-4); 4 = 4 waves on 64-row tiles, two independent workgroups per CU (61 KB of LDS each:
// Cin, Cout <= 256): they fall out of phase by themselves, so one's waits, barrier and epilogue run under the other's depthwise arithmetic. Pays only below the power limit
// (the fp32 counterpart measured equal at the limit: profiles/r05/j_*) — and, measured, not there either: lab only (launch2).
// H32 (round 5): Cin = 32 with every depthwise lane busy. The padded form above leaves lanes c4 >= 4 idle (half the VALU lanes of a VALU-bound kernel). Here FOUR lanes
// cover a pixel pair's 32 channels, so the 512 lanes cover 128 pairs = a 256-row tile; the A tile keeps its 128-byte rows with 16-byte slots 0..3 written (k = 0..31)
// and the MFMAs run over k-groups 0 and 1 only; 8 waves as 4 x 2 of 64 x 64 (the 256-column variant's wave tile). Block 4-5 of the 0.5x network (32 -> 64, stride 2).
template <int S, int BN, bool DBG, bool M16, bool FO = false, int NW = 8, bool H32 = false>
__global__ __launch_bounds__(64 * NW) void dwpw2_bf16(DwPw2Args a)
{
    static_assert(!H32 || (BN == 128 && NW == 8 && !M16 && !DBG), "H32: the shipped 8-wave 128-column form");
    constexpr int NT = 64 * NW, BM = (H32 ? 32 : 16) * NW;
    static_assert(!H32 || BM == BM_H32, "the launcher sizes the H32 grid by BM_H32");
    constexpr int CMAX = NW == 4 ? CMAX4 : CMAX_G, NOUT = NW == 4 ? NOUT4 : NOUT_G;
    const int dbg = DBG ? a.dbg : 0;
    constexpr int WN = 64, WM = (BN == 256 || H32) ? 64 : 32;   // wave tile: 8 waves as 2 x 4 (BN 256), 4 x 2 (BN 128), 4 x 2 of 64 x 64 (H32: 256 rows)
    constexpr int WAVES_N = BN / WN;
    static_assert((BM / WM) * WAVES_N == NW, "wave grid");
    constexpr int MI = WM / 32, NI = WN / 32;
    constexpr int B_LD = BN * 8 / NT;                  // 16-B filter pieces per lane per chunk (2 / 4)
    constexpr int XC = S + 3;                          // input columns feeding 2 adjacent output pixels
    constexpr int NX = 3 * XC;                         // buffer loads per lane per chunk
    constexpr int ABUF = BM * BKF, BBUF = BN * BKF;
    __shared__ __attribute__((aligned(16))) float lds[2 * ABUF + 2 * BBUF + 11 * CMAX + 2 * NOUT];
    float *const a_s0 = lds, *const b_s0 = lds + 2 * ABUF, *const wd_s = b_s0 + 2 * BBUF, *const sb_s = wd_s + 9 * CMAX;
    // pointwise scale | shift of all Cout channels: the epilogue reads them with ds_read. As global loads they were the wave's
    // youngest vector-memory operations, and waiting for them (in-order vmcnt) drained the x-window loads and the filter DMA
    // already in flight for the next steps: 1700-3100 cycles per tile in the stamps (profiles/r02/g_dwpw2_stamps.txt)
    float *const sc3_s = sb_s + 2 * CMAX, *const sh3_s = sc3_s + NOUT;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Cin = 32 (round 5: block 4-5 of the 0.5x network) is half a 64-channel chunk: one chunk whose upper 32 channels are padding — lanes c4 >= 4 load
    // nothing (out-of-range offsets), write zeros into the A tile, and the filter pieces of k >= 32 come from beyond the descriptor (zeros)
    const int nk = (a.cin + 63) / 64, nwg = a.mt * a.nt;
    const unsigned mtot = (unsigned)a.m;

    for (int i = tid * 4; i < 9 * a.cin; i += NT * 4) *reinterpret_cast<f4 *>(wd_s + i) = *reinterpret_cast<const f4 *>(a.wd + i);
    for (int i = tid * 4; i < a.cin; i += NT * 4) {
        *reinterpret_cast<f4 *>(sb_s + i) = *reinterpret_cast<const f4 *>(a.s2 + i);
        *reinterpret_cast<f4 *>(sb_s + a.cin + i) = *reinterpret_cast<const f4 *>(a.b2 + i);
    }
    for (int i = tid; i < a.cout; i += NT) { sc3_s[i] = a.s3[i]; sh3_s[i] = a.b3[i]; }
    __syncthreads();
    if ((int)blockIdx.x >= nwg) return;

    // ---- roles of this lane
    const int c4 = H32 ? (tid & 3) : (tid & 7), pair = H32 ? (tid >> 2) : (tid >> 3);   // depthwise: tile rows 2*pair, 2*pair+1, channels 8*c4..+7 of the chunk
    const int c8 = tid & 7;                                         // 16-byte slot of the filter row this lane stages
    const int wm = (wave_u / WAVES_N) * WM, wn = (wave_u % WAVES_N) * WN;      // MFMA tile origin inside the workgroup tile
    const int li = lane & 31, lh = lane >> 5;
    const __amdgpu_buffer_rsrc_t irsrc = mbn_make_rsrc(a.in, a.in_bytes);
    const __amdgpu_buffer_rsrc_t wrsrc = mbn_make_rsrc(a.wp, a.wp_bytes);
    const __amdgpu_buffer_rsrc_t orsrc = mbn_make_rsrc(a.out, (unsigned)(a.m * a.cout * 2));
    const int aw0 = swz(2 * pair, c4), aw1 = swz(2 * pair + 1, c4);           // A-tile slots this lane writes
    int fr_a[4], fr_b[4];                                                       // fragment slots this lane reads
#pragma unroll
    for (int g = 0; g < 4; g++) {
        fr_a[g] = swz(wm + li, 2 * g + lh);
        fr_b[g] = swz(wn + li, 2 * g + lh);
    }
    unsigned b_vo[B_LD];                                                        // filter piece offsets: fixed for the kernel, the tile's
#pragma unroll                                                                  // column origin and the chunk go into the scalar offset
    for (int p = 0; p < B_LD; p++) {
        const int row = (p * NT + tid) >> 3;
        // channel-paired column blocks (mbn_epilogue.h): LDS filter row `row` holds output channel mbn_pair_channel(row)
        const unsigned kch = (unsigned)(((c8 ^ (row >> 1)) & 7) * 8);                   // first k of this 16-byte piece
        b_vo[p] = kch < (unsigned)a.cin ? ((unsigned)mbn_pair_channel(row) * (unsigned)a.cin + kch) * 2u : OOB;
    }
    const bool cok = H32 || c4 * 8 < a.cin;                                            // this lane's 8 channels exist (false only for Cin = 32, c4 >= 4)
    const float *wk = wd_s + c4 * 8;                                           // depthwise taps of this lane's 8 channels (+ kc*64 + tap*cin)
    const float *sk = sb_s + c4 * 8;

    unsigned off[3][XC];
    auto set_offsets_general = [&](unsigned m0) __attribute__((always_inline)) {
        const unsigned m = m0 + 2 * pair;
        const bool mok = m < mtot && cok;
        const unsigned q = a.wo_m ? __umulhi(m, a.wo_m) >> a.wo_s : m;
        const unsigned x = m - q * (unsigned)a.wo;
        const unsigned n = a.ho_m ? __umulhi(q, a.ho_m) >> a.ho_s : q;
        const unsigned y = q - n * (unsigned)a.ho;
        const int iy0 = (int)y * S - a.pad_top, ix0 = (int)x * S - a.pad_left;
        const unsigned cs = (unsigned)a.cin * 2u, rs = (unsigned)a.w * cs;                    // column / row stride in bytes
        const unsigned base = ((n * a.h + iy0) * a.w + ix0) * cs + (unsigned)(c4 * 8) * 2u;         // wraps for taps that are masked out below
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
    auto set_offsets_fast = [&](unsigned m0) __attribute__((always_inline)) {
        const unsigned q0 = a.wo_m ? __umulhi(m0, a.wo_m) >> a.wo_s : m0;                    // wave-uniform: scalar unit
        const unsigned x0 = m0 - q0 * (unsigned)a.wo;
        const unsigned n0 = a.ho_m ? __umulhi(q0, a.ho_m) >> a.ho_s : q0;
        const unsigned y0 = q0 - n0 * (unsigned)a.ho;
        const unsigned r = x0 + 2u * (unsigned)pair;
        const unsigned q1 = (unsigned)__builtin_fmaf((float)r, a.inv_wo, 0.5f * a.inv_wo);     // exact: r < wo + 128, see the fp32 kernel
        const unsigned x = r - q1 * (unsigned)a.wo;
        const unsigned yy = y0 + q1;
        const unsigned q2 = (unsigned)__builtin_fmaf((float)yy, a.inv_ho, 0.5f * a.inv_ho);
        const unsigned y = yy - q2 * (unsigned)a.ho;
        const unsigned n = n0 + q2;
        const bool mok = m0 + 2u * (unsigned)pair < mtot && cok;
        const int iy0 = (int)y * S - a.pad_top, ix0 = (int)x * S - a.pad_left;
        const unsigned cs = (unsigned)a.cin * 2u, rs = (unsigned)a.w * cs;
        const int pix = __mul24((int)(n * (unsigned)a.h) + iy0, a.w) + ix0;
        const unsigned base = (unsigned)pix * cs + (unsigned)(c4 * 8) * 2u;
        unsigned rowv[3], colv[XC];
#pragma unroll
        for (int dy = 0; dy < 3; dy++) rowv[dy] = (mok && (unsigned)(iy0 + dy) < (unsigned)a.h) ? base + dy * rs : 0x80000000u;
#pragma unroll
        for (int j = 0; j < XC; j++) colv[j] = ((unsigned)(ix0 + j) < (unsigned)a.w) ? j * cs : 0x70000000u;
#pragma unroll
        for (int dy = 0; dy < 3; dy++)
#pragma unroll
            for (int j = 0; j < XC; j++) off[dy][j] = rowv[dy] + colv[j];
    };
    auto set_offsets = [&](unsigned m0) __attribute__((always_inline)) {
        if constexpr (FO) set_offsets_fast(m0);
        else set_offsets_general(m0);
    };
    u4v xr[3][XC];                                                            // the window stays packed (4 VGPRs per vector)
    auto ldx = [&](int kc) __attribute__((always_inline)) {
#pragma unroll
        for (int dy = 0; dy < 3; dy++)
#pragma unroll
            for (int j = 0; j < XC; j++)
                xr[dy][j] = __builtin_amdgcn_raw_buffer_load_b128(irsrc, off[dy][j], kc * 128, 0);
    };
    // depthwise + BN + ReLU6 of the chunk in xr into A buffer `buf`, rounded to bf16 (the layer output); the window is
    // widened one row at a time, each vector once; taps and scale/shift come from LDS (fp32)
    auto dw = [&](int kc, const int buf) __attribute__((always_inline)) {
        f8 acc0, acc1;
#pragma unroll
        for (int i = 0; i < 8; i++) acc0[i] = acc1[i] = 0.f;
#pragma unroll
        for (int dy = 0; dy < 3; dy++) {
            f8 row[XC];
#pragma unroll
            for (int j = 0; j < XC; j++) row[j] = widen8(xr[dy][j]);
#pragma unroll
            for (int dx = 0; dx < 3; dx++) {
                const f8 w = ld8(wk + kc * 64 + (dy * 3 + dx) * a.cin);
                acc0 = __builtin_elementwise_fma(row[dx], w, acc0);
                acc1 = __builtin_elementwise_fma(row[dx + S], w, acc1);
            }
        }
        const f8 sc = ld8(sk + kc * 64), sh = ld8(sk + a.cin + kc * 64);
        bf8 o0, o1;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            o0[i] = (__bf16)fminf(fmaxf(fmaf(acc0[i], sc[i], sh[i]), 0.f), 6.f);
            o1[i] = (__bf16)fminf(fmaxf(fmaf(acc1[i], sc[i], sh[i]), 0.f), 6.f);
        }
        if (!cok) {                               // padded channels (Cin = 32): exact zeros, whatever the LDS words behind the taps held
            const u4v z = { 0u, 0u, 0u, 0u };
            o0 = __builtin_bit_cast(bf8, z);
            o1 = o0;
        }
        *reinterpret_cast<bf8 *>(a_s0 + buf * ABUF + aw0) = o0;
        *reinterpret_cast<bf8 *>(a_s0 + buf * ABUF + aw1) = o1;
    };

    constexpr int MI16 = WM / 16, NI16 = WN / 16;
    // 64 x 64 wave tiles in the 16x16x32 form hold 32 fragment registers per k-group: read behind the depthwise part, not ahead of it (no spills)
    constexpr bool LATEFRAG = M16 && BN == 256;
    [[maybe_unused]] f16v acc[M16 ? 1 : MI][M16 ? 1 : NI];
    [[maybe_unused]] f4 acc16[M16 ? MI16 : 1][M16 ? NI16 : 1];
    // fragment slots: 32x32x16 — four k-groups of 16, group g in slot g & 1; 16x16x32 — two k-groups of 32, group kg in slot kg
    f4 fa[2][M16 ? MI16 : MI], fb[2][M16 ? NI16 : NI];
    [[maybe_unused]] int fr16_a[2], fr16_b[2];
#pragma unroll
    for (int kg = 0; kg < 2; kg++) {
        fr16_a[kg] = swz(wm + (lane & 15), 4 * kg + (lane >> 4));
        fr16_b[kg] = swz(wn + (lane & 15), 4 * kg + (lane >> 4));
    }
    auto ldfrag = [&](const int buf, int g, int slot) __attribute__((always_inline)) {
        if constexpr (M16) {
#pragma unroll
            for (int i = 0; i < MI16; i++) fa[slot][i] = *reinterpret_cast<const f4 *>(a_s0 + buf * ABUF + fr16_a[g] + i * 16 * BKF);
#pragma unroll
            for (int j = 0; j < NI16; j++) fb[slot][j] = *reinterpret_cast<const f4 *>(b_s0 + buf * BBUF + fr16_b[g] + j * 16 * BKF);
        } else {
#pragma unroll
        for (int mi = 0; mi < MI; mi++) fa[slot][mi] = *reinterpret_cast<const f4 *>(a_s0 + buf * ABUF + fr_a[g] + mi * 32 * BKF);
#pragma unroll
        for (int ni = 0; ni < NI; ni++) fb[slot][ni] = *reinterpret_cast<const f4 *>(b_s0 + buf * BBUF + fr_b[g] + ni * 32 * BKF);
        }
    };
    auto mfma_group = [&](int slot) __attribute__((always_inline)) {
        if constexpr (M16) {
#pragma unroll
            for (int i = 0; i < MI16; i++)
#pragma unroll
                for (int j = 0; j < NI16; j++)
                    acc16[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, fa[slot][i]), __builtin_bit_cast(bf8, fb[slot][j]),
                                                                          acc16[i][j], 0, 0, 0);
        } else {
#pragma unroll
        for (int mi = 0; mi < MI; mi++)
#pragma unroll
            for (int ni = 0; ni < NI; ni++)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, fa[slot][mi]), __builtin_bit_cast(bf8, fb[slot][ni]),
                                                                      acc[mi][ni], 0, 0, 0);
        }
    };
    // first MFMA group of a step; fresh (wave-uniform: the chunk is its tile's first): C = the inline-constant zero instead of zeroing the accumulators per tile
    auto mfma_group_first = [&](int slot, bool fresh) __attribute__((always_inline)) {
        if constexpr (M16) {
            const f4 z4 = f4{ 0.f, 0.f, 0.f, 0.f };
#pragma unroll
            for (int i = 0; i < MI16; i++)
#pragma unroll
                for (int j = 0; j < NI16; j++) {
                    if (fresh) acc16[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, fa[slot][i]), __builtin_bit_cast(bf8, fb[slot][j]), z4, 0, 0, 0);
                    else acc16[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, fa[slot][i]), __builtin_bit_cast(bf8, fb[slot][j]), acc16[i][j], 0, 0, 0);
                }
        } else {
            const f16v z = f16v{ 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f };
            if (fresh) {
#pragma unroll
                for (int mi = 0; mi < MI; mi++)
#pragma unroll
                    for (int ni = 0; ni < NI; ni++)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, fa[slot][mi]), __builtin_bit_cast(bf8, fb[slot][ni]), z, 0, 0, 0);
            } else {
#pragma unroll
                for (int mi = 0; mi < MI; mi++)
#pragma unroll
                    for (int ni = 0; ni < NI; ni++)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, fa[slot][mi]), __builtin_bit_cast(bf8, fb[slot][ni]), acc[mi][ni], 0, 0, 0);
            }
        }
    };
    auto zero_acc = [&]() __attribute__((always_inline)) {
        if constexpr (M16) {
#pragma unroll
            for (int i = 0; i < MI16; i++)
#pragma unroll
                for (int j = 0; j < NI16; j++) acc16[i][j] = f4{ 0.f, 0.f, 0.f, 0.f };
        } else {
#pragma unroll
        for (int mi = 0; mi < MI; mi++)
#pragma unroll
            for (int ni = 0; ni < NI; ni++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[mi][ni][r] = 0.f;
        }
    };

    // ---- three cursors over the flattened (tile, chunk) sequence of this workgroup: L (x loads) one chunk ahead of
    //      D (depthwise + filter DMA) one chunk ahead of M (MFMA). A cursor = (virtual block id, chunk, tile origin).
    int vbM, kM, n0M; unsigned m0M;
    int vbD, kD, n0D; unsigned m0D;
    bool validD;
    auto origin = [&](int vb, unsigned &m0, int &n0) __attribute__((always_inline)) {
        const int lid = xcd_remap(vb, nwg);
        n0 = (lid % a.nt) * BN;
        m0 = (unsigned)(lid / a.nt) * BM;
    };

    // prologue: L(0), D(0) into buffer 0, L(1)
    vbM = blockIdx.x; kM = 0;
    origin(vbM, m0M, n0M);
    set_offsets(m0M);
    ldx(0);
    dma_filter<B_LD, NT>(wrsrc, b_s0, b_vo, (n0M * a.cin + 0) * 2, wave_u);
    dw(0, 0);
    // D cursor = successor of M
    vbD = vbM; kD = 1; m0D = m0M; n0D = n0M; validD = true;
    if (kD >= nk) {
        kD = 0; vbD += gridDim.x; validD = vbD < nwg;
        if (validD) { origin(vbD, m0D, n0D); set_offsets(m0D); }
    }
    if (validD) {
        ldx(kD);
    }
    zero_acc();
    if (validD) lds_barrier<NX>();        // filter chunk 0 landed; the NX newer loads may fly
    else lds_barrier<0>();

    // One chunk step with the MFMA chunk in buffer P (a literal at both call sites).
    // Returns false when the sequence is finished.
#define MBN_DWPW2_STEP(P)                                                                                               \
    {                                                                                                                   \
        if (!LATEFRAG) ldfrag(P, 0, 0);                                                                                 \
        bool validL = false;                                                                                            \
        int vbL = vbD, kL = kD + 1, n0L = n0D;                                                                          \
        unsigned m0L = m0D;                                                                                             \
        if (validD) {                                                                                                   \
            if (!(dbg & 8)) dma_filter<B_LD, NT>(wrsrc, b_s0 + (P ^ 1) * BBUF, b_vo, (n0D * a.cin + kD * 64) * 2, wave_u); \
            if (!(dbg & 2)) dw(kD, P ^ 1);                                                                            \
            validL = true;                                                                                              \
            if (kL >= nk) {                                                                                             \
                kL = 0; vbL += gridDim.x; validL = vbL < nwg;                                                           \
                if (validL) { origin(vbL, m0L, n0L); set_offsets(m0L); }                                                \
            }                                                                                                           \
            if (validL && !(dbg & 1)) ldx(kL);                                                                        \
        }                                                                                                               \
        /* a wave whose 64-column group lies past Cout (Cout = 64 mod 128) has nothing to multiply: no fragment reads, no MFMAs (wave-uniform) */ \
        if (!(dbg & 16) && n0M + wn < a.cout) {                                                                       \
        if constexpr (M16) {                                                                                            \
            if (LATEFRAG) ldfrag(P, 0, 0);                                                                              \
            ldfrag(P, 1, 1);                                                                                            \
            __builtin_amdgcn_sched_barrier(0);                                                                          \
            mfma_group_first(0, kM == 0);                                                                               \
            __builtin_amdgcn_sched_barrier(0);                                                                          \
            mfma_group(1);                                                                                              \
        } else {                                                                                                        \
        _Pragma("unroll") for (int g = 0; g < (H32 ? 1 : 3); g++) {      /* H32: K = 32 = k-groups 0 and 1 */           \
            ldfrag(P, g + 1, (g + 1) & 1);                                                                              \
            __builtin_amdgcn_sched_barrier(0);                                                                          \
            if (g == 0) mfma_group_first(0, kM == 0); else mfma_group(g & 1);                                           \
            __builtin_amdgcn_sched_barrier(0);                                                                          \
        }                                                                                                               \
        mfma_group(1);                                                                                                  \
        }                                                                                                               \
        }                                                                                                               \
        if (validL) lds_barrier<NX>();                                                                                  \
        else lds_barrier<0>();                                                                                          \
        /* Cout = 64 (mod 128), round 5: a wave whose 64-column group lies past Cout multiplied zeros (its filter rows are beyond the descriptor: the  */ \
        /* LDS-DMA wrote zeros) and stores nothing. Block 6-7 of the 0.5x network (64 -> 64 channels) is inside the envelope with it.               */ \
        if (kM == nk - 1 && !(dbg & 4) && n0M + wn < a.cout) {                                                        \
            if constexpr (M16) {                                                                                        \
            if (m0M + BM <= mtot) mbn_store_relu6_bf16_pair16<MI16, NI16, 0>(orsrc, (unsigned)a.cout, m0M + wm, n0M + wn, lane, acc16, sc3_s, sh3_s); \
            else mbn_store_relu6_bf16_pair16<MI16, NI16, 1>(orsrc, (unsigned)a.cout, m0M + wm, n0M + wn, lane, acc16, sc3_s, sh3_s);               \
            } else {                                                                                                    \
            if (m0M + BM <= mtot) mbn_store_relu6_bf16_pair<MI, NI, 0>(orsrc, (unsigned)a.cout, m0M + wm, n0M + wn, lane, acc, sc3_s, sh3_s); \
            else mbn_store_relu6_bf16_pair<MI, NI, 1>(orsrc, (unsigned)a.cout, m0M + wm, n0M + wn, lane, acc, sc3_s, sh3_s);               \
            }                                                                                                           \
        }                                                                                                               \
        if (!validD) break;                                                                                             \
        vbM = vbD; kM = kD; m0M = m0D; n0M = n0D;                                                                       \
        vbD = vbL; kD = kL; m0D = m0L; n0D = n0L; validD = validL;                                                      \
    }

    for (;;) {
        MBN_DWPW2_STEP(0)
        MBN_DWPW2_STEP(1)
    }
#undef MBN_DWPW2_STEP
}

template <int S, int BN>
void launch2(DwPw2Args &a, hipStream_t s, int num_cus)
{
    constexpr int BM = BM8, NT = 512;
    if constexpr (BN == 128) {
        bool h32 = a.cin == 32 && a.fast_off;                                      // H32: Cin = 32 on the 256-row tile (BM_H32 rows: the kernel's BM)
#ifdef MBN_LAB
        if (g_mbn_tune.exp0 == 53) h32 = false;                                    // lab A/B: the padded 128-row form instead
#endif
        if (h32) {
            a.mt = (int)((a.m + BM_H32 - 1) / BM_H32);
            a.nt = (a.cout + BN - 1) / BN;
            long g32 = num_cus;
            if (g32 > (long)a.mt * a.nt) g32 = (long)a.mt * a.nt;
            hipLaunchKernelGGL((dwpw2_bf16<S, 128, false, false, true, 8, true>), dim3((unsigned)g32), dim3(NT), 0, s, a);
            return;
        }
    }
    a.mt = (int)((a.m + BM - 1) / BM);
    a.nt = (a.cout + BN - 1) / BN;          // Cout = 64 (mod 128): the last tile's upper 64 columns are padding (round 5, see the epilogue)
    const long nwg = (long)a.mt * a.nt;
    long grid = num_cus;
    if (grid > nwg) grid = nwg;
#ifdef MBN_LAB
    if (a.dbg) { hipLaunchKernelGGL((dwpw2_bf16<S, BN, true, false>), dim3((unsigned)grid), dim3(NT), 0, s, a); return; }
    if (g_mbn_tune.misc == 32) { hipLaunchKernelGGL((dwpw2_bf16<S, BN, false, true>), dim3((unsigned)grid), dim3(NT), 0, s, a); return; }   // A/B: the 16x16x32 form
#endif
#ifdef MBN_LAB
    if constexpr (BN == 128) {
        // LAB (exp2 = 44; profiles/r05/u_*): two 4-wave workgroups per CU on 64-row tiles (see the kernel's NW). Measured equal to slower at bf16 0.5x160 (block group
        // 0.3161-0.3191 -> 0.3205-0.3223 ms) and at 1.0x224 (0.706 -> 0.748): the bf16 blocks keep the VALU busy in either form. Not shipped.
        const long mt4 = (a.m + 63) / 64, nwg4 = mt4 * a.nt;
        if (a.fast_off && a.use4 && a.cin <= CMAX4 && a.cout <= NOUT4 && nwg4 >= 4L * num_cus) {
            a.mt = (int)mt4;
            long g4 = 2L * num_cus;
            hipLaunchKernelGGL((dwpw2_bf16<S, 128, false, false, true, 4>), dim3((unsigned)g4), dim3(256), 0, s, a);
            return;
        }
    }
#endif
    if (a.fast_off) hipLaunchKernelGGL((dwpw2_bf16<S, BN, false, false, true>), dim3((unsigned)grid), dim3(NT), 0, s, a);
    else hipLaunchKernelGGL((dwpw2_bf16<S, BN, false, false, false>), dim3((unsigned)grid), dim3(NT), 0, s, a);
}

}   // namespace

// Unified-wave form of mbn_launch_bf16_dwpw (same envelope: mbn_bf16_dwpw_check has been passed by the caller).
int mbn_launch_bf16_dwpw2(mbn_context *ctx, hipStream_t stream, void *out, const void *in, const float *wd, const float *s2,
                          const float *b2, const void *wp, const float *s3, const float *b3, int batch, int in_rows,
                          int in_cols, int out_rows, int out_cols, int cin, int cout, int stride, int pad_top, int pad_left)
{
    DwPw2Args a;
    a.out = (__bf16 *)out; a.in = (const __bf16 *)in; a.wp = (const __bf16 *)wp;
    a.wd = wd; a.s2 = s2; a.b2 = b2; a.s3 = s3; a.b3 = b3;
    a.m = (long)batch * out_rows * out_cols;
    a.h = in_rows; a.w = in_cols; a.ho = out_rows; a.wo = out_cols;
    a.cin = cin; a.cout = cout; a.pad_top = pad_top; a.pad_left = pad_left;
    mbn_udiv_magic((unsigned)out_cols, &a.wo_m, &a.wo_s);
    mbn_udiv_magic((unsigned)out_rows, &a.ho_m, &a.ho_s);
    const int variant = g_mbn_tune.dwpw_variant;
    a.in_bytes = (unsigned)(2.0 * batch * in_rows * in_cols * cin);
    a.wp_bytes = (unsigned)(2.0 * cin * cout);
    a.dbg = variant >= 100 ? variant - 100 : 0;
    a.inv_wo = 1.0f / (float)out_cols;
    a.inv_ho = 1.0f / (float)out_rows;
    a.use4 = 0;
    // the full-rate offsets' range: every input byte offset PLUS a left-pad column stays below the invalid-column constant 0x70000000
    // (a left-pad tap of image 0 / row 0 has base = -pad_left * cs: its sum with the constant must not wrap into the descriptor's range), the
    // row / image quotients stay in exact float range, (n h + iy0) in mul24 range
    const double cs_b = 2.0 * cin;
    a.fast_off = (2.0 * batch * in_rows * in_cols * cin + (pad_left + 1) * cs_b <= (double)0x70000000u && (double)batch * in_rows < 8388000.0 &&
                  in_cols < 32768 && out_cols < 32768 && out_rows < 32768 && pad_left <= 1) ? 1 : 0;
#ifdef MBN_LAB
    a.use4 = g_mbn_tune.exp2 == 44 ? 1 : 0;
    if (g_mbn_tune.exp0 == 51) a.fast_off = 0;                                     // lab A/B: the general offsets
#endif
    // 256-column tiles only when they alone fill the chip; pw_tile=1: force the 128-column tile (A/B hook)
    const bool wide = (cout % 256) == 0 && g_mbn_tune.pw_tile != 1 && ((a.m + BM8 - 1) / BM8) * (cout / 256) >= ctx->num_cus;
    if (stride == 1) {
        if (wide) launch2<1, 256>(a, stream, ctx->num_cus);
        else launch2<1, 128>(a, stream, ctx->num_cus);
    } else {
        if (wide) launch2<2, 256>(a, stream, ctx->num_cus);
        else launch2<2, 128>(a, stream, ctx->num_cus);
    }
    return MBN_OK;
}
