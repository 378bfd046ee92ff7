// mbn_f32_dwpw2.hip — fused depthwise 3x3 -> pointwise 1x1 block for gfx950, fp32, UNIFIED-WAVE form (round 2).
// Same contract as mbn_f32_dwpw.hip (one launch replaces a `depthwise` + `pointwise` pair of the reference's sequence,
// kernel.cl:62-92 + 94-114, pairs L4-5 ... of MobileNet.c:322-2599; bit-identical to the two separate launches):
//
//   out[m][n] = relu6( s3[n] * sum_c relu6( s2[c] * sum_{dy,dx} in[pix(m)+(dy,dx)][c] * wd[dy][dx][c] + b2[c] ) * wp[n][c] + b3[n] )
//
// Why a second form. On a gfx950 SIMD fp32 MFMA time and VALU time ADD (tools/micro/: v_mfma_f32_32x32x2_f32 runs at the
// fp32 vector rate and a VALU wave gets no issue slot beside an MFMA-streaming wave), so the best a fused block can do
// is (MFMA cycles) + (VALU instructions x ~5 cycles). The round-1 kernel split the workgroup into 8 producer (VALU) and
// 8 consumer (MFMA) waves: the producers were starved while the consumers streamed, so their whole dependent chain —
// LDS-DMA wait, eleven LDS weight reads, 52 FMAs, two LDS writes — ran after the consumers' MFMAs with every latency
// exposed: ~3000 cycles per 32-channel chunk on top of 4096-8192 MFMA cycles (0.36 HBM / 0.50 MFMA, BENCH_r01).
// Here every wave does BOTH jobs in one in-order instruction stream: its 1/8 of the depthwise chunk (2 pixels x 4
// channels per lane) and its 32x64 / 64x64 MFMA tile, so nothing is starved, there is one barrier per chunk and no
// cross-wave hand-over besides it, and each operand of the VALU part is in registers before the part starts:
//   * x window of chunk g+1: buffer_load_dwordx4 issued one chunk ahead (taps outside the image carry an out-of-range
//     offset, the buffer unit returns 0: zero padding costs no VALU and no branches);
//   * depthwise taps + scale/shift of chunk g+1: read from LDS before the previous barrier (44 VGPRs; 8 waves per
//     workgroup = 2 per SIMD = 256 VGPRs per wave make room for them next to the 64 accumulators);
//   * pointwise filter chunk [BN][32]: buffer_load ... lds (descriptor + fixed per-lane offset + scalar K offset),
//     issued first thing after the barrier; its wait sits in front of the NEXT barrier behind a whole chunk of MFMAs.
// Per chunk and wave: D(g+1) = 36 v_pk_fma + 8 fma + 8 med3 + 2 ds_write_b128; L(g+2) = 12-15 buffer loads;
// M(g) = 32 / 64 MFMAs with the fragment reads of group i+1 ahead of the MFMAs of group i (as in mbn_f32_pw.hip).
// Measured and rejected here (profiles/r02/b_block_kernel_variants.txt): the transposed product (operands swapped so a
// lane holds 4 consecutive output channels and the epilogue is 8 buffer_store_dwordx4 instead of 32 dword stores per
// 32x64 wave tile): 4-8 % SLOWER on every block — 64 scattered 16-byte segments per store instruction cost more in the
// memory pipeline than two full 128-byte lines. Ablation of one step (block 6-7, same file): without the epilogue
// stores -13 %, without the x-window loads -13 %, without the depthwise math -7 %, without the filter DMA -4 %.
// Also measured and rejected (profiles/r02/b_block_kernel_variants.txt, second part): issuing a finished tile's epilogue one
// step later, behind the next step's filter DMA and x loads, so the counted wait in front of the barrier need not drain
// the stores — the accumulators then stay live across the step (256 VGPRs in the 256-column variants) and every block
// measured 3-7 % slower than with the epilogue right behind the barrier. The channel-paired 8-byte stores are neutral
// in fp32 (within 1 %): the epilogue's cost is the burst itself, not the number of store instructions.
// The loop is unrolled by two so the LDS buffer index is a literal in every address (the waitcnt pass then keeps the
// LDS-DMA of buffer p^1 apart from the fragment reads of buffer p instead of draining vmcnt before each ds_read).
#include "mbn_internal.h"
#include "mbn_epilogue.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef mbn_f16v f16v;

constexpr int BKF = 32;
constexpr int BM8 = 128;                       // rows of the shipped 8-wave tile (2 waves per SIMD, 256 VGPRs each); the 16-wave lab form: 256
constexpr int NOUT = 1024;                     // widest pointwise output whose scale/shift the LDS copy holds
constexpr int CMAX = 1024;                     // largest Cin (depthwise constants resident in LDS: 44 KB)
constexpr int CMAX4 = 256, NOUT4 = 256;        // ... of the 4-wave form (two workgroups per CU: 66 KB of LDS each)
constexpr unsigned OOB = 0xF0000000u;          // byte offset beyond any supported tensor: the load returns zeros

// In-kernel stamps (diagnostic: dwpw_variant = 100 + 64): workgroup 0 records s_memtime at five points of its first 96 steps,
// per wave; read back with mbn_debug_dwpw2_stamps (tools/stamp_dwpw2.py). No output value depends on them.
__device__ unsigned long long g_dwpw2_stamps[8][96][6];

struct DwPw2Args {
    float *out;
    const float *in, *wd, *s2, *b2, *wp, *s3, *b3;
    long m;                 // output pixels = batch * ho * wo
    int h, w, ho, wo;       // input / output map sides
    int cin, cout;
    int pad_top, pad_left;
    int mt, nt;
    unsigned in_bytes, wp_bytes;
    int stagger;            // start delay of a workgroup in units of 1024 cycles x its phase (see the kernel): 0 = all workgroups in phase
    int dbg;                // experiments (tune dwpw_variant = 100 + bits): 1 = no x loads after the first, 2 = no depthwise math, 4 = no output stores, 128 = x loads as one burst (before: -2.5 %), 256 = no s_setprio around the depthwise part (-1 %),
                            // 8 = no filter DMA, 16 = no MFMA, 32 = unpaired column blocks (4-byte stores); round 5 (timing only): 1024 = no tap reads after the first chunk,
                            // 2048 = only the first fragment read of a step (the MFMAs reuse registers), 4096 = no barrier (waits only), 8192 = no epilogue arithmetic/zeroing either (with 4)
    unsigned wo_m, wo_s, ho_m, ho_s;   // floor(v / wo) = umulhi(v, wo_m) >> wo_s for v < 2^31 (m == 0: the divisor is 1)
    int fast_off;           // launcher: 1 = the FO instantiation (set_offsets in its full-rate form: input < 0x70000000 bytes); 0 = the general form (rounds 2-4; lab A/B: exp0 = 51)
    float inv_wo, inv_ho;   // 1 / wo, 1 / ho
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
template <int B_LD, int NT, int BN>
__device__ __forceinline__ void dma_filter(__amdgpu_buffer_rsrc_t rsrc, float *lds_b, const unsigned *voff, int soff, int wave_u)
{
#pragma unroll
    for (int p = 0; p < B_LD; p++)
        if ((B_LD * NT == BN * 8) || p * (NT / 8) + wave_u * 8 < BN)       // 12 waves: the second round of pieces exists for waves 0-3 only (wave-uniform)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)(lds_b + (p * (NT / 8) + wave_u * 8) * BKF),
                                                     16, voff[p], soff, 0, 0);
}

// DBG = false: the shipped kernel — every experiment switch (a.dbg) folds away at compile time, which takes ~40 scalar branches and the
// conservative waits around them out of the step. DBG = true is launched only for tune dwpw_variant >= 100.
// NW = 16 (lab, BN = 128 only): the same kernel on a 256-row tile with 16 waves = 4 per SIMD at <= 128 VGPRs (taps read from LDS inside the step)
// XA2 (round 4): the x window of a chunk is requested TWO steps ahead of the depthwise part that consumes it, into a second register set. In the one-ahead
// form the window's four load groups are issued under the MFMA groups of the step BEFORE the one that needs them — the last group a few hundred cycles
// ahead of the barrier behind which the depthwise part starts — so part of every window waits out its HBM latency inside D (stamps, profiles/r02/g_*:
// D 1283 cycles for ~54 VALU instructions). Two ahead, every load has a whole step (~7 k cycles) before the barrier that (vmcnt in order) completes it.
// Costs 4 NX VGPRs: fits next to the taps (PRE) for S = 1 / BN = 128, with the taps read inside the step (PRE = false) for S = 1 / BN = 256 and S = 2 / BN = 128.
// IL (round 5, the shipped form): the depthwise part of a step is no longer one VALU burst in front of the MFMAs. The stamps (profiles/r02/g_*) show what the
// burst form does on a SIMD with two waves: both waves run their depthwise parts at the same time right behind the barrier (no MFMA in flight: every LDS /
// vector-memory / scalar instruction and every wait of that part is exposed), then their MFMA parts one after the other (waves 0-3: M 2300 cycles + 2600-4000
// waiting at the barrier; waves 4-7 the reverse). A VALU instruction gets no issue slot beside another wave's MFMA stream and costs its issue time on top of
// the wave's own (tools/micro/mfma_valu_*), so VALU time adds whatever the order — but everything else need not. Here the part is cut by filter row: row dy's
// 12 v_pk_fma + the reload of the x row they consumed sit in front of MFMA group dy, BN + ReLU6 + the two LDS writes in front of group 3; the filter DMA opens
// the step. The two waves fall one group apart by themselves (one's MFMA group runs while the other does its loads, LDS reads, scalar work and waits) and the
// matrix pipe has work at every point of the step. Same fma order: bit-identical. The last step of a workgroup issues its DMA / loads on stale cursors
// (valid addresses, results unused) instead of branching around them: the counted waits are the same in every step.
// FO: set_offsets in its full-rate form (see there); the launcher takes the FO = false instantiation for inputs outside that form's range.
template <int S, int BN, bool PRE, bool DBG, int NW = 8, bool XA2 = false, bool IL = false, bool FO = false>
__global__ __launch_bounds__(64 * NW) void dwpw2_f32(DwPw2Args a)
{
    static_assert(!XA2 || (!DBG && NW == 8), "XA2: shipped 8-wave form only");
    static_assert(!IL || ((NW == 8 || NW == 4) && !XA2), "IL: 8-wave form, or 4 waves on a 64-row tile with two workgroups per CU");
    const int dbg = DBG ? a.dbg : 0;
    constexpr int NT = 64 * NW, BM = 16 * NW;
    constexpr int WN = 64, WM = BN == 256 ? 64 : 32;   // wave tile: 8 waves as 2 x 4 (BN 256) or 4 x 2 (BN 128)
    constexpr int WAVES_N = BN / WN;
    static_assert((BM / WM) * WAVES_N == NW, "wave grid");
    constexpr int MI = WM / 32, NI = WN / 32;
    constexpr int B_LD = (BN * 8 + NT - 1) / NT;       // 16-B filter pieces per lane per chunk (2 / 4)
    constexpr int XC = S + 3;                          // input columns feeding 2 adjacent output pixels
    constexpr int NX = 3 * XC;                         // buffer loads per lane per chunk
    constexpr int ABUF = BM * BKF, BBUF = BN * BKF;
    // NW = 4 (two workgroups per CU): depthwise constants of <= 256 input channels (11 KB), scale/shift of <= 256 outputs: 66 KB per workgroup
    constexpr int CMAXK = NW == 4 ? CMAX4 : CMAX, NOUTK = NW == 4 ? NOUT4 : NOUT;
    __shared__ __attribute__((aligned(16))) float lds[2 * ABUF + 2 * BBUF + 11 * CMAXK + 2 * NOUTK];
    float *const a_s0 = lds, *const b_s0 = lds + 2 * ABUF, *const wd_s = b_s0 + 2 * BBUF, *const sb_s = wd_s + 9 * CMAXK;
    // pointwise scale | shift of all Cout channels: the epilogue reads them with ds_read. As global loads they were the wave's
    // youngest vector-memory operations, and waiting for them (in-order vmcnt) drained the x-window loads and the filter DMA
    // already in flight for the next steps: 1700-3100 cycles per tile in the stamps (profiles/r02/g_dwpw2_stamps.txt)
    float *const sc3_s = sb_s + 2 * CMAXK, *const sh3_s = sc3_s + NOUTK;

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
    // De-phasing. All workgroups run the same steps on the same amount of work, so they reach their epilogues together: a
    // chip-wide store burst during which — vmcnt retires in order — no wave sees a younger load complete (round 3: loads and stores of
    // such a kernel add up instead of overlapping). A quarter of the workgroups starts immediately, the others 1, 2, 3 x `stagger`
    // kcycles later, so that at any time some workgroups load while others store. Arithmetic untouched.
    if (a.stagger > 0) {
        const int phase = ((int)blockIdx.x >> 3) & 3;
        const long long t0 = __builtin_readcyclecounter();
        const long long wait = (long long)phase * a.stagger * 1024;
        while (__builtin_readcyclecounter() - t0 < wait) __builtin_amdgcn_s_sleep(32);
    }

    // ---- roles of this lane
    const int c4 = tid & 7, pair = tid >> 3;                        // depthwise: tile rows 2*pair, 2*pair+1, channels 4*c4..+3 of the chunk
    const int wm = (wave_u / WAVES_N) * WM, wn = (wave_u % WAVES_N) * WN;      // MFMA tile origin inside the workgroup tile
    const int li = lane & 31, lh = lane >> 5;
    const __amdgpu_buffer_rsrc_t irsrc = mbn_make_rsrc(a.in, a.in_bytes);
    const __amdgpu_buffer_rsrc_t wrsrc = mbn_make_rsrc(a.wp, a.wp_bytes);
    const __amdgpu_buffer_rsrc_t orsrc = mbn_make_rsrc(a.out, (unsigned)(a.m * a.cout * 4));
    const int aw0 = swz(2 * pair, c4), aw1 = swz(2 * pair + 1, c4);           // A-tile slots this lane writes
    int fr_a[4], fr_b[4];                                                       // fragment slots this lane reads
#pragma unroll
    for (int g = 0; g < 4; g++) {
        fr_a[g] = swz(wm + li, 2 * g + lh);
        fr_b[g] = swz(wn + li, 2 * g + lh);
    }
    const bool paired = !(dbg & 32);
    unsigned b_vo[B_LD];                                                        // filter piece offsets: fixed for the kernel, the tile's
#pragma unroll                                                                  // column origin and the chunk go into the scalar offset
    for (int p = 0; p < B_LD; p++) {
        const int row = ((p * NT + tid) >> 3) & (BN - 1);       // (12 waves: the unused pieces of the second round wrap; never issued)
        // channel-paired column blocks (mbn_epilogue.h): LDS filter row `row` holds output channel mbn_pair_channel(row), so the
        // epilogue stores 8 bytes per lane, 256 contiguous bytes per pixel row; `pair` = dbg bit 5 switches it off (A/B)
        b_vo[p] = ((unsigned)(paired ? mbn_pair_channel(row) : row) * (unsigned)a.cin + (unsigned)(((c4 ^ (row >> 1)) & 7) * 4)) * 4u;
    }
    const float *wk = wd_s + c4 * 4;                                           // depthwise taps of this lane's 4 channels (+ kc*32 + tap*cin)
    const float *sk = sb_s + c4 * 4;

    unsigned off[3][XC];
    auto set_offsets_general = [&](unsigned m0) __attribute__((always_inline)) {
        const unsigned m = m0 + 2 * pair;
        const bool mok = m < mtot;
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
    };
    // Round 5: the same offsets from full-rate instructions. The stamps of the general form (profiles/r05/d_*) read ~1100-1250 cycles per tile for it on a
    // SIMD's two waves: two v_mul_hi_u32 and six v_mul_lo_u32 (quarter rate), a 64-bit mad, and 12-15 compare / select pairs under exec-mask branches.
    // Here: (n, y, x) of the tile's first pixel on the scalar unit (magic division); the lane's own pixel from it by two float reciprocal divisions of
    // small numbers (r < wo + 128, exact: (r + 0.5) / wo is >= 0.5 / wo away from an integer, the float error is < 2e-5); ONE 32-bit multiply for the
    // byte offset; validity separable by row and column: off[dy][j] = rowv[dy] + colv[j] where an invalid row is 0x80000000 and an invalid column
    // 0x70000000, so that any sum with an invalid term lies in [0x70000000, 0xF0005000) — beyond the descriptor's num_records (launch2 takes this form
    // only for inputs < 0x70000000 bytes) and without wrapping. Same offsets for every valid tap, zeros for every other: bit-identical results.
    auto set_offsets_fast = [&](unsigned m0) __attribute__((always_inline)) {
        const unsigned q0 = a.wo_m ? __umulhi(m0, a.wo_m) >> a.wo_s : m0;                    // wave-uniform: scalar unit
        const unsigned x0 = m0 - q0 * (unsigned)a.wo;
        const unsigned n0 = a.ho_m ? __umulhi(q0, a.ho_m) >> a.ho_s : q0;
        const unsigned y0 = q0 - n0 * (unsigned)a.ho;
        const unsigned r = x0 + 2u * (unsigned)pair;
        const unsigned q1 = (unsigned)__builtin_fmaf((float)r, a.inv_wo, 0.5f * a.inv_wo);
        const unsigned x = r - q1 * (unsigned)a.wo;                                       // q1 < 2^8, wo < 2^16: mul24 range
        const unsigned yy = y0 + q1;
        const unsigned q2 = (unsigned)__builtin_fmaf((float)yy, a.inv_ho, 0.5f * a.inv_ho);
        const unsigned y = yy - q2 * (unsigned)a.ho;
        const unsigned n = n0 + q2;
        const bool mok = m0 + 2u * (unsigned)pair < mtot;
        const int iy0 = (int)y * S - a.pad_top, ix0 = (int)x * S - a.pad_left;
        const unsigned cs = (unsigned)a.cin * 4u, rs = (unsigned)a.w * cs;
        const int pix = __mul24((int)(n * (unsigned)a.h) + iy0, a.w) + ix0;               // (n h + iy0) < 2^23 (launch2 checks), w < 2^16
        const unsigned base = (unsigned)pix * cs + (unsigned)(c4 * 4) * 4u;
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
    f4 xr[XA2 ? 2 : 1][3][XC];
    auto ldx_set = [&](int kc, const int set) __attribute__((always_inline)) {
#pragma unroll
        for (int dy = 0; dy < 3; dy++)
#pragma unroll
            for (int j = 0; j < XC; j++)
                xr[set][dy][j] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(irsrc, off[dy][j], kc * 128, 0));
    };
    auto ldx = [&](int kc) __attribute__((always_inline)) { ldx_set(kc, 0); };
    // the same loads in four parts (3+3+3+3 or 4+4+4+3), one in front of each MFMA group of the step: issued as one burst
    // right after D, the 8 waves' 96-120 loads queue up behind each other in the CU's one address unit — ~800 cycles of issue
    // stall per wave and step in the stamps (profiles/r02/g_dwpw2_stamps.txt); spread out they issue under the MFMAs
    auto ldx_part_set = [&](int kc, const int part, const int set) __attribute__((always_inline)) {
        constexpr int PER = (NX + 3) / 4;
#pragma unroll
        for (int i = part * PER; i < (part + 1) * PER && i < NX; i++)
            xr[set][i / XC][i % XC] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(irsrc, off[i / XC][i % XC], kc * 128, 0));
    };
    auto ldx_part = [&](int kc, const int part) __attribute__((always_inline)) { ldx_part_set(kc, part, 0); };
    f4 wreg[11];                                                               // 9 taps, scale, shift of the chunk D works on
    auto ldw = [&](int kc) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 9; t++) wreg[t] = *reinterpret_cast<const f4 *>(wk + kc * 32 + t * a.cin);
        wreg[9] = *reinterpret_cast<const f4 *>(sk + kc * 32);
        wreg[10] = *reinterpret_cast<const f4 *>(sk + a.cin + kc * 32);
    };
    // depthwise + BN + ReLU6 of the chunk in xr/wreg into A buffer `buf` (same fma order as mbn_f32_dw.hip: bit-identical)
    auto dw_set = [&](int kc, const int buf, const int set) __attribute__((always_inline)) {
        f4 acc0 = f4{ 0.f, 0.f, 0.f, 0.f }, acc1 = acc0;
        if constexpr (NW > 8) {
            // register-lean forms (> 2 waves per SIMD): one filter row of taps at a time, read where it is used (same fma order: same bits)
#pragma unroll
            for (int dy = 0; dy < 3; dy++) {
                f4 t3[3];
#pragma unroll
                for (int dx = 0; dx < 3; dx++) t3[dx] = *reinterpret_cast<const f4 *>(wk + kc * 32 + (dy * 3 + dx) * a.cin);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int dx = 0; dx < 3; dx++) {
                    acc0 = __builtin_elementwise_fma(xr[set][dy][dx], t3[dx], acc0);
                    acc1 = __builtin_elementwise_fma(xr[set][dy][dx + S], t3[dx], acc1);
                }
            }
            const f4 sc2 = *reinterpret_cast<const f4 *>(sk + kc * 32), sh2 = *reinterpret_cast<const f4 *>(sk + a.cin + kc * 32);
            *reinterpret_cast<f4 *>(a_s0 + buf * ABUF + aw0) = bn_relu6(acc0, sc2, sh2);
            *reinterpret_cast<f4 *>(a_s0 + buf * ABUF + aw1) = bn_relu6(acc1, sc2, sh2);
            return;
        }
        if (!PRE) ldw(kc);
#pragma unroll
        for (int dy = 0; dy < 3; dy++)
#pragma unroll
            for (int dx = 0; dx < 3; dx++) {
                acc0 = __builtin_elementwise_fma(xr[set][dy][dx], wreg[dy * 3 + dx], acc0);
                acc1 = __builtin_elementwise_fma(xr[set][dy][dx + S], wreg[dy * 3 + dx], acc1);
            }
        *reinterpret_cast<f4 *>(a_s0 + buf * ABUF + aw0) = bn_relu6(acc0, wreg[9], wreg[10]);
        *reinterpret_cast<f4 *>(a_s0 + buf * ABUF + aw1) = bn_relu6(acc1, wreg[9], wreg[10]);
    };
    auto dw = [&](int kc, const int buf) __attribute__((always_inline)) { dw_set(kc, buf, 0); };
    // IL: the same arithmetic in pieces. ldx_row: the XC loads of one window row; dw_row: that row's taps into the two running sums (dy = 0 starts them);
    // dw_fin: BN + ReLU6 + the two A-tile writes.
    f4 dacc0 = f4{ 0.f, 0.f, 0.f, 0.f }, dacc1 = dacc0;
    auto ldx_row = [&](int kc, const int dy) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < XC; j++)
            xr[0][dy][j] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(irsrc, off[dy][j], kc * 128, 0));
    };
    // taps just in time (IL): one filter row (3 x f4) in registers at a time — row dy + 1 is read from LDS right behind the FMAs of row dy and has a
    // whole MFMA group to arrive; scale / shift behind row 2; row 0 of the next chunk behind dw_fin. 20 VGPRs instead of wreg's 44 (190 instead of 208 for
    // the 128-column tile). Same 11 ds_read_b128 per step.
    f4 wrow[3], wss[2];
    auto ldw_row = [&](int kc, const int dy) __attribute__((always_inline)) {
#pragma unroll
        for (int dx = 0; dx < 3; dx++) wrow[dx] = *reinterpret_cast<const f4 *>(wk + kc * 32 + (dy * 3 + dx) * a.cin);
    };
    auto ldw_ss = [&](int kc) __attribute__((always_inline)) {
        wss[0] = *reinterpret_cast<const f4 *>(sk + kc * 32);
        wss[1] = *reinterpret_cast<const f4 *>(sk + a.cin + kc * 32);
    };
    auto dw_row = [&](int kc, const int dy) __attribute__((always_inline)) {
        if (dy == 0) {
            dacc0 = f4{ 0.f, 0.f, 0.f, 0.f };
            dacc1 = dacc0;
        }
#pragma unroll
        for (int dx = 0; dx < 3; dx++) {
            dacc0 = __builtin_elementwise_fma(xr[0][dy][dx], wrow[dx], dacc0);
            dacc1 = __builtin_elementwise_fma(xr[0][dy][dx + S], wrow[dx], dacc1);
        }
        if (dy < 2) ldw_row(kc, dy + 1);
        else ldw_ss(kc);
    };
    auto dw_fin = [&](const int buf) __attribute__((always_inline)) {
        *reinterpret_cast<f4 *>(a_s0 + buf * ABUF + aw0) = bn_relu6(dacc0, wss[0], wss[1]);
        *reinterpret_cast<f4 *>(a_s0 + buf * ABUF + aw1) = bn_relu6(dacc1, wss[0], wss[1]);
    };

    f16v acc[MI][NI];
    f4 fa[2][MI], fb[2][NI];
    auto ldfrag = [&](const int buf, int g, int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int mi = 0; mi < MI; mi++) fa[slot][mi] = *reinterpret_cast<const f4 *>(a_s0 + buf * ABUF + fr_a[g] + mi * 32 * BKF);
#pragma unroll
        for (int ni = 0; ni < NI; ni++) fb[slot][ni] = *reinterpret_cast<const f4 *>(b_s0 + buf * BBUF + fr_b[g] + ni * 32 * BKF);
    };
    auto mfma_group = [&](int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < 4; s++)
#pragma unroll
            for (int mi = 0; mi < MI; mi++)
#pragma unroll
                for (int ni = 0; ni < NI; ni++)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[slot][mi][s], fb[slot][ni][s], acc[mi][ni], 0, 0, 0);
    };
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int mi = 0; mi < MI; mi++)
#pragma unroll
            for (int ni = 0; ni < NI; ni++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[mi][ni][r] = 0.f;
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

    // XA2: a third persistent cursor, L = successor of D (its window is already in flight / in registers: set (chunk ordinal) & 1)
    [[maybe_unused]] int vbL = 0, kL = 0, n0L = 0; [[maybe_unused]] unsigned m0L = 0;
    [[maybe_unused]] bool validL1 = false;

    // prologue: L(0), D(0) into buffer 0, L(1)  [XA2: and L(2)]
    vbM = blockIdx.x; kM = 0;
    origin(vbM, m0M, n0M);
    set_offsets(m0M);
    ldx(0);                                   // chunk 0 -> x set 0
    if (PRE) ldw(0);
    dma_filter<B_LD, NT, BN>(wrsrc, b_s0, b_vo, (n0M * a.cin + 0) * 4, wave_u);
    dw(0, 0);
    // D cursor = successor of M
    vbD = vbM; kD = 1; m0D = m0M; n0D = n0M; validD = true;
    if (kD >= nk) {
        kD = 0; vbD += gridDim.x; validD = vbD < nwg;
        if (validD) { origin(vbD, m0D, n0D); set_offsets(m0D); }
    }
    if (validD) {
        ldx_set(kD, XA2 ? 1 : 0);             // chunk 1 -> x set 1 (XA2) / the one set
        if (PRE && !IL) ldw(kD);
    }
    if constexpr (IL) ldw_row(kD, 0);         // IL: filter row 0 of the chunk the first step works on (kD = 0 when there is none: never used)
    if constexpr (XA2) {
        vbL = vbD; kL = kD + 1; m0L = m0D; n0L = n0D; validL1 = validD;
        if (validL1 && kL >= nk) {
            kL = 0; vbL += gridDim.x; validL1 = vbL < nwg;
            if (validL1) { origin(vbL, m0L, n0L); set_offsets(m0L); }
        }
        if (validL1) ldx_set(kL, 0);          // chunk 2 -> x set 0 (chunk 0's window has been consumed by dw(0, 0))
    }
    zero_acc();
    if constexpr (XA2) {
        if (validL1) lds_barrier<2 * NX>();   // filter chunk 0 landed; the two newer windows may fly
        else if (validD) lds_barrier<NX>();
        else lds_barrier<0>();
    } else {
    if (validD) lds_barrier<NX>();        // filter chunk 0 landed; the NX newer loads may fly
    else lds_barrier<0>();
    }

    // The finished tile's epilogue is issued in the NEXT step, behind that step's depthwise part and filter DMA and ahead of its MFMAs
    // (round 3). vmcnt retires in order: with the 16 / 32 stores issued right behind the barrier — ahead of the next step's filter DMA —
    // the counted wait in front of the next barrier (filter landed) also waited for every store to be acknowledged, once per tile.
    // Behind the DMA they are YOUNGER than what that wait needs and stay in flight (lds_barrier<NX + NST>). The accumulators are not
    // touched by the depthwise part, so no second set is needed (the r2 attempt deferred the stores behind the x loads too, i.e. under
    // the MFMAs, and paid for it in registers). Measured (profiles/r03/q_block_more_waves.txt, second table): block 6-7 -0.7 %, blocks
    // 4-5 and 8-9 unchanged — like r2's three-slot experiment it says the "-13 % without the stores" of the ablation is the store
    // traffic itself, not the wait for it.
    constexpr int NST = 16 * MI * (NI / 2);            // buffer_store_dwordx2 per lane of the channel-paired epilogue, both modes
    bool pendE = false;
    unsigned m0E = 0;
    int n0E = 0;
    auto epilogue = [&](unsigned m0, int n0) __attribute__((always_inline)) {
        if (paired) {
            if (m0 + BM <= mtot) mbn_store_relu6_f32_pair<MI, NI, 0>(orsrc, (unsigned)a.cout, m0 + wm, n0 + wn, lane, acc, sc3_s, sh3_s);
            else mbn_store_relu6_f32_pair<MI, NI, 1>(orsrc, (unsigned)a.cout, m0 + wm, n0 + wn, lane, acc, sc3_s, sh3_s);
        } else if (m0 + BM <= mtot) mbn_store_relu6_f32<MI, NI, 0>(orsrc, (unsigned)a.cout, m0 + wm, n0 + wn, lane, acc, sc3_s, sh3_s, mtot, a.cout);
        else mbn_store_relu6_f32<MI, NI, 1>(orsrc, (unsigned)a.cout, m0 + wm, n0 + wn, lane, acc, sc3_s, sh3_s, mtot, a.cout);
    };
    int stepno = 0;
    const bool stamping = (dbg & 64) && blockIdx.x == 0;
#define STAMP(k) do { if (stamping && stepno < 96) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (lane == 0) g_dwpw2_stamps[wave_u][stepno][k] = t_; } } while (0)
    // One chunk step with the MFMA chunk in buffer P (a literal at both call sites).
    // Returns false when the sequence is finished.
#define MBN_DWPW2_STEP(P)                                                                                               \
    {                                                                                                                   \
        STAMP(0);                                                                                                       \
        if (!(dbg & 256)) __builtin_amdgcn_s_setprio(3);       /* depthwise part ahead of the other wave's MFMAs, see below */ \
        ldfrag(P, 0, 0);                                                                                                \
        bool validL = false;                                                                                            \
        int vbL = vbD, kL = kD + 1, n0L = n0D;                                                                          \
        unsigned m0L = m0D;                                                                                             \
        if (validD) {                                                                                                   \
            if (!(dbg & 8) && (dbg & 512)) dma_filter<B_LD, NT, BN>(wrsrc, b_s0 + (P ^ 1) * BBUF, b_vo, (n0D * a.cin + kD * 32) * 4, wave_u); /* dbg 512: DMA ahead of the depthwise part (before) */ \
            if (!(dbg & 2)) dw(kD, P ^ 1);                                                                            \
            /* the filter DMA BEHIND the depthwise part: issued ahead of it, its two LDS-DMA operations were the wave's youngest */ \
            /* vector-memory operations when the depthwise math needed the x window, and the compiler's wait for the window     */ \
            /* (s_waitcnt vmcnt(1), vmcnt(0) in the ISA) waited out the DMA's whole L2 round trip at the start of every step   */ \
            if (!(dbg & 8) && !(dbg & 512)) dma_filter<B_LD, NT, BN>(wrsrc, b_s0 + (P ^ 1) * BBUF, b_vo, (n0D * a.cin + kD * 32) * 4, wave_u); \
            validL = true;                                                                                              \
            if (kL >= nk) {                                                                                             \
                kL = 0; vbL += gridDim.x; validL = vbL < nwg;                                                           \
                if (validL) { origin(vbL, m0L, n0L); set_offsets(m0L); }                                                \
            }                                                                                                           \
            STAMP(1);                                                                                                   \
            if (validL && !(dbg & 1) && ((dbg & 128) || (dbg & 16))) ldx(kL);      /* dbg 128: the burst form (A/B) */ \
        }                                                                                                               \
        const bool spreadL = validL && !(dbg & 1) && !(dbg & 128) && !(dbg & 16);                                 \
        const bool didE = pendE && paired && !(dbg & 4);       /* exactly NST stores are in flight behind the DMA */      \
        /* the counted waits below (vmcnt(NX + NST)) are right only if the NST stores are YOUNGER than this step's filter DMA and older than */ \
        /* its x loads in the instruction stream: pin the order (tools/check_counted_waits.py reads the ISA) */                    \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
        if (pendE) {                                                                                                    \
            if (!(dbg & 4)) epilogue(m0E, n0E);                                                                       \
            if (!(dbg & 8192)) zero_acc();                                                                              \
            pendE = false;                                                                                              \
        }                                                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
        __builtin_amdgcn_s_setprio(0);                                                                                  \
        STAMP(2);                                                                                                       \
        if (!(dbg & 16)) {                                                                                            \
        _Pragma("unroll") for (int g = 0; g < 3; g++) {                                                                 \
            if (!(dbg & 2048)) ldfrag(P, g + 1, (g + 1) & 1);                                                           \
            if (spreadL) ldx_part(kL, g);                                                                               \
            __builtin_amdgcn_sched_barrier(0);                                                                          \
            mfma_group(g & 1);                                                                                          \
            __builtin_amdgcn_sched_barrier(0);                                                                          \
        }                                                                                                               \
        if (spreadL) ldx_part(kL, 3);                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
        mfma_group(1);                                                                                                  \
        }                                                                                                               \
        STAMP(3);                                                                                                       \
        if (PRE && validL && !(dbg & 1024)) ldw(kL);                                                                    \
        if (dbg & 4096) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); }                                 \
        else if (validL) { if (didE) lds_barrier<NX + NST>(); else lds_barrier<NX>(); }                                 \
        else { if (didE) lds_barrier<NST>(); else lds_barrier<0>(); }                                                   \
        STAMP(4);                                                                                                       \
        if (kM == nk - 1) { pendE = true; m0E = m0M; n0E = n0M; }      /* stored in the next step (or behind the loop) */ \
        STAMP(5);                                                                                                       \
        stepno++;                                                                                                       \
        if (!validD) break;                                                                                             \
        vbM = vbD; kM = kD; m0M = m0D; n0M = n0D;                                                                       \
        vbD = vbL; kD = kL; m0D = m0L; n0D = n0L; validD = validL;                                                      \
    }

    // XA2 step with the MFMA chunk in buffer P: M(c_s) from A/B buffers P; D(c_s+1) from x set P^1 into buffers P^1; the window of c_s+3 is
    // requested into x set P^1 (just consumed) under the MFMA groups; taps (PRE) of c_s+2 behind the MFMAs. Barrier: the filter DMA of this step has
    // landed — everything older has too (vmcnt in order: c_s+2's window, requested a whole step ago) — the NX (+ NST) younger operations may fly.
#define MBN_DWPW2_STEP2(P)                                                                                              \
    {                                                                                                                   \
        __builtin_amdgcn_s_setprio(3);                                                                                  \
        ldfrag(P, 0, 0);                                                                                                \
        bool validL2 = false;                                                                                           \
        int vbL2 = vbL, kL2 = kL + 1, n0L2 = n0L;                                                                       \
        unsigned m0L2 = m0L;                                                                                            \
        if (validD) {                                                                                                   \
            dw_set(kD, P ^ 1, P ^ 1);                                                                                   \
            dma_filter<B_LD, NT, BN>(wrsrc, b_s0 + (P ^ 1) * BBUF, b_vo, (n0D * a.cin + kD * 32) * 4, wave_u);          \
            if (validL1) {                                                                                              \
                validL2 = true;                                                                                         \
                if (kL2 >= nk) {                                                                                        \
                    kL2 = 0; vbL2 += gridDim.x; validL2 = vbL2 < nwg;                                                   \
                    if (validL2) { origin(vbL2, m0L2, n0L2); set_offsets(m0L2); }                                       \
                }                                                                                                       \
            }                                                                                                           \
        }                                                                                                               \
        const bool didE = pendE;                               /* exactly NST stores are in flight behind the DMA */      \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
        if (pendE) {                                                                                                    \
            epilogue(m0E, n0E);                                                                                         \
            zero_acc();                                                                                                 \
            pendE = false;                                                                                              \
        }                                                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
        __builtin_amdgcn_s_setprio(0);                                                                                  \
        _Pragma("unroll") for (int g = 0; g < 3; g++) {                                                                 \
            ldfrag(P, g + 1, (g + 1) & 1);                                                                              \
            if (validL2) ldx_part_set(kL2, g, P ^ 1);                                                                   \
            __builtin_amdgcn_sched_barrier(0);                                                                          \
            mfma_group(g & 1);                                                                                          \
            __builtin_amdgcn_sched_barrier(0);                                                                          \
        }                                                                                                               \
        if (validL2) ldx_part_set(kL2, 3, P ^ 1);                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
        mfma_group(1);                                                                                                  \
        if (PRE && validL1) ldw(kL);                                                                                    \
        if (validL2) { if (didE) lds_barrier<NX + NST>(); else lds_barrier<NX>(); }                                     \
        else { if (didE) lds_barrier<NST>(); else lds_barrier<0>(); }                                                   \
        if (kM == nk - 1) { pendE = true; m0E = m0M; n0E = n0M; }                                                       \
        if (!validD) break;                                                                                             \
        vbM = vbD; kM = kD; m0M = m0D; n0M = n0D;                                                                       \
        vbD = vbL; kD = kL; m0D = m0L; n0D = n0L; validD = validL1;                                                     \
        vbL = vbL2; kL = kL2; m0L = m0L2; n0L = n0L2; validL1 = validL2;                                                \
    }

    // IL step with the MFMA chunk in buffer P (see the template comment). Vector-memory order inside a step: filter DMA (B_LD pieces), [NST stores of the
    // previous tile], NX x loads — so the barrier's wait for the DMA leaves exactly the NX (+ NST) youngest operations in flight, in every step.
    // Measured and NOT taken (round 5, profiles/r05/g_*, h_*): the step in which a tile's epilogue runs is +2000 cycles on block 6-7 and +5000 on block 4-5
    // (stamps) — the 8 waves' 128 store instructions go through the CU's one texture unit right behind the barrier with no MFMA in flight. Computing the
    // tile's 32 outputs into temporaries and issuing the stores four at a time under the MFMA groups needs 32 more live VGPRs; with the second loop body it
    // takes the compiler stops updating the accumulators in place (two sets of 32, alternating by step) and spills 10-125 VGPRs, and every scratch reload is
    // a `s_waitcnt vmcnt(0)` that drains the x loads: block 6-7 0.31 -> 0.34 ms, block 4-5 0.245 -> 0.50 ms. Left as the next thing to try with the MFMAs
    // pinned in place.
#define MBN_DWPW2_IL_GROUP(P, G)                                                                                        \
        ldfrag(P, (G) + 1, ((G) + 1) & 1);                                                                              \
        dw_row(kD, G);                                                                                                  \
        ldx_row(kL, G);                                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
        mfma_group((G) & 1);                                                                                            \
        __builtin_amdgcn_sched_barrier(0);

#define MBN_DWPW2_STEP_IL(P)                                                                                            \
    {                                                                                                                   \
        STAMP(0);                                                                                                       \
        ldfrag(P, 0, 0);                                                                                                \
        int vbL = vbD, kL = kD + 1, n0L = n0D;                                                                          \
        unsigned m0L = m0D;                                                                                             \
        bool validL = validD;                                                                                           \
        if (kL >= nk) {                                                                                                 \
            kL = 0; vbL += gridDim.x; validL = validD && vbL < nwg;                                                     \
            if (validL) { origin(vbL, m0L, n0L); set_offsets(m0L); }                                                    \
        }                                                                                                               \
        dma_filter<B_LD, NT, BN>(wrsrc, b_s0 + (P ^ 1) * BBUF, b_vo, (n0D * a.cin + kD * 32) * 4, wave_u);              \
        STAMP(1);                                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
        const bool didE = pendE;                                                                                        \
        if (pendE) {                                                                                                    \
            epilogue(m0E, n0E);                                                                                         \
            zero_acc();                                                                                                 \
            pendE = false;                                                                                              \
        }                                                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
        STAMP(2);                                                                                                       \
        MBN_DWPW2_IL_GROUP(P, 0)                                                                                        \
        MBN_DWPW2_IL_GROUP(P, 1)                                                                                        \
        MBN_DWPW2_IL_GROUP(P, 2)                                                                                        \
        dw_fin(P ^ 1);                                                                                                  \
        ldw_row(kL, 0);                                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
        mfma_group(1);                                                                                                  \
        STAMP(3);                                                                                                       \
        if (didE) lds_barrier<NX + NST>(); else lds_barrier<NX>();                                                      \
        STAMP(4);                                                                                                       \
        if (kM == nk - 1) { pendE = true; m0E = m0M; n0E = n0M; }      /* stored in the next step (or behind the loop) */ \
        STAMP(5);                                                                                                       \
        stepno++;                                                                                                       \
        if (!validD) break;                                                                                             \
        vbM = vbD; kM = kD; m0M = m0D; n0M = n0D;                                                                       \
        vbD = vbL; kD = kL; m0D = m0L; n0D = n0L; validD = validL;                                                      \
    }

    if constexpr (IL) {
        for (;;) {
            MBN_DWPW2_STEP_IL(0)
            MBN_DWPW2_STEP_IL(1)
        }
    } else if constexpr (XA2) {
        for (;;) {
            MBN_DWPW2_STEP2(0)
            MBN_DWPW2_STEP2(1)
        }
    } else {
    for (;;) {
        MBN_DWPW2_STEP(0)
        MBN_DWPW2_STEP(1)
    }
    }
    if (pendE && !(dbg & 4)) epilogue(m0E, n0E);               // the workgroup's last tile
#undef MBN_DWPW2_STEP
#undef MBN_DWPW2_STEP2
#undef MBN_DWPW2_STEP_IL
#undef MBN_DWPW2_IL_GROUP
#undef STAMP
}

template <int S, int BN>
void launch2(DwPw2Args &a, hipStream_t s, int num_cus, bool pre)
{
    constexpr int BM = BM8, NT = 512;
#ifdef MBN_LAB
    if (BN == 128 && (g_mbn_tune.dwpw_variant == 5 || g_mbn_tune.dwpw_variant == 6)) {      // r3 A/B: 16 / 12 waves on a 256- / 192-row tile
        const int nw = g_mbn_tune.dwpw_variant == 5 ? 16 : 12;
        a.mt = (int)((a.m + 16 * nw - 1) / (16 * nw));
        a.nt = a.cout / BN;
        long g16 = num_cus;
        if (g16 > (long)a.mt * a.nt) g16 = (long)a.mt * a.nt;
        if (nw == 16) hipLaunchKernelGGL((dwpw2_f32<S, 128, false, false, 16>), dim3((unsigned)g16), dim3(1024), 0, s, a);
        else hipLaunchKernelGGL((dwpw2_f32<S, 128, false, false, 12>), dim3((unsigned)g16), dim3(768), 0, s, a);
        return;
    }
#endif
#ifdef MBN_LAB
    if (BN == 128 && g_mbn_tune.dwpw_variant == 10 && a.cin <= CMAX4 && a.cout <= NOUT4) {      // r5 A/B: two 4-wave workgroups per CU on 64-row tiles (they de-phase by themselves)
        a.mt = (int)((a.m + 63) / 64);
        a.nt = a.cout / BN;
        long g4 = 2L * num_cus;
        if (g4 > (long)a.mt * a.nt) g4 = (long)a.mt * a.nt;
        if constexpr (BN == 128) {
            if (a.fast_off) hipLaunchKernelGGL((dwpw2_f32<S, 128, true, false, 4, false, true, true>), dim3((unsigned)g4), dim3(256), 0, s, a);
            else hipLaunchKernelGGL((dwpw2_f32<S, 128, true, false, 4, false, true, false>), dim3((unsigned)g4), dim3(256), 0, s, a);
        }
        return;
    }
#endif
    a.mt = (int)((a.m + BM - 1) / BM);
    a.nt = a.cout / BN;
    const long nwg = (long)a.mt * a.nt;
    long grid = num_cus;
    if (grid > nwg) grid = nwg;
#ifdef MBN_LAB                                                       // ablation / stamp builds and the taps-inside-the-step form: dwpw_variant
    if (g_mbn_tune.dwpw_variant == 7 || g_mbn_tune.dwpw_variant == 8) {      // r4 A/B: x window two steps ahead (7: where it fits with the taps in registers; 8: also the PRE = false forms)
        if constexpr (S == 1 && BN == 128) { hipLaunchKernelGGL((dwpw2_f32<S, BN, true, false, 8, true>), dim3((unsigned)grid), dim3(NT), 0, s, a); return; }
        else if constexpr (!(S == 2 && BN == 256)) {
            if (g_mbn_tune.dwpw_variant == 8) { hipLaunchKernelGGL((dwpw2_f32<S, BN, false, false, 8, true>), dim3((unsigned)grid), dim3(NT), 0, s, a); return; }
        }
    }
    if (a.dbg & 16384) {                                                     // r5: the stamps (bit 64) in the interleaved form
        if constexpr (BN == 128) { hipLaunchKernelGGL((dwpw2_f32<S, BN, true, true, 8, false, true, true>), dim3((unsigned)grid), dim3(NT), 0, s, a); return; }
    }
    if (a.dbg) {
        if (pre) hipLaunchKernelGGL((dwpw2_f32<S, BN, true, true>), dim3((unsigned)grid), dim3(NT), 0, s, a);
        else hipLaunchKernelGGL((dwpw2_f32<S, BN, false, true>), dim3((unsigned)grid), dim3(NT), 0, s, a);
        return;
    }
    if (!pre) { hipLaunchKernelGGL((dwpw2_f32<S, BN, false, false>), dim3((unsigned)grid), dim3(NT), 0, s, a); return; }
    if (g_mbn_tune.dwpw_variant == 9) { hipLaunchKernelGGL((dwpw2_f32<S, BN, true, false>), dim3((unsigned)grid), dim3(NT), 0, s, a); return; }   // r5 A/B: the burst form (rounds 2-4)
#endif
    (void)pre;
    constexpr bool FOK = BN == 128;                    // the 256-column shapes are at the 256-VGPR limit without it (spills with): general offsets there
    if (FOK && a.fast_off) hipLaunchKernelGGL((dwpw2_f32<S, BN, true, false, 8, false, true, FOK>), dim3((unsigned)grid), dim3(NT), 0, s, a);
    else hipLaunchKernelGGL((dwpw2_f32<S, BN, true, false, 8, false, true, false>), dim3((unsigned)grid), dim3(NT), 0, s, a);
}

}   // namespace

// Unified-wave form of mbn_launch_f32_dwpw (same envelope: mbn_f32_dwpw_check has been passed by the caller).
int mbn_launch_f32_dwpw2(mbn_context *ctx, hipStream_t stream, float *out, const float *in, const float *wd,
                         const float *s2, const float *b2, const float *wp, const float *s3, const float *b3, int batch,
                         int in_rows, int in_cols, int out_rows, int out_cols, int cin, int cout, int stride, int pad_top,
                         int pad_left)
{
    DwPw2Args a;
    a.out = out; a.in = in; a.wd = wd; a.s2 = s2; a.b2 = b2; a.wp = wp; a.s3 = s3; a.b3 = b3;
    a.m = (long)batch * out_rows * out_cols;
    a.h = in_rows; a.w = in_cols; a.ho = out_rows; a.wo = out_cols;
    a.cin = cin; a.cout = cout; a.pad_top = pad_top; a.pad_left = pad_left;
    mbn_udiv_magic((unsigned)out_cols, &a.wo_m, &a.wo_s);
    mbn_udiv_magic((unsigned)out_rows, &a.ho_m, &a.ho_s);
    const int variant = g_mbn_tune.dwpw_variant;
    a.in_bytes = (unsigned)(4.0 * batch * in_rows * in_cols * cin);
    a.wp_bytes = (unsigned)(4.0 * cin * cout);
    a.dbg = variant >= 100 ? variant - 100 : 0;
    a.inv_wo = 1.0f / (float)out_cols;
    a.inv_ho = 1.0f / (float)out_rows;
    // the full-rate offsets need every input byte offset below the invalid-column constant, and (n h + iy0) in mul24 range
    //  ... PLUS a left-pad column: a left-pad tap of image 0 / row 0 has base = -pad_left * cs, and its sum with the invalid-column
    // constant must stay beyond num_records without wrapping; out_rows bounded so the row quotient stays in exact float range
    const double cs_b = 4.0 * cin;
    a.fast_off = (4.0 * batch * in_rows * in_cols * cin + (pad_left + 1) * cs_b <= (double)0x70000000u && (double)batch * in_rows < 8388000.0 &&
                  in_cols < 32768 && out_cols < 32768 && out_rows < 32768 && pad_left <= 1) ? 1 : 0;
#ifdef MBN_LAB
    if (g_mbn_tune.exp0 == 51) a.fast_off = 0;                                     // lab A/B: the general offsets
#endif
    a.stagger = g_mbn_tune.exp2;                                             // lab: start stagger of the workgroups in kcycles per phase
    const bool pre = variant != 3;                                       // 3: taps read from LDS inside the step (A/B hook)
    // 256-column tiles only when they alone fill the chip (see mbn_dwpw_fused); pw_tile=1: force the 128-column tile (A/B hook)
    const bool wide = (cout % 256) == 0 && g_mbn_tune.pw_tile != 1 && ((a.m + BM8 - 1) / BM8) * (cout / 256) >= ctx->num_cus;
    if (stride == 1) {
        if (wide) launch2<1, 256>(a, stream, ctx->num_cus, pre);
        else launch2<1, 128>(a, stream, ctx->num_cus, pre);
    } else {
        if (wide) launch2<2, 256>(a, stream, ctx->num_cus, pre);
        else launch2<2, 128>(a, stream, ctx->num_cus, pre);
    }
    return MBN_OK;
}

// Diagnostic: copies the stamps of the last stamped launch (workgroup 0: [wave 8][step 96][point 6] 64-bit cycle counts).
extern "C" int mbn_debug_dwpw2_stamps(unsigned long long *host, size_t bytes)
{
    if (!host || bytes < sizeof(g_dwpw2_stamps)) return MBN_EINVAL;
    if (hipDeviceSynchronize() != hipSuccess) return MBN_EDEVICE;
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_dwpw2_stamps), sizeof(g_dwpw2_stamps)) == hipSuccess ? MBN_OK : MBN_EDEVICE;
}
