// mbn_f32_dw.hip — NHWC 3x3 depthwise conv + folded-BN scale/shift + ReLU/ReLU6 for gfx950 (fp32 or bf16 storage,
// fp32 arithmetic). Replaces the arithmetic of the reference's `depthwise` kernel (kernel.cl:62-92) in the mode the
// metric measures. HBM-bound (0.9-2.2 flop/B): the design goal is to read every input byte once and write every
// output byte once with wide, coalesced accesses, and to keep enough loads in flight.
//
// Decomposition ("column march"): a lane owns 4 consecutive channels (one float4 / four bf16) of TW adjacent output
// columns and walks DOWN the output rows of one segment, keeping the 3 x (TW*S+2) input window in registers as fp32.
// Each new output row needs only STRIDE new input rows, so an input element is requested (TW*S+2)/(TW*S) times
// per row instead of 9/S^2 times; the left/right halo requests are served by the L1 of the same workgroup because
// the lanes that own the neighbouring columns sit in the same wave or the next one.
// Lane layout inside a wave: CW lanes along channels (CW*16 contiguous bytes of one pixel), then columns. For wide
// layers (C >= 64) CW is capped at 16 so that a 256-lane workgroup covers >= 16*TW columns of one 64-channel slab:
// with CW = C/4 a workgroup of a 512-channel layer would hold 2 pixels and every halo request would miss the L1.
// The 9 filter taps and scale/shift of the lane's 4 channels stay in registers for the whole march.
// Measured and rejected (profiles/r01): non-temporal output stores (-5..-22 %), one column per lane (TW=1, -10 %),
// channel-fastest lanes across the full C (-15..-40 % on the 512/1024-channel layers).
#include "mbn_internal.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f4 ld4(const float *p) { return *reinterpret_cast<const f4 *>(p); }
__device__ __forceinline__ f4 ld4(const __bf16 *p)
{
    const bf4 v = *reinterpret_cast<const bf4 *>(p);
    return f4{ (float)v.x, (float)v.y, (float)v.z, (float)v.w };
}
__device__ __forceinline__ void st4(float *p, f4 v) { *reinterpret_cast<f4 *>(p) = v; }
__device__ __forceinline__ void st4(__bf16 *p, f4 v)
{
    *reinterpret_cast<bf4 *>(p) = bf4{ (__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w };   // RNE
}
__device__ __forceinline__ f4 fma4(f4 a, f4 b, f4 c)
{
    return f4{ fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w) };
}
__device__ __forceinline__ f4 act4(f4 v, int act)
{
    if (act == MBN_ACT_RELU6) {
        v.x = fminf(fmaxf(v.x, 0.f), 6.f); v.y = fminf(fmaxf(v.y, 0.f), 6.f);
        v.z = fminf(fmaxf(v.z, 0.f), 6.f); v.w = fminf(fmaxf(v.w, 0.f), 6.f);
    } else if (act == MBN_ACT_RELU) {
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    }
    return v;
}

struct DwArgs {
    void *out;
    const void *in;
    const float *filt, *scale, *shift;
    int batch, in_rows, in_cols, rows, cols, ch, pad_top, pad_left, act;
    int seg_rows, nseg;     // output rows per segment / segments per image
    int prio;               // wave priority of the whole kernel (3: a memory-bound kernel beside another stream's MFMA-streaming GEMM gets its few VALU slots)
    int cw;                 // lanes along channels inside a slab (channels per slab = 4*cw)
    int nslab;              // ch / (4*cw)
    int lcols;              // lane-columns per row = ceil(cols / TW)
    long total;             // lanes with work
};

// One input row for a lane: NC = TW*STRIDE+2 channel-quads at columns ix0 .. ix0+NC-1; zero outside the image.
template <int NC, typename T>
__device__ __forceinline__ void load_row(const DwArgs &a, const T *img, int iy, int ix0, int c, f4 (&r)[NC])
{
    const bool rowok = iy >= 0 && iy < a.in_rows;
    const T *row = img + ((long)iy * a.in_cols) * a.ch + c;
#pragma unroll
    for (int j = 0; j < NC; j++) {
        const int ix = ix0 + j;
        r[j] = (rowok && ix >= 0 && ix < a.in_cols) ? ld4(row + (long)ix * a.ch) : f4{ 0.f, 0.f, 0.f, 0.f };
    }
}

template <int STRIDE, int TW, typename T>
__global__ __launch_bounds__(256) void dw3x3_nhwc(DwArgs a)
{
    constexpr int NC = TW * STRIDE + 2;
    if (a.prio) __builtin_amdgcn_s_setprio(3);
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.total) return;
    // lane -> (channel-in-slab fastest, lane-column, slab, segment, image)
    const int cl = (int)(t % a.cw);
    long q = t / a.cw;
    const int lc = (int)(q % a.lcols);
    q /= a.lcols;
    const int slab = (int)(q % a.nslab);
    q /= a.nslab;
    const int seg = (int)(q % a.nseg);
    const int n = (int)(q / a.nseg);
    const int c = (slab * a.cw + cl) << 2;
    const int ox0 = lc * TW;

    f4 w[9];
#pragma unroll
    for (int k = 0; k < 9; k++) w[k] = ld4(a.filt + (long)k * a.ch + c);
    const f4 sc = a.scale ? ld4(a.scale + c) : f4{ 1.f, 1.f, 1.f, 1.f };
    const f4 sh = a.shift ? ld4(a.shift + c) : f4{ 0.f, 0.f, 0.f, 0.f };

    const T *img = reinterpret_cast<const T *>(a.in) + (long)n * a.in_rows * a.in_cols * a.ch;
    T *op = reinterpret_cast<T *>(a.out) + (((long)n * a.rows) * a.cols + ox0) * a.ch + c;
    const int oy0 = seg * a.seg_rows;
    const int oy1 = min(oy0 + a.seg_rows, a.rows);
    const int ix0 = ox0 * STRIDE - a.pad_left;

    f4 r0[NC], r1[NC], r2[NC];
    int iy = oy0 * STRIDE - a.pad_top;
    load_row<NC, T>(a, img, iy, ix0, c, r0);
    if (STRIDE == 1) load_row<NC, T>(a, img, iy + 1, ix0, c, r1);

    for (int oy = oy0; oy < oy1; oy++) {
        iy = oy * STRIDE - a.pad_top;
        if (STRIDE == 2) load_row<NC, T>(a, img, iy + 1, ix0, c, r1);
        load_row<NC, T>(a, img, iy + 2, ix0, c, r2);
#pragma unroll
        for (int p = 0; p < TW; p++) {
            const int j = p * STRIDE;
            f4 acc = f4{ 0.f, 0.f, 0.f, 0.f };
            acc = fma4(r0[j], w[0], acc); acc = fma4(r0[j + 1], w[1], acc); acc = fma4(r0[j + 2], w[2], acc);
            acc = fma4(r1[j], w[3], acc); acc = fma4(r1[j + 1], w[4], acc); acc = fma4(r1[j + 2], w[5], acc);
            acc = fma4(r2[j], w[6], acc); acc = fma4(r2[j + 1], w[7], acc); acc = fma4(r2[j + 2], w[8], acc);
            acc = act4(fma4(acc, sc, sh), a.act);
            if (TW == 1 || ox0 + p < a.cols) st4(op + ((long)oy * a.cols + p) * a.ch, acc);
        }
#pragma unroll
        for (int j = 0; j < NC; j++) {
            if (STRIDE == 1) { r0[j] = r1[j]; r1[j] = r2[j]; }
            else r0[j] = r2[j];
        }
    }
}

// ---- round 3 (LAB ONLY: measured 9-20 % SLOWER at stride 1, equal at stride 2): the same column march with BRANCH-FREE loads and
// a row of look-ahead. What it shows: the guarded loads of dw3x3_nhwc are not what holds the kernel at 0.80-0.84 of the copy
// control (profiles/r03/c_depthwise_copy_control.txt) — issuing the same requests back to back, or twice as many of them, is
// slower, and so was a third form (own columns only, halo columns from the neighbouring lanes by ds_bpermute; removed).
// dw3x3_nhwc above guards every load with (row inside && column inside) — in the ISA that is an exec-masked branch per load
// (65 branches in the stride-1 kernel) with the compiler's s_waitcnt vmcnt(0) behind each row, so a wave has ONE input row
// (4 x 16 B per lane) in flight and nothing else. Here the loads go through a buffer descriptor: a tap outside the image
// carries an out-of-range offset and the hardware returns 0, so a row is NC loads back to back, and (PF = 1) the row of
// output row oy+1 is requested before output row oy is computed: two rows in flight per wave at +16 VGPRs.
// Same taps, same fma order as dw3x3_nhwc: bit-identical. The offset of a tap is (row part) + (column part) with NO select per
// load — a conditional offset came out of the compiler as a branch per load with s_waitcnt vmcnt(0) inside (15-30 % slower than
// the guarded loads): an invalid row or column part is 2^30, the input tensor must be smaller than that (1 GiB), so any sum with
// an invalid part is out of range and none wraps.
constexpr unsigned DW_OOB = 0x40000000u;

template <int STRIDE, int TW, typename T, int PF>
__global__ __launch_bounds__(256) void dw3x3_nhwc_b(DwArgs a)
{
    constexpr int NC = TW * STRIDE + 2;
    constexpr unsigned ES = sizeof(T);
    if (a.prio) __builtin_amdgcn_s_setprio(3);
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.total) return;
    const int cl = (int)(t % a.cw);
    long q = t / a.cw;
    const int lc = (int)(q % a.lcols);
    q /= a.lcols;
    const int slab = (int)(q % a.nslab);
    q /= a.nslab;
    const int seg = (int)(q % a.nseg);
    const int n = (int)(q / a.nseg);
    const int c = (slab * a.cw + cl) << 2;
    const int ox0 = lc * TW;

    f4 w[9];
#pragma unroll
    for (int k = 0; k < 9; k++) w[k] = ld4(a.filt + (long)k * a.ch + c);
    const f4 sc = a.scale ? ld4(a.scale + c) : f4{ 1.f, 1.f, 1.f, 1.f };
    const f4 sh = a.shift ? ld4(a.shift + c) : f4{ 0.f, 0.f, 0.f, 0.f };

    const __amdgpu_buffer_rsrc_t irs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.in), 0,
                                                                        (unsigned)((long)a.batch * a.in_rows * a.in_cols * a.ch * ES), 0x00020000);
    T *op = reinterpret_cast<T *>(a.out) + (((long)n * a.rows) * a.cols + ox0) * a.ch + c;
    const int oy0 = seg * a.seg_rows;
    const int oy1 = min(oy0 + a.seg_rows, a.rows);
    const int ix0 = ox0 * STRIDE - a.pad_left;
    // byte offset of (n, row 0, column ix0 + j, channel c), or out of range for columns outside the image
    unsigned coff[NC];
#pragma unroll
    for (int j = 0; j < NC; j++) {
        const int ix = ix0 + j;
        coff[j] = (ix >= 0 && ix < a.in_cols) ? (unsigned)((((long)n * a.in_rows) * a.in_cols + ix) * a.ch + c) * ES : DW_OOB;
    }
    const unsigned rstride = (unsigned)a.in_cols * (unsigned)a.ch * ES;
    auto load_row_b = [&](int iy, f4 (&r)[NC]) __attribute__((always_inline)) {
        const unsigned ro = (unsigned)iy < (unsigned)a.in_rows ? (unsigned)iy * rstride : DW_OOB;
#pragma unroll
        for (int j = 0; j < NC; j++) {
            const unsigned off = ro + coff[j];
            if constexpr (sizeof(T) == 4) r[j] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(irs, off, 0, 0));
            else {
                typedef unsigned u2v __attribute__((ext_vector_type(2)));
                const u2v p = __builtin_bit_cast(u2v, __builtin_amdgcn_raw_buffer_load_b64(irs, off, 0, 0));
                r[j] = f4{ __builtin_bit_cast(float, p.x << 16), __builtin_bit_cast(float, p.x & 0xffff0000u),
                           __builtin_bit_cast(float, p.y << 16), __builtin_bit_cast(float, p.y & 0xffff0000u) };
            }
        }
    };

    f4 r0[NC], r1[NC], r2[NC], nx[NC], nx2[NC];
    int iy = oy0 * STRIDE - a.pad_top;
    load_row_b(iy, r0);
    if (STRIDE == 1) load_row_b(iy + 1, r1);
    if (PF) {                                        // the new row(s) of the first output row
        if (STRIDE == 2) load_row_b(iy + 1, nx2);
        load_row_b(iy + 2, nx);
    }
    for (int oy = oy0; oy < oy1; oy++) {
        iy = oy * STRIDE - a.pad_top;
        if (PF) {
#pragma unroll
            for (int j = 0; j < NC; j++) { r2[j] = nx[j]; if (STRIDE == 2) r1[j] = nx2[j]; }
            // the next output row's new input rows, requested before this row's arithmetic (past the segment: one dropped row)
            const int iyn = (oy + 1) * STRIDE - a.pad_top;
            const bool more = oy + 1 < oy1;
            if (STRIDE == 2) load_row_b(more ? iyn + 1 : -1, nx2);
            load_row_b(more ? iyn + 2 : -1, nx);
        } else {
            if (STRIDE == 2) load_row_b(iy + 1, r1);
            load_row_b(iy + 2, r2);
        }
#pragma unroll
        for (int p = 0; p < TW; p++) {
            const int j = p * STRIDE;
            f4 acc = f4{ 0.f, 0.f, 0.f, 0.f };
            acc = fma4(r0[j], w[0], acc); acc = fma4(r0[j + 1], w[1], acc); acc = fma4(r0[j + 2], w[2], acc);
            acc = fma4(r1[j], w[3], acc); acc = fma4(r1[j + 1], w[4], acc); acc = fma4(r1[j + 2], w[5], acc);
            acc = fma4(r2[j], w[6], acc); acc = fma4(r2[j + 1], w[7], acc); acc = fma4(r2[j + 2], w[8], acc);
            acc = act4(fma4(acc, sc, sh), a.act);
            if (TW == 1 || ox0 + p < a.cols) st4(op + ((long)oy * a.cols + p) * a.ch, acc);
        }
#pragma unroll
        for (int j = 0; j < NC; j++) {
            if (STRIDE == 1) { r0[j] = r1[j]; r1[j] = r2[j]; }
            else r0[j] = r2[j];
        }
    }
}

// bf16 storage, 8 channels (16 bytes) per lane: the fp32 kernel's decomposition with 4-channel lanes moves 8 bytes per
// lane-load in bf16 and is instruction-bound at ~2.2 TB/s (27 % of HBM); here a lane loads whole 16-byte vectors, widens
// them to fp32 once (a shift / a mask per element) and keeps the 3 x NC window in fp32 registers. Weights, scale, shift stay fp32 in registers; arithmetic and rounding are the 4-channel kernel's.
typedef unsigned u4v __attribute__((ext_vector_type(4)));
typedef float f8 __attribute__((ext_vector_type(8)));

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
__device__ __forceinline__ f8 ld8f(const float *p)
{
    const f4 a = *reinterpret_cast<const f4 *>(p), b = *reinterpret_cast<const f4 *>(p + 4);
    return f8{ a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w };
}

template <int NC>
__device__ __forceinline__ void load_row8(const DwArgs &a, const __bf16 *img, int iy, int ix0, int c, f8 (&rf)[NC])
{
    u4v r[NC];
    const bool rowok = iy >= 0 && iy < a.in_rows;
    const __bf16 *row = img + ((long)iy * a.in_cols) * a.ch + c;
#pragma unroll
    for (int j = 0; j < NC; j++) {
        const int ix = ix0 + j;
        r[j] = (rowok && ix >= 0 && ix < a.in_cols) ? *reinterpret_cast<const u4v *>(row + (long)ix * a.ch) : u4v{ 0u, 0u, 0u, 0u };
    }
#pragma unroll
    for (int j = 0; j < NC; j++) rf[j] = widen8(r[j]);          // each element is widened once, not once per tap
}

// the packed half of load_row8: issue the loads of one input row, widen later (row prefetch, stride 1)
template <int NC>
__device__ __forceinline__ void load_row8_packed(const DwArgs &a, const __bf16 *img, int iy, int ix0, int c, u4v (&r)[NC])
{
    const bool rowok = iy >= 0 && iy < a.in_rows;
    const __bf16 *row = img + ((long)iy * a.in_cols) * a.ch + c;
#pragma unroll
    for (int j = 0; j < NC; j++) {
        const int ix = ix0 + j;
        r[j] = (rowok && ix >= 0 && ix < a.in_cols) ? *reinterpret_cast<const u4v *>(row + (long)ix * a.ch) : u4v{ 0u, 0u, 0u, 0u };
    }
}

template <int STRIDE, int TW>
__global__ __launch_bounds__(256) void dw3x3_nhwc_bf16x8(DwArgs a)
{
    constexpr int NC = TW * STRIDE + 2;
    if (a.prio) __builtin_amdgcn_s_setprio(3);
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.total) return;
    const int cl = (int)(t % a.cw);                  // a.cw = lanes along channels inside a slab (8 channels each)
    long q = t / a.cw;
    const int lc = (int)(q % a.lcols);
    q /= a.lcols;
    const int slab = (int)(q % a.nslab);
    q /= a.nslab;
    const int seg = (int)(q % a.nseg);
    const int n = (int)(q / a.nseg);
    const int c = (slab * a.cw + cl) << 3;
    const int ox0 = lc * TW;

    f8 w[9];
#pragma unroll
    for (int k = 0; k < 9; k++) w[k] = ld8f(a.filt + (long)k * a.ch + c);
    const f8 one = f8{ 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f }, zero = one - one;
    const f8 sc = a.scale ? ld8f(a.scale + c) : one;
    const f8 sh = a.shift ? ld8f(a.shift + c) : zero;

    const __bf16 *img = reinterpret_cast<const __bf16 *>(a.in) + (long)n * a.in_rows * a.in_cols * a.ch;
    __bf16 *op = reinterpret_cast<__bf16 *>(a.out) + (((long)n * a.rows) * a.cols + ox0) * a.ch + c;
    const int oy0 = seg * a.seg_rows;
    const int oy1 = min(oy0 + a.seg_rows, a.rows);
    const int ix0 = ox0 * STRIDE - a.pad_left;

    f8 r0[NC], r1[NC], r2[NC];
    int iy = oy0 * STRIDE - a.pad_top;
    load_row8<NC>(a, img, iy, ix0, c, r0);
    if (STRIDE == 1) load_row8<NC>(a, img, iy + 1, ix0, c, r1);
    // Stride 1: the new input row of output row oy+1 is requested (packed: 16 VGPRs) BEFORE output row oy is computed, so a
    // lane has two rows of loads in flight instead of one. With the fp32 window, nine fp32 tap vectors and scale/shift this
    // kernel sits at ~190 VGPRs = 2 waves per SIMD, and with one row (4 KB per wave) in flight the stride-1 layers were
    // latency-bound at 3.7-4.3 TB/s (the stride-2 lanes request twice the bytes per step and reach 4.9-5.6).
    u4v pk[NC];
    if (STRIDE == 1) load_row8_packed<NC>(a, img, iy + 2, ix0, c, pk);

    for (int oy = oy0; oy < oy1; oy++) {
        iy = oy * STRIDE - a.pad_top;
        if (STRIDE == 2) {
            load_row8<NC>(a, img, iy + 1, ix0, c, r1);
            load_row8<NC>(a, img, iy + 2, ix0, c, r2);
        } else {
            u4v nx[NC];
            if (oy + 1 < oy1) load_row8_packed<NC>(a, img, iy + 3, ix0, c, nx);       // next output row's new input row
#pragma unroll
            for (int j = 0; j < NC; j++) r2[j] = widen8(pk[j]);
#pragma unroll
            for (int j = 0; j < NC; j++) pk[j] = nx[j];
        }
#pragma unroll
        for (int p = 0; p < TW; p++) {
            const int j = p * STRIDE;
            f8 acc = zero;                                     // same tap order as the 4-channel kernel
            acc = __builtin_elementwise_fma(r0[j], w[0], acc);
            acc = __builtin_elementwise_fma(r0[j + 1], w[1], acc);
            acc = __builtin_elementwise_fma(r0[j + 2], w[2], acc);
            acc = __builtin_elementwise_fma(r1[j], w[3], acc);
            acc = __builtin_elementwise_fma(r1[j + 1], w[4], acc);
            acc = __builtin_elementwise_fma(r1[j + 2], w[5], acc);
            acc = __builtin_elementwise_fma(r2[j], w[6], acc);
            acc = __builtin_elementwise_fma(r2[j + 1], w[7], acc);
            acc = __builtin_elementwise_fma(r2[j + 2], w[8], acc);
            acc = __builtin_elementwise_fma(acc, sc, sh);
            if (TW == 1 || ox0 + p < a.cols) {
                __bf16 *o = op + ((long)oy * a.cols + p) * a.ch;
                const f4 lo = act4(f4{ acc[0], acc[1], acc[2], acc[3] }, a.act), hi = act4(f4{ acc[4], acc[5], acc[6], acc[7] }, a.act);
                typedef __bf16 bf8v __attribute__((ext_vector_type(8)));
                *reinterpret_cast<bf8v *>(o) = bf8v{ (__bf16)lo.x, (__bf16)lo.y, (__bf16)lo.z, (__bf16)lo.w,
                                                    (__bf16)hi.x, (__bf16)hi.y, (__bf16)hi.z, (__bf16)hi.w };   // RNE
            }
        }
#pragma unroll
        for (int j = 0; j < NC; j++) {
            if (STRIDE == 1) { r0[j] = r1[j]; r1[j] = r2[j]; }
            else r0[j] = r2[j];
        }
    }
}

#ifdef MBN_LAB
// ---- round 3, LAB ONLY (measured EQUAL to dw3x3_nhwc_bf16x8 on the 14 x 14 layers, -4 % on 28 x 28, +30 % on 7 x 7: profiles/r03/y_bf16_depthwise_lds.txt — the bf16
// depthwise is bound by its widening / packing VALU work at two waves per SIMD, not by loads in flight): the LDS-staged form in the bf16 mode, stride 1, C % 64 == 0, for the NARROW maps (14 x 14, 28 x 28, 7 x 7: the stand-alone depthwise launches of
// the bf16 network). dw3x3_nhwc_bf16x8 above is latency-bound there: ~190 VGPRs = 2 waves per SIMD with two input rows (8 KB per wave) in flight, a lane walks
// its 14 rows one memory round trip after the other (42.8 us for the 103 + 103 MB of a 14 x 14 x 512 layer at batch 512 = 0.60 of 8 TB/s; 33.5 us is the
// read + write control). Here a workgroup owns G IMAGES side by side (G = 4 for 14-wide maps: image g occupies ring pixels g*SW .. g*SW+SW-1 with its zero
// halo columns, SW = W + 2 rounded up to even), one 64-channel slab (128 B per pixel) and a row segment; the input rows go through a ring of RING LDS rows
// (8 KB each: 64 pixels x 64 bf16) filled by buffer_load ... lds LAO rows ahead, so the loads in flight cost no VGPRs and do not depend on the occupancy;
// the arithmetic is dw3x3_nhwc_bf16x8's (a lane = 2 adjacent pixels x 8 channels marching down the rows, every element widened once, same fma order: same
// bits), its row loads replaced by four ds_read_b128.
struct DwLdsBfArgs {
    __bf16 *out;
    const __bf16 *in;
    const float *filt, *scale, *shift;
    int batch, h, w, ch, pad_top, pad_left, act;
    int nslab, nseg, seg_rows, G, SW;
    unsigned tensor_bytes;
};

template <int LAO>
__global__ __launch_bounds__(256) void dw3x3_lds_bf16(DwLdsBfArgs a)
{
    constexpr int RING = LAO + 3;
    constexpr int ROWF = 2048;                          // floats per ring row (8 KB): 64 pixels x 32 words (64 bf16)
    constexpr unsigned OOB = 0xF0000000u;
    __shared__ __attribute__((aligned(16))) float ring[RING * ROWF];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    int b = blockIdx.x;
    const int seg = b % a.nseg; b /= a.nseg;
    const int slab = b % a.nslab;
    const int n0 = (b / a.nslab) * a.G;
    const int c0 = slab * 64, q = tid & 7, pi = tid >> 3;                  // this lane: channels c0 + 8 q .. + 7 of ring pixels 2 pi, 2 pi + 1
    const int oy0 = seg * a.seg_rows, oy1 = min(oy0 + a.seg_rows, a.h);
    const __amdgpu_buffer_rsrc_t irsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16 *>(a.in), 0, a.tensor_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, a.tensor_bytes, 0x00020000);
    const unsigned img_bytes = (unsigned)(a.h * a.w * a.ch * 2), row_bytes = (unsigned)(a.w * a.ch * 2);

    // the two DMA pieces of this wave per input row: ring pixels (2 wave + k) * 8 + lane / 8 (8 lanes x 16 B per pixel)
    unsigned col_off[2];
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int sp = (wave_u * 2 + k) * 8 + (lane >> 3);
        const int g = sp / a.SW, ix = sp % a.SW - a.pad_left;
        const bool ok = g < a.G && n0 + g < a.batch && ix >= 0 && ix < a.w;
        col_off[k] = ok ? (unsigned)(n0 + g) * img_bytes + (unsigned)((ix * a.ch + c0 + (lane & 7) * 8) * 2) : OOB;
    }
    auto issue_row = [&](int iy) __attribute__((always_inline)) {
        const int slot = (iy + 2 * RING) % RING;                                    // iy >= -1
        const bool rok = iy >= 0 && iy < a.h;
        const int soff = rok ? iy * (int)row_bytes : 0;
#pragma unroll
        for (int k = 0; k < 2; k++)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(irsrc, (__attribute__((address_space(3))) void *)(ring + slot * ROWF + (wave_u * 2 + k) * 256),
                                                     16, rok ? col_off[k] : OOB, soff, 0, 0);
    };
    f8 w[9];
#pragma unroll
    for (int k = 0; k < 9; k++) w[k] = ld8f(a.filt + (long)k * a.ch + c0 + q * 8);
    const f8 one = f8{ 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f }, zero = one - one;
    const f8 sc = a.scale ? ld8f(a.scale + c0 + q * 8) : one;
    const f8 sh = a.shift ? ld8f(a.shift + c0 + q * 8) : zero;
    // outputs of this lane: centre pixels 2 pi + p (p = 0, 1); window columns 2 pi - 1 .. 2 pi + 2 (clamped: the clamped ones feed invalid outputs only)
    unsigned st_off[2];
#pragma unroll
    for (int p = 0; p < 2; p++) {
        const int sp = 2 * pi + p, g = sp / a.SW, ox = sp % a.SW - a.pad_left;
        const bool ok = g < a.G && n0 + g < a.batch && ox >= 0 && ox < a.w;
        st_off[p] = ok ? (unsigned)(n0 + g) * img_bytes + (unsigned)((ox * a.ch + c0 + q * 8) * 2) : OOB;
    }
    int rd[4];                                                                      // word offsets of the four window columns inside a ring row
#pragma unroll
    for (int j = 0; j < 4; j++) rd[j] = min(max(2 * pi - 1 + j, 0), 63) * 32 + q * 4;
    auto read_row = [&](int iy, f8 (&r)[4]) __attribute__((always_inline)) {
        const float *rp = ring + ((iy + 2 * RING) % RING) * ROWF;
#pragma unroll
        for (int j = 0; j < 4; j++) r[j] = widen8(*reinterpret_cast<const u4v *>(rp + rd[j]));
    };
    const int iy_first = oy0 - a.pad_top;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                // the tap / scale loads: keep the counted waits exact
    for (int r = 0; r < 3 + (LAO - 1); r++) issue_row(iy_first + r);
    f8 r0[4], r1[4], r2[4];
    for (int oy = oy0; oy < oy1; oy++) {
        const int iy = oy - a.pad_top;
        const int t = oy - oy0;
#define MBN_DWB_WAIT(N) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((N) > 63 ? 63 : (N)) : "memory")
        if (t >= LAO) MBN_DWB_WAIT(2 + (LAO - 1) * 4);
        else if (t == 0) MBN_DWB_WAIT((LAO - 1) * 2);
        else if (t == 1) MBN_DWB_WAIT((LAO - 1) * 2 + 2);
        else if (t == 2) MBN_DWB_WAIT((LAO - 1) * 2 + 4);
        else MBN_DWB_WAIT((LAO - 1) * 2 + 6);
#undef MBN_DWB_WAIT
        static_assert(LAO <= 4, "the chain above covers t < LAO <= 4");
        issue_row(iy + 2 + LAO);                                                    // the new row of output row oy + LAO
        if (t == 0) { read_row(iy, r0); read_row(iy + 1, r1); }
        read_row(iy + 2, r2);
        const unsigned orow = (unsigned)oy * row_bytes;
#pragma unroll
        for (int p = 0; p < 2; p++) {
            f8 acc = zero;                                                         // same tap order as dw3x3_nhwc_bf16x8
            acc = __builtin_elementwise_fma(r0[p], w[0], acc);
            acc = __builtin_elementwise_fma(r0[p + 1], w[1], acc);
            acc = __builtin_elementwise_fma(r0[p + 2], w[2], acc);
            acc = __builtin_elementwise_fma(r1[p], w[3], acc);
            acc = __builtin_elementwise_fma(r1[p + 1], w[4], acc);
            acc = __builtin_elementwise_fma(r1[p + 2], w[5], acc);
            acc = __builtin_elementwise_fma(r2[p], w[6], acc);
            acc = __builtin_elementwise_fma(r2[p + 1], w[7], acc);
            acc = __builtin_elementwise_fma(r2[p + 2], w[8], acc);
            acc = __builtin_elementwise_fma(acc, sc, sh);
            const f4 lo = act4(f4{ acc[0], acc[1], acc[2], acc[3] }, a.act), hi = act4(f4{ acc[4], acc[5], acc[6], acc[7] }, a.act);
            typedef __bf16 bf8v __attribute__((ext_vector_type(8)));
            const bf8v o = bf8v{ (__bf16)lo.x, (__bf16)lo.y, (__bf16)lo.z, (__bf16)lo.w, (__bf16)hi.x, (__bf16)hi.y, (__bf16)hi.z, (__bf16)hi.w };   // RNE
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4v, o), orsrc, st_off[p] == OOB ? OOB : st_off[p] + orow, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) { r0[j] = r1[j]; r1[j] = r2[j]; }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

static int launch_dw_lds_bf16(const mbn_call &c, const DwArgs &a, int channels, int nseg_force)
{
    DwLdsBfArgs l;
    l.out = (__bf16 *)a.out; l.in = (const __bf16 *)a.in; l.filt = a.filt; l.scale = a.scale; l.shift = a.shift;
    l.batch = a.batch; l.h = a.rows; l.w = a.cols; l.ch = a.ch; l.pad_top = a.pad_top; l.pad_left = a.pad_left; l.act = a.act;
    l.SW = (a.cols + 2 + 1) & ~1;
    if (l.SW > 64 || a.pad_left < 0 || a.pad_left > 1 || a.pad_top < 0 || a.pad_top > 1) return MBN_EUNSUPPORTED;
    l.G = 64 / l.SW;
    l.nslab = channels / 64;
    const double bytes = (double)a.batch * a.rows * a.cols * channels * 2;
    if (bytes >= 3.5e9) return MBN_EUNSUPPORTED;
    l.tensor_bytes = (unsigned)bytes;
    const long groups = (a.batch + l.G - 1) / l.G, base = groups * l.nslab;
    const long slots = 2L * c.ctx->num_cus;                          // ~200 VGPRs: two workgroups per CU (56 KB of LDS each: 7 ring rows, 4 rows ahead)
    int ns = 1;
    if (nseg_force > 0) ns = nseg_force;
    else if (base * 10 < slots * 6) {                                 // under-filled: whole rounds with the fewest segments
        ns = (int)((slots + base - 1) / base);
        if (ns > a.rows / 2) ns = a.rows / 2 > 0 ? a.rows / 2 : 1;
    }
    l.seg_rows = (a.rows + ns - 1) / ns;
    l.nseg = (a.rows + l.seg_rows - 1) / l.seg_rows;
    if ((double)base * l.nseg >= 2147483647.0) return MBN_EUNSUPPORTED;
    hipLaunchKernelGGL(dw3x3_lds_bf16<4>, dim3((unsigned)(base * l.nseg)), dim3(256), 0, c.stream, l);
    return MBN_OK;
}
#endif

// Generic fallback (any stride / filtersize / channel count): one lane per output element.
template <typename T>
__global__ __launch_bounds__(256) void dw_generic_nhwc(DwArgs a, int fs, int stride)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)a.batch * a.rows * a.cols * a.ch;
    if (t >= total) return;
    const int c = (int)(t % a.ch);
    long q = t / a.ch;
    const int ox = (int)(q % a.cols);
    q /= a.cols;
    const int oy = (int)(q % a.rows);
    const int n = (int)(q / a.rows);
    const T *img = reinterpret_cast<const T *>(a.in) + (long)n * a.in_rows * a.in_cols * a.ch;
    float acc = 0.f;
    for (int ky = 0; ky < fs; ky++) {
        int iy = oy * stride + ky - a.pad_top;
        if (iy < 0 || iy >= a.in_rows) continue;
        for (int kx = 0; kx < fs; kx++) {
            int ix = ox * stride + kx - a.pad_left;
            if (ix < 0 || ix >= a.in_cols) continue;
            acc = fmaf((float)img[((long)iy * a.in_cols + ix) * a.ch + c], a.filt[(long)(ky * fs + kx) * a.ch + c], acc);
        }
    }
    float v = fmaf(acc, a.scale ? a.scale[c] : 1.f, a.shift ? a.shift[c] : 0.f);
    if (a.act == MBN_ACT_RELU6) v = fminf(fmaxf(v, 0.f), 6.f);
    else if (a.act == MBN_ACT_RELU) v = fmaxf(v, 0.f);
    reinterpret_cast<T *>(a.out)[t] = (T)v;
}

// ---- round 3: the `north_star` form ("LDS-staged 3x3 input halos"), fp32, C % 32 == 0. SHIPPED for stride 1 on maps at least 50 pixels wide
// whose input + output do not fit the Infinity Cache (layers 2 and 6 at batch 256: -6 % / -5 % against the register column march below,
// profiles/r03/f_depthwise_variants.txt (e)); every other instantiation and the forcing knobs are lab-only (narrow maps, stride 2 and
// cache-resident sizes measured equal or slower). A workgroup owns (image, 32-channel slab,
// column strip, row segment) and marches down its output rows; the input rows of the strip (64 pixels x 128 B = 8 KB each) go through a
// RING of LDS rows filled by buffer_load ... lds, LAO output rows ahead of the row being computed (no VGPRs in flight; a pixel outside
// the image carries an out-of-range offset and the DMA writes zeros = the padding). A lane computes ONE output pixel x 4 channels per
// pass (two passes per row: 62 / 31 output pixels per strip) from nine ds_read_b128; stride 2 stores the strip's even input pixels in
// the first half of a ring row and the odd ones in the second, so adjacent lanes read adjacent LDS pixels at both strides (no bank
// conflicts). One raw s_barrier per output row with a counted vmcnt (the stores of a row are issued unconditionally — out-of-range when
// the lane has no pixel — so the count is a constant). Same fma order as dw3x3_nhwc: same bits.
struct DwLdsArgs {
    float *out;
    const float *in, *filt, *scale, *shift;
    int batch, in_rows, in_cols, rows, cols, ch, pad_top, pad_left, act;
    int nslab, nstrip, nseg, seg_rows, tw;      // tw = output pixels per strip
    unsigned in_img_bytes, out_img_bytes;
};

// S = stride, LAO = look-ahead in output rows, CH = channels per slab: a ring row is 8 KB = 64 pixels x 32 channels or 32 pixels x 64 channels
// (narrow maps: 28-wide rows fill a 32-pixel ring row, and 16 lanes = one pixel's 256 bytes, conflict-free as well)
template <int S, int LAO, int CH>
__global__ __launch_bounds__(256) void dw3x3_lds(DwLdsArgs a)
{
    constexpr int RING = S * LAO + 3;                   // ring rows of 8 KB
    constexpr int ROWF = 2048;                          // floats per ring row
    constexpr int PX = ROWF / CH;                       // pixels per ring row (64 / 32)
    constexpr int QL = CH / 4;                          // lanes per pixel (8 / 16)
    constexpr int PP = 256 / QL;                        // pixels per pass (32 / 16): two passes per row
    constexpr int PPC = 64 / QL;                        // pixels per DMA piece (8 / 4)
    constexpr unsigned OOB = 0xF0000000u;
    __shared__ __attribute__((aligned(16))) float ring[RING * ROWF];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    int b = blockIdx.x;
    const int strip = b % a.nstrip; b /= a.nstrip;
    const int seg = b % a.nseg; b /= a.nseg;
    const int slab = b % a.nslab;
    const int n = b / a.nslab;
    const int c0 = slab * CH, q = tid % QL;
    const int ox0 = strip * a.tw;
    const int oy0 = seg * a.seg_rows, oy1 = min(oy0 + a.seg_rows, a.rows);
    const int ix0 = ox0 * S - a.pad_left;               // input column of relative column 0
    const __amdgpu_buffer_rsrc_t irsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.in) + (size_t)n * a.in_rows * a.in_cols * a.ch, 0, a.in_img_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(a.out + (size_t)n * a.rows * a.cols * a.ch, 0, a.out_img_bytes, 0x00020000);

    // the two DMA pieces of this wave per input row: ring pixels (2 wave + k) * PPC + lane / QL; column part of the source offset, or OOB
    unsigned col_off[2];
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int sp = (wave_u * 2 + k) * PPC + lane / QL;                            // ring pixel slot 0..PX-1
        const int rel = S == 1 ? sp : (sp < PX / 2 ? 2 * sp : 2 * (sp - PX / 2) + 1); // relative input column held in that slot
        const int ix = ix0 + rel;
        const bool ok = ix >= 0 && ix < a.in_cols && rel < a.tw * S + 2;
        col_off[k] = ok ? (unsigned)((ix * a.ch + c0 + (lane % QL) * 4) * 4) : OOB;
    }
    const unsigned row_bytes = (unsigned)(a.in_cols * a.ch * 4);
    auto issue_row = [&](int iy) __attribute__((always_inline)) {                 // input row iy -> ring slot iy mod RING (iy may be outside: zeros)
        const int slot = (iy + 2 * RING) % RING;                                    // iy >= -1
        const bool rok = iy >= 0 && iy < a.in_rows;
        const int soff = rok ? iy * (int)row_bytes : 0;
#pragma unroll
        for (int k = 0; k < 2; k++)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(irsrc, (__attribute__((address_space(3))) void *)(ring + slot * ROWF + (wave_u * 2 + k) * 256),
                                                     16, rok ? col_off[k] : OOB, soff, 0, 0);
    };
    f4 w[9];
#pragma unroll
    for (int k = 0; k < 9; k++) w[k] = ld4(a.filt + (long)k * a.ch + c0 + q * 4);
    const f4 sc = a.scale ? ld4(a.scale + c0 + q * 4) : f4{ 1.f, 1.f, 1.f, 1.f };
    const f4 sh = a.shift ? ld4(a.shift + c0 + q * 4) : f4{ 0.f, 0.f, 0.f, 0.f };
    // this lane's output pixels (pass 0 / 1) and their LDS pixel slots of the three taps of a row
    int px[2];
    unsigned st_off[2];
#pragma unroll
    for (int ps = 0; ps < 2; ps++) {
        px[ps] = tid / QL + PP * ps;
        const bool ok = px[ps] < a.tw && ox0 + px[ps] < a.cols;
        st_off[ps] = ok ? (unsigned)(((ox0 + px[ps]) * a.ch + c0 + q * 4) * 4) : OOB;
        if (!ok) px[ps] = 0;                                                        // reads stay inside the ring row
    }
    auto tap_slot = [&](int p, int dx) __attribute__((always_inline)) {
        if (S == 1) return p + dx;
        return dx == 1 ? PX / 2 + p : p + (dx >> 1);                                // 2p (even), 2p+1 (odd half), 2p+2 (even)
    };
    const int iy_first = oy0 * S - a.pad_top;
    // prologue: the rows of output rows oy0 .. oy0 + LAO - 1 (first one: 3 rows, then S per row); every iteration then issues S rows
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                // the tap / scale loads above: keep the counted waits below exact
    for (int r = 0; r < 3 + S * (LAO - 1); r++) issue_row(iy_first + r);
    for (int oy = oy0; oy < oy1; oy++) {
        const int iy = oy * S - a.pad_top;
        // rows iy .. iy+2 have landed for every wave. Younger than them in the steady state (t >= LAO): 2 stores of iteration t - LAO and
        // (LAO - 1) x (S rows x 2 pieces + 2 stores); iterations t < LAO still wait for rows of the prologue: 2 S (LAO - 1 - t) younger
        // pieces of it + t x (2 S pieces + 2 stores) = 2 S (LAO - 1) + 2 t
        const int t = oy - oy0;
#define MBN_DWL_WAIT(N) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((N) > 63 ? 63 : (N)) : "memory")
        if (t >= LAO) MBN_DWL_WAIT(2 + (LAO - 1) * (2 * S + 2));
        else if (t == 0) MBN_DWL_WAIT((LAO - 1) * 2 * S);
        else if (t == 1) MBN_DWL_WAIT((LAO - 1) * 2 * S + 2);
        else if (t == 2) MBN_DWL_WAIT((LAO - 1) * 2 * S + 4);
        else if (t == 3) MBN_DWL_WAIT((LAO - 1) * 2 * S + 6);
        else if (t == 4) MBN_DWL_WAIT((LAO - 1) * 2 * S + 8);
        else MBN_DWL_WAIT((LAO - 1) * 2 * S + 10);
#undef MBN_DWL_WAIT
        static_assert(LAO <= 6, "the chain above covers t < LAO <= 6");
        // rows of output row oy + LAO (their ring slots held rows of output row oy - 1 and older: every wave is past them)
#pragma unroll
        for (int r = 0; r < S; r++) issue_row(iy + 2 + S * (LAO - 1) + 1 + r);
        const unsigned orow = (unsigned)(oy * a.cols * a.ch * 4);
#pragma unroll
        for (int ps = 0; ps < 2; ps++) {
            f4 acc = f4{ 0.f, 0.f, 0.f, 0.f };
#pragma unroll
            for (int dy = 0; dy < 3; dy++) {
                const float *rp = ring + ((iy + dy + 2 * RING) % RING) * ROWF + q * 4;
#pragma unroll
                for (int dx = 0; dx < 3; dx++) acc = fma4(*reinterpret_cast<const f4 *>(rp + tap_slot(px[ps], dx) * CH), w[dy * 3 + dx], acc);
            }
            acc = act4(fma4(acc, sc, sh), a.act);
            typedef unsigned u4e __attribute__((ext_vector_type(4)));
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4e, acc), orsrc, st_off[ps] == OOB ? OOB : st_off[ps] + orow, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

// launch of the LDS-staged form; `la` = look-ahead variant (0 = short ring, 1 = long ring), ch64 = 64-channel slabs (32-pixel ring rows)
static int launch_dw_lds(const mbn_call &c, const DwArgs &a, int rows, int cols, int stride, int channels, bool ch64, int la, int nseg_force)
{
    DwLdsArgs l;
    l.out = (float *)a.out; l.in = (const float *)a.in; l.filt = a.filt; l.scale = a.scale; l.shift = a.shift;
    l.batch = a.batch; l.in_rows = a.in_rows; l.in_cols = a.in_cols; l.rows = a.rows; l.cols = a.cols; l.ch = a.ch;
    l.pad_top = a.pad_top; l.pad_left = a.pad_left; l.act = a.act;
    const int px = ch64 ? 32 : 64;
    l.tw = stride == 1 ? px - 2 : (px - 1) / 2;
    l.nstrip = (cols + l.tw - 1) / l.tw;
    l.tw = (cols + l.nstrip - 1) / l.nstrip;                       // even strips (112 -> 2 x 56)
    l.nslab = channels / (ch64 ? 64 : 32);
    const int lao = stride == 1 ? (la ? 6 : 3) : (la ? 3 : 2);
    const int ring_kb = 8 * (stride * lao + 3);
    const long slots = (long)c.ctx->num_cus * (160 / ring_kb);      // resident workgroups
    const long base = (long)c.batch * l.nslab * l.nstrip;
    // row segments: a whole number of rounds of the resident workgroups (no tail), as few segments as that allows (each re-reads halo rows)
    int ns = 1;
    if (nseg_force > 0) ns = nseg_force;
    else if (base > slots || base * 10 < slots * 6) {
        double best = 1e30;
        for (int k = 1; k <= rows / 8 && k <= 16; k++) {
            const int sr = (rows + k - 1) / k, kk = (rows + sr - 1) / sr;
            const double w = (double)base * kk / slots;
            const double rounds = w <= 1.0 ? (w >= 0.6 ? w : 0.6) : ceil(w);      // under-filled is fine down to 60 % of the slots
            const double cost = rounds / w * (1.0 + 0.5 / sr);
            if (cost < best - 1e-9) { best = cost; ns = k; }
        }
    }
    if (ns > rows) ns = rows;
    l.seg_rows = (rows + ns - 1) / ns;
    l.nseg = (rows + l.seg_rows - 1) / l.seg_rows;
    l.in_img_bytes = (unsigned)((size_t)a.in_rows * a.in_cols * channels * 4);      // one image: < 2^31 bytes (callers check), so the kernel's
    l.out_img_bytes = (unsigned)((size_t)rows * cols * channels * 4);              // 32-bit row offsets cannot wrap
    if ((double)base * l.nseg >= 2147483647.0) return MBN_EUNSUPPORTED;
    const dim3 g((unsigned)(base * l.nseg));
#define MBN_DWL(S_, L_, C_) hipLaunchKernelGGL((dw3x3_lds<S_, L_, C_>), g, dim3(256), 0, c.stream, l)
#ifdef MBN_LAB
    if (stride == 1) {
        if (ch64) { if (la) MBN_DWL(1, 6, 64); else MBN_DWL(1, 3, 64); }
        else { if (la) MBN_DWL(1, 6, 32); else MBN_DWL(1, 3, 32); }
    } else {
        if (ch64) { if (la) MBN_DWL(2, 3, 64); else MBN_DWL(2, 2, 64); }
        else { if (la) MBN_DWL(2, 3, 32); else MBN_DWL(2, 2, 32); }
    }
#else
    if (stride != 1 || ch64 || la) return MBN_EUNSUPPORTED;       // the shipped library holds the one instantiation its rule can reach
    MBN_DWL(1, 3, 32);
#endif
#undef MBN_DWL
    return MBN_OK;
}

template <typename T>
int launch_dw(const mbn_call &c, DwArgs &a, int rows, int cols, int fs, int stride, int channels)
{
    const size_t io_align = sizeof(T) * 4;             // one channel-quad
    const bool fast = fs == 3 && (stride == 1 || stride == 2) && (channels % 4) == 0 &&
                      ((uintptr_t)a.in % io_align) == 0 && ((uintptr_t)a.out % io_align) == 0 &&
                      ((uintptr_t)a.filt % 16) == 0 && (!c.scale || ((uintptr_t)c.scale % 16) == 0) &&
                      (!c.shift || ((uintptr_t)c.shift % 16) == 0);
    if (!fast) {
        a.seg_rows = rows; a.nseg = 1; a.total = 0; a.cw = a.nslab = a.lcols = 1;
        long total = (long)c.batch * rows * cols * channels;
        hipLaunchKernelGGL(dw_generic_nhwc<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c.stream, a, fs, stride);
        return MBN_OK;
    }
    const bool cache_resident = (double)c.batch * ((double)a.in_rows * a.in_cols + (double)rows * cols) * channels * sizeof(T) < 64.0 * 1048576;
    // variant: bits 0..1 = TW (0 = default 2), bit 4 = lanes across the full C instead of 64-channel slabs,
    // bit 5 = bf16 with 4-channel lanes (the first bf16 version; A/B hook)
    const int var = g_mbn_tune.dw_variant;
    if (sizeof(T) == 2 && !(var & 32) && (channels % 8) == 0 && ((uintptr_t)a.in % 16) == 0 && ((uintptr_t)a.out % 16) == 0) {
        const int c8 = channels / 8;
        int cw = c8 > 8 ? 8 : c8;                    // slab = 8 lanes x 8 channels = 64 channels = 128 B per pixel
        while (c8 % cw) cw--;
        a.cw = cw;
        a.nslab = c8 / cw;
        const int tw8 = (var & 3) == 1 ? 1 : 2;      // 2 columns per lane measured faster on every layer (tune dw_variant=1: one)
        a.lcols = (cols + tw8 - 1) / tw8;
        const long row_lanes = (long)c.batch * a.lcols * c8;
        const long target = (long)c.ctx->num_cus * 64 * (stride == 1 ? 12 : 6);
        int nseg = 1;
        if (g_mbn_tune.dw_nseg > 0) nseg = g_mbn_tune.dw_nseg;
        else if (row_lanes < target) {
            nseg = (int)((target + row_lanes - 1) / row_lanes);
            int max_seg = rows / 4 > 0 ? rows / 4 : 1;
            if (row_lanes * max_seg < (long)c.ctx->num_cus * 64 || cache_resident) max_seg = rows;   // latency-bound (see the fp32 branch)
            if (nseg > max_seg) nseg = max_seg;
        }
        if (nseg > rows) nseg = rows;
        a.seg_rows = (rows + nseg - 1) / nseg;
        a.nseg = (rows + a.seg_rows - 1) / a.seg_rows;
        a.total = row_lanes * a.nseg;
        dim3 grid((unsigned)((a.total + 255) / 256));
#ifdef MBN_LAB
        // LAB: the LDS-staged bf16 form (exp0 = 8): stride 1, C % 64 == 0, maps up to 62 pixels wide (same bits as the kernel below)
        if (stride == 1 && (channels % 64) == 0 && a.in_rows == rows && a.in_cols == cols && g_mbn_tune.exp0 == 8) {
            const int forced_seg = g_mbn_tune.dw_nseg;
            const int rc = launch_dw_lds_bf16(c, a, channels, forced_seg);
            if (rc != MBN_EUNSUPPORTED) return rc;
        }
#endif
#ifdef MBN_LAB
        if (tw8 == 1) {
            if (stride == 1) hipLaunchKernelGGL((dw3x3_nhwc_bf16x8<1, 1>), grid, dim3(256), 0, c.stream, a);
            else hipLaunchKernelGGL((dw3x3_nhwc_bf16x8<2, 1>), grid, dim3(256), 0, c.stream, a);
            return MBN_OK;
        }
#endif
        if (stride == 1) hipLaunchKernelGGL((dw3x3_nhwc_bf16x8<1, 2>), grid, dim3(256), 0, c.stream, a);
        else hipLaunchKernelGGL((dw3x3_nhwc_bf16x8<2, 2>), grid, dim3(256), 0, c.stream, a);
        return MBN_OK;
    }
    int tw = (var & 3) ? (var & 3) : 2;
    if (tw > 2) tw = 2;
    const int c4 = channels / 4;
    int cw = c4;
    if (!(var & 16)) {                               // slab of <= 16 lanes along channels (must divide C/4)
        cw = c4 > 16 ? 16 : c4;
        while (c4 % cw) cw--;
    }
    a.cw = cw;
    a.nslab = c4 / cw;
    a.lcols = (cols + tw - 1) / tw;
    // Row segments: every extra segment re-reads 2 (stride 1) or 1 (stride 2) halo rows from HBM — measured as
    // 19-36 % over-fetch (FETCH_SIZE, profiles/r01) when segmenting for "two full rounds" of lanes — so segment
    // only when a full-height march leaves the chip under-filled: ~12 waves/CU for stride 1, ~6 for stride 2
    // (whose lanes keep 10 loads in flight per step). tools/layer_bench.py --tune dw_nseg=... is the sweep.
    const long row_lanes = (long)c.batch * a.lcols * c4;
    const long target = (long)c.ctx->num_cus * 64 * (stride == 1 ? 12 : 6);
    int nseg = 1;
    if (g_mbn_tune.dw_nseg > 0) nseg = g_mbn_tune.dw_nseg;
    else if (row_lanes < target) {
        nseg = (int)((target + row_lanes - 1) / row_lanes);
        int max_seg = rows / 4 > 0 ? rows / 4 : 1;     // keep >= 4 output rows per segment ...
        // ... unless even that leaves less than one wave per CU (a few images): then the launch is bound by the length of a
        // lane's row march (one dependent memory round trip per row: 9-10 us for a 14-row map at batch 1), not by bytes, and
        // one output row per segment is best
        // ... or the tensors sit in L2 / Infinity Cache anyway (input + output under 64 MB: 5 ... 64 images on the 14 x 14 and 7 x 7 maps): the
        // halo rows an extra segment re-reads come from cache, and shorter marches are what the launch is short of — measured 1-5 us per
        // launch, 2.5-4.5 % of a forward at 8 ... 32 images (profiles/r03/w_depthwise_segments_small_batch.txt)
        if (row_lanes * max_seg < (long)c.ctx->num_cus * 64 || cache_resident) max_seg = rows;
        if (nseg > max_seg) nseg = max_seg;
    }
    // Round 4 (profiles/r04/i_depthwise_stride2_segments.txt): stride 2 on an input of streaming size (>= 512 MB: layer 4 from batch 256 up, 822 MB) runs 5-10 % faster
    // with TWO output rows per segment although a full-height march already fills the chip — 0.1988 -> 0.1861 ms at batch 256, 0.4194 -> 0.3772 at 512,
    // 0.2148 -> 0.1923 at 320 x 320 / 128 images — and 1-20 % slower below that size (411 MB at batch 128: +1 %; cache-sized inputs: +10-20 %): a lane's
    // 56-row march is one dependent HBM round trip per row, and the row a segment re-reads was fetched by its neighbour microseconds earlier.
    // The rule covers the shape it was measured on: output maps of >= 40 rows (the 112 -> 56 layer 4, the 320 -> 160 case). On the 56 -> 28 layer 8
    // the same segmentation measured flat to +6 % (0.0916 -> 0.0971 ms at 411 MB; 822 MB at batch 512 was never measured): not taken there.
    if (g_mbn_tune.dw_nseg <= 0 && sizeof(T) == 4 && stride == 2 && rows >= 40 &&
        (double)c.batch * a.in_rows * a.in_cols * channels * sizeof(T) >= 512.0 * 1048576)
        nseg = rows / 2;
    if (nseg > rows) nseg = rows;
    a.seg_rows = (rows + nseg - 1) / nseg;
    a.nseg = (rows + a.seg_rows - 1) / a.seg_rows;
    a.total = row_lanes * a.nseg;
    dim3 grid((unsigned)((a.total + 255) / 256));
#ifdef MBN_LAB
    if (tw == 1) {
        if (stride == 1) hipLaunchKernelGGL((dw3x3_nhwc<1, 1, T>), grid, dim3(256), 0, c.stream, a);
        else hipLaunchKernelGGL((dw3x3_nhwc<2, 1, T>), grid, dim3(256), 0, c.stream, a);
        return MBN_OK;
    }
#endif
    // the LDS-staged form where it measured faster: fp32, stride 1, maps >= 50 pixels wide, tensors beyond the Infinity Cache
    // (lab: exp0 = 1 never)
    if (sizeof(T) == 4 && stride == 1 && cols >= 50 && (channels % 32) == 0 && g_mbn_tune.exp0 == 0 &&
        (double)c.batch * (a.in_rows * a.in_cols + rows * cols) * channels * 4 >= 512.0 * 1048576 &&
        (double)a.in_rows * a.in_cols * channels * 4 < 2.0e9 && (double)rows * cols * channels * 4 < 2.0e9)
    {
        const int rc = launch_dw_lds(c, a, rows, cols, stride, channels, false, 0, g_mbn_tune.dw_nseg);
        if (rc != MBN_EUNSUPPORTED) return rc;                 // (grid too large: the column march below)
    }
#ifdef MBN_LAB
    // LAB: the LDS-staged form forced: exp0 = 6 (32-channel slabs) / 7 (64-channel slabs), + 10 = long ring
    {
        const int e = g_mbn_tune.exp0;
        const int kind = e % 10, la = e >= 10 && e < 20 ? 1 : 0;
        if ((kind == 6 || kind == 7) && e < 20 && sizeof(T) == 4 && (channels % (kind == 7 ? 64 : 32)) == 0 &&
            (double)a.in_rows * a.in_cols * channels * 4 < 2.0e9 && (double)rows * cols * channels * 4 < 2.0e9)
        {
            const int rc = launch_dw_lds(c, a, rows, cols, stride, channels, kind == 7, la, g_mbn_tune.dw_nseg);
            if (rc != MBN_EUNSUPPORTED) return rc;
        }
    }
    // LAB ONLY (slower, profiles/r03/f_depthwise_variants.txt): branch-free buffer loads, exp0 = 2 without / 3 with one row of look-ahead
    const bool small_in = (double)c.batch * a.in_rows * a.in_cols * channels * sizeof(T) < 1073741824.0;
    const int dv = g_mbn_tune.exp0;
    if (small_in && (dv == 2 || dv == 3)) {
        if (dv == 2) {
            if (stride == 1) hipLaunchKernelGGL((dw3x3_nhwc_b<1, 2, T, 0>), grid, dim3(256), 0, c.stream, a);
            else hipLaunchKernelGGL((dw3x3_nhwc_b<2, 2, T, 0>), grid, dim3(256), 0, c.stream, a);
        } else {
            if (stride == 1) hipLaunchKernelGGL((dw3x3_nhwc_b<1, 2, T, 1>), grid, dim3(256), 0, c.stream, a);
            else hipLaunchKernelGGL((dw3x3_nhwc_b<2, 2, T, 1>), grid, dim3(256), 0, c.stream, a);
        }
        return MBN_OK;
    }
#endif
    if (stride == 1) hipLaunchKernelGGL((dw3x3_nhwc<1, 2, T>), grid, dim3(256), 0, c.stream, a);
    else hipLaunchKernelGGL((dw3x3_nhwc<2, 2, T>), grid, dim3(256), 0, c.stream, a);
    return MBN_OK;
}

}   // namespace

// `out`/`in` are fp32 or bf16 NHWC according to c.dtype; the filter and scale/shift are always fp32.
int mbn_launch_f32_depthwise(const mbn_call &c, void *out, const void *in, const float *filt, int rows, int cols,
                             int fs, int stride, int channels)
{
    DwArgs a;
    a.out = out; a.in = in; a.filt = filt; a.scale = c.scale; a.shift = c.shift;
    a.batch = c.batch; a.in_rows = c.in_rows; a.in_cols = c.in_cols; a.rows = rows; a.cols = cols; a.ch = channels;
    a.pad_top = c.pad_top >= 0 ? c.pad_top : mbn_same_pad(c.in_rows, rows, fs, stride);
    a.pad_left = c.pad_left >= 0 ? c.pad_left : mbn_same_pad(c.in_cols, cols, fs, stride);
    a.act = c.act;
    a.prio = (g_mbn_tune.dw_variant & 128) ? 0 : 1;       // dw_variant bit 7: A/B hook, no raised priority
    if (c.dtype == MBN_DT_BF16) return launch_dw<__bf16>(c, a, rows, cols, fs, stride, channels);
    return launch_dw<float>(c, a, rows, cols, fs, stride, channels);
}
