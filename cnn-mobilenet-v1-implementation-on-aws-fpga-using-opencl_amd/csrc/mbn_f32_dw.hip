// mbn_f32_dw.hip — fp32 NHWC 3x3 depthwise conv + folded-BN scale/shift + ReLU/ReLU6 for gfx950.
// Replaces the arithmetic of the reference's `depthwise` kernel (kernel.cl:62-92) in the fp32 mode the metric
// measures. HBM-bound (0.9-2.2 flop/B): the design goal is to read every input byte once and write every
// output byte once with 16-B-per-lane, fully coalesced accesses, and to keep enough loads in flight.
//
// Decomposition ("column march"): a lane owns 4 consecutive channels (one float4) of one output column `ox`
// and walks DOWN the image rows of one segment, keeping the 3x3 input window of float4s in registers. Each
// new output row needs only STRIDE new input rows (3 float4 loads each), so an input element is requested
// 3x (stride 1) / 1.5x (stride 2) in total; the two neighbour-column requests hit the L1/TA path of the same
// workgroup (the lanes owning ox-1 / ox+1 fetch the same lines), so HBM sees each line once.
// Lanes are laid out channel-fastest then column, so a wave's load of one input row is one contiguous
// 1-KiB span of the NHWC row (C floats per pixel x consecutive pixels) — the coalescing NHWC was chosen for.
// The 9 filter taps and the scale/shift of the lane's 4 channels live in registers for the whole march.
#include "mbn_internal.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f4 ld4(const float *p) { return *reinterpret_cast<const f4 *>(p); }
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
    float *out;
    const float *in, *filt, *scale, *shift;
    int batch, in_rows, in_cols, rows, cols, ch, pad_top, pad_left, act;
    int seg_rows, nseg;     // output rows per segment / segments per image
    long total;             // lanes with work: batch * nseg * cols * (ch/4)
};

// Load the 3 float4 of input row `iy` around this lane's columns (ix0, ix0+1, ix0+2); zero outside the image.
__device__ __forceinline__ void load_row(const DwArgs &a, const float *img, int iy, int ix0, int c, f4 &l, f4 &m, f4 &r)
{
    const f4 z = f4{ 0.f, 0.f, 0.f, 0.f };
    l = m = r = z;
    if (iy < 0 || iy >= a.in_rows) return;
    const float *row = img + ((long)iy * a.in_cols) * a.ch + c;
    if (ix0 >= 0 && ix0 < a.in_cols) l = ld4(row + (long)ix0 * a.ch);
    if (ix0 + 1 >= 0 && ix0 + 1 < a.in_cols) m = ld4(row + (long)(ix0 + 1) * a.ch);
    if (ix0 + 2 >= 0 && ix0 + 2 < a.in_cols) r = ld4(row + (long)(ix0 + 2) * a.ch);
}

template <int STRIDE>
__global__ __launch_bounds__(256) void dw3x3_f32_nhwc(DwArgs a)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.total) return;
    const int c4n = a.ch >> 2;
    const int c = (int)(t % c4n) << 2;
    long q = t / c4n;
    const int ox = (int)(q % a.cols);
    q /= a.cols;
    const int seg = (int)(q % a.nseg);
    const int n = (int)(q / a.nseg);

    f4 w[9];
#pragma unroll
    for (int k = 0; k < 9; k++) w[k] = ld4(a.filt + (long)k * a.ch + c);
    const f4 sc = a.scale ? ld4(a.scale + c) : f4{ 1.f, 1.f, 1.f, 1.f };
    const f4 sh = a.shift ? ld4(a.shift + c) : f4{ 0.f, 0.f, 0.f, 0.f };

    const float *img = a.in + (long)n * a.in_rows * a.in_cols * a.ch;
    float *op = a.out + (((long)n * a.rows) * a.cols + ox) * a.ch + c;
    const int oy0 = seg * a.seg_rows;
    const int oy1 = min(oy0 + a.seg_rows, a.rows);
    const int ix0 = ox * STRIDE - a.pad_left;

    // window rows: r0 = input row oy*S - pad, r1 = +1, r2 = +2
    f4 r0l, r0m, r0r, r1l, r1m, r1r, r2l, r2m, r2r;
    int iy = oy0 * STRIDE - a.pad_top;
    load_row(a, img, iy, ix0, c, r0l, r0m, r0r);
    if (STRIDE == 1) load_row(a, img, iy + 1, ix0, c, r1l, r1m, r1r);

    for (int oy = oy0; oy < oy1; oy++) {
        iy = oy * STRIDE - a.pad_top;
        if (STRIDE == 1) {
            load_row(a, img, iy + 2, ix0, c, r2l, r2m, r2r);
        } else {
            load_row(a, img, iy + 1, ix0, c, r1l, r1m, r1r);
            load_row(a, img, iy + 2, ix0, c, r2l, r2m, r2r);
        }
        f4 acc = f4{ 0.f, 0.f, 0.f, 0.f };
        acc = fma4(r0l, w[0], acc); acc = fma4(r0m, w[1], acc); acc = fma4(r0r, w[2], acc);
        acc = fma4(r1l, w[3], acc); acc = fma4(r1m, w[4], acc); acc = fma4(r1r, w[5], acc);
        acc = fma4(r2l, w[6], acc); acc = fma4(r2m, w[7], acc); acc = fma4(r2r, w[8], acc);
        acc = act4(fma4(acc, sc, sh), a.act);
        *reinterpret_cast<f4 *>(op + (long)oy * a.cols * a.ch) = acc;
        if (STRIDE == 1) {
            r0l = r1l; r0m = r1m; r0r = r1r;
            r1l = r2l; r1m = r2m; r1r = r2r;
        } else {
            r0l = r2l; r0m = r2m; r0r = r2r;
        }
    }
}

// Generic fallback (any stride / filtersize / channel count): one lane per output element.
__global__ __launch_bounds__(256) void dw_generic_f32_nhwc(DwArgs a, int fs, int stride)
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
    const float *img = a.in + (long)n * a.in_rows * a.in_cols * a.ch;
    float acc = 0.f;
    for (int ky = 0; ky < fs; ky++) {
        int iy = oy * stride + ky - a.pad_top;
        if (iy < 0 || iy >= a.in_rows) continue;
        for (int kx = 0; kx < fs; kx++) {
            int ix = ox * stride + kx - a.pad_left;
            if (ix < 0 || ix >= a.in_cols) continue;
            acc = fmaf(img[((long)iy * a.in_cols + ix) * a.ch + c], a.filt[(long)(ky * fs + kx) * a.ch + c], acc);
        }
    }
    float v = fmaf(acc, a.scale ? a.scale[c] : 1.f, a.shift ? a.shift[c] : 0.f);
    if (a.act == MBN_ACT_RELU6) v = fminf(fmaxf(v, 0.f), 6.f);
    else if (a.act == MBN_ACT_RELU) v = fmaxf(v, 0.f);
    a.out[t] = v;
}

}   // namespace

int mbn_launch_f32_depthwise(const mbn_call &c, float *out, const float *in, const float *filt, int rows, int cols,
                             int fs, int stride, int channels)
{
    DwArgs a;
    a.out = out; a.in = in; a.filt = filt; a.scale = c.scale; a.shift = c.shift;
    a.batch = c.batch; a.in_rows = c.in_rows; a.in_cols = c.in_cols; a.rows = rows; a.cols = cols; a.ch = channels;
    a.pad_top = c.pad_top >= 0 ? c.pad_top : mbn_same_pad(c.in_rows, rows, fs, stride);
    a.pad_left = c.pad_left >= 0 ? c.pad_left : mbn_same_pad(c.in_cols, cols, fs, stride);
    a.act = c.act;
    const bool fast = fs == 3 && (stride == 1 || stride == 2) && (channels % 4) == 0 &&
                      ((uintptr_t)in % 16) == 0 && ((uintptr_t)out % 16) == 0 && ((uintptr_t)filt % 16) == 0 &&
                      (!c.scale || ((uintptr_t)c.scale % 16) == 0) && (!c.shift || ((uintptr_t)c.shift % 16) == 0);
    if (!fast) {
        a.seg_rows = rows; a.nseg = 1; a.total = 0;
        long total = (long)c.batch * rows * cols * channels;
        hipLaunchKernelGGL(dw_generic_f32_nhwc, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c.stream, a, fs,
                           stride);
        return MBN_OK;
    }
    // Segment the rows only when a full-height march would leave the chip under-filled: each extra segment
    // re-reads 2 halo rows (stride 1). Target >= ~2 resident rounds of 256 CUs x 2048 lanes.
    const long cols_lanes = (long)c.batch * cols * (channels / 4);
    const long target = (long)c.ctx->num_cus * 2048 * 2;
    int nseg = 1;
    if (cols_lanes < target) {
        nseg = (int)((target + cols_lanes - 1) / cols_lanes);
        int max_seg = rows / 4 > 0 ? rows / 4 : 1;     // keep >= 4 output rows per segment
        if (nseg > max_seg) nseg = max_seg;
    }
    a.seg_rows = (rows + nseg - 1) / nseg;
    a.nseg = (rows + a.seg_rows - 1) / a.seg_rows;
    a.total = cols_lanes * a.nseg;
    dim3 grid((unsigned)((a.total + 255) / 256));
    if (stride == 1) hipLaunchKernelGGL(dw3x3_f32_nhwc<1>, grid, dim3(256), 0, c.stream, a);
    else hipLaunchKernelGGL(dw3x3_f32_nhwc<2>, grid, dim3(256), 0, c.stream, a);
    return MBN_OK;
}
