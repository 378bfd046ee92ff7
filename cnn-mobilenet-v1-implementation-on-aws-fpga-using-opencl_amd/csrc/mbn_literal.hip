// mbn_literal.hip — LITERAL mode on gfx950: the integer semantics of the reference's four OpenCL kernels
// (kernel.cl:2-132): uint8 planar NCHW activations, int32 filters [oc][ic][ky][kx], int32 accumulate,
// ReLU, truncating int->uchar store. Bit-exact against oracle/ (tests/test_parity_gpu.py).
//
// Work decomposition follows what the arithmetic allows, not the reference's launch: one lane per emulated
// work-item (tx,ty) with the output-channel loop inside the lane, because MBN_Q_CARRY_SUM (kernel.cl:10,69,99:
// `sum` survives from one output channel to the next) makes the channels of one pixel a serial chain.
// Lanes run along tx, so every uint8 plane read and every output store is contiguous across the wave.
// Filters are wave-uniform and come through the scalar cache. This path is HBM/latency-light integer work;
// the fp32 NHWC kernels (mbn_f32_*.hip) are the performance path.
#include "mbn_internal.h"

namespace {

__device__ __forceinline__ int mac_i32(int sum, unsigned a, int w) { return (int)((unsigned)sum + a * (unsigned)w); }

// One tap: kernel.cl:19-25. `lit` selects the reference's own flat index expression.
__device__ __forceinline__ unsigned lit_tap(const uint8_t *__restrict__ plane, long limit, bool lit, int ty, int tx,
                                            int i, int j, int stride, int g0, int in_rows, int in_cols)
{
    if (lit) {
        int yi = ty + i, xi = tx + j;
        if (yi < 0 || xi < 0) return 0u;
        long idx = (long)yi * g0 * stride + (long)xi * stride;
        return idx < limit ? plane[idx] : 0u;
    }
    int iy = ty * stride + i, ix = tx * stride + j;
    if (iy < 0 || ix < 0 || iy >= in_rows || ix >= in_cols) return 0u;
    return plane[(long)iy * in_cols + ix];
}

__global__ __launch_bounds__(256) void lit_convolute_k(uint8_t *__restrict__ out, const uint8_t *__restrict__ r,
                                                       const uint8_t *__restrict__ g, const uint8_t *__restrict__ b,
                                                       const int *__restrict__ filt, int rows, int cols, int fs,
                                                       int stride, int op_size, unsigned quirks, int g0, int g1,
                                                       long out_plane, long in_image, long out_image)
{
    const int half = fs / 2;
    const bool lit = quirks & MBN_Q_LITERAL_INDEX, carry = quirks & MBN_Q_CARRY_SUM;
    const int n = blockIdx.z;
    const long wi = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (wi >= (long)g0 * g1) return;
    const int ty = (int)(wi / g0), tx = (int)(wi % g0);
    const uint8_t *planes[3] = { r + n * in_image, g + n * in_image, b + n * in_image };
    const long limit = (long)rows * cols;
    uint8_t *o = out + n * out_image + ((long)ty * g0 + tx);
    int sum = 0, findex = 0;
    for (int oc = 0; oc < op_size; oc++) {
        if (!carry) sum = 0;
        for (int p = 0; p < 3; p++)
            for (int i = -half; i <= half; i++)
                for (int j = -half; j <= half; j++, findex++)
                    sum = mac_i32(sum, lit_tap(planes[p], limit, lit, ty, tx, i, j, stride, g0, rows, cols),
                                  filt[findex]);
        if (sum <= 0) sum = 0;
        o[out_plane * oc] = (uint8_t)sum;
    }
}

__global__ __launch_bounds__(256) void lit_depthwise_k(uint8_t *__restrict__ out, const uint8_t *__restrict__ in,
                                                       const int *__restrict__ filt, int rows, int cols, int fs,
                                                       int stride, int op_size, unsigned quirks, int g0, int g1,
                                                       int in_rows, int in_cols)
{
    const int half = fs / 2;
    const bool lit = quirks & MBN_Q_LITERAL_INDEX, carry = quirks & MBN_Q_CARRY_SUM, plane0 = quirks & MBN_Q_DW_PLANE0;
    const int n = blockIdx.z;
    const long wi = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (wi >= (long)g0 * g1) return;
    const int ty = (int)(wi / g0), tx = (int)(wi % g0);
    const long in_plane = (long)in_rows * in_cols, total = in_plane * op_size, out_plane = (long)rows * cols;
    const uint8_t *img = in + n * total;
    uint8_t *o = out + n * out_plane * op_size + ((long)ty * g0 + tx);
    int sum = 0, findex = 0;
    for (int oc = 0; oc < op_size; oc++) {
        if (!carry) sum = 0;
        const long base = plane0 ? 0 : in_plane * oc;
        for (int i = -half; i <= half; i++)
            for (int j = -half; j <= half; j++, findex++)
                sum = mac_i32(sum, lit_tap(img + base, total - base, lit, ty, tx, i, j, stride, g0, in_rows, in_cols),
                              filt[findex]);
        if (sum <= 0) sum = 0;
        o[out_plane * oc] = (uint8_t)sum;
    }
}

__global__ __launch_bounds__(256) void lit_pointwise_k(uint8_t *__restrict__ out, const uint8_t *__restrict__ in,
                                                       const int *__restrict__ filt, long plane, int cin, int op_size,
                                                       unsigned quirks, long in_image)
{
    const bool carry = quirks & MBN_Q_CARRY_SUM;
    const int n = blockIdx.z;
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= plane) return;
    const uint8_t *ip = in + n * in_image + p;
    uint8_t *o = out + n * plane * op_size + p;
    int sum = 0, findex = 0;
    for (int oc = 0; oc < op_size; oc++) {
        if (!carry) sum = 0;
        for (int i = 0; i < cin; i++, findex++) sum = mac_i32(sum, ip[plane * i], filt[findex]);
        if (sum <= 0) sum = 0;
        o[plane * oc] = (uint8_t)sum;
    }
}

// kernel.cl:116-132 as work-item (0,0) computes it. Without the carry quirk channels are independent (one lane
// per channel); with it the per-channel window sums are prefix-summed along the channel axis by one wave.
__global__ __launch_bounds__(256) void lit_pool_k(uint8_t *__restrict__ out, const uint8_t *__restrict__ in, long plane,
                                                  int taps, int op_size, int div)
{
    const int n = blockIdx.z;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= op_size) return;
    const uint8_t *ip = in + ((long)n * op_size + c) * plane;
    unsigned s = 0;
    for (int i = 0; i < taps; i++) s += ip[i];
    out[(long)n * op_size + c] = (uint8_t)((int)s / div);
}

__global__ __launch_bounds__(64) void lit_pool_carry_k(uint8_t *__restrict__ out, const uint8_t *__restrict__ in,
                                                       long plane, int taps, int op_size, int div)
{
    // one wave per image: lane-strided window sums, then a wave-wide inclusive scan carried across chunks of 64
    const int n = blockIdx.x, lane = threadIdx.x;
    unsigned carry = 0;
    for (int c0 = 0; c0 < op_size; c0 += 64) {
        int c = c0 + lane;
        unsigned s = 0;
        if (c < op_size) {
            const uint8_t *ip = in + ((long)n * op_size + c) * plane;
            for (int i = 0; i < taps; i++) s += ip[i];
        }
        for (int d = 1; d < 64; d <<= 1) {   // Hillis-Steele inclusive scan over the 64 lanes
            unsigned t = __shfl_up(s, d, 64);
            if (lane >= d) s += t;
        }
        s += carry;
        if (c < op_size) out[(long)n * op_size + c] = (uint8_t)((int)s / div);
        carry = __shfl(s, 63, 64);
    }
}

}   // namespace

// The emulated NDRange must not make two work-items write one byte (the reference's 224x224 launch over a
// 112x112 output plane does, MobileNet.c:291-292 — a data race, not reproducible): require G0*G1 <= plane.
static int check_gsize(int &g0, int &g1, int ocol, int orow, long out_plane)
{
    if (g0 <= 0) g0 = ocol;
    if (g1 <= 0) g1 = orow;
    if ((long)g0 * g1 > out_plane) return MBN_EINVAL;
    return MBN_OK;
}

int mbn_launch_lit_convolute(const mbn_call &c, uint8_t *out, const uint8_t *r, const uint8_t *g, const uint8_t *b,
                             const int32_t *filt, int rows, int cols, int fs, int stride, int op_size)
{
    const int orow = rows / stride, ocol = cols / stride;
    const bool lit = c.quirks & MBN_Q_LITERAL_INDEX;
    const long out_plane = lit ? (long)(rows / 2) * (cols / 2) : (long)orow * ocol;   // kernel.cl:14
    if (out_plane <= 0) return MBN_EINVAL;
    int g0 = c.g0, g1 = c.g1;
    int rc = check_gsize(g0, g1, ocol, orow, out_plane);
    if (rc) return rc;
    dim3 grid((unsigned)(((long)g0 * g1 + 255) / 256), 1, c.batch);
    hipLaunchKernelGGL(lit_convolute_k, grid, dim3(256), 0, c.stream, out, r, g, b, filt, rows, cols, fs, stride,
                       op_size, c.quirks, g0, g1, out_plane, (long)rows * cols, out_plane * op_size);
    return MBN_OK;
}

int mbn_launch_lit_depthwise(const mbn_call &c, uint8_t *out, const uint8_t *in, const int32_t *filt, int rows,
                             int cols, int fs, int stride, int op_size)
{
    int g0 = c.g0, g1 = c.g1;
    int rc = check_gsize(g0, g1, cols, rows, (long)rows * cols);
    if (rc) return rc;
    dim3 grid((unsigned)(((long)g0 * g1 + 255) / 256), 1, c.batch);
    hipLaunchKernelGGL(lit_depthwise_k, grid, dim3(256), 0, c.stream, out, in, filt, rows, cols, fs, stride, op_size,
                       c.quirks, g0, g1, c.in_rows, c.in_cols);
    return MBN_OK;
}

int mbn_launch_lit_pointwise(const mbn_call &c, uint8_t *out, const uint8_t *in, const int32_t *filt, int rows,
                             int cols, int cin, int op_size)
{
    const long plane = (long)rows * cols;
    dim3 grid((unsigned)((plane + 255) / 256), 1, c.batch);
    // the input image holds `cin` planes (the caller's buffer may hold more; batch stride uses cin)
    hipLaunchKernelGGL(lit_pointwise_k, grid, dim3(256), 0, c.stream, out, in, filt, plane, cin, op_size, c.quirks,
                       plane * cin);
    return MBN_OK;
}

int mbn_launch_lit_pool(const mbn_call &c, uint8_t *out, const uint8_t *in, int rows, int cols, int fs, int op_size)
{
    const long plane = (long)rows * cols;
    const int taps = fs * fs;
    const int div = (c.quirks & MBN_Q_POOL_DIV49) ? 49 : taps;
    if (c.quirks & MBN_Q_CARRY_SUM)
        hipLaunchKernelGGL(lit_pool_carry_k, dim3(c.batch), dim3(64), 0, c.stream, out, in, plane, taps, op_size, div);
    else
        hipLaunchKernelGGL(lit_pool_k, dim3((op_size + 255) / 256, 1, c.batch), dim3(256), 0, c.stream, out, in, plane,
                           taps, op_size, div);
    return MBN_OK;
}
