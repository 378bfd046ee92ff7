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

// ------------------------------------------------------------------------------------------------ pointwise on v_dot4
// the reference's integer MACs (kernel.cl:106-108: sum += in[p + rows*cols*i] * filter[findex]) on the
// packed-int8 dot unit. Preconditions, checked on the device: MBN_Q_CARRY_SUM off (channels independent) and every filter
// value in [-128, 127]; otherwise the kernel runs the scalar loop of lit_pointwise_k for the same work (no host round trip
// to decide). Bit-exact: int32 arithmetic is modular, so  sum_i w_i x_i  =  sum_i w_i (x_i - 128)  +  128 sum_i w_i  (mod 2^32),
// and x_i - 128 is x_i ^ 0x80 read as int8 — v_dot4_i32_i8 is signed x signed, the activations are unsigned.
//   pre-pass  lit_pack_filter_k: int32 [oc][cin] -> int8x4 dwords laid out [cin/4][ocp] (8 consecutive oc = one
//             s_load_dwordx8), per-oc sum of weights, and the "fits" flag;
//   main      one workgroup = 64 pixels x all output channels: the uint8 planes of the 64 pixels are gathered once into
//             LDS as [cin/4][64] dwords of 4 consecutive input channels (conflict-free both ways); wave w owns every 16th
//             block of 8 output channels: per 4 input channels one ds_read_b32, one scalar x8 weight load, eight v_dot4.
constexpr int LD_PIX = 64, LD_WAVES = 16, LD_OCR = 8;

__global__ __launch_bounds__(256) void lit_pack_filter_k(unsigned *__restrict__ w8, int *__restrict__ wsum, int *__restrict__ bad,
                                                         const int *__restrict__ filt, int cin, int op_size, int cin4, int ocp)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)cin4 * ocp) return;
    const int oc = (int)(t % ocp), g = (int)(t / ocp);
    unsigned pk = 0;
    int sum = 0;
    bool ok = true;
    if (oc < op_size)
        for (int j = 0; j < 4; j++) {
            const int ic = 4 * g + j;
            const int w = ic < cin ? filt[(long)oc * cin + ic] : 0;
            ok = ok && w >= -128 && w <= 127;
            pk |= ((unsigned)w & 0xffu) << (8 * j);
            sum += w;
        }
    w8[t] = pk;
    if (sum) atomicAdd(&wsum[oc], sum);
    if (!ok) atomicOr(bad, 1);
}

__global__ __launch_bounds__(64 * LD_WAVES) void lit_pointwise_dot_k(uint8_t *__restrict__ out, const uint8_t *__restrict__ in,
                                                                     const int *__restrict__ filt, const unsigned *__restrict__ w8,
                                                                     const int *__restrict__ wsum, const int *__restrict__ bad,
                                                                     long plane, int cin, int op_size, int cin4, int ocp, long in_image)
{
    extern __shared__ unsigned xs[];                       // [cin4][64] dwords: 4 consecutive input channels of one pixel, x ^ 0x80
    const int n = blockIdx.z, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long p = (long)blockIdx.x * LD_PIX + lane;
    const bool pok = p < plane;
    const uint8_t *ip = in + n * in_image + p;
    uint8_t *o = out + n * plane * op_size + p;
    if (*bad) {
        // a filter value outside int8: the scalar loop (lit_pointwise_k without the carry), output channels split over the waves
        if (!pok) return;
        for (int oc = wave; oc < op_size; oc += LD_WAVES) {
            int sum = 0;
            const int *f = filt + (long)oc * cin;
            for (int i = 0; i < cin; i++) sum = mac_i32(sum, ip[plane * i], f[i]);
            if (sum <= 0) sum = 0;
            o[plane * oc] = (uint8_t)sum;
        }
        return;
    }
    // gather: wave w takes the channel quads w, w+16, ...; lanes run along pixels, so the four byte loads are contiguous
    for (int g = wave; g < cin4; g += LD_WAVES) {
        unsigned pk = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int ic = 4 * g + j;
            const unsigned b = (pok && ic < cin) ? ip[plane * ic] : 0u;
            pk |= b << (8 * j);
        }
        xs[g * LD_PIX + lane] = pk ^ 0x80808080u;
    }
    __syncthreads();
    for (int ob = wave * LD_OCR; ob < ocp; ob += LD_WAVES * LD_OCR) {
        int acc[LD_OCR];
#pragma unroll
        for (int j = 0; j < LD_OCR; j++) acc[j] = 0;
        for (int g = 0; g < cin4; g++) {
            const int x = (int)xs[g * LD_PIX + lane];
            const unsigned *wr = w8 + (long)g * ocp + ob;                    // wave-uniform: scalar loads
#pragma unroll
            for (int j = 0; j < LD_OCR; j++) acc[j] = __builtin_amdgcn_sdot4(x, (int)wr[j], acc[j], false);
        }
        if (pok) {
#pragma unroll
            for (int j = 0; j < LD_OCR; j++) {
                const int oc = ob + j;
                if (oc < op_size) {
                    int sum = (int)((unsigned)acc[j] + 128u * (unsigned)wsum[oc]);   // + 128 * sum of weights (mod 2^32)
                    if (sum <= 0) sum = 0;
                    o[plane * oc] = (uint8_t)sum;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------ pointwise on the int8 matrix cores (round 5)
// SURVEY.md 8f-4's other form: the same integer MACs of kernel.cl:106-108 on v_mfma_i32_32x32x32_i8 (gfx950: 32 x 32 outputs x 32 input channels per
// instruction = what 128 v_dot4 instructions do). Same preconditions as the v_dot4 kernel (carry quirk off, every filter value in int8 — else the scalar
// loop on the device), same algebra (x - 128 = x ^ 0x80 as int8, + 128 * sum(w) afterwards, all mod 2^32), bit-exact.
//   pre-pass  lit_pack_filter_rows_k: int32 [oc][cin] -> int8 rows [ocp32][cin32] (zero padded: a lane's A operand is 16 consecutive input channels of one
//             output channel = one 16-byte load), per-oc sum of weights, the "fits" flag.
//   main      one workgroup = 64 pixels x all output channels, the pixels' uint8 planes gathered once into LDS as [cin32/4][64] dwords of 4 consecutive input
//             channels (the v_dot4 kernel's image: the planar NCHW layout puts k at plane stride, this is the transposition). A wave owns 32 x 32 (oc x pixel)
//             tiles: per 32 input channels its B operand is four ds_read_b32 (k = 16 * (lane / 32) + 0..15 of pixel lane % 32), its A operand one 16-byte
//             global load (the filter is L2-resident), then one MFMA. The k-to-byte assignment inside a lane is the same function for A and B, and integer
//             sums do not depend on the order, so any such assignment gives the exact dot product.
//   C/D       lane l holds pixel l % 32 and output channels (r & 3) + 8 * (r >> 2) + 4 * (l / 32), r = 0..15: 32 lanes store 32 consecutive bytes of a plane.
constexpr int LM_PIX = 64, LM_WAVES = 8;
typedef int lit_v4i __attribute__((ext_vector_type(4)));
typedef int lit_v16i __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void lit_pack_filter_rows_k(unsigned *__restrict__ w8r, int *__restrict__ wsum, int *__restrict__ bad,
                                                              const int *__restrict__ filt, int cin, int op_size, int cin32, int ocp32)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;           // one dword = 4 consecutive input channels of one output channel
    const int q = cin32 / 4;
    if (t >= (long)q * ocp32) return;
    const int oc = (int)(t / q), g = (int)(t % q);
    unsigned pk = 0;
    int sum = 0;
    bool ok = true;
    if (oc < op_size)
        for (int j = 0; j < 4; j++) {
            const int ic = 4 * g + j;
            const int w = ic < cin ? filt[(long)oc * cin + ic] : 0;
            ok = ok && w >= -128 && w <= 127;
            pk |= ((unsigned)w & 0xffu) << (8 * j);
            sum += w;
        }
    w8r[t] = pk;
    if (sum) atomicAdd(&wsum[oc], sum);
    if (!ok) atomicOr(bad, 1);
}

__global__ __launch_bounds__(64 * LM_WAVES) void lit_pointwise_mfma_k(uint8_t *__restrict__ out, const uint8_t *__restrict__ in,
                                                                      const int *__restrict__ filt, const unsigned *__restrict__ w8r,
                                                                      const int *__restrict__ wsum, const int *__restrict__ bad,
                                                                      long plane, int cin, int op_size, int cin32, int ocp32, long in_image)
{
    extern __shared__ unsigned xs[];                       // [cin32 / 4][64] dwords: 4 consecutive input channels of one pixel, x ^ 0x80
    const int n = blockIdx.z, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long p0 = (long)blockIdx.x * LM_PIX;
    const long p = p0 + lane;
    const bool pok = p < plane;
    const uint8_t *ip = in + n * in_image + p;
    if (*bad) {
        // a filter value outside int8: the scalar loop (lit_pointwise_k without the carry), output channels split over the waves
        if (!pok) return;
        uint8_t *o = out + n * plane * op_size + p;
        for (int oc = wave; oc < op_size; oc += LM_WAVES) {
            int sum = 0;
            const int *f = filt + (long)oc * cin;
            for (int i = 0; i < cin; i++) sum = mac_i32(sum, ip[plane * i], f[i]);
            if (sum <= 0) sum = 0;
            o[plane * oc] = (uint8_t)sum;
        }
        return;
    }
    const int q = cin32 / 4;
    for (int g = wave; g < q; g += LM_WAVES) {             // channels past cin: byte 0 ^ 0x80 meets a zero weight
        unsigned pk = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int ic = 4 * g + j;
            const unsigned b = (pok && ic < cin) ? ip[plane * ic] : 0u;
            pk |= b << (8 * j);
        }
        xs[g * LM_PIX + lane] = pk ^ 0x80808080u;
    }
    __syncthreads();
    const int li = lane & 31, lh = lane >> 5;
    const int ntile = (ocp32 / 32) * (LM_PIX / 32);
    for (int t = wave; t < ntile; t += LM_WAVES) {
        const int oc0 = (t >> 1) * 32, px0 = (t & 1) * 32;
        lit_v16i acc;
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] = 0;
        const lit_v4i *arow = reinterpret_cast<const lit_v4i *>(w8r + ((long)(oc0 + li) * cin32 + 16 * lh) / 4);
        const unsigned *brow = xs + (4 * lh) * LM_PIX + px0 + li;
        for (int ks = 0; ks < cin32 / 32; ks++) {
            const lit_v4i a = arow[2 * ks];                                    // 32 input channels = two 16-byte pieces per row: this lane's half
            lit_v4i b;
#pragma unroll
            for (int j = 0; j < 4; j++) b[j] = (int)brow[(8 * ks + j) * LM_PIX];
            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc, 0, 0, 0);
        }
        const long px = p0 + px0 + li;
        if (px < plane) {
            uint8_t *o = out + n * plane * op_size + px;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int oc = oc0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (oc < op_size) {
                    int sum = (int)((unsigned)acc[r] + 128u * (unsigned)wsum[oc]);   // + 128 * sum of weights (mod 2^32)
                    if (sum <= 0) sum = 0;
                    o[plane * oc] = (uint8_t)sum;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ 3x3 taps on v_dot4 (round 3)
// SURVEY.md 8f-4 for the two 3x3 kernels (kernel.cl:16-50 convolute, :75-86 depthwise): the three taps of one filter ROW are
// three adjacent bytes of a uint8 plane (x-1, x, x+1 — also at stride 2: 2x-1, 2x, 2x+1), so one unaligned dword load brings
// them (and a fourth byte that meets a zero weight) and one v_dot4_i32_i8 replaces three load + multiply-add groups with their
// index arithmetic. Preconditions, wave-uniform per output channel and checked in the kernel: the carry quirk off (channels
// independent) and the channel's weights inside int8; per lane: all three taps of the row inside the image / the plane and the
// fourth byte readable — otherwise that row takes the tap-by-tap path of lit_tap (image borders; LITERAL_INDEX at stride 2,
// whose taps are two bytes apart). Bit-exact by the same modular identity as the pointwise path: x - 128 is x ^ 0x80 as int8,
// and  sum w x = sum w (x - 128) + 128 sum w  (mod 2^32).
struct RowW { int pk; int bias; bool fits; int w[3]; };     // packed int8 weights (byte 3 = 0), 128 * (w0 + w1 + w2), range check

__device__ __forceinline__ RowW pack_row(const int *__restrict__ f)
{
    RowW r;
    r.w[0] = f[0]; r.w[1] = f[1]; r.w[2] = f[2];
    r.fits = r.w[0] >= -128 && r.w[0] <= 127 && r.w[1] >= -128 && r.w[1] <= 127 && r.w[2] >= -128 && r.w[2] <= 127;
    r.pk = (r.w[0] & 0xff) | ((r.w[1] & 0xff) << 8) | ((r.w[2] & 0xff) << 16);
    r.bias = 128 * (r.w[0] + r.w[1] + r.w[2]);
    return r;
}

// sum += the three taps (i, -1..1) of one filter row. `plane` = first byte of the input plane, `limit` = bytes readable from it.
__device__ __forceinline__ int row3(int sum, const uint8_t *__restrict__ plane, long limit, bool lit, int ty, int tx, int i,
                                    int stride, int g0, int in_rows, int in_cols, const RowW &rw)
{
    long idx;
    bool fast;
    if (lit) {                                         // kernel.cl:24: in[(ty+i)*G0*stride + (tx+j)*stride], zero when ty+i < 0 or tx+j < 0
        const int yi = ty + i, xi = tx - 1;
        idx = (long)yi * g0 * stride + (long)xi * stride;
        fast = stride == 1 && yi >= 0 && xi >= 0 && idx + 3 < limit;
    } else {
        const int iy = ty * stride + i, ix = tx * stride - 1;
        idx = (long)iy * in_cols + ix;
        fast = iy >= 0 && iy < in_rows && ix >= 0 && ix + 2 < in_cols && idx + 3 < limit;
    }
    if (fast && rw.fits) {
        unsigned x;
        __builtin_memcpy(&x, plane + idx, 4);          // unaligned dword load
        return (int)((unsigned)__builtin_amdgcn_sdot4((int)(x ^ 0x80808080u), rw.pk, sum, false) + (unsigned)rw.bias);
    }
#pragma unroll
    for (int j = -1; j <= 1; j++) sum = mac_i32(sum, lit_tap(plane, limit, lit, ty, tx, i, j, stride, g0, in_rows, in_cols), rw.w[j + 1]);
    return sum;
}

__global__ __launch_bounds__(256) void lit_depthwise_dot_k(uint8_t *__restrict__ out, const uint8_t *__restrict__ in,
                                                           const int *__restrict__ filt, int rows, int cols, int stride,
                                                           int op_size, unsigned quirks, int g0, int g1, int in_rows, int in_cols)
{
    const bool lit = quirks & MBN_Q_LITERAL_INDEX, plane0 = quirks & MBN_Q_DW_PLANE0;
    const int n = blockIdx.z;
    const long wi = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (wi >= (long)g0 * g1) return;
    const int ty = (int)(wi / g0), tx = (int)(wi % g0);
    const long in_plane = (long)in_rows * in_cols, total = in_plane * op_size, out_plane = (long)rows * cols;
    const uint8_t *img = in + n * total;
    uint8_t *o = out + n * out_plane * op_size + ((long)ty * g0 + tx);
    // the channels are independent without the carry quirk: one grid row per channel (weights wave-uniform), not a serial loop in
    // the lane — a 14 x 14 x 512 layer is 100 k work-items instead of 196 lanes with 512 iterations each
    for (int oc = blockIdx.y; oc < op_size; oc += gridDim.y) {
        const long base = plane0 ? 0 : in_plane * oc;
        int sum = 0;
#pragma unroll
        for (int i = -1; i <= 1; i++)
            sum = row3(sum, img + base, total - base, lit, ty, tx, i, stride, g0, in_rows, in_cols, pack_row(filt + oc * 9 + (i + 1) * 3));
        if (sum <= 0) sum = 0;
        o[out_plane * oc] = (uint8_t)sum;
    }
}

__global__ __launch_bounds__(256) void lit_convolute_dot_k(uint8_t *__restrict__ out, const uint8_t *__restrict__ r,
                                                           const uint8_t *__restrict__ g, const uint8_t *__restrict__ b,
                                                           const int *__restrict__ filt, int rows, int cols, int stride, int op_size,
                                                           unsigned quirks, int g0, int g1, long out_plane, long in_image, long out_image)
{
    const bool lit = quirks & MBN_Q_LITERAL_INDEX;
    const int n = blockIdx.z;
    const long wi = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (wi >= (long)g0 * g1) return;
    const int ty = (int)(wi / g0), tx = (int)(wi % g0);
    const uint8_t *planes[3] = { r + n * in_image, g + n * in_image, b + n * in_image };
    const long limit = (long)rows * cols;
    uint8_t *o = out + n * out_image + ((long)ty * g0 + tx);
    for (int oc = blockIdx.y; oc < op_size; oc += gridDim.y) {
        int sum = 0;
#pragma unroll
        for (int p = 0; p < 3; p++)
#pragma unroll
            for (int i = -1; i <= 1; i++)
                sum = row3(sum, planes[p], limit, lit, ty, tx, i, stride, g0, rows, cols, pack_row(filt + oc * 27 + p * 9 + (i + 1) * 3));
        if (sum <= 0) sum = 0;
        o[out_plane * oc] = (uint8_t)sum;
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
    // 3x3 without the carry quirk: filter rows on v_dot4 (tune lit_dot = 1 keeps the tap-by-tap kernel)
    if (fs == 3 && !(c.quirks & MBN_Q_CARRY_SUM) && g_mbn_tune.lit_dot != 1) {
        grid.y = (unsigned)(op_size < 65535 ? op_size : 65535);
        hipLaunchKernelGGL(lit_convolute_dot_k, grid, dim3(256), 0, c.stream, out, r, g, b, filt, rows, cols, stride, op_size, c.quirks,
                           g0, g1, out_plane, (long)rows * cols, out_plane * op_size);
        return MBN_OK;
    }
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
    if (fs == 3 && !(c.quirks & MBN_Q_CARRY_SUM) && g_mbn_tune.lit_dot != 1) {
        grid.y = (unsigned)(op_size < 65535 ? op_size : 65535);
        hipLaunchKernelGGL(lit_depthwise_dot_k, grid, dim3(256), 0, c.stream, out, in, filt, rows, cols, stride, op_size, c.quirks,
                           g0, g1, c.in_rows, c.in_cols);
        return MBN_OK;
    }
    hipLaunchKernelGGL(lit_depthwise_k, grid, dim3(256), 0, c.stream, out, in, filt, rows, cols, fs, stride, op_size,
                       c.quirks, g0, g1, c.in_rows, c.in_cols);
    return MBN_OK;
}

int mbn_launch_lit_pointwise(const mbn_call &c, uint8_t *out, const uint8_t *in, const int32_t *filt, int rows,
                             int cols, int cin, int op_size)
{
    const long plane = (long)rows * cols;
    // matrix-core / v_dot4 paths: channels independent (no carry quirk), on the context's own stream (the packed filter lives in a
    // per-context workspace), LDS tile within 64 KB. tune lit_dot: 0 = int8 MFMA where it measured faster (K >= 512, or >= 16384 pixels in the call:
    // profiles/r05/q_literal_int8_mfma.txt: 1.1-1.4x there, 0.7-0.8x on one image of the K <= 256 layers, where the call is its three launches), else v_dot4
    // where eligible; 1 = the scalar kernel; 2 = v_dot4 (never the MFMA form); 3 = the MFMA form wherever eligible: A/B hooks and what the tests force
    const int lit_dot = g_mbn_tune.lit_dot;
    const bool indep = !(c.quirks & MBN_Q_CARRY_SUM) && c.stream == c.ctx->stream && lit_dot != 1;
    const int cin32 = (cin + 31) / 32 * 32, ocp32 = (op_size + 31) / 32 * 32;
    if (indep && lit_dot != 2 && cin >= 16 && op_size >= 16 && (long)cin32 * LM_PIX <= 65536 &&
        (lit_dot == 3 || cin >= 512 || (double)c.batch * plane >= 16384.0)) {
        const size_t need = (size_t)cin32 * ocp32 + (size_t)ocp32 * 4 + 256;
        if (c.ctx->lit_ws_bytes < need) {
            if (c.ctx->lit_ws) { (void)hipStreamSynchronize(c.stream); (void)hipFree(c.ctx->lit_ws); c.ctx->lit_ws = nullptr; c.ctx->lit_ws_bytes = 0; }
            if (hipMalloc(&c.ctx->lit_ws, need) != hipSuccess) return MBN_ENOMEM;
            c.ctx->lit_ws_bytes = need;
        }
        unsigned *w8r = (unsigned *)c.ctx->lit_ws;
        int *wsum = (int *)(w8r + (size_t)cin32 * ocp32 / 4), *bad = wsum + ocp32;
        (void)hipMemsetAsync(wsum, 0, (size_t)ocp32 * 4 + 4, c.stream);
        const long nt = (long)(cin32 / 4) * ocp32;
        hipLaunchKernelGGL(lit_pack_filter_rows_k, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, c.stream, w8r, wsum, bad, filt, cin, op_size,
                           cin32, ocp32);
        dim3 grid((unsigned)((plane + LM_PIX - 1) / LM_PIX), 1, c.batch);
        hipLaunchKernelGGL(lit_pointwise_mfma_k, grid, dim3(64 * LM_WAVES), (size_t)cin32 * LM_PIX, c.stream, out, in, filt, w8r, wsum, bad, plane,
                           cin, op_size, cin32, ocp32, plane * cin);
        return MBN_OK;
    }
    const int cin4 = (cin + 3) / 4, ocp = (op_size + LD_OCR - 1) / LD_OCR * LD_OCR;
    if (indep && cin4 * LD_PIX * 4 <= 65536 && cin >= 8 && op_size >= 8) {
        const size_t need = (size_t)cin4 * ocp * 4 + (size_t)ocp * 4 + 256;
        if (c.ctx->lit_ws_bytes < need) {
            if (c.ctx->lit_ws) { (void)hipStreamSynchronize(c.stream); (void)hipFree(c.ctx->lit_ws); c.ctx->lit_ws = nullptr; c.ctx->lit_ws_bytes = 0; }
            if (hipMalloc(&c.ctx->lit_ws, need) != hipSuccess) return MBN_ENOMEM;
            c.ctx->lit_ws_bytes = need;
        }
        unsigned *w8 = (unsigned *)c.ctx->lit_ws;
        int *wsum = (int *)(w8 + (size_t)cin4 * ocp), *bad = wsum + ocp;
        (void)hipMemsetAsync(wsum, 0, (size_t)ocp * 4 + 4, c.stream);
        const long nt = (long)cin4 * ocp;
        hipLaunchKernelGGL(lit_pack_filter_k, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, c.stream, w8, wsum, bad, filt, cin,
                           op_size, cin4, ocp);
        dim3 grid((unsigned)((plane + LD_PIX - 1) / LD_PIX), 1, c.batch);
        hipLaunchKernelGGL(lit_pointwise_dot_k, grid, dim3(64 * LD_WAVES), (size_t)cin4 * LD_PIX * 4, c.stream, out, in, filt, w8,
                           wsum, bad, plane, cin, op_size, cin4, ocp, plane * cin);
        return MBN_OK;
    }
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
