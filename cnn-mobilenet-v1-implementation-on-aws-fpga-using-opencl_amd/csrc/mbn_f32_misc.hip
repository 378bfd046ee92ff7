// mbn_f32_misc.hip — the remaining fp32 NHWC stages of the path for gfx950:
//   conv1   (kernel.cl:2-60  `convolute`, 3x3xCin -> Cout, stride 2)      HBM-bound (AI 9.8 flop/B)
//   pool    (kernel.cl:116-132 `pool`, global average)                    HBM-bound
//   softmax (MobileNet.c:2771-2792, host loop in the reference)           classifier tail on device
//   u8->f32 normalise (front-end for MobileNet.c:215-238's uint8 image)   HBM-bound
#include "mbn_internal.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));

typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st4(float *p, f4 v) { *reinterpret_cast<f4 *>(p) = v; }
__device__ __forceinline__ void st4(__bf16 *p, f4 v)
{
    *reinterpret_cast<bf4 *>(p) = bf4{ (__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w };   // RNE
}

struct ConvArgs {
    void *out;
    const float *in, *filt, *scale, *shift;
    int batch, rows, cols, cin, fs, stride, orow, ocol, cout, pad_top, pad_left, act;
    long total;   // batch*orow*ocol*(cout/4)
    const uint8_t *in8;   // MBN_IO_IN_U8: the raw uint8 HWC image instead of `in`, normalised at load (x/127.5 - 1)
};

// Input element i of the image tensor: fp32 as stored, or uint8 with the Keras MobileNet preprocessing applied — the same
// fmaf as normalize_u8_f32, so conv(u8) == conv(normalize(u8)) bit for bit.
constexpr float NORM_SCALE = 1.0f / 127.5f, NORM_BIAS = -1.0f;
__device__ __forceinline__ float norm_u8(unsigned v) { return fmaf((float)v, NORM_SCALE, NORM_BIAS); }
__device__ __forceinline__ float ld_in(const ConvArgs &a, long i) { return a.in8 ? norm_u8(a.in8[i]) : a.in[i]; }

// One lane = 4 consecutive output channels of one output pixel; the cout/4 lanes of a pixel are adjacent, so a
// wave's store is one contiguous span of the NHWC output (the 73 % of this stage's traffic). Input taps of a pixel
// are shared by those lanes (same-address loads, one request); the whole filter (fs*fs*cin*cout floats) sits in LDS
// and is read as wave-broadcast float4s.
template <typename TO>
__global__ __launch_bounds__(256) void conv_f32_nhwc(ConvArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float wlds[];
    const int wcount = a.fs * a.fs * a.cin * a.cout;
    for (int i = threadIdx.x * 4; i < wcount; i += blockDim.x * 4)
        *reinterpret_cast<f4 *>(wlds + i) = *reinterpret_cast<const f4 *>(a.filt + i);
    __syncthreads();
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.total) return;
    const int c4n = a.cout >> 2;
    const int oc = (int)(t % c4n) << 2;
    long q = t / c4n;
    const int ox = (int)(q % a.ocol);
    q /= a.ocol;
    const int oy = (int)(q % a.orow);
    const int n = (int)(q / a.orow);
    const long img = (long)n * a.rows * a.cols * a.cin;
    f4 acc = f4{ 0.f, 0.f, 0.f, 0.f };
    for (int ky = 0; ky < a.fs; ky++) {
        const int iy = oy * a.stride + ky - a.pad_top;
        if (iy < 0 || iy >= a.rows) continue;
        for (int kx = 0; kx < a.fs; kx++) {
            const int ix = ox * a.stride + kx - a.pad_left;
            if (ix < 0 || ix >= a.cols) continue;
            const long ip = img + ((long)iy * a.cols + ix) * a.cin;
            const float *wp = wlds + (long)((ky * a.fs + kx) * a.cin) * a.cout + oc;
            for (int ci = 0; ci < a.cin; ci++) {
                const float v = ld_in(a, ip + ci);
                const f4 w = *reinterpret_cast<const f4 *>(wp + (long)ci * a.cout);
                acc.x = fmaf(v, w.x, acc.x); acc.y = fmaf(v, w.y, acc.y);
                acc.z = fmaf(v, w.z, acc.z); acc.w = fmaf(v, w.w, acc.w);
            }
        }
    }
    const f4 sc = a.scale ? *reinterpret_cast<const f4 *>(a.scale + oc) : f4{ 1.f, 1.f, 1.f, 1.f };
    const f4 sh = a.shift ? *reinterpret_cast<const f4 *>(a.shift + oc) : f4{ 0.f, 0.f, 0.f, 0.f };
    f4 v = f4{ fmaf(acc.x, sc.x, sh.x), fmaf(acc.y, sc.y, sh.y), fmaf(acc.z, sc.z, sh.z), fmaf(acc.w, sc.w, sh.w) };
    if (a.act == MBN_ACT_RELU6) {
        v.x = fminf(fmaxf(v.x, 0.f), 6.f); v.y = fminf(fmaxf(v.y, 0.f), 6.f);
        v.z = fminf(fmaxf(v.z, 0.f), 6.f); v.w = fminf(fmaxf(v.w, 0.f), 6.f);
    } else if (a.act == MBN_ACT_RELU) {
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    }
    st4(reinterpret_cast<TO *>(a.out) + t * 4, v);
}

// Specialisation for the network's first layer (3x3, Cin = 3, stride 2): one lane = 4 consecutive output channels of
// PX = 4 consecutive output pixels of one row. Per filter row the lane reads the 9 input pixels x 3 channels it needs
// as 6 aligned float4 + 3 floats (27 contiguous floats of the NHWC row; the cout/4 lanes of a pixel group read the same
// addresses, one request), and each LDS weight float4 is reused for the 4 pixels: 432 FMAs per 21 global + 27 LDS
// loads instead of 108 per 27 + 27. Stores stay whole 128-B lines per pixel.
// Measured and rejected: the same layer as an im2col GEMM on the matrix cores (14 v_mfma_f32_32x32x2_f32 per 32-pixel
// tile, operands gathered with 14 dword loads per lane, no LDS) was parity-correct but 8 % slower — the scattered
// 4-byte gather makes it address-rate bound, not instruction bound.
template <typename TO>
__global__ __launch_bounds__(256) void conv3x3s2c3_f32_nhwc(ConvArgs a)
{
    constexpr int PX = 4;
    extern __shared__ __attribute__((aligned(16))) float wlds[];
    const int wcount = 27 * a.cout;
    for (int i = threadIdx.x * 4; i < wcount; i += blockDim.x * 4)
        *reinterpret_cast<f4 *>(wlds + i) = *reinterpret_cast<const f4 *>(a.filt + i);
    __syncthreads();
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.total) return;
    const int c4n = a.cout >> 2;
    const int oc = (int)(t % c4n) << 2;
    long q = t / c4n;
    const int gcols = a.ocol / PX;
    const int ox0 = (int)(q % gcols) * PX;
    q /= gcols;
    const int oy = (int)(q % a.orow);
    const int n = (int)(q / a.orow);
    const long img = (long)n * a.rows * a.cols * 3;
    const int ix0 = ox0 * 2;                       // pad_left == 0 on this path
    const bool last_ok = ix0 + 2 * PX < a.cols;    // the 9th pixel is the right zero-pad column for the last group
    f4 acc[PX];
#pragma unroll
    for (int p = 0; p < PX; p++) acc[p] = f4{ 0.f, 0.f, 0.f, 0.f };
#pragma unroll
    for (int ky = 0; ky < 3; ky++) {
        const int iy = oy * 2 + ky - a.pad_top;
        if (iy < 0 || iy >= a.rows) continue;      // wave-uniform except across the oy boundary of a wave
        const long rowi = img + ((long)iy * a.cols + ix0) * 3;
        float v[27];
        if (a.in8) {                               // 27 contiguous bytes: 6 aligned dwords (cols % 4 == 0) + 3 bytes
            const uint8_t *row8 = a.in8 + rowi;
#pragma unroll
            for (int j = 0; j < 6; j++) {
                const unsigned x = *reinterpret_cast<const unsigned *>(row8 + 4 * j);
                v[4 * j] = norm_u8(x & 0xff); v[4 * j + 1] = norm_u8((x >> 8) & 0xff);
                v[4 * j + 2] = norm_u8((x >> 16) & 0xff); v[4 * j + 3] = norm_u8(x >> 24);
            }
            v[24] = last_ok ? norm_u8(row8[24]) : 0.f;
            v[25] = last_ok ? norm_u8(row8[25]) : 0.f;
            v[26] = last_ok ? norm_u8(row8[26]) : 0.f;
        } else {
            const float *row = a.in + rowi;
#pragma unroll
            for (int j = 0; j < 6; j++) {
                const f4 x = *reinterpret_cast<const f4 *>(row + 4 * j);
                v[4 * j] = x.x; v[4 * j + 1] = x.y; v[4 * j + 2] = x.z; v[4 * j + 3] = x.w;
            }
            v[24] = last_ok ? row[24] : 0.f;
            v[25] = last_ok ? row[25] : 0.f;
            v[26] = last_ok ? row[26] : 0.f;
        }
#pragma unroll
        for (int kx = 0; kx < 3; kx++)
#pragma unroll
            for (int ci = 0; ci < 3; ci++) {
                const f4 w = *reinterpret_cast<const f4 *>(wlds + ((ky * 3 + kx) * 3 + ci) * a.cout + oc);
#pragma unroll
                for (int p = 0; p < PX; p++) {
                    const float x = v[(2 * p + kx) * 3 + ci];
                    acc[p].x = fmaf(x, w.x, acc[p].x); acc[p].y = fmaf(x, w.y, acc[p].y);
                    acc[p].z = fmaf(x, w.z, acc[p].z); acc[p].w = fmaf(x, w.w, acc[p].w);
                }
            }
    }
    const f4 sc = a.scale ? *reinterpret_cast<const f4 *>(a.scale + oc) : f4{ 1.f, 1.f, 1.f, 1.f };
    const f4 sh = a.shift ? *reinterpret_cast<const f4 *>(a.shift + oc) : f4{ 0.f, 0.f, 0.f, 0.f };
    TO *op = reinterpret_cast<TO *>(a.out) + ((((long)n * a.orow + oy) * a.ocol + ox0) * a.cout) + oc;
#pragma unroll
    for (int p = 0; p < PX; p++) {
        f4 r = f4{ fmaf(acc[p].x, sc.x, sh.x), fmaf(acc[p].y, sc.y, sh.y), fmaf(acc[p].z, sc.z, sh.z),
                   fmaf(acc[p].w, sc.w, sh.w) };
        if (a.act == MBN_ACT_RELU6) {
            r.x = fminf(fmaxf(r.x, 0.f), 6.f); r.y = fminf(fmaxf(r.y, 0.f), 6.f);
            r.z = fminf(fmaxf(r.z, 0.f), 6.f); r.w = fminf(fmaxf(r.w, 0.f), 6.f);
        } else if (a.act == MBN_ACT_RELU) {
            r.x = fmaxf(r.x, 0.f); r.y = fmaxf(r.y, 0.f); r.z = fmaxf(r.z, 0.f); r.w = fmaxf(r.w, 0.f);
        }
        st4(op + (long)p * a.cout, r);
    }
}

// Round 4: the network's first layer at 32 output channels in fp32 (alpha = 1) as a K = 27 -> 28 GEMM on v_mfma_f32_16x16x4_f32 — the arithmetic of
// the fused stem's conv1 phase (mbn_f32_stem.hip, MCF), instruction for instruction, so that the stem stays bit-identical to the three separate
// launches: a wave = 16 adjacent output pixels of one row; lane (pc = l & 15, kg = l >> 4) supplies tap k = 4 t + kg of pixel pc as the B operand
// and w[k][16 h + pc] as the A operand of step t = 0..6, one accumulator chain per 16-channel half h, and receives channels 16 h + 4 kg .. + 3 of
// pixel pc (one float4 of BN + ReLU6, one 16-byte store: 128 contiguous bytes per pixel across h and kg). Taps outside the image and k = 27 are 0.
// The fmaf-chain kernel above stays for every other shape (and, as the comment there says, was 8 % faster than a 32x32x2 gather form in round 1;
// this one is off the default path — the fused stem runs layers 1-3 — and within a few percent of it: profiles/r04/f_stem_conv1_mfma.txt).
__global__ __launch_bounds__(256) void conv1_mfma_f32(ConvArgs a)
{
    const int lane = threadIdx.x & 63, kg = lane >> 4, pc = lane & 15;
    const int bpr = a.ocol >> 4;                                          // 16-pixel blocks per output row
    const long nblk = (long)a.batch * a.orow * bpr;
    const long nwave = ((long)gridDim.x * blockDim.x) >> 6;
    float w[2][7];
    int dk[7], drem[7];
#pragma unroll
    for (int t = 0; t < 7; t++) {
        const int k = 4 * t + kg;
        dk[t] = k / 9;                                                    // filter row (3: the padding tap k = 27)
        drem[t] = k % 9;                                                  // 3 kx + ci
#pragma unroll
        for (int h = 0; h < 2; h++) w[h][t] = k < 27 ? a.filt[k * 32 + 16 * h + pc] : 0.f;
    }
    f4 s1[2], b1[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        s1[h] = *reinterpret_cast<const f4 *>(a.scale + 16 * h + 4 * kg);
        b1[h] = *reinterpret_cast<const f4 *>(a.shift + 16 * h + 4 * kg);
    }
    for (long blk = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6; blk < nblk; blk += nwave) {
        const int bx = (int)(blk % bpr);
        const long q = blk / bpr;
        const int oy = (int)(q % a.orow);
        const long n = q / a.orow;
        const int ox = bx * 16 + pc;
        const long img = n * a.rows * a.cols * 3;
        float xv[7];
#pragma unroll
        for (int t = 0; t < 7; t++) {
            const int iy = oy * 2 + dk[t] - a.pad_top;
            const int fx = ox * 6 + drem[t];                              // pad_left == 0 on this path: float index inside the row
            const bool ok = dk[t] < 3 && iy >= 0 && iy < a.rows && fx < a.cols * 3;
            const long idx = img + (long)iy * a.cols * 3 + fx;
            xv[t] = ok ? (a.in8 ? norm_u8(a.in8[idx]) : a.in[idx]) : 0.f;
        }
        float *op = reinterpret_cast<float *>(a.out) + (((n * a.orow + oy) * a.ocol + ox) * 32) + 4 * kg;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            f4 acc = f4{ 0.f, 0.f, 0.f, 0.f };
#pragma unroll
            for (int t = 0; t < 7; t++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[h][t], xv[t], acc, 0, 0, 0);
            f4 r = f4{ fmaf(acc.x, s1[h].x, b1[h].x), fmaf(acc.y, s1[h].y, b1[h].y), fmaf(acc.z, s1[h].z, b1[h].z), fmaf(acc.w, s1[h].w, b1[h].w) };
            r.x = fminf(fmaxf(r.x, 0.f), 6.f); r.y = fminf(fmaxf(r.y, 0.f), 6.f);
            r.z = fminf(fmaxf(r.z, 0.f), 6.f); r.w = fminf(fmaxf(r.w, 0.f), 6.f);
            *reinterpret_cast<f4 *>(op + 16 * h) = r;
        }
    }
}

// any cout / unaligned: one lane per output element, filter from global memory
template <typename TO>
__global__ __launch_bounds__(256) void conv_generic_f32_nhwc(ConvArgs a)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)a.batch * a.orow * a.ocol * a.cout;
    if (t >= total) return;
    const int oc = (int)(t % a.cout);
    long q = t / a.cout;
    const int ox = (int)(q % a.ocol);
    q /= a.ocol;
    const int oy = (int)(q % a.orow);
    const int n = (int)(q / a.orow);
    const long img = (long)n * a.rows * a.cols * a.cin;
    float acc = 0.f;
    for (int ky = 0; ky < a.fs; ky++) {
        const int iy = oy * a.stride + ky - a.pad_top;
        if (iy < 0 || iy >= a.rows) continue;
        for (int kx = 0; kx < a.fs; kx++) {
            const int ix = ox * a.stride + kx - a.pad_left;
            if (ix < 0 || ix >= a.cols) continue;
            for (int ci = 0; ci < a.cin; ci++)
                acc = fmaf(ld_in(a, img + ((long)iy * a.cols + ix) * a.cin + ci),
                           a.filt[((long)(ky * a.fs + kx) * a.cin + ci) * a.cout + oc], acc);
        }
    }
    float v = fmaf(acc, a.scale ? a.scale[oc] : 1.f, a.shift ? a.shift[oc] : 0.f);
    if (a.act == MBN_ACT_RELU6) v = fminf(fmaxf(v, 0.f), 6.f);
    else if (a.act == MBN_ACT_RELU) v = fmaxf(v, 0.f);
    reinterpret_cast<TO *>(a.out)[t] = (TO)v;
}

// global average pool: one lane per (image, channel); lanes run along channels (coalesced NHWC reads).
template <typename T>
__global__ __launch_bounds__(256) void pool_f32_nhwc(T *__restrict__ out, const T *__restrict__ in, int batch,
                                                     int rows, int cols, int fr, int fc, int ch)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)batch * ch) return;
    const int c = (int)(t % ch);
    const int n = (int)(t / ch);
    const T *ip = in + (long)n * rows * cols * ch + c;
    float acc = 0.f;
    if (fr == 7 && fc == 7) {
        // the network's 7x7 window: all 49 loads are issued before the first add (the plain loop below is a chain of
        // dependent load -> add pairs: 16 us at batch 1, one memory round trip per tap); same left-to-right sum, same bits
        float v[49];
#pragma unroll
        for (int i = 0; i < 49; i++) v[i] = (float)ip[((long)(i / 7) * cols + (i % 7)) * ch];
#pragma unroll
        for (int i = 0; i < 49; i++) acc += v[i];
    } else
    for (int y = 0; y < fr; y++)
        for (int x = 0; x < fc; x++) acc += (float)ip[((long)y * cols + x) * ch];
    out[t] = (T)(acc / (float)(fr * fc));
}

// Global average pool + FC in ONE launch for 1...4 images (MobileNet.c:2601-2739: the `pool` launch and the `pointwise` launch
// with rows = cols = 1; at batch 1 a forward is launch-bound, so a launch saved is ~3 % of it). Workgroup (ks, ns)
// owns channels [64 ks, 64 ks + 64) and classes [ns * npc, (ns + 1) * npc):
//   1. pooled[b][c] for its 64 channels: the same left-to-right sum over the window and the same division as pool_f32_nhwc
//      (bit-identical pooled values), lanes along channels (256-byte rows);
//   2. its 64-channel slice of every class row (256 contiguous bytes per row: 16 lanes x float4) against the pooled slice,
//      16-lane butterfly sum -> partial[ks][b][n] in the workspace;
//   3. the LAST workgroup of a class range to finish (one counter per range, agent-scope release/acquire around it) adds the
//      K/64 partials in slice order and the bias: a fixed summation order, no float atomics, independent of the batch.
// The filter (4 MB) is read once per forward, the pool input once per class range.
struct PoolFcArgs {
    float *out;              // [batch][classes]
    const float *in;         // [batch][pix][ch]
    const float *w, *bias;   // [classes][ch], [classes] or null
    float *ws;               // [nks][4][classes] partial sums
    unsigned *cnt;           // [nns <= 64] arrival counters + [64] the range counter of the one-launch tail, zero between launches
    int batch, pix, ch, classes, nks, nns, npc;
    // k > 0: the whole classifier tail in this launch (MobileNet.c:2601-2792; SURVEY 8f-3 as written): the LAST class range to finish
    // (a second counter, cnt[64]) reads the complete logits back into LDS and runs softmax + top-k for every image
    int k;
    float *probs;            // [batch][classes] or null
    int *topk_idx;           // [batch][k]
    float *topk_prob;        // [batch][k]
};
constexpr int TAIL_CLASSES_MAX = 1024;      // one image's logits in LDS (4 KB)

__device__ __forceinline__ void softmax_topk_wave(float *__restrict__ probs, int *__restrict__ topk_idx, float *__restrict__ topk_prob,
                                                  const float *l, long n, int classes, int k);

// TAIL (a.k > 0): the one-launch classifier tail; the plain pool + FC launch (k = 0: mbn_pool_fc) is the TAIL = false instantiation, which has neither the
// 16 KB of logits in LDS nor the second hand-over.
template <bool TAIL>
__global__ __launch_bounds__(256) void poolfc_f32(PoolFcArgs a)
{
    __shared__ float pooled[4][64];
    __shared__ unsigned s_last;
    const int ks = blockIdx.x % a.nks, ns = blockIdx.x / a.nks;
    const int tid = threadIdx.x;
    {
        const int b = tid >> 6, c = tid & 63;
        float acc = 0.f;
        if (b < a.batch) {
            const float *ip = a.in + (long)b * a.pix * a.ch + ks * 64 + c;
            if (a.pix == 49) {
                float v[49];
#pragma unroll
                for (int i = 0; i < 49; i++) v[i] = ip[(long)i * a.ch];
#pragma unroll
                for (int i = 0; i < 49; i++) acc += v[i];
            } else
                for (int i = 0; i < a.pix; i++) acc += ip[(long)i * a.ch];
            acc = acc / (float)a.pix;
        }
        pooled[b][c] = acc;
    }
    __syncthreads();
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int r = tid >> 4, q = tid & 15;                       // 16 class rows per pass, 16 float4 per 64-channel slice
    f4 pv[4];
#pragma unroll
    for (int b = 0; b < 4; b++) pv[b] = *reinterpret_cast<const f4 *>(&pooled[b][q * 4]);
    const int n_lo = ns * a.npc, n_hi = min(a.classes, n_lo + a.npc);
    for (int n0 = n_lo; n0 < n_hi; n0 += 16) {
        const int n = n0 + r;
        const bool ok = n < n_hi;
        const f4 wv = ok ? *reinterpret_cast<const f4 *>(a.w + (long)n * a.ch + ks * 64 + q * 4) : f4{ 0.f, 0.f, 0.f, 0.f };
#pragma unroll
        for (int b = 0; b < 4; b++) {
            float p = fmaf(wv.w, pv[b].w, fmaf(wv.z, pv[b].z, fmaf(wv.y, pv[b].y, wv.x * pv[b].x)));
            p += __shfl_xor(p, 8, 64);
            p += __shfl_xor(p, 4, 64);
            p += __shfl_xor(p, 2, 64);
            p += __shfl_xor(p, 1, 64);
            // agent-scope store (sc1: written through to where every XCD sees it) — a plain store would need the release fence to write the
            // whole L2 of this XCD back (buffer_wbl2: ~20 us with the previous layers' activations dirty in it, measured)
            if (q == 0 && ok && b < a.batch) __hip_atomic_store(a.ws + ((long)ks * 4 + b) * a.classes + n, p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // hand-over: every wave waits for the acknowledgement of its own write-through stores (explicit s_waitcnt vmcnt(0): the workgroup-scope
    // fence + barrier lower to `s_waitcnt lgkmcnt(0); s_barrier` only, which would let the arrival be counted with partials still in flight —
    // ADVICE r3) before the barrier that precedes the counter atomic; the counter and the reads of the last workgroup are agent-scope
    // accesses too, so no cache is written back or invalidated
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (tid == 0) s_last = __hip_atomic_fetch_add(a.cnt + ns, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (s_last != (unsigned)a.nks - 1u) return;
    for (int i = tid; i < (n_hi - n_lo) * a.batch; i += 256) {
        const int b = i / (n_hi - n_lo), n = n_lo + i % (n_hi - n_lo);
        // all K/64 <= 16 partials requested before the first add (one memory round trip, not one per slice); slices past nks read slice 0
        float pt[16];
#pragma unroll
        for (int k = 0; k < 16; k++)
            pt[k] = __hip_atomic_load(a.ws + ((long)(k < a.nks ? k : 0) * 4 + b) * a.classes + n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 16; k++) v += k < a.nks ? pt[k] : 0.f;
        const float lg = v + (a.bias ? a.bias[n] : 0.f);
        // with the tail in this launch the logits are read back by another workgroup (possibly on another XCD): agent-scope stores, like the partials
        if constexpr (TAIL) __hip_atomic_store(a.out + (long)b * a.classes + n, lg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else a.out[(long)b * a.classes + n] = lg;
    }
    if (tid == 0) __hip_atomic_store(a.cnt + ns, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch on this workspace (stream order)
    if constexpr (TAIL) {
    // ---- second hand-over: this range's logits are acknowledged (explicit vmcnt(0) in every wave, then the barrier) before it is counted as done;
    // the range that counts nns - 1 is the last one and owns the softmax + top-k of every image
    __shared__ float s_logits[4][TAIL_CLASSES_MAX];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) s_last = __hip_atomic_fetch_add(a.cnt + 64, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (s_last != (unsigned)a.nns - 1u) return;
    for (int i = tid; i < a.batch * a.classes; i += 256)
        s_logits[i / a.classes][i % a.classes] = __hip_atomic_load(a.out + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid == 0) __hip_atomic_store(a.cnt + 64, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    // one WAVE per image (the images side by side instead of one after the other: 18 / 27 / 45 us for 1 / 2 / 4 images with the block form, measured)
    const int wv = tid >> 6;
    if (wv < a.batch) softmax_topk_wave(a.probs, a.topk_idx, a.topk_prob, s_logits[wv], wv, a.classes, a.k);
    }
}

// softmax + argmax: one 256-lane workgroup per image; wave shuffles then a 4-entry LDS combine.
__global__ __launch_bounds__(256) void softmax_f32(float *__restrict__ probs, int *__restrict__ argmax,
                                                   const float *__restrict__ logits, int classes)
{
    __shared__ float s_val[4];
    __shared__ int s_idx[4];
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *l = logits + (long)n * classes;
    float mx = -INFINITY;
    int am = 0x7fffffff;
    for (int k = tid; k < classes; k += 256) {
        float v = l[k];
        if (v > mx) { mx = v; am = k; }
    }
    for (int d = 32; d >= 1; d >>= 1) {
        float ov = __shfl_xor(mx, d, 64);
        int oi = __shfl_xor(am, d, 64);
        if (ov > mx || (ov == mx && oi < am)) { mx = ov; am = oi; }
    }
    if (lane == 0) { s_val[wave] = mx; s_idx[wave] = am; }
    __syncthreads();
    mx = s_val[0]; am = s_idx[0];
    for (int w = 1; w < 4; w++)
        if (s_val[w] > mx || (s_val[w] == mx && s_idx[w] < am)) { mx = s_val[w]; am = s_idx[w]; }
    __syncthreads();
    float sum = 0.f;
    for (int k = tid; k < classes; k += 256) sum += expf(l[k] - mx);
    for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d, 64);
    if (lane == 0) s_val[wave] = sum;
    __syncthreads();
    sum = s_val[0] + s_val[1] + s_val[2] + s_val[3];
    if (probs)
        for (int k = tid; k < classes; k += 256) probs[(long)n * classes + k] = expf(l[k] - mx) / sum;
    if (argmax && tid == 0) argmax[n] = am;
}

// softmax + top-k per image (k <= 8): one workgroup per image. The k winners are found by k block-wide arg-max passes
// over the logits in the total order (value descending, index ascending) — pass j only admits entries strictly after
// pass j-1's winner in that order, so ties resolve to the lowest index like the oracle's strict '>' scan and no
// "taken" list is needed. probs (may be NULL) gets the full distribution; topk_prob the winners' probabilities.
// One image by one 256-lane workgroup; `l` may point to global memory or to an LDS copy of the image's logits (the one-launch classifier tail).
__device__ __forceinline__ void softmax_topk_block(float *__restrict__ probs, int *__restrict__ topk_idx, float *__restrict__ topk_prob,
                                                   const float *l, long n, int classes, int k, float *s_val, int *s_idx)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float pv = INFINITY, mx = 0.f, sum = 0.f;     // previous winner (value, index); +inf admits everything
    int pi = -1;
    for (int j = 0; j < k; j++) {
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        for (int c = tid; c < classes; c += 256) {
            const float v = l[c];
            const bool admitted = v < pv || (v == pv && c > pi);
            if (admitted && (v > bv || (v == bv && c < bi))) { bv = v; bi = c; }
        }
        for (int d = 32; d >= 1; d >>= 1) {
            const float ov = __shfl_xor(bv, d, 64);
            const int oi = __shfl_xor(bi, d, 64);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) { s_val[wave] = bv; s_idx[wave] = bi; }
        __syncthreads();
        bv = s_val[0]; bi = s_idx[0];
        for (int w = 1; w < 4; w++)
            if (s_val[w] > bv || (s_val[w] == bv && s_idx[w] < bi)) { bv = s_val[w]; bi = s_idx[w]; }
        __syncthreads();
        if (j == 0) {                              // the first winner is the maximum: normaliser of the softmax
            mx = bv;
            for (int c = tid; c < classes; c += 256) sum += expf(l[c] - mx);
            for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d, 64);
            if (lane == 0) s_val[wave] = sum;
            __syncthreads();
            sum = s_val[0] + s_val[1] + s_val[2] + s_val[3];
            __syncthreads();
            if (probs)
                for (int c = tid; c < classes; c += 256) probs[(long)n * classes + c] = expf(l[c] - mx) / sum;
        }
        if (tid == 0) {
            const bool found = bi != 0x7fffffff;   // fewer than k classes
            topk_idx[(long)n * k + j] = found ? bi : -1;
            topk_prob[(long)n * k + j] = found ? expf(bv - mx) / sum : 0.f;
        }
        pv = bv; pi = bi;
    }
}

// The same result by ONE wave, bit for bit: lane L plays lanes L of the block form's four waves — four partial sums per lane over the classes
// c = 64 w + L + 256 j in the same order, each reduced by the same xor tree, added left to right — and the winners come from a total order, which
// no summation order can change. `l` is an LDS copy of the image's logits. No workgroup barrier: the images of a call run side by side, one per wave.
__device__ __forceinline__ void softmax_topk_wave(float *__restrict__ probs, int *__restrict__ topk_idx, float *__restrict__ topk_prob,
                                                  const float *l, long n, int classes, int k)
{
    // the lane's classes c = lane + 64 i (i < 16: classes <= TAIL_CLASSES_MAX) live in registers: the k ranking passes and the softmax are then
    // straight-line VALU (with the logits re-read from LDS in every pass a wave took 16 us per image: one dependent LDS round trip per class and pass)
    constexpr int PL = TAIL_CLASSES_MAX / 64;
    const int lane = threadIdx.x & 63;
    float v[PL];
#pragma unroll
    for (int i = 0; i < PL; i++) v[i] = lane + 64 * i < classes ? l[lane + 64 * i] : -INFINITY;
    float pv = INFINITY, mx = 0.f, sum = 0.f;
    int pi = -1;
    for (int j = 0; j < k; j++) {
        float bv = -INFINITY;
        int bi = 0x7fffffff;
#pragma unroll
        for (int i = 0; i < PL; i++) {
            const int c = lane + 64 * i;
            const bool admitted = c < classes && (v[i] < pv || (v[i] == pv && c > pi));
            if (admitted && (v[i] > bv || (v[i] == bv && c < bi))) { bv = v[i]; bi = c; }
        }
        for (int d = 32; d >= 1; d >>= 1) {
            const float ov = __shfl_xor(bv, d, 64);
            const int oi = __shfl_xor(bi, d, 64);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (j == 0) {
            mx = bv;
            float ps[4] = { 0.f, 0.f, 0.f, 0.f };
#pragma unroll
            for (int i = 0; i < PL; i++)                                   // class 64 i + lane = 64 (i & 3) + lane + 256 (i >> 2): wave i & 3, trip i >> 2 of the block form
                if (lane + 64 * i < classes) ps[i & 3] += expf(v[i] - mx);
#pragma unroll
            for (int w = 0; w < 4; w++)
                for (int d = 32; d >= 1; d >>= 1) ps[w] += __shfl_xor(ps[w], d, 64);
            sum = ps[0] + ps[1] + ps[2] + ps[3];
            if (probs) {
#pragma unroll
                for (int i = 0; i < PL; i++)
                    if (lane + 64 * i < classes) probs[n * classes + lane + 64 * i] = expf(v[i] - mx) / sum;
            }
        }
        if (lane == 0) {
            const bool found = bi != 0x7fffffff;   // fewer than k classes
            topk_idx[n * k + j] = found ? bi : -1;
            topk_prob[n * k + j] = found ? expf(bv - mx) / sum : 0.f;
        }
        pv = bv; pi = bi;
    }
}

__global__ __launch_bounds__(256) void softmax_topk_f32(float *__restrict__ probs, int *__restrict__ topk_idx,
                                                        float *__restrict__ topk_prob, const float *__restrict__ logits,
                                                        int classes, int k)
{
    __shared__ float s_val[4];
    __shared__ int s_idx[4];
    softmax_topk_block(probs, topk_idx, topk_prob, logits + (long)blockIdx.x * classes, blockIdx.x, classes, k, s_val, s_idx);
}

__global__ __launch_bounds__(256) void normalize_u8_f32(float *__restrict__ out, const uint8_t *__restrict__ in,
                                                        size_t count, float scale, float bias)
{
    const size_t t = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (t + 3 < count) {
        const uchar4 v = *reinterpret_cast<const uchar4 *>(in + t);
        *reinterpret_cast<f4 *>(out + t) =
            f4{ fmaf((float)v.x, scale, bias), fmaf((float)v.y, scale, bias), fmaf((float)v.z, scale, bias),
                fmaf((float)v.w, scale, bias) };
    } else {
        for (size_t i = t; i < count; i++) out[i] = fmaf((float)in[i], scale, bias);
    }
}

// fp32 <-> bf16 (round to nearest even), 4 elements per lane
__global__ __launch_bounds__(256) void convert_f32_bf16(__bf16 *__restrict__ dst, const float *__restrict__ src, size_t count)
{
    const size_t t = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (t + 3 < count) st4(dst + t, *reinterpret_cast<const f4 *>(src + t));
    else for (size_t i = t; i < count; i++) dst[i] = (__bf16)src[i];
}
__global__ __launch_bounds__(256) void convert_bf16_f32(float *__restrict__ dst, const __bf16 *__restrict__ src, size_t count)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < count) dst[t] = (float)src[t];
}

}   // namespace

int mbn_launch_convert(mbn_context *, hipStream_t s, void *dst, const void *src, size_t count, int to_bf16)
{
    if (to_bf16) {
        if (((uintptr_t)dst % 8) || ((uintptr_t)src % 16)) return MBN_EINVAL;
        hipLaunchKernelGGL(convert_f32_bf16, dim3((unsigned)(((count + 3) / 4 + 255) / 256)), dim3(256), 0, s, (__bf16 *)dst,
                           (const float *)src, count);
    } else {
        hipLaunchKernelGGL(convert_bf16_f32, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, (float *)dst,
                           (const __bf16 *)src, count);
    }
    return MBN_OK;
}

int mbn_launch_f32_conv(const mbn_call &c, void *out, const void *in_v, const float *filt, int rows, int cols, int fs,
                        int stride, int op_size)
{
    const bool bf = c.dtype == MBN_DT_BF16;
    const bool u8in = (c.io_flags & MBN_IO_IN_U8) != 0;
    if (bf && !(c.io_flags & (MBN_IO_IN_F32 | MBN_IO_IN_U8))) return MBN_EUNSUPPORTED;   // the first layer reads the fp32 or raw image
    const float *in = (const float *)in_v;
    ConvArgs a;
    a.in8 = u8in ? (const uint8_t *)in_v : nullptr;
    a.out = out; a.in = in; a.filt = filt; a.scale = c.scale; a.shift = c.shift;
    a.batch = c.batch; a.rows = rows; a.cols = cols; a.cin = c.cin; a.fs = fs; a.stride = stride;
    a.orow = (rows + stride - 1) / stride; a.ocol = (cols + stride - 1) / stride; a.cout = op_size;
    a.pad_top = c.pad_top >= 0 ? c.pad_top : mbn_same_pad(rows, a.orow, fs, stride);
    a.pad_left = c.pad_left >= 0 ? c.pad_left : mbn_same_pad(cols, a.ocol, fs, stride);
    a.act = c.act;
    const size_t wbytes = (size_t)fs * fs * c.cin * op_size * sizeof(float);
    const bool fast = (op_size % 4) == 0 && wbytes <= 64 * 1024 && ((uintptr_t)out % (bf ? 8 : 16)) == 0 &&
                      ((uintptr_t)filt % 16) == 0 && (!c.scale || ((uintptr_t)c.scale % 16) == 0) &&
                      (!c.shift || ((uintptr_t)c.shift % 16) == 0);
    const bool first_layer = fast && fs == 3 && c.cin == 3 && stride == 2 && a.pad_left == 0 && (cols % 4) == 0 &&
                             (a.ocol % 4) == 0 && 2 * a.ocol == cols && ((uintptr_t)in % (u8in ? 4 : 16)) == 0 &&
                             g_mbn_tune.conv_variant != 1;
    // alpha = 1 in fp32: the MFMA form — the fused stem's conv1 arithmetic (see conv1_mfma_f32); lab conv_variant = 6: the fmaf-chain kernel instead (A/B)
    if (first_layer && !bf && op_size == 32 && (a.ocol % 16) == 0 && c.act == MBN_ACT_RELU6 && c.scale && c.shift && g_mbn_tune.conv_variant != 6 &&
        (double)c.batch * rows * cols * 3 < 9.0e18) {
        const long nblk = (long)c.batch * a.orow * (a.ocol / 16);
        long wgs = (nblk + 3) / 4;                                        // 4 waves per workgroup, one block per wave and trip
        const long cap = (long)c.ctx->num_cus * 8;
        if (wgs > cap) wgs = cap;
        a.total = nblk;
        hipLaunchKernelGGL(conv1_mfma_f32, dim3((unsigned)wgs), dim3(256), 0, c.stream, a);
    } else if (first_layer) {
        a.total = (long)c.batch * a.orow * (a.ocol / 4) * (op_size / 4);
        const dim3 grid((unsigned)((a.total + 255) / 256));
        if (bf) hipLaunchKernelGGL(conv3x3s2c3_f32_nhwc<__bf16>, grid, dim3(256), wbytes, c.stream, a);
        else hipLaunchKernelGGL(conv3x3s2c3_f32_nhwc<float>, grid, dim3(256), wbytes, c.stream, a);
    } else if (fast) {
        a.total = (long)c.batch * a.orow * a.ocol * (op_size / 4);
        const dim3 grid((unsigned)((a.total + 255) / 256));
        if (bf) hipLaunchKernelGGL(conv_f32_nhwc<__bf16>, grid, dim3(256), wbytes, c.stream, a);
        else hipLaunchKernelGGL(conv_f32_nhwc<float>, grid, dim3(256), wbytes, c.stream, a);
    } else {
        a.total = 0;
        long total = (long)c.batch * a.orow * a.ocol * op_size;
        const dim3 grid((unsigned)((total + 255) / 256));
        if (bf) hipLaunchKernelGGL(conv_generic_f32_nhwc<__bf16>, grid, dim3(256), 0, c.stream, a);
        else hipLaunchKernelGGL(conv_generic_f32_nhwc<float>, grid, dim3(256), 0, c.stream, a);
    }
    return MBN_OK;
}

int mbn_launch_f32_pool(const mbn_call &c, void *out, const void *in, int rows, int cols, int fs, int channels)
{
    const int fr = fs < rows ? fs : rows, fc = fs < cols ? fs : cols;
    const long total = (long)c.batch * channels;
    const dim3 grid((unsigned)((total + 255) / 256));
    if (c.dtype == MBN_DT_BF16)
        hipLaunchKernelGGL(pool_f32_nhwc<__bf16>, grid, dim3(256), 0, c.stream, (__bf16 *)out, (const __bf16 *)in, c.batch,
                           rows, cols, fr, fc, channels);
    else
        hipLaunchKernelGGL(pool_f32_nhwc<float>, grid, dim3(256), 0, c.stream, (float *)out, (const float *)in, c.batch,
                           rows, cols, fr, fc, channels);
    return MBN_OK;
}

// workspace: 64 range-arrival counters + the tail's counter (512 bytes at the front), then [ch/64][4][classes] floats; zeroed once by the caller
size_t mbn_pool_fc_ws_bytes(int channels, int classes)
{
    return 512 + (size_t)(channels / 64) * 4 * (size_t)classes * sizeof(float);
}

// k > 0: softmax + top-k of every image in the same launch (probs may be null); k = 0: pool + FC only
int mbn_launch_f32_pool_fc(mbn_context *ctx, hipStream_t s, float *out, const float *in, const float *w, const float *bias, void *ws,
                           int batch, int pix, int channels, int classes, int k, float *probs, int32_t *topk_idx, float *topk_prob)
{
    if (batch < 1 || batch > 4 || channels < 64 || channels > 1024 || (channels % 64) != 0 || pix <= 0 || classes <= 0) return MBN_EUNSUPPORTED;
    if (((uintptr_t)w % 16) != 0 || ((uintptr_t)ws % 16) != 0) return MBN_EUNSUPPORTED;
    if (k < 0 || k > 8 || (k > 0 && (classes > TAIL_CLASSES_MAX || !topk_idx || !topk_prob))) return k > 0 && classes > TAIL_CLASSES_MAX ? MBN_EUNSUPPORTED : MBN_EINVAL;
    PoolFcArgs a;
    a.out = out; a.in = in; a.w = w; a.bias = bias;
    a.k = k; a.probs = probs; a.topk_idx = (int *)topk_idx; a.topk_prob = topk_prob;
    a.cnt = (unsigned *)ws;
    a.ws = (float *)((char *)ws + 512);
    a.batch = batch; a.pix = pix; a.ch = channels; a.classes = classes;
    a.nks = channels / 64;
    // class ranges: enough workgroups for every CU to hold one (the filter slice of a workgroup is npc x 256 bytes), at most 64 ranges
    int nns = (ctx->num_cus + a.nks - 1) / a.nks;
    if (nns > 64) nns = 64;
    if (nns > (classes + 15) / 16) nns = (classes + 15) / 16;
    a.npc = ((classes + nns - 1) / nns + 15) / 16 * 16;
    a.nns = (classes + a.npc - 1) / a.npc;
    if (a.k > 0) hipLaunchKernelGGL(poolfc_f32<true>, dim3((unsigned)(a.nks * a.nns)), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(poolfc_f32<false>, dim3((unsigned)(a.nks * a.nns)), dim3(256), 0, s, a);
    return MBN_OK;
}

int mbn_launch_f32_softmax(mbn_context *, hipStream_t s, float *probs, int32_t *argmax, const float *logits, int batch,
                           int classes)
{
    hipLaunchKernelGGL(softmax_f32, dim3(batch), dim3(256), 0, s, probs, argmax, logits, classes);
    return MBN_OK;
}

int mbn_launch_f32_softmax_topk(mbn_context *, hipStream_t s, float *probs, int32_t *topk_idx, float *topk_prob,
                                const float *logits, int batch, int classes, int k)
{
    if (k < 1 || k > 8) return MBN_EINVAL;
    hipLaunchKernelGGL(softmax_topk_f32, dim3((unsigned)batch), dim3(256), 0, s, probs, topk_idx, topk_prob, logits, classes, k);
    return MBN_OK;
}

int mbn_launch_normalize(mbn_context *, hipStream_t s, float *out, const uint8_t *in, size_t count, float scale,
                         float bias)
{
    const size_t lanes = (count + 3) / 4;
    const bool aligned = ((uintptr_t)in % 4) == 0 && ((uintptr_t)out % 16) == 0;
    if (!aligned) return MBN_EINVAL;
    hipLaunchKernelGGL(normalize_u8_f32, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, s, out, in, count, scale,
                       bias);
    return MBN_OK;
}
