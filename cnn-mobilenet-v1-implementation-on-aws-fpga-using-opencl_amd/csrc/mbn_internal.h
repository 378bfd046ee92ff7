// mbn_internal.h — shared between the C-ABI translation unit and the gfx950 kernel files.
// Not part of the public boundary (that is include/mbn.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <atomic>
#include <map>
#include <mutex>
#include <vector>

#include "mbn.h"

struct mbn_emul_img {
    void *p = nullptr;           // the image (device)
    size_t bytes = 0;            // its size
    size_t src_bytes = 0;        // size of the fp32 filter it was split from (for invalidation by mbn_upload / mbn_memset / mbn_free)
    bool built = false;          // pw_emul_static: the image holds the split of the current filter contents
};

struct mbn_context {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    bool profiling = false;
    bool ev_valid = false;
    uint32_t literal_quirks = MBN_QUIRKS_KERNEL_CL;
    char last_error[256] = {0};
    char name[128] = {0};
    int num_cus = 256;
    std::vector<hipEvent_t> sync_events;         // mbn_stream_wait: reusable fork/join events (round robin)
    size_t sync_next = 0;
    std::vector<hipEvent_t> pool;                // mbn_profile_begin/end: 2 events per recorded call
    int pool_cap = 0, pool_used = 0;
    bool pool_on = false;
    std::vector<hipEvent_t> marks;               // mbn_mark / mbn_marks_read: step markers (reused across reads)
    size_t marks_used = 0;
    void *lit_ws = nullptr;                      // LITERAL pointwise on v_dot4: packed int8 filter + per-channel weight sums + flag
    size_t lit_ws_bytes = 0;
    std::map<std::pair<uintptr_t, int>, mbn_emul_img> emul_ws;   // pw_emul: pre-split filter images, by (filter pointer, layout) (mbn_f32_pw_x6.hip)
    std::mutex mu;
    std::map<uintptr_t, size_t> allocs;          // buffers handed out by mbn_alloc: base address -> bytes (ordered: mbn_span_check
                                                 // finds the allocation that CONTAINS an interior pointer)
};

static inline int mbn_record_hip_error(mbn_context *ctx, hipError_t e, const char *what)
{
    if (ctx) snprintf(ctx->last_error, sizeof(ctx->last_error), "%s: %s", what, hipGetErrorString(e));
    return MBN_EDEVICE;
}

#define MBN_HIP_TRY(ctx, expr)                                                    \
    do {                                                                          \
        hipError_t _e = (expr);                                                   \
        if (_e != hipSuccess) return mbn_record_hip_error((ctx), _e, #expr);      \
    } while (0)

// Process-wide switches (mbn_tune_set). 0 = shipped default everywhere.
// Two kinds. PRODUCT switches select between code paths that every build contains (the opt-in pw_emul arithmetic, the
// split-K and LITERAL dot4 regimes a test must be able to force, the sub-batch stagger, a forced GEMM tile among the compiled
// ones): relaxed atomics, read once per launch on the host. LAB knobs are A/B hooks of the experiments recorded in
// profiles/LOG.md: they exist only in the lab build (make lab -> libmbn_lab.so, -DMBN_LAB); in the shipped library they are
// compile-time zeros, so every branch on them folds away, the template instantiations only they reach are not compiled,
// and mbn_tune_set answers MBN_EUNSUPPORTED for their keys.
#ifdef MBN_LAB
#define MBN_LAB_KNOB(name) std::atomic<int> name{0}
#define MBN_LAB_BUILD 1
#else
#define MBN_LAB_KNOB(name) static constexpr int name = 0
#define MBN_LAB_BUILD 0
#endif
struct mbn_tunables {
    // ---- product switches
    std::atomic<int> pw_tile{0};      // pointwise GEMM: force a tile shape (0 = per-layer rule); a shape this build does not contain -> MBN_EUNSUPPORTED
    std::atomic<int> lit_dot{0};      // LITERAL pointwise: 0 = v_dot4 path where eligible, 1 = always the scalar kernel
    std::atomic<int> pw_splitk{0};    // fp32 pointwise in the few-tile regime: 0 = split-K kernel (mbn_f32_pw_splitk.hip), 1 = always pw_gemm, 2 = split-K wherever eligible
    std::atomic<int> pw_emul{0};      // fp32 pointwise on the bf16 matrix cores from exact three-way operand splits (mbn_f32_pw_x6.hip): 0 = off, 6 or 9 products
    std::atomic<int> pw_emul_static{0}; // pw_emul: 1 = a filter's image is split once and reused until the filter is rewritten through this library
    std::atomic<int> net_stagger{2};  // layers by which consecutive sub-batch streams are staggered (mbn_net_set_streams)
    std::atomic<int> pw_clock{0};     // 1 = pw_gemm launches accumulate their held core clock (mbn_pw_clock_read)
    std::atomic<int> cu_mask{0};      // streams made by mbn_stream_create alternate between two halves of the chip: 1 = XCDs 0-3 / 4-7, 2 = the two halves of every XCD (round 4 experiment)
    // ---- lab knobs
    MBN_LAB_KNOB(dw_variant);         // depthwise kernel variant
    MBN_LAB_KNOB(dw_nseg);            // force row segments per image (0 = heuristic)
    MBN_LAB_KNOB(pw_stage);           // 1 = register staging instead of direct-to-LDS loads
    MBN_LAB_KNOB(conv_variant);       // conv1 / stem / GEMM-epilogue variants
    MBN_LAB_KNOB(misc);
    MBN_LAB_KNOB(pw_ring);            // bf16 pointwise: 0 = ring kernel for K = 64, 1 = always pw_gemm, 2 = ring wherever eligible
    MBN_LAB_KNOB(pw_xn);              // pointwise GEMM: XCD groups along n (0 = by filter size, 1 = off, 2, 4)
    MBN_LAB_KNOB(dwpw_variant);       // fused block kernel: 0 = shipped choice per shape, 1 = round-1 producer/consumer kernel, 2 = unified-wave kernel,
                                      // 3 = unified with the taps read inside the step, 100 + bits = unified with parts switched off (ablation)
    MBN_LAB_KNOB(exp0);               // scratch knobs of the experiment at hand (kernel files say what they mean where they read them)
    MBN_LAB_KNOB(exp1);
    MBN_LAB_KNOB(exp2);
};
extern mbn_tunables g_mbn_tune;

// Resolved per-call view of (positional args + ext) handed to the kernel launchers.
struct mbn_call {
    mbn_context *ctx;
    hipStream_t stream;
    int batch;
    int act;
    int pad_top, pad_left;
    int in_rows, in_cols;
    int cin;
    int g0, g1;
    uint32_t quirks;
    const float *scale, *shift;
    int dtype;        // MBN_DT_F32 or MBN_DT_BF16 for the NHWC launchers (storage type of activations)
    int io_flags;     // MBN_IO_IN_F32 / MBN_IO_OUT_F32 (bf16 mode only), MBN_IO_IN_U8 (convolute)
};

// ---- launchers implemented in the kernel files; each returns MBN_* and launches on c.stream ----
// LITERAL (uint8 planar NCHW / int32) — mbn_literal.hip
int mbn_launch_lit_convolute(const mbn_call &c, uint8_t *out, const uint8_t *r, const uint8_t *g, const uint8_t *b,
                             const int32_t *filt, int rows, int cols, int fs, int stride, int op_size);
int mbn_launch_lit_depthwise(const mbn_call &c, uint8_t *out, const uint8_t *in, const int32_t *filt, int rows,
                             int cols, int fs, int stride, int op_size);
int mbn_launch_lit_pointwise(const mbn_call &c, uint8_t *out, const uint8_t *in, const int32_t *filt, int rows,
                             int cols, int cin, int op_size);
int mbn_launch_lit_pool(const mbn_call &c, uint8_t *out, const uint8_t *in, int rows, int cols, int fs, int op_size);

// NHWC fp32 / bf16 storage (c.dtype), fp32 arithmetic — mbn_f32_dw.hip / mbn_f32_pw.hip / mbn_f32_misc.hip
int mbn_launch_f32_conv(const mbn_call &c, void *out, const void *in, const float *filt, int rows, int cols, int fs,
                        int stride, int op_size);
int mbn_launch_f32_depthwise(const mbn_call &c, void *out, const void *in, const float *filt, int rows, int cols,
                             int fs, int stride, int channels);
int mbn_launch_f32_pointwise(const mbn_call &c, void *out, const void *in, const void *filt, long m, int cin,
                             int op_size);
int mbn_launch_f32_pw_splitk(const mbn_call &c, float *out, const float *in, const float *filt, long m, int cin, int op_size);
int mbn_launch_f32_pw_emul(const mbn_call &c, float *out, const float *in, const float *filt, long m, int cin, int op_size);
int mbn_pw_emul_filter_image(mbn_context *ctx, hipStream_t stream, const float *filt, int n, int k, int bn, int paired,
                             const unsigned **img, unsigned *bytes);
void mbn_pw_emul_invalidate(mbn_context *ctx, const void *dst, size_t bytes);   // caller holds no lock
int mbn_launch_f32_dwpw2_x6(mbn_context *ctx, hipStream_t stream, float *out, const float *in, const float *wd, const float *s2,
                            const float *b2, const float *wp, const float *s3, const float *b3, int batch, int in_rows, int in_cols,
                            int out_rows, int out_cols, int cin, int cout, int stride, int pad_top, int pad_left);
int mbn_launch_bf16_pw_ring(const mbn_call &c, void *out, const void *in, const void *filt, long m, int cin, int op_size);      // lab build only
int mbn_launch_bf16_pw_wide(const mbn_call &c, void *out, const void *in, const void *fpk, long m, int cin, int op_size);
int mbn_launch_pack_filter_bf16(mbn_context *ctx, hipStream_t s, void *dst, const void *src, int n, int k);
int mbn_bf16_pw_wide_eligible(long m, int cin, int op_size);
int mbn_launch_bf16_pw_rf(const mbn_call &c, void *out, const void *in, const void *filt, long m, int cin, int op_size);          // lab build only (round 6): K = 512, filter in registers
int mbn_launch_bf16_pw_stream(const mbn_call &c, void *out, const void *in, const void *filt, long m, int cin, int op_size, bool m16 = false);
int mbn_launch_bf16_pw_big(const mbn_call &c, void *out, const void *in, const void *filt, long m, int cin, int op_size, long *rows_done);
int mbn_launch_f32_pool(const mbn_call &c, void *out, const void *in, int rows, int cols, int fs, int channels);
size_t mbn_pool_fc_ws_bytes(int channels, int classes);
int mbn_launch_f32_pool_fc(mbn_context *ctx, hipStream_t s, float *out, const float *in, const float *w, const float *bias, void *ws,
                           int batch, int pix, int channels, int classes, int k, float *probs, int32_t *topk_idx, float *topk_prob);
// floor(v / d) == umulhi(v, *m) >> *s for every v < 2^31 (d >= 2); d == 1 gives *m = 0 (callers skip the multiply)
static inline void mbn_udiv_magic(unsigned d, unsigned *m, unsigned *s)
{
    if (d <= 1) { *m = 0; *s = 0; return; }
    unsigned l = 0;
    while ((1u << l) < d) l++;                           // l = ceil(log2 d) >= 1
    const unsigned long long p = 1ull << (31 + l);
    *m = (unsigned)((p + d - 1) / d);                    // ceil(2^(31+l) / d) <= 2^32 - 1 for d >= 2 (== 2^31 for powers of two)
    *s = l - 1;
}

int mbn_f32_dwpw_check(const float *out, const float *in, const float *wd, const float *s2, const float *b2,
                       const float *wp, const float *s3, const float *b3, int batch, int in_rows, int in_cols,
                       int out_rows, int out_cols, int cin, int cout, int stride, int pad_top, int pad_left);
int mbn_launch_f32_dwpw(mbn_context *ctx, hipStream_t stream, float *out, const float *in, const float *wd,
                        const float *s2, const float *b2, const float *wp, const float *s3, const float *b3, int batch,
                        int in_rows, int in_cols, int out_rows, int out_cols, int cin, int cout, int stride, int pad_top,
                        int pad_left);
int mbn_launch_f32_dwpw2(mbn_context *ctx, hipStream_t stream, float *out, const float *in, const float *wd,
                         const float *s2, const float *b2, const float *wp, const float *s3, const float *b3, int batch,
                         int in_rows, int in_cols, int out_rows, int out_cols, int cin, int cout, int stride, int pad_top,
                         int pad_left);
// short-K pointwise GEMM with the filter slice resident in LDS (mbn_f32_pw3.hip, round 6). Taken where it measured faster than pw_gemm (profiles/r06/l_*:
// layers 5 / 7 / 9 / 11 at batch 256: -22 / -12 / -12 / -7 %, batch 64: -17 / -8 / -9 / -2 %, batch 16: equal; layer 13 (N = 512, four slices): +2 ... +40 %):
// Cin <= 256, Cout <= 256, at least 128 pixels per CU. Same bits as pw_gemm, so the rule may depend on M.
#define MBN_PW3_DEFAULT(m, cin, n, cus) ((cin) <= 256 && (n) <= 256 && (m) >= 128L * (cus))
int mbn_launch_f32_pw3(const mbn_call &c, float *out, const float *in, const float *filt, long m, int cin, int op_size);
// a run of equal bf16 blocks with the activations resident in LDS (mbn_bf16_res.hip, round 6)
int mbn_bf16_res_eligible(int rows, int cols, int channels, int nblocks);
int mbn_bf16_tail_eligible(int rows, int cols, int c0, int c1);                      // round 6: the last two blocks + the pool in one launch (mbn_bf16_tail.hip)
int mbn_launch_bf16_tail(mbn_context *ctx, hipStream_t stream, void *out, const void *in, const mbn_block_params *blocks, int batch, int rows, int cols, int c0, int c1);
int mbn_launch_bf16_res_blocks(mbn_context *ctx, hipStream_t stream, void *out, const void *in, const mbn_block_params *blocks, int nblocks, int batch,
                               int rows, int cols, int channels);
// wave-private form of the fp32 block (mbn_f32_dwpw3.hip, round 6): the default for stride-1 blocks with Cin >= 128 (blocks 6-7 and 10-11 of the
// 1.0x network: -9...-12 % and -4...-6 % against dwpw2 in alternating runs, profiles/r06/f_*; equal on 8-9, 8 % slower on 4-5: those stay on dwpw2)
#define MBN_DWPW3_DEFAULT(stride, cin) ((stride) == 1 && (cin) >= 128)
int mbn_f32_dwpw3_eligible(const mbn_context *ctx, int batch, int in_rows, int in_cols, int out_rows, int out_cols, int cin, int cout, int stride,
                           int pad_top, int pad_left);
int mbn_launch_f32_dwpw3(mbn_context *ctx, hipStream_t stream, float *out, const float *in, const float *wd,
                         const float *s2, const float *b2, const float *wp, const float *s3, const float *b3, int batch,
                         int in_rows, int in_cols, int out_rows, int out_cols, int cin, int cout, int stride, int pad_top,
                         int pad_left);
int mbn_bf16_dwpw_check(const void *out, const void *in, const float *wd, const float *s2, const float *b2, const void *wp,
                        const float *s3, const float *b3, int batch, int in_rows, int in_cols, int out_rows, int out_cols,
                        int cin, int cout, int stride, int pad_top, int pad_left);
int mbn_launch_bf16_dwpw(mbn_context *ctx, hipStream_t stream, void *out, const void *in, const float *wd, const float *s2,
                         const float *b2, const void *wp, const float *s3, const float *b3, int batch, int in_rows,
                         int in_cols, int out_rows, int out_cols, int cin, int cout, int stride, int pad_top, int pad_left);
int mbn_launch_bf16_dwpw2(mbn_context *ctx, hipStream_t stream, void *out, const void *in, const float *wd, const float *s2,
                          const float *b2, const void *wp, const float *s3, const float *b3, int batch, int in_rows,
                          int in_cols, int out_rows, int out_cols, int cin, int cout, int stride, int pad_top, int pad_left);
int mbn_launch_f32_stem(mbn_context *ctx, hipStream_t stream, float *out, const float *in, const float *w1,
                        const float *s1, const float *b1, const float *wd, const float *s2, const float *b2,
                        const float *wp, const float *s3, const float *b3, int batch, int res, int c1, int c3, int in_u8, int bf16);
int mbn_launch_convert(mbn_context *ctx, hipStream_t s, void *dst, const void *src, size_t count, int to_bf16);
int mbn_launch_f32_softmax(mbn_context *ctx, hipStream_t s, float *probs, int32_t *argmax, const float *logits,
                           int batch, int classes);
int mbn_launch_f32_softmax_topk(mbn_context *ctx, hipStream_t s, float *probs, int32_t *topk_idx, float *topk_prob,
                                const float *logits, int batch, int classes, int k);
int mbn_launch_normalize(mbn_context *ctx, hipStream_t s, float *out, const uint8_t *in, size_t count, float scale,
                         float bias);

static inline int mbn_same_pad(int in, int out, int k, int stride)
{
    int total = (out - 1) * stride + k - in;
    if (total < 0) total = 0;
    return total / 2;
}
