/*
 * mbn.h — C-ABI of the MI355X-native MobileNet-V1 hot path.
 *
 * Drop-in boundary for the OpenCL "kernel-by-name + positional args" protocol of
 * the reference (anerisheth19/CNN-MobileNet-V1-implementation-on-AWS-FPGA-using-OpenCL):
 * every layer entry point below keeps the NAME and the LEADING POSITIONAL
 * PARAMETERS of one `__kernel` in the reference's kernel.cl, extended by one
 * trailing `const mbn_layer_ext*` that may be NULL.
 *
 *   reference interface replaced                         entry point here
 *   ---------------------------------------------------  -------------------------
 *   kernel.cl:2-3   __kernel convolute(...)              mbn_convolute
 *   kernel.cl:62    __kernel depthwise(...)              mbn_depthwise
 *   kernel.cl:94    __kernel pointwise(...) (also FC,    mbn_pointwise
 *                   MobileNet.c:2681-2763)
 *   kernel.cl:116   __kernel pool(...)                   mbn_pool
 *   MobileNet.c:145-213  CL platform/ctx/queue/program   mbn_init
 *   MobileNet.c:2809-2831 clRelease*                     mbn_shutdown
 *   MobileNet.c:340-342  clCreateBuffer                  mbn_alloc / mbn_free
 *   MobileNet.c:350-351  clEnqueueWriteBuffer            mbn_upload
 *   MobileNet.c:395      clEnqueueReadBuffer             mbn_download
 *   MobileNet.c:390-391  clWaitForEvents + clFinish      mbn_sync
 *   MobileNet.c:301-305  clGetEventProfilingInfo         mbn_last_kernel_ms
 *   MobileNet.c:31-47    readSquezeNetKernel             readSquezeNetKernel (kept) / mbn_read_text_weights
 *   MobileNet.c:49-57    decode_image                    decode_image (kept) / mbn_read_ppm
 *   MobileNet.c:218-238  RGB de-interleave               mbn_split_rgb
 *   MobileNet.c:2771-2792 softmax + argmax               mbn_softmax_argmax_u8 / mbn_softmax_f32
 *   keras.py:1-8 (role only: Keras .h5 -> host weights)  mbn_h5_open / mbn_h5_get / mbn_weights_from_h5
 *   MobileNet.c:13-26 + per-layer literals (topology)    mbn_plan_build
 *   MobileNet.c:240-2763 (29 hand-unrolled layer blocks) mbn_net_create / mbn_net_forward
 *
 * Conventions:
 *   - plain C, plain pointers and sizes; no C++/torch types cross this boundary;
 *   - every function returns int: 0 = MBN_OK, <0 = MBN_E*; nothing exits/aborts;
 *   - device memory is addressed by raw device pointers (void*) so a caller may
 *     also pass memory it allocated itself (hipMalloc, a torch tensor's data_ptr);
 *   - layer calls are asynchronous on the context's HIP stream (or ext->stream),
 *     ordered; mbn_sync() waits. One context per GPU; a context is not re-entrant.
 *
 * Two arithmetic modes:
 *   LITERAL (ext == NULL or ext->dtype == MBN_DT_U8): the integer semantics of
 *     kernel.cl — uint8 activations, planar NCHW, int32 filters `[oc][ic][ky][kx]`,
 *     int32 accumulate, ReLU, truncating int->uchar store. Bit-exact vs oracle/.
 *   F32 (ext->dtype == MBN_DT_F32): what BASELINE.json's metric measures — fp32,
 *     NHWC, conv -> per-channel scale/shift (folded BatchNorm) -> ReLU/ReLU6,
 *     TF-"SAME" padding. Tolerances: tests/test_parity_gpu.py.
 */
#ifndef MBN_H
#define MBN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- status */
#define MBN_OK            0
#define MBN_EINVAL       -1   /* bad argument (NULL pointer, non-positive size, unsupported combo) */
#define MBN_ENOMEM       -2   /* host or device allocation failed */
#define MBN_EDEVICE      -3   /* HIP runtime error (see mbn_last_device_error) */
#define MBN_EIO          -4   /* file missing / short read */
#define MBN_EFORMAT      -5   /* file is not what it claims (HDF5 / PPM / text weights) */
#define MBN_ENOTFOUND    -6   /* named object not in the .h5 */
#define MBN_ESHAPE       -7   /* dataset shape does not match the layer table */
#define MBN_EUNSUPPORTED -8   /* valid request this build does not implement */
#define MBN_ENODEVICE    -9   /* no HIP device / ordinal out of range */

const char *mbn_strerror(int code);

/* ------------------------------------------------------------- enumerations */
enum { MBN_DT_U8 = 0,   /* LITERAL: uint8 act / int32 filter / int32 acc (kernel.cl) */
       MBN_DT_F32 = 1,  /* fp32 act / fp32 filter / fp32 acc                         */
       MBN_DT_BF16 = 2  /* bf16 act / bf16 pointwise filter / fp32 acc (config 5)    */ };

enum { MBN_LAYOUT_NCHW_PLANAR = 0,  /* plane c at offset c*rows*cols (kernel.cl:73,107) */
       MBN_LAYOUT_NHWC = 1 };

enum { MBN_ACT_NONE = 0, MBN_ACT_RELU = 1 /* kernel.cl:87-89 */, MBN_ACT_RELU6 = 2 /* Keras */ };

/* LITERAL-mode switches; each reproduces one well-defined behaviour of kernel.cl
 * (SURVEY.md Appendix B). ext == NULL uses the context default (mbn_set_literal_quirks;
 * initial default MBN_QUIRKS_KERNEL_CL = what kernel.cl computes wherever it is in bounds). */
#define MBN_Q_CARRY_SUM      0x1u /* B1: `sum` is not reset between output channels (kernel.cl:10,69,99,121) */
#define MBN_Q_DW_PLANE0      0x2u /* B2: depthwise reads input plane 0 for every channel (kernel.cl:83) */
#define MBN_Q_LITERAL_INDEX  0x4u /* kernel.cl:24,83 index expression verbatim:
                                     in[(ty+i)*G0*stride + (tx+j)*stride] over the WHOLE input buffer
                                     (a column past the row end wraps into the next row, a row past the
                                     plane end lands in the next plane); an index past the end of the input
                                     buffer reads 0 (the reference would read out of bounds there: B4/B5) */
#define MBN_Q_POOL_DIV49     0x8u /* kernel.cl:129 divides by the literal 49 whatever filtersize is */
#define MBN_QUIRKS_NONE      0x0u
#define MBN_QUIRKS_KERNEL_CL (MBN_Q_CARRY_SUM | MBN_Q_DW_PLANE0 | MBN_Q_LITERAL_INDEX | MBN_Q_POOL_DIV49)

/* Trailing extension of every layer call. Zero-initialise, set struct_size = sizeof(mbn_layer_ext). */
typedef struct mbn_layer_ext {
    uint32_t struct_size;
    int32_t  batch;        /* N images; 0 or 1 => one image (the reference processes one: MobileNet.c:215) */
    int32_t  dtype;        /* MBN_DT_* */
    int32_t  layout;       /* MBN_LAYOUT_*; F32 requires NHWC, U8 requires NCHW_PLANAR */
    int32_t  act;          /* MBN_ACT_*; LITERAL ignores it (always ReLU, kernel.cl:87) */
    int32_t  pad_top;      /* F32: rows of zero padding above; -1 = TF "SAME" (computed from in/out size) */
    int32_t  pad_left;     /* F32: same for columns. LITERAL always pads filtersize/2 top/left (kernel.cl:79) */
    int32_t  in_rows;      /* depthwise: input plane size; 0 => rows*stride. convolute: ignored */
    int32_t  in_cols;
    int32_t  cin;          /* convolute F32: input channels of the NHWC image (0 => 3) */
    int32_t  gsize0;       /* LITERAL: NDRange of the emulated launch, x then y (get_global_size); 0 => output size */
    int32_t  gsize1;
    uint32_t quirks;       /* LITERAL: MBN_Q_* bitmask. Honoured only when quirks_valid != 0 */
    int32_t  quirks_valid;
    const void *scale;     /* F32: device ptr, op_size floats, multiplies the conv sum (folded BN); NULL => 1 */
    const void *shift;     /* F32: device ptr, op_size floats, added after scale (folded BN / FC bias); NULL => 0 */
    void    *stream;       /* hipStream_t; NULL => the context's stream */
    int32_t  io_flags;     /* MBN_DT_BF16 only: MBN_IO_* (which side of the call is fp32 instead of bf16) */
    int32_t  reserved;
} mbn_layer_ext;

/* bf16 mode (BASELINE config 5): activations bf16 NHWC in HBM, pointwise/FC filters bf16 [Cout][Cin], everything
 * else (conv1/depthwise filters, scale/shift, accumulation) fp32. */
#define MBN_IO_IN_F32   0x1   /* the input tensor is fp32 (convolute: the normalised image) */
#define MBN_IO_OUT_F32  0x2   /* the output tensor is fp32 (pointwise as FC: the logits) */
#define MBN_IO_FILT_PACKED 0x8 /* pointwise, MBN_DT_BF16: `filter_k` holds the plain [Cout][Cin] bf16 filter FOLLOWED, at byte offset
                                * mbn_packed_filter_offset(Cout, Cin), by its packed image (mbn_pack_filter_bf16): the wide-layer GEMM
                                * (mbn_bf16_pw_wide.hip) loads the filter in matrix-operand order straight into registers */
#define MBN_IO_IN_U8    0x4   /* convolute, fp32/bf16 mode: the input is the raw uint8 HWC image (what decode_image /
                               * mbn_read_ppm produce, MobileNet.c:49-57); the Keras MobileNet preprocessing x/127.5 - 1
                               * is applied at load — SURVEY §8f-2, replaces a separate mbn_normalize_u8_to_f32 pass */

typedef struct mbn_context mbn_context;   /* opaque; one per GPU */

/* ----------------------------------------------------------------- lifecycle */
/* Replaces MobileNet.c:145-213 (platform, device, context, queue with profiling, program build). */
int  mbn_init(int device_ordinal, mbn_context **ctx);
/* Replaces MobileNet.c:2809-2831. Frees every buffer still owned by the context. NULL is a no-op. */
int  mbn_shutdown(mbn_context *ctx);
int  mbn_device_count(int *count);                       /* 0 devices => *count = 0, MBN_OK */
int  mbn_device_name(mbn_context *ctx, char *buf, size_t buflen);
int  mbn_device_cus(mbn_context *ctx, int *count);       /* compute units of the context's GPU (256 on MI355X): the counterpart of
                                                          * CL_DEVICE_MAX_COMPUTE_UNITS; the net runner sizes its small-batch choices by it */
int  mbn_device_pci_bus_id(mbn_context *ctx, char *buf, size_t buflen);   /* "0000:05:00.0": which physical GPU this context holds. The
                                                          * reference picks device 0 of platform 0 (MobileNet.c:155); with one context
                                                          * per rank the multi-GPU bench line proves its ranks held N cards */
const char *mbn_last_device_error(mbn_context *ctx);     /* text of the last HIP error seen by this context */
int  mbn_set_literal_quirks(mbn_context *ctx, uint32_t quirks);
int  mbn_get_stream(mbn_context *ctx, void **stream);    /* the context's hipStream_t */
/* Extra streams of the context's device, for overlapping independent work (ext->stream selects one per call).
 * mbn_stream_wait(ctx, waiter, signaler): everything queued on `waiter` after this call runs after everything queued on
 * `signaler` before it (an event record + stream wait, no host sync); NULL names the context's own stream. */
int  mbn_stream_create(mbn_context *ctx, void **stream);
int  mbn_stream_destroy(mbn_context *ctx, void *stream);
int  mbn_stream_wait(mbn_context *ctx, void *waiter, void *signaler);
/* hipGraph capture of a sequence of layer calls (launch-bound small-batch loops): everything queued on `stream`
 * (NULL = the context's stream) between begin and end is recorded instead of executed; mbn_graph_launch replays it as
 * one submission. Blocking calls (mbn_upload/download/sync/alloc/free) must not be made on that stream while capturing;
 * the per-call profiling events are skipped inside a capture. */
int  mbn_graph_begin(mbn_context *ctx, void *stream);
int  mbn_graph_end(mbn_context *ctx, void *stream, void **graph_exec);
int  mbn_graph_launch(mbn_context *ctx, void *graph_exec, void *stream);
int  mbn_graph_destroy(mbn_context *ctx, void *graph_exec);

/* ---------------------------------------------------- buffers (clCreateBuffer &c.) */
/* Every buffer handed out by mbn_alloc is remembered with its size. When a pointer passed to a layer call, a fused call,
 * the classifier calls or mbn_upload/download/memset lies inside one of them (interior pointers included), the bytes the
 * call will touch — computed from its shape arguments, ext->batch and dtype — must fit in the rest of that buffer;
 * otherwise the call returns MBN_EINVAL and launches nothing (mbn_last_device_error names the operand). A uint8 image
 * handed to the fp32 first layer, a logits buffer sized for a smaller batch, a filter shorter than op_size x filtersize
 * are caller errors here, not GPU memory faults. Memory the caller allocated itself is not tracked and not checked. */
int  mbn_alloc(mbn_context *ctx, size_t bytes, void **dptr);                 /* MobileNet.c:340-342 */
int  mbn_free(mbn_context *ctx, void *dptr);
int  mbn_upload(mbn_context *ctx, void *dst_dev, const void *src_host, size_t bytes);   /* blocking, :350 */
int  mbn_download(mbn_context *ctx, void *dst_host, const void *src_dev, size_t bytes); /* blocking, :395 */
int  mbn_memset(mbn_context *ctx, void *dst_dev, int byte, size_t bytes);    /* async on ctx stream */
/* Tell the library that `bytes` of device memory at `dptr` which the CALLER owns are about to be freed or were rewritten
 * behind its back: every library-owned image derived from that range (the pre-split filter images of the opt-in pw_emul
 * form) is released after the device has gone idle. mbn_free does the same for the buffers it hands out; mbn_net_destroy
 * calls this for a caller-provided parameter blob. */
int  mbn_forget(mbn_context *ctx, const void *dptr, size_t bytes);
int  mbn_sync(mbn_context *ctx);                                             /* :390-391 */
/* Time of the most recent layer call in milliseconds (hipEvent pair recorded around its launch on the
 * stream it ran on; waits for it). Replaces clGetEventProfilingInfo START/END (MobileNet.c:301-305). */
int  mbn_last_kernel_ms(mbn_context *ctx, float *ms);
int  mbn_set_profiling(mbn_context *ctx, int enabled);   /* default 0: no events recorded */
/* Event pool for measuring every layer call of a timed region without synchronising inside it:
 * after mbn_profile_begin each layer call records a hipEvent pair on the stream it is launched on into the
 * next free slot (calls beyond `capacity` are not recorded); mbn_profile_end waits for the stream and returns
 * the per-call milliseconds in call order. */
int  mbn_profile_begin(mbn_context *ctx, int capacity);
int  mbn_profile_end(mbn_context *ctx, float *ms, int ms_capacity, int *count);
int  mbn_profile_pause(mbn_context *ctx, int paused);   /* 1: stop recording (slots keep their order), 0: resume */
/* Calibration of the pool: records one event pair exactly as a layer call does, around nothing (with_kernel = 0) or around
 * an empty one-wave kernel (with_kernel = 1), on `stream` (NULL = the context's stream). What such a pair reads is what
 * the pair itself adds to every recorded call (bench.py reports it as `event_overhead_us` and subtracts it, so that short
 * kernels agree with a rocprofv3 kernel trace). */
int  mbn_profile_null(mbn_context *ctx, int with_kernel, void *stream);
/* Step markers: mbn_mark queues one timing event on `stream` (NULL = the context's stream) without synchronising;
 * mbn_marks_read waits for the last one and returns the milliseconds between consecutive marks, in order
 * (count = marks - 1), then forgets them. bench.py brackets every timed step with one mark to report the median and
 * the p10/p90 of the per-step time next to the wall-clock figure. */
int  mbn_mark(mbn_context *ctx, void *stream);
int  mbn_marks_read(mbn_context *ctx, float *ms_between, int capacity, int *count);

/* --------------------------------------------------------------- layer calls
 * Positional parameters are kernel.cl's, in kernel.cl's order and meaning:
 *
 * convolute (kernel.cl:2-3): first 3x3xCin conv.
 *   rows, cols  = INPUT plane size (224); output plane = (rows/stride) x (cols/stride)
 *                 (kernel.cl:14 hard-codes rows/2 * cols/2 as the output plane stride;
 *                 F32 uses ceil division)
 *   LITERAL: inp_r/g/b three uint8 planes; filter int32 [op_size][3][ky][kx]
 *   F32    : inp_r = NHWC image [N][rows][cols][cin]; inp_g/inp_b ignored;
 *            filter fp32 [ky][kx][cin][op_size] (Keras HWIO)
 */
int mbn_convolute(mbn_context *ctx, void *output, const void *inp_image_r, const void *inp_image_g,
                  const void *inp_image_b, const void *filter_k, int rows, int cols, int filtersize,
                  int stride, int op_size, const mbn_layer_ext *ext);

/* depthwise (kernel.cl:62): per-channel 3x3, stride 1 or 2.
 *   rows, cols = OUTPUT plane size (the only use kernel.cl makes of them: output_shift, :73);
 *                input plane = ext->in_rows x in_cols, default rows*stride x cols*stride
 *   op_size    = channels
 *   LITERAL: filter int32 [op_size][ky][kx];  F32: filter fp32 [ky][kx][op_size]
 */
int mbn_depthwise(mbn_context *ctx, void *output, const void *inp_image, const void *filter_k,
                  int rows, int cols, int filtersize, int stride, int op_size, const mbn_layer_ext *ext);

/* pointwise (kernel.cl:94): 1x1 conv = per-pixel [op_size x filtersize] mat-vec; FC when rows=cols=1.
 *   filtersize = number of INPUT channels (loop bound kernel.cl:106, plane stride :107)
 *   filter     = [op_size][filtersize] in both modes (int32 / fp32)
 *   F32 summation order over the input channels: sequential (pw_gemm) for every call of 5 images and more; a call of
 *   ext->batch <= 4 images of a layer with few output tiles (its 4-image form has fewer 64x64 tiles than half the compute
 *   units; K >= 128, K % 64 == 0) runs the split-K kernel: K in 4 / 8 / 16 slices (by K alone) summed in fixed order.
 *   Either way a result depends on the layer and on "1..4 images or more", never on the batch-mates or the slot.
 */
int mbn_pointwise(mbn_context *ctx, void *output, const void *inp_image, const void *filter_k,
                  int rows, int cols, int filtersize, int op_size, const mbn_layer_ext *ext);

/* pool (kernel.cl:116): global filtersize x filtersize average per channel -> [op_size].
 *   rows, cols = input plane size. LITERAL: integer division (by 49 under MBN_Q_POOL_DIV49).
 */
int mbn_pool(mbn_context *ctx, void *output, const void *inp_image, int rows, int cols, int filtersize,
             int op_size, const mbn_layer_ext *ext);

/* Fused stem (SURVEY §8f-1 applied to the first block): layers 1-3 of the sequence — convolute 3x3x3 stride 2 ->
 * depthwise 3x3 stride 1 -> pointwise 1x1, each followed by its folded-BN scale/shift and ReLU6 — in one kernel, so
 * the two 112x112x32 intermediates never reach HBM (MobileNet.c:240-498 does three launches and six PCIe copies).
 * fp32 NHWC only; image [batch][res][res][3], out [batch][res/2][res/2][c3]; filters in the layouts of the separate
 * calls (w1 [3][3][3][c1], wd [3][3][c1], wp [c3][c1]). Returns MBN_EUNSUPPORTED unless (c1, c3) = (32, 64) (alpha = 1) or
 * (16, 32) (alpha = 0.5) and res is a multiple of 32 — callers then issue the three layer calls instead. */
int mbn_stem_fused(mbn_context *ctx, void *out, const void *image, const void *w1, const void *s1, const void *b1,
                   const void *wd, const void *s2, const void *b2, const void *wp, const void *s3, const void *b3,
                   int batch, int res, int c1, int c3, void *stream);
/* The same with the raw uint8 HWC image [batch][res][res][3] as input: x/127.5 - 1 is applied while the
 * input patch is loaded, bit-identical to mbn_normalize_u8_to_f32 followed by mbn_stem_fused. */
int mbn_stem_fused_u8(mbn_context *ctx, void *out, const void *image_u8, const void *w1, const void *s1, const void *b1,
                      const void *wd, const void *s2, const void *b2, const void *wp, const void *s3, const void *b3,
                      int batch, int res, int c1, int c3, void *stream);

/* General form: flags = MBN_STEM_IN_U8 (image is raw uint8 HWC) | MBN_STEM_BF16 (the network's bf16 mode, BASELINE
 * config 5: `out` is bf16, `wp` is the bf16 copy of the pointwise filter, both on-chip intermediates are rounded to bf16
 * where the separate launches would store them; everything else — image unless IN_U8, the other filters, scale/shift,
 * arithmetic — stays fp32). */
#define MBN_STEM_IN_U8  0x1
#define MBN_STEM_BF16   0x2
int mbn_stem_fused_ex(mbn_context *ctx, void *out, const void *image, const void *w1, const void *s1, const void *b1,
                      const void *wd, const void *s2, const void *b2, const void *wp, const void *s3, const void *b3,
                      int batch, int res, int c1, int c3, int flags, void *stream);

/* Fused block: a depthwise 3x3 (stride 1 or 2) + pointwise 1x1 pair of the sequence (the pairs L4-5 ...
 * L26-27, MobileNet.c:322-2599; kernel.cl:62-92 + 94-114) in one kernel, each stage followed by its folded-BN
 * scale/shift and ReLU6; the depthwise output never reaches HBM. fp32 NHWC: in [batch][in_rows][in_cols][cin],
 * out [batch][out_rows][out_cols][cout]; wd [3][3][cin], wp [cout][cin] as in the separate calls; pad_top/pad_left as in
 * mbn_layer_ext (zero padding; the high side needs none stated). Bit-identical to mbn_depthwise followed by
 * mbn_pointwise. Returns MBN_EUNSUPPORTED unless cin is a multiple of 32 and <= 1024, cout a multiple of 128 and <= 1024, out_cols
 * even and the input under 3.75 GiB — callers then issue the two layer calls instead. Three kernels sit behind it, all with the same bits
 * (round 6: stride 1 with cin 128 / 256 runs on the wave-private form mbn_f32_dwpw3.hip, everything else on mbn_f32_dwpw2.hip / mbn_f32_dwpw.hip). */
int mbn_dwpw_fused(mbn_context *ctx, void *out, const void *in, const void *wd, const void *s2, const void *b2,
                   const void *wp, const void *s3, const void *b3, int batch, int in_rows, int in_cols, int out_rows,
                   int out_cols, int cin, int cout, int stride, int pad_top, int pad_left, void *stream);

/* A RUN of equal fused blocks with the activations resident on chip (round 6; bf16 mode): `nblocks` consecutive depthwise 3x3 (stride 1, zero
 * padding 1) + pointwise 1x1 pairs of the sequence (MobileNet.c:322-2599; kernel.cl:62-92 + 94-114) with `channels` channels in and out on a
 * rows x cols map, each stage followed by its folded-BN scale/shift and ReLU6, in ONE launch: an image's map stays in LDS from the first block's input to
 * the last block's output, only the filters come from memory. in / out: bf16 NHWC [batch][rows][cols][channels]; per block: wd [3][3][channels],
 * s2, b2 [channels] fp32, wp_bf16 [channels][channels] bf16, s3, b3 [channels] fp32, as in mbn_dwpw_fused_bf16. Same arithmetic as the separate bf16
 * launches within the bf16 tolerance. Returns MBN_EUNSUPPORTED unless channels == 256, rows * cols <= 104, (rows + 2) * (cols + 2) <= 144 and
 * 1 <= nblocks <= 8 (the five 256 -> 256 blocks on the 10 x 10 map of the 0.5x160 network) — callers then issue the blocks one by one. */
typedef struct mbn_block_params {
    const void *wd, *s2, *b2;       /* depthwise filter, scale, shift (fp32) */
    const void *wp_bf16;            /* pointwise filter [cout][cin], bf16 */
    const void *s3, *b3;            /* pointwise scale, shift (fp32) */
} mbn_block_params;
int mbn_blocks_resident_bf16(mbn_context *ctx, void *out, const void *in, const mbn_block_params *blocks, int nblocks, int batch,
                             int rows, int cols, int channels, void *stream);

/* The network's last two blocks and the global average pool in ONE launch with an image's maps resident on chip (round 6; bf16 mode): depthwise 3x3
 * stride 2 on c0 channels -> pointwise c0 -> c1 -> depthwise 3x3 stride 1 -> pointwise c1 -> c1 -> average over the whole map (layers 24-28 of the
 * sequence: MobileNet.c:322-2599 pairs + the `pool` launch :2601-2679; kernel.cl:62-92, 94-114, 116-132), every stage with its folded-BN scale / shift
 * and ReLU6 and rounded to bf16 as the separate launches store it. in: bf16 NHWC [batch][rows][cols][c0]; out: bf16 [batch][c1] (what mbn_pool writes:
 * the FC layer's input); blocks[0] / blocks[1]: the parameters of the two blocks as in mbn_dwpw_fused_bf16 (blocks[0].wp_bf16 is [c1][c0]).
 * Same arithmetic as the five separate bf16 launches within the bf16 tolerance (depthwise sums and the pool's sum bit for bit; the pointwise sums
 * group their products by 16 instead of 32). Returns MBN_EUNSUPPORTED unless c0 == 256, c1 == 512 and rows, cols are even and <= 10 (the 0.5x160
 * network: 10 x 10 x 256 -> 5 x 5 x 512 -> 512) — callers then issue the layers one by one. */
int mbn_tail_resident_bf16(mbn_context *ctx, void *out, const void *in, const mbn_block_params *blocks, int batch, int rows, int cols, int c0, int c1,
                           void *stream);

/* Classifier tail on device, fp32 (replaces the host loop MobileNet.c:2771-2792):
 * probs[n][k] = softmax(logits[n][:]) and argmax[n] (0-based). probs or argmax may be NULL. */
int mbn_softmax_f32(mbn_context *ctx, void *probs, void *argmax_i32, const void *logits, int batch,
                    int classes, void *stream);

/* The same block in the network's bf16 mode (BASELINE config 5): in/out are bf16 NHWC, wp_bf16 is the bf16 copy of the
 * pointwise filter [cout][cin]; the depthwise filter and all scale/shift vectors stay fp32, arithmetic is fp32, the
 * depthwise output is rounded to bf16 where the separate launch would store it. cin must be a multiple of 64 or 32 itself (half a chunk, padded), cout a multiple
 * of 64 (round 5; a 64-column remainder — block 6-7 of the 0.5x network: 64 -> 64 channels — runs on a 128-column tile whose
 * upper half multiplies zeros and stores nothing). */
int mbn_dwpw_fused_bf16(mbn_context *ctx, void *out, const void *in, const void *wd, const void *s2, const void *b2,
                        const void *wp_bf16, const void *s3, const void *b3, int batch, int in_rows, int in_cols, int out_rows,
                        int out_cols, int cin, int cout, int stride, int pad_top, int pad_left, void *stream);

/* softmax + top-k on device: for every image the k <= 8 most probable classes, most probable first
 * (ties -> lowest index), topk_idx [batch][k] int32 (0-based, -1 when classes < k) and topk_prob [batch][k] fp32;
 * probs (may be NULL) additionally receives the full [batch][classes] distribution. Only 2k values per image have to
 * cross PCIe instead of the 1000 logits of MobileNet.c:2744 + the host exp loop :2771-2792. */
int mbn_softmax_topk_f32(mbn_context *ctx, void *probs, void *topk_idx_i32, void *topk_prob_f32, const void *logits,
                         int batch, int classes, int k, void *stream);
/* The classifier tail of the sequence as one call (MobileNet.c:2601-2792): global average pool (kernel.cl:116) ->
 * FC = pointwise with rows = cols = 1 (kernel.cl:94; bias, no ReLU: B15) -> softmax + top-k. fp32 NHWC,
 * in [batch][rows][cols][channels], fc_w [classes][channels], fc_bias [classes] or NULL; pooled_scratch
 * [batch][channels] and logits_scratch [batch][classes] are caller-provided device buffers (the logits stay readable
 * there). Three launches on `stream`. */
int mbn_classifier_tail(mbn_context *ctx, void *topk_idx_i32, void *topk_prob_f32, void *probs, void *logits_scratch,
                        void *pooled_scratch, const void *in, const void *fc_w, const void *fc_bias, int batch, int rows,
                        int cols, int channels, int classes, int k, void *stream);
/* Global average pool + FC as ONE launch for 1...4 images (MobileNet.c:2601-2739: the `pool` launch, kernel.cl:116, and the
 * `pointwise` launch with rows = cols = 1, kernel.cl:94; bias, no ReLU). fp32 NHWC in [batch][rows][cols][channels] (the window
 * is the whole map), fc_w [classes][channels], fc_bias [classes] or NULL, logits [batch][classes]. The pooled values are those of
 * mbn_pool bit for bit; the FC sums 64-channel slices in a fixed order (no float atomics): the result of an image does not depend
 * on the batch it is in. `workspace`: mbn_pool_fc_workspace_bytes(channels, classes) bytes of device memory, ZEROED ONCE by the
 * caller (mbn_memset) and then reused launch after launch on one stream at a time. MBN_EUNSUPPORTED for batch > 4 or channels
 * not a multiple of 64 (the caller then uses mbn_pool + mbn_pointwise). */
size_t mbn_pool_fc_workspace_bytes(int channels, int classes);
int mbn_pool_fc(mbn_context *ctx, void *logits, const void *in, const void *fc_w, const void *fc_bias, int batch, int rows, int cols,
                int channels, int classes, void *workspace, size_t workspace_bytes, void *stream);
/* The WHOLE classifier tail as ONE launch for 1...4 images (SURVEY 8f-3 as written; MobileNet.c:2601-2792: the `pool` launch, the FC
 * `pointwise` launch with its blocking read-back of 1000 logits, the host softmax loop :2771-2779 and the host arg-max :2781-2792): global
 * average pool -> FC + bias -> softmax -> top-k, results identical to mbn_pool_fc followed by mbn_softmax_topk_f32 bit for bit. The class
 * range that finishes last (a second arrival counter in the workspace) reads the complete logits back and normalises / ranks them. Same
 * workspace and envelope as mbn_pool_fc, plus classes <= 1024; `logits` [batch][classes] stay readable, `probs` may be NULL.
 * Measured on MI355X (profiles/r04/h_classifier_tail_one_launch.txt): see there — at 1...4 images the tail is launch-latency bound and the
 * one launch is offered as an option, not taken by default. */
int mbn_classifier_tail_fused(mbn_context *ctx, void *topk_idx_i32, void *topk_prob_f32, void *probs, void *logits, const void *in, const void *fc_w,
                              const void *fc_bias, int batch, int rows, int cols, int channels, int classes, int k, void *workspace,
                              size_t workspace_bytes, void *stream);

/* Input front-end on device: uint8 HWC [N][rows][cols][3] -> fp32 NHWC x*scale+bias
 * (Keras MobileNet preprocessing is scale=1/127.5, bias=-1). */
int mbn_normalize_u8_to_f32(mbn_context *ctx, void *out_f32, const void *in_u8, size_t count, float scale,
                            float bias, void *stream);

/* fp32 <-> bf16 (round to nearest even) on device; used to build the bf16 copy of the pointwise/FC filters. */
int mbn_convert_f32_to_bf16(mbn_context *ctx, void *dst_bf16, const void *src_f32, size_t count, void *stream);
int mbn_convert_bf16_to_f32(mbn_context *ctx, void *dst_f32, const void *src_bf16, size_t count, void *stream);
/* LAB BUILD ONLY as far as any kernel reads it (mbn_bf16_pw_wide.hip: 196 x 256 tiles with the filter loaded from this image straight
 * into matrix-operand registers — built and measured in round 3, slower than the tiled GEMM, not shipped; profiles/LOG.md). The shipped
 * library accepts MBN_IO_FILT_PACKED and ignores the image; mbn_pack_filter_bf16 answers MBN_EUNSUPPORTED there.
 * Packed image of a bf16 pointwise filter [cout][cin]: the same cout * cin bf16 values in the order
 * [cout / 256][cin / 64][8 waves][4 k16 steps][64 lanes][8], i.e. the B operand of v_mfma_f32_32x32x16_bf16 per 32-column
 * wave, so that a wave's share of a k-tile is four contiguous 1-KB loads. mbn_packed_filter_offset = where the image goes
 * inside the filter buffer (cout * cin * 2 bytes rounded up to 256), 0 when the shape has no packed form (cout % 256 or
 * cin % 64 != 0). mbn_pack_filter_bf16 writes the image from the plain filter at the start of `filter_buf` (which must hold
 * offset + cout * cin * 2 bytes). */
size_t mbn_packed_filter_offset(int cout, int cin);
int mbn_pack_filter_bf16(mbn_context *ctx, void *filter_buf, int cout, int cin, void *stream);

/* ------------------------------------------------------------------ loaders
 * The two reference symbols are kept with their exact signatures (CPU only). They return without
 * touching the destination when the file is missing instead of dereferencing NULL
 * (the reference does not check fopen: MobileNet.c:37,52). */
void readSquezeNetKernel(int *m, int read_size);                        /* MobileNet.c:31 — reads "weights_c.txt" */
int  decode_image(unsigned char frame[], char filename[]);              /* MobileNet.c:49 — raw 224*224*3 bytes from offset 0 */

int  mbn_read_text_weights(const char *path, int *m, int read_size);    /* checked form of readSquezeNetKernel */
int  mbn_read_text_weights_f32(const char *path, float *m, size_t read_size, size_t skip);
int  mbn_read_ppm(const char *path, unsigned char *rgb, int *width, int *height, int max_pixels); /* P6, header skipped (B14) */
int  mbn_write_ppm(const char *path, const unsigned char *rgb, int width, int height);
int  mbn_split_rgb(const unsigned char *hwc, int pixels, unsigned char *r, unsigned char *g,
                   unsigned char *b);                                   /* MobileNet.c:218-238 */
/* MobileNet.c:2771-2792: double softmax over uint8 logits + argmax; *location is 1-based like the reference
 * (and defined — 1 — when class 0 wins, which the reference leaves uninitialised: B11). */
int  mbn_softmax_argmax_u8(const unsigned char *logits, int n, double *probs, int *location, double *maximum);

/* ------------------------------------------------------------- Keras .h5 reader */
typedef struct mbn_h5 mbn_h5;   /* opaque: a memory-mapped HDF5 file */
int  mbn_h5_open(const char *path, mbn_h5 **h5);
int  mbn_h5_close(mbn_h5 *h5);
/* Look up a dataset by absolute path ("/conv1/conv1/kernel:0"; the leading "/model_weights" of a full
 * Keras model file is tried too). On success *data points into the mapping (valid until close),
 * shape[0..*ndim-1] is filled (ndim capacity 8). Only contiguous little-endian float32 datasets. */
int  mbn_h5_get(mbn_h5 *h5, const char *name, int *ndim, int64_t shape[8], const float **data);
/* Enumerate every dataset path in the file, depth-first; cb returns non-zero to stop. */
int  mbn_h5_visit(mbn_h5 *h5, int (*cb)(const char *path, int ndim, const int64_t *shape, void *user),
                  void *user);
/* Minimal writer (weight-format tooling): creates a v0-superblock file with old-style
 * groups and contiguous float32 datasets — the subset the reader and libhdf5 both accept. */
typedef struct mbn_h5_writer mbn_h5_writer;
int  mbn_h5_create(const char *path, mbn_h5_writer **w);
int  mbn_h5_put(mbn_h5_writer *w, const char *name, int ndim, const int64_t *shape, const float *data);
int  mbn_h5_finish(mbn_h5_writer *w);   /* writes the file and frees the writer */

/* ------------------------------------------------------ topology (layer table) */
enum { MBN_L_CONV = 1, MBN_L_DW = 2, MBN_L_PW = 3, MBN_L_POOL = 4, MBN_L_FC = 5 };

typedef struct mbn_layer_desc {
    int32_t index;        /* 1-based reference layer number (1 conv, 2..27 dw/pw, 28 pool, 29 FC) */
    int32_t kind;         /* MBN_L_* */
    int32_t in_rows, in_cols, in_ch;
    int32_t out_rows, out_cols, out_ch;
    int32_t stride;
    int32_t pad_top, pad_left;       /* TF-SAME */
    int64_t w_offset;     /* float offset of this layer's filter in the packed blob */
    int64_t w_count;
    int64_t scale_offset; /* float offset of per-channel scale (out_ch floats); -1 = none */
    int64_t shift_offset; /* float offset of per-channel shift / bias; -1 = none */
} mbn_layer_desc;

#define MBN_MAX_LAYERS 32
typedef struct mbn_plan {
    int32_t  n_layers;          /* 29 */
    int32_t  res;               /* input resolution (224, 192, 160, 128 or any multiple of 32) */
    float    alpha;             /* width multiplier */
    int32_t  classes;           /* 1000 */
    int64_t  blob_floats;       /* size of the packed, BN-folded parameter blob */
    int64_t  max_act_floats;    /* largest per-image activation (floats) across layers */
    mbn_layer_desc layer[MBN_MAX_LAYERS];
} mbn_plan;

/* MobileNet.c:13-26 + SURVEY §2.1 table, parameterised: channels = max(8, int(c*alpha)), sizes from res. */
int  mbn_plan_build(float alpha, int res, int classes, mbn_plan *plan);

/* Host-side packed weights: one contiguous fp32 blob (this is what is broadcast over RCCL). */
typedef struct mbn_weights {
    mbn_plan plan;
    float   *blob;        /* plan.blob_floats floats, malloc'd; free with mbn_weights_free */
} mbn_weights;

/* Read a Keras-applications MobileNet-V1 weights file: finds conv1 / conv_dw_i / conv_pw_i / *_bn /
 * conv_preds by name, checks every shape against the plan, folds BatchNorm (eps = 1e-3) into per-channel
 * scale/shift, repacks HWIO -> kernel layouts (pointwise -> [Cout][Cin]). alpha <= 0 => infer from conv1. */
int  mbn_weights_from_h5(const char *path, float alpha, int res, mbn_weights *w);
/* Deterministic synthetic weights: N(0, 2/fan_in) kernels, BN gamma U[.5,1.5], beta N(0,.1),
 * mean N(0,.1), var U[.5,1.5]; written in Keras layout so the same reader path loads them. */
int  mbn_weights_synthetic_h5(const char *path, float alpha, int classes, uint64_t seed);
int  mbn_weights_free(mbn_weights *w);

/* -------------------------------------------- whole-network runner (the C host)
 * Replaces the 29 hand-unrolled per-layer blocks of MobileNet.c:240-2763: same layer order, but
 * activations stay resident in HBM (no per-layer D2H/H2D, MobileNet.c:350,395) and weights are uploaded once. */
typedef struct mbn_net mbn_net;
int  mbn_net_create(mbn_context *ctx, const mbn_weights *w, int max_batch, mbn_net **net);
/* Same, but the packed blob already lives on the device (e.g. received by an RCCL broadcast). The net
 * does not take ownership of dev_blob. */
int  mbn_net_create_from_device_blob(mbn_context *ctx, const mbn_plan *plan, const void *dev_blob,
                                     int max_batch, mbn_net **net);
int  mbn_net_destroy(mbn_net *net);
/* MBN_DT_F32 (default) or MBN_DT_BF16: in bf16 mode the net keeps a bf16 copy of the pointwise/FC filters (made on
 * the device from the fp32 blob), activations are bf16, images stay fp32 [batch][res][res][3], logits stay fp32. */
int  mbn_net_set_dtype(mbn_net *net, int dtype);
/* Pipeline every forward over `n` contiguous sub-batches on n streams (1 <= n <= 8; default 1). The HBM-bound
 * depthwise kernels of one sub-batch then overlap the MFMA-bound pointwise kernels of another and fill their tails
 * (fp32 1.0x224 batch 256 with the round-2 kernels: +4.7 % at n = 2, -5 % at n = 3, -3 % at n = 4 — smaller sub-batches
 * quantise the GEMM grids worse; bf16 batch 512: -3 % at n = 2; profiles/r02/k_streams_ab.txt). Sub-batches of fewer
 * than 5 images are not forked. The call still behaves as ONE asynchronous operation on the
 * context's stream: the sub-streams fork from it and join back into it. */
int  mbn_net_set_streams(mbn_net *net, int n);
/* With n > 1 streams: 1 = the caller guarantees that `images` is not being produced by work still pending on the
 * context's stream (e.g. it was uploaded with the blocking mbn_upload, or is a resident benchmark input), so the
 * sub-streams need not wait for the context's stream at the start of a forward. Consecutive forwards then overlap
 * across the step boundary (sub-batch j of step k+1 only waits for sub-batch j of step k); the join into the
 * context's stream is still queued, so mbn_sync / later work on that stream stays ordered after the results. */
int  mbn_net_set_free_running(mbn_net *net, int enabled);
/* 1 = capture the 29 launches of a forward into a hipGraph the first time a (images, logits, batch, last_layer)
 * combination is seen and replay it afterwards (re-captured when any of them changes). For launch-bound batches:
 * measured 0.285 -> see DESIGN.md ms per forward at batch 1. Ignored with more than one stream or in timed forwards. */
int  mbn_net_set_graph(mbn_net *net, int enabled);
/* 1 (default) = run layers 1-3 through mbn_stem_fused when the plan allows it (fp32, alpha = 1, activations not
 * kept, at least 3 layers requested); 0 = always issue the 29 separate layer calls. mbn_net_fused_layers reports how
 * many leading layers the next forward(batch, last_layer) would fuse (0 or 3). */
int  mbn_net_set_fuse_stem(mbn_net *net, int enabled);
/* 1 = the `images` handed to mbn_net_forward / _timed / _classify are raw uint8 HWC [batch][res][res][3] (device);
 * layer 1 (or the fused stem) normalises them at load (MBN_IO_IN_U8). 0 (default) = fp32 NHWC, already normalised. */
int  mbn_net_set_input_u8(mbn_net *net, int enabled);
int  mbn_net_fused_layers(const mbn_net *net, int last_layer, int *count);
/* Fused depthwise->pointwise blocks (mbn_dwpw_fused): bit L of `mask` (L = 1-based number of a depthwise layer) lets
 * layers L and L+1 run as one launch when the plan is fp32, activations are not kept and the shapes are inside the
 * kernel's envelope. Default (until this is called) = MBN_FUSE_BLOCKS_DEFAULT in fp32, MBN_FUSE_BLOCKS_DEFAULT_BF16 in bf16
 * mode: the blocks measured faster fused at batch 256 / 512 on MI355X; 0 = every layer its own launch.
 * Under the fp32 DEFAULT a block is fused only when its launch has at least (compute units / 2) output tiles of 128 x 128
 * (1.0x224: blocks 4-7 from 6 images, blocks 8-11 from 11): below that two shorter launches are faster, and batch 1 runs the
 * stem + 26 single layers; an explicit mask is taken as given. fp32 results are bit-identical either way for calls of 5
 * images and more; for 1..4 images a stand-alone pointwise layer in the few-tile regime takes the split-K kernel
 * (mbn_pointwise), whose summation order differs from the block kernel's in the last bits. */
#define MBN_FUSE_BLOCKS_DEFAULT ((1u << 4) | (1u << 6) | (1u << 8) | (1u << 10))
/* bf16 mode: the matrix work is a sixteenth of fp32's and a block is bound by HBM and the depthwise VALU work: one launch
 * that never writes the depthwise output wins on every block inside the kernel's envelope whose pointwise layer is at
 * most 256 channels wide (one column tile: no depthwise recompute); wider blocks stay two launches under this default
 * and can be forced with an explicit mask. */
#define MBN_FUSE_BLOCKS_DEFAULT_BF16 0x0FFFFFFEu
int  mbn_net_set_fuse_blocks(mbn_net *net, unsigned mask);
int  mbn_net_get_fuse_blocks(const mbn_net *net, unsigned *mask);
/* Back to the state before any mbn_net_set_fuse_blocks call: the measured default of the current dtype WITH its
 * default-only rules (few-tile rule in fp32, one-column-tile rule in bf16), which an explicit mask switches off. */
int  mbn_net_reset_fuse_blocks(mbn_net *net);
/* Pool + FC as one launch (mbn_pool_fc) for calls of 1...4 images in fp32. DEFAULT OFF: measured on MI355X the one launch takes
 * 6.7 us against 1.9 + 2.1 us for the two (the cross-workgroup hand-over of the partial sums is three dependent memory round trips),
 * a forward of one image 0.1371 against 0.1348 ms (profiles/r03/p_pool_fc_one_launch.txt). The pooled values are identical either
 * way; the FC sums in another (fixed) order. */
int  mbn_net_set_fuse_tail(mbn_net *net, int enabled);
/* Runs of equal bf16 blocks on a small map (256 channels, stride 1, at most 10 x 10 pixels: the five 10 x 10 blocks of the 0.5x160 network) as ONE launch with
 * the map resident in LDS (mbn_blocks_resident_bf16), and the 0.5x160 network's last two blocks + pool as another (mbn_tail_resident_bf16). Default 1; 0 = one fused
 * launch per block / one launch per layer as before round 6. bf16 mode only; logits within the bf16 tolerance either way (the arithmetic is the same). */
int  mbn_net_set_fuse_resident(mbn_net *net, int enabled);
/* The launches the next forward(batch, last_layer) issues per (sub-)batch: launch j covers n_layers[j] layers starting
 * at the 1-based layer first_layer[j] (3 = fused stem, 2 = fused block or fused pool + FC, 1 = single layer). *count = number of
 * launches; the arrays (may be NULL) receive at most `capacity` entries. */
/* (The list assumes what mbn_net_forward's own buffers guarantee — 16-byte aligned tensors; a caller-provided `logits` /
 * last-layer buffer that is not 16-byte aligned makes the fused call answer MBN_EUNSUPPORTED and the forward issue the two
 * layer calls instead, one launch more than listed.) */
int  mbn_net_launches(const mbn_net *net, int batch, int last_layer, int *first_layer, int *n_layers, int capacity, int *count);
/* images: device fp32 NHWC [batch][res][res][3]; logits: device fp32 [batch][classes]. Asynchronous.
 * last_layer: run layers 1..last_layer only (0 or 29 => all; 5 and 13 = BASELINE configs 1-2), in which
 * case `logits` receives that layer's NHWC activation instead. */
int  mbn_net_forward(mbn_net *net, const void *images, void *logits, int batch, int last_layer);
/* forward + classifier read-out on device: the k <= 8 most probable classes per image (mbn_softmax_topk_f32 over the
 * net's own logits buffer): topk_idx [batch][k] int32, topk_prob [batch][k] fp32, device pointers. What MobileNet.c:2744-2792
 * does with a 1000-byte D2H and a host loop, with 2k values per image to download. */
int  mbn_net_classify(mbn_net *net, const void *images, int batch, int k, void *topk_idx_i32, void *topk_prob_f32);
/* Per-layer milliseconds of the most recent mbn_net_forward_timed call (hipEvents between layers). */
int  mbn_net_forward_timed(mbn_net *net, const void *images, void *logits, int batch, float *layer_ms,
                           int n_layer_ms);
int  mbn_net_plan(const mbn_net *net, mbn_plan *plan);
/* Device pointer of layer `index`'s output from the most recent forward (valid until the next forward that
 * overwrites the ping-pong buffer; with keep_activations every layer has its own buffer). */
int  mbn_net_set_keep_activations(mbn_net *net, int keep);
int  mbn_net_layer_output(mbn_net *net, int index, void **dptr, size_t *floats_per_image);

/* ------------------------------------------------------------------ multi-GPU (north_star: "batched images
 * shard naturally across the 8 GPUs of one node with an RCCL broadcast of weights over xGMI and no cross-GPU reduction").
 * The reference takes exactly one device (MobileNet.c:155 clGetDeviceIDs(..., 1, &device_id, ...)); this is the form a C
 * host uses to drive all GPUs of a node from one process, one thread per GPU:
 *   mbn_dist_init       one mbn_context per GPU (device_ordinals NULL = 0..n-1) + an RCCL communicator over them
 *                       (ncclCommInitAll; RCCL is bound with dlopen at this call, n_gpus = 1 needs none — with MBN_DIST_FORCE_RCCL=1 in the
 *                       environment one is built for a single GPU too: a rehearsal of the RCCL leg on a one-GPU box);
 *   mbn_dist_context    rank r's context: every other call of this header works on it, from the thread that owns rank r;
 *   mbn_dist_broadcast  dev_ptrs[r] = rank r's device buffer of `bytes` bytes; root's bytes overwrite the others' (one
 *                       grouped ncclBroadcast on the ranks' context streams; returns when all ranks have it). The one
 *                       collective of the path: the packed parameter blob, once, off the timed path;
 *   mbn_dist_sync       mbn_sync on every rank;
 *   mbn_shard_range     the contiguous slice [first, first+count) of `total` images owned by `rank` of `world`: total/world
 *                       each, the first total%world ranks take one more (host arithmetic; also in libmbn_host.so).
 * Contexts are independent, so per-GPU threads need no locking among themselves. */
typedef struct mbn_dist mbn_dist;
int  mbn_dist_init(int n_gpus, const int *device_ordinals, mbn_dist **dist);
int  mbn_dist_size(const mbn_dist *dist, int *n_gpus);
int  mbn_dist_context(mbn_dist *dist, int rank, mbn_context **ctx);
int  mbn_dist_broadcast(mbn_dist *dist, void *const *dev_ptrs, size_t bytes, int root);
int  mbn_dist_sync(mbn_dist *dist);
int  mbn_dist_shutdown(mbn_dist *dist);
const char *mbn_dist_last_error(const mbn_dist *dist);
int  mbn_shard_range(int total, int world, int rank, int *first, int *count);
/* One host thread per rank, for callers that drive several GPUs from one process (`mobilenet --gpus G`): mbn_run_ranks starts
 * n threads, opens a gate once ALL of them exist and runs fn(rank, arg, sync) in each; if a thread cannot be created nobody
 * runs fn and the call returns MBN_ENOMEM (bare pthread_create + pthread_barrier would strand the started ranks in the
 * barrier). Inside fn, mbn_rank_barrier(sync) is a barrier over the n ranks that returns MBN_EDEVICE instead of blocking once
 * any rank has failed (fn returned != MBN_OK, or called mbn_rank_fail). Returns the first failing rank's code;
 * rank_rc (may be NULL) receives every rank's (MBN_EUNSUPPORTED = its job did not run). fail_create_at >= 0 simulates a
 * failed thread creation at that rank (tests). Host code only: also in libmbn_host.so. */
typedef struct mbn_rank_sync mbn_rank_sync;
typedef int (*mbn_rank_fn)(int rank, void *arg, mbn_rank_sync *sync);
int  mbn_run_ranks(int n, mbn_rank_fn fn, void *arg, int fail_create_at, int *rank_rc);
int  mbn_rank_barrier(mbn_rank_sync *sync);
int  mbn_rank_fail(mbn_rank_sync *sync);

const char *mbn_version(void);

/* Process-wide switches. 0 always means "the shipped default". Two kinds:
 *   PRODUCT switches (every build): pw_tile, net_stagger, lit_dot, pw_splitk, pw_emul, pw_emul_static, pw_clock — they select between code
 *     paths the library always contains (a regime a caller or a test wants to force, the opt-in pw_emul arithmetic).
 *   LAB knobs (dw_variant, dw_nseg, pw_stage, conv_variant, misc, pw_ring, pw_xn, dwpw_variant, exp0..2): A/B hooks of the
 *     experiments recorded in profiles/LOG.md. They exist in the lab build only (make lab -> libmbn_lab.so, loaded by the tools with
 *     MBN_LAB=1); the shipped libmbn.so compiles them to zero together with every kernel instantiation only they can reach, and
 *     mbn_tune_set answers MBN_EUNSUPPORTED for them (mbn_tune_get: 0). mbn_lab_build() tells which library this is.
 * Unknown key => MBN_ENOTFOUND.
 *   pw_tile      force a GEMM tile shape of mbn_pointwise (mbn_f32_pw.hip; with pw_emul: mbn_f32_pw_x6.hip); a shape the build does
 *                not contain => MBN_EUNSUPPORTED from the call. 1 also forces the 128-column tile of mbn_dwpw_fused(_bf16);
 *                9 = the short-K resident-filter GEMM (mbn_f32_pw3.hip) wherever eligible, 10 = never (0: where it measured faster; same bits either way)
 *   misc         workgroups per CU of the persistent GEMM grid (1000 = one tile per workgroup); in the split-K kernel 16 / 32 =
 *                force the 16x16 / 32x32 workgroup tile
 *   pw_stage     1 = stage GEMM operands through registers instead of direct-to-LDS loads
 *   conv_variant 1 = generic conv1 kernel; 2 = bf16 fused stem with conv1 on the VALU instead of the bf16 MFMA; 8 = GEMM without the software-pipelined k-loop; 9 = GEMM with the general epilogue;
 *                16 / 32 (with pw_emul): split GEMM without the raised wave priority over the split / with one filter buffer
 *   dw_variant   depthwise: bits 0-1 = output columns per lane, bit 4 = lanes across all channels, bit 5 = bf16 with
 *                4-channel lanes, bit 7 = no raised wave priority
 *   dw_nseg      depthwise: row segments per image
 *   net_stagger  layers between the starts of consecutive sub-batch streams (mbn_net_set_streams)
 *   lit_dot      LITERAL pointwise: 0 = v_mfma_i32_32x32x32_i8 where eligible (no carry quirk, filter fits int8, Cin and Cout >= 16) and measured
 *                faster (K >= 512 or >= 16384 pixels in the call), else the v_dot4_i32_i8 path where eligible; 1 = scalar kernel; 2 = v_dot4 (never
 *                the matrix cores); 3 = the matrix cores wherever eligible. All bit-exact (kernel.cl:94-114)
 *   pw_ring      bf16 pointwise: 0 = streaming ring kernel for K = 64 (shipped), 1 = always the tiled GEMM, 2 = ring wherever eligible
 *   pw_splitk    fp32 pointwise of 1..4 images in the few-tile regime: 0 = split-K kernel (mbn_f32_pw_splitk.hip), 1 = always the
 *                tiled GEMM, 2 = split-K wherever the shape allows (K >= 128, K % 64 == 0), whatever the batch; 16 / 32 = as 2 with
 *                the 16x16 / 32x32 workgroup tile forced (same bits; tests)
 *   pw_emul      fp32 pointwise, OPT-IN arithmetic form (mbn_f32_pw_x6.hip): 0 = v_mfma_f32_32x32x2_f32 (default); 6 or 9 = every fp32
 *                operand split EXACTLY into three bf16 values (x = h + m + l, 24 bits kept) and the product formed from 6 (or all 9)
 *                bf16 x bf16 partial products on v_mfma_f32_32x32x16_bf16 with the fp32 accumulator; fp32 in, fp32 out. The 3 dropped
 *                partial products of 6 are below 2^-24 of the product. Measured error against a float64 product on the network's
 *                layer shapes: equal to or smaller than the fp32 MFMA kernel's (profiles/r02/m_pw_emul.txt); layers 13-27 at
 *                batch 256: 1.41 -> 1.14 ms; with the fused blocks (mbn_dwpw_fused takes mbn_f32_dwpw2_x6.hip for Cin <= 512) and the
 *                pointwise phase of mbn_stem_fused the whole step 89 k -> 109-110 k images/s. Applies to pointwise calls with K % 32 == 0 and at least as many 128x128 tiles as
 *                CUs; everything else takes the default kernels — which form a layer takes therefore depends on the size of the call,
 *                and forward(n)[:k] == forward(k) holds bit for bit only between calls whose layers take the same forms (fused blocks
 *                and stand-alone pairs agree bit for bit under the same value). Each filter pointer gets a pre-split image (6 bytes per
 *                weight) in the context on first use (not inside a hipGraph capture: the unsplit-filter kernel is taken there).
 *                Operand range: the split is exact for 2^-110 <= |x| < 2^127 (measured, profiles/r02/m_pw_emul.txt (19)); in fp32's top
 *                binade h = bf16(x) can round to infinity, below 2^-110 the matrix cores flush the bf16 denormals of the low planes
 *                (2e-6 relative at 2^-115). Post-ReLU6 activations and BN-folded weights are far inside.
 *   pw_emul_static  with pw_emul: 0 = a filter's pre-split image is rewritten by every call that uses it (the filter may change between
 *                calls like any other argument); 1 = the caller's promise that filters are only ever written through mbn_upload /
 *                mbn_memset (or freed through mbn_free): an image is then split once and reused until one of those calls touches its
 *                filter — 13 launches of ~5 us less per forward of the network. bench.py --pw-emul and mobilenet --pw-emul set it
 *                (weights are uploaded once there).
 *   pw_xn        pointwise GEMM tile order: XCD groups along n (0 = by filter size, 1 = single ordering, 2, 4)
 *   dwpw_variant fused block kernel: 0 = shipped choice, 1 = round-1 producer/consumer kernels, 2 = unified-wave kernels,
 *                3 = unified fp32 with the taps read inside the step, 100 + bits = unified with parts switched off
 *   pw_clock     1 = every pw_gemm launch adds the core-clock cycles (s_memtime) and the 100 MHz reference ticks (s_memrealtime) its
 *                first eight workgroups (one per XCD) lived to device counters; mbn_pw_clock_read turns them into the clock the
 *                chip HELD under the fp32 MFMA stream. Off (default): two scalar compares per launch. bench.py sets it for its
 *                untimed profiled steps only, so that `roofline.frac` can be compared across boxes (DVFS: 2.1-2.4 GHz by box;
 *                product switch, every build) */
int  mbn_tune_set(const char *key, int value);
/* Mean core clock (GHz) over the pw_gemm launches recorded since the last reset (tune key pw_clock), and how many launches that was.
 * Waits for the device. reset != 0 clears the counters after reading. No launches recorded => *ghz = 0. The counterpart in the
 * reference is nothing: its only timing is clGetEventProfilingInfo (MobileNet.c:301-305). */
int  mbn_pw_clock_read(mbn_context *ctx, int reset, double *ghz, long long *launches);
int  mbn_lab_build(void);      /* 1 = built with -DMBN_LAB (all A/B variants and knobs), 0 = the shipped library */
int  mbn_tune_get(const char *key, int *value);

#ifdef __cplusplus
}
#endif
#endif /* MBN_H */
