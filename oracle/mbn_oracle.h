/*
 * mbn_oracle.h — CPU restatement of the reference's device kernels. TEST INFRASTRUCTURE ONLY.
 *
 * PARITY UNPINNED: the reference (anerisheth19/CNN-MobileNet-V1-implementation-on-AWS-FPGA-using-OpenCL)
 * ships no tests, golden vectors or expected outputs, cannot run in the build container
 * (zero OpenCL devices; its inputs Cat_Image0.ppm / weights_c.txt / the .h5 are absent) and cannot be
 * compiled for the CPU without writing stand-ins for the OpenCL device runtime. This oracle is therefore
 * pinned only by known-answer vectors derived by hand from kernel.cl (tests/golden/kat_literal.json)
 * and cross-checked against torch.nn.functional.conv2d as an independent implementation.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything in oracle/.
 * The product (include/mbn.h, libmbn.so) never links, loads or calls it.
 *
 * Each function cites the reference lines it restates.
 */
#ifndef MBN_ORACLE_H
#define MBN_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* quirk bits — same numeric values as include/mbn.h MBN_Q_* (kept separate on purpose) */
#define ORC_Q_CARRY_SUM      0x1u  /* kernel.cl:10,69,99,121 */
#define ORC_Q_DW_PLANE0      0x2u  /* kernel.cl:83 */
#define ORC_Q_LITERAL_INDEX  0x4u  /* kernel.cl:24,36,48,83 */
#define ORC_Q_POOL_DIV49     0x8u  /* kernel.cl:129 */
#define ORC_QUIRKS_KERNEL_CL 0xFu

enum { ORC_ACT_NONE = 0, ORC_ACT_RELU = 1, ORC_ACT_RELU6 = 2 };

/* ---------------- LITERAL mode: uint8 planar NCHW, int32 filters, int32 accumulate ---------------- */

/* kernel.cl:2-60. rows, cols = input plane size; output plane (rows/stride)x(cols/stride) with plane
 * stride (rows/2)*(cols/2) when stride==2 exactly as kernel.cl:14. g0,g1 = emulated NDRange (0 => output size). */
void orc_lit_convolute(uint8_t *output, const uint8_t *inp_r, const uint8_t *inp_g, const uint8_t *inp_b,
                       const int32_t *filter_k, int rows, int cols, int filtersize, int stride, int op_size,
                       uint32_t quirks, int g0, int g1);

/* kernel.cl:62-92. rows, cols = OUTPUT plane size; input plane in_rows x in_cols. */
void orc_lit_depthwise(uint8_t *output, const uint8_t *inp_image, const int32_t *filter_k, int rows,
                       int cols, int filtersize, int stride, int op_size, int in_rows, int in_cols,
                       uint32_t quirks, int g0, int g1);

/* kernel.cl:94-114. filtersize = number of input channels actually summed. */
void orc_lit_pointwise(uint8_t *output, const uint8_t *inp_image, const int32_t *filter_k, int rows,
                       int cols, int filtersize, int op_size, uint32_t quirks);

/* kernel.cl:116-132, as computed by work-item (0,0) (the only one whose store lands where intended). */
void orc_lit_pool(uint8_t *output, const uint8_t *inp_image, int rows, int cols, int filtersize,
                  int op_size, uint32_t quirks);

/* MobileNet.c:2771-2792. location is 1-based; 1 when class 0 wins. */
void orc_softmax_argmax_u8(const uint8_t *logits, int n, double *probs, int *location, double *maximum);

/* ---------------- F32 mode: fp32 NHWC, conv -> scale/shift -> activation ---------------- */
/* Sums are accumulated in double in (ky,kx,ci) order and rounded to float once. */

/* in [N][rows][cols][cin]; filter [k][k][cin][op_size]; out [N][orow][ocol][op_size], orow=ceil(rows/stride) */
void orc_f32_conv(float *out, const float *in, const float *filter, const float *scale, const float *shift,
                  int batch, int rows, int cols, int cin, int filtersize, int stride, int op_size,
                  int pad_top, int pad_left, int act);

/* in [N][in_rows][in_cols][C]; filter [k][k][C]; out [N][rows][cols][C] */
void orc_f32_depthwise(float *out, const float *in, const float *filter, const float *scale,
                       const float *shift, int batch, int rows, int cols, int in_rows, int in_cols,
                       int filtersize, int stride, int channels, int pad_top, int pad_left, int act);

/* in [M][cin]; filter [op_size][cin]; out [M][op_size]; M = batch*rows*cols */
void orc_f32_pointwise(float *out, const float *in, const float *filter, const float *scale,
                       const float *shift, long m, int cin, int op_size, int act);

/* in [N][rows][cols][C] -> out [N][C], mean over the filtersize x filtersize top-left window */
void orc_f32_pool(float *out, const float *in, int batch, int rows, int cols, int filtersize, int channels);

void orc_f32_softmax(float *probs, int32_t *argmax, const float *logits, int batch, int classes);

int  orc_same_pad(int in, int out, int k, int stride);   /* TF "SAME": leading pad */

/* ---------------- topology + whole-net forward (MobileNet.c:240-2763 order) ---------------- */
enum { ORC_L_CONV = 1, ORC_L_DW = 2, ORC_L_PW = 3, ORC_L_POOL = 4, ORC_L_FC = 5 };
typedef struct orc_layer {
    int index, kind, in_rows, in_cols, in_ch, out_rows, out_cols, out_ch, stride, pad_top, pad_left;
    long w_offset, w_count, scale_offset, shift_offset;
} orc_layer;
typedef struct orc_plan {
    int n_layers, res, classes;
    float alpha;
    long blob_floats, max_act_floats;
    orc_layer layer[32];
} orc_plan;

int  orc_plan_build(float alpha, int res, int classes, orc_plan *plan);

/* Runs layers 1..last_layer (0 => all) over `batch` NHWC images with the packed blob.
 * out receives the last layer's activation. threads<=1 => serial; else OpenMP over output pixels.
 * If layer_out != NULL, layer_out[i] (i = 0..n-1, may individually be NULL) receives a copy of layer i+1's output. */
int  orc_net_forward(const orc_plan *plan, const float *blob, const float *images, float *out, int batch,
                     int last_layer, int threads, float **layer_out);

int  orc_num_threads(void);   /* omp_get_max_threads() or 1 */

/* ---------------- bf16 mode (BASELINE config 5): emulated on the CPU ----------------
 * Activations are bf16 in memory: every layer's output is rounded to bf16 (round-to-nearest-even) before the next
 * layer reads it; pointwise/FC filters are bf16; depthwise/conv1 filters, scale/shift, accumulation and the FC
 * logits stay fp32. The input image is fp32. */
float orc_bf16_round(float x);
void  orc_bf16_round_array(float *x, long n);
int   orc_net_forward_bf16(const orc_plan *plan, const float *blob, const float *images, float *out, int batch,
                           int last_layer, int threads, float **layer_out);

#ifdef __cplusplus
}
#endif
#endif
