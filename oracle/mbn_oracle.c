/*
 * mbn_oracle.c — CPU restatement of kernel.cl (LITERAL mode) and of the fp32 MobileNet-V1 the metric
 * measures (F32 mode). TEST INFRASTRUCTURE ONLY — see mbn_oracle.h. PARITY UNPINNED (no reference
 * fixtures exist); pinned by hand-derived known-answer vectors in tests/golden/.
 *
 * Written from the semantics of /root/reference/kernel.cl and the layer table of
 * /root/reference/MobileNet.c; no reference source is copied. Every function carries the lines it follows.
 *
 * Build: make -C oracle   (gcc -O2 -fopenmp -shared)
 */
#include "mbn_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* int32 arithmetic with defined wrap-around (OpenCL `int` overflow wraps on every device the
 * reference targeted; C signed overflow is UB, so do it unsigned). */
static inline int32_t mac_i32(int32_t sum, uint8_t a, int32_t w)
{
    return (int32_t)((uint32_t)sum + (uint32_t)a * (uint32_t)w);
}

/* kernel.cl:52-56 / 87-90 / 109-112: `if (sum <= 0) sum = 0;` then store int -> uchar (truncation). */
static inline uint8_t relu_store_u8(int32_t *sum)
{
    if (*sum <= 0) *sum = 0;
    return (uint8_t)(*sum & 0xFF);
}

/* One tap of kernel.cl:19-25 (and :31-37, :43-49, :78-84).
 *   literal_index: the reference's own expression  in[yindex*G0*stride + xindex*stride]  where
 *     yindex = ty+i, xindex = tx+j are WORK-ITEM coordinates; `limit` = elements available from
 *     `plane` to the end of the whole input buffer (reads past it return 0: B4/B5 deviation).
 *   otherwise: intended strided geometry: row ty*stride+i, col tx*stride+j of an
 *     in_rows x in_cols plane, zero outside the plane on the high side. */
static inline uint8_t lit_tap(const uint8_t *plane, long limit, int literal_index, int ty, int tx, int i,
                              int j, int stride, int g0, int in_rows, int in_cols)
{
    if (literal_index) {
        int yindex = ty + i, xindex = tx + j;
        if (yindex < 0 || xindex < 0) return 0;                     /* kernel.cl:20 */
        long idx = (long)yindex * g0 * stride + (long)xindex * stride; /* kernel.cl:24 */
        return idx < limit ? plane[idx] : 0;
    } else {
        int iy = ty * stride + i, ix = tx * stride + j;
        if (iy < 0 || ix < 0) return 0;                             /* top/left pad only (B6) */
        if (iy >= in_rows || ix >= in_cols) return 0;               /* bounds-safe high side (B5) */
        return plane[(long)iy * in_cols + ix];
    }
}

/* kernel.cl:2-60 */
void orc_lit_convolute(uint8_t *output, const uint8_t *inp_r, const uint8_t *inp_g, const uint8_t *inp_b,
                       const int32_t *filter_k, int rows, int cols, int filtersize, int stride, int op_size,
                       uint32_t quirks, int g0, int g1)
{
    const int half = filtersize / 2;                                /* kernel.cl:8 */
    const int orow = rows / stride, ocol = cols / stride;
    const int lit = (quirks & ORC_Q_LITERAL_INDEX) != 0;
    const int carry = (quirks & ORC_Q_CARRY_SUM) != 0;
    const long out_plane = lit ? (long)(rows / 2) * (cols / 2)      /* kernel.cl:14 */
                               : (long)orow * ocol;
    if (g0 <= 0) g0 = ocol;
    if (g1 <= 0) g1 = orow;
    const uint8_t *planes[3] = { inp_r, inp_g, inp_b };
    const long limit = (long)rows * cols;
    for (int ty = 0; ty < g1; ty++) {
        for (int tx = 0; tx < g0; tx++) {
            int32_t sum = 0;                                        /* kernel.cl:10 */
            int findex = 0;
            for (int oc = 0; oc < op_size; oc++) {                  /* kernel.cl:13 */
                if (!carry) sum = 0;
                for (int p = 0; p < 3; p++)                         /* r :16-27, g :28-39, b :40-51 */
                    for (int i = -half; i <= half; i++)
                        for (int j = -half; j <= half; j++, findex++)
                            sum = mac_i32(sum,
                                          lit_tap(planes[p], limit, lit, ty, tx, i, j, stride, g0, rows, cols),
                                          filter_k[findex]);
                output[((long)ty * g0 + tx) + out_plane * oc] = relu_store_u8(&sum); /* :52-56 */
            }
        }
    }
}

/* kernel.cl:62-92 */
void orc_lit_depthwise(uint8_t *output, const uint8_t *inp_image, const int32_t *filter_k, int rows,
                       int cols, int filtersize, int stride, int op_size, int in_rows, int in_cols,
                       uint32_t quirks, int g0, int g1)
{
    const int half = filtersize / 2;
    const int lit = (quirks & ORC_Q_LITERAL_INDEX) != 0;
    const int carry = (quirks & ORC_Q_CARRY_SUM) != 0;
    const int plane0 = (quirks & ORC_Q_DW_PLANE0) != 0;
    if (in_rows <= 0) in_rows = rows * stride;
    if (in_cols <= 0) in_cols = cols * stride;
    if (g0 <= 0) g0 = cols;
    if (g1 <= 0) g1 = rows;
    const long in_plane = (long)in_rows * in_cols;
    const long total = in_plane * op_size;
    for (int ty = 0; ty < g1; ty++) {
        for (int tx = 0; tx < g0; tx++) {
            int32_t sum = 0;                                        /* kernel.cl:69 */
            int findex = 0;
            for (int oc = 0; oc < op_size; oc++) {
                if (!carry) sum = 0;
                const long base = plane0 ? 0 : in_plane * oc;       /* kernel.cl:83 has no channel offset */
                for (int i = -half; i <= half; i++)
                    for (int j = -half; j <= half; j++, findex++)
                        sum = mac_i32(sum,
                                      lit_tap(inp_image + base, total - base, lit, ty, tx, i, j, stride, g0,
                                              in_rows, in_cols),
                                      filter_k[findex]);
                output[((long)ty * g0 + tx) + (long)rows * cols * oc] = relu_store_u8(&sum); /* :73,90 */
            }
        }
    }
}

/* kernel.cl:94-114 */
void orc_lit_pointwise(uint8_t *output, const uint8_t *inp_image, const int32_t *filter_k, int rows,
                       int cols, int filtersize, int op_size, uint32_t quirks)
{
    const int carry = (quirks & ORC_Q_CARRY_SUM) != 0;
    const long plane = (long)rows * cols;
    for (long p = 0; p < plane; p++) {                              /* p = ty*gsize0 + tx */
        int32_t sum = 0;                                            /* kernel.cl:99 */
        int findex = 0;
        for (int oc = 0; oc < op_size; oc++) {
            if (!carry) sum = 0;
            for (int i = 0; i < filtersize; i++, findex++)          /* kernel.cl:106-108 */
                sum = mac_i32(sum, inp_image[p + plane * i], filter_k[findex]);
            output[p + plane * oc] = relu_store_u8(&sum);
        }
    }
}

/* kernel.cl:116-132, work-item (0,0) */
void orc_lit_pool(uint8_t *output, const uint8_t *inp_image, int rows, int cols, int filtersize,
                  int op_size, uint32_t quirks)
{
    const int carry = (quirks & ORC_Q_CARRY_SUM) != 0;
    const int div = (quirks & ORC_Q_POOL_DIV49) ? 49 : filtersize * filtersize; /* kernel.cl:129 */
    int32_t sum = 0;                                                /* kernel.cl:121 */
    for (int c = 0; c < op_size; c++) {
        if (!carry) sum = 0;
        const long shift = (long)rows * cols * c;                   /* kernel.cl:125 */
        for (int i = 0; i < filtersize * filtersize; i++)           /* kernel.cl:126-128: flat, not 2-D */
            sum = (int32_t)((uint32_t)sum + inp_image[i + shift]);
        output[c] = (uint8_t)((sum / div) & 0xFF);
    }
}

/* MobileNet.c:2771-2792 */
void orc_softmax_argmax_u8(const uint8_t *logits, int n, double *probs, int *location, double *maximum)
{
    double sum = 0.0;
    for (int k = 0; k < n; k++) {
        probs[k] = exp((double)logits[k]);
        sum += exp((double)logits[k]);
    }
    for (int k = 0; k < n; k++) probs[k] = probs[k] / sum;
    double mx = probs[0];
    int loc = 1;                                                    /* reference leaves it uninitialised (B11) */
    for (int k = 1; k < n; k++)
        if (probs[k] > mx) { mx = probs[k]; loc = k + 1; }
    *location = loc;
    *maximum = mx;
}

/* ------------------------------------------------------------------ F32 mode */

static inline float act_f32(float v, int act)
{
    if (act == ORC_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == ORC_ACT_RELU6) return v < 0.f ? 0.f : (v > 6.f ? 6.f : v);
    return v;
}

static inline float bn_act(double acc, const float *scale, const float *shift, int c, int act)
{
    float v = (float)acc;
    float s = scale ? scale[c] : 1.f, b = shift ? shift[c] : 0.f;
    return act_f32(fmaf(v, s, b), act);
}

int orc_same_pad(int in, int out, int k, int stride)
{
    int total = (out - 1) * stride + k - in;
    if (total < 0) total = 0;
    return total / 2;
}

/* intended math of kernel.cl:2-60 (3x3xCin conv) in fp32 NHWC */
void orc_f32_conv(float *out, const float *in, const float *filter, const float *scale, const float *shift,
                  int batch, int rows, int cols, int cin, int filtersize, int stride, int op_size,
                  int pad_top, int pad_left, int act)
{
    const int orow = (rows + stride - 1) / stride, ocol = (cols + stride - 1) / stride;
    if (pad_top < 0) pad_top = orc_same_pad(rows, orow, filtersize, stride);
    if (pad_left < 0) pad_left = orc_same_pad(cols, ocol, filtersize, stride);
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < batch; n++)
        for (int oy = 0; oy < orow; oy++)
            for (int ox = 0; ox < ocol; ox++) {
                float *o = out + (((long)n * orow + oy) * ocol + ox) * op_size;
                for (int oc = 0; oc < op_size; oc++) {
                    double acc = 0.0;
                    for (int ky = 0; ky < filtersize; ky++) {
                        int iy = oy * stride + ky - pad_top;
                        if (iy < 0 || iy >= rows) continue;
                        for (int kx = 0; kx < filtersize; kx++) {
                            int ix = ox * stride + kx - pad_left;
                            if (ix < 0 || ix >= cols) continue;
                            const float *ip = in + (((long)n * rows + iy) * cols + ix) * cin;
                            const float *fp = filter + ((long)(ky * filtersize + kx) * cin) * op_size + oc;
                            for (int ci = 0; ci < cin; ci++) acc += (double)ip[ci] * (double)fp[(long)ci * op_size];
                        }
                    }
                    o[oc] = bn_act(acc, scale, shift, oc, act);
                }
            }
}

/* intended math of kernel.cl:62-92 in fp32 NHWC */
void orc_f32_depthwise(float *out, const float *in, const float *filter, const float *scale,
                       const float *shift, int batch, int rows, int cols, int in_rows, int in_cols,
                       int filtersize, int stride, int channels, int pad_top, int pad_left, int act)
{
    if (in_rows <= 0) in_rows = rows * stride;
    if (in_cols <= 0) in_cols = cols * stride;
    if (pad_top < 0) pad_top = orc_same_pad(in_rows, rows, filtersize, stride);
    if (pad_left < 0) pad_left = orc_same_pad(in_cols, cols, filtersize, stride);
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < batch; n++)
        for (int oy = 0; oy < rows; oy++)
            for (int ox = 0; ox < cols; ox++) {
                float *o = out + (((long)n * rows + oy) * cols + ox) * channels;
                for (int c = 0; c < channels; c++) {
                    double acc = 0.0;
                    for (int ky = 0; ky < filtersize; ky++) {
                        int iy = oy * stride + ky - pad_top;
                        if (iy < 0 || iy >= in_rows) continue;
                        for (int kx = 0; kx < filtersize; kx++) {
                            int ix = ox * stride + kx - pad_left;
                            if (ix < 0 || ix >= in_cols) continue;
                            acc += (double)in[(((long)n * in_rows + iy) * in_cols + ix) * channels + c] *
                                   (double)filter[(long)(ky * filtersize + kx) * channels + c];
                        }
                    }
                    o[c] = bn_act(acc, scale, shift, c, act);
                }
            }
}

/* intended math of kernel.cl:94-114 in fp32: out[m][oc] = sum_i in[m][i] * filter[oc][i] */
void orc_f32_pointwise(float *out, const float *in, const float *filter, const float *scale,
                       const float *shift, long m, int cin, int op_size, int act)
{
#pragma omp parallel for schedule(static)
    for (long r = 0; r < m; r++) {
        const float *ip = in + r * cin;
        float *o = out + r * op_size;
        for (int oc = 0; oc < op_size; oc++) {                      /* pixel-outer, channel-inner like kernel.cl:100 */
            const float *fp = filter + (long)oc * cin;
            double acc = 0.0;
            for (int i = 0; i < cin; i++) acc += (double)ip[i] * (double)fp[i];
            o[oc] = bn_act(acc, scale, shift, oc, act);
        }
    }
}

/* intended math of kernel.cl:116-132 in fp32 */
void orc_f32_pool(float *out, const float *in, int batch, int rows, int cols, int filtersize, int channels)
{
    int fr = filtersize < rows ? filtersize : rows, fc = filtersize < cols ? filtersize : cols;
    for (int n = 0; n < batch; n++)
        for (int c = 0; c < channels; c++) {
            double acc = 0.0;
            for (int y = 0; y < fr; y++)
                for (int x = 0; x < fc; x++) acc += in[(((long)n * rows + y) * cols + x) * channels + c];
            out[(long)n * channels + c] = (float)(acc / (double)(fr * fc));
        }
}

void orc_f32_softmax(float *probs, int32_t *argmax, const float *logits, int batch, int classes)
{
    for (int n = 0; n < batch; n++) {
        const float *l = logits + (long)n * classes;
        float mx = l[0];
        int am = 0;
        for (int k = 1; k < classes; k++)
            if (l[k] > mx) { mx = l[k]; am = k; }
        double sum = 0.0;
        for (int k = 0; k < classes; k++) sum += exp((double)l[k] - (double)mx);
        if (probs)
            for (int k = 0; k < classes; k++)
                probs[(long)n * classes + k] = (float)(exp((double)l[k] - (double)mx) / sum);
        if (argmax) argmax[n] = am;
    }
}

/* ------------------------------------------------------------------ topology */

static long align64(long x) { return (x + 63) & ~63L; }

/* MobileNet.c:13-26 (#define FILTER_SIZE_*) + the per-layer literals tabulated in SURVEY.md §2.1 */
int orc_plan_build(float alpha, int res, int classes, orc_plan *plan)
{
    static const int base_ch[14] = { 32, 64, 128, 128, 256, 256, 512, 512, 512, 512, 512, 512, 1024, 1024 };
    static const int dw_stride[13] = { 1, 2, 1, 2, 1, 2, 1, 1, 1, 1, 1, 2, 1 };
    if (!plan || alpha <= 0.f || res < 32 || classes <= 0) return -1;
    memset(plan, 0, sizeof(*plan));
    plan->alpha = alpha;
    plan->res = res;
    plan->classes = classes;
    long off = 0, max_act = (long)res * res * 3;
    int L = 0, h = res, c = 3;
    orc_layer *l = &plan->layer[L++];
    int oc = (int)(base_ch[0] * alpha);
    l->index = 1; l->kind = ORC_L_CONV; l->in_rows = l->in_cols = h; l->in_ch = c; l->stride = 2;
    l->out_rows = l->out_cols = (h + 1) / 2; l->out_ch = oc;
    l->pad_top = l->pad_left = orc_same_pad(h, l->out_rows, 3, 2);
    l->w_offset = off; l->w_count = 27L * oc; off = align64(off + l->w_count);
    l->scale_offset = off; off = align64(off + oc);
    l->shift_offset = off; off = align64(off + oc);
    h = l->out_rows; c = oc;
    if ((long)h * h * c > max_act) max_act = (long)h * h * c;
    for (int b = 0; b < 13; b++) {
        l = &plan->layer[L++];
        l->index = L; l->kind = ORC_L_DW; l->in_rows = l->in_cols = h; l->in_ch = c; l->stride = dw_stride[b];
        l->out_rows = l->out_cols = (h + l->stride - 1) / l->stride; l->out_ch = c;
        l->pad_top = l->pad_left = orc_same_pad(h, l->out_rows, 3, l->stride);
        l->w_offset = off; l->w_count = 9L * c; off = align64(off + l->w_count);
        l->scale_offset = off; off = align64(off + c);
        l->shift_offset = off; off = align64(off + c);
        h = l->out_rows;
        if ((long)h * h * c > max_act) max_act = (long)h * h * c;
        oc = (int)(base_ch[b + 1] * alpha);
        l = &plan->layer[L++];
        l->index = L; l->kind = ORC_L_PW; l->in_rows = l->in_cols = h; l->in_ch = c; l->stride = 1;
        l->out_rows = l->out_cols = h; l->out_ch = oc;
        l->w_offset = off; l->w_count = (long)oc * c; off = align64(off + l->w_count);
        l->scale_offset = off; off = align64(off + oc);
        l->shift_offset = off; off = align64(off + oc);
        c = oc;
        if ((long)h * h * c > max_act) max_act = (long)h * h * c;
    }
    l = &plan->layer[L++];
    l->index = L; l->kind = ORC_L_POOL; l->in_rows = l->in_cols = h; l->in_ch = c; l->stride = 1;
    l->out_rows = l->out_cols = 1; l->out_ch = c; l->w_offset = off; l->scale_offset = l->shift_offset = -1;
    l = &plan->layer[L++];
    l->index = L; l->kind = ORC_L_FC; l->in_rows = l->in_cols = 1; l->in_ch = c; l->stride = 1;
    l->out_rows = l->out_cols = 1; l->out_ch = classes;
    l->w_offset = off; l->w_count = (long)classes * c; off = align64(off + l->w_count);
    l->scale_offset = -1;
    l->shift_offset = off; off = align64(off + classes);
    plan->n_layers = L;
    plan->blob_floats = off;
    plan->max_act_floats = max_act;
    return 0;
}

int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

float orc_bf16_round(float x)
{
    uint32_t u;
    memcpy(&u, &x, 4);
    if ((u & 0x7F800000u) == 0x7F800000u) { u &= 0xFFFF0000u; memcpy(&x, &u, 4); return x; }   /* inf/nan: truncate */
    u = (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;      /* round to nearest, ties to even */
    memcpy(&x, &u, 4);
    return x;
}

void orc_bf16_round_array(float *x, long n)
{
    for (long i = 0; i < n; i++) x[i] = orc_bf16_round(x[i]);
}

static int net_forward_impl(const orc_plan *plan, const float *blob, const float *images, float *out, int batch,
                            int last_layer, int threads, float **layer_out, int bf16);

int orc_net_forward(const orc_plan *plan, const float *blob, const float *images, float *out, int batch,
                    int last_layer, int threads, float **layer_out)
{
    return net_forward_impl(plan, blob, images, out, batch, last_layer, threads, layer_out, 0);
}

int orc_net_forward_bf16(const orc_plan *plan, const float *blob_in, const float *images, float *out, int batch,
                         int last_layer, int threads, float **layer_out)
{
    if (!plan || !blob_in) return -1;
    float *blob = (float *)malloc(sizeof(float) * plan->blob_floats);
    if (!blob) return -2;
    memcpy(blob, blob_in, sizeof(float) * plan->blob_floats);
    for (int i = 0; i < plan->n_layers; i++)                     /* pointwise / FC filters are stored as bf16 */
        if (plan->layer[i].kind == ORC_L_PW || plan->layer[i].kind == ORC_L_FC)
            orc_bf16_round_array(blob + plan->layer[i].w_offset, plan->layer[i].w_count);
    int rc = net_forward_impl(plan, blob, images, out, batch, last_layer, threads, layer_out, 1);
    free(blob);
    return rc;
}

/* MobileNet.c:240-2763: the 29 layers in order; activations ping-pong between two host buffers. */
static int net_forward_impl(const orc_plan *plan, const float *blob, const float *images, float *out, int batch,
                            int last_layer, int threads, float **layer_out, int bf16)
{
    if (!plan || !blob || !images || !out || batch <= 0) return -1;
    if (last_layer <= 0 || last_layer > plan->n_layers) last_layer = plan->n_layers;
#ifdef _OPENMP
    int prev_threads = omp_get_max_threads();
    omp_set_num_threads(threads > 1 ? threads : 1);
#else
    (void)threads;
#endif
    long cap = plan->max_act_floats * batch;
    float *buf[2] = { (float *)malloc(sizeof(float) * cap), (float *)malloc(sizeof(float) * cap) };
    if (!buf[0] || !buf[1]) { free(buf[0]); free(buf[1]); return -2; }
    const float *cur = images;
    int which = 0;
    for (int i = 0; i < last_layer; i++) {
        const orc_layer *l = &plan->layer[i];
        float *dst = buf[which];
        const float *w = blob + l->w_offset;
        const float *sc = l->scale_offset >= 0 ? blob + l->scale_offset : NULL;
        const float *sh = l->shift_offset >= 0 ? blob + l->shift_offset : NULL;
        switch (l->kind) {
        case ORC_L_CONV:
            orc_f32_conv(dst, cur, w, sc, sh, batch, l->in_rows, l->in_cols, l->in_ch, 3, l->stride, l->out_ch,
                         l->pad_top, l->pad_left, ORC_ACT_RELU6);
            break;
        case ORC_L_DW:
            orc_f32_depthwise(dst, cur, w, sc, sh, batch, l->out_rows, l->out_cols, l->in_rows, l->in_cols, 3,
                              l->stride, l->out_ch, l->pad_top, l->pad_left, ORC_ACT_RELU6);
            break;
        case ORC_L_PW:
            orc_f32_pointwise(dst, cur, w, sc, sh, (long)batch * l->out_rows * l->out_cols, l->in_ch, l->out_ch,
                              ORC_ACT_RELU6);
            break;
        case ORC_L_POOL:
            orc_f32_pool(dst, cur, batch, l->in_rows, l->in_cols, l->in_rows, l->in_ch);
            break;
        case ORC_L_FC:
            orc_f32_pointwise(dst, cur, w, NULL, sh, batch, l->in_ch, l->out_ch, ORC_ACT_NONE);
            break;
        default:
            free(buf[0]); free(buf[1]);
            return -3;
        }
        long cnt = (long)batch * l->out_rows * l->out_cols * l->out_ch;
        if (bf16 && l->kind != ORC_L_FC) orc_bf16_round_array(dst, cnt);    /* activations live in HBM as bf16 */
        if (layer_out && layer_out[i]) memcpy(layer_out[i], dst, sizeof(float) * cnt);
        if (i == last_layer - 1) memcpy(out, dst, sizeof(float) * cnt);
        cur = dst;
        which ^= 1;
    }
    free(buf[0]);
    free(buf[1]);
#ifdef _OPENMP
    omp_set_num_threads(prev_threads);
#endif
    return 0;
}
