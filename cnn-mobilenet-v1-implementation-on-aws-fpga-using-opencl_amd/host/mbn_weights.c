/*
 * mbn_weights.c — Keras MobileNet-V1 .h5 -> packed, BatchNorm-folded parameter blob (and a deterministic
 * synthetic generator that writes the same Keras layout, so benchmarks and tests go through the real loader).
 *
 * Replaces the role keras.py:1-8 + readSquezeNetKernel (MobileNet.c:31-47) play in the reference: getting
 * trained Keras weights into the host arrays the kernels consume. The reference's route cannot work
 * (keras.py dumps raw HDF5 bytes; the C side truncates doubles in (-1,1) to int 0: SURVEY.md Appendix B, B8).
 *
 * Keras-applications layout (SURVEY.md Appendix C — external knowledge, not evidenced in the reference):
 *   /conv1/conv1/kernel:0 (3,3,3,C)                   /conv1_bn/conv1_bn/{gamma,beta,moving_mean,moving_variance}:0
 *   /conv_dw_i/conv_dw_i/depthwise_kernel:0 (3,3,C,1) /conv_dw_i_bn/...        i = 1..13
 *   /conv_pw_i/conv_pw_i/kernel:0 (1,1,Cin,Cout)      /conv_pw_i_bn/...
 *   /conv_preds/conv_preds/kernel:0 (1,1,C,classes)   /conv_preds/conv_preds/bias:0 (classes)
 * BatchNorm epsilon 1e-3. Folding: scale = gamma / sqrt(var + eps), shift = beta - mean * scale.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mbn.h"

#define BN_EPS 1e-3

static int64_t shape_count(int ndim, const int64_t *shape)
{
    int64_t c = 1;
    for (int i = 0; i < ndim; i++) c *= shape[i];
    return c;
}

/* fetch "<group>/<group>/<leaf>" and require exactly `want` elements */
static int get_param(mbn_h5 *h5, const char *group, const char *leaf, int64_t want, const float **data)
{
    char name[256];
    int ndim;
    int64_t shape[8];
    if (snprintf(name, sizeof(name), "/%s/%s/%s", group, group, leaf) >= (int)sizeof(name)) return MBN_EINVAL;
    int rc = mbn_h5_get(h5, name, &ndim, shape, data);
    if (rc != MBN_OK) return rc;
    if (shape_count(ndim, shape) != want) return MBN_ESHAPE;
    return MBN_OK;
}

static int fold_bn(mbn_h5 *h5, const char *bn_group, int ch, float *scale, float *shift)
{
    const float *gamma, *beta, *mean, *var;
    int rc;
    if ((rc = get_param(h5, bn_group, "gamma:0", ch, &gamma)) != MBN_OK) return rc;
    if ((rc = get_param(h5, bn_group, "beta:0", ch, &beta)) != MBN_OK) return rc;
    if ((rc = get_param(h5, bn_group, "moving_mean:0", ch, &mean)) != MBN_OK) return rc;
    if ((rc = get_param(h5, bn_group, "moving_variance:0", ch, &var)) != MBN_OK) return rc;
    for (int c = 0; c < ch; c++) {
        double s = (double)gamma[c] / sqrt((double)var[c] + BN_EPS);
        scale[c] = (float)s;
        shift[c] = (float)((double)beta[c] - (double)mean[c] * s);
    }
    return MBN_OK;
}

int mbn_weights_free(mbn_weights *w)
{
    if (!w) return MBN_OK;
    free(w->blob);
    w->blob = NULL;
    return MBN_OK;
}

int mbn_weights_from_h5(const char *path, float alpha, int res, mbn_weights *w)
{
    if (!path || !w) return MBN_EINVAL;
    memset(w, 0, sizeof(*w));
    mbn_h5 *h5 = NULL;
    int rc = mbn_h5_open(path, &h5);
    if (rc != MBN_OK) return rc;
    if (res <= 0) res = 224;

    int ndim;
    int64_t shape[8];
    const float *p;
    rc = mbn_h5_get(h5, "/conv1/conv1/kernel:0", &ndim, shape, &p);
    if (rc == MBN_OK && (ndim != 4 || shape[0] != 3 || shape[1] != 3 || shape[2] != 3)) rc = MBN_ESHAPE;
    if (rc == MBN_OK && alpha <= 0.f) alpha = (float)shape[3] / 32.f;
    int classes = 0;
    if (rc == MBN_OK) {
        rc = mbn_h5_get(h5, "/conv_preds/conv_preds/bias:0", &ndim, shape, &p);
        if (rc == MBN_OK) classes = (int)shape_count(ndim, shape);
    }
    if (rc == MBN_OK) rc = mbn_plan_build(alpha, res, classes, &w->plan);
    if (rc == MBN_OK) {
        w->blob = (float *)calloc((size_t)w->plan.blob_floats, sizeof(float));
        if (!w->blob) rc = MBN_ENOMEM;
    }
    int dw = 0, pw = 0;
    for (int i = 0; rc == MBN_OK && i < w->plan.n_layers; i++) {
        const mbn_layer_desc *l = &w->plan.layer[i];
        float *dst = w->blob + l->w_offset;
        char g[64], gbn[64];
        switch (l->kind) {
        case MBN_L_CONV:                                       /* HWIO == [ky][kx][ci][co]: straight copy */
            rc = get_param(h5, "conv1", "kernel:0", l->w_count, &p);
            if (rc == MBN_OK) memcpy(dst, p, sizeof(float) * (size_t)l->w_count);
            if (rc == MBN_OK) rc = fold_bn(h5, "conv1_bn", l->out_ch, w->blob + l->scale_offset, w->blob + l->shift_offset);
            break;
        case MBN_L_DW:                                         /* (3,3,C,1) == [ky][kx][C] */
            dw++;
            snprintf(g, sizeof(g), "conv_dw_%d", dw);
            snprintf(gbn, sizeof(gbn), "conv_dw_%d_bn", dw);
            rc = get_param(h5, g, "depthwise_kernel:0", l->w_count, &p);
            if (rc == MBN_OK) memcpy(dst, p, sizeof(float) * (size_t)l->w_count);
            if (rc == MBN_OK) rc = fold_bn(h5, gbn, l->out_ch, w->blob + l->scale_offset, w->blob + l->shift_offset);
            break;
        case MBN_L_PW:                                         /* (1,1,Cin,Cout) -> [Cout][Cin] (kernel.cl order) */
            pw++;
            snprintf(g, sizeof(g), "conv_pw_%d", pw);
            snprintf(gbn, sizeof(gbn), "conv_pw_%d_bn", pw);
            rc = get_param(h5, g, "kernel:0", l->w_count, &p);
            if (rc == MBN_OK)
                for (int ci = 0; ci < l->in_ch; ci++)
                    for (int co = 0; co < l->out_ch; co++) dst[(size_t)co * l->in_ch + ci] = p[(size_t)ci * l->out_ch + co];
            if (rc == MBN_OK) rc = fold_bn(h5, gbn, l->out_ch, w->blob + l->scale_offset, w->blob + l->shift_offset);
            break;
        case MBN_L_FC:
            rc = get_param(h5, "conv_preds", "kernel:0", l->w_count, &p);
            if (rc == MBN_OK)
                for (int ci = 0; ci < l->in_ch; ci++)
                    for (int co = 0; co < l->out_ch; co++) dst[(size_t)co * l->in_ch + ci] = p[(size_t)ci * l->out_ch + co];
            if (rc == MBN_OK) rc = get_param(h5, "conv_preds", "bias:0", l->out_ch, &p);
            if (rc == MBN_OK) memcpy(w->blob + l->shift_offset, p, sizeof(float) * (size_t)l->out_ch);
            break;
        default:
            break;
        }
    }
    mbn_h5_close(h5);
    if (rc != MBN_OK) mbn_weights_free(w);
    return rc;
}

/* ------------------------------------------------------------------ synthetic generator */
typedef struct { uint64_t s; int have; double spare; } rng_t;

static uint64_t rng_next(rng_t *r)                      /* splitmix64 */
{
    uint64_t z = (r->s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static double rng_uniform(rng_t *r) { return (double)(rng_next(r) >> 11) * (1.0 / 9007199254740992.0); }
static double rng_normal(rng_t *r)
{
    if (r->have) { r->have = 0; return r->spare; }
    double u1 = rng_uniform(r), u2 = rng_uniform(r);
    if (u1 < 1e-300) u1 = 1e-300;
    double m = sqrt(-2.0 * log(u1));
    r->spare = m * sin(6.283185307179586 * u2);
    r->have = 1;
    return m * cos(6.283185307179586 * u2);
}

static int put_normal(mbn_h5_writer *hw, rng_t *r, const char *group, const char *leaf, int ndim, const int64_t *shape,
                      double mean, double std)
{
    char name[256];
    int64_t n = shape_count(ndim, shape);
    float *buf = (float *)malloc(sizeof(float) * (size_t)(n ? n : 1));
    if (!buf) return MBN_ENOMEM;
    for (int64_t i = 0; i < n; i++) buf[i] = (float)(mean + std * rng_normal(r));
    snprintf(name, sizeof(name), "/%s/%s/%s", group, group, leaf);
    int rc = mbn_h5_put(hw, name, ndim, shape, buf);
    free(buf);
    return rc;
}

static int put_uniform(mbn_h5_writer *hw, rng_t *r, const char *group, const char *leaf, int64_t n, double lo, double hi)
{
    char name[256];
    float *buf = (float *)malloc(sizeof(float) * (size_t)(n ? n : 1));
    if (!buf) return MBN_ENOMEM;
    for (int64_t i = 0; i < n; i++) buf[i] = (float)(lo + (hi - lo) * rng_uniform(r));
    snprintf(name, sizeof(name), "/%s/%s/%s", group, group, leaf);
    int rc = mbn_h5_put(hw, name, 1, &n, buf);
    free(buf);
    return rc;
}

static int put_bn(mbn_h5_writer *hw, rng_t *r, const char *group, int64_t ch)
{
    int rc;
    if ((rc = put_uniform(hw, r, group, "gamma:0", ch, 0.5, 1.5)) != MBN_OK) return rc;
    if ((rc = put_normal(hw, r, group, "beta:0", 1, &ch, 0.0, 0.1)) != MBN_OK) return rc;
    if ((rc = put_normal(hw, r, group, "moving_mean:0", 1, &ch, 0.0, 0.1)) != MBN_OK) return rc;
    return put_uniform(hw, r, group, "moving_variance:0", ch, 0.5, 1.5);
}

int mbn_weights_synthetic_h5(const char *path, float alpha, int classes, uint64_t seed)
{
    if (!path) return MBN_EINVAL;
    mbn_plan plan;
    int rc = mbn_plan_build(alpha, 224, classes, &plan);
    if (rc != MBN_OK) return rc;
    mbn_h5_writer *hw = NULL;
    if ((rc = mbn_h5_create(path, &hw)) != MBN_OK) return rc;
    rng_t r = { seed, 0, 0.0 };
    int dw = 0, pw = 0;
    for (int i = 0; rc == MBN_OK && i < plan.n_layers; i++) {
        const mbn_layer_desc *l = &plan.layer[i];
        char g[64], gbn[64];
        if (l->kind == MBN_L_CONV) {
            int64_t shp[4] = { 3, 3, 3, l->out_ch };
            rc = put_normal(hw, &r, "conv1", "kernel:0", 4, shp, 0.0, sqrt(2.0 / 27.0));
            if (rc == MBN_OK) rc = put_bn(hw, &r, "conv1_bn", l->out_ch);
        } else if (l->kind == MBN_L_DW) {
            dw++;
            snprintf(g, sizeof(g), "conv_dw_%d", dw);
            snprintf(gbn, sizeof(gbn), "conv_dw_%d_bn", dw);
            int64_t shp[4] = { 3, 3, l->out_ch, 1 };
            rc = put_normal(hw, &r, g, "depthwise_kernel:0", 4, shp, 0.0, sqrt(2.0 / 9.0));
            if (rc == MBN_OK) rc = put_bn(hw, &r, gbn, l->out_ch);
        } else if (l->kind == MBN_L_PW) {
            pw++;
            snprintf(g, sizeof(g), "conv_pw_%d", pw);
            snprintf(gbn, sizeof(gbn), "conv_pw_%d_bn", pw);
            int64_t shp[4] = { 1, 1, l->in_ch, l->out_ch };
            rc = put_normal(hw, &r, g, "kernel:0", 4, shp, 0.0, sqrt(2.0 / (double)l->in_ch));
            if (rc == MBN_OK) rc = put_bn(hw, &r, gbn, l->out_ch);
        } else if (l->kind == MBN_L_FC) {
            int64_t shp[4] = { 1, 1, l->in_ch, l->out_ch };
            int64_t nb = l->out_ch;
            rc = put_normal(hw, &r, "conv_preds", "kernel:0", 4, shp, 0.0, sqrt(1.0 / (double)l->in_ch));
            if (rc == MBN_OK) rc = put_normal(hw, &r, "conv_preds", "bias:0", 1, &nb, 0.0, 0.1);
        }
    }
    int rc2 = mbn_h5_finish(hw);
    return rc != MBN_OK ? rc : rc2;
}
