/*
 * mbn_plan.c — the network topology as data.
 *
 * The reference spells MobileNet-V1 out as compile-time constants and 29 hand-unrolled blocks
 * (MobileNet.c:13-26 `#define FILTER_SIZE_L*`, and the per-layer rows/cols/stride/op_size literals at
 * :268-292, :326-381, ... :2682-2739; tabulated in SURVEY.md §2.1). Here it is one table, parameterised by the
 * width multiplier alpha and the input resolution, which also fixes the layout of the packed parameter blob:
 *
 *   per layer, in order:  [filter][scale (out_ch)][shift (out_ch)]   each segment 256-byte aligned
 *     conv1     filter [3][3][3][C]          (Keras HWIO)
 *     depthwise filter [3][3][C]             (Keras (3,3,C,1))
 *     pointwise filter [Cout][Cin]           (kernel.cl's own [oc][ic] order; Keras (1,1,Cin,Cout) transposed)
 *     FC        filter [classes][Cin], no scale, shift = bias
 *
 * Deviations from the reference's literals, on purpose: layer 13 consumes layer 12
 * (B9), layer 26 has stride 1 (B10), pointwise layers sum over all Cin (B3).
 */
#include <string.h>

#include "mbn.h"

static int64_t align_seg(int64_t x) { return (x + 63) & ~(int64_t)63; }   /* 64 floats = 256 B */

static int same_pad(int in, int out, int k, int stride)
{
    int total = (out - 1) * stride + k - in;
    return total > 0 ? total / 2 : 0;
}

int mbn_plan_build(float alpha, int res, int classes, mbn_plan *plan)
{
    /* output channels of conv1 and of the 13 pointwise layers at alpha = 1 (MobileNet.c:16-25) */
    static const int width[14] = { 32, 64, 128, 128, 256, 256, 512, 512, 512, 512, 512, 512, 1024, 1024 };
    /* strides of the 13 depthwise layers (MobileNet.c: L4 :503, L8 :865, L12 :1219, L24 :2264 are 2) */
    static const int dstride[13] = { 1, 2, 1, 2, 1, 2, 1, 1, 1, 1, 1, 2, 1 };

    if (!plan) return MBN_EINVAL;
    if (!(alpha > 0.f) || alpha > 4.f || res < 32 || res > 4096 || classes <= 0) return MBN_EINVAL;
    /* Odd feature maps are where TF-"SAME" (out = ceil(h/2), pad_top = total/2: what this table computes) and Keras
     * MobileNet (ZeroPadding2D(((0,1),(0,1))) + 'valid': pad_top = 0, out = floor((h-2)/2)+1) disagree — 25 -> 13 vs 12.
     * With res a multiple of 32 every stride-2 layer sees an even map and the two coincide, so that is the supported set. */
    if (res % 32) return MBN_EUNSUPPORTED;
    memset(plan, 0, sizeof(*plan));
    plan->alpha = alpha;
    plan->res = res;
    plan->classes = classes;

    int64_t off = 0, max_act = (int64_t)res * res * 3;
    int n = 0, h = res, ch = 3;

    for (int blk = -1; blk < 13; blk++) {
        if (blk >= 0) {                                   /* depthwise half of block blk */
            mbn_layer_desc *d = &plan->layer[n++];
            d->index = n;
            d->kind = MBN_L_DW;
            d->in_rows = d->in_cols = h;
            d->in_ch = d->out_ch = ch;
            d->stride = dstride[blk];
            d->out_rows = d->out_cols = (h + d->stride - 1) / d->stride;
            d->pad_top = d->pad_left = same_pad(h, d->out_rows, 3, d->stride);
            d->w_offset = off; d->w_count = 9 * (int64_t)ch; off = align_seg(off + d->w_count);
            d->scale_offset = off; off = align_seg(off + ch);
            d->shift_offset = off; off = align_seg(off + ch);
            h = d->out_rows;
            if ((int64_t)h * h * ch > max_act) max_act = (int64_t)h * h * ch;
        }
        int oc = (int)(width[blk + 1] * alpha);           /* Keras: int(filters * alpha) */
        if (oc < 1) return MBN_EINVAL;
        mbn_layer_desc *l = &plan->layer[n++];
        l->index = n;
        l->in_rows = l->in_cols = h;
        l->in_ch = ch;
        l->out_ch = oc;
        if (blk < 0) {                                    /* conv1: 3x3x3, stride 2 (MobileNet.c:123-124,268-292) */
            l->kind = MBN_L_CONV;
            l->stride = 2;
            l->out_rows = l->out_cols = (h + 1) / 2;
            l->pad_top = l->pad_left = same_pad(h, l->out_rows, 3, 2);
            l->w_count = 27 * (int64_t)oc;
        } else {                                          /* pointwise half */
            l->kind = MBN_L_PW;
            l->stride = 1;
            l->out_rows = l->out_cols = h;
            l->w_count = (int64_t)oc * ch;
        }
        l->w_offset = off; off = align_seg(off + l->w_count);
        l->scale_offset = off; off = align_seg(off + oc);
        l->shift_offset = off; off = align_seg(off + oc);
        h = l->out_rows;
        ch = oc;
        if ((int64_t)h * h * ch > max_act) max_act = (int64_t)h * h * ch;
    }

    mbn_layer_desc *p = &plan->layer[n++];                /* L28 global average pool (MobileNet.c:2601-2679) */
    p->index = n;
    p->kind = MBN_L_POOL;
    p->in_rows = p->in_cols = h;
    p->in_ch = p->out_ch = ch;
    p->stride = 1;
    p->out_rows = p->out_cols = 1;
    p->w_offset = off;
    p->scale_offset = p->shift_offset = -1;

    mbn_layer_desc *f = &plan->layer[n++];                /* L29 FC = pointwise with rows=cols=1 (:2681-2763) */
    f->index = n;
    f->kind = MBN_L_FC;
    f->in_rows = f->in_cols = f->out_rows = f->out_cols = 1;
    f->in_ch = ch;
    f->out_ch = classes;
    f->stride = 1;
    f->w_offset = off; f->w_count = (int64_t)classes * ch; off = align_seg(off + f->w_count);
    f->scale_offset = -1;
    f->shift_offset = off; off = align_seg(off + classes);

    plan->n_layers = n;
    plan->blob_floats = off;
    plan->max_act_floats = max_act;
    return MBN_OK;
}

/* Batch sharding of the multi-GPU path: rank's contiguous slice of `total` independent images. Same
 * arithmetic as dist.py shard_range (the torch.distributed form) — tested against each other. */
int mbn_shard_range(int total, int world, int rank, int *first, int *count)
{
    if (!first || !count || world <= 0 || rank < 0 || rank >= world || total < 0) return MBN_EINVAL;
    const int q = total / world, r = total % world;
    *first = rank * q + (rank < r ? rank : r);
    *count = q + (rank < r ? 1 : 0);
    return MBN_OK;
}
