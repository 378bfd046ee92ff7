/*
 * mbn_net.c — the C host that sequences MobileNet-V1 through the C-ABI (include/mbn.h).
 *
 * Counterpart of the reference's main(): MobileNet.c:240-2763 is 29 hand-unrolled blocks, each doing
 * clCreateKernel -> readSquezeNetKernel -> clCreateBuffer x3 -> blocking H2D x2 -> clSetKernelArg ->
 * clEnqueueNDRangeKernel -> clFinish -> blocking D2H (template at MobileNet.c:322-403). Here the same
 * layer order is a loop over the plan; weights are uploaded once, activations ping-pong between two buffers
 * that stay in HBM, every call is asynchronous on the context's stream, and nothing crosses PCIe between layers.
 *
 * Plain C: this file uses nothing but the functions declared in mbn.h.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "mbn.h"

struct mbn_net {
    mbn_context *ctx;
    mbn_plan plan;
    int max_batch;
    int num_cus;               /* compute units of the context's GPU (small-batch fusion rule) */
    void *dev_blob;
    int own_blob;
    void *act[2];
    int keep;
    int dtype;                 /* MBN_DT_F32 or MBN_DT_BF16 */
    int fuse_stem;             /* mbn_net_set_fuse_stem (default 1) */
    int input_u8;              /* mbn_net_set_input_u8: images are raw uint8 HWC */
    unsigned fuse_blocks;      /* mbn_net_set_fuse_blocks: bit L = run the depthwise layer L and the pointwise layer L+1 as one launch */
    int fuse_blocks_set;       /* the caller chose a mask; otherwise the default of the current dtype applies */
    int use_graph;             /* mbn_net_set_graph */
    void *graph;               /* instantiated hipGraph of one forward, valid for the key below */
    int g_emul;                /* pw_emul value the graph was captured under */
    const void *g_images;
    void *g_logits;
    int g_batch, g_last, g_dtype, g_keep;
    int free_running;          /* no fork dependency on the context's stream (mbn_net_set_free_running) */
    const void *fr_images;     /* free-running: the (images, logits, batch, last_layer) of the previous forward; a call that */
    void *fr_logits;           /* differs re-inserts the fork wait, because sub-batch slices of act[] move with the batch */
    int fr_batch, fr_last, fr_ns, fr_stagger;
    int nstreams;              /* sub-batch pipelining (mbn_net_set_streams); 1 = everything on the context's stream */
    void *streams[8];
    void *bf16_filt[MBN_MAX_LAYERS];   /* bf16 copies of the pointwise / FC filters (bf16 mode) */
    unsigned char bf16_packed[MBN_MAX_LAYERS];   /* the copy is followed by its packed image (mbn_pack_filter_bf16) */
    void *keep_buf[MBN_MAX_LAYERS];
    void *logits_buf;          /* mbn_net_classify: [max_batch][classes] fp32 */
    void *poolfc_ws;           /* workspace of mbn_pool_fc (1...4 images: pool + FC in one launch), allocated and zeroed at creation */
    size_t poolfc_ws_bytes;
    int fuse_tail;             /* mbn_net_set_fuse_tail (default 0) */
    int fuse_resident;         /* mbn_net_set_fuse_resident (default 1): runs of equal bf16 blocks on a small map as one launch, the map resident in LDS */
    void *last_out[MBN_MAX_LAYERS];
};

static const float *blob_at(const mbn_net *net, int64_t off)
{
    return off < 0 ? NULL : (const float *)net->dev_blob + off;
}

static int net_alloc_common(mbn_context *ctx, const mbn_plan *plan, int max_batch, mbn_net **out)
{
    if (!ctx || !plan || !out || max_batch <= 0 || plan->n_layers <= 0 || plan->n_layers > MBN_MAX_LAYERS)
        return MBN_EINVAL;
    mbn_net *net = (mbn_net *)calloc(1, sizeof(*net));
    if (!net) return MBN_ENOMEM;
    net->ctx = ctx;
    net->dtype = MBN_DT_F32;
    net->nstreams = 1;
    net->fuse_stem = 1;
    net->fuse_blocks = MBN_FUSE_BLOCKS_DEFAULT;
    net->plan = *plan;
    net->max_batch = max_batch;
    net->num_cus = 256;
    (void)mbn_device_cus(ctx, &net->num_cus);
    size_t bytes = (size_t)plan->max_act_floats * (size_t)max_batch * sizeof(float);
    int rc = mbn_alloc(ctx, bytes, &net->act[0]);
    if (rc == MBN_OK) rc = mbn_alloc(ctx, bytes, &net->act[1]);
    net->fuse_tail = 0;            /* measured slower than the two launches (mbn.h: mbn_net_set_fuse_tail): opt-in */
    net->fuse_resident = 1;
    if (rc == MBN_OK && plan->n_layers >= 2 && plan->layer[plan->n_layers - 2].kind == MBN_L_POOL &&
        plan->layer[plan->n_layers - 1].kind == MBN_L_FC) {
        const mbn_layer_desc *fc = &plan->layer[plan->n_layers - 1];
        net->poolfc_ws_bytes = mbn_pool_fc_workspace_bytes(fc->in_ch, fc->out_ch);     /* 0: channels outside the kernel's envelope */
        if (net->poolfc_ws_bytes) {
            rc = mbn_alloc(ctx, net->poolfc_ws_bytes, &net->poolfc_ws);
            if (rc == MBN_OK) rc = mbn_memset(ctx, net->poolfc_ws, 0, net->poolfc_ws_bytes);
        }
    }
    if (rc != MBN_OK) { mbn_net_destroy(net); return rc; }
    *out = net;
    return MBN_OK;
}

int mbn_net_create(mbn_context *ctx, const mbn_weights *w, int max_batch, mbn_net **out)
{
    if (!w || !w->blob || !out) return MBN_EINVAL;
    *out = NULL;
    mbn_net *net = NULL;
    int rc = net_alloc_common(ctx, &w->plan, max_batch, &net);
    if (rc != MBN_OK) return rc;
    size_t bytes = (size_t)w->plan.blob_floats * sizeof(float);
    rc = mbn_alloc(ctx, bytes, &net->dev_blob);
    if (rc == MBN_OK) { net->own_blob = 1; rc = mbn_upload(ctx, net->dev_blob, w->blob, bytes); }   /* once, not per layer */
    if (rc != MBN_OK) { mbn_net_destroy(net); return rc; }
    *out = net;
    return MBN_OK;
}

int mbn_net_create_from_device_blob(mbn_context *ctx, const mbn_plan *plan, const void *dev_blob, int max_batch,
                                    mbn_net **out)
{
    if (!dev_blob || !out) return MBN_EINVAL;
    *out = NULL;
    mbn_net *net = NULL;
    int rc = net_alloc_common(ctx, plan, max_batch, &net);
    if (rc != MBN_OK) return rc;
    net->dev_blob = (void *)dev_blob;
    net->own_blob = 0;
    *out = net;
    return MBN_OK;
}

int mbn_net_destroy(mbn_net *net)
{
    if (!net) return MBN_OK;
    mbn_sync(net->ctx);
    if (net->graph) mbn_graph_destroy(net->ctx, net->graph);
    if (net->logits_buf) mbn_free(net->ctx, net->logits_buf);
    if (net->poolfc_ws) mbn_free(net->ctx, net->poolfc_ws);
    for (int j = 0; j < 8; j++)
        if (net->streams[j]) mbn_stream_destroy(net->ctx, net->streams[j]);
    for (int i = 0; i < MBN_MAX_LAYERS; i++) {
        if (net->keep_buf[i]) mbn_free(net->ctx, net->keep_buf[i]);
        if (net->bf16_filt[i]) mbn_free(net->ctx, net->bf16_filt[i]);
    }
    if (net->act[0]) mbn_free(net->ctx, net->act[0]);
    if (net->act[1]) mbn_free(net->ctx, net->act[1]);
    if (net->own_blob && net->dev_blob) mbn_free(net->ctx, net->dev_blob);
    else if (net->dev_blob) mbn_forget(net->ctx, net->dev_blob, (size_t)net->plan.blob_floats * sizeof(float));   /* images derived from the caller's blob */
    free(net);
    return MBN_OK;
}

int mbn_net_plan(const mbn_net *net, mbn_plan *plan)
{
    if (!net || !plan) return MBN_EINVAL;
    *plan = net->plan;
    return MBN_OK;
}

int mbn_net_set_dtype(mbn_net *net, int dtype)
{
    if (!net || (dtype != MBN_DT_F32 && dtype != MBN_DT_BF16)) return MBN_EINVAL;
    if (dtype == MBN_DT_BF16) {
        for (int i = 0; i < net->plan.n_layers; i++) {
            const mbn_layer_desc *l = &net->plan.layer[i];
            if ((l->kind != MBN_L_PW && l->kind != MBN_L_FC) || net->bf16_filt[i]) continue;
            /* lab build only (the wide-tile GEMM measured slower and is not shipped): wide pointwise layers (Cout % 256 == 0,
             * Cin % 64 == 0) carry the packed image of their filter behind the plain bf16 copy (MBN_IO_FILT_PACKED) */
            const size_t pk = (l->kind == MBN_L_PW && mbn_lab_build()) ? mbn_packed_filter_offset(l->out_ch, l->in_ch) : 0;
            int rc = mbn_alloc(net->ctx, pk ? pk + (size_t)l->w_count * 2 : (size_t)l->w_count * 2, &net->bf16_filt[i]);
            if (rc == MBN_OK)
                rc = mbn_convert_f32_to_bf16(net->ctx, net->bf16_filt[i], blob_at(net, l->w_offset), (size_t)l->w_count, NULL);
            if (rc == MBN_OK && pk) rc = mbn_pack_filter_bf16(net->ctx, net->bf16_filt[i], l->out_ch, l->in_ch, NULL);
            if (rc != MBN_OK) return rc;
            net->bf16_packed[i] = pk != 0;
        }
        int rc = mbn_sync(net->ctx);     /* the copies are ready before any (possibly free-running) sub-stream reads them */
        if (rc != MBN_OK) return rc;
    }
    /* kept activations change element size with the dtype: drop them */
    if (dtype != net->dtype) {
        mbn_sync(net->ctx);
        for (int i = 0; i < MBN_MAX_LAYERS; i++)
            if (net->keep_buf[i]) { mbn_free(net->ctx, net->keep_buf[i]); net->keep_buf[i] = NULL; }
    }
    net->dtype = dtype;
    return MBN_OK;
}

int mbn_net_set_streams(mbn_net *net, int n)
{
    if (!net || n < 1 || n > 8) return MBN_EINVAL;
    for (int j = 0; j < n; j++)
        if (!net->streams[j]) {
            int rc = mbn_stream_create(net->ctx, &net->streams[j]);
            if (rc != MBN_OK) return rc;
        }
    net->nstreams = n;
    net->fr_images = NULL;             /* the next multi-stream forward orders itself behind the previous one's tails again */
    return MBN_OK;
}

/* layers 1-3 can go through mbn_stem_fused: conv 3x3 s2 -> dw s1 -> pw with 32 -> 32 -> 64 channels, fp32, nothing kept */
static int stem_fusable(const mbn_net *net, int last_layer)
{
    const mbn_layer_desc *l = net->plan.layer;
    /* the fused kernels load with 16-byte accesses: a caller-provided device blob must be aligned (the plan's segments are
     * 256-byte aligned inside it); mbn_net_launches and the forward both decide here, so they cannot disagree */
    if (((uintptr_t)net->dev_blob % 16) != 0) return 0;
    return net->fuse_stem && !net->keep && last_layer >= 3 && net->plan.n_layers >= 3 &&
           (net->dtype == MBN_DT_F32 || (net->dtype == MBN_DT_BF16 && net->bf16_filt[2])) &&
           l[0].kind == MBN_L_CONV && l[1].kind == MBN_L_DW && l[2].kind == MBN_L_PW && l[0].in_ch == 3 &&
           ((l[0].out_ch == 32 && l[2].out_ch == 64) || (l[0].out_ch == 16 && l[2].out_ch == 32)) && l[1].stride == 1 &&
           (net->plan.res % 32) == 0;
}

int mbn_net_set_fuse_stem(mbn_net *net, int enabled)
{
    if (!net) return MBN_EINVAL;
    if (net->fuse_stem != (enabled != 0) && net->graph) {
        mbn_sync(net->ctx);
        mbn_graph_destroy(net->ctx, net->graph);
        net->graph = NULL;
    }
    net->fuse_stem = enabled != 0;
    return MBN_OK;
}

int mbn_net_fused_layers(const mbn_net *net, int last_layer, int *count)
{
    if (!net || !count) return MBN_EINVAL;
    if (last_layer <= 0 || last_layer > net->plan.n_layers) last_layer = net->plan.n_layers;
    *count = stem_fusable(net, last_layer) ? 3 : 0;
    return MBN_OK;
}

/* the mask in force: the caller's, or the measured default of the current dtype (mbn.h) */
static unsigned fuse_mask(const mbn_net *net)
{
    if (net->fuse_blocks_set) return net->fuse_blocks;
    return net->dtype == MBN_DT_BF16 ? MBN_FUSE_BLOCKS_DEFAULT_BF16 : MBN_FUSE_BLOCKS_DEFAULT;
}

/* layers i+1 (depthwise) and i+2 (pointwise), 0-based index i, can go through mbn_dwpw_fused for `count` images: fp32,
 * nothing kept, enabled in the mask, and inside the kernel's envelope (mbn.h: mbn_dwpw_fused) */
static int block_fusable(const mbn_net *net, int i, int count, int last_layer)
{
    const int bf = net->dtype == MBN_DT_BF16;
    if ((net->dtype != MBN_DT_F32 && !bf) || net->keep || i + 2 > last_layer || i + 1 >= net->plan.n_layers || i + 1 >= 32) return 0;
    if (bf && (!net->bf16_filt[i + 1] || ((net->plan.layer[i].in_ch % 64) != 0 && net->plan.layer[i].in_ch != 32))) return 0;   /* 32: half a chunk, padded (round 5) */
    if (!((fuse_mask(net) >> (i + 1)) & 1u)) return 0;
    /* bf16 default: the block kernel recomputes the depthwise chunk once per 256-column tile and is bound by that VALU work,
     * so a block wider than one tile measures slower fused than as two launches; an explicit mask overrides */
    if (bf && !net->fuse_blocks_set && net->plan.layer[i + 1].out_ch > 256) return 0;
    /* default, fp32: a block kernel workgroup walks 128 output pixels through the WHOLE block (all depthwise chunks, then
     * K/2 matrix instructions per chunk, one after the other), so when the launch has fewer 128 x 128 output tiles than half
     * the compute units the two separate launches — more, shorter workgroups — are faster (batch 1: -28 us per forward,
     * break-even at batch 4-8 for blocks 4-7 and 16 for blocks 8-11: profiles/r02/i_small_batch.txt). fp32 results are
     * bit-identical either way, so the choice may depend on the batch. In bf16 the fused blocks stay (measured faster at every
     * batch from 1 up: the stand-alone bf16 pointwise has no few-tile form). */
    if (!bf && !net->fuse_blocks_set) {
        const mbn_layer_desc *q = &net->plan.layer[i + 1];
        const long tiles = (((long)count * q->out_rows * q->out_cols + 127) / 128) * ((q->out_ch + 127) / 128);
        if (tiles * 2 < net->num_cus) return 0;
    }
    if (((uintptr_t)net->dev_blob % 16) != 0) return 0;            /* see stem_fusable */
    const mbn_layer_desc *d = &net->plan.layer[i], *p = &net->plan.layer[i + 1];
    if (d->kind != MBN_L_DW || p->kind != MBN_L_PW || (d->stride != 1 && d->stride != 2)) return 0;
    if (d->in_ch < 32 || (d->in_ch % 32) != 0 || d->in_ch > 1024 || p->out_ch > 1024) return 0;
    if (bf ? (p->out_ch < 64 || (p->out_ch % 64) != 0) : (p->out_ch < 128 || (p->out_ch % 128) != 0)) return 0;   /* bf16 (round 5): 64-column remainders on a padded tile */
    if ((d->out_cols & 1) || p->in_ch != d->out_ch) return 0;
    const double es = bf ? 2.0 : 4.0;
    if (es * count * d->in_rows * d->in_cols * d->in_ch >= 4026531840.0) return 0;
    if (es * count * p->out_rows * p->out_cols * p->out_ch >= 4294967296.0) return 0;
    return 1;
}

/* Blocks starting at layer index i (0-based; depthwise, pointwise, depthwise, ...) that can run as ONE launch with the map resident in LDS
 * (mbn_blocks_resident_bf16, round 6): bf16, each of them fusable under the mask in force, stride 1 with pad 1, 256 channels in and out, equal small maps.
 * Returns the number of blocks (2 ... 8) or 0. The five 256 -> 256 blocks on the 10 x 10 map of the 0.5x160 network (layers 14-23). */
static int resident_run(const mbn_net *net, int i, int count, int last_layer)
{
    if (net->dtype != MBN_DT_BF16 || !net->fuse_resident || net->keep) return 0;
    int k = 0;
    while (k < 8 && block_fusable(net, i + 2 * k, count, last_layer)) {
        const mbn_layer_desc *d = &net->plan.layer[i + 2 * k], *p = &net->plan.layer[i + 2 * k + 1], *d0 = &net->plan.layer[i];
        if (d->stride != 1 || d->pad_top != 1 || d->pad_left != 1 || d->in_ch != 256 || p->out_ch != 256 || d->out_rows != d->in_rows ||
            d->out_cols != d->in_cols || d->in_rows != d0->in_rows || d->in_cols != d0->in_cols || d->in_rows * d->in_cols > 104 ||
            (d->in_rows + 2) * (d->in_cols + 2) > 144)
            break;
        k++;
    }
    return k >= 2 ? k : 0;
}

/* Layers i+1 ... i+5 (0-based index i) are the network's last two blocks and the global pool in the shape mbn_tail_resident_bf16 takes (round 6): bf16, nothing
 * kept, resident launches enabled, both blocks enabled in the mask in force; depthwise stride 2 without top / left padding on an even map of at most 10 x 10 x 256,
 * pointwise 256 -> 512, depthwise stride 1 with pad 1, pointwise 512 -> 512, pool over the whole map (layers 24-28 of the 0.5x160 network). */
static int tail_run(const mbn_net *net, int i, int count, int last_layer)
{
    (void)count;
    if (net->dtype != MBN_DT_BF16 || !net->fuse_resident || net->keep || i + 5 > last_layer || i + 5 > net->plan.n_layers || i + 4 >= 32) return 0;
    const mbn_layer_desc *d0 = &net->plan.layer[i], *p0 = &net->plan.layer[i + 1], *d1 = &net->plan.layer[i + 2], *p1 = &net->plan.layer[i + 3],
                         *po = &net->plan.layer[i + 4];
    if (d0->kind != MBN_L_DW || p0->kind != MBN_L_PW || d1->kind != MBN_L_DW || p1->kind != MBN_L_PW || po->kind != MBN_L_POOL) return 0;
    if (!net->bf16_filt[i + 1] || !net->bf16_filt[i + 3]) return 0;
    const unsigned mask = fuse_mask(net);
    if (!((mask >> (i + 1)) & 1u) || !((mask >> (i + 3)) & 1u)) return 0;
    if (((uintptr_t)net->dev_blob % 16) != 0) return 0;
    if (d0->stride != 2 || d0->pad_top != 0 || d0->pad_left != 0 || d0->in_ch != 256 || p0->out_ch != 512 || (d0->in_rows & 1) || (d0->in_cols & 1) ||
        d0->in_rows > 10 || d0->in_cols > 10 || d0->out_rows != d0->in_rows / 2 || d0->out_cols != d0->in_cols / 2)
        return 0;
    if (d1->stride != 1 || d1->pad_top != 1 || d1->pad_left != 1 || d1->in_ch != 512 || p1->out_ch != 512 || d1->out_rows != d0->out_rows ||
        d1->out_cols != d0->out_cols)
        return 0;
    if (po->in_rows != d0->out_rows || po->in_cols != d0->out_cols || po->out_ch != 512) return 0;
    return 1;
}

/* layers i+1 (pool) and i+2 (FC), 0-based index i, as one launch: fp32, 1...4 images, nothing kept, the last two layers of the call */
static int tail_fusable(const mbn_net *net, int i, int count, int last_layer)
{
    if (!net->fuse_tail || !net->poolfc_ws || net->dtype != MBN_DT_F32 || net->keep || count < 1 || count > 4) return 0;
    if (i + 2 != last_layer || last_layer != net->plan.n_layers) return 0;
    return net->plan.layer[i].kind == MBN_L_POOL && net->plan.layer[i + 1].kind == MBN_L_FC;
}

/* a captured graph bakes the launch list in: drop it when the fusion settings change */
static void drop_graph(mbn_net *net)
{
    if (net->graph) {
        mbn_sync(net->ctx);
        mbn_graph_destroy(net->ctx, net->graph);
        net->graph = NULL;
    }
}

int mbn_net_set_input_u8(mbn_net *net, int enabled)
{
    if (!net) return MBN_EINVAL;
    if (net->input_u8 != (enabled != 0)) drop_graph(net);
    net->input_u8 = enabled != 0;
    return MBN_OK;
}

int mbn_net_set_fuse_blocks(mbn_net *net, unsigned mask)
{
    if (!net) return MBN_EINVAL;
    /* an explicit mask also switches off the default-only rules of block_fusable (few tiles in fp32, wide blocks in bf16),
     * so the launch list can change even when the mask's value does not */
    if (fuse_mask(net) != mask || !net->fuse_blocks_set) drop_graph(net);
    net->fuse_blocks = mask;
    net->fuse_blocks_set = 1;
    return MBN_OK;
}

int mbn_net_set_fuse_tail(mbn_net *net, int enabled)
{
    if (!net) return MBN_EINVAL;
    if (net->fuse_tail != (enabled != 0)) drop_graph(net);
    net->fuse_tail = enabled != 0;
    return MBN_OK;
}

int mbn_net_set_fuse_resident(mbn_net *net, int enabled)
{
    if (!net) return MBN_EINVAL;
    if (net->fuse_resident != (enabled != 0)) drop_graph(net);
    net->fuse_resident = enabled != 0;
    return MBN_OK;
}

int mbn_net_reset_fuse_blocks(mbn_net *net)
{
    if (!net) return MBN_EINVAL;
    if (net->fuse_blocks_set) drop_graph(net);
    net->fuse_blocks_set = 0;
    net->fuse_blocks = MBN_FUSE_BLOCKS_DEFAULT;
    return MBN_OK;
}

int mbn_net_get_fuse_blocks(const mbn_net *net, unsigned *mask)
{
    if (!net || !mask) return MBN_EINVAL;
    *mask = fuse_mask(net);
    return MBN_OK;
}

int mbn_net_launches(const mbn_net *net, int batch, int last_layer, int *first_layer, int *n_layers, int capacity, int *count)
{
    if (!net || !count || batch <= 0) return MBN_EINVAL;
    if (last_layer <= 0 || last_layer > net->plan.n_layers) last_layer = net->plan.n_layers;
    int ns = net->nstreams;
    if (batch < 5 * ns) ns = 1;                        /* as forward_impl */
    const int sub = ns > 1 ? batch / ns + (batch % ns ? 1 : 0) : batch;      /* the largest sub-batch decides the envelope */
    int n = 0, i = 0;
    while (i < last_layer) {
        int span = 1;
        if (i == 0 && stem_fusable(net, last_layer)) span = 3;
        else if (resident_run(net, i, sub, last_layer)) span = 2 * resident_run(net, i, sub, last_layer);
        else if (tail_run(net, i, sub, last_layer)) span = 5;
        else if (block_fusable(net, i, sub, last_layer)) span = 2;
        else if (tail_fusable(net, i, sub, last_layer)) span = 2;
        if (first_layer && n_layers && n < capacity) { first_layer[n] = i + 1; n_layers[n] = span; }
        n++;
        i += span;
    }
    *count = n;
    return MBN_OK;
}

int mbn_net_set_graph(mbn_net *net, int enabled)
{
    if (!net) return MBN_EINVAL;
    net->use_graph = enabled != 0;
    if (!enabled && net->graph) {
        mbn_sync(net->ctx);
        mbn_graph_destroy(net->ctx, net->graph);
        net->graph = NULL;
    }
    return MBN_OK;
}

int mbn_net_set_free_running(mbn_net *net, int enabled)
{
    if (!net) return MBN_EINVAL;
    net->free_running = enabled != 0;
    net->fr_images = NULL;
    return MBN_OK;
}

int mbn_net_set_keep_activations(mbn_net *net, int keep)
{
    if (!net) return MBN_EINVAL;
    net->keep = keep != 0;
    return MBN_OK;
}

int mbn_net_layer_output(mbn_net *net, int index, void **dptr, size_t *floats_per_image)
{
    if (!net || index < 1 || index > net->plan.n_layers || !dptr) return MBN_EINVAL;
    const mbn_layer_desc *l = &net->plan.layer[index - 1];
    *dptr = net->last_out[index - 1];
    if (floats_per_image) *floats_per_image = (size_t)l->out_rows * l->out_cols * l->out_ch;
    return *dptr ? MBN_OK : MBN_ENOTFOUND;
}

/* One layer through the C-ABI: the positional arguments are kernel.cl's (see mbn.h). */
static int run_layer(mbn_net *net, const mbn_layer_desc *l, const void *src, void *dst, int batch, void *stream)
{
    const int bf = net->dtype == MBN_DT_BF16;
    mbn_layer_ext ext;
    memset(&ext, 0, sizeof(ext));
    ext.struct_size = sizeof(ext);
    ext.batch = batch;
    ext.dtype = net->dtype;
    ext.layout = MBN_LAYOUT_NHWC;
    ext.act = MBN_ACT_RELU6;
    ext.pad_top = l->pad_top;
    ext.pad_left = l->pad_left;
    ext.scale = blob_at(net, l->scale_offset);
    ext.shift = blob_at(net, l->shift_offset);
    ext.stream = stream;
    const void *filt = blob_at(net, l->w_offset);
    if (bf && (l->kind == MBN_L_PW || l->kind == MBN_L_FC)) filt = net->bf16_filt[l->index - 1];
    switch (l->kind) {
    case MBN_L_CONV:                       /* MobileNet.c:268-292: rows/cols = input size, stride 2 */
        ext.cin = l->in_ch;
        if (bf) ext.io_flags = MBN_IO_IN_F32;          /* the normalised image is fp32 */
        if (net->input_u8) ext.io_flags = MBN_IO_IN_U8; /* raw image, normalised at load */
        return mbn_convolute(net->ctx, dst, src, NULL, NULL, filt, l->in_rows, l->in_cols, 3, l->stride, l->out_ch, &ext);
    case MBN_L_DW:                         /* MobileNet.c:326-381 */
        ext.in_rows = l->in_rows;
        ext.in_cols = l->in_cols;
        return mbn_depthwise(net->ctx, dst, src, filt, l->out_rows, l->out_cols, 3, l->stride, l->out_ch, &ext);
    case MBN_L_PW:                         /* MobileNet.c:417-470, with filtersize = true Cin (B3) */
        if (bf && net->bf16_packed[l->index - 1]) ext.io_flags |= MBN_IO_FILT_PACKED;
        return mbn_pointwise(net->ctx, dst, src, filt, l->out_rows, l->out_cols, l->in_ch, l->out_ch, &ext);
    case MBN_L_POOL:                       /* MobileNet.c:2603-2656 */
        ext.act = MBN_ACT_NONE;
        return mbn_pool(net->ctx, dst, src, l->in_rows, l->in_cols, l->in_rows, l->out_ch, &ext);
    case MBN_L_FC:                         /* MobileNet.c:2682-2739: pointwise with rows = cols = 1; bias, no ReLU (B15) */
        ext.act = MBN_ACT_NONE;
        if (bf) ext.io_flags = MBN_IO_OUT_F32;         /* logits stay fp32 */
        return mbn_pointwise(net->ctx, dst, src, filt, 1, 1, l->in_ch, l->out_ch, &ext);
    default:
        return MBN_EINVAL;
    }
}

/* bytes per element of layer i's output in the current mode (FC logits are fp32 in both) */
static size_t out_esize(const mbn_net *net, const mbn_layer_desc *l)
{
    return (net->dtype == MBN_DT_BF16 && l->kind != MBN_L_FC) ? 2 : 4;
}

/* Layers 1..last_layer for images [first, first+count) on `stream` (NULL = the context's stream). Sub-batches use
 * disjoint slices of the two ping-pong buffers, so several of them can be in flight on different streams. */
static int forward_range(mbn_net *net, const void *images, void *logits, int first, int count, int last_layer,
                         void *stream, float *layer_ms, int n_layer_ms, void *next_stream, int stagger)
{
    const size_t img_floats = (size_t)net->plan.res * net->plan.res * 3;
    const char *src = (const char *)images + (size_t)first * img_floats * (net->input_u8 ? 1 : sizeof(float));
    const size_t slot = (size_t)first * (size_t)net->plan.max_act_floats * sizeof(float);
    int which = 0, i0 = 0;
    if (!layer_ms && stem_fusable(net, last_layer)) {
        /* layers 1-3 in one kernel; the 112x112x32 intermediates stay on chip */
        const mbn_layer_desc *l = net->plan.layer;
        const int bf = net->dtype == MBN_DT_BF16;
        const size_t per_img = (size_t)l[2].out_rows * l[2].out_cols * l[2].out_ch * (bf ? 2 : sizeof(float));
        char *dst = last_layer == 3 ? (char *)logits + (size_t)first * per_img : (char *)net->act[which] + slot;
        int rc = mbn_stem_fused_ex(net->ctx, dst, src, blob_at(net, l[0].w_offset), blob_at(net, l[0].scale_offset),
                                   blob_at(net, l[0].shift_offset), blob_at(net, l[1].w_offset), blob_at(net, l[1].scale_offset),
                                   blob_at(net, l[1].shift_offset), bf ? net->bf16_filt[2] : blob_at(net, l[2].w_offset),
                                   blob_at(net, l[2].scale_offset), blob_at(net, l[2].shift_offset), count, net->plan.res,
                                   l[0].out_ch, l[2].out_ch, (net->input_u8 ? MBN_STEM_IN_U8 : 0) | (bf ? MBN_STEM_BF16 : 0), stream);
        if (rc == MBN_OK) {
            if (last_layer != 3) which ^= 1;
            if (first == 0) { net->last_out[0] = net->last_out[1] = NULL; net->last_out[2] = dst; }
            src = dst;
            i0 = 3;
            if (next_stream && stagger >= 1 && stagger <= 3) {
                rc = mbn_stream_wait(net->ctx, next_stream, stream);
                if (rc != MBN_OK) return rc;
            }
        } else if (rc != MBN_EUNSUPPORTED) return rc;
    }
    for (int i = i0; i < last_layer; i++) {
        const mbn_layer_desc *l = &net->plan.layer[i];
        const int rr = layer_ms ? 0 : resident_run(net, i, count, last_layer);
        if (rr) {
            /* rr blocks in one launch, the map resident in LDS from the first block's input to the last block's output */
            mbn_block_params bp[8];
            for (int k = 0; k < rr; k++) {
                const mbn_layer_desc *d = &net->plan.layer[i + 2 * k], *p = &net->plan.layer[i + 2 * k + 1];
                bp[k].wd = blob_at(net, d->w_offset); bp[k].s2 = blob_at(net, d->scale_offset); bp[k].b2 = blob_at(net, d->shift_offset);
                bp[k].wp_bf16 = net->bf16_filt[i + 2 * k + 1]; bp[k].s3 = blob_at(net, p->scale_offset); bp[k].b3 = blob_at(net, p->shift_offset);
            }
            const int lastl = i + 2 * rr - 1;                      /* index of the run's last (pointwise) layer */
            const mbn_layer_desc *lp = &net->plan.layer[lastl];
            const size_t per_img2 = (size_t)lp->out_rows * lp->out_cols * lp->out_ch * 2;
            char *dst2 = (lastl == last_layer - 1) ? (char *)logits + (size_t)first * per_img2 : (char *)net->act[which] + slot;
            int rc = mbn_blocks_resident_bf16(net->ctx, dst2, src, bp, rr, count, l->in_rows, l->in_cols, l->in_ch, stream);
            if (rc == MBN_OK) {
                if (lastl != last_layer - 1) which ^= 1;
                if (first == 0) {
                    for (int k = i; k < lastl; k++) net->last_out[k] = NULL;
                    net->last_out[lastl] = dst2;
                }
                src = dst2;
                if (next_stream && stagger > i && stagger <= lastl + 1) {
                    rc = mbn_stream_wait(net->ctx, next_stream, stream);
                    if (rc != MBN_OK) return rc;
                }
                i = lastl;
                continue;
            }
            if (rc != MBN_EUNSUPPORTED) return rc;
        }
        if (!layer_ms && tail_run(net, i, count, last_layer)) {
            /* the last two blocks and the pool in one launch, an image's maps resident in LDS */
            mbn_block_params bp[2];
            for (int k = 0; k < 2; k++) {
                const mbn_layer_desc *d = &net->plan.layer[i + 2 * k], *p = &net->plan.layer[i + 2 * k + 1];
                bp[k].wd = blob_at(net, d->w_offset); bp[k].s2 = blob_at(net, d->scale_offset); bp[k].b2 = blob_at(net, d->shift_offset);
                bp[k].wp_bf16 = net->bf16_filt[i + 2 * k + 1]; bp[k].s3 = blob_at(net, p->scale_offset); bp[k].b3 = blob_at(net, p->shift_offset);
            }
            const int lastl = i + 4;                               /* the pool */
            const mbn_layer_desc *lp = &net->plan.layer[lastl];
            const size_t per_img2 = (size_t)lp->out_ch * 2;
            char *dst2 = (lastl == last_layer - 1) ? (char *)logits + (size_t)first * per_img2 : (char *)net->act[which] + slot;
            int rc = mbn_tail_resident_bf16(net->ctx, dst2, src, bp, count, l->in_rows, l->in_cols, l->in_ch, lp->out_ch, stream);
            if (rc == MBN_OK) {
                if (lastl != last_layer - 1) which ^= 1;
                if (first == 0) {
                    for (int k = i; k < lastl; k++) net->last_out[k] = NULL;
                    net->last_out[lastl] = dst2;
                }
                src = dst2;
                if (next_stream && stagger > i && stagger <= lastl + 1) {
                    rc = mbn_stream_wait(net->ctx, next_stream, stream);
                    if (rc != MBN_OK) return rc;
                }
                i = lastl;
                continue;
            }
            if (rc != MBN_EUNSUPPORTED) return rc;
        }
        if (!layer_ms && block_fusable(net, i, count, last_layer)) {
            /* depthwise + pointwise in one kernel; the depthwise output stays in LDS */
            const mbn_layer_desc *lp = &net->plan.layer[i + 1];
            const int bf = net->dtype == MBN_DT_BF16;
            const size_t per_img2 = (size_t)lp->out_rows * lp->out_cols * lp->out_ch * (bf ? 2 : sizeof(float));
            char *dst2 = (i + 1 == last_layer - 1) ? (char *)logits + (size_t)first * per_img2 : (char *)net->act[which] + slot;
            int rc = (bf ? mbn_dwpw_fused_bf16 : mbn_dwpw_fused)(
                net->ctx, dst2, src, blob_at(net, l->w_offset), blob_at(net, l->scale_offset), blob_at(net, l->shift_offset),
                bf ? net->bf16_filt[i + 1] : blob_at(net, lp->w_offset), blob_at(net, lp->scale_offset),
                blob_at(net, lp->shift_offset), count, l->in_rows, l->in_cols, l->out_rows, l->out_cols, l->in_ch, lp->out_ch,
                l->stride, l->pad_top, l->pad_left, stream);
            if (rc == MBN_OK) {
                if (i + 1 != last_layer - 1) which ^= 1;
                if (first == 0) { net->last_out[i] = NULL; net->last_out[i + 1] = dst2; }
                src = dst2;
                if (next_stream && (i + 1 == stagger || i + 2 == stagger)) {
                    rc = mbn_stream_wait(net->ctx, next_stream, stream);
                    if (rc != MBN_OK) return rc;
                }
                i++;
                continue;
            }
            if (rc != MBN_EUNSUPPORTED) return rc;
        }
        if (!layer_ms && tail_fusable(net, i, count, last_layer)) {
            /* pool + FC in one launch (1...4 images: the forward is launch-bound there) */
            const mbn_layer_desc *fc = &net->plan.layer[i + 1];
            char *dst2 = (char *)logits + (size_t)first * fc->out_ch * sizeof(float);
            int rc = mbn_pool_fc(net->ctx, dst2, src, blob_at(net, fc->w_offset), blob_at(net, fc->shift_offset), count, l->in_rows,
                                 l->in_cols, l->out_ch, fc->out_ch, net->poolfc_ws, net->poolfc_ws_bytes, stream);
            if (rc == MBN_OK) {
                if (first == 0) { net->last_out[i] = NULL; net->last_out[i + 1] = dst2; }
                break;
            }
            if (rc != MBN_EUNSUPPORTED) return rc;
        }
        const size_t per_img = (size_t)l->out_rows * l->out_cols * l->out_ch * out_esize(net, l);
        char *dst;
        if (i == last_layer - 1) dst = (char *)logits + (size_t)first * per_img;
        else if (net->keep) {
            if (!net->keep_buf[i]) {
                size_t bytes = (size_t)l->out_rows * l->out_cols * l->out_ch * sizeof(float) * (size_t)net->max_batch;
                int rc = mbn_alloc(net->ctx, bytes, &net->keep_buf[i]);
                if (rc != MBN_OK) return rc;
            }
            dst = (char *)net->keep_buf[i] + (size_t)first * per_img;
        } else {
            dst = (char *)net->act[which] + slot;
            which ^= 1;
        }
        int rc = run_layer(net, l, src, dst, count, stream);
        if (rc != MBN_OK) return rc;
        if (layer_ms && i < n_layer_ms) {
            rc = mbn_last_kernel_ms(net->ctx, &layer_ms[i]);
            if (rc != MBN_OK) return rc;
        }
        if (first == 0) net->last_out[i] = dst;
        src = dst;
        /* stagger: the next sub-batch's stream may start only when this one has finished `stagger` layers, so the
         * streams run a layer or two apart and unlike kernels (HBM-bound vs MFMA-bound) meet each other */
        if (next_stream && i + 1 == stagger) {
            rc = mbn_stream_wait(net->ctx, next_stream, stream);
            if (rc != MBN_OK) return rc;
        }
    }
    return MBN_OK;
}

static int forward_impl(mbn_net *net, const void *images, void *logits, int batch, int last_layer, float *layer_ms,
                        int n_layer_ms)
{
    if (!net || !images || !logits || batch <= 0 || batch > net->max_batch) return MBN_EINVAL;
    const int n = net->plan.n_layers;
    if (last_layer <= 0 || last_layer > n) last_layer = n;
    int ns = net->nstreams;
    if (layer_ms || batch < 5 * ns) ns = 1;            /* per-layer timing serialises; sub-batches of fewer than 5 images are not
                                                        * forked: they would take the 1..4-image kernels (mbn_f32_pw_splitk.hip), whose
                                                        * summation order differs from the single-stream forward of the whole batch */
    if (ns <= 1) net->fr_images = NULL;                /* a single-stream forward in between: the next multi-stream one forks again */
    if (ns <= 1 && net->use_graph && !layer_ms) {
        /* launch-bound batches: replay the 29 launches as one hipGraph; re-capture when the call's key changes */
        int emul = 0;
        (void)mbn_tune_get("pw_emul", &emul);              /* the arithmetic form of the pointwise layers is baked into the capture */
        if (net->graph && (net->g_images != images || net->g_logits != logits || net->g_batch != batch ||
                           net->g_last != last_layer || net->g_dtype != net->dtype || net->g_keep != net->keep || net->g_emul != emul)) {
            mbn_sync(net->ctx);
            mbn_graph_destroy(net->ctx, net->graph);
            net->graph = NULL;
        }
        if (!net->graph) {
            if (net->keep)                                   /* allocations are not allowed inside a capture */
                for (int i = 0; i < last_layer - 1; i++)
                    if (!net->keep_buf[i]) {
                        const mbn_layer_desc *l = &net->plan.layer[i];
                        size_t bytes = (size_t)l->out_rows * l->out_cols * l->out_ch * sizeof(float) * (size_t)net->max_batch;
                        int rc = mbn_alloc(net->ctx, bytes, &net->keep_buf[i]);
                        if (rc != MBN_OK) return rc;
                    }
            /* one eager pass first: kernels that allocate a workspace on first use (the pre-split filter images of pw_emul) must
             * have done so before the capture, inside which nothing may be allocated */
            int rc = forward_range(net, images, logits, 0, batch, last_layer, NULL, NULL, 0, NULL, 0);
            if (rc != MBN_OK) return rc;
            rc = mbn_graph_begin(net->ctx, NULL);
            if (rc != MBN_OK) return rc;
            rc = forward_range(net, images, logits, 0, batch, last_layer, NULL, NULL, 0, NULL, 0);
            void *g = NULL;
            int rc2 = mbn_graph_end(net->ctx, NULL, &g);     /* always close the capture */
            if (rc != MBN_OK || rc2 != MBN_OK) { if (g) mbn_graph_destroy(net->ctx, g); return rc != MBN_OK ? rc : rc2; }
            net->graph = g;
            net->g_images = images; net->g_logits = logits; net->g_batch = batch; net->g_last = last_layer;
            net->g_dtype = net->dtype; net->g_keep = net->keep; net->g_emul = emul;
        }
        return mbn_graph_launch(net->ctx, net->graph, NULL);
    }
    if (ns <= 1) return forward_range(net, images, logits, 0, batch, last_layer, NULL, layer_ms, n_layer_ms, NULL, 0);
    int stagger = 2;
    (void)mbn_tune_get("net_stagger", &stagger);
    if (stagger > last_layer) stagger = last_layer;
    /* fork: every sub-stream starts after what is already queued on the context's stream (e.g. the producer of
     * `images`); sub-batches are launched stream after stream, which staggers them by a layer or two so that one
     * stream's depthwise (HBM-bound) meets another's pointwise (MFMA-bound); join: the context's stream waits for all. */
    const int q = batch / ns, r = batch % ns;
    int first = 0;
    /* free-running skips the fork only between IDENTICAL consecutive calls: with another batch the sub-batch slices of the
     * ping-pong buffers move, with other images/logits the caller may have queued their producer/consumer on the context's
     * stream — both need the ordering back; another stream count or stagger moves the slices too */
    const int fork = !net->free_running || net->fr_images != images || net->fr_logits != logits || net->fr_batch != batch ||
                     net->fr_last != last_layer || net->fr_ns != ns || net->fr_stagger != stagger;
    net->fr_images = images; net->fr_logits = logits; net->fr_batch = batch; net->fr_last = last_layer;
    net->fr_ns = ns; net->fr_stagger = stagger;
    if (fork && net->free_running)
        for (int j = 0; j < ns; j++) {                       /* the previous free-running forward's tails, on every sub-stream */
            int rc = mbn_stream_wait(net->ctx, NULL, net->streams[j]);
            if (rc != MBN_OK) return rc;
        }
    for (int j = 0; j < ns; j++) {
        const int count = q + (j < r ? 1 : 0);
        int rc = fork ? mbn_stream_wait(net->ctx, net->streams[j], NULL) : MBN_OK;
        if (rc == MBN_OK)
            rc = forward_range(net, images, logits, first, count, last_layer, net->streams[j], NULL, 0,
                               j + 1 < ns ? net->streams[j + 1] : NULL, stagger);
        if (rc != MBN_OK) return rc;
        first += count;
    }
    for (int j = 0; j < ns; j++) {
        int rc = mbn_stream_wait(net->ctx, NULL, net->streams[j]);
        if (rc != MBN_OK) return rc;
    }
    return MBN_OK;
}

int mbn_net_forward(mbn_net *net, const void *images, void *logits, int batch, int last_layer)
{
    return forward_impl(net, images, logits, batch, last_layer, NULL, 0);
}

int mbn_net_classify(mbn_net *net, const void *images, int batch, int k, void *topk_idx_i32, void *topk_prob_f32)
{
    if (!net || !images || !topk_idx_i32 || !topk_prob_f32 || batch <= 0 || batch > net->max_batch || k < 1 || k > 8)
        return MBN_EINVAL;
    const mbn_layer_desc *fc = &net->plan.layer[net->plan.n_layers - 1];
    if (!net->logits_buf) {
        int rc = mbn_alloc(net->ctx, (size_t)net->max_batch * fc->out_ch * sizeof(float), &net->logits_buf);
        if (rc != MBN_OK) return rc;
    }
    int rc = forward_impl(net, images, net->logits_buf, batch, 0, NULL, 0);
    if (rc != MBN_OK) return rc;
    return mbn_softmax_topk_f32(net->ctx, NULL, topk_idx_i32, topk_prob_f32, net->logits_buf, batch, fc->out_ch, k, NULL);
}

int mbn_net_forward_timed(mbn_net *net, const void *images, void *logits, int batch, float *layer_ms, int n_layer_ms)
{
    if (!net || !layer_ms || n_layer_ms <= 0) return MBN_EINVAL;
    int rc = mbn_set_profiling(net->ctx, 1);
    if (rc != MBN_OK) return rc;
    rc = forward_impl(net, images, logits, batch, 0, layer_ms, n_layer_ms);
    mbn_set_profiling(net->ctx, 0);
    return rc;
}
