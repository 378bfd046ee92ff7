/*
 * mobilenet_main.c — command-line C host; the counterpart of the reference's ./out (MobileNet.c main, :113-2833).
 *
 *   mobilenet --h5 weights.h5 [--ppm image.ppm] [--batch N] [--res 224] [--alpha 1.0]      fp32 path
 *   mobilenet --synthetic SEED [--alpha A] [--res R] [--batch N]                              fp32, synthetic weights
 *   mobilenet --literal [--weights weights_c.txt] [--image Cat_Image0.ppm] [--ref-args]      the reference's own mode
 *   mobilenet --gpus G --batch N [--steps K --warmup W --streams S --pw-emul 6] (--h5 F | --synthetic SEED)  N images sharded over G GPUs
 *   mobilenet --inspect weights.h5                                                             list the datasets of a .h5
 *   mobilenet --convert weights.h5 out.txt                                                     folded blob as text (one %.9g per line)
 *
 * --literal runs the 29 uint8/int32 layers exactly in MobileNet.c's order through the NULL-ext C-ABI calls, with the
 * reference's loaders (readSquezeNetKernel re-reads the same file prefix for every layer, decode_image keeps the PPM
 * header bytes). --ref-args additionally passes the reference's own argument literals where they differ from the
 * intended network (filtersize = K_P = 1 for pointwise, MobileNet.c:451; stride 2 at layer 26, :2432).
 * Prints the same two kinds of line as the reference: per-layer kernel time (MobileNet.c:315) and the argmax
 * line (MobileNet.c:2792).
 */
#include <pthread.h>
#include <stdio.h>
#include <unistd.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "mbn.h"

#define CHECK(call)                                                                                   \
    do {                                                                                              \
        int _rc = (call);                                                                             \
        if (_rc != MBN_OK) {                                                                          \
            fprintf(stderr, "Error: %s -> %s (%s)\n", #call, mbn_strerror(_rc),                       \
                    ctx ? mbn_last_device_error(ctx) : "");                                           \
            return 1;                                                                                 \
        }                                                                                             \
    } while (0)

static int run_literal(mbn_context *ctx, const char *wfile, const char *image, int ref_args)
{
    mbn_plan plan;
    CHECK(mbn_plan_build(1.0f, 224, 1000, &plan));
    static unsigned char frame[224 * 224 * 3];
    unsigned char *r = malloc(224 * 224), *g = malloc(224 * 224), *b = malloc(224 * 224);
    int *filter = malloc(sizeof(int) * 1024 * 1024);
    if (!r || !g || !b || !filter) return 1;
    memset(frame, 0, sizeof(frame));
    if (decode_image(frame, (char *)image) != 0) fprintf(stderr, "warning: %s not readable; using a zero image\n", image);
    mbn_split_rgb(frame, 224 * 224, r, g, b);
    size_t act_bytes = 112 * 112 * 64;
    void *d_r, *d_g, *d_b, *d_f, *d_act[2];
    CHECK(mbn_alloc(ctx, 224 * 224, &d_r));
    CHECK(mbn_alloc(ctx, 224 * 224, &d_g));
    CHECK(mbn_alloc(ctx, 224 * 224, &d_b));
    CHECK(mbn_alloc(ctx, sizeof(int) * 1024 * 1024, &d_f));
    CHECK(mbn_alloc(ctx, act_bytes, &d_act[0]));
    CHECK(mbn_alloc(ctx, act_bytes, &d_act[1]));
    CHECK(mbn_upload(ctx, d_r, r, 224 * 224));
    CHECK(mbn_upload(ctx, d_g, g, 224 * 224));
    CHECK(mbn_upload(ctx, d_b, b, 224 * 224));
    CHECK(mbn_set_profiling(ctx, 1));
    const void *src = NULL;
    int which = 0;
    for (int i = 0; i < plan.n_layers; i++) {
        const mbn_layer_desc *l = &plan.layer[i];
        void *dst = d_act[which];
        int count = (int)l->w_count;
        if (l->kind != MBN_L_POOL) {
            memset(filter, 0, sizeof(int) * (size_t)count);
            int rc = mbn_read_text_weights(wfile, filter, count);     /* same prefix every layer, MobileNet.c:37 */
            if (rc != MBN_OK && i == 0) fprintf(stderr, "warning: %s: %s; missing weights read as 0\n", wfile, mbn_strerror(rc));
            CHECK(mbn_upload(ctx, d_f, filter, sizeof(int) * (size_t)count));
        }
        switch (l->kind) {
        case MBN_L_CONV:
            CHECK(mbn_convolute(ctx, dst, d_r, d_g, d_b, d_f, 224, 224, 3, 2, l->out_ch, NULL));
            break;
        case MBN_L_DW: {
            int stride = (ref_args && l->index == 26) ? 2 : l->stride;
            CHECK(mbn_depthwise(ctx, dst, src, d_f, l->out_rows, l->out_cols, 3, stride, l->out_ch, NULL));
            break;
        }
        case MBN_L_PW:
        case MBN_L_FC:
            CHECK(mbn_pointwise(ctx, dst, src, d_f, l->out_rows, l->out_cols, ref_args ? 1 : l->in_ch, l->out_ch, NULL));
            break;
        case MBN_L_POOL:
            CHECK(mbn_pool(ctx, dst, src, l->in_rows, l->in_cols, 7, l->out_ch, NULL));
            break;
        }
        float ms = 0.f;
        CHECK(mbn_last_kernel_ms(ctx, &ms));
        if (l->kind == MBN_L_FC) printf("Kernel Execution time for Fully Connected Layer: %f\n", ms / 1000.0);
        else printf("Kernel Execution time for Layer %d: %f\n", l->index, ms / 1000.0);
        src = dst;
        which ^= 1;
    }
    unsigned char logits[1000];
    double probs[1000], maximum;
    int location;
    CHECK(mbn_download(ctx, logits, src, 1000));
    mbn_softmax_argmax_u8(logits, 1000, probs, &location, &maximum);
    printf("Highest Probability of the element is present at location %d and it's value is %f.\n", location, maximum);
    free(r); free(g); free(b); free(filter);
    return 0;
}

static int inspect_cb(const char *path, int ndim, const int64_t *shape, void *user)
{
    long *total = (long *)user, n = 1;
    printf("%-56s (", path);
    for (int i = 0; i < ndim; i++) { printf(i ? ", %lld" : "%lld", (long long)shape[i]); n *= (long)shape[i]; }
    printf(")\n");
    *total += n;
    return 0;
}

/* Weight-format tooling (no GPU needed): what keras.py:1-8 was meant to do, done by parsing the HDF5 file. */
static int inspect_h5(const char *path)
{
    mbn_h5 *h5 = NULL;
    int rc = mbn_h5_open(path, &h5);
    if (rc != MBN_OK) { fprintf(stderr, "Error: %s: %s\n", path, mbn_strerror(rc)); return 1; }
    long total = 0;
    rc = mbn_h5_visit(h5, inspect_cb, &total);
    mbn_h5_close(h5);
    if (rc != MBN_OK) { fprintf(stderr, "Error: walking %s: %s\n", path, mbn_strerror(rc)); return 1; }
    printf("%ld float32 parameters\n", total);
    return 0;
}

static int convert_h5(const char *path, const char *out)
{
    mbn_weights w;
    int rc = mbn_weights_from_h5(path, 0.f, 224, &w);
    if (rc != MBN_OK) { fprintf(stderr, "Error: %s: %s\n", path, mbn_strerror(rc)); return 1; }
    FILE *fp = fopen(out, "w");
    if (!fp) { mbn_weights_free(&w); return 1; }
    fprintf(fp, "# mbn folded blob: alpha %g classes %d floats %lld\n", w.plan.alpha, w.plan.classes, (long long)w.plan.blob_floats);
    for (int64_t i = 0; i < w.plan.blob_floats; i++) fprintf(fp, "%.9g\n", w.blob[i]);
    fclose(fp);
    printf("alpha %g, %d classes, %lld floats -> %s\n", w.plan.alpha, w.plan.classes, (long long)w.plan.blob_floats, out);
    mbn_weights_free(&w);
    return 0;
}

/* ---------------------------------------------------------------------------------------------- --gpus G
 * The multi-GPU form of the host (BASELINE config 4): the batch is cut into G contiguous shards
 * (mbn_shard_range), GPU 0 receives the parameter blob from the host and ONE RCCL broadcast (mbn_dist_broadcast) hands it
 * to the other GPUs over xGMI, then one host thread per GPU runs the identical single-GPU pipeline on its shard. No
 * collective on the data path. Timing: W untimed steps, a thread barrier, K steps, device sync, barrier; the job's time
 * is the slowest GPU's (MAX), images/s = K * N / that. The reference drives exactly one device (MobileNet.c:155). */
typedef struct {
    int rank, world, batch, first, count, res, steps, warmup;
    mbn_dist *dist;
    const mbn_plan *plan;
    void *dev_blob;
    double seconds;
    int top1;            /* class of this shard's first image */
    float *logits;       /* host copy of the shard's logits after the timed steps [count][classes] (--verify, checksum) */
    unsigned long long fnv;
} gpu_job;

static double now_s(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

static int g_streams = 1;      /* --streams S: sub-batch streams per GPU in --gpus mode (mbn_net_set_streams) */

/* image i of the whole batch is generated from its GLOBAL index, so a shard sees the same pixels as the same images would in
 * a single-GPU run (sharding property: concat of shards == the unsharded batch) */
static void synth_images(unsigned char *u8, int first, int n, size_t img)
{
    for (int i = 0; i < n; i++) {
        unsigned long long s = 0xC0FFEEULL + (unsigned long long)(first + i) * 0x9E3779B97F4A7C15ULL;
        unsigned char *p = u8 + (size_t)i * img;
        for (size_t k = 0; k < img; k++) { s = s * 6364136223846793005ULL + 1442695040888963407ULL; p[k] = (unsigned char)(s >> 56); }
    }
}

static unsigned long long fnv1a(const void *p, size_t n)
{
    const unsigned char *b = (const unsigned char *)p;
    unsigned long long h = 0xcbf29ce484222325ULL;
    for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 0x100000001b3ULL; }
    return h;
}

/* the forward of images [first, first + n) on `ctx`, exactly as a rank runs it (same net settings, same call sizes): used by the
 * ranks and by --verify, which replays every shard on GPU 0 */
static int shard_forward(mbn_context *ctx, const mbn_plan *plan, void *dev_blob, int res, int first, int n, int warmup, int steps,
                         float *host_logits, int *top1, double *seconds, mbn_rank_sync *sync)
{
    mbn_net *net = NULL;
    void *d_u8 = NULL, *d_logits = NULL, *d_idx = NULL, *d_prob = NULL;
    unsigned char *u8 = NULL;
    const int nn = n > 0 ? n : 1;                             /* a rank without images still takes part in the barriers */
    const size_t img = (size_t)res * res * 3;
    int rc = mbn_net_create_from_device_blob(ctx, plan, dev_blob, nn, &net);
    if (rc == MBN_OK) rc = mbn_net_set_input_u8(net, 1);
    if (rc == MBN_OK && g_streams > 1) {
        rc = mbn_net_set_streams(net, g_streams);
        if (rc == MBN_OK) rc = mbn_net_set_free_running(net, 1);   /* the images are uploaded with the blocking mbn_upload below */
    }
    if (rc == MBN_OK) rc = mbn_alloc(ctx, (size_t)nn * img, &d_u8);
    if (rc == MBN_OK) rc = mbn_alloc(ctx, (size_t)nn * plan->classes * sizeof(float), &d_logits);
    if (rc == MBN_OK) rc = mbn_alloc(ctx, sizeof(int) * (size_t)nn, &d_idx);
    if (rc == MBN_OK) rc = mbn_alloc(ctx, sizeof(float) * (size_t)nn, &d_prob);
    if (rc == MBN_OK && !(u8 = malloc((size_t)nn * img))) rc = MBN_ENOMEM;
    if (rc == MBN_OK) {
        synth_images(u8, first, nn, img);
        rc = mbn_upload(ctx, d_u8, u8, (size_t)nn * img);
    }
    for (int w = 0; rc == MBN_OK && w < warmup && n > 0; w++) rc = mbn_net_forward(net, d_u8, d_logits, nn, 0);
    if (rc == MBN_OK) rc = mbn_sync(ctx);
    if (sync && (rc != MBN_OK ? (mbn_rank_fail(sync), 1) : mbn_rank_barrier(sync) != MBN_OK)) rc = rc != MBN_OK ? rc : MBN_EDEVICE;
    const double t0 = now_s();
    for (int k = 0; rc == MBN_OK && k < steps && n > 0; k++) rc = mbn_net_forward(net, d_u8, d_logits, nn, 0);
    if (rc == MBN_OK) rc = mbn_sync(ctx);
    if (seconds) *seconds = now_s() - t0;
    if (sync && rc == MBN_OK && mbn_rank_barrier(sync) != MBN_OK) rc = MBN_EDEVICE;
    if (rc == MBN_OK && n > 0) {
        int idx = -1;
        rc = mbn_softmax_topk_f32(ctx, NULL, d_idx, d_prob, d_logits, nn, plan->classes, 1, NULL);
        if (rc == MBN_OK) rc = mbn_download(ctx, &idx, d_idx, sizeof(int));
        if (top1) *top1 = idx + 1;
        if (rc == MBN_OK && host_logits) rc = mbn_download(ctx, host_logits, d_logits, (size_t)n * plan->classes * sizeof(float));
    }
    if (net) mbn_net_destroy(net);
    if (d_u8) mbn_free(ctx, d_u8);
    if (d_logits) mbn_free(ctx, d_logits);
    if (d_idx) mbn_free(ctx, d_idx);
    if (d_prob) mbn_free(ctx, d_prob);
    free(u8);
    return rc;
}

static int gpu_rank(int rank, void *arg, mbn_rank_sync *sync)
{
    gpu_job *j = &((gpu_job *)arg)[rank];
    mbn_context *ctx = NULL;
    int rc = mbn_dist_context(j->dist, j->rank, &ctx);
    if (rc == MBN_OK)
        rc = shard_forward(ctx, j->plan, j->dev_blob, j->res, j->first, j->count, j->warmup, j->steps, j->logits, &j->top1, &j->seconds, sync);
    if (rc == MBN_OK && j->count > 0) j->fnv = fnv1a(j->logits, (size_t)j->count * j->plan->classes * sizeof(float));
    if (rc != MBN_OK) fprintf(stderr, "Error: GPU %d: %s (%s)\n", j->rank, mbn_strerror(rc), ctx ? mbn_last_device_error(ctx) : "");
    return rc;
}

/* --gpus G [--verify]. Every failure leaves through `done`, which shuts the communicators and contexts down. */
static int run_multi(int gpus, int batch, int res, int steps, int warmup, const mbn_weights *w, int verify)
{
    mbn_context *ctx = NULL;
    mbn_dist *dist = NULL;
    int bad = 1;
    int rc = mbn_dist_init(gpus, NULL, &dist);
    if (rc != MBN_OK) {
        fprintf(stderr, "Error: mbn_dist_init(%d) -> %s%s\n", gpus, mbn_strerror(rc),
                rc == MBN_ENODEVICE ? " (MBN_ENODEVICE: --gpus asks for more GPUs than this node shows)" : "");
        return 1;
    }
    const size_t blob_bytes = (size_t)w->plan.blob_floats * sizeof(float);
    void **blobs = calloc((size_t)gpus, sizeof(void *));
    gpu_job *jobs = calloc((size_t)gpus, sizeof(gpu_job));
    if (!blobs || !jobs) goto done;
#define TRY(call) do { rc = (call); if (rc != MBN_OK) { fprintf(stderr, "Error: %s -> %s (%s)\n", #call, mbn_strerror(rc), ctx ? mbn_last_device_error(ctx) : ""); goto done; } } while (0)
    for (int r = 0; r < gpus; r++) {
        TRY(mbn_dist_context(dist, r, &ctx));
        TRY(mbn_alloc(ctx, blob_bytes, &blobs[r]));
    }
    TRY(mbn_dist_context(dist, 0, &ctx));
    TRY(mbn_upload(ctx, blobs[0], w->blob, blob_bytes));        /* host -> GPU 0 once ... */
    rc = mbn_dist_broadcast(dist, blobs, blob_bytes, 0);         /* ... GPU 0 -> everybody over xGMI: the path's one collective */
    if (rc != MBN_OK) { fprintf(stderr, "Error: mbn_dist_broadcast -> %s (%s)\n", mbn_strerror(rc), mbn_dist_last_error(dist)); goto done; }
    for (int r = 0; r < gpus; r++) {
        gpu_job *j = &jobs[r];
        j->rank = r; j->world = gpus; j->batch = batch; j->res = res; j->steps = steps; j->warmup = warmup;
        j->dist = dist; j->plan = &w->plan; j->dev_blob = blobs[r];
        TRY(mbn_shard_range(batch, gpus, r, &j->first, &j->count));
        j->logits = malloc((size_t)(j->count > 0 ? j->count : 1) * w->plan.classes * sizeof(float));
        if (!j->logits) goto done;
    }
    rc = mbn_run_ranks(gpus, gpu_rank, jobs, -1, NULL);          /* one thread per GPU, gate + cancellable barrier (mbn_ranks.c) */
    if (rc != MBN_OK) { fprintf(stderr, "Error: a rank failed: %s\n", mbn_strerror(rc)); goto done; }
    double worst = 0.0;
    for (int r = 0; r < gpus; r++)
        if (jobs[r].seconds > worst) worst = jobs[r].seconds;
    for (int r = 0; r < gpus; r++)
        printf("GPU %d: images [%d, %d) %d steps in %.6f s; first image -> class %d; logits fnv1a %016llx\n", r, jobs[r].first,
               jobs[r].first + jobs[r].count, steps, jobs[r].seconds, jobs[r].top1, jobs[r].fnv);
    printf("%d GPUs, batch %d (%s), %d steps: %.1f images/sec (slowest GPU %.6f s, %.3f ms/step)\n", gpus, batch,
           "contiguous shards, weights broadcast once over RCCL", steps, (double)steps * batch / worst, worst,
           1000.0 * worst / steps);
    bad = 0;
    if (verify) {
        /* every shard again on GPU 0 — same images (global index), same call size, GPU 0's own copy of the blob — compared bit for
         * bit with what the shard's GPU produced: a wrong broadcast, a wrong shard offset or a GPU that computes differently shows */
        int differ = 0;
        bad = 1;                                                /* until every shard has compared identical (a failed TRY below left 0) */
        TRY(mbn_dist_context(dist, 0, &ctx));
        for (int r = 0; r < gpus; r++) {
            if (jobs[r].count <= 0) continue;
            const size_t nb = (size_t)jobs[r].count * w->plan.classes * sizeof(float);
            float *again = malloc(nb);
            if (!again) goto done;
            rc = shard_forward(ctx, &w->plan, blobs[0], res, jobs[r].first, jobs[r].count, 0, 1, again, NULL, NULL, NULL);
            const int same = rc == MBN_OK && memcmp(again, jobs[r].logits, nb) == 0;
            printf("verify: shard %d (GPU %d) vs the same images on GPU 0: %s (fnv1a %016llx)\n", r, r,
                   rc != MBN_OK ? mbn_strerror(rc) : same ? "identical" : "DIFFERENT", rc == MBN_OK ? fnv1a(again, nb) : 0ULL);
            free(again);
            if (!same) differ = 1;
        }
        bad = differ;
    }
#undef TRY
done:
    if (jobs)
        for (int r = 0; r < gpus; r++) free(jobs[r].logits);
    mbn_dist_shutdown(dist);                                    /* frees every buffer the contexts own, the communicators and the contexts */
    free(blobs); free(jobs);
    return bad;
}

int main(int argc, char **argv)
{
    if (argc == 3 && !strcmp(argv[1], "--inspect")) return inspect_h5(argv[2]);
    if (argc == 4 && !strcmp(argv[1], "--convert")) return convert_h5(argv[2], argv[3]);
    const char *h5 = NULL, *ppm = NULL, *wfile = "weights_c.txt", *image = "Cat_Image0.ppm";
    int literal = 0, ref_args = 0, batch = 1, res = 224, have_seed = 0, gpus = 0, steps = 20, warmup = 3, verify = 0;
    unsigned long long seed = 0;
    float alpha = 0.f;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--h5") && i + 1 < argc) h5 = argv[++i];
        else if (!strcmp(argv[i], "--ppm") && i + 1 < argc) ppm = argv[++i];
        else if (!strcmp(argv[i], "--weights") && i + 1 < argc) wfile = argv[++i];
        else if (!strcmp(argv[i], "--image") && i + 1 < argc) image = argv[++i];
        else if (!strcmp(argv[i], "--batch") && i + 1 < argc) batch = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--res") && i + 1 < argc) res = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--alpha") && i + 1 < argc) alpha = (float)atof(argv[++i]);
        else if (!strcmp(argv[i], "--synthetic") && i + 1 < argc) { seed = strtoull(argv[++i], NULL, 0); have_seed = 1; }
        else if (!strcmp(argv[i], "--gpus") && i + 1 < argc) gpus = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--steps") && i + 1 < argc) steps = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--streams") && i + 1 < argc) g_streams = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--pw-emul") && i + 1 < argc) {          /* opt-in arithmetic form of the pointwise layers (mbn.h: pw_emul) */
            if (mbn_tune_set("pw_emul", atoi(argv[++i])) != MBN_OK || mbn_tune_set("pw_emul_static", 1) != MBN_OK) { fprintf(stderr, "bad --pw-emul\n"); return 2; }
        }
        else if (!strcmp(argv[i], "--warmup") && i + 1 < argc) warmup = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--verify")) verify = 1;
        else if (!strcmp(argv[i], "--literal")) literal = 1;
        else if (!strcmp(argv[i], "--ref-args")) ref_args = 1;
        else {
            fprintf(stderr, "usage: %s [--h5 F | --synthetic SEED | --literal] [--ppm F] [--batch N] [--res R] [--alpha A] "
                            "[--gpus G [--steps K] [--warmup W] [--streams S] [--pw-emul 6] [--verify]]\n", argv[0]);
            return 2;
        }
    }
    mbn_context *ctx = NULL;
    if (gpus > 0 && !literal) {
        if (batch < 1 || steps < 1 || warmup < 0 || g_streams < 1 || g_streams > 8) { fprintf(stderr, "bad --batch/--steps/--warmup/--streams\n"); return 2; }
        char tmpm[] = "/tmp/mbn_synth_XXXXXX.h5";
        if (!h5) {
            if (!have_seed) { fprintf(stderr, "need --h5 or --synthetic with --gpus\n"); return 2; }
            int tfd = mkstemps(tmpm, 3);
            if (tfd < 0) { fprintf(stderr, "cannot create a temporary file in /tmp\n"); return 2; }
            close(tfd);
            CHECK(mbn_weights_synthetic_h5(tmpm, alpha > 0.f ? alpha : 1.0f, 1000, seed));
            h5 = tmpm;
        }
        mbn_weights wm;
        CHECK(mbn_weights_from_h5(h5, alpha, res, &wm));
        if (h5 == tmpm) remove(tmpm);
        int rc = run_multi(gpus, batch, res, steps, warmup, &wm, verify);
        mbn_weights_free(&wm);
        return rc;
    }
    printf("Initializing HIP device...\n");
    CHECK(mbn_init(0, &ctx));
    char name[128];
    mbn_device_name(ctx, name, sizeof(name));
    printf("device: %s  (%s)\n", name, mbn_version());
    if (literal) {
        int rc = run_literal(ctx, wfile, image, ref_args);
        mbn_shutdown(ctx);
        return rc;
    }
    char tmp[] = "/tmp/mbn_synth_XXXXXX.h5";
    if (!h5) {
        if (!have_seed) { fprintf(stderr, "need --h5, --synthetic or --literal\n"); return 2; }
        int tfd = mkstemps(tmp, 3);                 /* O_EXCL-created, unpredictable name: no symlink following, no clash between runs */
        if (tfd < 0) { fprintf(stderr, "cannot create a temporary file in /tmp\n"); return 2; }
        close(tfd);
        CHECK(mbn_weights_synthetic_h5(tmp, alpha > 0.f ? alpha : 1.0f, 1000, seed));
        h5 = tmp;
    }
    mbn_weights w;
    CHECK(mbn_weights_from_h5(h5, alpha, res, &w));
    if (have_seed) remove(tmp);
    mbn_net *net = NULL;
    CHECK(mbn_net_create(ctx, &w, batch, &net));
    const int classes = w.plan.classes;
    size_t img_count = (size_t)batch * res * res * 3;
    unsigned char *u8 = malloc(img_count);
    if (!u8) return 1;
    int pw = 0, ph = 0;
    if (ppm && mbn_read_ppm(ppm, u8, &pw, &ph, res * res) == MBN_OK && pw == res && ph == res) {
        for (int n = 1; n < batch; n++) memcpy(u8 + (size_t)n * res * res * 3, u8, (size_t)res * res * 3);
    } else {
        if (ppm) fprintf(stderr, "warning: %s is not a %dx%d P6 image; using a synthetic one\n", ppm, res, res);
        unsigned long long s = 0xC0FFEEULL;
        for (size_t i = 0; i < img_count; i++) { s = s * 6364136223846793005ULL + 1442695040888963407ULL; u8[i] = (unsigned char)(s >> 56); }
    }
    void *d_u8, *d_logits, *d_probs, *d_arg;
    CHECK(mbn_alloc(ctx, img_count, &d_u8));
    CHECK(mbn_alloc(ctx, (size_t)batch * classes * sizeof(float), &d_logits));
    CHECK(mbn_alloc(ctx, (size_t)batch * classes * sizeof(float), &d_probs));
    CHECK(mbn_alloc(ctx, (size_t)batch * 8 * sizeof(int), &d_arg));
    CHECK(mbn_upload(ctx, d_u8, u8, img_count));
    CHECK(mbn_net_set_input_u8(net, 1));   /* layer 1 reads the raw image and applies the Keras x/127.5 - 1 at load */
    float ms[MBN_MAX_LAYERS];
    CHECK(mbn_net_forward_timed(net, d_u8, d_logits, batch, ms, MBN_MAX_LAYERS));
    for (int i = 0; i < w.plan.n_layers; i++) {
        if (w.plan.layer[i].kind == MBN_L_FC) printf("Kernel Execution time for Fully Connected Layer: %f\n", ms[i] / 1000.0);
        else printf("Kernel Execution time for Layer %d: %f\n", i + 1, ms[i] / 1000.0);
    }
    /* classifier read-out on the device: only the 5 best (index, probability) pairs per image cross PCIe, instead of the
     * reference's 1000 logits + host exp loop (MobileNet.c:2744-2792) */
    const int topk = classes < 5 ? classes : 5;
    CHECK(mbn_softmax_topk_f32(ctx, NULL, d_arg, d_probs, d_logits, batch, classes, topk, NULL));
    int *arg = malloc(sizeof(int) * (size_t)batch * topk);
    float *probs = malloc(sizeof(float) * (size_t)batch * topk);
    if (!arg || !probs) return 1;
    CHECK(mbn_download(ctx, arg, d_arg, sizeof(int) * (size_t)batch * topk));
    CHECK(mbn_download(ctx, probs, d_probs, sizeof(float) * (size_t)batch * topk));
    printf("Highest Probability of the element is present at location %d and it's value is %f.\n", arg[0] + 1, probs[0]);
    printf("top-%d:", topk);
    for (int j = 0; j < topk; j++) printf(" %d (%f)", arg[j] + 1, probs[j]);
    printf("\n");
    mbn_net_destroy(net);
    mbn_weights_free(&w);
    mbn_shutdown(ctx);
    free(u8); free(arg); free(probs);
    return 0;
}
