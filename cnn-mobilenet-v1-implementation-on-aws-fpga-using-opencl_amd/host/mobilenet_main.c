/*
 * mobilenet_main.c — command-line C host; the counterpart of the reference's ./out (MobileNet.c main, :113-2833).
 *
 *   mobilenet --h5 weights.h5 [--ppm image.ppm] [--batch N] [--res 224] [--alpha 1.0]      fp32 path
 *   mobilenet --synthetic SEED [--alpha A] [--res R] [--batch N]                              fp32, synthetic weights
 *   mobilenet --literal [--weights weights_c.txt] [--image Cat_Image0.ppm] [--ref-args]      the reference's own mode
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
#include <stdio.h>
#include <unistd.h>
#include <stdlib.h>
#include <string.h>

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

int main(int argc, char **argv)
{
    if (argc == 3 && !strcmp(argv[1], "--inspect")) return inspect_h5(argv[2]);
    if (argc == 4 && !strcmp(argv[1], "--convert")) return convert_h5(argv[2], argv[3]);
    const char *h5 = NULL, *ppm = NULL, *wfile = "weights_c.txt", *image = "Cat_Image0.ppm";
    int literal = 0, ref_args = 0, batch = 1, res = 224, have_seed = 0;
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
        else if (!strcmp(argv[i], "--literal")) literal = 1;
        else if (!strcmp(argv[i], "--ref-args")) ref_args = 1;
        else {
            fprintf(stderr, "usage: %s [--h5 F | --synthetic SEED | --literal] [--ppm F] [--batch N] [--res R] [--alpha A]\n",
                    argv[0]);
            return 2;
        }
    }
    mbn_context *ctx = NULL;
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
