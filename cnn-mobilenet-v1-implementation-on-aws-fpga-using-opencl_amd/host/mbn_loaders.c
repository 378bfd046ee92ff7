/*
 * mbn_loaders.c — host-side loaders of the path, plain C.
 *
 * Keeps the reference's two loader symbols with their exact signatures (MobileNet.c:31 readSquezeNetKernel,
 * MobileNet.c:49 decode_image) and adds checked forms that return MBN_E* instead of crashing on a missing file
 * (the reference never checks fopen: MobileNet.c:37,52). Also: RGB de-interleave (MobileNet.c:218-238),
 * a PPM reader that skips the header (the reference reads the header bytes as pixels: B14), and the host
 * softmax/argmax of MobileNet.c:2771-2792.
 */
#include <ctype.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mbn.h"

#define MBN_REF_HEIGHT 224
#define MBN_REF_WIDTH 224

/* One whitespace-delimited token -> double, as the reference does with fscanf("%s") + atof (MobileNet.c:41-42).
 * Tokens longer than the reference's 255-byte buffer are truncated, not overflowed. */
static int next_token_double(FILE *fp, double *v)
{
    char buff[256];
    if (fscanf(fp, "%255s", buff) != 1) return 0;
    *v = atof(buff);
    return 1;
}

int mbn_read_text_weights(const char *path, int *m, int read_size)
{
    if (!path || !m || read_size < 0) return MBN_EINVAL;
    FILE *fp = fopen(path, "r");
    if (!fp) return MBN_EIO;
    for (int i = 0; i < read_size; i++) {
        double n;
        if (!next_token_double(fp, &n)) { fclose(fp); return MBN_EFORMAT; }   /* fewer tokens than asked */
        m[i] = (int)n;                    /* double -> int truncation, MobileNet.c:43 (B8) */
    }
    fclose(fp);
    return MBN_OK;
}

int mbn_read_text_weights_f32(const char *path, float *m, size_t read_size, size_t skip)
{
    if (!path || !m) return MBN_EINVAL;
    FILE *fp = fopen(path, "r");
    if (!fp) return MBN_EIO;
    double n;
    for (size_t i = 0; i < skip; i++)
        if (!next_token_double(fp, &n)) { fclose(fp); return MBN_EFORMAT; }
    for (size_t i = 0; i < read_size; i++) {
        if (!next_token_double(fp, &n)) { fclose(fp); return MBN_EFORMAT; }
        m[i] = (float)n;
    }
    fclose(fp);
    return MBN_OK;
}

/* MobileNet.c:31-47 — same name, same signature, same file name, same "re-open and read the first read_size
 * tokens every call" behaviour. A missing or short file leaves the remaining entries untouched. */
void readSquezeNetKernel(int *m, int read_size)
{
    (void)mbn_read_text_weights("weights_c.txt", m, read_size);
}

/* MobileNet.c:49-57 — raw read of 224*224*3 bytes from offset 0 (PPM header bytes included, B14).
 * Returns 0 like the reference on success; MBN_EIO when the file cannot be opened or is short. */
int decode_image(unsigned char frame[], char filename[])
{
    if (!frame || !filename) return MBN_EINVAL;
    FILE *fp = fopen(filename, "r");
    if (!fp) return MBN_EIO;
    size_t want = (size_t)MBN_REF_HEIGHT * MBN_REF_WIDTH * 3;
    size_t got = fread(frame, 1, want, fp);
    fclose(fp);
    return got == want ? 0 : MBN_EIO;
}

static int ppm_next_int(FILE *fp, int *v)
{
    int c;
    for (;;) {                              /* skip whitespace and '#' comments */
        c = fgetc(fp);
        if (c == EOF) return 0;
        if (c == '#') {
            while ((c = fgetc(fp)) != EOF && c != '\n') {}
            continue;
        }
        if (!isspace(c)) break;
    }
    if (!isdigit(c)) return 0;
    long n = 0;
    while (c != EOF && isdigit(c)) {
        n = n * 10 + (c - '0');
        if (n > 1 << 24) return 0;
        c = fgetc(fp);
    }
    /* the single whitespace byte after the last header field has now been consumed */
    *v = (int)n;
    return 1;
}

int mbn_read_ppm(const char *path, unsigned char *rgb, int *width, int *height, int max_pixels)
{
    if (!path || !rgb || !width || !height || max_pixels <= 0) return MBN_EINVAL;
    FILE *fp = fopen(path, "rb");
    if (!fp) return MBN_EIO;
    int rc = MBN_EFORMAT, w = 0, h = 0, maxv = 0;
    if (fgetc(fp) == 'P' && fgetc(fp) == '6' && ppm_next_int(fp, &w) && ppm_next_int(fp, &h) &&
        ppm_next_int(fp, &maxv) && w > 0 && h > 0 && maxv > 0 && maxv < 256) {
        if ((long)w * h > max_pixels) rc = MBN_EINVAL;
        else {
            size_t want = (size_t)w * h * 3;
            rc = fread(rgb, 1, want, fp) == want ? MBN_OK : MBN_EIO;
            *width = w;
            *height = h;
        }
    }
    fclose(fp);
    return rc;
}

int mbn_write_ppm(const char *path, const unsigned char *rgb, int width, int height)
{
    if (!path || !rgb || width <= 0 || height <= 0) return MBN_EINVAL;
    FILE *fp = fopen(path, "wb");
    if (!fp) return MBN_EIO;
    fprintf(fp, "P6\n%d %d\n255\n", width, height);
    size_t want = (size_t)width * height * 3;
    int rc = fwrite(rgb, 1, want, fp) == want ? MBN_OK : MBN_EIO;
    fclose(fp);
    return rc;
}

/* MobileNet.c:218-238: interleaved HWC bytes -> three planes. */
int mbn_split_rgb(const unsigned char *hwc, int pixels, unsigned char *r, unsigned char *g, unsigned char *b)
{
    if (!hwc || !r || !g || !b || pixels < 0) return MBN_EINVAL;
    for (int i = 0; i < pixels; i++) {
        r[i] = hwc[3 * i];
        g[i] = hwc[3 * i + 1];
        b[i] = hwc[3 * i + 2];
    }
    return MBN_OK;
}

/* MobileNet.c:2771-2792: exp() in double over the uint8 logits, normalise, arg-max (first maximum wins, 1-based).
 * A logit is one of 256 values, so exp is taken once per distinct value instead of twice per class as the reference
 * does; the sum still runs over the classes in index order, which keeps the reference's rounding. */
int mbn_softmax_argmax_u8(const unsigned char *logits, int n, double *probs, int *location, double *maximum)
{
    if (!logits || !probs || n <= 0) return MBN_EINVAL;
    double e[256];
    unsigned char seen[256] = { 0 };
    double sum = 0.0;
    int best = 0;
    for (int k = 0; k < n; k++) {
        const unsigned v = logits[k];
        if (!seen[v]) { e[v] = exp((double)v); seen[v] = 1; }
        sum += e[v];
        if (v > logits[best]) best = k;                  /* exp is monotone: the arg-max of the logits is the arg-max of the probabilities */
    }
    const double inv_free_sum = sum;                     /* divide (not multiply by a reciprocal): same quotient as the reference */
    for (int k = 0; k < n; k++) probs[k] = e[logits[k]] / inv_free_sum;
    if (location) *location = best + 1;
    if (maximum) *maximum = probs[best];
    return MBN_OK;
}
