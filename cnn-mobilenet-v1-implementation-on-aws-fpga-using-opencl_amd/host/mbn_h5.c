/*
 * mbn_h5.c — a self-contained reader (and minimal writer) for the subset of HDF5 that Keras weight files use.
 *
 * Role in the path: the reference gets its weights from a text file that nothing in its tree produces
 * (MobileNet.c:31-47 reads "weights_c.txt"), and its only .h5 tool, keras.py:1-8, reinterprets the raw HDF5
 * file bytes as float64 — it is not a parser. This file is the real loader the north star asks for: it walks
 * the HDF5 structures directly (no libhdf5 on the GPU box is assumed), memory-maps the file, and hands out
 * pointers to contiguous little-endian float32 datasets by path name.
 *
 * Supported on read (enough for h5py/Keras `save_weights` output and for libhdf5 with default or "latest"
 * format bounds as long as groups stay compact):
 *   superblock v0/v1/v2/v3; object headers v1 and v2 (with continuation blocks);
 *   groups as symbol tables (B-tree v1 "TREE" + "SNOD" + local heap "HEAP") or as compact Link messages;
 *   dataspace v1/v2 (simple); datatype class 1 IEEE float32 LE; data layout v1-v4 contiguous or compact.
 * Not supported (returns MBN_EUNSUPPORTED): dense groups (fractal heap), chunked/compressed datasets,
 * other datatypes, external/soft links.
 *
 * Writer: superblock v0, one symbol-table group node per group, v1 object headers, contiguous float32 datasets.
 */
#include <errno.h>
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "mbn.h"

#define H5_UNDEF UINT64_MAX
#define H5_MAX_DEPTH 16
#define H5_MAX_PATH 512

struct mbn_h5 {
    int fd;
    const uint8_t *map;
    size_t size;
    uint64_t base;          /* superblock base address */
    int so, sl;             /* size of offsets / lengths */
    uint64_t root_ohdr;     /* root group object header */
    void **copies;          /* aligned copies of datasets whose bytes sit unaligned in the file (compact layout) */
    int ncopies;
};

/* ------------------------------------------------------------------ byte helpers */
static uint64_t rd_le(const uint8_t *p, int n)
{
    uint64_t v = 0;
    for (int i = n - 1; i >= 0; i--) v = (v << 8) | p[i];
    return v;
}

static const uint8_t *at(const mbn_h5 *h, uint64_t addr, uint64_t len)
{
    if (addr == H5_UNDEF || addr > UINT64_MAX - h->base) return NULL;    /* the sum below must not wrap */
    uint64_t a = addr + h->base;
    if (a > h->size || len > h->size - a) return NULL;
    return h->map + a;
}

static uint64_t rd_off(const mbn_h5 *h, const uint8_t *p)
{
    uint64_t v = rd_le(p, h->so);
    if (h->so < 8 && v == ((1ULL << (8 * h->so)) - 1)) return H5_UNDEF;
    return v;
}

/* ------------------------------------------------------------------ object-header message iteration */
typedef int (*msg_cb)(const mbn_h5 *h, int type, const uint8_t *data, size_t size, void *user);

/* v1 block: messages with 8-byte headers, sizes padded to 8. Returns <0 error, >0 stop, 0 continue. */
static int iter_v1_block(const mbn_h5 *h, uint64_t addr, uint64_t len, int *budget, msg_cb cb, void *user, int depth)
{
    const uint8_t *blk = at(h, addr, len);
    if (!blk) return MBN_EFORMAT;
    uint64_t pos = 0;
    while (pos + 8 <= len && *budget > 0) {
        int type = (int)rd_le(blk + pos, 2);
        size_t size = (size_t)rd_le(blk + pos + 2, 2);
        if (pos + 8 + size > len) return MBN_EFORMAT;
        const uint8_t *data = blk + pos + 8;
        (*budget)--;
        if (type == 0x0010) {                                /* continuation */
            if (size < (size_t)(h->so + h->sl) || depth > 64) return MBN_EFORMAT;
            int rc = iter_v1_block(h, rd_off(h, data), rd_le(data + h->so, h->sl), budget, cb, user, depth + 1);
            if (rc) return rc;
        } else if (type != 0) {
            int rc = cb(h, type, data, size, user);
            if (rc) return rc;
        }
        pos += 8 + size;
    }
    return 0;
}

static int iter_v2_block(const mbn_h5 *h, const uint8_t *blk, uint64_t len, int track_order, msg_cb cb, void *user,
                         int depth)
{
    const int mh = 4 + (track_order ? 2 : 0);
    uint64_t pos = 0;
    while (pos + mh <= len) {
        int type = blk[pos];
        size_t size = (size_t)rd_le(blk + pos + 1, 2);
        if (pos + mh + size > len) return MBN_EFORMAT;
        const uint8_t *data = blk + pos + mh;
        if (type == 0x10) {
            if (size < (size_t)(h->so + h->sl) || depth > 64) return MBN_EFORMAT;
            uint64_t caddr = rd_off(h, data), clen = rd_le(data + h->so, h->sl);
            const uint8_t *c = at(h, caddr, clen);
            if (!c || clen < 8 || memcmp(c, "OCHK", 4) != 0) return MBN_EFORMAT;
            int rc = iter_v2_block(h, c + 4, clen - 8, track_order, cb, user, depth + 1);
            if (rc) return rc;
        } else if (type != 0) {
            int rc = cb(h, type, data, size, user);
            if (rc) return rc;
        }
        pos += mh + size;
    }
    return 0;
}

static int iter_messages(const mbn_h5 *h, uint64_t ohdr, msg_cb cb, void *user)
{
    const uint8_t *p = at(h, ohdr, 16);
    if (!p) return MBN_EFORMAT;
    if (memcmp(p, "OHDR", 4) == 0) {                          /* version 2 */
        if (p[4] != 2) return MBN_EFORMAT;
        int flags = p[5];
        uint64_t pos = 6;
        if (flags & 0x20) pos += 16;
        if (flags & 0x10) pos += 4;
        int csz = 1 << (flags & 3);
        const uint8_t *q = at(h, ohdr, pos + csz);
        if (!q) return MBN_EFORMAT;
        uint64_t chunk0 = rd_le(q + pos, csz);
        pos += csz;
        /* an 8-byte size field comes straight from the file: chunk0 + 4 wraps for chunk0 >= 2^64 - 4 and would pass at()
         * with a tiny length while iter_v2_block walks chunk0 bytes */
        if (chunk0 > h->size) return MBN_EFORMAT;
        const uint8_t *blk = at(h, ohdr + pos, chunk0 + 4);
        if (!blk) return MBN_EFORMAT;
        int rc = iter_v2_block(h, blk, chunk0, (flags & 0x04) != 0, cb, user, 0);
        return rc < 0 ? rc : 0;
    }
    if (p[0] != 1) return MBN_EFORMAT;                        /* version 1 */
    int nmsgs = (int)rd_le(p + 2, 2);
    uint64_t hsize = rd_le(p + 8, 4);
    int budget = nmsgs;
    int rc = iter_v1_block(h, ohdr + 16, hsize, &budget, cb, user, 0);
    return rc < 0 ? rc : 0;
}

/* ------------------------------------------------------------------ object classification */
typedef struct {
    int has_stab, has_links, dense, is_dataset;
    uint64_t btree, heap;
    /* dataset */
    int ndim;
    int64_t shape[8];
    int dtype_ok, dtype_seen;
    int layout_class;      /* -1 none, 0 compact, 1 contiguous, 2 chunked */
    uint64_t data_addr, data_size;
    const uint8_t *compact;
} obj_info;

static int info_cb(const mbn_h5 *h, int type, const uint8_t *d, size_t size, void *user)
{
    obj_info *o = (obj_info *)user;
    switch (type) {
    case 0x0011:                                             /* symbol table */
        if (size < (size_t)(2 * h->so)) return MBN_EFORMAT;
        o->has_stab = 1;
        o->btree = rd_off(h, d);
        o->heap = rd_off(h, d + h->so);
        break;
    case 0x0002: {                                           /* link info: dense storage if the heap address is set */
        if (size < 2) return MBN_EFORMAT;
        size_t pos = 2;
        if (d[1] & 1) pos += 8;
        if (pos + 2 * (size_t)h->so > size) return MBN_EFORMAT;
        if (rd_off(h, d + pos) != H5_UNDEF) o->dense = 1;
        o->has_links = 1;
        break;
    }
    case 0x0006:
        o->has_links = 1;
        break;
    case 0x0001: {                                           /* dataspace */
        if (size < 4) return MBN_EFORMAT;
        int ver = d[0], rank = d[1], flags = d[2];
        size_t pos = ver == 1 ? 8 : 4;
        if (ver != 1 && ver != 2) return MBN_EUNSUPPORTED;
        if (rank > 8 || pos + (size_t)rank * h->sl > size) return MBN_EFORMAT;
        (void)flags;
        o->ndim = rank;
        for (int i = 0; i < rank; i++) o->shape[i] = (int64_t)rd_le(d + pos + (size_t)i * h->sl, h->sl);
        o->is_dataset = 1;
        break;
    }
    case 0x0003: {                                           /* datatype */
        if (size < 8) return MBN_EFORMAT;
        int cls = d[0] & 0x0f;
        uint32_t tsz = (uint32_t)rd_le(d + 4, 4);
        o->dtype_seen = 1;
        o->dtype_ok = (cls == 1 && tsz == 4 && (d[1] & 0x01) == 0);   /* float, 4 bytes, little-endian */
        break;
    }
    case 0x0008: {                                           /* data layout */
        if (size < 2) return MBN_EFORMAT;
        int ver = d[0];
        if (ver == 3 || ver == 4) {
            o->layout_class = d[1];
            if (d[1] == 1) {
                if (size < (size_t)(2 + h->so + h->sl)) return MBN_EFORMAT;
                o->data_addr = rd_off(h, d + 2);
                o->data_size = rd_le(d + 2 + h->so, h->sl);
            } else if (d[1] == 0) {
                if (size < 4) return MBN_EFORMAT;
                o->data_size = rd_le(d + 2, 2);
                if (4 + o->data_size > size) return MBN_EFORMAT;
                o->compact = d + 4;
            }
        } else if (ver == 1 || ver == 2) {
            if (size < 8) return MBN_EFORMAT;
            int rank = d[1];
            o->layout_class = d[2];
            if (d[2] == 1) {
                if (size < (size_t)(8 + h->so)) return MBN_EFORMAT;
                o->data_addr = rd_off(h, d + 8);
                o->data_size = 0;                            /* size comes from the dataspace */
            } else if (d[2] == 0) {
                size_t pos = 8 + (size_t)rank * 4;
                if (pos + 4 > size) return MBN_EFORMAT;
                o->data_size = rd_le(d + pos, 4);
                if (pos + 4 + o->data_size > size) return MBN_EFORMAT;
                o->compact = d + pos + 4;
            }
        } else return MBN_EUNSUPPORTED;
        break;
    }
    default:
        break;
    }
    return 0;
}

static int object_info(const mbn_h5 *h, uint64_t ohdr, obj_info *o)
{
    memset(o, 0, sizeof(*o));
    o->layout_class = -1;
    o->btree = o->heap = o->data_addr = H5_UNDEF;
    return iter_messages(h, ohdr, info_cb, o);
}

/* ------------------------------------------------------------------ child enumeration */
typedef int (*child_cb)(const mbn_h5 *h, const char *name, uint64_t ohdr, void *user);

static int walk_btree(const mbn_h5 *h, uint64_t node, const uint8_t *heap_data, uint64_t heap_size, child_cb cb,
                      void *user, int depth)
{
    if (depth > 32) return MBN_EFORMAT;
    const uint8_t *p = at(h, node, 8 + 2 * (uint64_t)h->so);
    if (!p) return MBN_EFORMAT;
    if (memcmp(p, "TREE", 4) == 0) {
        if (p[4] != 0) return MBN_EFORMAT;                   /* node type 0 = group */
        int level = p[5], used = (int)rd_le(p + 6, 2);
        uint64_t body = 8 + 2 * (uint64_t)h->so;
        uint64_t need = body + (uint64_t)used * (h->sl + h->so) + h->sl;
        p = at(h, node, need);
        if (!p) return MBN_EFORMAT;
        for (int i = 0; i < used; i++) {
            uint64_t child = rd_off(h, p + body + (uint64_t)i * (h->sl + h->so) + h->sl);
            int rc = walk_btree(h, child, heap_data, heap_size, cb, user, depth + 1);
            if (rc) return rc;
        }
        (void)level;
        return 0;
    }
    if (memcmp(p, "SNOD", 4) == 0) {
        int nsym = (int)rd_le(p + 6, 2);
        uint64_t esz = 2 * (uint64_t)h->so + 24;
        p = at(h, node, 8 + (uint64_t)nsym * esz);
        if (!p) return MBN_EFORMAT;
        for (int i = 0; i < nsym; i++) {
            const uint8_t *e = p + 8 + (uint64_t)i * esz;
            uint64_t noff = rd_le(e, h->so), ohdr = rd_off(h, e + h->so);
            if (noff >= heap_size) return MBN_EFORMAT;
            const char *name = (const char *)heap_data + noff;
            if (!memchr(name, 0, heap_size - noff)) return MBN_EFORMAT;
            int rc = cb(h, name, ohdr, user);
            if (rc) return rc;
        }
        return 0;
    }
    return MBN_EFORMAT;
}

typedef struct { child_cb cb; void *user; } link_ctx;

static int link_msg_cb(const mbn_h5 *h, int type, const uint8_t *d, size_t size, void *user)
{
    if (type != 0x0006) return 0;
    link_ctx *lc = (link_ctx *)user;
    if (size < 2 || d[0] != 1) return MBN_EFORMAT;
    int flags = d[1];
    size_t pos = 2;
    int ltype = 0;
    if (flags & 0x08) { if (pos >= size) return MBN_EFORMAT; ltype = d[pos++]; }
    if (flags & 0x04) pos += 8;
    if (flags & 0x10) pos += 1;
    int lsz = 1 << (flags & 3);
    if (pos + lsz > size) return MBN_EFORMAT;
    uint64_t nlen = rd_le(d + pos, lsz);
    pos += lsz;
    if (nlen >= H5_MAX_PATH || pos + nlen > size) return MBN_EFORMAT;
    char name[H5_MAX_PATH];
    memcpy(name, d + pos, nlen);
    name[nlen] = 0;
    pos += nlen;
    if (ltype != 0) return 0;                                /* soft/external links are skipped */
    if (pos + h->so > size) return MBN_EFORMAT;
    return lc->cb(h, name, rd_off(h, d + pos), lc->user);
}

static int for_each_child(const mbn_h5 *h, uint64_t ohdr, const obj_info *o, child_cb cb, void *user)
{
    if (o->has_stab) {
        const uint8_t *hp = at(h, o->heap, 8 + 2 * (uint64_t)h->sl + h->so);
        if (!hp || memcmp(hp, "HEAP", 4) != 0) return MBN_EFORMAT;
        uint64_t dsize = rd_le(hp + 8, h->sl);
        uint64_t daddr = rd_off(h, hp + 8 + 2 * h->sl);
        const uint8_t *hd = at(h, daddr, dsize);
        if (!hd) return MBN_EFORMAT;
        return walk_btree(h, o->btree, hd, dsize, cb, user, 0);
    }
    if (o->dense) return MBN_EUNSUPPORTED;
    if (o->has_links) {
        link_ctx lc = { cb, user };
        return iter_messages(h, ohdr, link_msg_cb, &lc);
    }
    return 0;
}

/* ------------------------------------------------------------------ open / close */
static int parse_superblock(mbn_h5 *h)
{
    static const uint8_t sig[8] = { 0x89, 'H', 'D', 'F', '\r', '\n', 0x1a, '\n' };
    for (uint64_t off = 0; off + 64 <= h->size; off = off ? off * 2 : 512) {
        const uint8_t *p = h->map + off;
        if (memcmp(p, sig, 8) != 0) continue;
        int ver = p[8];
        if (ver == 0 || ver == 1) {
            h->so = p[13];
            h->sl = p[14];
            if ((h->so != 4 && h->so != 8) || (h->sl != 4 && h->sl != 8)) return MBN_EFORMAT;
            uint64_t pos = 24 + (ver == 1 ? 4 : 0);
            if (off + pos + 4 * (uint64_t)h->so + 2 * h->so + 24 > h->size) return MBN_EFORMAT;
            h->base = 0;
            uint64_t base = rd_off(h, p + pos);
            pos += 4 * (uint64_t)h->so;                      /* base, free-space, EOF, driver info */
            h->root_ohdr = rd_off(h, p + pos + h->so);       /* root symbol table entry: name off, ohdr addr */
            h->base = base == H5_UNDEF ? 0 : base;
            return MBN_OK;
        }
        if (ver == 2 || ver == 3) {
            h->so = p[9];
            h->sl = p[10];
            if ((h->so != 4 && h->so != 8) || (h->sl != 4 && h->sl != 8)) return MBN_EFORMAT;
            if (off + 12 + 4 * (uint64_t)h->so + 4 > h->size) return MBN_EFORMAT;
            h->base = 0;
            uint64_t base = rd_off(h, p + 12);
            h->root_ohdr = rd_off(h, p + 12 + 3 * h->so);
            h->base = base == H5_UNDEF ? 0 : base;
            return MBN_OK;
        }
        return MBN_EUNSUPPORTED;
    }
    return MBN_EFORMAT;
}

int mbn_h5_open(const char *path, mbn_h5 **out)
{
    if (!path || !out) return MBN_EINVAL;
    *out = NULL;
    int fd = open(path, O_RDONLY);
    if (fd < 0) return MBN_EIO;
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size < 64) { close(fd); return st.st_size < 64 ? MBN_EFORMAT : MBN_EIO; }
    void *map = mmap(NULL, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (map == MAP_FAILED) { close(fd); return MBN_EIO; }
    mbn_h5 *h = (mbn_h5 *)calloc(1, sizeof(*h));
    if (!h) { munmap(map, (size_t)st.st_size); close(fd); return MBN_ENOMEM; }
    h->fd = fd;
    h->map = (const uint8_t *)map;
    h->size = (size_t)st.st_size;
    int rc = parse_superblock(h);
    if (rc == MBN_OK && !at(h, h->root_ohdr, 16)) rc = MBN_EFORMAT;
    if (rc != MBN_OK) { mbn_h5_close(h); return rc; }
    *out = h;
    return MBN_OK;
}

int mbn_h5_close(mbn_h5 *h)
{
    if (!h) return MBN_OK;
    if (h->map) munmap((void *)h->map, h->size);
    if (h->fd >= 0) close(h->fd);
    for (int i = 0; i < h->ncopies; i++) free(h->copies[i]);
    free(h->copies);
    free(h);
    return MBN_OK;
}

/* ------------------------------------------------------------------ lookup */
typedef struct { const char *want; size_t wlen; uint64_t found; } find_ctx;

static int find_cb(const mbn_h5 *h, const char *name, uint64_t ohdr, void *user)
{
    (void)h;
    find_ctx *f = (find_ctx *)user;
    if (strlen(name) == f->wlen && memcmp(name, f->want, f->wlen) == 0) {
        f->found = ohdr;
        return 1;
    }
    return 0;
}

static int resolve_path(const mbn_h5 *h, const char *path, uint64_t *ohdr_out)
{
    uint64_t cur = h->root_ohdr;
    const char *p = path;
    int depth = 0;
    while (*p) {
        while (*p == '/') p++;
        if (!*p) break;
        const char *e = strchr(p, '/');
        size_t len = e ? (size_t)(e - p) : strlen(p);
        if (++depth > H5_MAX_DEPTH) return MBN_EFORMAT;
        obj_info o;
        int rc = object_info(h, cur, &o);
        if (rc < 0) return rc;
        find_ctx f = { p, len, H5_UNDEF };
        rc = for_each_child(h, cur, &o, find_cb, &f);
        if (rc < 0) return rc;
        if (f.found == H5_UNDEF) return MBN_ENOTFOUND;
        cur = f.found;
        p += len;
    }
    *ohdr_out = cur;
    return MBN_OK;
}

static int dataset_view(const mbn_h5 *h, uint64_t ohdr, int *ndim, int64_t shape[8], const float **data)
{
    obj_info o;
    int rc = object_info(h, ohdr, &o);
    if (rc < 0) return rc;
    if (!o.is_dataset || o.layout_class < 0) return MBN_ENOTFOUND;      /* a group, not a dataset */
    if (!o.dtype_seen || !o.dtype_ok) return MBN_EUNSUPPORTED;
    uint64_t count = 1;
    for (int i = 0; i < o.ndim; i++) {
        if (o.shape[i] < 0 || (o.shape[i] && count > (UINT64_MAX / 8) / (uint64_t)o.shape[i])) return MBN_EFORMAT;
        count *= (uint64_t)o.shape[i];
    }
    const uint8_t *p = NULL;
    if (o.layout_class == 1) {
        if (o.data_addr == H5_UNDEF) {                                   /* never written: legal only for an empty dataset */
            if (count != 0) return MBN_EFORMAT;
            if (ndim) *ndim = o.ndim;
            if (shape) for (int i = 0; i < o.ndim; i++) shape[i] = o.shape[i];
            if (data) *data = NULL;
            return MBN_OK;
        }
        p = at(h, o.data_addr, count * 4);
        if (!p) return MBN_EFORMAT;
    } else if (o.layout_class == 0) {
        if (o.data_size < count * 4) return MBN_EFORMAT;
        p = o.compact;
    } else return MBN_EUNSUPPORTED;                                      /* chunked */
    if (((uintptr_t)p & 3) != 0) {                                      /* cannot hand out an unaligned float*: copy */
        mbn_h5 *hm = (mbn_h5 *)h;
        void **nc = (void **)realloc(hm->copies, sizeof(void *) * (size_t)(hm->ncopies + 1));
        if (!nc) return MBN_ENOMEM;
        hm->copies = nc;
        void *cp = malloc(count ? count * 4 : 4);
        if (!cp) return MBN_ENOMEM;
        memcpy(cp, p, count * 4);
        hm->copies[hm->ncopies++] = cp;
        p = (const uint8_t *)cp;
    }
    if (ndim) *ndim = o.ndim;
    if (shape) for (int i = 0; i < o.ndim; i++) shape[i] = o.shape[i];
    if (data) *data = (const float *)p;
    return MBN_OK;
}

int mbn_h5_get(mbn_h5 *h, const char *name, int *ndim, int64_t shape[8], const float **data)
{
    if (!h || !name) return MBN_EINVAL;
    uint64_t ohdr;
    int rc = resolve_path(h, name, &ohdr);
    if (rc == MBN_ENOTFOUND) {                              /* full-model files keep weights under /model_weights */
        char alt[H5_MAX_PATH];
        if (snprintf(alt, sizeof(alt), "/model_weights/%s", name[0] == '/' ? name + 1 : name) < (int)sizeof(alt))
            rc = resolve_path(h, alt, &ohdr);
    }
    if (rc != MBN_OK) return rc;
    return dataset_view(h, ohdr, ndim, shape, data);
}

/* ------------------------------------------------------------------ visit */
typedef struct {
    int (*cb)(const char *, int, const int64_t *, void *);
    void *user;
    char path[H5_MAX_PATH];
    int depth;
    int rc;
} visit_ctx;

static int visit_cb(const mbn_h5 *h, const char *name, uint64_t ohdr, void *user)
{
    visit_ctx *v = (visit_ctx *)user;
    size_t plen = strlen(v->path), nlen = strlen(name);
    if (plen + 1 + nlen + 1 > sizeof(v->path) || v->depth >= H5_MAX_DEPTH) return MBN_EFORMAT;
    v->path[plen] = '/';
    memcpy(v->path + plen + 1, name, nlen + 1);
    obj_info o;
    int rc = object_info(h, ohdr, &o);
    if (rc < 0) return rc;
    if (o.is_dataset) {
        rc = v->cb(v->path, o.ndim, o.shape, v->user);
    } else {
        v->depth++;
        rc = for_each_child(h, ohdr, &o, visit_cb, v);
        v->depth--;
    }
    v->path[plen] = 0;
    return rc;
}

int mbn_h5_visit(mbn_h5 *h, int (*cb)(const char *path, int ndim, const int64_t *shape, void *user), void *user)
{
    if (!h || !cb) return MBN_EINVAL;
    visit_ctx v;
    memset(&v, 0, sizeof(v));
    v.cb = cb;
    v.user = user;
    obj_info o;
    int rc = object_info(h, h->root_ohdr, &o);
    if (rc < 0) return rc;
    rc = for_each_child(h, h->root_ohdr, &o, visit_cb, &v);
    return rc < 0 ? rc : MBN_OK;
}

/* ================================================================== writer */
typedef struct wnode {
    char *name;
    struct wnode **kids;
    int nkids, cap;
    int is_dataset, ndim;
    int64_t shape[8];
    float *data;
    uint64_t count;
    /* assigned addresses */
    uint64_t ohdr, btree, heap, heap_data, snod, raw;
    uint64_t heap_data_size;
    uint64_t *name_off;
} wnode;

struct mbn_h5_writer {
    char *path;
    wnode *root;
    int max_kids;
    int failed;
};

static wnode *wnode_new(const char *name, size_t len)
{
    wnode *n = (wnode *)calloc(1, sizeof(*n));
    if (!n) return NULL;
    n->name = (char *)malloc(len + 1);
    if (!n->name) { free(n); return NULL; }
    memcpy(n->name, name, len);
    n->name[len] = 0;
    return n;
}

static void wnode_free(wnode *n)
{
    if (!n) return;
    for (int i = 0; i < n->nkids; i++) wnode_free(n->kids[i]);
    free(n->kids);
    free(n->name);
    free(n->data);
    free(n->name_off);
    free(n);
}

static wnode *wnode_child(wnode *p, const char *name, size_t len, int create)
{
    for (int i = 0; i < p->nkids; i++)
        if (strlen(p->kids[i]->name) == len && memcmp(p->kids[i]->name, name, len) == 0) return p->kids[i];
    if (!create) return NULL;
    if (p->nkids == p->cap) {
        int nc = p->cap ? p->cap * 2 : 8;
        wnode **nk = (wnode **)realloc(p->kids, sizeof(wnode *) * nc);
        if (!nk) return NULL;
        p->kids = nk;
        p->cap = nc;
    }
    wnode *c = wnode_new(name, len);
    if (!c) return NULL;
    p->kids[p->nkids++] = c;
    return c;
}

int mbn_h5_create(const char *path, mbn_h5_writer **out)
{
    if (!path || !out) return MBN_EINVAL;
    mbn_h5_writer *w = (mbn_h5_writer *)calloc(1, sizeof(*w));
    if (!w) return MBN_ENOMEM;
    w->path = strdup(path);
    w->root = wnode_new("", 0);
    if (!w->path || !w->root) { free(w->path); wnode_free(w->root); free(w); return MBN_ENOMEM; }
    *out = w;
    return MBN_OK;
}

int mbn_h5_put(mbn_h5_writer *w, const char *name, int ndim, const int64_t *shape, const float *data)
{
    if (!w || !name || ndim < 0 || ndim > 8 || (ndim && !shape) || !data) return MBN_EINVAL;
    wnode *cur = w->root;
    const char *p = name;
    for (;;) {
        while (*p == '/') p++;
        if (!*p) return MBN_EINVAL;
        const char *e = strchr(p, '/');
        size_t len = e ? (size_t)(e - p) : strlen(p);
        int last = !e || !e[1];
        wnode *c = wnode_child(cur, p, len, 1);
        if (!c) return MBN_ENOMEM;
        if (last) {
            if (c->is_dataset || c->nkids) return MBN_EINVAL;           /* duplicate */
            uint64_t count = 1;
            for (int i = 0; i < ndim; i++) {
                if (shape[i] < 0) return MBN_EINVAL;
                count *= (uint64_t)shape[i];
            }
            c->data = (float *)malloc(count ? count * 4 : 4);
            if (!c->data) return MBN_ENOMEM;
            memcpy(c->data, data, count * 4);
            c->is_dataset = 1;
            c->ndim = ndim;
            c->count = count;
            for (int i = 0; i < ndim; i++) c->shape[i] = shape[i];
            return MBN_OK;
        }
        if (c->is_dataset) return MBN_EINVAL;
        cur = c;
        p += len;
    }
}

static int cmp_nodes(const void *a, const void *b)
{
    return strcmp((*(wnode *const *)a)->name, (*(wnode *const *)b)->name);
}

static uint64_t al8(uint64_t x) { return (x + 7) & ~(uint64_t)7; }

#define W_BTREE_K 16
#define W_BTREE_SIZE (24 + (2 * W_BTREE_K + 1) * 8 + 2 * W_BTREE_K * 8)

static void wr_le(uint8_t *p, uint64_t v, int n)
{
    for (int i = 0; i < n; i++) { p[i] = (uint8_t)(v & 0xff); v >>= 8; }
}

/* pass 1: sort children, assign addresses depth-first. leaf_k fixes the SNOD size for the whole file. */
static void w_layout(wnode *n, uint64_t *pos, int leaf_k)
{
    if (n->is_dataset) {
        n->ohdr = *pos;
        uint64_t msgs = (8 + 8 + 8 * (uint64_t)n->ndim) + (8 + 24) + (8 + 8) + (8 + 24);
        *pos = al8(*pos + 16 + msgs);
        *pos = (*pos + 63) & ~(uint64_t)63;                  /* raw data 64-byte aligned in the file */
        n->raw = *pos;
        *pos = al8(*pos + n->count * 4);
        return;
    }
    qsort(n->kids, (size_t)n->nkids, sizeof(wnode *), cmp_nodes);
    n->ohdr = *pos;  *pos += 16 + 8 + 16;
    n->btree = *pos; *pos += W_BTREE_SIZE;
    n->snod = *pos;  *pos += 8 + 2 * (uint64_t)leaf_k * 40;
    n->heap = *pos;  *pos += 32;
    n->heap_data = *pos;
    n->name_off = (uint64_t *)calloc((size_t)(n->nkids ? n->nkids : 1), sizeof(uint64_t));
    uint64_t hp = 8;                                         /* offset 0 holds the empty string */
    for (int i = 0; i < n->nkids; i++) {
        n->name_off[i] = hp;
        hp = al8(hp + strlen(n->kids[i]->name) + 1);
    }
    n->heap_data_size = hp + 16;                             /* + one free block */
    *pos = al8(*pos + n->heap_data_size);
    for (int i = 0; i < n->nkids; i++) w_layout(n->kids[i], pos, leaf_k);
}

static void w_emit(const wnode *n, uint8_t *f, int leaf_k)
{
    if (n->is_dataset) {
        uint8_t *p = f + n->ohdr;
        uint64_t sp_sz = 8 + 8 * (uint64_t)n->ndim;
        p[0] = 1; wr_le(p + 2, 4, 2); wr_le(p + 4, 1, 4);
        wr_le(p + 8, (8 + sp_sz) + (8 + 24) + (8 + 8) + (8 + 24), 4);
        p += 16;
        /* dataspace v1 */
        wr_le(p, 0x0001, 2); wr_le(p + 2, sp_sz, 2); p += 8;
        p[0] = 1; p[1] = (uint8_t)n->ndim;
        for (int i = 0; i < n->ndim; i++) wr_le(p + 8 + 8 * i, (uint64_t)n->shape[i], 8);
        p += sp_sz;
        /* datatype: IEEE float32 little-endian, flagged constant (bit 0 of message flags) like libhdf5 does */
        wr_le(p, 0x0003, 2); wr_le(p + 2, 24, 2); p[4] = 1; p += 8;
        p[0] = 0x11; p[1] = 0x20; p[2] = 0x1f; p[3] = 0x00; wr_le(p + 4, 4, 4);
        wr_le(p + 8, 0, 2); wr_le(p + 10, 32, 2); p[12] = 23; p[13] = 8; p[14] = 0; p[15] = 23; wr_le(p + 16, 127, 4);
        p += 24;
        /* fill value v2: allocate early, never write fill, undefined value */
        wr_le(p, 0x0005, 2); wr_le(p + 2, 8, 2); p += 8;
        p[0] = 2; p[1] = 1; p[2] = 0; p[3] = 0;
        p += 8;
        /* layout v3 contiguous */
        wr_le(p, 0x0008, 2); wr_le(p + 2, 24, 2); p += 8;
        p[0] = 3; p[1] = 1; wr_le(p + 2, n->count ? n->raw : H5_UNDEF, 8); wr_le(p + 10, n->count * 4, 8);
        memcpy(f + n->raw, n->data, n->count * 4);
        return;
    }
    uint8_t *p = f + n->ohdr;                                /* object header v1: one symbol-table message */
    p[0] = 1; wr_le(p + 2, 1, 2); wr_le(p + 4, 1, 4); wr_le(p + 8, 24, 4);
    wr_le(p + 16, 0x0011, 2); wr_le(p + 18, 16, 2);
    wr_le(p + 24, n->btree, 8); wr_le(p + 32, n->heap, 8);
    p = f + n->btree;                                        /* B-tree root: level 0, one child (the SNOD) */
    memcpy(p, "TREE", 4); p[4] = 0; p[5] = 0; wr_le(p + 6, n->nkids ? 1 : 0, 2);
    wr_le(p + 8, H5_UNDEF, 8); wr_le(p + 16, H5_UNDEF, 8);
    wr_le(p + 24, 0, 8);                                     /* key 0: the empty string */
    if (n->nkids) {
        wr_le(p + 32, n->snod, 8);
        wr_le(p + 40, n->name_off[n->nkids - 1], 8);         /* key 1: largest name in the child */
    }
    p = f + n->snod;
    memcpy(p, "SNOD", 4); p[4] = 1; wr_le(p + 6, (uint64_t)n->nkids, 2);
    for (int i = 0; i < n->nkids; i++) {
        uint8_t *e = p + 8 + 40 * (uint64_t)i;
        const wnode *c = n->kids[i];
        wr_le(e, n->name_off[i], 8); wr_le(e + 8, c->ohdr, 8);
        if (!c->is_dataset) {                                /* cache type 1: B-tree + heap addresses */
            wr_le(e + 16, 1, 4); wr_le(e + 24, c->btree, 8); wr_le(e + 32, c->heap, 8);
        }
    }
    (void)leaf_k;
    p = f + n->heap;
    memcpy(p, "HEAP", 4);
    wr_le(p + 8, n->heap_data_size, 8); wr_le(p + 16, n->heap_data_size - 16, 8); wr_le(p + 24, n->heap_data, 8);
    uint8_t *hd = f + n->heap_data;
    for (int i = 0; i < n->nkids; i++) strcpy((char *)hd + n->name_off[i], n->kids[i]->name);
    wr_le(hd + n->heap_data_size - 16, 1, 8);                /* free block: next = 1 (end of list), size 16 */
    wr_le(hd + n->heap_data_size - 8, 16, 8);
    for (int i = 0; i < n->nkids; i++) w_emit(n->kids[i], f, leaf_k);
}

static int max_fanout(const wnode *n)
{
    int m = n->nkids;
    for (int i = 0; i < n->nkids; i++) {
        int c = max_fanout(n->kids[i]);
        if (c > m) m = c;
    }
    return m;
}

int mbn_h5_finish(mbn_h5_writer *w)
{
    if (!w) return MBN_EINVAL;
    int rc = MBN_OK;
    int fan = max_fanout(w->root);
    int leaf_k = (fan + 1) / 2;                              /* one SNOD (2K entries) holds every child of a group */
    if (leaf_k < 4) leaf_k = 4;
    if (leaf_k > 16384) rc = MBN_EUNSUPPORTED;
    uint64_t pos = 96;
    uint8_t *f = NULL;
    if (rc == MBN_OK) {
        w_layout(w->root, &pos, leaf_k);
        f = (uint8_t *)calloc(1, pos);
        if (!f) rc = MBN_ENOMEM;
    }
    if (rc == MBN_OK) {
        static const uint8_t sig[8] = { 0x89, 'H', 'D', 'F', '\r', '\n', 0x1a, '\n' };
        memcpy(f, sig, 8);
        f[13] = 8; f[14] = 8;                                /* versions all 0; offsets/lengths 8 bytes */
        wr_le(f + 16, (uint64_t)leaf_k, 2); wr_le(f + 18, W_BTREE_K, 2);
        wr_le(f + 24, 0, 8); wr_le(f + 32, H5_UNDEF, 8); wr_le(f + 40, pos, 8); wr_le(f + 48, H5_UNDEF, 8);
        wr_le(f + 56, 0, 8); wr_le(f + 64, w->root->ohdr, 8); wr_le(f + 72, 1, 4);   /* root entry, cache type 1 */
        wr_le(f + 80, w->root->btree, 8); wr_le(f + 88, w->root->heap, 8);
        w_emit(w->root, f, leaf_k);
        FILE *fp = fopen(w->path, "wb");
        if (!fp) rc = MBN_EIO;
        else {
            if (fwrite(f, 1, pos, fp) != pos) rc = MBN_EIO;
            if (fclose(fp) != 0) rc = MBN_EIO;
        }
    }
    free(f);
    wnode_free(w->root);
    free(w->path);
    free(w);
    return rc;
}
