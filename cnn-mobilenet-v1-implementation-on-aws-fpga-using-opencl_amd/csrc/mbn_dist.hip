// mbn_dist.hip — multi-GPU plumbing of the C-ABI (include/mbn.h, "multi-GPU"): one mbn_context per GPU of the node inside
// ONE host process, RCCL for the single collective of the path. SURVEY.md §8e: inference over independent images shards
// by batch with no data-path exchange (the reference processes exactly one image on exactly one device,
// MobileNet.c:155,215); what crosses xGMI is one broadcast of the packed, BatchNorm-folded parameter blob (~17 MB fp32)
// from GPU 0 at start-up. The C host (host/mobilenet_main.c --gpus N) drives one thread per GPU over these calls;
// bench.py keeps the one-process-per-GPU torch.distributed form the driver launches.
//
// RCCL is bound at run time (dlopen "librccl.so.1", then "librccl.so"): libmbn.so itself has no load-time dependency on
// it, a single-GPU caller never touches it, and a process that already holds a copy (PyTorch ships one) reuses that copy.
#include "mbn_internal.h"

#include <cstdlib>
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <new>

struct mbn_dist {
    int n = 0;
    std::vector<mbn_context *> ctx;
    std::vector<ncclComm_t> comm;
    void *lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    char last_error[256] = {0};
};

namespace {

int dist_fail(mbn_dist *d, ncclResult_t r, const char *what)
{
    if (d) snprintf(d->last_error, sizeof(d->last_error), "%s: %s", what, d->GetErrorString ? d->GetErrorString(r) : "rccl error");
    return MBN_EDEVICE;
}

template <typename F>
bool bind(void *lib, const char *name, F &fn)
{
    fn = reinterpret_cast<F>(dlsym(lib, name));
    return fn != nullptr;
}

}   // namespace

extern "C" {

int mbn_dist_init(int n_gpus, const int *device_ordinals, mbn_dist **out)
{
    if (!out || n_gpus <= 0 || n_gpus > 64) return MBN_EINVAL;
    *out = nullptr;
    int have = 0;
    if (hipGetDeviceCount(&have) != hipSuccess || have <= 0) return MBN_ENODEVICE;
    std::vector<int> devs(n_gpus);
    for (int i = 0; i < n_gpus; i++) {
        devs[i] = device_ordinals ? device_ordinals[i] : i;
        if (devs[i] < 0 || devs[i] >= have) return MBN_ENODEVICE;
        for (int j = 0; j < i; j++)
            if (devs[j] == devs[i]) return MBN_EINVAL;                      // one rank per GPU
    }
    mbn_dist *d = new (std::nothrow) mbn_dist();
    if (!d) return MBN_ENOMEM;
    d->n = n_gpus;
    for (int i = 0; i < n_gpus; i++) {
        mbn_context *c = nullptr;
        const int rc = mbn_init(devs[i], &c);
        if (rc != MBN_OK) { mbn_dist_shutdown(d); return rc; }
        d->ctx.push_back(c);
    }
    // a single GPU needs no communicator — unless MBN_DIST_FORCE_RCCL=1 asks for one anyway (round 4 rehearsal on the one-GPU boxes: dlopen, symbol
    // binding, ncclCommInitAll, the grouped ncclBroadcast on the context's stream and the teardown all run for real, only the xGMI transfer does not)
    const char *force = getenv("MBN_DIST_FORCE_RCCL");
    if (n_gpus > 1 || (force && force[0] == '1')) {
        d->lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!d->lib) d->lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (!d->lib || !bind(d->lib, "ncclCommInitAll", d->CommInitAll) || !bind(d->lib, "ncclCommDestroy", d->CommDestroy) ||
            !bind(d->lib, "ncclGroupStart", d->GroupStart) || !bind(d->lib, "ncclGroupEnd", d->GroupEnd) ||
            !bind(d->lib, "ncclBroadcast", d->Broadcast) || !bind(d->lib, "ncclGetErrorString", d->GetErrorString)) {
            mbn_dist_shutdown(d);
            return MBN_EUNSUPPORTED;                                         // no RCCL on this box
        }
        d->comm.resize(n_gpus);
        const ncclResult_t r = d->CommInitAll(d->comm.data(), n_gpus, devs.data());
        if (r != ncclSuccess) {
            d->comm.clear();
            mbn_dist_shutdown(d);
            return MBN_EDEVICE;
        }
    }
    *out = d;
    return MBN_OK;
}

int mbn_dist_size(const mbn_dist *d, int *n_gpus)
{
    if (!d || !n_gpus) return MBN_EINVAL;
    *n_gpus = d->n;
    return MBN_OK;
}

int mbn_dist_context(mbn_dist *d, int rank, mbn_context **ctx)
{
    if (!d || !ctx || rank < 0 || rank >= d->n) return MBN_EINVAL;
    *ctx = d->ctx[rank];
    return MBN_OK;
}

const char *mbn_dist_last_error(const mbn_dist *d) { return d ? d->last_error : "no communicator"; }

// dev_ptrs[r] = rank r's device buffer of `bytes` bytes (on rank r's GPU); root's content overwrites the others'.
// One grouped ncclBroadcast over xGMI, queued on every rank's context stream, then all of them are drained.
int mbn_dist_broadcast(mbn_dist *d, void *const *dev_ptrs, size_t bytes, int root)
{
    if (!d || !dev_ptrs || root < 0 || root >= d->n) return MBN_EINVAL;
    for (int r = 0; r < d->n; r++)
        if (!dev_ptrs[r]) return MBN_EINVAL;
    if (bytes == 0 || d->comm.empty()) return MBN_OK;                         // no communicator: one GPU (root's buffer is the only one)
    ncclResult_t r = d->GroupStart();
    if (r != ncclSuccess) return dist_fail(d, r, "ncclGroupStart");
    for (int i = 0; i < d->n && r == ncclSuccess; i++) {
        (void)hipSetDevice(d->ctx[i]->device);
        r = d->Broadcast(dev_ptrs[i], dev_ptrs[i], bytes, ncclUint8, root, d->comm[i], d->ctx[i]->stream);
    }
    const ncclResult_t e = d->GroupEnd();
    if (r != ncclSuccess) return dist_fail(d, r, "ncclBroadcast");
    if (e != ncclSuccess) return dist_fail(d, e, "ncclGroupEnd");
    return mbn_dist_sync(d);
}

int mbn_dist_sync(mbn_dist *d)
{
    if (!d) return MBN_EINVAL;
    for (int i = 0; i < d->n; i++) {
        const int rc = mbn_sync(d->ctx[i]);
        if (rc != MBN_OK) return rc;
    }
    return MBN_OK;
}

int mbn_dist_shutdown(mbn_dist *d)
{
    if (!d) return MBN_OK;
    for (size_t i = 0; i < d->comm.size(); i++)
        if (d->comm[i] && d->CommDestroy) (void)d->CommDestroy(d->comm[i]);
    for (mbn_context *c : d->ctx) (void)mbn_shutdown(c);
    if (d->lib) dlclose(d->lib);
    delete d;
    return MBN_OK;
}

}   // extern "C"
