// mbn_abi.hip — the C-ABI of include/mbn.h: context lifecycle, buffers, and the four layer entry points
// that replace the reference's clSetKernelArg + clEnqueueNDRangeKernel protocol
// (MobileNet.c:272-292 convolute, :363-382 depthwise, :453-470 pointwise, :2640-2656 pool).
//
// No CPU fallback exists: without a HIP device mbn_init returns MBN_ENODEVICE and nothing computes.
#include "mbn_internal.h"

#include <new>

mbn_tunables g_mbn_tune;

namespace {

// Buffer-size guard. Every buffer handed out by mbn_alloc is recorded with its size; when a pointer given to a layer call
// lies inside one of them, the bytes the call is about to touch must fit in what is left of that allocation, otherwise
// the call is a caller error (MBN_EINVAL) instead of a GPU memory fault. Caller-owned memory (hipMalloc, a torch tensor)
// is not in the map and is not checked. Cause on record (round 1, gpurun_out/fault.log): the C host uploaded a uint8
// image (3*96*96*3 bytes) and issued the fp32 first-layer kernel on it, which read 4x the buffer.
struct Span { const void *p; double bytes; const char *what; };

int span_check(mbn_context *ctx, const Span *sp, int n)
{
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (ctx->allocs.empty()) return MBN_OK;
    for (int i = 0; i < n; i++) {
        if (!sp[i].p || sp[i].bytes <= 0) continue;
        const uintptr_t a = (uintptr_t)sp[i].p;
        auto it = ctx->allocs.upper_bound(a);
        if (it == ctx->allocs.begin()) continue;
        --it;
        if (a >= it->first + it->second) continue;                 // not inside any of our buffers
        const double room = (double)(it->first + it->second - a);
        if (sp[i].bytes > room) {
            snprintf(ctx->last_error, sizeof(ctx->last_error), "%s: call touches %.0f bytes, %.0f left in its %zu-byte mbn_alloc buffer",
                     sp[i].what, sp[i].bytes, room, it->second);
            return MBN_EINVAL;
        }
    }
    return MBN_OK;
}
#define MBN_SPANS(ctx, ...)                                                            \
    do {                                                                               \
        const Span _sp[] = { __VA_ARGS__ };                                            \
        const int _rc = span_check((ctx), _sp, (int)(sizeof(_sp) / sizeof(_sp[0])));   \
        if (_rc != MBN_OK) return _rc;                                                 \
    } while (0)

// element size of an activation tensor of the call: bf16 mode stores 2 bytes unless the io flag says fp32
inline double esz_in(const mbn_call &c) { return c.dtype == MBN_DT_BF16 && !(c.io_flags & MBN_IO_IN_F32) ? 2.0 : 4.0; }
inline double esz_out(const mbn_call &c) { return c.dtype == MBN_DT_BF16 && !(c.io_flags & MBN_IO_OUT_F32) ? 2.0 : 4.0; }

// Library-owned images derived from [lo, hi) (the pre-split filter images of pw_emul): taken out of the map under the lock,
// released by the caller once the device is idle (they may be in use on any of the net runner's sub-batch streams).
std::vector<void *> take_derived(mbn_context *ctx, uintptr_t lo, uintptr_t hi)
{
    std::vector<void *> gone;
    for (auto it = ctx->emul_ws.begin(); it != ctx->emul_ws.end();) {
        if (it->first.first < hi && it->first.first + it->second.src_bytes > lo) {
            gone.push_back(it->second.p);
            it = ctx->emul_ws.erase(it);
        } else ++it;
    }
    return gone;
}

}   // namespace

extern "C" {

static std::atomic<int> *tune_slot(const char *key, bool *lab_only)
{
    *lab_only = false;
    if (!key) return nullptr;
    if (!strcmp(key, "pw_tile")) return &g_mbn_tune.pw_tile;
    if (!strcmp(key, "net_stagger")) return &g_mbn_tune.net_stagger;
    if (!strcmp(key, "pw_splitk")) return &g_mbn_tune.pw_splitk;
    if (!strcmp(key, "pw_emul")) return &g_mbn_tune.pw_emul;
    if (!strcmp(key, "pw_emul_static")) return &g_mbn_tune.pw_emul_static;
    if (!strcmp(key, "lit_dot")) return &g_mbn_tune.lit_dot;
    if (!strcmp(key, "pw_clock")) return &g_mbn_tune.pw_clock;
    static const char *const lab_keys[] = { "dw_variant", "dw_nseg", "pw_stage", "conv_variant", "misc", "pw_ring", "pw_xn", "dwpw_variant",
                                            "exp0", "exp1", "exp2", "cu_mask" };
    for (const char *k : lab_keys)
        if (!strcmp(key, k)) *lab_only = true;
#ifdef MBN_LAB
    if (!strcmp(key, "dw_variant")) return &g_mbn_tune.dw_variant;
    if (!strcmp(key, "dw_nseg")) return &g_mbn_tune.dw_nseg;
    if (!strcmp(key, "pw_stage")) return &g_mbn_tune.pw_stage;
    if (!strcmp(key, "conv_variant")) return &g_mbn_tune.conv_variant;
    if (!strcmp(key, "misc")) return &g_mbn_tune.misc;
    if (!strcmp(key, "pw_ring")) return &g_mbn_tune.pw_ring;
    if (!strcmp(key, "pw_xn")) return &g_mbn_tune.pw_xn;
    if (!strcmp(key, "dwpw_variant")) return &g_mbn_tune.dwpw_variant;
    if (!strcmp(key, "exp0")) return &g_mbn_tune.exp0;
    if (!strcmp(key, "exp1")) return &g_mbn_tune.exp1;
    if (!strcmp(key, "exp2")) return &g_mbn_tune.exp2;
    if (!strcmp(key, "cu_mask")) return &g_mbn_tune.cu_mask;
#endif
    return nullptr;
}

int mbn_lab_build(void) { return MBN_LAB_BUILD; }

int mbn_tune_set(const char *key, int value)
{
    bool lab_only;
    std::atomic<int> *p = tune_slot(key, &lab_only);
    if (!p) return lab_only ? MBN_EUNSUPPORTED : MBN_ENOTFOUND;      // a lab knob in the shipped library
    p->store(value, std::memory_order_relaxed);
    return MBN_OK;
}

int mbn_tune_get(const char *key, int *value)
{
    bool lab_only;
    std::atomic<int> *p = tune_slot(key, &lab_only);
    if (!value) return MBN_EINVAL;
    if (!p) {
        if (!lab_only) return MBN_ENOTFOUND;
        *value = 0;                                                   // what the shipped library behaves like
        return MBN_OK;
    }
    *value = p->load(std::memory_order_relaxed);
    return MBN_OK;
}

int mbn_device_count(int *count)
{
    if (!count) return MBN_EINVAL;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) n = 0;
    *count = n;
    return MBN_OK;
}

int mbn_init(int device_ordinal, mbn_context **out)
{
    if (!out) return MBN_EINVAL;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return MBN_ENODEVICE;
    if (device_ordinal < 0 || device_ordinal >= n) return MBN_ENODEVICE;
    mbn_context *ctx = new (std::nothrow) mbn_context();
    if (!ctx) return MBN_ENOMEM;
    ctx->device = device_ordinal;
    hipError_t e = hipSetDevice(device_ordinal);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&ctx->ev_start);
    if (e == hipSuccess) e = hipEventCreate(&ctx->ev_stop);
    if (e != hipSuccess) {
        if (ctx->ev_start) (void)hipEventDestroy(ctx->ev_start);
        if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
        delete ctx;
        return MBN_EDEVICE;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_ordinal) == hipSuccess) {
        snprintf(ctx->name, sizeof(ctx->name), "%s (%s)", prop.name, prop.gcnArchName);
        ctx->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    *out = ctx;
    return MBN_OK;
}

int mbn_shutdown(mbn_context *ctx)
{
    if (!ctx) return MBN_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto &kv : ctx->allocs) (void)hipFree((void *)kv.first);
    if (ctx->lit_ws) (void)hipFree(ctx->lit_ws);
    for (auto &kv : ctx->emul_ws) (void)hipFree(kv.second.p);
    ctx->allocs.clear();
    for (hipEvent_t e : ctx->pool) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->marks) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->sync_events) (void)hipEventDestroy(e);
    (void)hipEventDestroy(ctx->ev_start);
    (void)hipEventDestroy(ctx->ev_stop);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return MBN_OK;
}

int mbn_device_name(mbn_context *ctx, char *buf, size_t buflen)
{
    if (!ctx || !buf || buflen == 0) return MBN_EINVAL;
    snprintf(buf, buflen, "%s", ctx->name);
    return MBN_OK;
}

int mbn_device_pci_bus_id(mbn_context *ctx, char *buf, size_t buflen)
{
    if (!ctx || !buf || buflen < 16) return MBN_EINVAL;
    buf[0] = 0;
    MBN_HIP_TRY(ctx, hipDeviceGetPCIBusId(buf, (int)buflen, ctx->device));
    return MBN_OK;
}

int mbn_device_cus(mbn_context *ctx, int *count)
{
    if (!ctx || !count) return MBN_EINVAL;
    *count = ctx->num_cus;
    return MBN_OK;
}

const char *mbn_last_device_error(mbn_context *ctx) { return ctx ? ctx->last_error : "no context"; }

int mbn_set_literal_quirks(mbn_context *ctx, uint32_t quirks)
{
    if (!ctx || (quirks & ~MBN_QUIRKS_KERNEL_CL)) return MBN_EINVAL;
    ctx->literal_quirks = quirks;
    return MBN_OK;
}

int mbn_get_stream(mbn_context *ctx, void **stream)
{
    if (!ctx || !stream) return MBN_EINVAL;
    *stream = (void *)ctx->stream;
    return MBN_OK;
}

int mbn_stream_create(mbn_context *ctx, void **stream)
{
    if (!ctx || !stream) return MBN_EINVAL;
    (void)hipSetDevice(ctx->device);
    hipStream_t s;
    const int cm = g_mbn_tune.cu_mask;
    if (cm == 1 || cm == 2) {
        // LAB experiment (round 4, profiles/r04/j_streams_cu_mask.txt: XCD halves equal to the shared chip within 0.2 %, halves of every XCD -2.3 %): sub-batch streams confined to disjoint halves of the chip, so that one stream's HBM-bound kernels run beside the
        // other's MFMA-bound ones instead of time-slicing the same CUs. CU-mask bit i is CU i / 8 of XCD i % 8 (the driver deals the bits round the XCDs)
        static std::atomic<int> seq{0};
        const int half = seq.fetch_add(1) & 1;
        uint32_t mask[8];
        for (int w = 0; w < 8; w++) {
            uint32_t m = 0;
            for (int b = 0; b < 32; b++) {
                const int i = 32 * w + b;
                const bool mine = cm == 1 ? ((i & 7) < 4) == (half == 0) : (i < 128) == (half == 0);
                if (mine) m |= 1u << b;
            }
            mask[w] = m;
        }
        MBN_HIP_TRY(ctx, hipExtStreamCreateWithCUMask(&s, 8, mask));
        *stream = (void *)s;
        return MBN_OK;
    }
    MBN_HIP_TRY(ctx, hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = (void *)s;
    return MBN_OK;
}

int mbn_stream_destroy(mbn_context *ctx, void *stream)
{
    if (!ctx || !stream) return MBN_EINVAL;
    (void)hipSetDevice(ctx->device);
    MBN_HIP_TRY(ctx, hipStreamSynchronize((hipStream_t)stream));
    MBN_HIP_TRY(ctx, hipStreamDestroy((hipStream_t)stream));
    return MBN_OK;
}

int mbn_stream_wait(mbn_context *ctx, void *waiter, void *signaler)
{
    if (!ctx) return MBN_EINVAL;
    hipStream_t w = waiter ? (hipStream_t)waiter : ctx->stream, sg = signaler ? (hipStream_t)signaler : ctx->stream;
    if (w == sg) return MBN_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->sync_events.size() < 64) {
        hipEvent_t e;
        MBN_HIP_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->sync_events.push_back(e);
        ctx->sync_next = ctx->sync_events.size() - 1;
    } else ctx->sync_next = (ctx->sync_next + 1) % ctx->sync_events.size();
    hipEvent_t e = ctx->sync_events[ctx->sync_next];      // re-recording an event is legal: earlier waits keep their capture
    MBN_HIP_TRY(ctx, hipEventRecord(e, sg));
    MBN_HIP_TRY(ctx, hipStreamWaitEvent(w, e, 0));
    return MBN_OK;
}

int mbn_graph_begin(mbn_context *ctx, void *stream)
{
    if (!ctx) return MBN_EINVAL;
    (void)hipSetDevice(ctx->device);
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    MBN_HIP_TRY(ctx, hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    return MBN_OK;
}

int mbn_graph_end(mbn_context *ctx, void *stream, void **graph_exec)
{
    if (!ctx || !graph_exec) return MBN_EINVAL;
    *graph_exec = nullptr;
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    hipGraph_t g = nullptr;
    MBN_HIP_TRY(ctx, hipStreamEndCapture(s, &g));
    hipGraphExec_t ge = nullptr;
    hipError_t e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess) return mbn_record_hip_error(ctx, e, "hipGraphInstantiate");
    *graph_exec = (void *)ge;
    return MBN_OK;
}

int mbn_graph_launch(mbn_context *ctx, void *graph_exec, void *stream)
{
    if (!ctx || !graph_exec) return MBN_EINVAL;
    (void)hipSetDevice(ctx->device);
    MBN_HIP_TRY(ctx, hipGraphLaunch((hipGraphExec_t)graph_exec, stream ? (hipStream_t)stream : ctx->stream));
    return MBN_OK;
}

int mbn_graph_destroy(mbn_context *ctx, void *graph_exec)
{
    if (!ctx) return MBN_EINVAL;
    if (!graph_exec) return MBN_OK;
    MBN_HIP_TRY(ctx, hipGraphExecDestroy((hipGraphExec_t)graph_exec));
    return MBN_OK;
}

int mbn_alloc(mbn_context *ctx, size_t bytes, void **dptr)
{
    if (!ctx || !dptr || bytes == 0) return MBN_EINVAL;
    *dptr = nullptr;
    (void)hipSetDevice(ctx->device);
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e == hipErrorOutOfMemory) return MBN_ENOMEM;
    if (e != hipSuccess) return mbn_record_hip_error(ctx, e, "hipMalloc");
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->allocs[(uintptr_t)p] = bytes;
    *dptr = p;
    return MBN_OK;
}

int mbn_free(mbn_context *ctx, void *dptr)
{
    if (!ctx) return MBN_EINVAL;
    if (!dptr) return MBN_OK;
    std::vector<void *> gone;
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        auto it = ctx->allocs.find((uintptr_t)dptr);
        if (it == ctx->allocs.end()) return MBN_EINVAL;   // not ours: caller-owned memory is never freed here
        const size_t freed = it->second;
        ctx->allocs.erase(it);
        gone = take_derived(ctx, (uintptr_t)dptr, (uintptr_t)dptr + freed);     // images of filters inside the freed buffer (they leaked)
    }
    (void)hipSetDevice(ctx->device);
    if (!gone.empty()) MBN_HIP_TRY(ctx, hipDeviceSynchronize());
    else MBN_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (void *p : gone) (void)hipFree(p);
    MBN_HIP_TRY(ctx, hipFree(dptr));
    return MBN_OK;
}

int mbn_forget(mbn_context *ctx, const void *dptr, size_t bytes)
{
    if (!ctx) return MBN_EINVAL;
    if (!dptr || bytes == 0) return MBN_OK;
    std::vector<void *> gone;
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        gone = take_derived(ctx, (uintptr_t)dptr, (uintptr_t)dptr + bytes);
    }
    if (gone.empty()) return MBN_OK;
    (void)hipSetDevice(ctx->device);
    MBN_HIP_TRY(ctx, hipDeviceSynchronize());
    for (void *p : gone) (void)hipFree(p);
    return MBN_OK;
}

int mbn_upload(mbn_context *ctx, void *dst, const void *src, size_t bytes)
{
    if (!ctx || !dst || !src) return MBN_EINVAL;
    if (bytes == 0) return MBN_OK;
    MBN_SPANS(ctx, { dst, (double)bytes, "upload destination" });
    mbn_pw_emul_invalidate(ctx, dst, bytes);
    (void)hipSetDevice(ctx->device);
    MBN_HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    MBN_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // blocking like CL_TRUE, MobileNet.c:350
    return MBN_OK;
}

int mbn_download(mbn_context *ctx, void *dst, const void *src, size_t bytes)
{
    if (!ctx || !dst || !src) return MBN_EINVAL;
    if (bytes == 0) return MBN_OK;
    MBN_SPANS(ctx, { src, (double)bytes, "download source" });
    (void)hipSetDevice(ctx->device);
    MBN_HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    MBN_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // blocking like CL_TRUE, MobileNet.c:395
    return MBN_OK;
}

int mbn_memset(mbn_context *ctx, void *dst, int byte, size_t bytes)
{
    if (!ctx || !dst) return MBN_EINVAL;
    if (bytes == 0) return MBN_OK;
    MBN_SPANS(ctx, { dst, (double)bytes, "memset destination" });
    mbn_pw_emul_invalidate(ctx, dst, bytes);
    (void)hipSetDevice(ctx->device);
    MBN_HIP_TRY(ctx, hipMemsetAsync(dst, byte, bytes, ctx->stream));
    return MBN_OK;
}

int mbn_sync(mbn_context *ctx)
{
    if (!ctx) return MBN_EINVAL;
    (void)hipSetDevice(ctx->device);
    MBN_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return MBN_OK;
}

int mbn_set_profiling(mbn_context *ctx, int enabled)
{
    if (!ctx) return MBN_EINVAL;
    ctx->profiling = enabled != 0;
    ctx->ev_valid = false;
    return MBN_OK;
}

int mbn_last_kernel_ms(mbn_context *ctx, float *ms)
{
    if (!ctx || !ms) return MBN_EINVAL;
    if (!ctx->ev_valid) return MBN_EINVAL;
    MBN_HIP_TRY(ctx, hipEventSynchronize(ctx->ev_stop));
    MBN_HIP_TRY(ctx, hipEventElapsedTime(ms, ctx->ev_start, ctx->ev_stop));
    return MBN_OK;
}

int mbn_profile_begin(mbn_context *ctx, int capacity)
{
    if (!ctx || capacity <= 0 || capacity > (1 << 20)) return MBN_EINVAL;
    (void)hipSetDevice(ctx->device);
    while ((int)ctx->pool.size() < 2 * capacity) {
        hipEvent_t e;
        MBN_HIP_TRY(ctx, hipEventCreate(&e));
        ctx->pool.push_back(e);
    }
    ctx->pool_cap = capacity;
    ctx->pool_used = 0;
    ctx->pool_on = true;
    return MBN_OK;
}

int mbn_profile_pause(mbn_context *ctx, int paused)
{
    if (!ctx) return MBN_EINVAL;
    ctx->pool_on = !paused && ctx->pool_cap > 0;
    return MBN_OK;
}

int mbn_profile_end(mbn_context *ctx, float *ms, int ms_capacity, int *count)
{
    if (!ctx || !ms || !count || ms_capacity < 0) return MBN_EINVAL;
    ctx->pool_on = false;
    (void)hipSetDevice(ctx->device);
    MBN_HIP_TRY(ctx, hipDeviceSynchronize());
    int n = ctx->pool_used < ms_capacity ? ctx->pool_used : ms_capacity;
    for (int i = 0; i < n; i++) MBN_HIP_TRY(ctx, hipEventElapsedTime(&ms[i], ctx->pool[2 * i], ctx->pool[2 * i + 1]));
    *count = n;
    return MBN_OK;
}

int mbn_mark(mbn_context *ctx, void *stream)
{
    if (!ctx) return MBN_EINVAL;
    (void)hipSetDevice(ctx->device);
    if (ctx->marks_used == ctx->marks.size()) {
        if (ctx->marks.size() >= (1u << 20)) return MBN_ENOMEM;
        hipEvent_t e;
        MBN_HIP_TRY(ctx, hipEventCreate(&e));
        ctx->marks.push_back(e);
    }
    MBN_HIP_TRY(ctx, hipEventRecord(ctx->marks[ctx->marks_used], stream ? (hipStream_t)stream : ctx->stream));
    ctx->marks_used++;
    return MBN_OK;
}

int mbn_marks_read(mbn_context *ctx, float *ms_between, int capacity, int *count)
{
    if (!ctx || !count || capacity < 0 || (capacity > 0 && !ms_between)) return MBN_EINVAL;
    (void)hipSetDevice(ctx->device);
    const size_t used = ctx->marks_used;
    ctx->marks_used = 0;
    *count = 0;
    if (used == 0) return MBN_OK;
    MBN_HIP_TRY(ctx, hipEventSynchronize(ctx->marks[used - 1]));
    int n = 0;
    for (size_t i = 1; i < used && n < capacity; i++, n++)
        MBN_HIP_TRY(ctx, hipEventElapsedTime(&ms_between[n], ctx->marks[i - 1], ctx->marks[i]));
    *count = n;
    return MBN_OK;
}

}   // extern "C"

// --------------------------------------------------------------------------- call resolution

namespace {


struct Scope {   // hipEvent pair around one layer call when profiling is on (MobileNet.c:301-305 analogue)
    mbn_context *ctx;
    hipStream_t s;
    int slot = -1;
    bool capturing = false;
    Scope(mbn_context *c, hipStream_t st) : ctx(c), s(st)
    {
        (void)hipSetDevice(ctx->device);
        if (ctx->profiling || ctx->pool_on) {
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            capturing = hipStreamIsCapturing(s, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
        }
        if (capturing) return;
        if (ctx->profiling) (void)hipEventRecord(ctx->ev_start, s);
        if (ctx->pool_on && ctx->pool_used < ctx->pool_cap) {
            slot = ctx->pool_used++;
            (void)hipEventRecord(ctx->pool[2 * slot], s);
        }
    }
    int finish(int rc)
    {
        if (slot >= 0) (void)hipEventRecord(ctx->pool[2 * slot + 1], s);
        if (ctx->profiling && !capturing) {
            (void)hipEventRecord(ctx->ev_stop, s);
            ctx->ev_valid = true;
        }
        if (rc != MBN_OK) return rc;
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return mbn_record_hip_error(ctx, e, "kernel launch");
        return MBN_OK;
    }
};

__global__ void mbn_null_kernel() {}

int resolve(mbn_context *ctx, const mbn_layer_ext *ext, mbn_call *c, int *dtype)
{
    if (!ctx) return MBN_EINVAL;
    memset(c, 0, sizeof(*c));
    c->ctx = ctx;
    c->stream = ctx->stream;
    c->batch = 1;
    c->act = MBN_ACT_RELU;
    c->pad_top = c->pad_left = -1;
    c->quirks = ctx->literal_quirks;
    *dtype = MBN_DT_U8;
    if (!ext) return MBN_OK;
    if (ext->struct_size != sizeof(mbn_layer_ext)) return MBN_EINVAL;
    *dtype = ext->dtype;
    if (ext->dtype != MBN_DT_U8 && ext->dtype != MBN_DT_F32 && ext->dtype != MBN_DT_BF16) return MBN_EINVAL;
    if (ext->dtype == MBN_DT_U8 && ext->layout != MBN_LAYOUT_NCHW_PLANAR) return MBN_EUNSUPPORTED;
    if (ext->dtype != MBN_DT_U8 && ext->layout != MBN_LAYOUT_NHWC) return MBN_EUNSUPPORTED;
    if (ext->io_flags & ~(MBN_IO_IN_F32 | MBN_IO_OUT_F32 | MBN_IO_IN_U8 | MBN_IO_FILT_PACKED)) return MBN_EINVAL;
    c->dtype = ext->dtype;
    // IN_F32 / OUT_F32 only mean something in bf16 mode; IN_U8 (raw image into convolute) applies to fp32 and bf16
    c->io_flags = ext->dtype == MBN_DT_BF16 ? ext->io_flags : (ext->io_flags & MBN_IO_IN_U8);
    if (ext->batch < 0) return MBN_EINVAL;
    c->batch = ext->batch > 0 ? ext->batch : 1;
    if (ext->act < MBN_ACT_NONE || ext->act > MBN_ACT_RELU6) return MBN_EINVAL;
    c->act = ext->act;
    c->pad_top = ext->pad_top;
    c->pad_left = ext->pad_left;
    c->in_rows = ext->in_rows;
    c->in_cols = ext->in_cols;
    c->cin = ext->cin;
    c->g0 = ext->gsize0;
    c->g1 = ext->gsize1;
    if (ext->quirks_valid) {
        if (ext->quirks & ~MBN_QUIRKS_KERNEL_CL) return MBN_EINVAL;
        c->quirks = ext->quirks;
    }
    c->scale = (const float *)ext->scale;
    c->shift = (const float *)ext->shift;
    if (ext->stream) c->stream = (hipStream_t)ext->stream;
    return MBN_OK;
}

}   // namespace

extern "C" {

int mbn_profile_null(mbn_context *ctx, int with_kernel, void *stream)
{
    if (!ctx) return MBN_EINVAL;
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    Scope sc(ctx, s);
    if (with_kernel) hipLaunchKernelGGL(mbn_null_kernel, dim3(1), dim3(64), 0, s);
    return sc.finish(MBN_OK);
}

int mbn_convolute(mbn_context *ctx, void *output, const void *inp_r, const void *inp_g, const void *inp_b,
                  const void *filter_k, int rows, int cols, int filtersize, int stride, int op_size,
                  const mbn_layer_ext *ext)
{
    mbn_call c;
    int dtype;
    int rc = resolve(ctx, ext, &c, &dtype);
    if (rc != MBN_OK) return rc;
    if (!output || !inp_r || !filter_k) return MBN_EINVAL;
    if (rows <= 0 || cols <= 0 || op_size <= 0 || stride <= 0 || filtersize <= 0 || !(filtersize & 1))
        return MBN_EINVAL;
    // every argument check comes before the profiling Scope: an MBN_EINVAL must not consume an event slot
    if (dtype == MBN_DT_U8) {
        if (!inp_g || !inp_b) return MBN_EINVAL;
        if (rows / stride <= 0 || cols / stride <= 0) return MBN_EINVAL;
        const double plane = (double)rows * cols, oplane = (c.quirks & MBN_Q_LITERAL_INDEX) ? (double)(rows / 2) * (cols / 2)
                                                                                               : (double)(rows / stride) * (cols / stride);
        MBN_SPANS(ctx, { inp_r, plane * c.batch, "convolute inp_image_r" }, { inp_g, plane * c.batch, "convolute inp_image_g" },
                  { inp_b, plane * c.batch, "convolute inp_image_b" }, { output, oplane * op_size * c.batch, "convolute output" },
                  { filter_k, 4.0 * op_size * 3 * filtersize * filtersize, "convolute filter" });
        Scope sc(ctx, c.stream);
        return sc.finish(mbn_launch_lit_convolute(c, (uint8_t *)output, (const uint8_t *)inp_r, (const uint8_t *)inp_g,
                                                  (const uint8_t *)inp_b, (const int32_t *)filter_k, rows, cols,
                                                  filtersize, stride, op_size));
    }
    if (c.cin <= 0) c.cin = 3;
    {
        const double orow = (rows + stride - 1) / stride, ocol = (cols + stride - 1) / stride;
        const double in_es = (c.io_flags & MBN_IO_IN_U8) ? 1.0 : esz_in(c);
        MBN_SPANS(ctx, { inp_r, in_es * c.batch * rows * cols * c.cin, "convolute image" },
                  { output, esz_out(c) * c.batch * orow * ocol * op_size, "convolute output" },
                  { filter_k, 4.0 * filtersize * filtersize * c.cin * op_size, "convolute filter" },
                  { c.scale, 4.0 * op_size, "convolute scale" }, { c.shift, 4.0 * op_size, "convolute shift" });
    }
    Scope sc(ctx, c.stream);
    return sc.finish(mbn_launch_f32_conv(c, output, inp_r, (const float *)filter_k, rows, cols, filtersize, stride, op_size));
}

int mbn_depthwise(mbn_context *ctx, void *output, const void *inp_image, const void *filter_k, int rows, int cols,
                  int filtersize, int stride, int op_size, const mbn_layer_ext *ext)
{
    mbn_call c;
    int dtype;
    int rc = resolve(ctx, ext, &c, &dtype);
    if (rc != MBN_OK) return rc;
    if (!output || !inp_image || !filter_k) return MBN_EINVAL;
    if (rows <= 0 || cols <= 0 || op_size <= 0 || stride <= 0 || filtersize <= 0 || !(filtersize & 1))
        return MBN_EINVAL;
    if (c.in_rows <= 0) c.in_rows = rows * stride;
    if (c.in_cols <= 0) c.in_cols = cols * stride;
    {
        const double es = dtype == MBN_DT_U8 ? 1.0 : esz_in(c), fes = 4.0 * filtersize * filtersize * op_size;
        MBN_SPANS(ctx, { inp_image, es * c.batch * c.in_rows * c.in_cols * op_size, "depthwise input" },
                  { output, (dtype == MBN_DT_U8 ? 1.0 : esz_out(c)) * c.batch * rows * cols * op_size, "depthwise output" },
                  { filter_k, fes, "depthwise filter" }, { c.scale, 4.0 * op_size, "depthwise scale" },
                  { c.shift, 4.0 * op_size, "depthwise shift" });
    }
    Scope sc(ctx, c.stream);
    if (dtype == MBN_DT_U8)
        return sc.finish(mbn_launch_lit_depthwise(c, (uint8_t *)output, (const uint8_t *)inp_image,
                                                  (const int32_t *)filter_k, rows, cols, filtersize, stride, op_size));
    return sc.finish(mbn_launch_f32_depthwise(c, output, inp_image, (const float *)filter_k, rows, cols, filtersize, stride,
                                              op_size));
}

int mbn_pointwise(mbn_context *ctx, void *output, const void *inp_image, const void *filter_k, int rows, int cols,
                  int filtersize, int op_size, const mbn_layer_ext *ext)
{
    mbn_call c;
    int dtype;
    int rc = resolve(ctx, ext, &c, &dtype);
    if (rc != MBN_OK) return rc;
    if (!output || !inp_image || !filter_k) return MBN_EINVAL;
    if (rows <= 0 || cols <= 0 || op_size <= 0 || filtersize <= 0) return MBN_EINVAL;
    {
        const double px = (double)c.batch * rows * cols, lit = dtype == MBN_DT_U8;
        MBN_SPANS(ctx, { inp_image, (lit ? 1.0 : esz_in(c)) * px * filtersize, "pointwise input" },
                  { output, (lit ? 1.0 : esz_out(c)) * px * op_size, "pointwise output" },
                  { filter_k, (dtype == MBN_DT_BF16 ? 2.0 : 4.0) * op_size * filtersize +
                                  ((dtype == MBN_DT_BF16 && (c.io_flags & MBN_IO_FILT_PACKED)) ? (double)mbn_packed_filter_offset(op_size, filtersize) : 0.0),
                    "pointwise filter" },
                  { c.scale, 4.0 * op_size, "pointwise scale" }, { c.shift, 4.0 * op_size, "pointwise shift" });
    }
    Scope sc(ctx, c.stream);
    if (dtype == MBN_DT_U8)
        return sc.finish(mbn_launch_lit_pointwise(c, (uint8_t *)output, (const uint8_t *)inp_image,
                                                  (const int32_t *)filter_k, rows, cols, filtersize, op_size));
    long m = (long)c.batch * rows * cols;
    return sc.finish(mbn_launch_f32_pointwise(c, output, inp_image, filter_k, m, filtersize, op_size));
}

int mbn_pool(mbn_context *ctx, void *output, const void *inp_image, int rows, int cols, int filtersize, int op_size,
             const mbn_layer_ext *ext)
{
    mbn_call c;
    int dtype;
    int rc = resolve(ctx, ext, &c, &dtype);
    if (rc != MBN_OK) return rc;
    if (!output || !inp_image) return MBN_EINVAL;
    if (rows <= 0 || cols <= 0 || op_size <= 0 || filtersize <= 0) return MBN_EINVAL;
    if (dtype == MBN_DT_U8 && (long)filtersize * filtersize > (long)rows * cols) return MBN_EINVAL;   // kernel.cl:126 would run past the plane
    {
        const double lit = dtype == MBN_DT_U8;
        MBN_SPANS(ctx, { inp_image, (lit ? 1.0 : esz_in(c)) * c.batch * rows * cols * op_size, "pool input" },
                  { output, (lit ? 1.0 : esz_out(c)) * c.batch * op_size, "pool output" });
    }
    Scope sc(ctx, c.stream);
    if (dtype == MBN_DT_U8) {
        return sc.finish(mbn_launch_lit_pool(c, (uint8_t *)output, (const uint8_t *)inp_image, rows, cols, filtersize,
                                             op_size));
    }
    return sc.finish(mbn_launch_f32_pool(c, output, inp_image, rows, cols, filtersize, op_size));
}

int mbn_softmax_f32(mbn_context *ctx, void *probs, void *argmax_i32, const void *logits, int batch, int classes,
                    void *stream)
{
    if (!ctx || !logits || batch <= 0 || classes <= 0 || (!probs && !argmax_i32)) return MBN_EINVAL;
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    MBN_SPANS(ctx, { logits, 4.0 * batch * classes, "softmax logits" }, { probs, 4.0 * batch * classes, "softmax probs" },
              { argmax_i32, 4.0 * batch, "softmax argmax" });
    Scope sc(ctx, s);
    return sc.finish(mbn_launch_f32_softmax(ctx, s, (float *)probs, (int32_t *)argmax_i32, (const float *)logits, batch,
                                            classes));
}

int mbn_softmax_topk_f32(mbn_context *ctx, void *probs, void *topk_idx_i32, void *topk_prob_f32, const void *logits,
                         int batch, int classes, int k, void *stream)
{
    if (!ctx || !logits || !topk_idx_i32 || !topk_prob_f32 || batch <= 0 || classes <= 0 || k < 1 || k > 8) return MBN_EINVAL;
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    MBN_SPANS(ctx, { logits, 4.0 * batch * classes, "softmax_topk logits" }, { probs, 4.0 * batch * classes, "softmax_topk probs" },
              { topk_idx_i32, 4.0 * batch * k, "softmax_topk idx" }, { topk_prob_f32, 4.0 * batch * k, "softmax_topk prob" });
    Scope sc(ctx, s);
    return sc.finish(mbn_launch_f32_softmax_topk(ctx, s, (float *)probs, (int32_t *)topk_idx_i32, (float *)topk_prob_f32,
                                                 (const float *)logits, batch, classes, k));
}

int mbn_classifier_tail(mbn_context *ctx, void *topk_idx_i32, void *topk_prob_f32, void *probs, void *logits_scratch,
                        void *pooled_scratch, const void *in, const void *fc_w, const void *fc_bias, int batch, int rows,
                        int cols, int channels, int classes, int k, void *stream)
{
    if (!ctx || !topk_idx_i32 || !topk_prob_f32 || !logits_scratch || !pooled_scratch || !in || !fc_w || batch <= 0 ||
        rows <= 0 || rows != cols || channels <= 0 || classes <= 0 || k < 1 || k > 8)
        return MBN_EINVAL;
    mbn_layer_ext e;
    memset(&e, 0, sizeof e);
    e.struct_size = sizeof e;
    e.batch = batch;
    e.dtype = MBN_DT_F32;
    e.layout = MBN_LAYOUT_NHWC;
    e.act = MBN_ACT_NONE;
    e.pad_top = e.pad_left = -1;
    e.stream = stream;
    int rc = mbn_pool(ctx, pooled_scratch, in, rows, cols, rows, channels, &e);           // MobileNet.c:2603-2656
    if (rc != MBN_OK) return rc;
    e.shift = fc_bias;                                                                     // bias, no ReLU (B15)
    rc = mbn_pointwise(ctx, logits_scratch, pooled_scratch, fc_w, 1, 1, channels, classes, &e);   // MobileNet.c:2682-2739
    if (rc != MBN_OK) return rc;
    return mbn_softmax_topk_f32(ctx, probs, topk_idx_i32, topk_prob_f32, logits_scratch, batch, classes, k, stream);
}

size_t mbn_pool_fc_workspace_bytes(int channels, int classes)
{
    if (channels < 64 || channels > 1024 || (channels % 64) != 0 || classes <= 0) return 0;
    return mbn_pool_fc_ws_bytes(channels, classes);
}

int mbn_pool_fc(mbn_context *ctx, void *logits, const void *in, const void *fc_w, const void *fc_bias, int batch, int rows, int cols,
                int channels, int classes, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!ctx || !logits || !in || !fc_w || !workspace || batch <= 0 || rows <= 0 || cols <= 0 || channels <= 0 || classes <= 0) return MBN_EINVAL;
    if (batch > 4 || channels < 64 || channels > 1024 || (channels % 64) != 0) return MBN_EUNSUPPORTED;
    if (workspace_bytes < mbn_pool_fc_ws_bytes(channels, classes)) return MBN_EINVAL;
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    MBN_SPANS(ctx, { in, 4.0 * batch * rows * cols * channels, "pool_fc input" }, { logits, 4.0 * batch * classes, "pool_fc logits" },
              { fc_w, 4.0 * classes * channels, "pool_fc filter" }, { workspace, (double)mbn_pool_fc_ws_bytes(channels, classes), "pool_fc workspace" });
    Scope sc(ctx, s);
    return sc.finish(mbn_launch_f32_pool_fc(ctx, s, (float *)logits, (const float *)in, (const float *)fc_w, (const float *)fc_bias, workspace,
                                            batch, rows * cols, channels, classes, 0, nullptr, nullptr, nullptr));
}

int mbn_classifier_tail_fused(mbn_context *ctx, void *topk_idx_i32, void *topk_prob_f32, void *probs, void *logits, const void *in, const void *fc_w,
                              const void *fc_bias, int batch, int rows, int cols, int channels, int classes, int k, void *workspace,
                              size_t workspace_bytes, void *stream)
{
    if (!ctx || !topk_idx_i32 || !topk_prob_f32 || !logits || !in || !fc_w || !workspace || batch <= 0 || rows <= 0 || cols <= 0 || channels <= 0 ||
        classes <= 0 || k < 1 || k > 8)
        return MBN_EINVAL;
    if (batch > 4 || channels < 64 || channels > 1024 || (channels % 64) != 0 || classes > 1024) return MBN_EUNSUPPORTED;
    if (workspace_bytes < mbn_pool_fc_ws_bytes(channels, classes)) return MBN_EINVAL;
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    MBN_SPANS(ctx, { in, 4.0 * batch * rows * cols * channels, "classifier_tail_fused input" }, { logits, 4.0 * batch * classes, "classifier_tail_fused logits" },
              { fc_w, 4.0 * classes * channels, "classifier_tail_fused filter" }, { workspace, (double)mbn_pool_fc_ws_bytes(channels, classes), "classifier_tail_fused workspace" },
              { probs, 4.0 * batch * classes, "classifier_tail_fused probs" }, { topk_idx_i32, 4.0 * batch * k, "classifier_tail_fused idx" },
              { topk_prob_f32, 4.0 * batch * k, "classifier_tail_fused prob" });
    Scope sc(ctx, s);
    return sc.finish(mbn_launch_f32_pool_fc(ctx, s, (float *)logits, (const float *)in, (const float *)fc_w, (const float *)fc_bias, workspace,
                                            batch, rows * cols, channels, classes, k, (float *)probs, (int32_t *)topk_idx_i32, (float *)topk_prob_f32));
}

static int stem_fused_impl(mbn_context *ctx, void *out, const void *image, const void *w1, const void *s1, const void *b1,
                           const void *wd, const void *s2, const void *b2, const void *wp, const void *s3, const void *b3,
                           int batch, int res, int c1, int c3, void *stream, int in_u8, int bf16)
{
    if (!ctx || !out || !image) return MBN_EINVAL;
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    // decide before opening the profiling scope so an unsupported shape does not consume an event slot
    if (!((c1 == 32 && c3 == 64) || (c1 == 16 && c3 == 32)) || res < 32 || (res % 32) != 0 || batch <= 0) return MBN_EUNSUPPORTED;
    MBN_SPANS(ctx, { image, (in_u8 ? 1.0 : 4.0) * batch * res * res * 3, "stem image" },
              { out, (bf16 ? 2.0 : 4.0) * batch * (res / 2) * (res / 2) * c3, "stem output" });
    Scope sc(ctx, s);
    return sc.finish(mbn_launch_f32_stem(ctx, s, (float *)out, (const float *)image, (const float *)w1, (const float *)s1,
                                         (const float *)b1, (const float *)wd, (const float *)s2, (const float *)b2,
                                         (const float *)wp, (const float *)s3, (const float *)b3, batch, res, c1, c3, in_u8, bf16));
}

int mbn_stem_fused(mbn_context *ctx, void *out, const void *image, const void *w1, const void *s1, const void *b1,
                   const void *wd, const void *s2, const void *b2, const void *wp, const void *s3, const void *b3,
                   int batch, int res, int c1, int c3, void *stream)
{
    return stem_fused_impl(ctx, out, image, w1, s1, b1, wd, s2, b2, wp, s3, b3, batch, res, c1, c3, stream, 0, 0);
}

int mbn_stem_fused_u8(mbn_context *ctx, void *out, const void *image_u8, const void *w1, const void *s1, const void *b1,
                      const void *wd, const void *s2, const void *b2, const void *wp, const void *s3, const void *b3,
                      int batch, int res, int c1, int c3, void *stream)
{
    return stem_fused_impl(ctx, out, image_u8, w1, s1, b1, wd, s2, b2, wp, s3, b3, batch, res, c1, c3, stream, 1, 0);
}

int mbn_stem_fused_ex(mbn_context *ctx, void *out, const void *image, const void *w1, const void *s1, const void *b1,
                      const void *wd, const void *s2, const void *b2, const void *wp, const void *s3, const void *b3,
                      int batch, int res, int c1, int c3, int flags, void *stream)
{
    if (flags & ~(MBN_STEM_IN_U8 | MBN_STEM_BF16)) return MBN_EINVAL;
    return stem_fused_impl(ctx, out, image, w1, s1, b1, wd, s2, b2, wp, s3, b3, batch, res, c1, c3, stream,
                           (flags & MBN_STEM_IN_U8) != 0, (flags & MBN_STEM_BF16) != 0);
}

int mbn_dwpw_fused(mbn_context *ctx, void *out, const void *in, const void *wd, const void *s2, const void *b2,
                   const void *wp, const void *s3, const void *b3, int batch, int in_rows, int in_cols, int out_rows,
                   int out_cols, int cin, int cout, int stride, int pad_top, int pad_left, void *stream)
{
    if (!ctx) return MBN_EINVAL;
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    // decide before opening the profiling scope so an unsupported shape does not consume an event slot
    const int rc = mbn_f32_dwpw_check((const float *)out, (const float *)in, (const float *)wd, (const float *)s2,
                                      (const float *)b2, (const float *)wp, (const float *)s3, (const float *)b3, batch,
                                      in_rows, in_cols, out_rows, out_cols, cin, cout, stride, pad_top, pad_left);
    if (rc != MBN_OK) return rc;
    MBN_SPANS(ctx, { in, 4.0 * batch * in_rows * in_cols * cin, "dwpw input" },
              { out, 4.0 * batch * out_rows * out_cols * cout, "dwpw output" }, { wd, 36.0 * cin, "dwpw depthwise filter" },
              { wp, 4.0 * cin * cout, "dwpw pointwise filter" });
    Scope sc(ctx, s);
    // unified-wave kernel (mbn_f32_dwpw2.hip) for the 128-column tile, where it measures 9-12 % faster than the round-1
    // producer/consumer kernel; the 256-column tile stays on the round-1 kernel (within +-3 % of each other there).
    // dwpw_variant: 1 = always round-1, 2 = always unified (A/B hooks, tools/block_bench.py)
    // small problems (few images): a 256-column tile leaves most CUs idle and every workgroup walks the whole K alone —
    // batch 1, block 10-11: 40 us fused against 24 us as two launches — so below one 256-wide tile per CU the block runs
    // on 128-column tiles (twice the workgroups, half the MFMA work per step), i.e. on the unified kernel
    // opt-in: pointwise products on the bf16 matrix cores from exact operand splits (mbn_f32_dwpw2_x6.hip; pw_emul = 6 | 9)
    if (g_mbn_tune.pw_emul != 0 && g_mbn_tune.dwpw_variant == 0 &&
        mbn_launch_f32_dwpw2_x6(ctx, s, (float *)out, (const float *)in, (const float *)wd, (const float *)s2, (const float *)b2,
                                (const float *)wp, (const float *)s3, (const float *)b3, batch, in_rows, in_cols, out_rows, out_cols,
                                cin, cout, stride, pad_top, pad_left) == MBN_OK)
        return sc.finish(MBN_OK);
    const int dv = g_mbn_tune.dwpw_variant;
    // round 6: the wave-private form (mbn_f32_dwpw3.hip: no barrier in the loop, filter slice resident in LDS) where it applies
    // (Cin 64 / 128 / 256). Lab A/B: dwpw_variant 11 = always where eligible, 12 = never, 300 + bits = its ablation build.
    if ((dv == 11 || dv >= 300 || (dv == 0 && MBN_DWPW3_DEFAULT(stride, cin))) &&
        mbn_f32_dwpw3_eligible(ctx, batch, in_rows, in_cols, out_rows, out_cols, cin, cout, stride, pad_top, pad_left))
        return sc.finish(mbn_launch_f32_dwpw3(ctx, s, (float *)out, (const float *)in, (const float *)wd, (const float *)s2,
                                              (const float *)b2, (const float *)wp, (const float *)s3, (const float *)b3, batch,
                                              in_rows, in_cols, out_rows, out_cols, cin, cout, stride, pad_top, pad_left));
    const long tiles256 = (((long)batch * out_rows * out_cols + 127) / 128) * (cout / 256);
    const bool small = tiles256 < ctx->num_cus;
    // stride-2 blocks (15 x-window loads per lane and step): with the loads spread under the MFMA groups the unified kernel also wins at
    // 256 columns (block 8-9: 0.1967 / 0.1964 -> 0.1931 / 0.1893 ms, profiles/r02/b_block_kernel_variants.txt); stride 1 at 256 columns
    // stays on the round-1 kernel (block 10-11: 0.291 vs 0.308 ms)
    if (dv != 1 && (dv >= 2 || (cout % 256) != 0 || small || stride == 2))
        return sc.finish(mbn_launch_f32_dwpw2(ctx, s, (float *)out, (const float *)in, (const float *)wd, (const float *)s2,
                                              (const float *)b2, (const float *)wp, (const float *)s3, (const float *)b3, batch,
                                              in_rows, in_cols, out_rows, out_cols, cin, cout, stride, pad_top, pad_left));
    return sc.finish(mbn_launch_f32_dwpw(ctx, s, (float *)out, (const float *)in, (const float *)wd, (const float *)s2,
                                         (const float *)b2, (const float *)wp, (const float *)s3, (const float *)b3, batch,
                                         in_rows, in_cols, out_rows, out_cols, cin, cout, stride, pad_top, pad_left));
}

int mbn_blocks_resident_bf16(mbn_context *ctx, void *out, const void *in, const mbn_block_params *blocks, int nblocks, int batch,
                             int rows, int cols, int channels, void *stream)
{
    if (!ctx || !out || !in || !blocks || batch <= 0) return MBN_EINVAL;
    if (!mbn_bf16_res_eligible(rows, cols, channels, nblocks)) return MBN_EUNSUPPORTED;
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    const double act = 2.0 * batch * rows * cols * channels;
    MBN_SPANS(ctx, { in, act, "resident blocks input" }, { out, act, "resident blocks output" });
    for (int i = 0; i < nblocks; i++) {
        if (!blocks[i].wd || !blocks[i].s2 || !blocks[i].b2 || !blocks[i].wp_bf16 || !blocks[i].s3 || !blocks[i].b3) return MBN_EINVAL;
        MBN_SPANS(ctx, { blocks[i].wd, 36.0 * channels, "resident blocks depthwise filter" }, { blocks[i].wp_bf16, 2.0 * channels * channels, "resident blocks pointwise filter" },
                  { blocks[i].s2, 4.0 * channels, "resident blocks scale" }, { blocks[i].b3, 4.0 * channels, "resident blocks shift" });
    }
    Scope sc(ctx, s);
    return sc.finish(mbn_launch_bf16_res_blocks(ctx, s, out, in, blocks, nblocks, batch, rows, cols, channels));
}

int mbn_tail_resident_bf16(mbn_context *ctx, void *out, const void *in, const mbn_block_params *blocks, int batch, int rows, int cols, int c0, int c1,
                           void *stream)
{
    if (!ctx || !out || !in || !blocks || batch <= 0) return MBN_EINVAL;
    if (!mbn_bf16_tail_eligible(rows, cols, c0, c1)) return MBN_EUNSUPPORTED;
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    MBN_SPANS(ctx, { in, 2.0 * batch * rows * cols * c0, "resident tail input" }, { out, 2.0 * batch * c1, "resident tail output" });
    for (int i = 0; i < 2; i++) {
        if (!blocks[i].wd || !blocks[i].s2 || !blocks[i].b2 || !blocks[i].wp_bf16 || !blocks[i].s3 || !blocks[i].b3) return MBN_EINVAL;
        const int ci = i ? c1 : c0;
        MBN_SPANS(ctx, { blocks[i].wd, 36.0 * ci, "resident tail depthwise filter" }, { blocks[i].wp_bf16, 2.0 * ci * c1, "resident tail pointwise filter" },
                  { blocks[i].s2, 4.0 * ci, "resident tail scale" }, { blocks[i].b2, 4.0 * ci, "resident tail shift" },
                  { blocks[i].s3, 4.0 * c1, "resident tail scale" }, { blocks[i].b3, 4.0 * c1, "resident tail shift" });
    }
    Scope sc(ctx, s);
    return sc.finish(mbn_launch_bf16_tail(ctx, s, out, in, blocks, batch, rows, cols, c0, c1));
}

int mbn_dwpw_fused_bf16(mbn_context *ctx, void *out, const void *in, const void *wd, const void *s2, const void *b2,
                        const void *wp_bf16, const void *s3, const void *b3, int batch, int in_rows, int in_cols, int out_rows,
                        int out_cols, int cin, int cout, int stride, int pad_top, int pad_left, void *stream)
{
    if (!ctx) return MBN_EINVAL;
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    const int rc = mbn_bf16_dwpw_check(out, in, (const float *)wd, (const float *)s2, (const float *)b2, wp_bf16,
                                       (const float *)s3, (const float *)b3, batch, in_rows, in_cols, out_rows, out_cols, cin,
                                       cout, stride, pad_top, pad_left);
    if (rc != MBN_OK) return rc;
    MBN_SPANS(ctx, { in, 2.0 * batch * in_rows * in_cols * cin, "dwpw input" },
              { out, 2.0 * batch * out_rows * out_cols * cout, "dwpw output" }, { wd, 36.0 * cin, "dwpw depthwise filter" },
              { wp_bf16, 2.0 * cin * cout, "dwpw pointwise filter" });
    Scope sc(ctx, s);
#ifdef MBN_LAB
    if (g_mbn_tune.dwpw_variant == 1 && (cout % 128) == 0 && (cin % 64) == 0)      // the round-1 producer/consumer kernel (A/B hook; mbn_bf16_dwpw.hip's kernels are lab-only; whole 128-column tiles only)
        return sc.finish(mbn_launch_bf16_dwpw(ctx, s, out, in, (const float *)wd, (const float *)s2, (const float *)b2, wp_bf16,
                                              (const float *)s3, (const float *)b3, batch, in_rows, in_cols, out_rows, out_cols,
                                              cin, cout, stride, pad_top, pad_left));
#endif
    // unified-wave kernel (mbn_bf16_dwpw2.hip)
    return sc.finish(mbn_launch_bf16_dwpw2(ctx, s, out, in, (const float *)wd, (const float *)s2, (const float *)b2, wp_bf16,
                                           (const float *)s3, (const float *)b3, batch, in_rows, in_cols, out_rows, out_cols,
                                           cin, cout, stride, pad_top, pad_left));
}

int mbn_convert_f32_to_bf16(mbn_context *ctx, void *dst_bf16, const void *src_f32, size_t count, void *stream)
{
    if (!ctx || !dst_bf16 || !src_f32) return MBN_EINVAL;
    if (count == 0) return MBN_OK;
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    MBN_SPANS(ctx, { src_f32, 4.0 * count, "convert source" }, { dst_bf16, 2.0 * count, "convert destination" });
    Scope sc(ctx, s);
    return sc.finish(mbn_launch_convert(ctx, s, dst_bf16, src_f32, count, 1));
}

size_t mbn_packed_filter_offset(int cout, int cin)
{
    if (cout <= 0 || cin <= 0 || (cout % 256) != 0 || (cin % 64) != 0) return 0;
    return ((size_t)cout * cin * 2 + 255) & ~(size_t)255;
}

int mbn_pack_filter_bf16(mbn_context *ctx, void *filter_buf, int cout, int cin, void *stream)
{
    if (!ctx || !filter_buf) return MBN_EINVAL;
    const size_t off = mbn_packed_filter_offset(cout, cin);
    if (off == 0) return MBN_EUNSUPPORTED;
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    MBN_SPANS(ctx, { filter_buf, (double)off + 2.0 * cout * cin, "packed filter buffer" });
#ifdef MBN_LAB
    Scope sc(ctx, s);
    return sc.finish(mbn_launch_pack_filter_bf16(ctx, s, (char *)filter_buf + off, filter_buf, cout, cin));
#else
    (void)s;
    return MBN_EUNSUPPORTED;           // the kernel that reads packed images is not part of the shipped library (see mbn.h)
#endif
}

int mbn_convert_bf16_to_f32(mbn_context *ctx, void *dst_f32, const void *src_bf16, size_t count, void *stream)
{
    if (!ctx || !dst_f32 || !src_bf16) return MBN_EINVAL;
    if (count == 0) return MBN_OK;
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    MBN_SPANS(ctx, { src_bf16, 2.0 * count, "convert source" }, { dst_f32, 4.0 * count, "convert destination" });
    Scope sc(ctx, s);
    return sc.finish(mbn_launch_convert(ctx, s, dst_f32, src_bf16, count, 0));
}

int mbn_normalize_u8_to_f32(mbn_context *ctx, void *out_f32, const void *in_u8, size_t count, float scale, float bias,
                            void *stream)
{
    if (!ctx || !out_f32 || !in_u8) return MBN_EINVAL;
    if (count == 0) return MBN_OK;
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    MBN_SPANS(ctx, { in_u8, 1.0 * count, "normalize input" }, { out_f32, 4.0 * count, "normalize output" });
    Scope sc(ctx, s);
    return sc.finish(mbn_launch_normalize(ctx, s, (float *)out_f32, (const uint8_t *)in_u8, count, scale, bias));
}

}   // extern "C"
