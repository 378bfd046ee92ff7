"""Python mirror of the C-ABI in include/mbn.h (ctypes; no torch types cross the boundary).

The product is the C/HIP library (libmbn.so, built by the Makefile next to this file); this module only
loads it and exposes thin wrappers so tests and bench.py can drive it. There is no CPU fallback:
`load()` raises if the library is missing, and `Context()` raises if there is no HIP device.

The directory name contains '-', so import it with `import_package()` from the repo-root `mbn_amd.py` shim
(or importlib on this file).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(PKG_DIR)
# MBN_LAB=1 in the environment selects the lab build (every A/B variant and mbn_tune_set knob: `make lab`); tools/*.py set it,
# bench.py and the tests run the shipped library unless the caller exports it
# (MBN_LAB=<file name>: another build of the lab library in the package directory — tools' A/B of two source states in one GPU call)
_lab = os.environ.get("MBN_LAB", "")
LIB_PATH = os.path.join(PKG_DIR, "libmbn_lab.so" if _lab == "1" else _lab if (_lab.startswith("libmbn") and _lab.endswith(".so")) else "libmbn.so")
HOST_LIB_PATH = os.path.join(PKG_DIR, "libmbn_host.so")
HEADER = os.path.join(REPO_ROOT, "include", "mbn.h")

RANK_FN = C.CFUNCTYPE(C.c_int, C.c_int, C.c_void_p, C.c_void_p)     # mbn_rank_fn(rank, arg, sync)
OK, EINVAL, ENOMEM, EDEVICE, EIO, EFORMAT, ENOTFOUND, ESHAPE, EUNSUPPORTED, ENODEVICE = 0, -1, -2, -3, -4, -5, -6, -7, -8, -9
DT_U8, DT_F32, DT_BF16 = 0, 1, 2
LAYOUT_NCHW_PLANAR, LAYOUT_NHWC = 0, 1
IO_IN_F32, IO_OUT_F32, IO_IN_U8, IO_FILT_PACKED = 1, 2, 4, 8
ACT_NONE, ACT_RELU, ACT_RELU6 = 0, 1, 2
Q_CARRY_SUM, Q_DW_PLANE0, Q_LITERAL_INDEX, Q_POOL_DIV49 = 1, 2, 4, 8
QUIRKS_NONE, QUIRKS_KERNEL_CL = 0, 0xF
L_CONV, L_DW, L_PW, L_POOL, L_FC = 1, 2, 3, 4, 5
MAX_LAYERS = 32


class MbnError(RuntimeError):
    def __init__(self, code, what=""):
        self.code = code
        super().__init__("mbn error %d (%s) %s" % (code, _strerror(code), what))


class BlockParams(C.Structure):
    """mbn_block_params (include/mbn.h): one depthwise + pointwise block of mbn_blocks_resident_bf16"""
    _fields_ = [("wd", C.c_void_p), ("s2", C.c_void_p), ("b2", C.c_void_p), ("wp_bf16", C.c_void_p), ("s3", C.c_void_p), ("b3", C.c_void_p)]


class LayerExt(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("batch", C.c_int32), ("dtype", C.c_int32), ("layout", C.c_int32),
                ("act", C.c_int32), ("pad_top", C.c_int32), ("pad_left", C.c_int32), ("in_rows", C.c_int32),
                ("in_cols", C.c_int32), ("cin", C.c_int32), ("gsize0", C.c_int32), ("gsize1", C.c_int32),
                ("quirks", C.c_uint32), ("quirks_valid", C.c_int32), ("scale", C.c_void_p), ("shift", C.c_void_p),
                ("stream", C.c_void_p), ("io_flags", C.c_int32), ("reserved", C.c_int32)]


class LayerDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("index", "kind", "in_rows", "in_cols", "in_ch", "out_rows", "out_cols",
                                        "out_ch", "stride", "pad_top", "pad_left")] + \
               [(n, C.c_int64) for n in ("w_offset", "w_count", "scale_offset", "shift_offset")]


class Plan(C.Structure):
    _fields_ = [("n_layers", C.c_int32), ("res", C.c_int32), ("alpha", C.c_float), ("classes", C.c_int32),
                ("blob_floats", C.c_int64), ("max_act_floats", C.c_int64), ("layer", LayerDesc * MAX_LAYERS)]


class Weights(C.Structure):
    _fields_ = [("plan", Plan), ("blob", C.POINTER(C.c_float))]


def build(force: bool = False) -> None:
    """Compile the HIP kernels for gfx950 + the C host (make; hipcc cross-compiles without a GPU): the shipped library and
    the lab build beside it (`all lab`)."""
    args = ["make", "-C", PKG_DIR, "-j8", "all", "lab"]
    if force:
        subprocess.check_call(["make", "-C", PKG_DIR, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(args, stdout=subprocess.DEVNULL)


_lib = None
_host = None


def _declare_host(lib):
    lib.mbn_strerror.restype = C.c_char_p
    lib.mbn_strerror.argtypes = [C.c_int]
    lib.mbn_plan_build.argtypes = [C.c_float, C.c_int, C.c_int, C.POINTER(Plan)]
    lib.mbn_weights_from_h5.argtypes = [C.c_char_p, C.c_float, C.c_int, C.POINTER(Weights)]
    lib.mbn_weights_synthetic_h5.argtypes = [C.c_char_p, C.c_float, C.c_int, C.c_uint64]
    lib.mbn_weights_free.argtypes = [C.POINTER(Weights)]
    lib.mbn_h5_open.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
    lib.mbn_h5_close.argtypes = [C.c_void_p]
    lib.mbn_h5_get.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int64),
                               C.POINTER(C.POINTER(C.c_float))]
    lib.mbn_h5_visit.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.mbn_h5_create.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
    lib.mbn_h5_put.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.POINTER(C.c_int64), C.c_void_p]
    lib.mbn_h5_finish.argtypes = [C.c_void_p]
    lib.mbn_read_text_weights.argtypes = [C.c_char_p, C.c_void_p, C.c_int]
    lib.mbn_read_text_weights_f32.argtypes = [C.c_char_p, C.c_void_p, C.c_size_t, C.c_size_t]
    lib.mbn_read_ppm.argtypes = [C.c_char_p, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int]
    lib.mbn_write_ppm.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_int]
    lib.mbn_split_rgb.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.mbn_softmax_argmax_u8.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_double)]
    lib.readSquezeNetKernel.argtypes = [C.c_void_p, C.c_int]
    lib.readSquezeNetKernel.restype = None
    lib.decode_image.argtypes = [C.c_void_p, C.c_char_p]
    lib.mbn_shard_range.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.mbn_run_ranks.argtypes = [C.c_int, RANK_FN, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    lib.mbn_rank_barrier.argtypes = [C.c_void_p]
    lib.mbn_rank_fail.argtypes = [C.c_void_p]
    return lib


def host_lib():
    """The C host alone (no HIP): plan, loaders, .h5 reader/writer. Usable on a CPU-only box."""
    global _host
    if _host is None:
        if not os.path.exists(HOST_LIB_PATH):
            raise FileNotFoundError("%s missing: run __graft_entry__.build() / make" % HOST_LIB_PATH)
        _host = _declare_host(C.CDLL(HOST_LIB_PATH))
    return _host


def load():
    """Load libmbn.so (HIP kernels + C-ABI + C host). Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError("%s missing: the HIP extension is not built (no fallback exists)" % LIB_PATH)
        lib = _declare_host(C.CDLL(LIB_PATH))
        vp, ci = C.c_void_p, C.c_int
        lib.mbn_version.restype = C.c_char_p
        lib.mbn_last_device_error.restype = C.c_char_p
        lib.mbn_last_device_error.argtypes = [vp]
        lib.mbn_init.argtypes = [ci, C.POINTER(vp)]
        lib.mbn_shutdown.argtypes = [vp]
        lib.mbn_device_count.argtypes = [C.POINTER(ci)]
        lib.mbn_device_name.argtypes = [vp, C.c_char_p, C.c_size_t]
        lib.mbn_device_cus.argtypes = [vp, C.POINTER(C.c_int)]
        lib.mbn_device_pci_bus_id.argtypes = [vp, C.c_char_p, C.c_size_t]
        lib.mbn_pw_clock_read.argtypes = [vp, ci, C.POINTER(C.c_double), C.POINTER(C.c_longlong)]
        lib.mbn_set_literal_quirks.argtypes = [vp, C.c_uint32]
        lib.mbn_get_stream.argtypes = [vp, C.POINTER(vp)]
        lib.mbn_alloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
        lib.mbn_free.argtypes = [vp, vp]
        lib.mbn_upload.argtypes = [vp, vp, vp, C.c_size_t]
        lib.mbn_download.argtypes = [vp, vp, vp, C.c_size_t]
        lib.mbn_memset.argtypes = [vp, vp, ci, C.c_size_t]
        lib.mbn_sync.argtypes = [vp]
        lib.mbn_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
        lib.mbn_set_profiling.argtypes = [vp, ci]
        lib.mbn_profile_begin.argtypes = [vp, ci]
        lib.mbn_profile_end.argtypes = [vp, C.POINTER(C.c_float), ci, C.POINTER(ci)]
        lib.mbn_profile_pause.argtypes = [vp, ci]
        lib.mbn_profile_null.argtypes = [vp, ci, vp]
        lib.mbn_mark.argtypes = [vp, vp]
        lib.mbn_dist_init.argtypes = [ci, C.POINTER(ci), C.POINTER(vp)]
        lib.mbn_dist_size.argtypes = [vp, C.POINTER(ci)]
        lib.mbn_dist_context.argtypes = [vp, ci, C.POINTER(vp)]
        lib.mbn_dist_broadcast.argtypes = [vp, C.POINTER(vp), C.c_size_t, ci]
        lib.mbn_dist_sync.argtypes = [vp]
        lib.mbn_dist_shutdown.argtypes = [vp]
        lib.mbn_dist_last_error.restype = C.c_char_p
        lib.mbn_dist_last_error.argtypes = [vp]
        lib.mbn_marks_read.argtypes = [vp, C.POINTER(C.c_float), ci, C.POINTER(ci)]
        ext = C.POINTER(LayerExt)
        lib.mbn_convolute.argtypes = [vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ext]
        lib.mbn_depthwise.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci, ci, ext]
        lib.mbn_pointwise.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci, ext]
        lib.mbn_pool.argtypes = [vp, vp, vp, ci, ci, ci, ci, ext]
        lib.mbn_softmax_f32.argtypes = [vp, vp, vp, vp, ci, ci, vp]
        lib.mbn_normalize_u8_to_f32.argtypes = [vp, vp, vp, C.c_size_t, C.c_float, C.c_float, vp]
        lib.mbn_convert_f32_to_bf16.argtypes = [vp, vp, vp, C.c_size_t, vp]
        lib.mbn_packed_filter_offset.restype = C.c_size_t
        lib.mbn_packed_filter_offset.argtypes = [ci, ci]
        lib.mbn_pack_filter_bf16.argtypes = [vp, vp, ci, ci, vp]
        lib.mbn_convert_bf16_to_f32.argtypes = [vp, vp, vp, C.c_size_t, vp]
        lib.mbn_net_set_dtype.argtypes = [vp, ci]
        lib.mbn_tune_set.argtypes = [C.c_char_p, ci]
        lib.mbn_tune_get.argtypes = [C.c_char_p, C.POINTER(ci)]
        lib.mbn_net_set_streams.argtypes = [vp, ci]
        lib.mbn_net_set_free_running.argtypes = [vp, ci]
        lib.mbn_net_set_graph.argtypes = [vp, ci]
        lib.mbn_net_set_fuse_stem.argtypes = [vp, ci]
        lib.mbn_net_fused_layers.argtypes = [vp, ci, C.POINTER(ci)]
        lib.mbn_net_set_fuse_blocks.argtypes = [vp, C.c_uint]
        lib.mbn_net_get_fuse_blocks.argtypes = [vp, C.POINTER(C.c_uint)]
        lib.mbn_net_reset_fuse_blocks.argtypes = [vp]
        lib.mbn_net_set_fuse_tail.argtypes = [vp, ci]
        lib.mbn_pool_fc_workspace_bytes.argtypes = [ci, ci]
        lib.mbn_pool_fc_workspace_bytes.restype = C.c_size_t
        lib.mbn_pool_fc.argtypes = [vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, vp, C.c_size_t, vp]
        lib.mbn_classifier_tail_fused.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, vp, C.c_size_t, vp]
        lib.mbn_forget.argtypes = [vp, vp, C.c_size_t]
        lib.mbn_net_classify.argtypes = [vp, vp, ci, ci, vp, vp]
        lib.mbn_net_launches.argtypes = [vp, ci, ci, C.POINTER(ci), C.POINTER(ci), ci, C.POINTER(ci)]
        lib.mbn_stem_fused.argtypes = [vp] + [vp] * 11 + [ci, ci, ci, ci, vp]
        lib.mbn_stem_fused_u8.argtypes = [vp] + [vp] * 11 + [ci, ci, ci, ci, vp]
        lib.mbn_stem_fused_ex.argtypes = [vp] + [vp] * 11 + [ci, ci, ci, ci, ci, vp]
        lib.mbn_net_set_input_u8.argtypes = [vp, ci]
        lib.mbn_net_set_fuse_resident.argtypes = [vp, ci]
        lib.mbn_dwpw_fused.argtypes = [vp] + [vp] * 8 + [ci] * 10 + [vp]
        lib.mbn_dwpw_fused_bf16.argtypes = [vp] + [vp] * 8 + [ci] * 10 + [vp]
        lib.mbn_blocks_resident_bf16.argtypes = [vp, vp, vp, C.POINTER(BlockParams), ci, ci, ci, ci, ci, vp]
        lib.mbn_tail_resident_bf16.argtypes = [vp, vp, vp, C.POINTER(BlockParams), ci, ci, ci, ci, ci, vp]
        lib.mbn_softmax_topk_f32.argtypes = [vp, vp, vp, vp, vp, ci, ci, ci, vp]
        lib.mbn_classifier_tail.argtypes = [vp] * 9 + [ci] * 6 + [vp]
        lib.mbn_graph_begin.argtypes = [vp, vp]
        lib.mbn_graph_end.argtypes = [vp, vp, C.POINTER(vp)]
        lib.mbn_graph_launch.argtypes = [vp, vp, vp]
        lib.mbn_graph_destroy.argtypes = [vp, vp]
        lib.mbn_stream_create.argtypes = [vp, C.POINTER(vp)]
        lib.mbn_stream_destroy.argtypes = [vp, vp]
        lib.mbn_stream_wait.argtypes = [vp, vp, vp]
        lib.mbn_net_create.argtypes = [vp, C.POINTER(Weights), ci, C.POINTER(vp)]
        lib.mbn_net_create_from_device_blob.argtypes = [vp, C.POINTER(Plan), vp, ci, C.POINTER(vp)]
        lib.mbn_net_destroy.argtypes = [vp]
        lib.mbn_net_forward.argtypes = [vp, vp, vp, ci, ci]
        lib.mbn_net_forward_timed.argtypes = [vp, vp, vp, ci, C.POINTER(C.c_float), ci]
        lib.mbn_net_plan.argtypes = [vp, C.POINTER(Plan)]
        lib.mbn_net_set_keep_activations.argtypes = [vp, ci]
        lib.mbn_net_layer_output.argtypes = [vp, ci, C.POINTER(vp), C.POINTER(C.c_size_t)]
        _lib = lib
    return _lib


def _strerror(code):
    for l in (_lib, _host):
        if l is not None:
            return l.mbn_strerror(code).decode()
    return "?"


def _chk(rc, what=""):
    if rc != OK:
        raise MbnError(rc, what)


def declared_symbols():
    """Every function name declared in include/mbn.h (for the export test)."""
    import re
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b((?:mbn_[a-z0-9_]+)|readSquezeNetKernel|decode_image)\s*\(", src)
    return sorted(set(n for n in names if n not in ("mbn_layer_ext", "mbn_context")))


def make_ext(batch=1, dtype=DT_F32, act=ACT_RELU6, pad_top=-1, pad_left=-1, in_rows=0, in_cols=0, cin=0, scale=None,
             shift=None, quirks=None, gsize=(0, 0), stream=None, io_flags=0) -> LayerExt:
    e = LayerExt()
    e.struct_size = C.sizeof(LayerExt)
    e.batch = batch
    e.dtype = dtype
    e.layout = LAYOUT_NCHW_PLANAR if dtype == DT_U8 else LAYOUT_NHWC
    e.io_flags = io_flags
    e.act = act
    e.pad_top, e.pad_left = pad_top, pad_left
    e.in_rows, e.in_cols, e.cin = in_rows, in_cols, cin
    e.gsize0, e.gsize1 = gsize
    if quirks is not None:
        e.quirks, e.quirks_valid = quirks, 1
    e.scale = scale
    e.shift = shift
    e.stream = stream
    return e


def bf16_bits_to_f32(raw: np.ndarray) -> np.ndarray:
    """uint16 bf16 bit patterns -> float32 values."""
    return (raw.astype(np.uint32) << 16).view(np.float32)


def f32_to_bf16_bits(x: np.ndarray) -> np.ndarray:
    """float32 -> bf16 bit patterns, round to nearest even (host-side helper for tests)."""
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)


def packed_filter_dev(ctx, filt_f32: np.ndarray):
    """Device buffer holding a bf16 pointwise filter [cout][cin] followed by its packed image (mbn_pack_filter_bf16), for calls with
    IO_FILT_PACKED; when the shape has no packed form the buffer holds the plain filter only. Returns (buffer, io_flag)."""
    cout, cin = filt_f32.shape
    off = ctx.lib.mbn_packed_filter_offset(cout, cin)
    bits = f32_to_bf16_bits(filt_f32)
    buf = ctx.alloc(off + bits.nbytes if off else bits.nbytes)
    buf.upload(bits)
    if off:
        _chk(ctx.lib.mbn_pack_filter_bf16(ctx.h, buf.ptr, cout, cin, None), ctx.last_error())
        ctx.sync()
    return buf, (IO_FILT_PACKED if off else 0)


class DeviceBuffer:
    """A device allocation owned by a Context (mbn_alloc / mbn_free)."""

    def __init__(self, ctx: "Context", nbytes: int):
        self.ctx, self.nbytes = ctx, int(nbytes)
        p = C.c_void_p()
        _chk(ctx.lib.mbn_alloc(ctx.h, max(self.nbytes, 1), C.byref(p)), "alloc %d" % nbytes)
        self.ptr = p.value

    def upload(self, arr: np.ndarray):
        a = np.ascontiguousarray(arr)
        assert a.nbytes <= self.nbytes
        _chk(self.ctx.lib.mbn_upload(self.ctx.h, self.ptr, a.ctypes.data, a.nbytes))
        return self

    def download(self, shape, dtype) -> np.ndarray:
        out = np.empty(shape, dtype)
        assert out.nbytes <= self.nbytes
        _chk(self.ctx.lib.mbn_download(self.ctx.h, out.ctypes.data, self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            self.ctx.lib.mbn_free(self.ctx.h, self.ptr)
            self.ptr = None


class Context:
    """mbn_init / mbn_shutdown. Raises MbnError(ENODEVICE) when no MI355X is visible."""

    def __init__(self, device: int = 0):
        self.lib = load()
        h = C.c_void_p()
        _chk(self.lib.mbn_init(device, C.byref(h)), "mbn_init(%d)" % device)
        self.h = h
        self._bufs = []

    def name(self) -> str:
        b = C.create_string_buffer(128)
        self.lib.mbn_device_name(self.h, b, 128)
        return b.value.decode()

    def pci_bus_id(self) -> str:
        b = C.create_string_buffer(64)
        _chk(self.lib.mbn_device_pci_bus_id(self.h, b, 64), self.last_error())
        return b.value.decode()

    def pw_clock(self, reset=True):
        """(GHz, launches): the core clock held inside the pw_gemm launches recorded since the last reset (tune key pw_clock)."""
        g, n = C.c_double(), C.c_longlong()
        _chk(self.lib.mbn_pw_clock_read(self.h, int(reset), C.byref(g), C.byref(n)), self.last_error())
        return g.value, n.value

    def alloc(self, nbytes) -> DeviceBuffer:
        b = DeviceBuffer(self, nbytes)
        self._bufs.append(b)
        return b

    def to_device(self, arr: np.ndarray) -> DeviceBuffer:
        a = np.ascontiguousarray(arr)
        return self.alloc(a.nbytes).upload(a)

    def sync(self):
        _chk(self.lib.mbn_sync(self.h), self.last_error())

    def last_error(self) -> str:
        return self.lib.mbn_last_device_error(self.h).decode()

    def stream(self) -> int:
        s = C.c_void_p()
        _chk(self.lib.mbn_get_stream(self.h, C.byref(s)))
        return s.value or 0

    def profile_begin(self, capacity: int):
        _chk(self.lib.mbn_profile_begin(self.h, capacity))

    def profile_pause(self, paused: bool):
        _chk(self.lib.mbn_profile_pause(self.h, int(paused)))

    def profile_end(self, capacity: int):
        ms = (C.c_float * capacity)()
        n = C.c_int()
        _chk(self.lib.mbn_profile_end(self.h, ms, capacity, C.byref(n)), self.last_error())
        return [ms[i] for i in range(n.value)]

    def profile_null_us(self, with_kernel: bool, n: int = 200) -> float:
        """Median reading (microseconds) of an event pair recorded around nothing / around an empty kernel (mbn_profile_null)."""
        self.profile_begin(n)
        for _ in range(n):
            _chk(self.lib.mbn_profile_null(self.h, int(with_kernel), None))
        ms = sorted(self.profile_end(n))
        return 1000.0 * ms[len(ms) // 2]

    def mark(self, stream=None):
        _chk(self.lib.mbn_mark(self.h, stream))

    def marks_read(self, capacity: int):
        ms = (C.c_float * max(capacity, 1))()
        n = C.c_int()
        _chk(self.lib.mbn_marks_read(self.h, ms, capacity, C.byref(n)), self.last_error())
        return [ms[i] for i in range(n.value)]

    def close(self):
        if self.h:
            self.lib.mbn_shutdown(self.h)
            self.h = None

    # ---- layer calls: positional arguments are kernel.cl's ----
    def convolute(self, out, inp_r, inp_g, inp_b, filt, rows, cols, filtersize, stride, op_size, ext=None):
        _chk(self.lib.mbn_convolute(self.h, out, inp_r, inp_g, inp_b, filt, rows, cols, filtersize, stride, op_size,
                                    None if ext is None else C.byref(ext)), self.last_error())

    def depthwise(self, out, inp, filt, rows, cols, filtersize, stride, op_size, ext=None):
        _chk(self.lib.mbn_depthwise(self.h, out, inp, filt, rows, cols, filtersize, stride, op_size,
                                    None if ext is None else C.byref(ext)), self.last_error())

    def pointwise(self, out, inp, filt, rows, cols, filtersize, op_size, ext=None):
        _chk(self.lib.mbn_pointwise(self.h, out, inp, filt, rows, cols, filtersize, op_size,
                                    None if ext is None else C.byref(ext)), self.last_error())

    def pool(self, out, inp, rows, cols, filtersize, op_size, ext=None):
        _chk(self.lib.mbn_pool(self.h, out, inp, rows, cols, filtersize, op_size,
                               None if ext is None else C.byref(ext)), self.last_error())

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def plan_build(alpha=1.0, res=224, classes=1000, lib=None) -> Plan:
    p = Plan()
    _chk((lib or host_lib()).mbn_plan_build(alpha, res, classes, C.byref(p)))
    return p


class HostWeights:
    """mbn_weights_from_h5 result; .blob is a numpy view of the packed fp32 parameters."""

    def __init__(self, path, alpha=0.0, res=224, lib=None):
        self.lib = lib or host_lib()
        self.w = Weights()
        _chk(self.lib.mbn_weights_from_h5(path.encode(), alpha, res, C.byref(self.w)), path)
        self.plan = self.w.plan
        self.blob = np.ctypeslib.as_array(self.w.blob, shape=(self.plan.blob_floats,))

    def free(self):
        self.lib.mbn_weights_free(C.byref(self.w))


def synthetic_h5(path, alpha=1.0, classes=1000, seed=0xC0FFEE, lib=None):
    _chk((lib or host_lib()).mbn_weights_synthetic_h5(path.encode(), alpha, classes, seed), path)


class Net:
    """mbn_net_*: the whole-network runner. `blob` may be a numpy array (uploaded once) or a device pointer."""

    def __init__(self, ctx: Context, plan: Plan, blob, max_batch: int):
        self.ctx, self.plan, self.max_batch = ctx, plan, max_batch
        h = C.c_void_p()
        if isinstance(blob, np.ndarray):
            self._dev_blob = ctx.to_device(np.ascontiguousarray(blob, np.float32))
            dev = self._dev_blob.ptr
        else:
            dev = int(blob)
        _chk(ctx.lib.mbn_net_create_from_device_blob(ctx.h, C.byref(plan), dev, max_batch, C.byref(h)), ctx.last_error())
        self.h = h

    def forward(self, images_ptr, out_ptr, batch, last_layer=0):
        _chk(self.ctx.lib.mbn_net_forward(self.h, images_ptr, out_ptr, batch, last_layer), self.ctx.last_error())

    def forward_timed(self, images_ptr, out_ptr, batch):
        ms = (C.c_float * MAX_LAYERS)()
        _chk(self.ctx.lib.mbn_net_forward_timed(self.h, images_ptr, out_ptr, batch, ms, MAX_LAYERS), self.ctx.last_error())
        return [ms[i] for i in range(self.plan.n_layers)]

    def set_streams(self, n, free_running=False):
        _chk(self.ctx.lib.mbn_net_set_streams(self.h, n), self.ctx.last_error())
        _chk(self.ctx.lib.mbn_net_set_free_running(self.h, int(free_running)))

    def set_fuse_stem(self, enabled=True):
        _chk(self.ctx.lib.mbn_net_set_fuse_stem(self.h, int(enabled)))

    def fused_layers(self, last_layer=0) -> int:
        n = C.c_int()
        _chk(self.ctx.lib.mbn_net_fused_layers(self.h, last_layer, C.byref(n)))
        return n.value

    def classify(self, images_dev, batch, k, topk_idx_dev, topk_prob_dev):
        _chk(self.ctx.lib.mbn_net_classify(self.h, images_dev, batch, k, topk_idx_dev, topk_prob_dev), self.ctx.last_error())

    def set_input_u8(self, enabled=True):
        _chk(self.ctx.lib.mbn_net_set_input_u8(self.h, int(enabled)))

    def set_fuse_blocks(self, mask):
        _chk(self.ctx.lib.mbn_net_set_fuse_blocks(self.h, int(mask)))

    def reset_fuse_blocks(self):
        _chk(self.ctx.lib.mbn_net_reset_fuse_blocks(self.h))

    def set_fuse_tail(self, enabled=True):
        _chk(self.ctx.lib.mbn_net_set_fuse_tail(self.h, int(enabled)))

    def set_fuse_resident(self, enabled=True):
        """Runs of equal small-map bf16 blocks as one launch with the map resident in LDS (default on)."""
        _chk(self.ctx.lib.mbn_net_set_fuse_resident(self.h, int(enabled)))

    def get_fuse_blocks(self) -> int:
        m = C.c_uint()
        _chk(self.ctx.lib.mbn_net_get_fuse_blocks(self.h, C.byref(m)))
        return m.value

    def launches(self, batch, last_layer=0):
        """[(first_layer, n_layers), ...] of the launches one forward issues per (sub-)batch."""
        cap = 64
        a, b, n = (C.c_int * cap)(), (C.c_int * cap)(), C.c_int()
        _chk(self.ctx.lib.mbn_net_launches(self.h, batch, last_layer, a, b, cap, C.byref(n)))
        return [(a[i], b[i]) for i in range(min(n.value, cap))]

    def set_graph(self, enabled=True):
        _chk(self.ctx.lib.mbn_net_set_graph(self.h, int(enabled)), self.ctx.last_error())

    def set_dtype(self, dtype):
        self.dtype = dtype
        _chk(self.ctx.lib.mbn_net_set_dtype(self.h, dtype), self.ctx.last_error())

    def keep_activations(self, keep=True):
        _chk(self.ctx.lib.mbn_net_set_keep_activations(self.h, int(keep)))

    def layer_output(self, index, batch, images=None) -> np.ndarray:
        """Kept activation of layer `index` (1-based): the whole batch, or only the listed images (full-size batches: a few images
        of an 800 MB tensor)."""
        p, n = C.c_void_p(), C.c_size_t()
        _chk(self.ctx.lib.mbn_net_layer_output(self.h, index, C.byref(p), C.byref(n)))
        l = self.plan.layer[index - 1]
        bf = getattr(self, "dtype", DT_F32) == DT_BF16 and l.kind != L_FC
        per = (l.out_rows, l.out_cols, l.out_ch)
        raw = np.empty((batch if images is None else len(images),) + per, np.uint16 if bf else np.float32)
        if images is None:
            _chk(self.ctx.lib.mbn_download(self.ctx.h, raw.ctypes.data, p, raw.nbytes))
        else:
            step = raw[0].nbytes
            for j, im in enumerate(images):
                assert 0 <= im < batch
                _chk(self.ctx.lib.mbn_download(self.ctx.h, raw[j].ctypes.data, p.value + im * step, step))
        return bf16_bits_to_f32(raw) if bf else raw

    def destroy(self):
        if self.h:
            self.ctx.lib.mbn_net_destroy(self.h)
            self.h = None
