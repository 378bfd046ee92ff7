"""ctypes/numpy front-end of the CPU oracle (oracle/mbn_oracle.c). TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: see oracle/mbn_oracle.h. Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; the product never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmbn_oracle.so")

Q_CARRY_SUM, Q_DW_PLANE0, Q_LITERAL_INDEX, Q_POOL_DIV49 = 1, 2, 4, 8
QUIRKS_KERNEL_CL = 0xF
QUIRKS_NONE = 0
ACT_NONE, ACT_RELU, ACT_RELU6 = 0, 1, 2
L_CONV, L_DW, L_PW, L_POOL, L_FC = 1, 2, 3, 4, 5


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "mbn_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libmbn_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


class OrcLayer(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("index", "kind", "in_rows", "in_cols", "in_ch", "out_rows", "out_cols",
                                      "out_ch", "stride", "pad_top", "pad_left")] + \
               [(n, C.c_long) for n in ("w_offset", "w_count", "scale_offset", "shift_offset")]


class OrcPlan(C.Structure):
    _fields_ = [("n_layers", C.c_int), ("res", C.c_int), ("classes", C.c_int), ("alpha", C.c_float),
                ("blob_floats", C.c_long), ("max_act_floats", C.c_long), ("layer", OrcLayer * 32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_plan_build.argtypes = [C.c_float, C.c_int, C.c_int, C.POINTER(OrcPlan)]
        _lib.orc_plan_build.restype = C.c_int
        _lib.orc_net_forward.restype = C.c_int
        _lib.orc_num_threads.restype = C.c_int
        _lib.orc_same_pad.restype = C.c_int
    return _lib


def _p(a, t=C.c_void_p):
    return None if a is None else a.ctypes.data_as(t)


def _u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


# ----------------------------------------------------------------------------- LITERAL

def lit_convolute(inp_r, inp_g, inp_b, filt, rows, cols, filtersize, stride, op_size, quirks=QUIRKS_KERNEL_CL,
                  g0=0, g1=0):
    """kernel.cl:2-60. planes: uint8 [rows*cols]; filt int32 [op_size][3][k][k]. Returns uint8 [op_size][orow][ocol]."""
    r, g, b, f = _u8(inp_r), _u8(inp_g), _u8(inp_b), _i32(filt)
    plane = (rows // 2) * (cols // 2) if quirks & Q_LITERAL_INDEX else (rows // stride) * (cols // stride)
    out = np.zeros(op_size * plane, np.uint8)
    lib().orc_lit_convolute(_p(out), _p(r), _p(g), _p(b), _p(f), rows, cols, filtersize, stride, op_size,
                            C.c_uint(quirks), g0, g1)
    return out


def lit_depthwise(inp, filt, rows, cols, filtersize, stride, op_size, in_rows=0, in_cols=0,
                  quirks=QUIRKS_KERNEL_CL, g0=0, g1=0):
    """kernel.cl:62-92. inp uint8 [C][in_rows][in_cols]; filt int32 [C][k][k]. Returns uint8 [C*rows*cols]."""
    i, f = _u8(inp), _i32(filt)
    out = np.zeros(op_size * rows * cols, np.uint8)
    lib().orc_lit_depthwise(_p(out), _p(i), _p(f), rows, cols, filtersize, stride, op_size, in_rows, in_cols,
                            C.c_uint(quirks), g0, g1)
    return out


def lit_pointwise(inp, filt, rows, cols, filtersize, op_size, quirks=QUIRKS_KERNEL_CL):
    """kernel.cl:94-114. inp uint8 [Cin][rows][cols]; filt int32 [op_size][filtersize]."""
    i, f = _u8(inp), _i32(filt)
    out = np.zeros(op_size * rows * cols, np.uint8)
    lib().orc_lit_pointwise(_p(out), _p(i), _p(f), rows, cols, filtersize, op_size, C.c_uint(quirks))
    return out


def lit_pool(inp, rows, cols, filtersize, op_size, quirks=QUIRKS_KERNEL_CL):
    """kernel.cl:116-132 (work-item (0,0))."""
    i = _u8(inp)
    out = np.zeros(op_size, np.uint8)
    lib().orc_lit_pool(_p(out), _p(i), rows, cols, filtersize, op_size, C.c_uint(quirks))
    return out


def softmax_argmax_u8(logits):
    """MobileNet.c:2771-2792 -> (probs float64[n], location (1-based), maximum)."""
    l = _u8(logits)
    probs = np.zeros(l.size, np.float64)
    loc, mx = C.c_int(0), C.c_double(0)
    lib().orc_softmax_argmax_u8(_p(l), int(l.size), _p(probs), C.byref(loc), C.byref(mx))
    return probs, loc.value, mx.value


# ----------------------------------------------------------------------------- F32

def f32_conv(x, filt, scale, shift, stride, act, pad_top=-1, pad_left=-1):
    """x [N][H][W][Cin], filt [k][k][Cin][Cout] -> [N][ceil(H/s)][ceil(W/s)][Cout]."""
    x, filt, scale, shift = _f32(x), _f32(filt), _f32(scale), _f32(shift)
    n, h, w, cin = x.shape
    k, cout = filt.shape[0], filt.shape[3]
    oh, ow = -(-h // stride), -(-w // stride)
    out = np.empty((n, oh, ow, cout), np.float32)
    lib().orc_f32_conv(_p(out), _p(x), _p(filt), _p(scale), _p(shift), n, h, w, cin, k, stride, cout, pad_top,
                       pad_left, act)
    return out


def f32_depthwise(x, filt, scale, shift, stride, act, out_rows=0, out_cols=0, pad_top=-1, pad_left=-1):
    """x [N][H][W][C], filt [k][k][C] -> [N][rows][cols][C] (default ceil(H/s))."""
    x, filt, scale, shift = _f32(x), _f32(filt), _f32(scale), _f32(shift)
    n, h, w, c = x.shape
    k = filt.shape[0]
    rows = out_rows or -(-h // stride)
    cols = out_cols or -(-w // stride)
    out = np.empty((n, rows, cols, c), np.float32)
    lib().orc_f32_depthwise(_p(out), _p(x), _p(filt), _p(scale), _p(shift), n, rows, cols, h, w, k, stride, c,
                            pad_top, pad_left, act)
    return out


def f32_pointwise(x, filt, scale, shift, act):
    """x [..., Cin], filt [Cout][Cin] -> [..., Cout]."""
    x, filt, scale, shift = _f32(x), _f32(filt), _f32(scale), _f32(shift)
    cin = x.shape[-1]
    cout = filt.shape[0]
    m = x.size // cin
    out = np.empty(x.shape[:-1] + (cout,), np.float32)
    lib().orc_f32_pointwise(_p(out), _p(x), _p(filt), _p(scale), _p(shift), C.c_long(m), cin, cout, act)
    return out


def f32_pool(x, filtersize=0):
    x = _f32(x)
    n, h, w, c = x.shape
    out = np.empty((n, c), np.float32)
    lib().orc_f32_pool(_p(out), _p(x), n, h, w, filtersize or h, c)
    return out


def f32_softmax(logits):
    l = _f32(logits)
    n, k = l.shape
    probs = np.empty((n, k), np.float32)
    am = np.empty(n, np.int32)
    lib().orc_f32_softmax(_p(probs), _p(am), _p(l), n, k)
    return probs, am


# ----------------------------------------------------------------------------- topology / whole net

def plan_build(alpha=1.0, res=224, classes=1000) -> OrcPlan:
    p = OrcPlan()
    rc = lib().orc_plan_build(C.c_float(alpha), res, classes, C.byref(p))
    if rc != 0:
        raise ValueError("orc_plan_build failed: %d" % rc)
    return p


def bf16_round(a):
    """Round-to-nearest-even to bf16 precision, returned as float32."""
    a = np.array(a, dtype=np.float32, copy=True)
    lib().orc_bf16_round_array(_p(a), C.c_long(a.size))
    return a


def net_forward(plan: OrcPlan, blob, images, last_layer=0, threads=1, keep_layers=False, bf16=False):
    """Returns (out, [per-layer outputs] or None). images [N][res][res][3] fp32."""
    blob, images = _f32(blob), _f32(images)
    n = images.shape[0]
    ll = last_layer if 0 < last_layer <= plan.n_layers else plan.n_layers
    L = plan.layer[ll - 1]
    out = np.empty((n, L.out_rows, L.out_cols, L.out_ch), np.float32)
    layer_arrays, ptrs = None, None
    if keep_layers:
        layer_arrays = []
        ptrs = (C.c_void_p * plan.n_layers)()
        for i in range(plan.n_layers):
            l = plan.layer[i]
            if i < ll:
                a = np.empty((n, l.out_rows, l.out_cols, l.out_ch), np.float32)
                layer_arrays.append(a)
                ptrs[i] = a.ctypes.data
            else:
                ptrs[i] = None
    fn = lib().orc_net_forward_bf16 if bf16 else lib().orc_net_forward
    rc = fn(C.byref(plan), _p(blob), _p(images), _p(out), n, ll, threads, ptrs)
    if rc != 0:
        raise RuntimeError("orc_net_forward failed: %d" % rc)
    return out, layer_arrays


def num_threads() -> int:
    return lib().orc_num_threads()
