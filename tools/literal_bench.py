#!/usr/bin/env python3
"""LITERAL (uint8 / int32, kernel.cl semantics, quirks off) pointwise: the int8 MFMA form (round 5, default) and the v_dot4_i32_i8 path (tune lit_dot=2) against the
scalar kernel (lit_dot=1), at the reference's layer shapes, one image as the reference processes it — and a batch of 64 images, where the forms differ."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("MBN_LAB", "1")      # the lab build: every A/B variant and mbn_tune_set knob (make lab)
sys.path.insert(0, ROOT)
from mbn_amd import import_package
pkg = import_package(); lib = pkg.load(); ctx = pkg.Context(0)
rng = np.random.default_rng(0)
for nb in (1, 64):
    print("batch %d" % nb)
    print("%-34s %12s %12s %12s %10s %10s" % ("pointwise layer (rows cin -> cout)", "scalar ms", "v_dot4 ms", "int8 MFMA ms", "MFMA/dot4", "GMAC/s"))
    for name, rows, cin, oc in (("L3", 112, 32, 64), ("L7", 56, 128, 128), ("L11", 28, 256, 256), ("L15", 14, 512, 512), ("L25", 7, 512, 1024), ("L27", 7, 1024, 1024), ("L29 (FC)", 1, 1024, 1000)):
        x = rng.integers(0, 256, (nb, cin, rows, rows), dtype=np.uint8)
        f = rng.integers(-4, 5, (oc, cin), dtype=np.int32)
        d_x, d_f, d_o = ctx.to_device(x), ctx.to_device(f), ctx.alloc(nb * oc * rows * rows)
        ext = pkg.make_ext(batch=nb, dtype=pkg.DT_U8, quirks=0)
        t = {}
        for mode in ((1, 2, 0) if nb == 1 else (2, 0)):
            lib.mbn_tune_set(b"lit_dot", mode)
            for _ in range(3):
                ctx.pointwise(d_o.ptr, d_x.ptr, d_f.ptr, rows, rows, cin, oc, ext)
            ctx.sync()
            reps = 10
            ctx.profile_begin(reps)
            for _ in range(reps):
                ctx.pointwise(d_o.ptr, d_x.ptr, d_f.ptr, rows, rows, cin, oc, ext)
            t[mode] = float(np.median(ctx.profile_end(reps)))
        lib.mbn_tune_set(b"lit_dot", 0)
        print("%-34s %12s %12.4f %12.4f %9.1fx %10.1f" % ("%s  %dx%d %d -> %d" % (name, rows, rows, cin, oc), ("%.4f" % t[1]) if 1 in t else "-", t[2], t[0], t[2] / t[0],
                                                           nb * rows * rows * cin * oc / t[0] / 1e6))
        for b in (d_x, d_f, d_o):
            b.free()

print()
print("%-34s %12s %12s %8s" % ("3x3 layer, quirks off", "tap-by-tap ms", "v_dot4 ms", "speedup"))
for name, rows, ch, stride in (("L2 depthwise", 112, 32, 1), ("L4 depthwise", 56, 64, 2), ("L6 depthwise", 56, 128, 1), ("L14 depthwise", 14, 512, 1), ("L26 depthwise", 7, 1024, 1)):
    x = rng.integers(0, 256, (ch, rows * stride, rows * stride), dtype=np.uint8)
    f = rng.integers(-4, 5, (ch, 3, 3), dtype=np.int32)
    d_x, d_f, d_o = ctx.to_device(x), ctx.to_device(f), ctx.alloc(ch * rows * rows)
    ext = pkg.make_ext(dtype=pkg.DT_U8, quirks=0)
    t = {}
    for mode in (1, 0):
        lib.mbn_tune_set(b"lit_dot", mode)
        for _ in range(3):
            ctx.depthwise(d_o.ptr, d_x.ptr, d_f.ptr, rows, rows, 3, stride, ch, ext)
        ctx.sync()
        reps = 10
        ctx.profile_begin(reps)
        for _ in range(reps):
            ctx.depthwise(d_o.ptr, d_x.ptr, d_f.ptr, rows, rows, 3, stride, ch, ext)
        t[mode] = float(np.median(ctx.profile_end(reps)))
    lib.mbn_tune_set(b"lit_dot", 0)
    print("%-34s %12.4f %12.4f %7.1fx" % ("%s  %dx%d x%d s%d" % (name, rows, rows, ch, stride), t[1], t[0], t[1] / t[0]))
r, g, b = (rng.integers(0, 256, 224 * 224, dtype=np.uint8) for _ in range(3))
f = rng.integers(-4, 5, (32, 3, 3, 3), dtype=np.int32)
d = [ctx.to_device(a) for a in (r, g, b, f)]
d_o = ctx.alloc(112 * 112 * 32)
t = {}
for mode in (1, 0):
    lib.mbn_tune_set(b"lit_dot", mode)
    reps = 10
    for _ in range(3):
        ctx.convolute(d_o.ptr, d[0].ptr, d[1].ptr, d[2].ptr, d[3].ptr, 224, 224, 3, 2, 32, pkg.make_ext(dtype=pkg.DT_U8, quirks=0))
    ctx.sync()
    ctx.profile_begin(reps)
    for _ in range(reps):
        ctx.convolute(d_o.ptr, d[0].ptr, d[1].ptr, d[2].ptr, d[3].ptr, 224, 224, 3, 2, 32, pkg.make_ext(dtype=pkg.DT_U8, quirks=0))
    t[mode] = float(np.median(ctx.profile_end(reps)))
lib.mbn_tune_set(b"lit_dot", 0)
print("%-34s %12.4f %12.4f %7.1fx" % ("L1 convolute 224x224x3 -> 32", t[1], t[0], t[1] / t[0]))
