#!/usr/bin/env python3
"""Classifier tail (MobileNet.c:2601-2792) at 1...4 images: three launches (mbn_classifier_tail: pool, FC, softmax + top-k), two (mbn_pool_fc +
mbn_softmax_topk_f32) and one (mbn_classifier_tail_fused). GPU time per tail = wall time of 2000 back-to-back tails on one stream / 2000."""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mbn_amd import import_package
pkg = import_package()
lib = pkg.load()
rng = np.random.default_rng(3)
ch, classes, h, k = 1024, 1000, 7, 5
with pkg.Context(0) as ctx:
    w = rng.normal(0, (1.0 / ch) ** 0.5, (classes, ch)).astype(np.float32)
    b = rng.normal(0, 0.5, classes).astype(np.float32)
    d_w, d_b = ctx.to_device(w), ctx.to_device(b)
    nb = lib.mbn_pool_fc_workspace_bytes(ch, classes)
    d_ws = ctx.alloc(nb)
    lib.mbn_memset(ctx.h, d_ws.ptr, 0, nb)
    print("#### tools/tail_bench.py: classifier tail, 7x7x1024 -> 1000 classes, top-5; microseconds of GPU time per tail (2000 back-to-back, one stream)")
    for n in (1, 2, 4):
        x = rng.uniform(0, 6, (n, h, h, ch)).astype(np.float32)
        d_x = ctx.to_device(x)
        d_lg, d_pool, d_p, d_i, d_v = ctx.alloc(n * classes * 4), ctx.alloc(n * ch * 4), ctx.alloc(n * classes * 4), ctx.alloc(n * 8 * 4), ctx.alloc(n * 8 * 4)
        forms = {
            "3 launches (pool, FC, softmax+top-k)": lambda: lib.mbn_classifier_tail(ctx.h, d_i.ptr, d_v.ptr, None, d_lg.ptr, d_pool.ptr, d_x.ptr, d_w.ptr, d_b.ptr, n, h, h, ch, classes, k, None),
            "2 launches (pool+FC, softmax+top-k)": lambda: (lib.mbn_pool_fc(ctx.h, d_lg.ptr, d_x.ptr, d_w.ptr, d_b.ptr, n, h, h, ch, classes, d_ws.ptr, nb, None),
                                                            lib.mbn_softmax_topk_f32(ctx.h, None, d_i.ptr, d_v.ptr, d_lg.ptr, n, classes, k, None)),
            "1 launch (mbn_classifier_tail_fused)": lambda: lib.mbn_classifier_tail_fused(ctx.h, d_i.ptr, d_v.ptr, None, d_lg.ptr, d_x.ptr, d_w.ptr, d_b.ptr, n, h, h, ch, classes, k, d_ws.ptr, nb, None),
        }
        res = {}
        for rep in range(3):
            for name, fn in forms.items():
                for _ in range(200):
                    fn()
                ctx.sync()
                t0 = time.perf_counter()
                for _ in range(2000):
                    fn()
                ctx.sync()
                res.setdefault(name, []).append((time.perf_counter() - t0) / 2000 * 1e6)
        for name, v in res.items():
            print("  %d image(s)  %-40s %6.2f us  (runs: %s)" % (n, name, float(np.median(v)), " ".join("%.2f" % t for t in v)))
