#!/usr/bin/env python3
"""Core clock held inside the bf16 streaming GEMM (lab build): workgroup 0 stamps s_memtime (core clock cycles) and s_memrealtime (100 MHz) at its start and end.
usage: stream_clock.py  — layer 15 at batch 512 (100352 x 512 x 512), the kernel's ablation builds (exp1) one after the other."""
import ctypes as C
import os
import sys
import numpy as np
os.environ.setdefault("MBN_LAB", "1")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mbn_amd import import_package
pkg = import_package()
lib = pkg.load()
m, k, n = 100352, 512, 512
rng = np.random.default_rng(0)
with pkg.Context(0) as ctx:
    x = pkg.f32_to_bf16_bits(rng.uniform(-1, 1, (m, k)).astype(np.float32))
    f = pkg.f32_to_bf16_bits(rng.normal(0, 0.06, (n, k)).astype(np.float32))
    d_x, d_f = ctx.to_device(x), ctx.to_device(f)
    d_sc, d_sh = ctx.to_device(np.ones(n, np.float32)), ctx.to_device(np.zeros(n, np.float32))
    d_o = ctx.alloc(m * n * 2)
    ext = pkg.make_ext(dtype=pkg.DT_BF16, act=2, scale=d_sc.ptr, shift=d_sh.ptr)
    lib.mbn_debug_stream_clock.argtypes = [C.POINTER(C.c_ulonglong)]
    names = {0: "full kernel", 1: "no LDS-DMA", 2: "no fragment reads", 3: "no DMA, no fragment reads (MFMA + stores)", 4: "no MFMA",
             5: "no DMA, no MFMA", 6: "no fragment reads, no MFMA (DMA + stores)", 7: "barriers + stores only", 16: "no stores"}
    for ring, tag in ((4, "128 x 128 tiles, two workgroups per CU"), (7, "256 x 256 tiles, one workgroup per CU")):
        assert lib.mbn_tune_set(b"pw_ring", ring) == 0
        assert lib.mbn_tune_set(b"exp2", 98 if ring == 7 else 0) == 0
        print(tag)
        for e in (0, 1, 2, 3, 4, 5, 6, 7, 16):
            assert lib.mbn_tune_set(b"exp1", e) == 0
            for _ in range(20):
                ctx.pointwise(d_o.ptr, d_x.ptr, d_f.ptr, m, 1, k, n, ext)
            ctx.sync()
            v = (C.c_ulonglong * 4)()
            assert lib.mbn_debug_stream_clock(v) == 0
            dt = (v[3] - v[2]) / 100e6
            dc = v[1] - v[0]
            print("  exp1=%-2d %-44s workgroup 0 alive %.1f us, %7d core cycles -> %.2f GHz" % (e, names[e], dt * 1e6, dc, dc / dt / 1e9))
