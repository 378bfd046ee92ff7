#!/usr/bin/env python3
"""Core clock held inside pw_gemm (lab build): workgroups 0..7 (one per XCD) stamp s_memtime (core cycles) and s_memrealtime (100 MHz) at its start and end.
usage: gemm_clock.py — the network's stand-alone pointwise shapes at batch 256 (fp32) and 512 (bf16)."""
import ctypes as C
import os
import sys
import numpy as np
os.environ.setdefault("MBN_LAB", "1")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mbn_amd import import_package
pkg = import_package()
lib = pkg.load()
lib.mbn_debug_pw_clock.argtypes = [C.POINTER(C.c_ulonglong)]
rng = np.random.default_rng(0)
with pkg.Context(0) as ctx:
    for dtype, batch in (("f32", 256), ("bf16", 512)):
        for name, px, k, n in (("L13", 196, 256, 512), ("L15", 196, 512, 512), ("L25", 49, 512, 1024), ("L27", 49, 1024, 1024)):
            m = batch * px
            x = rng.uniform(-1, 1, (m, k)).astype(np.float32)
            f = rng.normal(0, (2.0 / k) ** 0.5, (n, k)).astype(np.float32)
            bf = dtype == "bf16"
            d_x = ctx.to_device(pkg.f32_to_bf16_bits(x) if bf else x)
            d_f = ctx.to_device(pkg.f32_to_bf16_bits(f) if bf else f)
            d_sc, d_sh = ctx.to_device(np.ones(n, np.float32)), ctx.to_device(np.zeros(n, np.float32))
            d_o = ctx.alloc(m * n * (2 if bf else 4))
            ext = pkg.make_ext(dtype=pkg.DT_BF16 if bf else pkg.DT_F32, act=2, scale=d_sc.ptr, shift=d_sh.ptr)
            if bf:
                assert lib.mbn_tune_set(b"pw_ring", 1) == 0          # pw_gemm<bf16>, not the streaming kernel
            for _ in range(30):
                ctx.pointwise(d_o.ptr, d_x.ptr, d_f.ptr, m, 1, k, n, ext)
            ctx.sync()
            v = (C.c_ulonglong * 32)()
            assert lib.mbn_debug_pw_clock(v) == 0
            ghz = [(v[4 * i + 1] - v[4 * i]) / ((v[4 * i + 3] - v[4 * i + 2]) / 100e6) / 1e9 for i in range(8)]
            span = (max(v[4 * i + 3] for i in range(8)) - min(v[4 * i + 2] for i in range(8))) / 100e6
            g = sum(ghz) / 8
            flops = 2.0 * m * k * n
            peak = (2500.0 if bf else 157.3) * g / 2.4
            print("%-5s %-4s M=%6d K=%4d N=%4d  first 8 workgroups (one per XCD) alive %6.1f us  core clock per XCD %s  mean %.2f GHz -> MFMA peak at that clock %.0f TFLOP/s"
                  % (dtype, name, m, k, n, span * 1e6, " ".join("%.2f" % x for x in ghz), g, peak))
            lib.mbn_tune_set(b"pw_ring", 0)
            for b in (d_x, d_f, d_sc, d_sh, d_o):
                b.free()
