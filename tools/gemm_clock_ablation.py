#!/usr/bin/env python3
"""Core clock held inside pw_gemm<float> and its ablation builds (lab): workgroups 0..7 (one per XCD) stamp s_memtime (core cycles) and
s_memrealtime (100 MHz) at start and end. M = 49152, K = N = 512 (whole rounds of every tile form), random operands, >= 2 s of back-to-back
launches before the stamped one (MI355X guide, 'DVFS give-back' item 6). Answers VERDICT r3 item 2: with the DMA, the barriers, the epilogue
stores AND the LDS fragment reads removed the kernel still runs at 0.86-0.88 of the 157.3 TFLOP/s peak — what clock does it hold?"""
import ctypes as C
import os
import sys
import time
import numpy as np
os.environ.setdefault("MBN_LAB", "1")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mbn_amd import import_package
pkg = import_package()
lib = pkg.load()
lib.mbn_debug_pw_clock.argtypes = [C.POINTER(C.c_ulonglong)]
rng = np.random.default_rng(0)
m, k, n = 49152, 512, 512
names = {0: "full kernel", 1: "no LDS-DMA in the k-loop", 2: "no barriers in the k-loop", 4: "no epilogue stores", 8: "no LDS fragment reads",
         9: "no DMA, no fragment reads", 15: "MFMAs only (no DMA, barriers, stores, fragment reads)"}
with pkg.Context(0) as ctx:
    x = rng.uniform(-1, 1, (m, k)).astype(np.float32)
    f = rng.normal(0, (2.0 / k) ** 0.5, (n, k)).astype(np.float32)
    d_x, d_f = ctx.to_device(x), ctx.to_device(f)
    d_sc, d_sh = ctx.to_device(np.ones(n, np.float32)), ctx.to_device(np.zeros(n, np.float32))
    d_o = ctx.alloc(m * n * 4)
    ext = pkg.make_ext(dtype=pkg.DT_F32, act=2, scale=d_sc.ptr, shift=d_sh.ptr)
    print("#### tools/gemm_clock_ablation.py: pw_gemm<float>, M=%d K=%d N=%d; tile 3 = 64x64 (4 waves of 32x32, 4 workgroups per CU, shipped), 5 = 128x128 (8 waves of 32x64, 2 per CU)" % (m, k, n))
    for tile in (3, 5):
        for abl in (0, 1, 4, 8, 9, 15):
            assert lib.mbn_tune_set(b"pw_tile", tile) == 0 and lib.mbn_tune_set(b"exp1", abl) == 0
            t0 = time.time()
            it = 0
            while time.time() - t0 < 2.0:                  # >= 2 s of back-to-back launches: the clock has settled
                for _ in range(200):
                    ctx.pointwise(d_o.ptr, d_x.ptr, d_f.ptr, m, 1, k, n, ext)
                ctx.sync()
                it += 200
            ts = []
            for _ in range(50):
                ctx.profile_begin(1)
                ctx.pointwise(d_o.ptr, d_x.ptr, d_f.ptr, m, 1, k, n, ext)
                ts.append(ctx.profile_end(1)[0])
            ctx.pointwise(d_o.ptr, d_x.ptr, d_f.ptr, m, 1, k, n, ext)
            ctx.sync()
            v = (C.c_ulonglong * 32)()
            assert lib.mbn_debug_pw_clock(v) == 0
            ghz = [(v[4 * i + 1] - v[4 * i]) / ((v[4 * i + 3] - v[4 * i + 2]) / 100e6) / 1e9 for i in range(8)]
            g = float(np.median(ghz))
            ms = float(np.median(ts)) - 0.0047             # event pair overhead (bench.py event_overhead_us)
            tf = 2.0 * m * k * n / ms / 1e9
            print("tile %d  exp1=%-2d %-56s %.4f ms  %6.1f TFLOP/s = %.3f of 157.3   in-kernel clock %.2f GHz (XCDs %.2f-%.2f) -> peak at that clock %.1f TFLOP/s, kernel at %.3f of it"
                  % (tile, abl, names[abl], ms, tf, tf / 157.3, g, min(ghz), max(ghz), 157.3 * g / 2.4, tf / (157.3 * g / 2.4)))
            sys.stdout.flush()
    lib.mbn_tune_set(b"pw_tile", 0); lib.mbn_tune_set(b"exp1", 0)
