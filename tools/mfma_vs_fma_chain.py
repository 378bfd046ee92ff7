#!/usr/bin/env python3
"""Is v_mfma_f32_32x32x2_f32's K-chain bit-identical to a sequential fmaf chain over k? pw_gemm<float> (MFMA) against pw_generic (one lane per
output element, fmaf over k in order; taken when the input pointer is not 16-byte aligned) on the same operands, bit for bit. Decides whether a
conv1-as-GEMM form of the stem (K = 27 taps in tap order) can stay bit-identical to the VALU conv1 kernel."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mbn_amd import import_package
pkg = import_package()
lib = pkg.load()
rng = np.random.default_rng(5)
with pkg.Context(0) as ctx:
    for m, k, n in ((4096, 32, 64), (4096, 28, 32), (8192, 128, 128), (2048, 512, 512)):
        x = rng.uniform(-1, 1, (m, k)).astype(np.float32)
        f = rng.normal(0, 0.3, (n, k)).astype(np.float32)
        sc, sh = rng.uniform(0.5, 1.5, n).astype(np.float32), rng.normal(0, 0.1, n).astype(np.float32)
        d_f, d_sc, d_sh = ctx.to_device(f), ctx.to_device(sc), ctx.to_device(sh)
        d_x = ctx.alloc(x.nbytes + 16)
        d_o1, d_o2 = ctx.alloc(m * n * 4), ctx.alloc(m * n * 4)
        ext = pkg.make_ext(dtype=pkg.DT_F32, act=2, scale=d_sc.ptr, shift=d_sh.ptr)
        lib.mbn_upload(ctx.h, d_x.ptr, x.ctypes.data, x.nbytes)
        assert lib.mbn_tune_set(b"pw_splitk", 1) == 0          # never the split-K kernel
        ctx.pointwise(d_o1.ptr, d_x.ptr, d_f.ptr, m, 1, k, n, ext)          # aligned: pw_gemm (MFMA)
        lib.mbn_upload(ctx.h, d_x.ptr + 4, x.ctypes.data, x.nbytes)
        ctx.pointwise(d_o2.ptr, d_x.ptr + 4, d_f.ptr, m, 1, k, n, ext)      # 4-byte aligned input: pw_generic (fmaf chain)
        ctx.sync()
        a, b = d_o1.download((m, n), np.float32), d_o2.download((m, n), np.float32)
        ref = np.clip((x.astype(np.float64) @ f.astype(np.float64).T) * sc + sh, 0, 6)
        print("M=%d K=%d N=%d: MFMA vs fmaf chain: %s (differing elements %d of %d, max |diff| %.3g); max err vs float64: MFMA %.3g, chain %.3g"
              % (m, k, n, "BIT-IDENTICAL" if np.array_equal(a, b) else "different", int((a != b).sum()), a.size, float(np.abs(a - b).max()),
                 float(np.abs(a - ref).max()), float(np.abs(b - ref).max())))
        lib.mbn_tune_set(b"pw_splitk", 0)
