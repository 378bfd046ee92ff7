#!/usr/bin/env python3
"""Accuracy of the pointwise GEMM forms against a float64 product of the same fp32 operands, on the network's layer shapes
(reduced M): pw_gemm<float> (fp32 MFMA), the split forms of mbn_f32_pw_x6.hip (pw_emul = 9, 6; 3 and 1 for scale), and a
sequential fp32 fmaf chain on the CPU (the reference's own arithmetic, kernel.cl:94-114). Prints max and rms error in units
of the per-output scale sum_k |a_k b_k| * 2^-24 (one fp32 rounding of the magnitude sum).

  python tools/pw_emul_check.py            (needs the GPU)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("MBN_LAB", "1")      # the lab build: every A/B variant and mbn_tune_set knob (make lab)
sys.path.insert(0, ROOT)
from mbn_amd import import_package  # noqa: E402


def main():
    pkg = import_package()
    lib = pkg.load()
    ctx = pkg.Context(0)
    shapes = [(4096, 32, 64), (4096, 64, 128), (4096, 128, 128), (4096, 256, 256), (4099, 512, 512), (2048, 1024, 1024), (2000, 512, 1000), (777, 96, 200)]
    print("%-20s %-10s %12s %12s %12s" % ("shape M,K,N", "form", "max/ulpsum", "rms/ulpsum", "max rel"))
    for m, k, n in shapes:
        rng = np.random.default_rng(m + k + n)
        x = np.clip(rng.normal(1.0, 1.5, (m, k)), 0, 6).astype(np.float32)            # post-ReLU6 activations
        f = rng.normal(0, (2.0 / k) ** 0.5, (n, k)).astype(np.float32)
        ref = x.astype(np.float64) @ f.astype(np.float64).T
        mag = np.abs(x).astype(np.float64) @ np.abs(f).astype(np.float64).T
        unit = mag * 2.0 ** -24
        d_x, d_f = ctx.to_device(x), ctx.to_device(f)
        d_o = ctx.alloc(m * n * 4)
        ext = pkg.make_ext(batch=1, act=0)
        rows = {}
        for form in (0, 9, 6, 3, 1):
            assert lib.mbn_tune_set(b"pw_splitk", 1) == 0
            assert lib.mbn_tune_set(b"pw_emul", form) == 0
            ctx.pointwise(d_o.ptr, d_x.ptr, d_f.ptr, m, 1, k, n, ext)
            ctx.sync()
            rows["fp32 MFMA" if form == 0 else "split x%d" % form] = d_o.download((m, n), np.float32).astype(np.float64)
        lib.mbn_tune_set(b"pw_emul", 0)
        lib.mbn_tune_set(b"pw_splitk", 0)
        # sequential fp32 chain on a sample of outputs (numpy float32 cumulative fma-free: multiply-add rounding twice; close enough for scale)
        sm = min(m, 256)
        acc = np.zeros((sm, n), np.float32)
        for kk in range(k):
            acc = (acc + x[:sm, kk:kk + 1] * f[None, :, kk]).astype(np.float32)
        for name, got in list(rows.items()) + [("cpu chain", None)]:
            if got is None:
                e = np.abs(acc.astype(np.float64) - ref[:sm]) / unit[:sm]
                rel = np.abs(acc.astype(np.float64) - ref[:sm]).max() / np.abs(ref[:sm]).max()
            else:
                e = np.abs(got - ref) / unit
                rel = np.abs(got - ref).max() / np.abs(ref).max()
            print("%-20s %-10s %12.3f %12.4f %12.3e" % ("%d,%d,%d" % (m, k, n), name, e.max(), np.sqrt((e ** 2).mean()), rel))
    range_probe(pkg, lib, ctx)
    ctx.close() if hasattr(ctx, "close") else None


def range_probe(pkg, lib, ctx):
    """Operand range of the exact split: x * 1.0 through the pw_emul = 6 kernel for x with random 24-bit significands in one binade.
    bf16 has fp32's exponent range, but rounding up in the top binade overflows (h = inf) and the matrix cores flush bf16
    denormals, so the l plane (2^-17 x) loses bits below 2^-110."""
    m, k, n = 256, 64, 128
    rng = np.random.default_rng(1)
    lib.mbn_tune_set(b"pw_splitk", 1); lib.mbn_tune_set(b"pw_emul", 6); lib.mbn_tune_set(b"pw_tile", 11)
    ext = pkg.make_ext(batch=1, act=0)
    d_o = ctx.alloc(m * n * 4)
    print("\nbinade of the operand: exactness of x * 1.0 (pw_emul 6)")
    for e in (1, 60, 120, 126, 127, -60, -100, -110, -115, -120, -126):
        bits = rng.integers(0, 1 << 23, (m, k), dtype=np.uint32) | (np.uint32(127 + e) << 23)
        x = bits.view(np.float32)
        f = np.zeros((n, k), np.float32)
        kk = (np.arange(n) * 7) % k
        f[np.arange(n), kk] = 1.0
        d_x, d_f = ctx.to_device(x), ctx.to_device(f)
        ctx.pointwise(d_o.ptr, d_x.ptr, d_f.ptr, m, 1, k, n, ext)
        ctx.sync()
        got, want = d_o.download((m, n), np.float32), x[:, kk]
        bad = got.view(np.uint32) != want.view(np.uint32)
        with np.errstate(all="ignore"):
            rel = np.abs((got.astype(np.float64) - want.astype(np.float64)) / want.astype(np.float64))
        print("2^%4d: exact %-5s mismatches %5d / %d  max rel err %.3e  finite %s" % (e, not bad.any(), bad.sum(), bad.size, np.nanmax(rel), np.isfinite(got).all()))
        d_x.free(); d_f.free()
    for key in (b"pw_splitk", b"pw_emul", b"pw_tile"):
        lib.mbn_tune_set(key, 0)


if __name__ == "__main__":
    main()
