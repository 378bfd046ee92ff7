#!/usr/bin/env python3
"""Layers 24-28 of the 0.5x160 network at batch 512 (bf16): one resident launch (mbn_tail_resident_bf16) against the five launches (depthwise, pointwise,
depthwise, pointwise, pool), interleaved in one process; ms from the library's event pool.   usage: tail_bench.py [--batch 512] [--reps 30] [--side 10]"""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mbn_amd import import_package  # noqa: E402
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=512)
ap.add_argument("--reps", type=int, default=30)
ap.add_argument("--side", type=int, default=10)
ap.add_argument("--tune", action="append", default=[])
args = ap.parse_args()
pkg = import_package(); lib = pkg.load(); ctx = pkg.Context(0)
n, h, c0, c1 = args.batch, args.side, 256, 512
h1 = h // 2
rng = np.random.default_rng(0)
x = rng.uniform(0, 4, (n, h, h, c0)).astype(np.float32)
d_x = ctx.to_device(pkg.f32_to_bf16_bits(x))
dev = []
for ci in (c0, c1):
    wd = rng.normal(0, 0.5, (3, 3, ci)).astype(np.float32)
    wp = rng.normal(0, (2.0 / ci) ** 0.5, (c1, ci)).astype(np.float32)
    s2, s3 = rng.uniform(0.5, 1.5, ci).astype(np.float32), rng.uniform(0.5, 1.5, c1).astype(np.float32)
    b2, b3 = rng.normal(0, 0.1, ci).astype(np.float32), rng.normal(0, 0.1, c1).astype(np.float32)
    dev.append([ctx.to_device(wd), ctx.to_device(s2), ctx.to_device(b2), ctx.to_device(pkg.f32_to_bf16_bits(wp)), ctx.to_device(s3), ctx.to_device(b3)])
arr = (pkg.BlockParams * 2)()
for i, d in enumerate(dev):
    arr[i].wd, arr[i].s2, arr[i].b2, arr[i].wp_bf16, arr[i].s3, arr[i].b3 = (t.ptr for t in d)
d_o, d_a, d_b = ctx.alloc(n * c1 * 2), ctx.alloc(n * h * h * c1 * 2), ctx.alloc(n * h * h * c1 * 2)
e = lambda d, **kw: pkg.make_ext(batch=n, dtype=pkg.DT_BF16, act=2, scale=d[0].ptr, shift=d[1].ptr, **kw)
def resident():
    for kv in args.tune:
        lib.mbn_tune_set(kv.split("=")[0].encode(), int(kv.split("=")[1]))
    rc = lib.mbn_tail_resident_bf16(ctx.h, d_o.ptr, d_x.ptr, arr, n, h, h, c0, c1, None)
    for kv in args.tune:
        lib.mbn_tune_set(kv.split("=")[0].encode(), 0)
    assert rc == 0
def one_by_one():
    ctx.depthwise(d_a.ptr, d_x.ptr, dev[0][0].ptr, h1, h1, 3, 2, c0, e(dev[0][1:3], pad_top=0, pad_left=0, in_rows=h, in_cols=h))
    ctx.pointwise(d_b.ptr, d_a.ptr, dev[0][3].ptr, h1, h1, c0, c1, e(dev[0][4:6]))
    ctx.depthwise(d_a.ptr, d_b.ptr, dev[1][0].ptr, h1, h1, 3, 1, c1, e(dev[1][1:3], pad_top=1, pad_left=1, in_rows=h1, in_cols=h1))
    ctx.pointwise(d_b.ptr, d_a.ptr, dev[1][3].ptr, h1, h1, c1, c1, e(dev[1][4:6]))
    ctx.pool(d_a.ptr, d_b.ptr, h1, h1, h1, c1, pkg.make_ext(batch=n, dtype=pkg.DT_BF16, act=0))
for _ in range(3):
    resident(); one_by_one()
ctx.sync()
ctx.profile_begin(6 * args.reps)
for _ in range(args.reps):
    resident(); one_by_one()
ms = np.asarray(ctx.profile_end(6 * args.reps)).reshape(args.reps, 6)
a = pkg.bf16_bits_to_f32(d_o.download((n, c1), np.uint16)); b = pkg.bf16_bits_to_f32(d_a.download((n, c1), np.uint16))
print("batch %d, %dx%dx%d -> %d pooled, bf16: resident launch %.4f ms (min %.4f); five launches %.4f ms (sum of medians; each %s); max rel diff %.2e"
      % (n, h, h, c0, c1, np.median(ms[:, 0]), ms[:, 0].min(), np.median(ms[:, 1:], axis=0).sum(), np.round(np.median(ms[:, 1:], axis=0), 4).tolist(),
         float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-6))))
