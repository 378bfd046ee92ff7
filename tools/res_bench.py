#!/usr/bin/env python3
"""The five 256 -> 256 blocks on the 10 x 10 map of the 0.5x160 network at batch 512 (bf16): one resident launch (mbn_blocks_resident_bf16) against five
mbn_dwpw_fused_bf16 launches, interleaved in one process; ms from the library's event pool.   usage: res_bench.py [--batch 512] [--reps 30] [--blocks 5] [--side 10]"""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mbn_amd import import_package  # noqa: E402
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=512)
ap.add_argument("--reps", type=int, default=30)
ap.add_argument("--blocks", type=int, default=5)
ap.add_argument("--side", type=int, default=10)
ap.add_argument("--tune", action="append", default=[])
args = ap.parse_args()
pkg = import_package(); lib = pkg.load(); ctx = pkg.Context(0)
n, h, c, nb = args.batch, args.side, 256, args.blocks
rng = np.random.default_rng(0)
x = rng.uniform(0, 4, (n, h, h, c)).astype(np.float32)
d_x = ctx.to_device(pkg.f32_to_bf16_bits(x))
dev = []
for _ in range(nb):
    wd = rng.normal(0, 0.5, (3, 3, c)).astype(np.float32)
    wp = rng.normal(0, (2.0 / c) ** 0.5, (c, c)).astype(np.float32)
    s2, s3 = rng.uniform(0.5, 1.5, c).astype(np.float32), rng.uniform(0.5, 1.5, c).astype(np.float32)
    b2, b3 = rng.normal(0, 0.1, c).astype(np.float32), rng.normal(0, 0.1, c).astype(np.float32)
    dev.append([ctx.to_device(wd), ctx.to_device(s2), ctx.to_device(b2), ctx.to_device(pkg.f32_to_bf16_bits(wp)), ctx.to_device(s3), ctx.to_device(b3)])
arr = (pkg.BlockParams * nb)()
for i, d in enumerate(dev):
    arr[i].wd, arr[i].s2, arr[i].b2, arr[i].wp_bf16, arr[i].s3, arr[i].b3 = (t.ptr for t in d)
d_o, d_p, d_q = ctx.alloc(x.size * 2), ctx.alloc(x.size * 2), ctx.alloc(x.size * 2)
def resident():
    for kv in args.tune:
        lib.mbn_tune_set(kv.split("=")[0].encode(), int(kv.split("=")[1]))
    assert lib.mbn_blocks_resident_bf16(ctx.h, d_o.ptr, d_x.ptr, arr, nb, n, h, h, c, None) == 0
    for kv in args.tune:
        lib.mbn_tune_set(kv.split("=")[0].encode(), 0)
def one_by_one():
    src, dst = d_x, d_p
    for d in dev:
        assert lib.mbn_dwpw_fused_bf16(ctx.h, dst.ptr, src.ptr, d[0].ptr, d[1].ptr, d[2].ptr, d[3].ptr, d[4].ptr, d[5].ptr, n, h, h, h, h, c, c, 1, 1, 1, None) == 0
        src, dst = dst, (d_q if dst is d_p else d_p)
    return src
for _ in range(3):
    resident(); one_by_one()
ctx.sync()
ctx.profile_begin((1 + nb) * args.reps)
for _ in range(args.reps):
    resident(); last = one_by_one()
ms = np.asarray(ctx.profile_end((1 + nb) * args.reps)).reshape(args.reps, 1 + nb)
a = pkg.bf16_bits_to_f32(d_o.download((n, h, h, c), np.uint16)); b = pkg.bf16_bits_to_f32(last.download((n, h, h, c), np.uint16))
print("batch %d, %d blocks on %dx%dx%d bf16: resident launch %.4f ms (min %.4f); %d fused launches %.4f ms (sum of medians; each %s); max rel diff %.2e"
      % (n, nb, h, h, c, np.median(ms[:, 0]), ms[:, 0].min(), nb, np.median(ms[:, 1:], axis=0).sum(), np.round(np.median(ms[:, 1:], axis=0), 4).tolist(),
         float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-6))))
