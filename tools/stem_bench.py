#!/usr/bin/env python3
"""Fused stem (layers 1-3, mbn_stem_fused) alone at batch 256 fp32: ms per launch from the library's event pool, variants interleaved in one process.
usage: stem_bench.py [--batch 256] [--reps 30] [--variants 0,6]     (lab conv_variant values: 0 shipped, 6 = conv1 rows unpadded, 5 = conv1 on the VALU)"""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("MBN_LAB", "1")
sys.path.insert(0, ROOT)
from mbn_amd import import_package  # noqa: E402
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--reps", type=int, default=30)
ap.add_argument("--variants", default="0,6")
args = ap.parse_args()
pkg = import_package(); lib = pkg.load(); ctx = pkg.Context(0)
n, res, c1, c3 = args.batch, 224, 32, 64
rng = np.random.default_rng(0)
img = rng.uniform(-1, 1, (n, res, res, 3)).astype(np.float32)
w1 = rng.normal(0, 0.3, (27 * c1,)).astype(np.float32)
wd = rng.normal(0, 0.5, (9 * c1,)).astype(np.float32)
wp = rng.normal(0, 0.25, (c3 * c1,)).astype(np.float32)
s1, s2, s3 = (rng.uniform(0.5, 1.5, c).astype(np.float32) for c in (c1, c1, c3))
b1, b2, b3 = (rng.normal(0, 0.1, c).astype(np.float32) for c in (c1, c1, c3))
d = [ctx.to_device(a) for a in (img, w1, s1, b1, wd, s2, b2, wp, s3, b3)]
h = res // 2
outs = {}
vs = [int(v) for v in args.variants.split(",")]
for v in vs:
    outs[v] = ctx.alloc(n * h * h * c3 * 4)
def run(v):
    lib.mbn_tune_set(b"conv_variant", v)
    rc = lib.mbn_stem_fused(ctx.h, outs[v].ptr, *[x.ptr for x in d], n, res, c1, c3, None)
    lib.mbn_tune_set(b"conv_variant", 0)
    assert rc == 0, rc
for _ in range(3):
    for v in vs: run(v)
ctx.sync()
ctx.profile_begin(len(vs) * args.reps)
for _ in range(args.reps):
    for v in vs: run(v)
ms = np.asarray(ctx.profile_end(len(vs) * args.reps)).reshape(args.reps, len(vs))
ref = outs[vs[0]].download((n * h * h, c3), np.float32)
for i, v in enumerate(vs):
    same = np.array_equal(ref, outs[v].download((n * h * h, c3), np.float32))
    print("conv_variant %d: median %.4f ms  min %.4f  (bytes %.1f MB -> %.2f TB/s)  %s" % (v, np.median(ms[:, i]), ms[:, i].min(), (img.nbytes + n * h * h * c3 * 4) / 1e6,
          (img.nbytes + n * h * h * c3 * 4) / np.median(ms[:, i]) / 1e9, "same bits as variant %d" % vs[0] if same else "DIFFERENT from variant %d" % vs[0]))
