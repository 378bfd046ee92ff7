#!/usr/bin/env python3
"""Where does the wave-private block (lab dwpw_variant = 11) differ from the shipped one (12)? Prints the mismatching 32-pixel tiles / channels.
usage: dwpw3_debug.py [--batch 64] [--block 6]"""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("MBN_LAB", "1")
sys.path.insert(0, ROOT)
from mbn_amd import import_package  # noqa: E402
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--block", type=int, default=6)
ap.add_argument("--tune", action="append", default=[])
args = ap.parse_args()
pkg = import_package(); lib = pkg.load(); ctx = pkg.Context(0)
plan = pkg.plan_build(1.0, 224, 1000, lib=lib)
ldw, lpw = plan.layer[args.block - 1], plan.layer[args.block]
n = args.batch
h, oh, cin, cout, s = ldw.in_rows, ldw.out_rows, ldw.in_ch, lpw.out_ch, ldw.stride
rng = np.random.default_rng(0)
x = rng.uniform(0, 6, (n, h, h, cin)).astype(np.float32)
wd = rng.normal(0, 0.5, (3, 3, cin)).astype(np.float32)
wp = rng.normal(0, (2.0 / cin) ** 0.5, (cout, cin)).astype(np.float32)
s2, s3 = rng.uniform(0.5, 1.5, cin).astype(np.float32), rng.uniform(0.5, 1.5, cout).astype(np.float32)
b2, b3 = rng.normal(0, 0.1, cin).astype(np.float32), rng.normal(0, 0.1, cout).astype(np.float32)
d = [ctx.to_device(a) for a in (x, wd, s2, b2, wp, s3, b3)]
outs = []
for v in (12, 11):
    o = ctx.alloc(n * oh * oh * cout * 4)
    lib.mbn_memset(ctx.h, o.ptr, 0xFF, n * oh * oh * cout * 4)
    lib.mbn_tune_set(b"dwpw_variant", v)
    for kv in args.tune:
        lib.mbn_tune_set(kv.split("=")[0].encode(), int(kv.split("=")[1]))
    rc = lib.mbn_dwpw_fused(ctx.h, o.ptr, d[0].ptr, d[1].ptr, d[2].ptr, d[3].ptr, d[4].ptr, d[5].ptr, d[6].ptr, n, h, h, oh, oh, cin, cout, s, ldw.pad_top, ldw.pad_left, None)
    assert rc == 0, rc
    ctx.sync()
    outs.append(o.download((n * oh * oh, cout), np.float32))
lib.mbn_tune_set(b"dwpw_variant", 0)
for kv in args.tune:
    lib.mbn_tune_set(kv.split("=")[0].encode(), 0)
a, b = outs
bad = (a.view(np.uint32) != b.view(np.uint32))
print("M", a.shape[0], "tiles", (a.shape[0] + 31) // 32, "mismatching elements", int(bad.sum()), "of", bad.size, "max abs diff", float(np.nanmax(np.abs(a - b))) if bad.any() else 0.0)
if bad.any():
    rows = np.flatnonzero(bad.any(axis=1))
    tiles = np.unique(rows // 32)
    print("bad rows", len(rows), "bad tiles", len(tiles), "first tiles", tiles[:40], "last", tiles[-10:])
    print("rows within tile histogram", np.bincount(rows % 32, minlength=32))
    cols = np.flatnonzero(bad.any(axis=0))
    print("bad cols", len(cols), cols[:32])
    r = rows[0]
    print("row", r, "shipped", a[r, cols[:6]], "dwpw3", b[r, cols[:6]])
    nanrows = np.flatnonzero(np.isnan(b).any(axis=1))
    print("rows never written by dwpw3 (NaN fill)", len(nanrows), nanrows[:20] // 32)
