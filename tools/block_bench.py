#!/usr/bin/env python3
"""Microbenchmark of the fused depthwise->pointwise block (mbn_dwpw_fused) against the two separate launches, per block
of MobileNet-V1 1.0x224 at the given batch. Per-kernel times come from the library's own HIP events (mbn_profile_*).
usage: block_bench.py [--batch 256] [--reps 20] [--tune key=value ...] [--blocks 4,6,8,10,12]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("MBN_LAB", "1")      # the lab build: every A/B variant and mbn_tune_set knob (make lab)
sys.path.insert(0, ROOT)
from mbn_amd import import_package  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--tune", action="append", default=[])
ap.add_argument("--blocks", default="4,6,8,10,12")
args = ap.parse_args()
pkg = import_package()
lib = pkg.load()
ctx = pkg.Context(0)
plan = pkg.plan_build(1.0, 224, 1000, lib=lib)
rng = np.random.default_rng(0)
n = args.batch
print("%-28s %9s %9s %9s %9s %8s %8s" % ("block", "dw ms", "pw ms", "sum ms", "fused ms", "speedup", "TFLOP/s"))
for first in [int(b) for b in args.blocks.split(",")]:
    ldw, lpw = plan.layer[first - 1], plan.layer[first]
    assert ldw.kind == pkg.L_DW and lpw.kind == pkg.L_PW
    h, oh, cin, cout, s = ldw.in_rows, ldw.out_rows, ldw.in_ch, lpw.out_ch, ldw.stride
    x = rng.uniform(0, 6, (n, h, h, cin)).astype(np.float32)
    wd = rng.normal(0, 0.5, (3, 3, cin)).astype(np.float32)
    wp = rng.normal(0, (2.0 / cin) ** 0.5, (cout, cin)).astype(np.float32)
    s2, s3 = rng.uniform(0.5, 1.5, cin).astype(np.float32), rng.uniform(0.5, 1.5, cout).astype(np.float32)
    b2, b3 = rng.normal(0, 0.1, cin).astype(np.float32), rng.normal(0, 0.1, cout).astype(np.float32)
    d = [ctx.to_device(a) for a in (x, wd, s2, b2, wp, s3, b3)]
    del x
    d_f, d_u = ctx.alloc(n * oh * oh * cout * 4), ctx.alloc(n * oh * oh * cout * 4)
    d_m = ctx.alloc(n * oh * oh * cin * 4)
    e_dw = pkg.make_ext(batch=n, act=2, pad_top=ldw.pad_top, pad_left=ldw.pad_left, in_rows=h, in_cols=h, scale=d[2].ptr, shift=d[3].ptr)
    e_pw = pkg.make_ext(batch=1, act=2, scale=d[5].ptr, shift=d[6].ptr)

    emul = [kv for kv in args.tune if kv.startswith("pw_emul=")]      # an arithmetic form, not a kernel variant: applies to both sides

    def unfused():
        for kv in emul:
            lib.mbn_tune_set(b"pw_emul", int(kv.split("=")[1]))
        ctx.depthwise(d_m.ptr, d[0].ptr, d[1].ptr, oh, oh, 3, s, cin, e_dw)
        ctx.pointwise(d_u.ptr, d_m.ptr, d[4].ptr, n * oh * oh, 1, cin, cout, e_pw)
        for kv in emul:
            lib.mbn_tune_set(b"pw_emul", 0)

    def fused():
        for kv in args.tune:
            k, v = kv.split("=")
            lib.mbn_tune_set(k.encode(), int(v))
        rc = lib.mbn_dwpw_fused(ctx.h, d_f.ptr, d[0].ptr, d[1].ptr, d[2].ptr, d[3].ptr, d[4].ptr, d[5].ptr, d[6].ptr,
                                n, h, h, oh, oh, cin, cout, s, ldw.pad_top, ldw.pad_left, None)
        for kv in args.tune:
            lib.mbn_tune_set(kv.split("=")[0].encode(), 0)
        assert rc == 0, rc

    for _ in range(3):
        unfused(); fused()
    ctx.sync()
    ctx.profile_begin(3 * args.reps)
    for _ in range(args.reps):          # interleaved A/B: both see the same clocks and cache state
        unfused(); fused()
    ms = np.asarray(ctx.profile_end(3 * args.reps)).reshape(args.reps, 3)
    t_dw, t_pw, t_f = np.median(ms, axis=0)
    same = np.array_equal(d_f.download((n * oh * oh, cout), np.float32)[:4096], d_u.download((n * oh * oh, cout), np.float32)[:4096])
    flops = 2.0 * n * oh * oh * cin * (cout + 9)
    print("L%d-%d %3dx%-3d %4d->%-4d s%d    %9.4f %9.4f %9.4f %9.4f %7.2fx %8.1f %s" % (
        first, first + 1, h, h, cin, cout, s, t_dw, t_pw, t_dw + t_pw, t_f, (t_dw + t_pw) / t_f, flops / t_f / 1e9,
        "" if same else "MISMATCH"))
    sys.stdout.flush()
    for b in d + [d_f, d_u, d_m]:
        b.free()
