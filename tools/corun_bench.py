#!/usr/bin/env python3
"""Do two layers on two streams run concurrently? Layer A (a pointwise GEMM) REPS times on one stream, layer B (a depthwise layer) REPS_B times on
another, alone and together: wall time (host clock around a sync) of each alone, of both queued together, and the sum. `together` near max(alone)
means the two kernels share the chip; near the sum means the second waits for the first's workgroups to leave.

  MBN_LAB=1 python tools/corun_bench.py --a 15 --b 14 --batch 128 [--tune misc=3]
"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("MBN_LAB", "1")
sys.path.insert(0, ROOT)
from mbn_amd import import_package  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--a", type=int, default=15)
    ap.add_argument("--b", type=int, default=14)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--reps", type=int, default=40)
    ap.add_argument("--tune", action="append", default=[])
    args = ap.parse_args()
    pkg = import_package()
    lib = pkg.load()
    lib.mbn_tune_set.argtypes = [C.c_char_p, C.c_int]
    plan = pkg.plan_build(1.0, 224, 1000, lib=lib)
    ctx = pkg.Context(0)
    rng = np.random.default_rng(0)
    d_blob = ctx.to_device(rng.normal(0, 0.05, plan.blob_floats).astype(np.float32))
    n = args.batch
    bufs = [ctx.alloc(int(plan.max_act_floats) * n * 4) for _ in range(4)]
    chunk = rng.uniform(-1, 1, 1 << 22).astype(np.float32)
    for d in (bufs[0], bufs[2]):
        for off in range(0, int(plan.max_act_floats) * n, chunk.size):
            m = min(chunk.size, int(plan.max_act_floats) * n - off)
            lib.mbn_upload(ctx.h, d.ptr + off * 4, chunk.ctypes.data, m * 4)
    streams = []
    for _ in range(2):
        s = C.c_void_p()
        assert lib.mbn_stream_create(ctx.h, C.byref(s)) == 0
        streams.append(s)

    def call(li, src, dst, stream):
        l = plan.layer[li - 1]
        ext = pkg.make_ext(batch=n, act=pkg.ACT_RELU6, pad_top=l.pad_top, pad_left=l.pad_left, stream=stream,
                           scale=(d_blob.ptr + 4 * l.scale_offset) if l.scale_offset >= 0 else None,
                           shift=(d_blob.ptr + 4 * l.shift_offset) if l.shift_offset >= 0 else None)
        filt = d_blob.ptr + 4 * l.w_offset
        if l.kind == pkg.L_DW:
            ext.in_rows, ext.in_cols = l.in_rows, l.in_cols
            ctx.depthwise(dst.ptr, src.ptr, filt, l.out_rows, l.out_cols, 3, l.stride, l.out_ch, ext)
        else:
            ctx.pointwise(dst.ptr, src.ptr, filt, l.out_rows, l.out_cols, l.in_ch, l.out_ch, ext)

    def sync_all():
        for s in streams:
            assert lib.mbn_stream_wait(ctx.h, None, s) == 0
        ctx.sync()

    def timed(fn):
        fn(); sync_all()
        best = []
        for _ in range(5):
            t0 = time.perf_counter(); fn(); sync_all(); best.append(time.perf_counter() - t0)
        return 1e3 * float(np.median(best))

    for t in [""] + args.tune:
        if t:
            k, v = t.split("=")
            assert lib.mbn_tune_set(k.encode(), int(v)) == 0
        ra = args.reps
        ta = timed(lambda: [call(args.a, bufs[0], bufs[1], streams[0]) for _ in range(ra)])
        tb1 = timed(lambda: [call(args.b, bufs[2], bufs[3], streams[1]) for _ in range(ra)])
        rb = max(1, int(round(ra * ta / tb1)))                      # as much depthwise work as the GEMM stream has
        tb = timed(lambda: [call(args.b, bufs[2], bufs[3], streams[1]) for _ in range(rb)])

        def both():
            ia = ib = 0
            while ia < ra or ib < rb:                               # interleaved submission
                if ia < ra:
                    call(args.a, bufs[0], bufs[1], streams[0]); ia += 1
                for _ in range(max(1, rb // ra)):
                    if ib < rb:
                        call(args.b, bufs[2], bufs[3], streams[1]); ib += 1
        tt = timed(both)
        print("[%s] layer %d x%d alone %.3f ms; layer %d x%d alone %.3f ms; together %.3f ms (sum %.3f, max %.3f): overlap %.0f %% of the shorter" % (
            t or "default", args.a, ra, ta, args.b, rb, tb, tt, ta + tb, max(ta, tb), 100 * (ta + tb - tt) / min(ta, tb)))
    ctx.close()


if __name__ == "__main__":
    main()
