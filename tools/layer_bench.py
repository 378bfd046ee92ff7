#!/usr/bin/env python3
"""Per-layer microbenchmark of the hot path on one MI355X: runs chosen layers of the plan in isolation through the
C-ABI (same calls the net runner makes), times them with the event pool, prints ms / GB/s / TFLOP/s per layer.

  python tools/layer_bench.py --layers 2,3,15 --iters 30 [--batch 256] [--tune key=value ...]

Used for A/B of kernel variants (interleaved in one process, cdna guide §5.4 rule 24) and as the target of
rocprofv3 --pmc passes (one kernel shape per run keeps the counter CSV readable).
"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("MBN_LAB", "1")      # the lab build: every A/B variant and mbn_tune_set knob (make lab)
sys.path.insert(0, ROOT)
from bench import layer_work, HBM_PEAK_GBS, MFMA_F32_PEAK_TFLOPS  # noqa: E402
from mbn_amd import import_package  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", default="all")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--alpha", type=float, default=1.0)
    ap.add_argument("--res", type=int, default=224)
    ap.add_argument("--tune", action="append", default=[], help="key=v1,v2,... : A/B over tuning values, interleaved")
    ap.add_argument("--json", action="store_true")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32")
    ap.add_argument("--packed", action="store_true", help="bf16: give every pointwise layer a filter buffer with the packed image behind it (IO_FILT_PACKED)")
    ap.add_argument("--custom-pw", default="", help="M,K,N[;M,K,N...]: time raw pointwise GEMMs of these shapes instead")
    args = ap.parse_args()

    pkg = import_package()
    lib = pkg.load()
    plan = pkg.plan_build(args.alpha, args.res, 1000, lib=lib)
    idxs = list(range(1, plan.n_layers + 1)) if args.layers == "all" else [int(x) for x in args.layers.split(",")]
    variants = [{}]
    for t in args.tune:
        k, vs = t.split("=")
        variants = [dict(v, **{k: int(x)}) for v in variants for x in vs.split(",")]

    ctx = pkg.Context(0)
    rng = np.random.default_rng(0)
    blob = rng.normal(0, 0.05, plan.blob_floats).astype(np.float32)
    d_blob = ctx.to_device(blob)
    n = args.batch
    d_a = ctx.alloc(int(plan.max_act_floats) * n * 4)
    d_b = ctx.alloc(int(plan.max_act_floats) * n * 4)
    # fill the input buffer with random data once (random operands: guide rule 25)
    chunk = rng.uniform(-1, 1, 1 << 22).astype(np.float32)
    tot = int(plan.max_act_floats) * n
    for off in range(0, tot, chunk.size):
        m = min(chunk.size, tot - off)
        lib.mbn_upload(ctx.h, d_a.ptr + off * 4, chunk.ctypes.data, m * 4)

    bf = args.dtype == "bf16"
    ab = 2.0 if bf else 4.0

    packed = {}

    def call(l):
        ext = pkg.make_ext(batch=n, dtype=pkg.DT_BF16 if bf else pkg.DT_F32, act=pkg.ACT_RELU6, pad_top=l.pad_top,
                           pad_left=l.pad_left,
                           io_flags=(pkg.IO_IN_F32 if l.kind == pkg.L_CONV else pkg.IO_OUT_F32 if l.kind == pkg.L_FC else 0) if bf else 0,
                           scale=(d_blob.ptr + 4 * l.scale_offset) if l.scale_offset >= 0 else None,
                           shift=(d_blob.ptr + 4 * l.shift_offset) if l.shift_offset >= 0 else None)
        filt = d_blob.ptr + 4 * l.w_offset
        if l.kind == pkg.L_CONV:
            ext.cin = l.in_ch
            ctx.convolute(d_b.ptr, d_a.ptr, None, None, filt, l.in_rows, l.in_cols, 3, l.stride, l.out_ch, ext)
        elif l.kind == pkg.L_DW:
            ext.in_rows, ext.in_cols = l.in_rows, l.in_cols
            ctx.depthwise(d_b.ptr, d_a.ptr, filt, l.out_rows, l.out_cols, 3, l.stride, l.out_ch, ext)
        elif l.kind == pkg.L_PW:
            if bf and args.packed:
                if l.index not in packed:
                    w = blob[l.w_offset:l.w_offset + l.w_count].reshape(l.out_ch, l.in_ch)
                    packed[l.index] = pkg.packed_filter_dev(ctx, w)
                filt, fl = packed[l.index][0].ptr, packed[l.index][1]
                ext.io_flags |= fl
            ctx.pointwise(d_b.ptr, d_a.ptr, filt, l.out_rows, l.out_cols, l.in_ch, l.out_ch, ext)
        elif l.kind == pkg.L_POOL:
            ctx.pool(d_b.ptr, d_a.ptr, l.in_rows, l.in_cols, l.in_rows, l.out_ch, ext)
        else:
            ext.act = pkg.ACT_NONE
            ctx.pointwise(d_b.ptr, d_a.ptr, filt, 1, 1, l.in_ch, l.out_ch, ext)

    def set_tune(v):
        for k, x in v.items():
            rc = lib.mbn_tune_set(k.encode(), x)
            if rc != 0:
                sys.exit("mbn_tune_set(%s) -> %d" % (k, rc))

    lib.mbn_tune_set.argtypes = [C.c_char_p, C.c_int]
    rows = []
    if args.custom_pw:
        for spec in args.custom_pw.split(";"):
            m, k, nn = (int(x) for x in spec.split(","))
            assert m * max(k, nn) <= tot and nn * k <= plan.blob_floats
            ext = pkg.make_ext(batch=1, act=pkg.ACT_RELU6)
            for vi, v in enumerate(variants):
                set_tune(v)
                ts = []
                for it in range(args.iters + args.warmup):
                    ctx.profile_begin(1)
                    ctx.pointwise(d_b.ptr, d_a.ptr, d_blob.ptr, m, 1, k, nn, ext)
                    t = ctx.profile_end(1)[0]
                    if it >= args.warmup:
                        ts.append(t)
                med = float(np.median(ts))
                print("pw M=%d K=%d N=%d %-18s med %.4f ms  %6.1f TF (%.0f%% MFMA)  %6.0f GB/s" % (
                    m, k, nn, json.dumps(v), med, 2.0 * m * k * nn / med / 1e9,
                    100 * 2.0 * m * k * nn / med / 1e9 / MFMA_F32_PEAK_TFLOPS, 4.0 * (m * k + m * nn) / med / 1e6))
        ctx.close()
        return
    for li in idxs:
        l = plan.layer[li - 1]
        for v in variants:
            set_tune(v)
            for _ in range(args.warmup):
                call(l)
        ctx.sync()
        times = {i: [] for i in range(len(variants))}
        for _ in range(args.iters):                 # interleave variants round by round
            for vi, v in enumerate(variants):
                set_tune(v)
                ctx.profile_begin(1)
                call(l)
                times[vi].append(ctx.profile_end(1)[0])
        f, b = layer_work(l, n, pkg, ab)
        for vi, v in enumerate(variants):
            med = float(np.median(times[vi]))
            mn = float(np.min(times[vi]))
            rows.append({"layer": li, "kind": int(l.kind), "variant": v, "ms_med": med, "ms_min": mn,
                         "GBps": b / med / 1e6, "TFLOPs": f / med / 1e9})
    if args.json:
        print(json.dumps(rows))
    else:
        for r in rows:
            print("L%-2d kind=%d %-24s med %.4f ms  min %.4f ms  %7.0f GB/s (%.0f%% HBM)  %6.1f TF (%.0f%% MFMA)" % (
                r["layer"], r["kind"], json.dumps(r["variant"]), r["ms_med"], r["ms_min"], r["GBps"],
                100 * r["GBps"] / HBM_PEAK_GBS, r["TFLOPs"], 100 * r["TFLOPs"] / MFMA_F32_PEAK_TFLOPS))
    ctx.close()


if __name__ == "__main__":
    main()
