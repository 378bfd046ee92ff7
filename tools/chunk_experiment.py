#!/usr/bin/env python3
"""Experiment: depth-first chunking of the HBM-bound early layers. Layers 1..L of a 256-image batch are run (a) once over
the whole batch and (b) chunk by chunk (c images at a time, all L layers per chunk) on one stream, so that a layer's
output (c x up to 3.2 MB) is still in the 256 MiB Infinity Cache when the next layer reads it.
usage: chunk_experiment.py [batch] [last_layer,...] [chunk,...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("MBN_LAB", "1")      # the lab build: every A/B variant and mbn_tune_set knob (make lab)
sys.path.insert(0, ROOT)
from mbn_amd import import_package  # noqa: E402

pkg = import_package()
lib = pkg.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
lasts = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [5, 7, 11, 13]
chunks = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [256, 128, 64, 32, 16]
steps = 20
plan = pkg.plan_build(1.0, 224, 1000, lib=lib)
blob = np.random.default_rng(0).normal(0, 0.05, plan.blob_floats).astype(np.float32)
imgs = np.random.default_rng(1).uniform(-1, 1, (B, 224, 224, 3)).astype(np.float32)
ctx = pkg.Context(0)
net = pkg.Net(ctx, plan, blob, B)
d_in = ctx.to_device(imgs)
img_bytes = 224 * 224 * 3 * 4
for last in lasts:
    l = plan.layer[last - 1]
    per_img = l.out_rows * l.out_cols * l.out_ch * 4
    d_out = ctx.alloc(B * per_img)
    ref = None
    for c in chunks:
        def once():
            for first in range(0, B, c):
                n = min(c, B - first)
                net.forward(d_in.ptr + first * img_bytes, d_out.ptr + first * per_img, n, last)
        for _ in range(3):
            once()
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            once()
        ctx.sync()
        ms = 1000 * (time.perf_counter() - t0) / steps
        got = d_out.download((B, per_img // 4), np.float32)
        if ref is None:
            ref = got
        print("layers 1..%-2d chunk %3d: %.3f ms  %s" % (last, c, ms, "same" if np.array_equal(ref, got) else "DIFFERENT"))
        sys.stdout.flush()
    d_out.free()
