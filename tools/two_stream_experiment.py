#!/usr/bin/env python3
"""Experiment: does running two half-batches on two HIP streams overlap the HBM-bound depthwise kernels of one with the
MFMA-bound pointwise kernels of the other?  Compares images/s of (a) one net, batch B; (b) two nets, batch B/2 each, on
two contexts (= two streams) of the same device, launched back to back from one host thread."""
import os
import sys
import time

import numpy as np
if os.environ.get("WITH_TORCH"):
    import torch  # noqa: F401  (loads torch's bundled HIP runtime first)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("MBN_LAB", "1")      # the lab build: every A/B variant and mbn_tune_set knob (make lab)
sys.path.insert(0, ROOT)
from mbn_amd import import_package  # noqa: E402

pkg = import_package()
lib = pkg.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
splits = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [B // 2, B // 2]
steps = 20
plan = pkg.plan_build(1.0, 224, 1000, lib=lib)
blob = np.random.default_rng(0).normal(0, 0.05, plan.blob_floats).astype(np.float32)
imgs = np.random.default_rng(1).uniform(-1, 1, (B, 224, 224, 3)).astype(np.float32)


def run(parts):
    ctxs = [pkg.Context(0) for _ in parts]
    nets, bufs = [], []
    off = 0
    for c, n in zip(ctxs, parts):
        nets.append(pkg.Net(c, plan, blob, n))
        bufs.append((c.to_device(imgs[off:off + n]), c.alloc(n * 4000)))
        off += n
    for _ in range(3):
        for net, (i, o), n in zip(nets, bufs, parts):
            net.forward(i.ptr, o.ptr, n)
    for c in ctxs:
        c.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        for net, (i, o), n in zip(nets, bufs, parts):
            net.forward(i.ptr, o.ptr, n)
    for c in ctxs:
        c.sync()
    dt = time.perf_counter() - t0
    for net in nets:
        net.destroy()
    for c in ctxs:
        c.close()
    return sum(parts) * steps / dt, 1000 * dt / steps


for parts in ([B], splits, [B // 4] * 4):
    ips, ms = run(parts)
    print("parts %-22s %9.0f images/s  %.3f ms per %d images" % (parts, ips, ms, sum(parts)))
