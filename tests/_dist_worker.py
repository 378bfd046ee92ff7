"""Worker for tests/test_dist_cpu.py: one of WORLD_SIZE gloo ranks rehearsing bench.py's multi-GPU protocol on the CPU.
The per-rank compute stand-in is the oracle (allowed: this is a test), the protocol code is the product's dist.py."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from mbn_amd import import_package  # noqa: E402
import oracle as orc  # noqa: E402

pkg = import_package()
from mbn_amd_pkg import dist as mdist  # noqa: E402

out_dir, total = sys.argv[1], int(sys.argv[2])
rank, world = mdist.init("gloo")
alpha, res, classes = 0.25, 32, 16
plan = pkg.plan_build(alpha, res, classes)
blob = torch.full((plan.blob_floats,), float("nan"))
if rank == 0:                                   # only rank 0 touches the weight file
    path = os.path.join(out_dir, "w.h5")
    pkg.synthetic_h5(path, alpha=alpha, classes=classes, seed=5)
    hw = pkg.HostWeights(path, res=res)
    blob.copy_(torch.from_numpy(hw.blob))
mdist.broadcast_blob(blob, 0)
assert torch.isfinite(blob).all()
imgs = np.random.default_rng(77).uniform(-1, 1, (total, res, res, 3)).astype(np.float32)   # same on every rank
lo, hi = mdist.shard_range(total, world, rank)
oplan = orc.plan_build(alpha, res, classes)
mdist.barrier()
t0 = time.perf_counter()
mine, _ = orc.net_forward(oplan, blob.numpy(), imgs[lo:hi]) if hi > lo else (np.zeros((0, 1, 1, classes), np.float32), None)
dt = time.perf_counter() - t0 + 0.01 * rank
mdist.barrier()
slowest = mdist.max_over_ranks(dt)
counts = [b - a for a, b in (mdist.shard_range(total, world, r) for r in range(world))]
allrows = mdist.gather_rows(torch.from_numpy(mine.reshape(-1, classes)), counts)
res_d = {"rank": rank, "lo": lo, "hi": hi, "slowest": slowest, "dt": dt, "blob_sum": float(blob.double().sum())}
if rank == 0:
    full, _ = orc.net_forward(oplan, blob.numpy(), imgs)
    res_d["match"] = bool(np.array_equal(allrows.numpy(), full.reshape(total, classes)))
json.dump(res_d, open(os.path.join(out_dir, "rank%d.json" % rank), "w"))
mdist.shutdown()
