#!/usr/bin/env python3
"""fp32 pointwise layer 15 at batch 256 (50176 x 512 x 512, pw_gemm<float,64,64>) and its ablation builds (lab exp1 bits: 1 no LDS-DMA, 2 no barriers, 4 no stores, 8 no fragment
reads; 15 = MFMAs + bookkeeping only) back to back for ~2.5 s each with package power and core clock sampled from rocm-smi: how much of the dominant kernel's energy is the MFMAs? (lab build)"""
import json, os, re, subprocess, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("MBN_LAB", "1")
sys.path.insert(0, ROOT)
from mbn_amd import import_package
pkg = import_package(); lib = pkg.load(); ctx = pkg.Context(0)
from bench import smi_card_of  # noqa: E402
try:
    SMI_CARD = smi_card_of(ctx.pci_bus_id())      # the card the context holds, by PCI bus id
except Exception:
    SMI_CARD = None
samples, stop = [], threading.Event()
def sampler():
    while not stop.is_set():
        try:
            r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5)
            c = json.loads(r.stdout); c = c[SMI_CARD or sorted(c)[0]]
            m = re.search(r"(\d+)", c.get("sclk clock speed:", ""))
            samples.append((time.time(), float(c.get("Current Socket Graphics Package Power (W)", "nan")), float(m.group(1)) if m else float("nan")))
        except Exception:
            pass
        time.sleep(0.04)
threading.Thread(target=sampler, daemon=True).start()
rng = np.random.default_rng(0)
names = {0: "full kernel", 1: "no LDS-DMA", 4: "no stores", 8: "no fragment reads", 9: "no DMA, no fragment reads", 12: "no stores, no fragment reads", 15: "MFMAs + bookkeeping only", 7: "no DMA, no barriers, no stores"}
for (m, k, n, tag) in ((50176, 512, 512, "L15"), (12544, 1024, 1024, "L27")):
    x = rng.uniform(-1, 1, (m, k)).astype(np.float32)
    f = rng.normal(0, (2.0 / k) ** 0.5, (n, k)).astype(np.float32)
    d_x, d_f = ctx.to_device(x), ctx.to_device(f)
    d_sc, d_sh = ctx.to_device(np.ones(n, np.float32)), ctx.to_device(np.zeros(n, np.float32))
    d_o = ctx.alloc(m * n * 4)
    ext = pkg.make_ext(act=2, scale=d_sc.ptr, shift=d_sh.ptr)
    flops = 2.0 * m * k * n
    print("fp32 pointwise %s: %d x %d x %d, %.1f GFLOP; 157.3 TFLOP/s -> %.1f us" % (tag, m, k, n, flops / 1e9, flops / 157.3e12 * 1e6))
    for e in (0, 15, 1, 4, 8, 9, 12):
        assert lib.mbn_tune_set(b"exp1", e) == 0
        call = lambda: ctx.pointwise(d_o.ptr, d_x.ptr, d_f.ptr, m, 1, k, n, ext)
        for _ in range(20): call()
        ctx.sync()
        t0 = time.time(); cnt = 0
        while time.time() - t0 < 2.5:
            for _ in range(100): call()
            ctx.sync(); cnt += 100
        t1 = time.time()
        mine = sorted((p, c) for (t, p, c) in samples if t0 + 0.5 <= t <= t1)
        p = mine[len(mine) // 2][0] if mine else float("nan")
        c = sorted(cc for _, cc in mine)[len(mine) // 2] if mine else float("nan")
        us = 1e6 * (t1 - t0) / cnt
        print("  exp1=%-2d %-34s %6.1f us  %5.1f TFLOP/s  power %5.0f W  sclk %4.0f MHz  -> %6.1f mJ per launch" % (e, names[e], us, flops / us / 1e6, p, c, us * p / 1e3)); sys.stdout.flush()
    lib.mbn_tune_set(b"exp1", 0)
    for b in (d_x, d_f, d_sc, d_sh, d_o): b.free()
stop.set()
