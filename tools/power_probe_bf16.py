#!/usr/bin/env python3
"""bf16 pointwise layer 15 at batch 512 (100352 x 512 x 512, pw_stream_bf16) and its ablation builds (lab exp1) back to back for ~2.5 s each with package power
and core clock sampled from rocm-smi: VERDICT r4 item 3 asked why matrix time and byte time ADD on this layer. (lab build)"""
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
m, k, n = 100352, 512, 512
rng = np.random.default_rng(0)
x = pkg.f32_to_bf16_bits(rng.uniform(-1, 1, (m, k)).astype(np.float32))
f = pkg.f32_to_bf16_bits(rng.normal(0, 0.06, (n, k)).astype(np.float32))
d_x, d_f = ctx.to_device(x), ctx.to_device(f)
d_sc, d_sh = ctx.to_device(np.ones(n, np.float32)), ctx.to_device(np.zeros(n, np.float32))
d_o = ctx.alloc(m * n * 2)
ext = pkg.make_ext(dtype=pkg.DT_BF16, act=2, scale=d_sc.ptr, shift=d_sh.ptr)
names = {0: "full kernel", 1: "no LDS-DMA (no activation / filter traffic)", 2: "no fragment reads", 3: "no DMA, no fragment reads (MFMA + stores)", 4: "no MFMA (DMA + fragment reads + stores)",
         5: "no DMA, no MFMA", 6: "no fragment reads, no MFMA (DMA + stores)", 7: "barriers + stores only", 16: "no stores"}
flops, bytes_ = 2.0 * m * k * n, 2.0 * (m * k + m * n + k * n)
print("bf16 pointwise 100352 x 512 x 512 (layer 15, batch 512): %.1f GFLOP, %.0f MB algorithmic; 2.5 PFLOP/s -> %.1f us, 8 TB/s -> %.1f us" % (flops / 1e9, bytes_ / 1e6, flops / 2.5e15 * 1e6, bytes_ / 8e12 * 1e6))
for e in (0, 4, 1, 3, 6, 2, 16, 5, 7):
    assert lib.mbn_tune_set(b"exp1", e) == 0
    call = lambda: ctx.pointwise(d_o.ptr, d_x.ptr, d_f.ptr, m, 1, k, n, ext)
    for _ in range(20): call()
    ctx.sync()
    t0 = time.time(); cnt = 0
    while time.time() - t0 < 2.5:
        for _ in range(200): call()
        ctx.sync(); cnt += 200
    t1 = time.time()
    mine = sorted((p, c) for (t, p, c) in samples if t0 + 0.5 <= t <= t1)
    p = mine[len(mine) // 2][0] if mine else float("nan")
    c = sorted(cc for _, cc in mine)[len(mine) // 2] if mine else float("nan")
    us = 1e6 * (t1 - t0) / cnt
    print("  exp1=%-2d %-52s %6.1f us  power %5.0f W  sclk %4.0f MHz  -> %5.1f mJ per launch" % (e, names[e], us, p, c, us * p / 1e3)); sys.stdout.flush()
lib.mbn_tune_set(b"exp1", 0)
stop.set()
