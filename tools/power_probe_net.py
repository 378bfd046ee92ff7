#!/usr/bin/env python3
"""Package power and core clock while the whole network step, and single layer kernels, run back to back for a few seconds each (rocm-smi samples every ~50 ms).
Round 5: the fused block kernel was found at the 1400 W package limit (tools/power_probe.py); this is the same reading for the step `value` is measured on."""
import json, os, re, subprocess, sys, tempfile, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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
            pw = float(c.get("Current Socket Graphics Package Power (W)", "nan"))
            m = re.search(r"(\d+)", c.get("sclk clock speed:", ""))
            samples.append((time.time(), pw, float(m.group(1)) if m else float("nan")))
        except Exception:
            pass
        time.sleep(0.04)
threading.Thread(target=sampler, daemon=True).start()
def run(name, fn, per_call_images=None, seconds=3.0):
    for _ in range(5): fn()
    ctx.sync()
    t0 = time.time(); k = 0
    while time.time() - t0 < seconds:
        for _ in range(50): fn()
        ctx.sync(); k += 50
    t1 = time.time()
    mine = [(p, c) for (t, p, c) in samples if t0 + 0.5 <= t <= t1]
    p = np.median([a for a, _ in mine]) if mine else float("nan")
    c = np.median([b for _, b in mine]) if mine else float("nan")
    ms = 1000 * (t1 - t0) / k
    extra = "  %.0f images/s, %.1f images/J" % (per_call_images / ms * 1000, per_call_images / ms * 1000 / p) if per_call_images else ""
    print("%-46s %8.4f ms/call  power %5.0f W  sclk %4.0f MHz  (%d samples)%s" % (name, ms, p, c, len(mine), extra)); sys.stdout.flush()
def net(dtype, alpha, res, batch, streams):
    plan = pkg.plan_build(alpha, res, 1000, lib=lib)
    fd, path = tempfile.mkstemp(suffix=".h5"); os.close(fd)
    pkg.synthetic_h5(path, alpha=alpha, classes=1000, seed=1, lib=lib)
    hw = pkg.HostWeights(path, alpha=alpha, res=res, lib=lib); os.remove(path)
    n = pkg.Net(ctx, hw.plan, hw.blob.copy(), batch)
    if dtype == "bf16": n.set_dtype(pkg.DT_BF16)
    if streams > 1: n.set_streams(streams, free_running=True)
    imgs = np.random.default_rng(0).random((batch, res, res, 3), dtype=np.float32) * 2 - 1
    d_in, d_out = ctx.to_device(imgs), ctx.alloc(batch * 4000)
    return n, (lambda: n.forward(d_in.ptr, d_out.ptr, batch))
if "--stem" in sys.argv:
    plan = pkg.plan_build(1.0, 224, 1000, lib=lib)
    fd, path = tempfile.mkstemp(suffix=".h5"); os.close(fd)
    pkg.synthetic_h5(path, alpha=1.0, classes=1000, seed=1, lib=lib)
    hw = pkg.HostWeights(path, alpha=1.0, res=224, lib=lib); os.remove(path)
    for dtype in ("f32", "bf16"):
        nb = 256 if dtype == "f32" else 512
        ns = pkg.Net(ctx, hw.plan, hw.blob.copy(), nb)
        if dtype == "bf16": ns.set_dtype(pkg.DT_BF16)
        imgs = np.random.default_rng(0).random((nb, 224, 224, 3), dtype=np.float32) * 2 - 1
        d_in, d_o = ctx.to_device(imgs), ctx.alloc(nb * 112 * 112 * 64 * 4)
        run("fused stem (layers 1-3) %s b%d" % (dtype, nb), lambda: ns.forward(d_in.ptr, d_o.ptr, nb, 3))
        ns.set_fuse_blocks(1 << 4)
        d_o2 = ctx.alloc(nb * 56 * 56 * 128 * 4)
        run("layers 1-5 (stem + fused block 4-5) %s b%d" % (dtype, nb), lambda: ns.forward(d_in.ptr, d_o2.ptr, nb, 5))
        ns.destroy(); d_in.free(); d_o.free(); d_o2.free()
    stop.set(); sys.exit(0)
n1, f = net("f32", 1.0, 224, 256, 2); run("net fp32 1.0x224 b256, 2 streams (headline)", f, 256)
n1.set_streams(1); run("net fp32 1.0x224 b256, 1 stream", f, 256)
assert lib.mbn_tune_set(b"pw_emul", 6) == 0 and lib.mbn_tune_set(b"pw_emul_static", 1) == 0
run("net fp32 b256, opt-in pw_emul 6, 1 stream", f, 256)
lib.mbn_tune_set(b"pw_emul", 0); lib.mbn_tune_set(b"pw_emul_static", 0)
n1.destroy()
n2, f = net("bf16", 1.0, 224, 512, 1); run("net bf16 1.0x224 b512", f, 512); n2.destroy()
n3, f = net("bf16", 0.5, 160, 512, 1); run("net bf16 0.5x160 b512", f, 512); n3.destroy()
rng = np.random.default_rng(0)
def pw(m, k, nn, bf=False):
    x = rng.uniform(-1, 1, (m, k)).astype(np.float32); fl = rng.normal(0, (2.0 / k) ** .5, (nn, k)).astype(np.float32)
    d_x = ctx.to_device(pkg.f32_to_bf16_bits(x) if bf else x); d_f = ctx.to_device(pkg.f32_to_bf16_bits(fl) if bf else fl)
    d_sc, d_sh = ctx.to_device(np.ones(nn, np.float32)), ctx.to_device(np.zeros(nn, np.float32))
    d_o = ctx.alloc(m * nn * (2 if bf else 4))
    ext = pkg.make_ext(dtype=pkg.DT_BF16 if bf else pkg.DT_F32, act=2, scale=d_sc.ptr, shift=d_sh.ptr)
    return lambda: ctx.pointwise(d_o.ptr, d_x.ptr, d_f.ptr, m, 1, k, nn, ext)
run("pw_gemm fp32 L15 (50176 x 512 x 512)", pw(50176, 512, 512))
run("pw_gemm fp32 L7 (802816 x 128 x 128)", pw(802816, 128, 128))
run("pw bf16 L15 b512 (100352 x 512 x 512)", pw(100352, 512, 512, True))
def dwl(nb, h, c, s):
    x = rng.uniform(0, 6, (nb, h, h, c)).astype(np.float32); w = rng.normal(0, .5, (3, 3, c)).astype(np.float32)
    d_x, d_w = ctx.to_device(x), ctx.to_device(w)
    d_sc, d_sh = ctx.to_device(np.ones(c, np.float32)), ctx.to_device(np.zeros(c, np.float32))
    oh = h // s
    d_o = ctx.alloc(nb * oh * oh * c * 4)
    ext = pkg.make_ext(batch=nb, act=2, in_rows=h, in_cols=h, scale=d_sc.ptr, shift=d_sh.ptr)
    return lambda: ctx.depthwise(d_o.ptr, d_x.ptr, d_w.ptr, oh, oh, 3, s, c, ext)
run("depthwise fp32 L2 (256 x 112 x 112 x 32)", dwl(256, 112, 32, 1))
run("depthwise fp32 L14 (256 x 14 x 14 x 512)", dwl(256, 14, 512, 1))
stop.set()
