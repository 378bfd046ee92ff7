#!/usr/bin/env python3
"""Does the chip run the fused block kernel at its power limit? Loops one variant of mbn_dwpw_fused (block 6-7, batch 256) for ~3 s per variant while a thread
samples `rocm-smi --showpower --showclocks --json` (lab build). Prints per variant: kernel ms (HIP events over the last launches), and the samples' median
power / sclk. usage: power_probe.py [--variants 0,9,113,15475]   (dwpw_variant values: 0 shipped, 9 burst form, 100 + ablation bits)"""
import argparse, json, os, subprocess, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("MBN_LAB", "1")
sys.path.insert(0, ROOT)
from mbn_amd import import_package
ap = argparse.ArgumentParser()
ap.add_argument("--variants", default="0,9,113,15475")
ap.add_argument("--block", type=int, default=6)
ap.add_argument("--seconds", type=float, default=3.0)
args = ap.parse_args()
pkg = import_package(); lib = pkg.load(); ctx = pkg.Context(0)
from bench import smi_card_of  # noqa: E402
try:
    SMI_CARD = smi_card_of(ctx.pci_bus_id())      # the card the context holds, by PCI bus id
except Exception:
    SMI_CARD = None
plan = pkg.plan_build(1.0, 224, 1000, lib=lib)
ldw, lpw = plan.layer[args.block - 1], plan.layer[args.block]
n, h, oh, cin, cout, s = 256, ldw.in_rows, ldw.out_rows, ldw.in_ch, lpw.out_ch, ldw.stride
rng = np.random.default_rng(0)
x = rng.uniform(0, 6, (n, h, h, cin)).astype(np.float32)
d = [ctx.to_device(a) for a in (x, rng.normal(0, .5, (3, 3, cin)).astype(np.float32), np.ones(cin, np.float32), np.zeros(cin, np.float32),
                                rng.normal(0, .1, (cout, cin)).astype(np.float32), np.ones(cout, np.float32), np.zeros(cout, np.float32))]
out = ctx.alloc(n * oh * oh * cout * 4)
samples, stop = [], threading.Event()
def sampler():
    while not stop.is_set():
        try:
            r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5)
            j = json.loads(r.stdout)
            c = j[SMI_CARD or sorted(j)[0]]
            samples.append((time.time(), c))
        except Exception as e:
            samples.append((time.time(), {"error": str(e)}))
        time.sleep(0.05)
th = threading.Thread(target=sampler, daemon=True); th.start()
def launch(v):
    lib.mbn_tune_set(b"dwpw_variant", v)
    rc = lib.mbn_dwpw_fused(ctx.h, out.ptr, d[0].ptr, d[1].ptr, d[2].ptr, d[3].ptr, d[4].ptr, d[5].ptr, d[6].ptr, n, h, h, oh, oh, cin, cout, s, ldw.pad_top, ldw.pad_left, None)
    lib.mbn_tune_set(b"dwpw_variant", 0); assert rc == 0
for v in [int(t) for t in args.variants.split(",")]:
    for _ in range(20): launch(v)
    ctx.sync()
    t0 = time.time(); k = 0
    while time.time() - t0 < args.seconds:
        for _ in range(200): launch(v)
        ctx.sync(); k += 200
    t1 = time.time()
    ctx.profile_begin(20)
    for _ in range(20): launch(v)
    ms = np.median(ctx.profile_end(20))
    mine = [c for (t, c) in samples if t0 + 0.5 <= t <= t1]
    def num(key_part):
        vals = []
        for c in mine:
            for kk, vv in c.items():
                if key_part in kk.lower():
                    try: vals.append(float(str(vv).strip("()MhzW ").replace("Mhz", "")))
                    except ValueError: pass
        return np.median(vals) if vals else float("nan")
    print("variant %6d: %.4f ms/launch (events), %.4f ms wall/launch over %d launches; samples %d: power %.0f W, sclk %.0f MHz, mclk %.0f MHz"
          % (v, ms, 1000 * (t1 - t0) / k, k, len(mine), num("power"), num("sclk"), num("mclk")))
    if v == int(args.variants.split(",")[0]) and mine:
        print("   (one raw sample: %s)" % json.dumps(mine[len(mine) // 2])[:400])
    sys.stdout.flush()
stop.set()
