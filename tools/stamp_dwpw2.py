#!/usr/bin/env python3
"""In-kernel cycle stamps of the unified fused block kernel (mbn_f32_dwpw2.hip, dwpw_variant = 164): per wave of workgroup 0,
per step: D = depthwise math + filter DMA issue, L = x-window load issue, M = MFMA part, W = counted wait + barrier,
E = epilogue (only where a tile ended). usage: stamp_dwpw2.py [--block 6] [--batch 256]"""
import argparse, ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("MBN_LAB", "1")      # the lab build: every A/B variant and mbn_tune_set knob (make lab)
sys.path.insert(0, ROOT)
from mbn_amd import import_package
ap = argparse.ArgumentParser(); ap.add_argument("--block", type=int, default=6); ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--variant", type=int, default=164, help="164 = stamps; add ablation bits: 1 no x loads, 2 no depthwise math, 4 no stores, 8 no filter DMA, 16 no MFMA")
ap.add_argument("--brief", action="store_true")
args = ap.parse_args()
pkg = import_package(); lib = pkg.load(); ctx = pkg.Context(0)
plan = pkg.plan_build(1.0, 224, 1000, lib=lib)
ldw, lpw = plan.layer[args.block - 1], plan.layer[args.block]
n, h, oh, cin, cout, s = args.batch, ldw.in_rows, ldw.out_rows, ldw.in_ch, lpw.out_ch, ldw.stride
rng = np.random.default_rng(0)
x = rng.uniform(0, 6, (n, h, h, cin)).astype(np.float32)
d = [ctx.to_device(a) for a in (x, rng.normal(0, .5, (3, 3, cin)).astype(np.float32), np.ones(cin, np.float32), np.zeros(cin, np.float32),
                                rng.normal(0, .1, (cout, cin)).astype(np.float32), np.ones(cout, np.float32), np.zeros(cout, np.float32))]
out = ctx.alloc(n * oh * oh * cout * 4)
def run(v):
    lib.mbn_tune_set(b"dwpw_variant", v)
    rc = lib.mbn_dwpw_fused(ctx.h, out.ptr, d[0].ptr, d[1].ptr, d[2].ptr, d[3].ptr, d[4].ptr, d[5].ptr, d[6].ptr, n, h, h, oh, oh, cin, cout, s, ldw.pad_top, ldw.pad_left, None)
    lib.mbn_tune_set(b"dwpw_variant", 0); assert rc == 0
for _ in range(3): run(2)
run(args.variant); ctx.sync()
st = np.zeros((8, 96, 6), np.uint64)
lib.mbn_debug_dwpw2_stamps.argtypes = [C.c_void_p, C.c_size_t]
assert lib.mbn_debug_dwpw2_stamps(st.ctypes.data, st.nbytes) == 0
st = st.astype(np.int64)
nk = cin // 32
print("block %d-%d: Cin %d Cout %d stride %d, %d chunks per tile; cycles (s_memtime), workgroup 0" % (args.block, args.block + 1, cin, cout, s, nk))
print("wave  step  D(dw+dma)  L(ldx)  M(mfma)  taps  W(wait+barrier)  E(epilogue)  total")
for w in (() if args.brief else (0, 3, 4, 7)):
    for j in range(8, 8 + 2 * nk):
        t = st[w, j]
        nxt = st[w, j + 1, 0]
        print("%4d %5d %9d %8d %8d %6d %10d %12d %8d" % (w, j, t[1] - t[0], t[2] - t[1], t[3] - t[2], 0, t[4] - t[3], t[5] - t[4], nxt - t[0]))
valid = int((st[0, :, 0] > 0).sum()) - 1                     # stamped steps of wave 0 (the last one has no successor)
lo, hi = min(8, valid // 4), valid
tot = st[:, lo + 1:hi, 0] - st[:, lo:hi - 1, 0]
part = lambda k: (st[:, lo:hi - 1, k + 1] - st[:, lo:hi - 1, k]).mean()
print("variant %d " % args.variant, end="")
print("mean step over waves x steps %d..%d: %.0f cycles; per part: D %.0f  L %.0f  M %.0f  W %.0f  E %.0f" % (
    lo, hi - 1, tot.mean(), part(0), part(1), part(2), part(3), part(4)))
