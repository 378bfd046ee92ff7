#!/usr/bin/env python3
"""Instruction census of the kernels in a gfx950 assembly listing (hipcc --cuda-device-only -S): per kernel the counts by class,
and with --loop the same for the largest loop body (label ... backward branch), i.e. the steady state of a persistent kernel.
usage: isa_count.py file.s [substring of the kernel name] [--loop] [--top N]"""
import collections
import re
import sys

args = [a for a in sys.argv[1:] if not a.startswith("--")]
want = args[1] if len(args) > 1 else ""
loop = "--loop" in sys.argv
top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 0
lines = open(args[0]).read().split("\n")
starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
for si, st in enumerate(starts):
    name = lines[st].split(":")[0]
    if want not in name:
        continue
    end = next((i for i in range(st, len(lines)) if "s_endpgm" in lines[i]), len(lines))
    body = lines[st:end]
    if loop:
        labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\w+):", l)] if m}
        best = None
        for i, l in enumerate(body):
            m = re.search(r"s_cbranch\w*\s+(\.LBB\w+)|s_branch\s+(\.LBB\w+)", l)
            if m:
                t = m.group(1) or m.group(2)
                if t in labels and labels[t] < i and (best is None or i - labels[t] > best[1] - best[0]):
                    best = (labels[t], i)
        if best:
            body = body[best[0]:best[1] + 1]
    ins = [l.strip().split()[0] for l in body if l.startswith("\t") and not l.strip().startswith((".", ";"))]
    c = collections.Counter(ins)
    cls = collections.Counter()
    for k, v in c.items():
        if k.startswith("v_mfma"): cls["mfma"] += v
        elif k.startswith("v_"): cls["valu"] += v
        elif k.startswith("s_waitcnt"): cls["waitcnt"] += v
        elif k.startswith("s_nop"): cls["nop"] += v
        elif k.startswith("s_"): cls["salu"] += v
        elif k.startswith("ds_read") or k.startswith("ds_load"): cls["ds_read"] += v
        elif k.startswith("ds_write") or k.startswith("ds_store"): cls["ds_write"] += v
        elif k.startswith("buffer_load") or k.startswith("global_load"): cls["vmem_load"] += v
        elif k.startswith("buffer_store") or k.startswith("global_store"): cls["vmem_store"] += v
        elif k.startswith("scratch_"): cls["scratch"] += v
        else: cls["other"] += v
    print(name[-48:], "total", len(ins), dict(cls))
    if top:
        print("   ", c.most_common(top))
