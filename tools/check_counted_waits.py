#!/usr/bin/env python3
"""Build-time check of the counted waits in pw_gemm (csrc/mbn_f32_pw.hip): `s_waitcnt vmcnt(NSTF); s_barrier` at the top of a tile is
right only if the fast epilogue of the previous tile emitted EXACTLY NSTF vector-memory stores and nothing else between the next tile's
first LDS-DMA and that wait. The source pins the order with sched_barrier(0); this script reads the ISA the compiler produced
(tools/isa.sh mbn_f32_pw -> /tmp/isa/mbn_f32_pw.s, ~10 s) and fails when, for a GLDS kernel with a counted wait vmcnt(N):
  * no basic block consists of exactly N `buffer_store_dword`(x2 for the paired bf16 form: `buffer_store_dword` too) and no other VMEM, or
  * some basic block holds both an LDS-DMA (`buffer_load_dwordx4 ... lds`) and a buffer store (the two were interleaved).
usage: check_counted_waits.py [isa file]   (exit 0 = ok)"""
import re
import subprocess
import sys
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernels(text):
    cur, name = None, None
    for line in text.splitlines():
        m = re.match(r"^(_ZN\S*pw_gemm\S*):", line)
        if m:
            name, cur = m.group(1), []
            continue
        if cur is not None:
            cur.append(line)
            if "s_endpgm" in line:
                yield name, cur
                cur = None


def check(path):
    text = open(path).read()
    bad, seen = [], 0
    for name, lines in kernels(text):
        waits = [int(m.group(1)) for i, l in enumerate(lines) if (m := re.search(r"s_waitcnt vmcnt\((\d+)\) lgkmcnt\(0\)", l))
                 and i > 0 and "ASMSTART" in lines[i - 1]]
        waits = [w for w in waits if w > 0]
        if not waits:
            continue
        seen += 1
        n = waits[0]
        blocks, cur = [], []
        for l in lines:
            if re.match(r"^\.LBB", l) or re.search(r"\bs_c?branch", l):
                blocks.append(cur)
                cur = []
            cur.append(l)
        blocks.append(cur)
        exact = False
        for b in blocks:
            vm = [l.split()[0] for l in b if re.match(r"\s+(buffer_|global_|flat_|scratch_)", l)]
            stores = [v for v in vm if v.startswith("buffer_store")]
            dma = [l for l in b if re.search(r"buffer_load_dwordx4 .* lds", l) or "global_load_lds" in l]
            if stores and dma:
                bad.append("%s: a block interleaves LDS-DMA and buffer stores" % name)
            if stores and len(stores) == n and len(vm) == n:
                exact = True
        if not exact:
            bad.append("%s: counted wait vmcnt(%d) but no block of exactly %d buffer stores" % (name, n, n))
    if seen == 0:
        bad.append("no pw_gemm kernel with a counted wait found in %s" % path)
    return seen, bad


if __name__ == "__main__":
    path = sys.argv[1] if len(sys.argv) > 1 else "/tmp/isa/mbn_f32_pw.s"
    if len(sys.argv) <= 1:
        subprocess.check_call(["bash", os.path.join(ROOT, "tools", "isa.sh"), "mbn_f32_pw"], stdout=subprocess.DEVNULL)
    seen, bad = check(path)
    print("%d kernels with a counted wait checked" % seen)
    for b in bad:
        print("FAIL:", b)
    sys.exit(1 if bad else 0)
