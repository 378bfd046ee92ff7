#!/usr/bin/env python3
"""Build-time check of the counted waits in pw_gemm (csrc/mbn_f32_pw.hip): `s_waitcnt vmcnt(NSTF); s_barrier` at the top of a tile is
right only if the fast epilogue of the previous tile emitted EXACTLY NSTF vector-memory stores and nothing else between the next tile's
first LDS-DMA and that wait. The source pins the order with sched_barrier(0); this script reads the ISA the compiler produced
(tools/isa.sh mbn_f32_pw -> /tmp/isa/mbn_f32_pw.s, ~10 s) and fails when, for a GLDS kernel with a counted wait vmcnt(N):
  * no basic block consists of exactly N `buffer_store_dword`(x2 for the paired bf16 form: `buffer_store_dword` too) and no other VMEM, or
  * some basic block holds both an LDS-DMA (`buffer_load_dwordx4 ... lds`) and a buffer store (the two were interleaved).
The fused block kernel (csrc/mbn_f32_dwpw2.hip, waits vmcnt(NX + NST)) is checked the same way for its NST = 16 / 32 buffer_store_dwordx2.
usage: check_counted_waits.py [isa file]   (exit 0 = ok)"""
import re
import subprocess
import sys
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernels(text, stem="pw_gemm"):
    cur, name = None, None
    for line in text.splitlines():
        m = re.match(r"^(_ZN\S*" + stem + r"\S*):", line)
        if m:
            name, cur = m.group(1), []
            continue
        if cur is not None:
            cur.append(line)
            if "s_endpgm" in line:
                yield name, cur
                cur = None


def check(path, stem="pw_gemm", nst_of=None):
    """nst_of: None = the counted wait's own value is the store count (pw_gemm); else a function kernel name -> store count (the fused block
    kernel waits for NX + NST: the stores are one term of its counts)."""
    text = open(path).read()
    bad, seen = [], 0
    for name, lines in kernels(text, stem):
        waits = [int(m.group(1)) for i, l in enumerate(lines) if (m := re.search(r"s_waitcnt vmcnt\((\d+)\) lgkmcnt\(0\)", l))
                 and i > 0 and "ASMSTART" in lines[i - 1]]
        waits = [w for w in waits if w > 0]
        if not waits:
            continue
        seen += 1
        n = nst_of(name) if nst_of else waits[0]
        if nst_of and not any(w >= n for w in waits):
            bad.append("%s: no counted wait leaves %d stores in flight (waits %s)" % (name, n, sorted(set(waits))))
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


def dwpw2_nst(name):
    """dwpw2_f32<S, BN, ...>: NST = 16 * MI * (NI / 2) buffer_store_dwordx2 per lane: 16 (BN = 128: MI 1, NI 2) or 32 (BN = 256: MI 2, NI 2)"""
    return 32 if "ILi1ELi256E" in name or "ILi2ELi256E" in name else 16


if __name__ == "__main__":
    path = sys.argv[1] if len(sys.argv) > 1 else "/tmp/isa/mbn_f32_pw.s"
    if len(sys.argv) <= 1:
        subprocess.check_call(["bash", os.path.join(ROOT, "tools", "isa.sh"), "mbn_f32_pw"], stdout=subprocess.DEVNULL)
        subprocess.check_call(["bash", os.path.join(ROOT, "tools", "isa.sh"), "mbn_f32_dwpw2"], stdout=subprocess.DEVNULL)
    seen, bad = check(path)
    if len(sys.argv) <= 1:
        s2, b2 = check("/tmp/isa/mbn_f32_dwpw2.s", "dwpw2_f32", dwpw2_nst)
        print("%d fused block kernels with counted waits checked" % s2)
        if s2 == 0:
            b2.append("no dwpw2_f32 kernel with a counted wait found")
        bad += b2
    print("%d kernels with a counted wait checked" % seen)
    for b in bad:
        print("FAIL:", b)
    sys.exit(1 if bad else 0)
