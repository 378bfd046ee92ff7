#!/usr/bin/env python3
"""Build-time check of the counted waits in pw_gemm (csrc/mbn_f32_pw.hip): `s_waitcnt vmcnt(NSTF); s_barrier` at the top of a tile is
right only if the fast epilogue of the previous tile emitted EXACTLY NSTF vector-memory stores and nothing else between the next tile's
first LDS-DMA and that wait. The source pins the order with sched_barrier(0); this script reads the ISA the compiler produced
(tools/isa.sh mbn_f32_pw -> a temporary directory, ~10 s) and fails when, for a GLDS kernel with a counted wait vmcnt(N):
  * no basic block consists of exactly N `buffer_store_dword`(x2 for the paired bf16 form: `buffer_store_dword` too) and no other VMEM
    AND sits where the wait assumes it: the nearest VMEM-carrying block in front of it holds the LDS-DMA (and no store), and a counted wait
    that leaves >= N operations in flight follows it before the next LDS-DMA (an unrelated block of N stores must not satisfy the check), or
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
    """The invariant behind `s_waitcnt vmcnt(W); s_barrier` with W > 0: the wait must not let the LDS-DMA it guards stay in flight, and should not
    wait for anything younger. On the kernel's control-flow graph (labels, s_branch / s_cbranch successors — the text order of the blocks is not
    the execution order: pw_gemm's epilogue sits at the loop's end, its DMA and wait at the head), walk every path from each LDS-DMA instruction
    forward to the next counted wait, counting the vector-memory instructions issued behind the DMA: for every counted wait W some path must carry exactly W of them (the walk
    is path-insensitive: the waits are chosen by run-time flags such as "the previous tile took the fast epilogue", so infeasible combinations
    show up as other counts; what must exist is the path on which the W youngest operations are exactly the ones issued behind the DMA —
    ADVICE r4: an unrelated block of N stores elsewhere in the kernel no longer satisfies the check), and a DMA's basic block holds no stores. nst_of (fused block kernel): kernel name ->
    NST; at least one wait must leave >= NST operations in flight, i.e. the deferred-epilogue path exists."""
    text = open(path).read()
    bad, seen = [], 0
    is_dma = lambda l: bool(re.search(r"buffer_load_dwordx4 .* lds", l) or "global_load_lds" in l)
    is_vm = lambda l: bool(re.match(r"\s+(buffer_|global_|flat_|scratch_)", l))
    for name, lines in kernels(text, stem):
        def wait_at(k):
            m = re.search(r"s_waitcnt vmcnt\((\d+)\) lgkmcnt\(0\)", lines[k])
            return int(m.group(1)) if (m and k > 0 and "ASMSTART" in lines[k - 1]) else None
        waits = [w for k in range(len(lines)) if (w := wait_at(k))]
        if not waits:
            continue
        seen += 1
        if nst_of and not any(w >= nst_of(name) for w in waits):
            bad.append("%s: no counted wait leaves %d stores in flight (waits %s)" % (name, nst_of(name), sorted(set(waits))))
        label = {m.group(1): k for k, l in enumerate(lines) if (m := re.match(r"^(\.LBB\w+):", l))}
        def succ(k):
            l = lines[k]
            m = re.search(r"\bs_(c?)branch\w*\s+(\.LBB\w+)", l)
            if m:
                return ([label[m.group(2)]] if m.group(2) in label else []) + ([k + 1] if m.group(1) else [])
            return [] if "s_endpgm" in l else [k + 1]
        starts = [k for k, l in enumerate(lines) if is_dma(l)]
        seen_state, stack, checked, reached = set(), [(k + 1, 0) for k in starts], 0, {}
        while stack:
            k, cnt = stack.pop()
            if k >= len(lines) or (k, cnt) in seen_state or cnt > 200:
                continue
            seen_state.add((k, cnt))
            l = lines[k]
            if is_dma(l):
                continue                                    # a younger DMA: its own walk starts there
            w = wait_at(k)
            if w is not None:
                if w > 0:
                    checked += 1
                    reached.setdefault(w, set()).add(cnt)
                continue                                    # this wait ends the DMA's exposure (w == 0 drains everything)
            if re.search(r"s_waitcnt vmcnt\(0\)", l) or "s_waitcnt_vscnt" in l:
                continue
            for nk in succ(k):
                stack.append((nk, cnt + (1 if is_vm(l) else 0)))
        if checked == 0:
            bad.append("%s: no path from an LDS-DMA reaches a counted wait" % name)
        for w, cnts in reached.items():
            if w not in cnts:
                bad.append("%s: no path issues exactly %d vector-memory instructions between an LDS-DMA and `s_waitcnt vmcnt(%d)` (seen: %s)"
                           % (name, w, w, sorted(cnts)[:8]))
        blk = []                                            # and the DMA group is never interleaved with stores inside one basic block
        for l in lines + [".LBB_end:"]:
            if re.match(r"^\.LBB", l) or re.search(r"\bs_c?branch", l):
                if any(is_dma(x) for x in blk) and any(x.split()[0].startswith("buffer_store") for x in blk if is_vm(x)):
                    bad.append("%s: a block interleaves LDS-DMA and buffer stores" % name)
                blk = []
            blk.append(l)
    if seen == 0:
        bad.append("no %s kernel with a counted wait found in %s" % (stem, path))
    return seen, sorted(set(bad))


def check_spills(path, stem, in_loop_only=False):
    """Shipped block kernels must not touch scratch where it costs: a scratch reload is an `s_waitcnt vmcnt(0)` that drains the window loads in flight
    (profiles/r05/h_*: 0.245 -> 0.50 ms with spills in the step). dwpw2 forms: no scratch instruction at all. dwpw3 (in_loop_only): none
    between a loop header and its backward branch (its 256-VGPR form reloads one register once, behind the loop, in front of the last tile's stores)."""
    text = open(path).read()
    bad, seen = [], 0
    for name, lines in kernels(text, stem):
        seen += 1
        idx = [k for k, l in enumerate(lines) if re.match(r"\s+scratch_", l)]
        if not idx:
            continue
        if not in_loop_only:
            bad.append("%s: %d scratch instructions (spills) in a shipped block kernel" % (name, len(idx)))
            continue
        label = {m.group(1): k for k, l in enumerate(lines) if (m := re.match(r"^(\.LBB\w+):", l))}
        loops = []
        for k, l in enumerate(lines):
            m = re.search(r"\bs_c?branch\w*\s+(\.LBB\w+)", l)
            if m and m.group(1) in label and label[m.group(1)] < k:
                loops.append((label[m.group(1)], k))
        inside = [k for k in idx if any(a <= k <= b for a, b in loops)]
        # pw3's K = 512 form (opt-in by pw_tile 9, never the default rule): one reload of a store offset in the tile's epilogue, none in its k loop
        if "pw3_f32ILi512E" in name and len(inside) <= 1:
            continue
        if inside:
            bad.append("%s: %d scratch instructions inside a loop" % (name, len(inside)))
    if seen == 0:
        bad.append("no %s kernel found in %s" % (stem, path))
    return seen, bad


def dwpw2_nst(name):
    """dwpw2_f32<S, BN, ...>: NST = 16 * MI * (NI / 2) buffer_store_dwordx2 per lane: 16 (BN = 128: MI 1, NI 2) or 32 (BN = 256: MI 2, NI 2)"""
    return 32 if "ILi1ELi256E" in name or "ILi2ELi256E" in name else 16


if __name__ == "__main__":
    import tempfile
    tmp = tempfile.mkdtemp(prefix="mbn_isa_")
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(tmp, "mbn_f32_pw.s")
    if len(sys.argv) <= 1:
        env = dict(os.environ, ISA_OUT=tmp)
        subprocess.check_call(["bash", os.path.join(ROOT, "tools", "isa.sh"), "mbn_f32_pw"], stdout=subprocess.DEVNULL, env=env)
        subprocess.check_call(["bash", os.path.join(ROOT, "tools", "isa.sh"), "mbn_f32_dwpw2"], stdout=subprocess.DEVNULL, env=env)
        subprocess.check_call(["bash", os.path.join(ROOT, "tools", "isa.sh"), "mbn_bf16_dwpw2"], stdout=subprocess.DEVNULL, env=env)
        subprocess.check_call(["bash", os.path.join(ROOT, "tools", "isa.sh"), "mbn_f32_dwpw3"], stdout=subprocess.DEVNULL, env=env)
        subprocess.check_call(["bash", os.path.join(ROOT, "tools", "isa.sh"), "mbn_f32_pw3"], stdout=subprocess.DEVNULL, env=env)
        subprocess.check_call(["bash", os.path.join(ROOT, "tools", "isa.sh"), "mbn_bf16_res"], stdout=subprocess.DEVNULL, env=env)
        subprocess.check_call(["bash", os.path.join(ROOT, "tools", "isa.sh"), "mbn_bf16_tail"], stdout=subprocess.DEVNULL, env=env)
    seen, bad = check(path)
    if len(sys.argv) <= 1:
        s2, b2 = check(os.path.join(tmp, "mbn_f32_dwpw2.s"), "dwpw2_f32", dwpw2_nst)
        print("%d fused block kernels with counted waits checked" % s2)
        if s2 == 0:
            b2.append("no dwpw2_f32 kernel with a counted wait found")
        bad += b2
        # round 6: the bf16 block kernel's counted waits (FO / H32 instantiations included: the shipped build's set), and no spills in any shipped block kernel
        s3, b3 = check(os.path.join(tmp, "mbn_bf16_dwpw2.s"), "dwpw2_bf16")
        print("%d bf16 fused block kernels with counted waits checked" % s3)
        bad += b3
        for f, st, loop_only in (("mbn_f32_dwpw2.s", "dwpw2_f32", False), ("mbn_bf16_dwpw2.s", "dwpw2_bf16", False), ("mbn_f32_dwpw3.s", "dwpw3_f32", True),
                                  ("mbn_f32_pw3.s", "pw3_f32", True), ("mbn_bf16_res.s", "res_blocks_bf16", False),
                                  ("mbn_bf16_tail.s", "tail_bf16", False)):
            s4, b4 = check_spills(os.path.join(tmp, f), st, loop_only)
            print("%d %s kernels checked for scratch use" % (s4, st))
            bad += b4
    print("%d kernels with a counted wait checked" % seen)
    for b in bad:
        print("FAIL:", b)
    sys.exit(1 if bad else 0)
