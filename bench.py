#!/usr/bin/env python3
"""bench.py — images/sec of the MobileNet-V1 hot path on MI355X (BASELINE.json metric).

  python bench.py --gpus 1 --steps K --warmup W            one GPU
  python bench.py --gpus N --steps K --warmup W            N > 1 without a launcher: starts the line below as a child process
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W            one rank per GPU (RCCL)

A "step" = one forward of all 29 layers (MobileNet.c:240-2763 order) over one batch of synthetic 224x224x3 fp32
images that is already resident in HBM. Workload at N=1 = BASELINE.json configs[2]: MobileNet-V1 1.0x224, fp32,
batch 256. With N ranks every rank runs the same per-GPU batch on its own images (weak scaling, independent images:
no data-path collective); the packed parameter blob is broadcast once from rank 0 over RCCL before timing.

The LAST line of stdout (rank 0) is ONE compact JSON line (< 4 KB: the driver keeps an 8 KB tail): metric, value, unit, n_gpus, steps,
warmup, ms_per_step, dtype, data, config, roofline, cpu_baseline, parity_check, stages_frac, configs_alt, and `full_record` = the path
of the side file that holds everything else (per-launch table, unfused per-stage table, CPU variants, pw_emul_alt, every `how` string).
  value         batch x K / the wall time of EXACTLY K forwards in the default configuration, barrier + device sync on both sides.
                Nothing else runs inside that region: the steps whose kernels are timed one by one run AFTER it.
  roofline      the dominant kernel (pw_gemm<float>: the stand-alone pointwise GEMMs): algorithmic FLOPs per launch / average launch
                duration from HIP events on the kernel's stream over `profiled_steps` untimed single-stream forwards, against the
                fp32 MFMA peak 157.3 TFLOP/s (MI355X_MICROARCH.md). `held_clock_ghz` = the core clock those launches held
                (s_memtime / s_memrealtime inside the kernel), `frac_at_held_clock` = achieved / (peak x held / 2.4 GHz): comparable
                across boxes. Flat scalars beside them: depthwise x13 against 8 TB/s, pointwise x13 against the MFMA peak, blocks, stem.
  cpu_baseline  the oracle's C restatement (kind "port": the reference has no CPU path and cannot be built here) timed on this box's
                host cores on a bounded sample of the same workload: the FIRST n images of the timed batch.
  parity_check  the oracle's logits for those n images against the logits the timed steps left on the device; the run exits
                non-zero above the tolerance (fp32 1e-3, bf16 2e-2 of max|ref|: SURVEY.md §8c, tests/test_parity_gpu.py).
  stages_frac   {stage: [ms, frac of 8 TB/s, frac of the MFMA peak]} of the default configuration and of the one-launch-per-layer pass.
  configs_alt   {name: [images/sec, roofline frac, parity ok]}: BASELINE.json configs[4] (bf16, batch 512) and configs[1] (batch 1).
  ranks         N > 1: per rank [rank, device ordinal, PCI bus id, images per step, seconds], `collective_world_size`, `backend`.
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32 matrix peak


def layer_work(l, batch, pkg, act_bytes=4.0):
    """Algorithmic FLOPs and HBM bytes of one layer launch (input read once, output
    written once; weights counted once per launch). act_bytes = 4 (fp32) or 2 (bf16 activations; conv1 still
    reads the fp32 image, FC still writes fp32 logits)."""
    in_b = (4.0 if l.kind == pkg.L_CONV else act_bytes) * l.in_rows * l.in_cols * l.in_ch * batch
    out_b = (4.0 if l.kind == pkg.L_FC else act_bytes) * l.out_rows * l.out_cols * l.out_ch * batch
    px = float(l.out_rows * l.out_cols * batch)
    if l.kind == pkg.L_CONV:
        flops, w = 2.0 * 27 * l.out_ch * px, 4.0 * 27 * l.out_ch
    elif l.kind == pkg.L_DW:
        flops, w = 2.0 * 9 * l.out_ch * px, 4.0 * 9 * l.out_ch
    elif l.kind in (pkg.L_PW, pkg.L_FC):
        flops, w = 2.0 * l.in_ch * l.out_ch * px, act_bytes * l.in_ch * l.out_ch
    else:
        flops, w = float(l.in_rows * l.in_cols * l.in_ch * batch), 0.0
    return flops, in_b + out_b + w


def layer_weight_bytes(l, pkg, act_bytes=4.0):
    if l.kind == pkg.L_CONV:
        return 4.0 * 27 * l.out_ch
    if l.kind == pkg.L_DW:
        return 4.0 * 9 * l.out_ch
    if l.kind in (pkg.L_PW, pkg.L_FC):
        return act_bytes * l.in_ch * l.out_ch
    return 0.0


def load_traffic(layers=None, key="f32"):
    """HBM bytes per launch of the dominant kernel from the PMC counters (FETCH_SIZE x2 on gfx950 + WRITE_SIZE), measured
    in separate rocprofv3 --pmc passes of this same workload (tools/pmc_pass.sh) and committed under profiles/.
    bench.py cannot run the profiler on itself, so it reports the committed measurement — averaged over the pointwise
    layers that are separate launches in this run — or null when there is none. Returns (bytes, source) where source
    names the git sha and command the committed figure was measured at."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        if key != "f32":
            t = t[key]
        src = {k: t.get(k) for k in ("git_sha", "command", "date") if t.get(k)} or None
        if layers:
            return sum(t["pointwise_layers"][str(l)]["bytes"] for l in layers) / len(layers), src
        return t["pointwise_avg_bytes_per_launch"], src
    except Exception:
        return None, None


def stage_table(plan, pkg, launches, layer_ms, batch, act_bytes, mfma_peak):
    """Per-launch and per-stage-group achieved GB/s / TFLOP/s from the launch list and its measured milliseconds."""
    def launch_work(idx):
        """Algorithmic work of one launch: FLOPs of every layer in it; bytes = first layer's input + last layer's
        output + every layer's weights (a fused launch's intermediates never reach HBM)."""
        fl = sum(layer_work(plan.layer[i], batch, pkg, act_bytes)[0] for i in idx)
        first, last = plan.layer[idx[0]], plan.layer[idx[-1]]
        by = sum(layer_weight_bytes(plan.layer[i], pkg, act_bytes) for i in idx)
        by += (4.0 if first.kind == pkg.L_CONV else act_bytes) * first.in_rows * first.in_cols * first.in_ch * batch
        by += (4.0 if last.kind == pkg.L_FC else act_bytes) * last.out_rows * last.out_cols * last.out_ch * batch
        return fl, by
    kind_name = {pkg.L_CONV: "conv1", pkg.L_DW: "depthwise", pkg.L_PW: "pointwise", pkg.L_POOL: "pool", pkg.L_FC: "fc"}
    stage_of = ["stem_fused" if len(idx) == 3 else "tail_fused" if len(idx) == 2 and plan.layer[idx[0]].kind == pkg.L_POOL else
                "block_fused" if len(idx) == 2 else "blocks_resident" if len(idx) > 3 else kind_name[plan.layer[idx[0]].kind] for idx in launches]
    stages, per_layer = {}, []
    for j, idx in enumerate(launches):
        f, b = launch_work(idx)
        per_layer.append({"layers": [i + 1 for i in idx], "stage": stage_of[j], "ms": round(float(layer_ms[j]), 5),
                          "GBps": round(b / layer_ms[j] / 1e6, 1), "TFLOPs": round(f / layer_ms[j] / 1e9, 2)})
    for name in ["stem_fused", "block_fused", "blocks_resident", "conv1", "depthwise", "pointwise", "pool", "fc", "tail_fused"]:
        js = [j for j in range(len(launches)) if stage_of[j] == name]
        if not js:
            continue
        fl = sum(launch_work(launches[j])[0] for j in js)
        by = sum(launch_work(launches[j])[1] for j in js)
        ms_sum = float(sum(layer_ms[j] for j in js))
        st = {"launches": len(js), "ms": round(ms_sum, 4), "GBps": round(by / ms_sum / 1e6, 1),
              "TFLOPs": round(fl / ms_sum / 1e9, 2)}
        st["frac_hbm"] = round(st["GBps"] / HBM_PEAK_GBS, 4)
        st["frac_mfma"] = round(st["TFLOPs"] / mfma_peak, 4)
        stages[name] = st
    if "stem_fused" in stages:
        stages["stem_fused"]["layers"] = "1-3 (conv1 + depthwise + pointwise in one kernel)"
    if "block_fused" in stages:
        stages["block_fused"]["layers"] = ", ".join("%d-%d" % (idx[0] + 1, idx[1] + 1) for j, idx in enumerate(launches) if stage_of[j] == "block_fused") \
                                          + " (depthwise + pointwise in one kernel each)"
    if "tail_fused" in stages:
        stages["tail_fused"]["layers"] = "28-29 (average pool + FC in one kernel: 1...4 images)"
    return stages, per_layer, stage_of


def smi_card_of(bus_id):
    """rocm-smi's card name ("card3") of the GPU with this PCI bus id ("0000:05:00.0", as mbn_device_pci_bus_id prints it), from `rocm-smi --showbus --json`.
    None when no card matches (on a multi-GPU box card0 need not be the GPU the context holds)."""
    import subprocess
    try:
        r = subprocess.run(["rocm-smi", "--showbus", "--json"], capture_output=True, text=True, timeout=10)
        cards = json.loads(r.stdout)
    except Exception:
        return None
    want = bus_id.strip().lower()
    for name in sorted(cards):
        v = cards[name]
        if isinstance(v, dict) and any(isinstance(x, str) and x.strip().lower() == want for x in v.values()):
            return name
    return None


def sample_power(step, sync, seconds=1.5, bus_id=None):
    """Package power and core clock while `step` runs back to back for `seconds` (AFTER the timed region, never inside it): `rocm-smi --showpower
    --showclocks --json` sampled every ~40 ms from a thread. Round 5 found the step at the package power limit (1350-1390 W of 1400 W:
    profiles/r05/l_power_probe_net.txt), which is what holds the clock under the fp32 MFMA stream. bus_id (round 6): the PCI bus id of the GPU the
    context holds — the samples are read from THAT card (None when rocm-smi lists no such card); without it, the first card. Returns None when
    rocm-smi is not usable."""
    import re
    import shutil
    import subprocess
    import threading
    if not shutil.which("rocm-smi"):
        return None
    card = None
    if bus_id:
        card = smi_card_of(bus_id)
        if card is None:
            return None
    samples, stop = [], threading.Event()

    def sampler():
        while not stop.is_set():
            try:
                r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5)
                c = json.loads(r.stdout)
                c = c[card if card else sorted(c)[0]]
                m = re.search(r"(\d+)", c.get("sclk clock speed:", ""))
                samples.append((time.time(), float(c.get("Current Socket Graphics Package Power (W)", "nan")), float(m.group(1)) if m else float("nan")))
            except Exception:
                pass
            time.sleep(0.04)

    cap = None
    try:
        r = subprocess.run(["rocm-smi", "--showmaxpower", "--json"], capture_output=True, text=True, timeout=10)
        c = json.loads(r.stdout)
        cap_card = card if card else sorted(c)[0]
        c = c[cap_card]
        cap = float(next(v for k, v in c.items() if "Max" in k and "Power" in k))
    except Exception:
        pass
    th = threading.Thread(target=sampler, daemon=True)
    th.start()
    t0 = time.time()
    n = 0
    while time.time() - t0 < seconds:
        for _ in range(20):
            step()
        sync()
        n += 20
    t1 = time.time()
    stop.set()
    th.join(timeout=6)
    mine = sorted((p, c) for (t, p, c) in samples if t0 + 0.4 <= t <= t1 and p == p)
    if len(mine) < 3:
        return None
    return {"package_w": mine[len(mine) // 2][0], "cap_w": cap, "sclk_mhz": sorted(c for _, c in mine)[len(mine) // 2], "samples": len(mine),
            "card": card if card else "first card listed", "pci_bus_id": bus_id,
            "ms_per_step_while_sampling": round(1000.0 * (t1 - t0) / n, 4),
            "how": "median of rocm-smi samples while the step ran back to back for %.1f s after the timed region" % seconds}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="images per GPU per step")
    ap.add_argument("--alpha", type=float, default=1.0)
    ap.add_argument("--res", type=int, default=224)
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="storage type of activations and pointwise filters (arithmetic is fp32 either way)")
    ap.add_argument("--streams", type=int, default=0,
                    help="0 (default) = 2 in fp32 at batch >= 64 and in bf16 for the big configurations (1.0x224 at batch 512), 1 otherwise. n > 1: pipeline each step over n sub-batches on "
                         "separate HIP streams (mbn_net_set_streams, bit-identical logits): the HBM-bound depthwise kernels of one "
                         "sub-batch overlap the MFMA-bound GEMMs of the other. The untimed steps whose kernels are timed one by one "
                         "(--profile-steps) run on ONE stream, so the per-kernel HIP-event durations behind `roofline` and "
                         "`stages_frac` are not stretched by a concurrent kernel (profiles/LOG.md, streams)")
    ap.add_argument("--dist-backend", default="nccl", help="rehearsal only: 'gloo' lets several ranks share one GPU")
    ap.add_argument("--device-override", type=int, default=-1, help="rehearsal only: every rank uses this device")
    ap.add_argument("--graph", action="store_true", help="replay each step as one hipGraph (mbn_net_set_graph)")
    ap.add_argument("--tune", action="append", default=[], help="key=value passed to mbn_tune_set (experiments; lab build)")
    ap.add_argument("--pw-emul", type=int, default=0, choices=[0, 6, 9],
                    help="fp32 only, OPT-IN: run the pointwise layers (stand-alone: mbn_f32_pw_x6.hip; inside fused blocks 4-11: mbn_f32_dwpw2_x6.hip) with every fp32 operand split exactly "
                         "into three bf16 values, 6 or 9 bf16 MFMA partial products per product, fp32 accumulate; include/mbn.h "
                         "tune key pw_emul). The default line measures the fp32-MFMA kernels and reports this form beside it "
                         "as `pw_emul_alt`")
    ap.add_argument("--no-pw-emul-alt", action="store_true", help="skip the untimed-by-`value` pw_emul=6 pass of the default fp32 line")
    ap.add_argument("--no-configs-alt", action="store_true",
                    help="skip `configs_alt`: after the headline line (N = 1, default workload only) the other single-GPU BASELINE.json "
                         "configs are measured in the same process — configs[4] (bf16 1.0x224 and 0.5x160, batch 512) and configs[1] (batch 1)")
    ap.add_argument("--no-fuse-stem", action="store_true", help="run layers 1-3 as three launches instead of mbn_stem_fused")
    ap.add_argument("--fuse-tail", action="store_true", help="1...4 images: pool and FC as one launch (mbn_pool_fc; off by default: measured slower)")
    ap.add_argument("--no-fuse-resident", action="store_true", help="bf16: the runs of equal small-map blocks (layers 14-23 at 0.5x160) as one fused launch per block instead of one resident launch (A/B)")
    ap.add_argument("--fuse-blocks", type=lambda v: int(v, 0), default=None,
                    help="mask for mbn_net_set_fuse_blocks (bit L = fuse depthwise layer L with pointwise L+1); default: library's")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-images", type=int, default=0, help="images in the CPU baseline sample (0 = auto)")
    ap.add_argument("--no-cpu-variants", action="store_true", help="skip the 1-thread / batch-1 / batch-8 CPU table")
    ap.add_argument("--no-unfused-stages", action="store_true", help="skip the untimed one-launch-per-layer pass")
    ap.add_argument("--cpu-threads", type=int, default=16, help="threads of the CPU baseline (cap; box share is 16)")
    ap.add_argument("--no-profile", action="store_true", help="skip the per-kernel HIP-event pass that follows the timed region")
    ap.add_argument("--profile-steps", type=int, default=5,
                    help="forwards AFTER the timed region whose kernels are timed one by one (HIP event pair per launch, one stream): "
                         "the evidence behind `roofline` and `stages_frac`. Never inside the region `value` comes from")
    ap.add_argument("--no-power", action="store_true", help="skip the untimed ~1.5 s loop that samples package power / core clock with rocm-smi")
    ap.add_argument("--record", default="", help="path of the full record (default: gpurun_out/bench_full_<workload>.json under the repo)")
    args = ap.parse_args(argv)
    if args.streams <= 0:
        # bf16 (round 6, profiles/r06/v_bf16_streams.txt): two streams +3.2 % at 1.0x224 batch 512 (282-284 k -> 291-293 k), equal at 0.5x160 (its step is
        # 0.5 ms of short launches), three lose 8-20 %: two from 1e7 batch * alpha^2 * res^2 up (1.0x224 at batch 512: 2.6e7; 0.5x160: 3.3e6)
        heavy = args.dtype == "f32" or args.batch * args.alpha * args.alpha * args.res * args.res >= 1e7
        args.streams = 2 if (heavy and args.batch >= 64 and not args.graph) else 1
    args.profile_steps = max(1, args.profile_steps)
    return args


def visible_gpus():
    """GPUs this node shows, counted in a short-lived CHILD process so that the launcher itself never holds a GPU runtime (when torch
    cannot use amdsmi, torch.cuda.device_count() falls back to hipGetDeviceCount and initialises HIP: ADVICE r4). -1 = could not tell."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True, timeout=300)
        return int(r.stdout.strip().splitlines()[-1])
    except Exception:
        return -1


def self_launch(args, argv):
    """`python bench.py --gpus N` with N > 1 and no rank environment: start the N ranks ourselves, the way the driver's own
    command line does (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ...`), as a CHILD process — never os.exec*,
    and this process never touches HIP or torch.cuda (the device count comes from another child). The rendezvous is
    `--standalone --local-addr 127.0.0.1`: torchrun picks a free port itself (no bind-then-close race). Rank 0's JSON line goes to our
    stdout through the inherited descriptor; the child's return code is ours. A rank count above the visible devices is refused here
    with a plain message instead of N ranks failing in hipSetDevice (the reference has one device, MobileNet.c:155)."""
    import subprocess
    if args.device_override < 0:
        have = visible_gpus()
        if have < args.gpus:
            sys.stderr.write("bench.py: --gpus %d but this node shows %d GPU(s) (MBN_ENODEVICE); for a rehearsal on one card use "
                             "--dist-backend gloo --device-override 0\n" % (args.gpus, have))
            return 19       # ENODEV
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           "--nproc-per-node", str(args.gpus), os.path.abspath(__file__)] + list(argv)
    sys.stdout.flush()
    return subprocess.run(cmd, env=env).returncode


LINE_LIMIT = 4096          # bytes of the printed line (the 20 KB line of round 4 did not fit the driver's 8 KB stdout tail)


def _r(x, n=4):
    return None if x is None else round(float(x), n)


def compact_line(out, record_path=None):
    """The ONE line bench.py prints, from run_one's full record: the contract's keys, `roofline` and `cpu_baseline` with scalar members
    only (the driver's parser keeps scalars), flat [ms, frac_hbm, frac_mfma] triples per stage, [value, frac, parity_ok] per alternative
    configuration. Everything else stays in the side file `full_record` names. Pure function: tests/test_host_cpu.py feeds it a
    synthetic record and checks length and parse."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    line = {k: out.get(k) for k in keep}
    line["value"], line["ms_per_step"] = _r(out.get("value"), 1), _r(out.get("ms_per_step"), 4)
    cfg = out.get("config") or {}
    line["config"] = {k: cfg[k] for k in ("workload", "global_batch", "per_gpu_batch", "parallelism", "streams", "device", "pw_emul") if k in cfg}
    rf = out.get("roofline")
    if rf:
        line["roofline"] = {k: (v if isinstance(v, (str, int)) or v is None else _r(v, 5 if "ms" in k else 4)) for k, v in rf.items()
                            if not isinstance(v, (dict, list)) and k not in ("note",)}
        if isinstance(line["roofline"].get("kernel"), str):
            line["roofline"]["kernel"] = line["roofline"]["kernel"][:110]
    sf = {}
    for name, st in (out.get("stages") or {}).items():
        sf[name] = [_r(st["ms"]), _r(st["frac_hbm"]), _r(st["frac_mfma"])]
    for name, st in (((out.get("unfused_stages") or {}).get("stages")) or {}).items():
        if name in ("depthwise", "pointwise", "conv1"):
            sf["unfused_%s_x%d" % (name, st["launches"])] = [_r(st["ms"]), _r(st["frac_hbm"]), _r(st["frac_mfma"])]
    if sf:
        line["stages_frac"] = sf
        line["stages_frac_cols"] = "ms,frac_of_8TBps,frac_of_mfma_peak"
        line["profiled_steps"] = out.get("profiled_steps")
    cb = out.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {"value": _r(cb["value"], 2), "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
                                "sample": str(cb.get("sample", ""))[:110]}
    pc = out.get("parity_check")
    if pc:
        line["parity_check"] = {"ok": pc["ok"], "max_rel_err": float("%.3e" % pc["max_rel_err"]), "tolerance": pc["tolerance"], "images": pc["images"]}
    if out.get("step_ms"):
        line["step_ms"] = [out["step_ms"]["median"], out["step_ms"]["p10"], out["step_ms"]["p90"]]
    ca = out.get("configs_alt")
    if ca:
        line["configs_alt"] = {name: [_r(c.get("value"), 1), _r((c.get("roofline") or {}).get("frac")), (c.get("parity_check") or {}).get("ok"),
                                      (c.get("power") or {}).get("package_w")] for name, c in ca.items()}
        line["configs_alt_cols"] = "images_per_sec,roofline_frac(bf16: pointwise of 8 TB/s; f32: of mfma peak),parity_ok,package_w"
    alt = out.get("pw_emul_alt")
    if alt:
        line["pw_emul_alt"] = [_r(alt["value"], 1), (alt.get("parity_check") or {}).get("ok")]
    for k in ("ranks", "ranks_cols", "collective_world_size", "backend"):
        if k in out:
            line[k] = out[k]
    if record_path:
        line["full_record"] = record_path
    text = json.dumps(line, separators=(",", ":"))
    if len(text) > LINE_LIMIT:                      # never print a line the driver cannot keep: drop the optional tables, largest first
        for k in ("stages_frac", "configs_alt", "ranks", "step_ms", "pw_emul_alt"):
            if k in line:
                line.pop(k)
                line.pop(k + "_cols", None)
                line["dropped"] = line.get("dropped", []) + [k]
                text = json.dumps(line, separators=(",", ":"))
                if len(text) <= LINE_LIMIT:
                    break
    return text


def write_record(args, out):
    """The full record (everything run_one measured) as a side file; returns the path the line names (relative to the repo when inside it)."""
    name = args.record or os.path.join(ROOT, "gpurun_out", "bench_full_%s_a%g_r%d_b%d_n%d.json" % (args.dtype, args.alpha, args.res, args.batch, out.get("n_gpus", 1)))
    try:
        os.makedirs(os.path.dirname(os.path.abspath(name)), exist_ok=True)
        with open(name, "w") as f:
            json.dump(out, f, indent=1)
    except OSError:
        fd, name = tempfile.mkstemp(prefix="mbn_bench_full_", suffix=".json")
        with os.fdopen(fd, "w") as f:
            json.dump(out, f, indent=1)
    return os.path.relpath(name, ROOT) if os.path.abspath(name).startswith(ROOT + os.sep) else name


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args, sys.argv[1:]))
    from mbn_amd import import_package
    pkg = import_package()
    from mbn_amd_pkg import dist as mdist      # the protocol rehearsed on gloo in tests/test_dist_cpu.py
    rank, local_rank, world = mdist.env_rank_world()
    if world != args.gpus:
        args.gpus = world

    import torch   # device plumbing only: RCCL broadcast, barrier, device-wide synchronize

    lib = pkg.load()          # raises if the HIP extension is missing: no fallback
    for kv in args.tune:
        k, v = kv.split("=")
        assert lib.mbn_tune_set(k.encode(), int(v)) == 0, kv

    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X (torch.cuda.is_available() is False)")
    if args.device_override >= 0:
        local_rank = args.device_override
    if local_rank >= torch.cuda.device_count():
        sys.exit("bench.py: rank %d wants GPU %d but this node shows %d (MBN_ENODEVICE)" % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    mdist.init(args.dist_backend, dev)         # nccl == RCCL on ROCm; no-op for one process
    env = {"pkg": pkg, "mdist": mdist, "torch": torch, "lib": lib, "rank": rank, "local_rank": local_rank, "world": world, "dev": dev}

    out = run_one(args, env)
    fail = None
    if rank == 0:
        headline = (world == 1 and args.dtype == "f32" and args.alpha == 1.0 and args.res == 224 and args.batch == 256
                    and not args.pw_emul and not args.graph and not args.tune)
        if headline and not args.no_configs_alt:
            out["configs_alt"] = configs_alt(args, env)
        if "parity_check" in out and not out["parity_check"]["ok"]:
            fail = "parity check failed: max rel err %.3e > %.1e" % (out["parity_check"]["max_rel_err"], out["parity_check"]["tolerance"])
        for name, c in (out.get("configs_alt") or {}).items():
            pc = c.get("parity_check")
            if pc and not pc["ok"] and not fail:
                fail = "configs_alt %s: parity check failed: max rel err %.3e > %.1e" % (name, pc["max_rel_err"], pc["tolerance"])
        if out.get("ranks_error") and not fail:
            fail = out["ranks_error"]
        sys.stderr.flush()
        print(compact_line(out, write_record(args, out)))       # the LAST line of stdout
        sys.stdout.flush()
    mdist.shutdown()
    if fail:
        sys.exit(fail)


def configs_alt(args, env):
    """The other single-GPU configurations BASELINE.json names, measured after the headline line in the same process and
    printed inside it (the driver's record should hold them): configs[4] = bf16 storage at 1.0x224 and
    0.5x160, batch 512; configs[1] = batch 1 (latency). Each entry is a full record of its own workload (value, ms_per_step,
    stages, roofline with the committed PMC traffic, parity_check against the oracle) minus the CPU tables; the printed line
    carries [value, roofline frac, parity ok] of each, the side file all of it."""
    import copy
    res = {}
    for name, kw in (("bf16_1.0x224_b512", dict(dtype="bf16", alpha=1.0, res=224, batch=512, streams=2)),
                     ("bf16_0.5x160_b512", dict(dtype="bf16", alpha=0.5, res=160, batch=512, streams=1)),
                     ("f32_1.0x224_b1", dict(dtype="f32", alpha=1.0, res=224, batch=1, streams=1))):
        a = copy.copy(args)
        for k, v in kw.items():
            setattr(a, k, v)
        a.steps, a.warmup = (200, 20) if a.batch == 1 else (60, 8)
        a.profile_steps = 5                      # after the timed region, like the headline's
        a.no_cpu_variants = a.no_unfused_stages = a.no_pw_emul_alt = True
        a.cpu_images = 1 if a.batch == 1 else 8
        o = run_one(a, env)
        keep = ("value", "unit", "ms_per_step", "steps", "warmup", "dtype", "config", "roofline", "stages", "sum_kernel_ms",
                "profiled_steps", "event_overhead_us", "step_ms", "parity_check", "power", "layers")
        res[name] = {k: o[k] for k in keep if k in o}
    return res


def run_one(args, env):
    """One workload: build the net, time `steps` forwards, return the bench line as a dict (rank 0; None elsewhere)."""
    pkg, mdist, torch, lib = env["pkg"], env["mdist"], env["torch"], env["lib"]
    rank, local_rank, world, dev = env["rank"], env["local_rank"], env["world"], env["dev"]
    import numpy as np
    if args.pw_emul and args.dtype == "f32":
        assert lib.mbn_tune_set(b"pw_emul", args.pw_emul) == 0
        assert lib.mbn_tune_set(b"pw_emul_static", 1) == 0      # the weights are uploaded once: filter images are split once

    # ---- parameters: rank 0 writes a synthetic Keras-layout .h5 and reads it back through the real loader
    plan = pkg.plan_build(args.alpha, args.res, 1000, lib=lib)
    blob_t = torch.empty(plan.blob_floats, dtype=torch.float32, device=dev)
    if rank == 0:
        fd, path = tempfile.mkstemp(prefix="mbn_bench_", suffix=".h5")      # O_EXCL, unpredictable name
        os.close(fd)
        pkg.synthetic_h5(path, alpha=args.alpha, classes=1000, seed=0xC0FFEE, lib=lib)
        hw = pkg.HostWeights(path, alpha=args.alpha, res=args.res, lib=lib)
        os.remove(path)
        assert hw.plan.blob_floats == plan.blob_floats
        blob_t.copy_(torch.from_numpy(hw.blob))
        hw.free()
    if args.dist_backend == "gloo" and world > 1:          # rehearsal: gloo moves host tensors
        host = blob_t.cpu()
        mdist.broadcast_blob(host, 0)
        blob_t.copy_(host)
    else:
        mdist.broadcast_blob(blob_t, 0)        # the one collective of the path: ~17 MB over xGMI, off the timed path
    torch.cuda.synchronize()

    ctx = pkg.Context(local_rank)
    net = pkg.Net(ctx, plan, blob_t.data_ptr(), args.batch)
    if args.no_fuse_stem:
        net.set_fuse_stem(False)
    if args.fuse_tail:
        net.set_fuse_tail(True)
    if args.no_fuse_resident:
        net.set_fuse_resident(False)
    if args.fuse_blocks is not None:
        net.set_fuse_blocks(args.fuse_blocks)
    if args.graph:
        net.set_graph(True)
    if args.streams > 1 and args.batch >= 2 * args.streams:
        net.set_streams(args.streams, free_running=True)   # the input batch is resident before timing starts
    bf16 = args.dtype == "bf16"
    if bf16:
        net.set_dtype(pkg.DT_BF16)
    act_bytes = 2.0 if bf16 else 4.0
    mfma_peak = 2500.0 if bf16 else MFMA_F32_PEAK_TFLOPS      # dense bf16 MFMA peak ~2.5 PFLOP/s (MI355X_MICROARCH.md)

    # ---- synthetic input, U[-1,1) (Keras x/127.5-1 range), generated on the host then uploaded once
    rng = np.random.default_rng(0xC0FFEE + rank)
    imgs = rng.random((args.batch, args.res, args.res, 3), dtype=np.float32) * 2.0 - 1.0
    d_in = ctx.alloc(imgs.nbytes)
    h0 = time.perf_counter()
    d_in.upload(imgs)                      # blocking H2D of one batch from pageable host memory (PCIe), not timed as a step
    h2d_ms = 1000.0 * (time.perf_counter() - h0)
    head = imgs[:64].copy()                # the CPU baseline / parity check runs the oracle on the first images of THIS batch
    del imgs
    d_out = ctx.alloc(args.batch * 1000 * 4)

    barrier = mdist.barrier

    for _ in range(args.warmup):
        net.forward(d_in.ptr, d_out.ptr, args.batch)
    ctx.sync()

    profile = not args.no_profile and not args.graph    # per-kernel events cannot be read back from inside a graph
    multi = args.streams > 1 and args.batch >= 2 * args.streams
    # launches of one single-stream pass, in order (mbn_net_launches): the fused stem (layers 1-3, mbn_stem_fused), fused
    # depthwise->pointwise blocks (mbn_dwpw_fused) and single layers
    launches = [list(range(f - 1, f - 1 + c)) for f, c in net.launches(args.batch)]
    n_launch = len(launches)

    # ---- the timed region: EXACTLY K forwards in the default configuration, barrier + device-wide sync on both sides; one HIP event
    # between steps (no sync) for the per-step median. No per-kernel events, no single-stream steps: `value` is one configuration.
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.mark()
    for step in range(args.steps):
        net.forward(d_in.ptr, d_out.ptr, args.batch)
        ctx.mark()
    torch.cuda.synchronize()
    barrier()
    t1 = time.perf_counter()
    elapsed_rank = t1 - t0
    step_ms = np.asarray(ctx.marks_read(args.steps + 1), dtype=np.float64)
    elapsed = mdist.max_over_ranks(elapsed_rank, "cpu" if args.dist_backend == "gloo" else dev)

    logits = d_out.download((args.batch, 1000), np.float32)      # what the timed steps left on the device
    if not np.isfinite(logits).all():
        sys.exit("non-finite logits")

    # ---- after the region: `profile_steps` forwards on ONE stream with a HIP event pair around every launch (roofline, stages_frac),
    # the pw_gemm launches also accumulating the core clock they held (tune key pw_clock)
    layer_ms = None
    held_ghz, held_launches = None, 0
    n_prof = args.profile_steps
    if profile:
        if multi:
            net.set_streams(1)
        net.forward(d_in.ptr, d_out.ptr, args.batch)              # one untimed pass in the single-stream configuration first
        ctx.sync()
        clock_ok = lib.mbn_tune_set(b"pw_clock", 1) == 0
        if clock_ok:
            ctx.pw_clock(reset=True)
        ctx.profile_begin(n_launch * n_prof)
        for _ in range(n_prof):
            net.forward(d_in.ptr, d_out.ptr, args.batch)
        ms = ctx.profile_end(n_launch * n_prof)
        if clock_ok:
            held_ghz, held_launches = ctx.pw_clock(reset=True)
            lib.mbn_tune_set(b"pw_clock", 0)
        layer_ms_raw = np.asarray(ms, dtype=np.float64).reshape(n_prof, n_launch).mean(axis=0)
        # What the event pair itself adds to every reading (stand-alone depthwise read 34.5 us by events, 31.2 us
        # in the rocprofv3 kernel trace): a pair recorded around NOTHING on the same stream reads the marker-to-marker time that is
        # also inside every pair with a kernel between them. Measured here and subtracted from every launch.
        ov_null_us = ctx.profile_null_us(False)
        ov_kernel_us = ctx.profile_null_us(True)
        layer_ms = np.maximum(layer_ms_raw - ov_null_us * 1e-3, 1e-6)
        if multi:
            net.set_streams(args.streams, free_running=True)

    # ---- the one-launch-per-layer pass (north_star's per-stage targets), UNTIMED, right behind the per-kernel pass of the default configuration: the
    # chip is in the state those launches were timed in (round 6: it used to run behind the 1.5 s power sampler and the pw_emul pass, on a chip that
    # had been loaded for seconds — the same GEMM launch held 2.06-2.13 GHz there against 2.24-2.28 in the default pass)
    unfused_raw = None
    if world == 1 and profile and not args.no_unfused_stages:
        if multi:
            net.set_streams(1)
        net.set_fuse_stem(False)
        net.set_fuse_blocks(0)
        net.set_fuse_tail(False)
        ul = [list(range(f - 1, f - 1 + c)) for f, c in net.launches(args.batch)]
        for _ in range(2):
            net.forward(d_in.ptr, d_out.ptr, args.batch)
        reps = 5
        # round 6 (the same pw_gemm launch read 8 % longer in this pass than in the default one, unexplained): the core
        # clock held inside the GEMM launches of THIS pass, read the same way as the default pass's `held_clock_ghz`
        u_clock_on = (not bf16) and lib.mbn_tune_set(b"pw_clock", 1) == 0
        if u_clock_on:
            ctx.pw_clock(reset=True)
        ctx.profile_begin(len(ul) * reps)
        for _ in range(reps):
            net.forward(d_in.ptr, d_out.ptr, args.batch)
        ums = np.asarray(ctx.profile_end(len(ul) * reps), dtype=np.float64).reshape(reps, len(ul)).mean(axis=0)
        u_ghz = None
        if u_clock_on:
            g_, n_ = ctx.pw_clock(reset=True)
            lib.mbn_tune_set(b"pw_clock", 0)
            u_ghz = round(g_, 3) if n_ > 0 else None
        ums = np.maximum(ums - ov_null_us * 1e-3, 1e-6)
        unfused_raw = (ul, ums, u_ghz, reps)
        net.set_fuse_stem(not args.no_fuse_stem)
        net.set_fuse_tail(args.fuse_tail)
        if args.fuse_blocks is not None:
            net.set_fuse_blocks(args.fuse_blocks)
        else:
            net.reset_fuse_blocks()          # back to the default WITH its default-only rules (an explicit mask would switch them off)
        if multi:
            net.set_streams(args.streams, free_running=True)

    power = None
    if world == 1 and not args.no_power and not args.graph:
        try:
            my_bus = ctx.pci_bus_id()
        except Exception:
            my_bus = None
        power = sample_power(lambda: net.forward(d_in.ptr, d_out.ptr, args.batch), ctx.sync, 1.0 if args.batch == 1 else 1.5, my_bus)

    # ---- the opt-in split form of the pointwise GEMM beside the default line (N = 1, fp32): same net, same buffers, same
    # stream configuration, 3 warm-up + 10 timed steps without per-kernel events. Never part of `value`.
    alt = None
    if world == 1 and not bf16 and not args.pw_emul and not args.no_pw_emul_alt and not args.graph:
        assert lib.mbn_tune_set(b"pw_emul", 6) == 0
        assert lib.mbn_tune_set(b"pw_emul_static", 1) == 0      # the weights are uploaded once: filter images are split once
        for _ in range(3):
            net.forward(d_in.ptr, d_out.ptr, args.batch)
        torch.cuda.synchronize()
        a0 = time.perf_counter()
        for _ in range(10):
            net.forward(d_in.ptr, d_out.ptr, args.batch)
        torch.cuda.synchronize()
        a1 = time.perf_counter()
        assert lib.mbn_tune_set(b"pw_emul", 0) == 0
        assert lib.mbn_tune_set(b"pw_emul_static", 0) == 0
        alt_logits = d_out.download((args.batch, 1000), np.float32)
        alt = {"pw_emul": 6, "value": args.batch * 10 / (a1 - a0), "unit": "images/sec", "ms_per_step": 100.0 * (a1 - a0),
               "steps": 10, "warmup": 3,
               "max_rel_diff_to_default_logits": float(np.abs(alt_logits.astype(np.float64) - logits).max() / max(float(np.abs(logits).max()), 1e-6)),
               "what": "mbn_tune_set(\"pw_emul\", 6): the pointwise layers 13-27 on mbn_f32_pw_x6.hip and the pointwise halves of the "
                       "fused blocks 4-11 on mbn_f32_dwpw2_x6.hip, the pointwise phase of the fused stem likewise - fp32 in/out, every "
                       "operand split exactly into three bf16 values, 6 bf16 MFMA partial products per fp32 product (the dropped "
                       "three are < 2^-24 of it), fp32 accumulate; measured error against float64 <= the fp32 MFMA kernel's "
                       "(profiles/r02/m_pw_emul.txt). Opt-in: `value` above is the fp32-MFMA path"}

    # ---- N > 1: every rank reports which card it held and what it did (off the timed path); rank 0 prints them
    ranks_info = None
    if world > 1 or mdist._force_pg():
        import torch.distributed as tdist
        # round 6: what each card held while ALL ranks ran — the >= 7x target at 8 GPUs is set by the slowest card of a chassis in
        # which every card sits at its power cap. After the timed region and behind a barrier every rank runs its step back to back for ~1 s, sampling
        # ITS card (chosen by PCI bus id) with rocm-smi and the core clock held inside the GEMM launches (mbn_pw_clock_read); None where unavailable.
        my_bus = ctx.pci_bus_id()
        rank_w = rank_sclk = rank_ghz = None
        if not args.no_power and not args.graph:
            try:
                tdist.barrier()
                clock_on = (not bf16) and lib.mbn_tune_set(b"pw_clock", 1) == 0
                if clock_on:
                    ctx.pw_clock(reset=True)
                pw = sample_power(lambda: net.forward(d_in.ptr, d_out.ptr, args.batch), ctx.sync, 1.0, my_bus)
                if clock_on:
                    g, nl = ctx.pw_clock(reset=True)
                    lib.mbn_tune_set(b"pw_clock", 0)
                    rank_ghz = round(g, 3) if nl > 0 else None
                if pw:
                    rank_w, rank_sclk = pw["package_w"], pw["sclk_mhz"]
            except Exception:
                pass
        mine = [rank, local_rank, my_bus, args.batch, round(elapsed_rank, 6), rank_w, rank_sclk, rank_ghz]
        ranks_info = [None] * tdist.get_world_size()
        tdist.all_gather_object(ranks_info, mine)

    if rank == 0:
        total_images = args.batch * world * args.steps
        out = {
            "metric": "images/sec MobileNet-V1 1.0x224 fp32, batch 256; per-stage HBM GB/s vs roofline",
            "value": total_images / elapsed,
            "unit": "images/sec",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1000.0 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": "MobileNet-V1 %.2gx%d %s, batch %d per GPU, 29 layers, 1000 classes%s"
                                   % (args.alpha, args.res, "bf16" if bf16 else "fp32", args.batch,
                                      " (BASELINE.json configs[2])" if (not bf16 and args.alpha == 1.0 and args.res == 224
                                                                         and args.batch == 256) else ""),
                       "global_batch": args.batch * world, "per_gpu_batch": args.batch,
                       "parallelism": "batch-sharded x%d, weights broadcast once over RCCL" % world,
                       "streams": args.streams if multi else 1,
                       "streams_note": ("sub-batches of %d images on %d forked streams in the timed region; the %d untimed steps whose "
                                        "kernels are timed one by one (stages, roofline) run on one stream" % (args.batch // args.streams, args.streams,
                                                                                                            n_prof if profile else 0)) if multi else None,
                       "device": ctx.name()},
        }
        if layer_ms is not None:
            stages, per_layer, stage_of = stage_table(plan, pkg, launches, layer_ms, args.batch, act_bytes, mfma_peak)
            # dominant kernel: the stand-alone pointwise GEMMs; at 1..4 images every block is one fused launch, so the fused
            # block launches are the dominant kernel and their pointwise layers carry the flops
            # ... and in bf16 at 0.5x160 since round 5 (every block but the two 5 x 5 ones is a fused launch): the dominant kernel is whichever of the
            # two groups takes more of the step
            dom = "pointwise" if "pointwise" in stages else "block_fused"
            if bf16 and "pointwise" in stages and "block_fused" in stages and stages["pointwise"]["launches"] < 4 and \
                    stages["block_fused"]["ms"] > stages["pointwise"]["ms"]:
                dom = "block_fused"       # 0.5x160: two stand-alone pointwise launches (the 5 x 5 layers) are not what the step spends its time in
            pw = stages[dom]
            pw_idx = [launches[j][-1] for j in range(n_launch) if stage_of[j] == dom]
            flops_per_launch = sum(layer_work(plan.layer[i], args.batch, pkg, act_bytes)[0] for i in pw_idx) / len(pw_idx)
            bytes_per_launch = sum(layer_work(plan.layer[i], args.batch, pkg, act_bytes)[1] for i in pw_idx) / len(pw_idx)
            if dom == "block_fused":          # a fused launch moves its depthwise layer's input and its pointwise layer's output (+ both filters)
                bytes_per_launch = pw["GBps"] * pw["ms"] * 1e6 / len(pw_idx)
            avg_ms = pw["ms"] / len(pw_idx)           # per LAYER (= per launch when --streams 1)
            traffic, traffic_src = load_traffic([i + 1 for i in pw_idx], "bf16_%gx%d" % (args.alpha, args.res) if bf16 else
                                                ("f32_pw_emul%d" % args.pw_emul if args.pw_emul else "f32"))
            if dom == "block_fused":
                traffic, traffic_src = None, None     # the committed PMC passes are per stand-alone layer
            if bf16:       # ridge ~312 flop/B: every pointwise layer is HBM-bound in bf16
                out["roofline"] = {
                    "kernel": ("bf16 pointwise GEMMs: pw_stream_bf16 (K >= 512 on 16x16x32 MFMAs; K <= 256 narrow layers) and pw_gemm<bf16> (%d pointwise 1x1 conv launches per step)" if dom == "pointwise" else
                               "dwpw2_bf16: fused depthwise->pointwise blocks (%d launches per step; input of the depthwise layer + output of the pointwise layer + filters)") % len(pw_idx),
                    "bound": "hbm", "achieved": round(bytes_per_launch / avg_ms / 1e6, 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(bytes_per_launch / avg_ms / 1e6 / HBM_PEAK_GBS, 4),
                    "traffic": traffic, "traffic_source": traffic_src, "avg_launch_ms": round(avg_ms, 5),
                    "algorithmic_bytes_per_launch": bytes_per_launch,
                }
            else:
                out["roofline"] = {
                    "kernel": ("pw_gemm<float> (%d pointwise 1x1 conv launches per step)" if dom == "pointwise" else
                               "dwpw_small_f32 (%d fused depthwise->pointwise launches per step; flops/bytes of their pointwise layers)") % len(pw_idx),
                    "bound": "mfma", "achieved": round(flops_per_launch / avg_ms / 1e9, 2), "peak": MFMA_F32_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(flops_per_launch / avg_ms / 1e9 / MFMA_F32_PEAK_TFLOPS, 4),
                    "traffic": traffic, "traffic_source": traffic_src, "avg_launch_ms": round(avg_ms, 5),
                    "algorithmic_flops_per_launch": flops_per_launch,
                    "algorithmic_bytes_per_launch": bytes_per_launch,
                }
            out["roofline"]["launches_per_step"] = len(pw_idx)
            if held_ghz and dom == "pointwise" and not bf16 and not args.pw_emul:
                # the clock the chip held inside these very launches (DVFS differs by box: 2.1-2.4 GHz); the peak is quoted at 2.4 GHz
                out["roofline"]["held_clock_ghz"] = round(held_ghz, 3)
                out["roofline"]["frac_at_held_clock"] = round(out["roofline"]["achieved"] / (MFMA_F32_PEAK_TFLOPS * held_ghz / 2.4), 4)
                out["roofline"]["held_clock_launches"] = int(held_launches)
            out["stages"] = stages
            out["layers"] = per_layer
            out["sum_kernel_ms"] = round(float(layer_ms.sum()), 4)
            out["profiled_steps"] = n_prof
            out["profiled_steps_where"] = "after the timed region, one stream"
            out["event_overhead_us"] = {"subtracted_per_launch": round(ov_null_us, 3), "empty_pair": round(ov_null_us, 3),
                                        "pair_around_empty_kernel": round(ov_kernel_us, 3),
                                        "sum_kernel_ms_raw": round(float(layer_ms_raw.sum()), 4),
                                        "how": "median of 200 event pairs recorded like a layer call's (mbn_profile_null) after the timed "
                                               "region; every per-launch time in stages/layers/roofline is the raw pair reading minus empty_pair"}
        if ranks_info is not None:
            import torch.distributed as tdist
            out["ranks"] = ranks_info
            out["ranks_cols"] = "rank,device_ordinal,pci_bus_id,images_per_step,seconds_for_the_K_steps,package_w,sclk_mhz,held_clock_ghz (all ranks loaded, after the timed region)"
            out["collective_world_size"] = tdist.get_world_size()
            out["backend"] = tdist.get_backend()
            buses = [r[2] for r in ranks_info]
            if len(set(buses)) != len(buses) and args.device_override < 0:
                out["ranks_error"] = "two ranks held the same GPU (PCI bus ids %s) without --device-override" % buses
            if len(ranks_info) != world:
                out["ranks_error"] = "the process group has %d ranks, the line claims %d" % (len(ranks_info), world)
        if power is not None:
            out["power"] = power
            if "roofline" in out:       # scalars the driver's record keeps: the step runs at the package power limit, which sets the held clock
                out["roofline"]["package_power_w"] = power["package_w"]
                out["roofline"]["power_cap_w"] = power["cap_w"]
                if power["package_w"] and power.get("ms_per_step_while_sampling"):      # at the limit, speed is energy per image
                    out["roofline"]["images_per_joule"] = round(args.batch / (power["ms_per_step_while_sampling"] * 1e-3) / power["package_w"], 2)
        out["h2d_ms_per_batch"] = round(h2d_ms, 3)   # PCIe-inclusive rate = batch / (ms_per_step + this)
        if step_ms.size:
            out["step_ms"] = {"median": round(float(np.median(step_ms)), 4), "p10": round(float(np.percentile(step_ms, 10)), 4),
                              "p90": round(float(np.percentile(step_ms, 90)), 4), "n": int(step_ms.size),
                              "how": "HIP event between steps on the kernels' stream, no sync inside the timed region"}
        if world == 1 and profile and not args.no_unfused_stages:
            # all 13 depthwise + 13 pointwise stages as their own launches (the metric names per-stage numbers; the timed
            # configuration above folds layers 1-11 into fused launches). UNTIMED: outside the region `value` comes from.
            ul, ums, u_ghz, reps = unfused_raw
            ust, ulayers, _ = stage_table(plan, pkg, ul, ums, args.batch, act_bytes, mfma_peak)
            out["unfused_stages"] = {"note": "untimed: %d forwards with one launch per layer (mbn_net_set_fuse_stem(0), "
                                             "mbn_net_set_fuse_blocks(0)); same batch, same buffers" % reps,
                                     "stages": ust, "layers": ulayers, "sum_kernel_ms": round(float(ums.sum()), 4), "held_clock_ghz": u_ghz}
            if "roofline" in out:       # flat scalars the driver's record keeps: north_star's two stage targets, and the fused launches
                if "depthwise" in ust:
                    out["roofline"]["dw_x%d_frac_hbm" % ust["depthwise"]["launches"]] = ust["depthwise"]["frac_hbm"]
                if "pointwise" in ust:
                    out["roofline"]["pw_x%d_frac_mfma" % ust["pointwise"]["launches"]] = ust["pointwise"]["frac_mfma"]
                    if u_ghz:
                        out["roofline"]["pw_x%d_held_clock_ghz" % ust["pointwise"]["launches"]] = u_ghz
                if "block_fused" in stages:
                    out["roofline"]["blocks_ms"] = stages["block_fused"]["ms"]
                if "stem_fused" in stages:
                    out["roofline"]["stem_ms"] = stages["stem_fused"]["ms"]
        if alt is not None:
            out["pw_emul_alt"] = alt
        if args.pw_emul and not bf16:
            out["config"]["pw_emul"] = args.pw_emul
            out["config"]["arithmetic"] = ("pointwise layers 13-27 and the pointwise halves of the fused blocks 4-11: fp32 operands split exactly "
                                           "into three bf16 values, %d bf16 MFMA partial products per product, fp32 accumulate "
                                           "(mbn_f32_pw_x6.hip, mbn_f32_dwpw2_x6.hip) and the pointwise phase of the fused stem; conv1, depthwise, pool, FC unchanged" % args.pw_emul)
            if "roofline" in out and out["roofline"].get("bound") == "mfma":
                r = out["roofline"]
                r["kernel"] = "pw_gemm_x (%d bf16 partial products per fp32 product) " % args.pw_emul + r["kernel"]
                r["fp32_equivalent_tflops"] = r["achieved"]
                r["achieved"] = round(r["achieved"] * args.pw_emul, 2)
                r["peak"] = 2500.0
                r["frac"] = round(r["achieved"] / 2500.0, 4)
                r["note"] = "achieved = %d x the algorithmic fp32 flops (the bf16 MFMA work actually issued) against the dense bf16 peak" % args.pw_emul
        if world == 1 and not args.no_cpu_baseline:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import oracle as orc   # cpu_baseline leg: the oracle is the thing timed here, never the product path
            oplan = orc.plan_build(args.alpha, args.res, 1000)
            cores = min(orc.num_threads(), args.cpu_threads)      # the one-GPU box's CPU share is 16 cores
            n_img = min(args.cpu_images or max(2, 4 * cores), head.shape[0], args.batch)   # ~1-2 s on 16 cores
            blob = blob_t.cpu().numpy()
            sample = head[:n_img]                                 # the first n images of the timed batch
            c0 = time.perf_counter()
            ref, _ = orc.net_forward(oplan, blob, sample, threads=cores, bf16=bf16)
            c1 = time.perf_counter()
            out["cpu_baseline"] = {"value": n_img / (c1 - c0), "unit": "images/sec", "cores": cores, "kind": "port",
                                   "sample": "first %d images of the timed batch, same network/weights, C restatement of "
                                             "kernel.cl semantics in fp32 NHWC (oracle/mbn_oracle.c), OpenMP over output "
                                             "pixels, %.1f s" % (n_img, c1 - c0)}
            # ---- parity of the timed workload itself: what the steps left in d_out vs the oracle, same images
            ref = np.asarray(ref, dtype=np.float64).reshape(n_img, 1000)
            got = logits[:n_img].astype(np.float64)
            scale = max(float(np.abs(ref).max()), 1e-6)
            err = float(np.abs(got - ref).max()) / scale
            tol = 2e-2 if bf16 else 1e-3
            if alt is not None:
                aerr = float(np.abs(alt_logits[:n_img].astype(np.float64) - ref).max()) / scale
                alt["parity_check"] = {"images": n_img, "max_rel_err": aerr, "tolerance": tol, "ok": bool(aerr <= tol),
                                       "argmax_agree": int((alt_logits[:n_img].argmax(1) == ref.argmax(1)).sum())}
            out["parity_check"] = {"images": n_img, "max_rel_err": err, "tolerance": tol, "ok": bool(err <= tol),
                                   "argmax_agree": int((got.argmax(1) == ref.argmax(1)).sum()),
                                   "against": "oracle/mbn_oracle.c F32 mode%s, logits of the first %d images of the timed batch"
                                              % (" (bf16 storage emulated)" if bf16 else "", n_img)}
            if not args.no_cpu_variants:
                # SURVEY.md §8d / BASELINE.md §3: single thread (faithful to the reference's one-work-item-at-a-time
                # semantics) and all cores, batch 1 and 8, median of 5
                def med5(nb, th):
                    ts = []
                    for _ in range(5):
                        a0 = time.perf_counter()
                        orc.net_forward(oplan, blob, head[:nb], threads=th, bf16=bf16)
                        ts.append(time.perf_counter() - a0)
                    return float(np.median(ts))
                var = {}
                for th in (1, cores):
                    for nb in (1, 8):
                        t = med5(nb, th)
                        var["threads%d_batch%d" % (th, nb)] = {"ms_per_image": round(1000.0 * t / nb, 2),
                                                               "images_per_sec": round(nb / t, 2)}
                out["cpu_baseline"]["variants"] = var
                out["cpu_baseline"]["variants_how"] = "median of 5, host cores of this box: %d used of %d visible" % (
                    cores, os.cpu_count() or cores)
        if world > 1 and not args.no_cpu_baseline:
            # N > 1: no cpu_baseline leg (contract: rank 0 at N = 1 only), but the logits rank 0's timed steps left on its GPU are
            # still checked against the oracle on the first 8 images of its shard (~0.3 s of host time after the timed region)
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import oracle as orc   # checker only
            n_img = min(8, head.shape[0], args.batch)
            ref, _ = orc.net_forward(orc.plan_build(args.alpha, args.res, 1000), blob_t.cpu().numpy(), head[:n_img],
                                     threads=min(orc.num_threads(), args.cpu_threads), bf16=bf16)
            ref = np.asarray(ref, dtype=np.float64).reshape(n_img, 1000)
            got = logits[:n_img].astype(np.float64)
            err = float(np.abs(got - ref).max()) / max(float(np.abs(ref).max()), 1e-6)
            tol = 2e-2 if bf16 else 1e-3
            out["parity_check"] = {"images": n_img, "max_rel_err": err, "tolerance": tol, "ok": bool(err <= tol),
                                   "argmax_agree": int((got.argmax(1) == ref.argmax(1)).sum()),
                                   "against": "oracle/mbn_oracle.c F32 mode%s, logits of the first %d images of rank 0's shard"
                                              % (" (bf16 storage emulated)" if bf16 else "", n_img)}
    if args.pw_emul and args.dtype == "f32":
        lib.mbn_tune_set(b"pw_emul", 0)
        lib.mbn_tune_set(b"pw_emul_static", 0)
    net.destroy()
    ctx.close()
    return out if rank == 0 else None


if __name__ == "__main__":
    main()
