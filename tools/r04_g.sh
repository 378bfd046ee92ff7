#!/bin/bash
# round 4, call G: block kernels without the per-tile accumulator zeroing (first MFMAs of a tile take C = 0): parity + A/B against the previous build
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out
python -m pytest tests -m gpu -x -q -k "dwpw or headline or net_per_layer or graft" > $O/r04g_pytest_lean.log 2>&1; echo "rc=$?" >> $O/r04g_pytest_lean.log; tail -n 3 $O/r04g_pytest_lean.log
{
echo "#### tools/block_bench.py --blocks 4,6,8,10 --reps 40 (batch 256 fp32): prev = per-tile v_mov zeroing (libmbn_lab_prev.so), new = first MFMAs of a tile with C = 0; alternating"
for i in 1 2 3; do
  echo "## prev (run $i)"; MBN_LAB=libmbn_lab_prev.so python tools/block_bench.py --blocks 4,6,8,10 --reps 40
  echo "## new (run $i)"; MBN_LAB=1 python tools/block_bench.py --blocks 4,6,8,10 --reps 40
done
echo "#### bf16 1.0x224 batch 512 in the network (bench.py --dtype bf16), same two builds"
B="python bench.py --dtype bf16 --batch 512 --steps 40 --warmup 8 --no-cpu-baseline --no-configs-alt --no-unfused-stages"
for i in 1 2; do
  MBN_LAB=libmbn_lab_prev.so $B > $O/r04g_bf16_prev_$i.json 2> $O/r04g_err.log
  MBN_LAB=1 $B > $O/r04g_bf16_new_$i.json 2>> $O/r04g_err.log
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04g_bf16_*.json")):
    o=json.loads(open(f).read().strip().splitlines()[-1])
    print("%-28s value %9.1f ms/step %.4f  block_fused %.4f ms  per block: %s" % (f.split('/')[-1], o['value'], o['ms_per_step'], o['stages']['block_fused']['ms'], [l['ms'] for l in o['layers'] if l['stage']=='block_fused']))
PY
} > $O/r04g_block_c0.txt 2>&1
grep -v "^block" $O/r04g_block_c0.txt
