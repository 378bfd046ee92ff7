#!/bin/bash
# round 4, call D: fused block kernel with the x window two steps ahead (dwpw_variant 7 / 8, lab): parity, per-block A/B, in-network A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out
MBN_LAB=1 python -m pytest tests -m gpu -x -q -k "test_f32_dwpw_fused and not emul and not envelope and not padding" > $O/r04d_pytest_lab.log 2>&1; echo "rc=$?" >> $O/r04d_pytest_lab.log; tail -n 3 $O/r04d_pytest_lab.log
{
echo "#### tools/block_bench.py --blocks 4,6,8,10 --reps 40 (batch 256, fp32), lab knob dwpw_variant: 0 = shipped (x window one step ahead), 7 = two steps ahead where it fits beside the taps (S=1, BN=128: block 6-7), 8 = also S=2 / BN=128 with the taps read in the step (block 4-5); alternating runs"
for i in 1 2 3; do for v in 0 7 8; do echo "## dwpw_variant=$v (run $i)"; python tools/block_bench.py --blocks 4,6,8,10 --reps 40 --tune dwpw_variant=$v; done; done
} > $O/r04d_block_xa2.txt 2>&1
grep -v "^block" $O/r04d_block_xa2.txt | tail -40
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-configs-alt --no-unfused-stages --no-pw-emul-alt"
for i in 1 2; do
  MBN_LAB=1 $B > $O/r04d_bench_v0_$i.json 2> $O/r04d_err.log
  MBN_LAB=1 $B --tune dwpw_variant=8 > $O/r04d_bench_v8_$i.json 2>> $O/r04d_err.log
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04d_bench_v*.json")):
    o=json.loads(open(f).read().strip().splitlines()[-1])
    print("%-28s value %9.1f (no-profile %s) ms/step %.4f  block_fused %.4f ms  per block: %s  pw frac %.4f" % (f.split('/')[-1], o['value'], o.get('value_no_profile',{}).get('value'), o['ms_per_step'], o['stages']['block_fused']['ms'], [l['ms'] for l in o['layers'] if l['stage']=='block_fused'], o['roofline']['frac']))
PY
