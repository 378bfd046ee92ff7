#!/bin/bash
# round 4, call E: fused stem with conv1 on the fp32 MFMA (lab conv_variant = 5): parity + A/B in the network
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out
MBN_LAB=1 python -m pytest tests -m gpu -x -q -k "fused_stem_conv1_on_fp32_mfma or fused_stem_equals" > $O/r04e_pytest_lab.log 2>&1; echo "rc=$?" >> $O/r04e_pytest_lab.log; tail -n 5 $O/r04e_pytest_lab.log
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-configs-alt --no-unfused-stages --no-pw-emul-alt"
for i in 1 2 3; do
  MBN_LAB=1 $B > $O/r04e_bench_v0_$i.json 2> $O/r04e_err.log
  MBN_LAB=1 $B --tune conv_variant=5 > $O/r04e_bench_v5_$i.json 2>> $O/r04e_err.log
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04e_bench_v*.json")):
    o=json.loads(open(f).read().strip().splitlines()[-1])
    print("%-28s value %9.1f (no-profile %s) ms/step %.4f  stem %.4f ms  blocks %.4f  pw frac %.4f  parity %s" % (f.split('/')[-1], o['value'], o.get('value_no_profile',{}).get('value'), o['ms_per_step'], o['stages']['stem_fused']['ms'], o['stages']['block_fused']['ms'], o['roofline']['frac'], o.get('parity_check')))
PY
