#!/bin/bash
# round 4, call F: conv1 on the fp32 MFMA shipped in the stem and in the stand-alone kernel: full GPU suite (lean), A/B in the network, stand-alone A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out
python -m pytest tests -m gpu -x -q > $O/r04f_pytest_lean.log 2>&1; echo "rc=$?" >> $O/r04f_pytest_lean.log; tail -n 4 $O/r04f_pytest_lean.log
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-configs-alt --no-unfused-stages --no-pw-emul-alt"
for i in 1 2 3; do
  MBN_LAB=1 $B --tune conv_variant=5 > $O/r04f_bench_valu_$i.json 2> $O/r04f_err.log
  MBN_LAB=1 $B > $O/r04f_bench_mfma_$i.json 2>> $O/r04f_err.log
done
{
echo "#### fused stem (layers 1-3, batch 256 fp32): conv1 phase on the VALU (v_pk_fma_f32 chain, round 3; lab conv_variant=5) vs on v_mfma_f32_16x16x4_f32 (shipped now); bench.py, lab build, alternating runs"
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04f_bench_*.json")):
    o=json.loads(open(f).read().strip().splitlines()[-1])
    print("%-28s value %9.1f (no-profile %s) ms/step %.4f  stem %.4f ms  blocks %.4f  pw frac %.4f" % (f.split('/')[-1], o['value'], o.get('value_no_profile',{}).get('value'), o['ms_per_step'], o['stages']['stem_fused']['ms'], o['stages']['block_fused']['ms'], o['roofline']['frac']))
PY
echo "#### stand-alone conv1 (layer 1 alone, tools/layer_bench.py --layers 1): conv_variant 0 = conv1_mfma_f32 (shipped for 32 channels), 6 = the fmaf-chain kernel conv3x3s2c3_f32_nhwc"
python tools/layer_bench.py --layers 1 --iters 60 --warmup 10 --tune conv_variant=0,6
} > $O/r04f_stem_conv1_mfma.txt 2>&1
cat $O/r04f_stem_conv1_mfma.txt
