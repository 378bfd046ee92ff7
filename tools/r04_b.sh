#!/bin/bash
# round 4, call B: bf16 fused block on 16x16x32 MFMAs (tests + A/B), the streaming GEMM's 16x16x32 form on the K <= 256 layers
# NOTE: the file names and the header text below are those of the commit this call ran at (e3-era: the block kernel's DEFAULT was the 16x16x32 form and
# `misc=32` selected the 32x32x16 form). At HEAD it is the other way round — the shipped fused bf16 block is 32x32x16 and `misc=32` selects the 16x16x32 lab form
# (mbn_bf16_dwpw2.hip launch2) — so a re-run from HEAD produces the two series under swapped names. Kept as the record of the call, not as a tool.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out
python -m pytest tests -m gpu -x -q -k "bf16_dwpw_fused or bf16_net_per_layer or headline_bf16 or bf16_fused_stem" > $O/r04b_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r04b_pytest.log
tail -4 $O/r04b_pytest.log
B="python bench.py --dtype bf16 --batch 512 --steps 40 --warmup 8 --no-cpu-baseline --no-configs-alt --no-unfused-stages"
for i in 1 2; do
  MBN_LAB=1 $B --tune misc=32 > $O/r04b_bf16_blocks_32x32x16_$i.json 2> $O/r04b_err.log
  MBN_LAB=1 $B > $O/r04b_bf16_blocks_16x16x32_$i.json 2>> $O/r04b_err.log
done
python - <<'PY' > gpurun_out/r04b_bf16_blocks_ab.txt
import json,glob
print("#### bf16 1.0x224 batch 512 (bench.py --dtype bf16, lab build): fused blocks 4-11 with the pointwise products on v_mfma_f32_32x32x16_bf16 (tune misc=32, round 3's form) vs 16x16x32 (shipped now); alternating runs")
for f in sorted(glob.glob("gpurun_out/r04b_bf16_blocks_*.json")):
    o=json.loads(open(f).read().strip().splitlines()[-1])
    print("%-52s value %9.1f img/s  ms/step %.4f  block_fused %.4f ms  stem %.4f  pointwise %.4f  depthwise %.4f" % (f.split('/')[-1], o['value'], o['ms_per_step'], o['stages']['block_fused']['ms'], o['stages']['stem_fused']['ms'], o['stages']['pointwise']['ms'], o['stages']['depthwise']['ms']))
    print("     per block:", [ (l['layers'], l['ms']) for l in o['layers'] if l['stage']=='block_fused'])
PY
cat $O/r04b_bf16_blocks_ab.txt
{
echo "#### bf16 pointwise, batch 512: pw_ring 1 = pw_gemm<bf16> (32x32x16), 4 = the streaming kernel wherever eligible; misc 16 = its 16x16x32 form"
python tools/layer_bench.py --dtype bf16 --batch 512 --layers 5,7,9,11,13,15 --iters 60 --warmup 10 --tune pw_ring=1,4 --tune misc=0,16
echo "#### 0.5x160"
python tools/layer_bench.py --dtype bf16 --batch 512 --alpha 0.5 --res 160 --layers 7,9,11,13,15,25,27 --iters 60 --warmup 10 --tune pw_ring=1,4 --tune misc=0,16
} > $O/r04b_bf16_pw_shapes.txt 2>&1
cat $O/r04b_bf16_pw_shapes.txt
