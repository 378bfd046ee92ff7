#!/bin/bash
# round 4, call M: the whole lab library built with -mllvm -amdgpu-sched-strategy=max-ilp (libmbn_lab_ilp.so) against the default scheduler
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out
MBN_LAB=libmbn_lab_ilp.so timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "headline or net_per_layer or fused_stem_equals or test_f32_dwpw_fused or f32_pointwise or f32_depthwise" > $O/r04m_pytest.log 2>&1; tail -n 2 $O/r04m_pytest.log
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-configs-alt --no-pw-emul-alt"
{
echo "#### bench.py (fp32 headline, lab builds): default instruction scheduler vs -mllvm -amdgpu-sched-strategy=max-ilp on every kernel file; alternating"
for i in 1 2 3; do for lib in 1 libmbn_lab_ilp.so; do
  MBN_LAB=$lib $B > $O/r04m_tmp.json 2>> $O/r04m_err.log
  python -c "
import json
o=json.loads(open('$O/r04m_tmp.json').read().strip().splitlines()[-1])
s=o['stages']; u=o['unfused_stages']['stages']
print('lib=%-22s run $i: value %.1f (no-profile %.1f)  stem %.4f  blocks %.4f  dw8 %.4f  pw8 %.4f (frac %.4f)  unfused dw13 %.4f pw13 %.4f' % ('$lib', o['value'], o['roofline']['value_no_profile'], s['stem_fused']['ms'], s['block_fused']['ms'], s['depthwise']['ms'], s['pointwise']['ms'], o['roofline']['frac'], u['depthwise']['ms'], u['pointwise']['ms']))"
done; done
} > $O/r04m_sched_strategy.txt 2>&1
cat $O/r04m_sched_strategy.txt
