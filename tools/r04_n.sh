#!/bin/bash
# round 4, call N: csrc/mbn_f32_dw.hip alone under other instruction-scheduling strategies (-mllvm -amdgpu-sched-strategy=...), all 13 depthwise layers
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out
{
echo "#### tools/layer_bench.py --layers 2,4,...,26 --iters 60 (batch 256 fp32), sum of the 13 medians; lab library with mbn_f32_dw.hip built with the named strategy; three alternating passes"
for i in 1 2 3; do for lib in 1 libmbn_lab_dw_max-ilp.so libmbn_lab_dw_max-memory-clause.so libmbn_lab_dw_iterative-minreg.so libmbn_lab_dw_iterative-ilp.so; do
  MBN_LAB=$lib python tools/layer_bench.py --layers 2,4,6,8,10,12,14,16,18,20,22,24,26 --iters 60 --warmup 8 --json > $O/r04n_tmp.json 2>> $O/r04n_err.log
  python -c "
import json
r=json.load(open('$O/r04n_tmp.json'))
print('pass $i %-40s sum13 %.4f ms  |' % ('$lib', sum(x['ms_med'] for x in r)), ' '.join('L%d %.4f' % (x['layer'], x['ms_med']) for x in r))"
done; done
} > $O/r04n_dw_sched.txt 2>&1
cat $O/r04n_dw_sched.txt | cut -c1-200
