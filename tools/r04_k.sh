#!/bin/bash
# round 4, call K: sub-batch stream stagger re-checked on this round's kernels (net_stagger = layers by which consecutive streams are offset), streams 2 / 3
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out
B="python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-configs-alt --no-unfused-stages --no-pw-emul-alt --no-profile"
{
echo "#### bench.py --no-profile, 40 steps; tune net_stagger x --streams; two passes"
for i in 1 2; do for st in 2 3; do for sg in 0 1 2 3 4 6 9; do
  $B --streams $st --tune net_stagger=$sg > $O/r04k_tmp.json 2>> $O/r04k_err.log
  python -c "
import json
o=json.loads(open('$O/r04k_tmp.json').read().strip().splitlines()[-1])
print('pass $i streams $st net_stagger $sg: value %.1f img/s  ms/step %.4f' % (o['value'], o['ms_per_step']))"
done; done; done
} > $O/r04k_stagger.txt 2>&1
cat $O/r04k_stagger.txt
