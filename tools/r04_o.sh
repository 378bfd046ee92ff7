#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out
{
echo "#### tools/block_bench.py --blocks 4,6 --reps 40: dwpw_variant 0 = shipped (8 waves, 128-row tiles), 6 = 12 waves on 192-row tiles (lab, round 3), alternating"
for i in 1 2 3; do for v in 0 6; do echo "## dwpw_variant=$v (run $i)"; python tools/block_bench.py --blocks 4,6 --reps 40 --tune dwpw_variant=$v; done; done
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-configs-alt --no-unfused-stages --no-pw-emul-alt"
for i in 1 2; do for v in 0 6; do
  MBN_LAB=1 $B --tune dwpw_variant=$v > $O/r04o_tmp.json 2>> $O/r04o_err.log
  python -c "
import json
o=json.loads(open('$O/r04o_tmp.json').read().strip().splitlines()[-1])
print('in the network dwpw_variant=$v run $i: value %.1f (no-profile %.1f) blocks %s' % (o['value'], o['roofline']['value_no_profile'], [l['ms'] for l in o['layers'] if l['stage']=='block_fused']))"
done; done
} > $O/r04o_block_12waves.txt 2>&1
grep -v "^block" $O/r04o_block_12waves.txt
