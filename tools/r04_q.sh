#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out
B="python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-configs-alt --no-unfused-stages --no-pw-emul-alt --no-profile"
{
echo "#### bench.py --no-profile, 40 steps: default (two sub-batch streams) vs --streams 1 vs --graph (one hipGraph per step, one stream); alternating"
for i in 1 2; do for mode in "" "--streams 1" "--graph"; do
  $B $mode > $O/r04q_tmp.json 2>> $O/r04q_err.log
  python -c "
import json
o=json.loads(open('$O/r04q_tmp.json').read().strip().splitlines()[-1])
print('run $i %-14s value %.1f img/s  ms/step %.4f' % ('[$mode]', o['value'], o['ms_per_step']))"
done; done
} > $O/r04q_graph.txt 2>&1
cat $O/r04q_graph.txt; tail -2 $O/r04q_err.log
