#!/bin/bash
# round 4, call J: sub-batch streams on disjoint halves of the chip (cu_mask 1 = XCDs 0-3 / 4-7, 2 = halves of every XCD) vs shared (0)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out
B="python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-configs-alt --no-unfused-stages --no-pw-emul-alt --no-profile"
{
echo "#### bench.py --no-profile (two sub-batch streams, 40 steps), tune cu_mask: 0 = both streams on the whole chip, 1 = XCDs 0-3 / 4-7, 2 = lower / upper half of every XCD; alternating"
for i in 1 2 3; do for m in 0 1 2; do
  $B --tune cu_mask=$m > $O/r04j_$m_$i.json 2>> $O/r04j_err.log
  python -c "
import json,sys
o=json.loads(open('$O/r04j_$m_$i.json').read().strip().splitlines()[-1])
print('cu_mask=$m run $i: value %.1f img/s  ms/step %.4f  step_ms median %.4f p10 %.4f p90 %.4f' % (o['value'], o['ms_per_step'], o['step_ms']['median'], o['step_ms']['p10'], o['step_ms']['p90']))"
done; done
} > $O/r04j_cu_mask.txt 2>&1
cat $O/r04j_cu_mask.txt; tail -3 $O/r04j_err.log
