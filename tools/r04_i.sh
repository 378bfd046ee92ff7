#!/bin/bash
# round 4, call I: depthwise early layers: forms (exp0: 0 rule, 1 column march, 6 LDS-staged 32-ch slabs, 7 64-ch slabs, 16 long ring) x row segments
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out
{
for L in 2 4 6 8 10; do
  python tools/layer_bench.py --layers $L --iters 40 --warmup 5 --tune exp0=0,1,6,7,16 --tune dw_nseg=0,1,2,3,4,7
done
} > $O/r04i_dw_sweep.txt 2>&1
python - <<'PY'
import re,collections
best=collections.defaultdict(list)
for l in open('gpurun_out/r04i_dw_sweep.txt'):
    m=re.match(r'L(\d+)\s+kind=\d+ (\{.*?\})\s+med ([\d.]+) ms',l)
    if m: best[int(m.group(1))].append((float(m.group(3)),m.group(2)))
for L,v in sorted(best.items()):
    v.sort()
    d=[x for x in v if x[1]=='{"exp0": 0, "dw_nseg": 0}']
    print("L%d default %s | best: %s" % (L, d, v[:4]))
PY
