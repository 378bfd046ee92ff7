#!/bin/bash
# tools/sweep_bench.sh "<streams list>" "<stagger list>" [extra bench args]: one line per (streams, stagger)
for s in $1; do for g in $2; do
  python bench.py --streams $s --tune net_stagger=$g --no-cpu-baseline --no-profile "${@:3}" 2>/dev/null > /tmp/sweep.json
  python - "$s" "$g" <<'PY'
import json,sys
d=json.load(open('/tmp/sweep.json'))
print("streams %s stagger %s  %8.0f img/s  %.3f ms/step" % (sys.argv[1], sys.argv[2], d["value"], d["ms_per_step"]))
PY
done; done
