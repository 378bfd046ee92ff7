#!/bin/bash
# round 5, step U: bf16 block kernel as two 4-wave workgroups per CU on 64-row tiles (lab exp2 = 44) against the 8-wave workgroup, bf16 0.5x160 and 1.0x224 (lab library both sides)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05u; mkdir -p $O
export MBN_LAB=1
show() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], (d.get('parity_check') or {}).get('max_rel_err'), {k:v[0] for k,v in d['stages_frac'].items()})"; }
A="--no-configs-alt --no-unfused-stages --no-pw-emul-alt --no-power --cpu-images 8 --no-cpu-variants"
for rep in 1 2 3; do
python3 bench.py --dtype bf16 --alpha 0.5 --res 160 --batch 512 --steps 200 --warmup 20 $A --record $O/a$rep.json | tail -n 1 | show "0.5x160 8-wave      "
python3 bench.py --dtype bf16 --alpha 0.5 --res 160 --batch 512 --steps 200 --warmup 20 $A --tune exp2=44 --record $O/b$rep.json | tail -n 1 | show "0.5x160 2 x 4-wave  "
done
python3 bench.py --dtype bf16 --batch 512 --steps 60 --warmup 10 $A --record $O/c.json | tail -n 1 | show "1.0x224 8-wave      "
python3 bench.py --dtype bf16 --batch 512 --steps 60 --warmup 10 $A --tune exp2=44 --record $O/d.json | tail -n 1 | show "1.0x224 2 x 4-wave  "
