#!/bin/bash
# round 5, step X: bf16 block kernel — constants staged behind the first window loads (this source state) against the previous state (libmbn_lab_prev.so), same box, alternating
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05x; mkdir -p $O
MBN_LAB=1 timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "bf16_dwpw or bf16_net or headline_bf16" 2>&1 | tail -n 2
show() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], {k:v[0] for k,v in d['stages_frac'].items()})"; }
A="--no-configs-alt --no-unfused-stages --no-pw-emul-alt --no-power --no-cpu-baseline"
for rep in 1 2 3; do
MBN_LAB=libmbn_lab_prev.so python3 bench.py --dtype bf16 --alpha 0.5 --res 160 --batch 512 --steps 200 --warmup 20 $A --record $O/p$rep.json | tail -n 1 | show "bf16 0.5x160 prev"
MBN_LAB=1 python3 bench.py --dtype bf16 --alpha 0.5 --res 160 --batch 512 --steps 200 --warmup 20 $A --record $O/n$rep.json | tail -n 1 | show "bf16 0.5x160 new "
done
MBN_LAB=libmbn_lab_prev.so python3 bench.py --dtype bf16 --batch 512 --steps 60 --warmup 10 $A --record $O/p4.json | tail -n 1 | show "bf16 1.0x224 prev"
MBN_LAB=1 python3 bench.py --dtype bf16 --batch 512 --steps 60 --warmup 10 $A --record $O/n4.json | tail -n 1 | show "bf16 1.0x224 new "
