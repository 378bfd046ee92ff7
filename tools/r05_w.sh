#!/bin/bash
# round 5, step W: compile-time tap stride in LDS (this source state) against the previous source state (libmbn_lab_prev.so, built from HEAD), same box, alternating
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05w; mkdir -p $O
show() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], {k:v[0] for k,v in d['stages_frac'].items()})"; }
A="--no-configs-alt --no-unfused-stages --no-pw-emul-alt --no-power --no-cpu-baseline"
for rep in 1 2 3; do
MBN_LAB=libmbn_lab_prev.so python3 bench.py --dtype bf16 --alpha 0.5 --res 160 --batch 512 --steps 200 --warmup 20 $A --record $O/p$rep.json | tail -n 1 | show "bf16 0.5x160 prev (runtime stride)   "
MBN_LAB=1 python3 bench.py --dtype bf16 --alpha 0.5 --res 160 --batch 512 --steps 200 --warmup 20 $A --record $O/n$rep.json | tail -n 1 | show "bf16 0.5x160 new (compile-time stride)"
done
MBN_LAB=libmbn_lab_prev.so python3 bench.py --dtype bf16 --batch 512 --steps 60 --warmup 10 $A --record $O/p4.json | tail -n 1 | show "bf16 1.0x224 prev"
MBN_LAB=1 python3 bench.py --dtype bf16 --batch 512 --steps 60 --warmup 10 $A --record $O/n4.json | tail -n 1 | show "bf16 1.0x224 new "
for rep in 1 2; do
echo "fp32 blocks prev"; MBN_LAB=libmbn_lab_prev.so python3 tools/block_bench.py --blocks 4,6,8 --reps 30 | grep "^L"
echo "fp32 blocks new"; MBN_LAB=1 python3 tools/block_bench.py --blocks 4,6,8 --reps 30 | grep "^L"
done
