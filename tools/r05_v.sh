#!/bin/bash
# round 5, step V: depthwise taps in LDS at a compile-time tap stride (both block kernels): parity + bf16 0.5x160 / 1.0x224 and the fp32 blocks
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05v; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "dwpw or fused_block or net_default or bf16_net or headline" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 3 $O/pytest.log
show() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], (d.get('parity_check') or {}).get('max_rel_err'), {k:v[0] for k,v in d['stages_frac'].items()})"; }
A="--no-configs-alt --no-unfused-stages --no-pw-emul-alt --no-power --cpu-images 8 --no-cpu-variants"
for rep in 1 2 3; do
python3 bench.py --dtype bf16 --alpha 0.5 --res 160 --batch 512 --steps 200 --warmup 20 $A --record $O/a$rep.json | tail -n 1 | show "bf16 0.5x160"
done
python3 bench.py --dtype bf16 --batch 512 --steps 60 --warmup 10 $A --record $O/c.json | tail -n 1 | show "bf16 1.0x224"
python3 bench.py --steps 30 --warmup 5 $A --record $O/d.json | tail -n 1 | show "fp32 1.0x224 b256"
python3 tools/block_bench.py --blocks 4,6,8 --reps 30 | grep "^L"
