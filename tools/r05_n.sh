#!/bin/bash
# round 5, step N: hipGraph replay of the step on the short configurations (bf16 0.5x160 batch 512: 19 launches in 0.58 ms; fp32 batch 1: 27 launches in 0.14 ms)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05n; mkdir -p $O
C="--no-cpu-baseline --no-configs-alt --no-unfused-stages --no-pw-emul-alt --no-power"
for g in "" "--graph"; do
  python3 bench.py --dtype bf16 --alpha 0.5 --res 160 --batch 512 --steps 200 --warmup 20 $C $g --record $O/bf16_05_160$g.json | tail -n 1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('bf16 0.5x160 b512 $g', d['value'], d['ms_per_step'])"
  python3 bench.py --batch 1 --steps 500 --warmup 50 $C $g --record $O/f32_b1$g.json | tail -n 1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('fp32 b1 $g', d['value'], d['ms_per_step'])"
  python3 bench.py --batch 8 --steps 300 --warmup 30 $C $g --record $O/f32_b8$g.json | tail -n 1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('fp32 b8 $g', d['value'], d['ms_per_step'])"
  python3 bench.py --dtype bf16 --batch 512 --steps 60 --warmup 10 $C $g --record $O/bf16_10_224$g.json | tail -n 1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('bf16 1.0x224 b512 $g', d['value'], d['ms_per_step'])"
done
