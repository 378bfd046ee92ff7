#!/bin/bash
# round 5, step S: bf16 block kernel for Cout = 64 (mod 128): parity, then bf16 0.5x160 with block 6-7 fused by the new default
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05s; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "bf16" > $O/pytest_bf16.log 2>&1; echo "pytest rc=$?"; tail -n 4 $O/pytest_bf16.log
C="--no-cpu-baseline --no-configs-alt --no-unfused-stages --no-pw-emul-alt --no-power"
show() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d.get('parity_check'), {k:v[0] for k,v in d['stages_frac'].items()})"; }
for rep in 1 2 3; do
python3 bench.py --dtype bf16 --alpha 0.5 --res 160 --batch 512 --steps 200 --warmup 20 --no-configs-alt --no-unfused-stages --no-pw-emul-alt --no-power --cpu-images 8 --no-cpu-variants --record $O/a$rep.json | tail -n 1 | show "0.5x160 default (6-7 fused)"
python3 bench.py --dtype bf16 --alpha 0.5 --res 160 --batch 512 --steps 200 --warmup 20 $C --fuse-blocks 0x0FFFFFBE --record $O/b$rep.json | tail -n 1 | show "0.5x160 without 6-7        "
done
