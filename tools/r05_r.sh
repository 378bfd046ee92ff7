#!/bin/bash
# round 5, step R: bf16 — the blocks wider than one 256-column tile (24-25, 26-27 at both sizes; 12-13 ... at 1.0x224) fused by an explicit mask against the default (two launches)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05r; mkdir -p $O
C="--no-cpu-baseline --no-configs-alt --no-unfused-stages --no-pw-emul-alt --no-power"
show() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], {k:v[0] for k,v in d['stages_frac'].items()})"; }
for rep in 1 2; do
python3 bench.py --dtype bf16 --alpha 0.5 --res 160 --batch 512 --steps 200 --warmup 20 $C --record $O/a.json | tail -n 1 | show "0.5x160 default          "
python3 bench.py --dtype bf16 --alpha 0.5 --res 160 --batch 512 --steps 200 --warmup 20 $C --fuse-blocks 0x0FFFFFFE --record $O/b.json | tail -n 1 | show "0.5x160 all blocks fused "
python3 bench.py --dtype bf16 --alpha 0.5 --res 160 --batch 512 --steps 200 --warmup 20 $C --fuse-blocks 0x05FFFFFE --record $O/c.json | tail -n 1 | show "0.5x160 all but 26-27    "
done
python3 bench.py --dtype bf16 --batch 512 --steps 60 --warmup 10 $C --record $O/d.json | tail -n 1 | show "1.0x224 default          "
python3 bench.py --dtype bf16 --batch 512 --steps 60 --warmup 10 $C --fuse-blocks 0x0FFFFFFE --record $O/e.json | tail -n 1 | show "1.0x224 all blocks fused "
