#!/bin/bash
# round 5, step Z: bf16 block kernel, Cin = 32 on a 256-row tile with four lanes per pixel pair (H32, default) against the padded 128-row form (lab exp0 = 53)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05z; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "bf16_dwpw or bf16_net or headline_bf16" 2>&1 | tail -n 2
export MBN_LAB=1
show() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], (d.get('parity_check') or {}).get('max_rel_err'), [(l['layers'], l['ms']) for l in json.load(open('$2'))['layers'][1:4]])"; }
A="--no-configs-alt --no-unfused-stages --no-pw-emul-alt --no-power --cpu-images 8 --no-cpu-variants"
for rep in 1 2 3; do
python3 bench.py --dtype bf16 --alpha 0.5 --res 160 --batch 512 --steps 200 --warmup 20 $A --tune exp0=53 --record $O/p$rep.json | tail -n 1 | show "0.5x160 padded 128-row form" $O/p$rep.json
python3 bench.py --dtype bf16 --alpha 0.5 --res 160 --batch 512 --steps 200 --warmup 20 $A --record $O/n$rep.json | tail -n 1 | show "0.5x160 H32 256-row form   " $O/n$rep.json
done
