#!/bin/bash
# round 5, step AA: bf16 block kernel — waves whose column group lies past Cout skip their MFMAs (this state) against the previous state (libmbn_lab_prev.so)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05aa; mkdir -p $O
MBN_LAB=1 timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "bf16_dwpw or bf16_net or headline_bf16" 2>&1 | tail -n 2
show() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], (d.get('parity_check') or {}).get('max_rel_err'), [(l['layers'], l['ms']) for l in json.load(open('$2'))['layers'][1:4]])"; }
A="--no-configs-alt --no-unfused-stages --no-pw-emul-alt --no-power --cpu-images 8 --no-cpu-variants"
for rep in 1 2 3; do
MBN_LAB=libmbn_lab_prev.so python3 bench.py --dtype bf16 --alpha 0.5 --res 160 --batch 512 --steps 200 --warmup 20 $A --record $O/p$rep.json | tail -n 1 | show "0.5x160 prev" $O/p$rep.json
MBN_LAB=1 python3 bench.py --dtype bf16 --alpha 0.5 --res 160 --batch 512 --steps 200 --warmup 20 $A --record $O/n$rep.json | tail -n 1 | show "0.5x160 new " $O/n$rep.json
done
