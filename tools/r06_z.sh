#!/bin/bash
# round 6, step Z: register-filter bf16 pointwise: ablations on layer 15 at batch 512 (exp0 = 700 + bits: 1 no LDS-DMA in the loop, 2 no fragment reads / MFMAs, 4 no stores)
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06z; mkdir -p $O
timeout -k 10 300 python3 tools/layer_bench.py --dtype bf16 --batch 512 --layers 15 --iters 30 --tune pw_ring=8 --tune exp0=0,701,702,704,705,706,707 | tee $O/rf_ablate.txt
