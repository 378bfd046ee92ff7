#!/bin/bash
# round 6, step Y: bf16 K = 512 pointwise with the filter in registers (mbn_bf16_pw_rf.hip): parity (lab: bit for bit against the M16 streaming kernel), then A/B on layers 15, 25
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06y; mkdir -p $O
MBN_LAB=1 timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "register_filter" > $O/pytest_rf.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -n 15 $O/pytest_rf.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python3 tools/layer_bench.py --dtype bf16 --batch 512 --layers 15,25 --iters 30 --tune pw_ring=0,8 | tee $O/rf_ab.txt
