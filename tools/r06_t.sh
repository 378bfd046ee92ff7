#!/bin/bash
# round 6, step T: a run of bf16 blocks resident in LDS (mbn_bf16_res.hip): parity, then the time of the five 10x10 blocks against five fused launches
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06t; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "blocks_resident" > $O/pytest_res.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -n 12 $O/pytest_res.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python3 tools/res_bench.py | tee $O/res_bench.txt
