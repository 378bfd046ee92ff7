#!/bin/bash
# round 6, step I: dwpw3 as the default of the stride-1 blocks with Cin >= 128: the whole GPU suite on the shipped library, then the bench line
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06i; mkdir -p $O
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu_lean.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -n 4 $O/pytest_gpu_lean.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 > $O/bench.log 2>&1; echo "bench rc=$?"; tail -n 1 $O/bench.log
