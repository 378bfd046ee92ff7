#!/bin/bash
# round 5, step P: full GPU suite on the shipped library after the block-kernel, pool+FC-template and depthwise-rule changes
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05p; mkdir -p $O
timeout -k 10 1150 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu_lean.log 2>&1; echo "pytest lean rc=$?"; tail -n 4 $O/pytest_gpu_lean.log
