#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05k; mkdir -p $O
rocm-smi --showpower --showclocks --json 2>&1 | head -c 1500; echo
rocm-smi --showmaxpower 2>&1 | tail -n 6
python3 tools/power_probe.py --variants 0,9,113,115,15475,116 | tee $O/power_probe_block6.txt
