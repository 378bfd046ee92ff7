#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05t; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "held_clock or bench_self or bench_rccl" 2>&1 | tail -n 3
python3 tools/power_probe_net.py --stem | tee $O/power_probe_stem.txt
