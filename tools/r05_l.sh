#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05l; mkdir -p $O
python3 tools/power_probe_net.py | tee $O/power_probe_net.txt
