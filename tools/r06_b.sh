#!/bin/bash
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06b; mkdir -p $O
for b in 6 4 10; do for n in 24 64 256; do echo "== block $b batch $n"; timeout -k 10 200 python3 tools/dwpw3_debug.py --block $b --batch $n 2>&1 | tee -a $O/debug.txt || exit 1; done; done
