#!/bin/bash
# round 6, step R: K = 512 pointwise on the resident-filter GEMM (64-channel slices, 12 waves per workgroup; pw_tile = 9) against pw_gemm (pw_tile = 10)
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06r; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "short_k_resident" > $O/pytest_pw3.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -n 3 $O/pytest_pw3.log
[ $rc -ne 0 ] && exit $rc
for rep in 1 2; do timeout -k 10 400 python3 tools/layer_bench.py --layers 15,25 --iters 30 --tune pw_tile=10,9 | tee -a $O/layers.txt || exit 1; done
