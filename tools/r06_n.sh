#!/bin/bash
# round 6, step N: short-K pointwise on the resident-filter GEMM (mbn_f32_pw3.hip, pw_tile = 9) against pw_gemm (pw_tile = 10): parity, then the layers
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06n; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "short_k_resident or test_f32_pointwise" > $O/pytest_pw3.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -n 3 $O/pytest_pw3.log
[ $rc -ne 0 ] && exit $rc
for rep in 1 2; do timeout -k 10 400 python3 tools/layer_bench.py --layers 5,7,9,11,13 --iters 30 --tune pw_tile=10,9 | tee -a $O/layers.txt || exit 1; done
timeout -k 10 400 python3 tools/layer_bench.py --layers 5,7,9,11,13 --batch 64 --iters 30 --tune pw_tile=10,9 | tee -a $O/layers_b64.txt || exit 1
timeout -k 10 400 python3 tools/layer_bench.py --layers 5,7,9,11,13 --batch 16 --iters 30 --tune pw_tile=10,9 | tee -a $O/layers_b16.txt || exit 1
