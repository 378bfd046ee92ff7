#!/bin/bash
# round 6, step O: pw3 prefetch depth (lab exp1 = 1: shallow, 0: up to two tiles ahead, 2: K = 256 eight half-rounds)
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06o; mkdir -p $O
MBN_LAB=1 timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "short_k_resident" > $O/pytest_pw3.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -n 3 $O/pytest_pw3.log
[ $rc -ne 0 ] && exit $rc
for rep in 1 2; do timeout -k 10 400 python3 tools/layer_bench.py --layers 5,7,9,11,13 --iters 30 --tune pw_tile=9 --tune exp1=1,0,2 | tee -a $O/layers.txt || exit 1; done
timeout -k 10 400 python3 tools/layer_bench.py --layers 5,7,9,11,13 --iters 30 --tune pw_tile=10 | tee -a $O/layers.txt || exit 1
