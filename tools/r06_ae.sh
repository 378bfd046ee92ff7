#!/bin/bash
# round 6, step AE: co-run of a K = 512 GEMM (layer 15) and a depthwise layer (14 / 4) on two streams, with 4 / 3 / 2 GEMM workgroups per CU (tools/corun_bench.py)
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06ae; mkdir -p $O
MBN_LAB=1 timeout -k 10 200 python3 tools/corun_bench.py --a 15 --b 14 --batch 128 --tune misc=3 --tune misc=2 | tee $O/corun_15_14.txt || exit 1
MBN_LAB=1 timeout -k 10 200 python3 tools/corun_bench.py --a 15 --b 4 --batch 128 --tune misc=3 --tune misc=2 | tee $O/corun_15_4.txt || exit 1
MBN_LAB=1 timeout -k 10 200 python3 tools/corun_bench.py --a 15 --b 2 --batch 128 --tune misc=3 --tune misc=2 | tee $O/corun_15_2.txt || exit 1
