#!/bin/bash
# round 4, call H: one-launch classifier tail (pool + FC + softmax + top-k): parity + timing
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "pool_fc or tail or softmax or classif" > $O/r04h_pytest.log 2>&1; echo "rc=$?" >> $O/r04h_pytest.log; tail -n 6 $O/r04h_pytest.log
timeout -k 10 300 python tools/tail_bench.py > $O/r04h_tail_bench.txt 2>&1; cat $O/r04h_tail_bench.txt
