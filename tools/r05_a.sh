#!/bin/bash
# round 5, step A: the compact bench line as the driver runs it + the tests that guard it + the new full-size bf16 per-layer tests
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05a; mkdir -p $O
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_stdout.txt 2> $O/bench_stderr.txt; echo "bench rc=$?" | tee $O/bench_rc.txt
tail -n 1 $O/bench_stdout.txt | wc -c | tee -a $O/bench_rc.txt
wc -l $O/bench_stdout.txt | tee -a $O/bench_rc.txt
cp gpurun_out/bench_full_f32_a1_r224_b256_n1.json $O/ 2>/dev/null
timeout -k 10 1500 python3 -m pytest tests -m gpu -x -q -k "bench or bf16_net or headline_bf16 or fused_stem or dist_rccl" > $O/pytest_sel.log 2>&1; echo "pytest rc=$?" | tee -a $O/bench_rc.txt
tail -n 5 $O/pytest_sel.log
