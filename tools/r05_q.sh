#!/bin/bash
# round 5, step Q: LITERAL pointwise on v_mfma_i32_32x32x32_i8: parity tests (bit-exact against oracle, scalar kernel, v_dot4 form, hand-derived KATs) and timing
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05q; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "literal or c_host" > $O/pytest_literal.log 2>&1; echo "pytest rc=$?"; tail -n 12 $O/pytest_literal.log
timeout -k 10 300 python3 tools/literal_bench.py 2>&1 | head -n 24 | tee $O/literal_bench.txt
