#!/bin/bash
# CPU-only sanitizer pass (GPU ASan is not available on the pool): builds the C host and the oracle with
# -fsanitize=address,undefined, swaps them in for the test run, restores the normal builds afterwards.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
P="$R/cnn-mobilenet-v1-implementation-on-aws-fpga-using-opencl_amd"
gcc -O1 -g -std=c11 -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer -I"$R/include" -D_GNU_SOURCE -shared \
    -o /tmp/libmbn_host_asan.so "$P"/host/mbn_status.c "$P"/host/mbn_loaders.c "$P"/host/mbn_plan.c "$P"/host/mbn_h5.c \
    "$P"/host/mbn_weights.c "$P"/host/mbn_ranks.c -lm -lpthread
make -C "$R/oracle" asan > /dev/null
cp "$P/libmbn_host.so" /tmp/libmbn_host_plain.so; cp "$R/oracle/libmbn_oracle.so" /tmp/libmbn_oracle_plain.so
trap 'cp /tmp/libmbn_host_plain.so "$P/libmbn_host.so"; cp /tmp/libmbn_oracle_plain.so "$R/oracle/libmbn_oracle.so"; rm -f "$R/oracle/libmbn_oracle_asan.so"' EXIT
cp /tmp/libmbn_host_asan.so "$P/libmbn_host.so"; cp "$R/oracle/libmbn_oracle_asan.so" "$R/oracle/libmbn_oracle.so"
cd "$R"
ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" \
python -m pytest tests/test_host_cpu.py tests/test_oracle_cpu.py -x -q -k "not abi_library and not struct_matches and not vs_torch"
