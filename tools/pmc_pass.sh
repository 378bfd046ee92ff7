#!/bin/bash
# tools/pmc_pass.sh <tag> <counters...> -- <layer_bench args>
# One rocprofv3 --pmc pass (counters in their own run, kernel-trace only: MI355X guide) over tools/layer_bench.py
# (or over the tool named by PMC_TARGET, e.g. PMC_TARGET=tools/block_bench.py).
# A counter set that over-subscribes a block's counter slots makes rocprofv3 abort() inside the first dispatch ("error code 38:
# Request exceeds the capabilities of the hardware to collect") and the python process then sits there: the pass runs under
# `timeout`, and that message in the log fails the pass at once (ADVICE r3 / VERDICT r3 item 6).
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
ctrs=()
while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
shift
cd /tmp && export TMPDIR=/tmp
set +e
timeout -k 10 ${PMC_TIMEOUT:-300} rocprofv3 --pmc "${ctrs[@]}" --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$tag -- python3 $R/${PMC_TARGET:-tools/layer_bench.py} "$@" > $R/gpurun_out/pmc_$tag.log 2>&1
rc=$?
set -e
if grep -q "Request exceeds the capabilities of the hardware" $R/gpurun_out/pmc_$tag.log; then
    echo "pmc_pass $tag: counter set [${ctrs[*]}] does not fit the hardware counter slots (rocprofv3 error 38) - split it" >&2
    exit 38
fi
exit $rc
