#!/bin/bash
# tools/pmc_pass.sh <tag> <counters...> -- <layer_bench args>
# One rocprofv3 --pmc pass (counters in their own run, kernel-trace only: MI355X guide) over tools/layer_bench.py
# (or over the tool named by PMC_TARGET, e.g. PMC_TARGET=tools/block_bench.py).
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
ctrs=()
while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc "${ctrs[@]}" --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$tag -- python3 $R/${PMC_TARGET:-tools/layer_bench.py} "$@" > $R/gpurun_out/pmc_$tag.log 2>&1
