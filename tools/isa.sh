#!/bin/bash
# tools/isa.sh <csrc file stem> [extra hipcc flags]: device ISA of one kernel file -> $ISA_OUT/<stem>.s (default: <tmpdir>/mbn_isa), prints the resource summary.
# The repo root comes from this script's own location (a checkout anywhere), like the other tools.
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
P="$ROOT/cnn-mobilenet-v1-implementation-on-aws-fpga-using-opencl_amd"
OUT="${ISA_OUT:-${TMPDIR:-/tmp}/mbn_isa}"
stem=$1; shift
mkdir -p "$OUT"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -I"$ROOT/include" -I"$P/csrc" -I"$P/host" -S --cuda-device-only "$@" -o "$OUT/$stem.s" "$P/csrc/$stem.hip" 2>&1 | grep -v "hip-link"
grep -E "^\s+\.(name|vgpr_count|sgpr_count|vgpr_spill_count|group_segment_fixed_size|private_segment_fixed_size):" "$OUT/$stem.s" | paste - - - - - - | sed 's/  */ /g'
