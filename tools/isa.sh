#!/bin/bash
# tools/isa.sh <csrc file stem> [extra hipcc flags]: device ISA of one kernel file -> /tmp/isa/<stem>.s, prints the resource summary
P=/root/repo/cnn-mobilenet-v1-implementation-on-aws-fpga-using-opencl_amd
stem=$1; shift
mkdir -p /tmp/isa
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -I/root/repo/include -I$P/csrc -I$P/host -S --cuda-device-only "$@" -o /tmp/isa/$stem.s $P/csrc/$stem.hip 2>&1 | grep -v "hip-link"
grep -E "^\s+\.(name|vgpr_count|sgpr_count|vgpr_spill_count|group_segment_fixed_size|private_segment_fixed_size):" /tmp/isa/$stem.s | paste - - - - - - | sed 's/  */ /g'
