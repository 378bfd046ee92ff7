#!/bin/bash
# round 5, step D: in-kernel stamps of the fused block kernel (burst form) with parts switched off: where do the cycles of a step go?
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05d; mkdir -p $O
python3 tools/stamp_dwpw2.py --block 6 --variant 164 | tee $O/stamps_block6_full.txt
for v in 180 169 185 191 195; do python3 tools/stamp_dwpw2.py --block 6 --variant $v --brief | tail -n 1 | tee -a $O/stamps_block6_ablations.txt; done
python3 tools/stamp_dwpw2.py --block 6 --variant 191 | tee $O/stamps_block6_skeleton.txt
