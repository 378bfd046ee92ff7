#!/bin/bash
# round 5, step G: stamps of the interleaved form (variant 100 + 64 + 16384) next to the burst form (164), block 6-7 and 4-5
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05g; mkdir -p $O
python3 tools/stamp_dwpw2.py --block 6 --variant 164 | tee $O/stamps_block6_burst.txt | tail -n 1
python3 tools/stamp_dwpw2.py --block 6 --variant 16548 | tee $O/stamps_block6_il.txt
python3 tools/stamp_dwpw2.py --block 4 --variant 164 | tee $O/stamps_block4_burst.txt | tail -n 1
python3 tools/stamp_dwpw2.py --block 4 --variant 16548 | tee $O/stamps_block4_il.txt | tail -n 20
