#!/bin/bash
# tools/r02_traffic.sh — regenerates profiles/traffic.json at HEAD: separate rocprofv3 --pmc passes for
# FETCH_SIZE and WRITE_SIZE (MI355X guide: the two do not fit one pass; counters in their own run with --kernel-trace only)
# over tools/layer_bench.py, for the three benched configurations:
#   f32            MobileNet-V1 1.0x224 fp32  batch 256   pointwise 3..27 + depthwise 2..26
#   bf16_1x224     1.0x224 bf16 batch 512                  pointwise 3..27 + depthwise
#   bf16_0.5x160   0.5x160 bf16 batch 512                  pointwise 3..27 + depthwise
# plus `rocprofv3 --kernel-trace --stats` of the default bench command. Run on the GPU box from the repo root:
#   bash tools/r02_traffic.sh <git-sha>
# Outputs: gpurun_out/r02_traffic/{traffic.json, *_counter_collection.csv, kernel_stats.csv, bench_under_rocprof.json}
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
SHA=${1:-unknown}
O=$R/gpurun_out/r02_traffic
mkdir -p $O
PW=3,5,7,9,11,13,15,17,19,21,23,25,27
DW=2,4,6,8,10,12,14,16,18,20,22,24,26
pass() {   # tag counter layer-list extra-args...
  local tag=$1 ctr=$2 layers=$3; shift 3
  bash $R/tools/pmc_pass.sh r02t_${tag} $ctr -- --layers $layers --iters 3 --warmup 1 "$@"
  cp $(ls $R/gpurun_out/pmc_r02t_${tag}/*/*counter_collection.csv | head -1) $O/${tag}_counter_collection.csv
  echo "pass $tag done"
}
for cfg in "f32|--batch 256" "bf16_1x224|--batch 512 --dtype bf16" "bf16_0.5x160|--batch 512 --dtype bf16 --alpha 0.5 --res 160"; do
  key=${cfg%%|*}; args=${cfg#*|}
  pass ${key}_pwF FETCH_SIZE $PW $args
  pass ${key}_pwW WRITE_SIZE $PW $args
  pass ${key}_dwF FETCH_SIZE $DW $args
  pass ${key}_dwW WRITE_SIZE $DW $args
  python3 $R/tools/make_traffic.py $R/gpurun_out/pmc_r02t_${key}_pwF $R/gpurun_out/pmc_r02t_${key}_pwW $PW 4 > $O/${key}_pw.json
  python3 $R/tools/make_traffic.py $R/gpurun_out/pmc_r02t_${key}_dwF $R/gpurun_out/pmc_r02t_${key}_dwW $DW 4 > $O/${key}_dw.json
done
python3 $R/tools/make_traffic.py assemble $O $SHA > $O/traffic.json
echo "traffic.json written"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --no-cpu-variants > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err || true
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv || true
echo "stats done"
