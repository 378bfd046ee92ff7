#!/bin/bash
# tools/r06_traffic.sh <git-sha> — PMC traffic passes of round 6 (FETCH_SIZE / WRITE_SIZE in separate rocprofv3 --pmc passes over tools/layer_bench.py) -> gpurun_out/r06_traffic/traffic.json
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_traffic
SHA=${1:-unknown}
mkdir -p $O
cd $R
PW=3,5,7,9,11,13,15,17,19,21,23,25,27
DW=2,4,6,8,10,12,14,16,18,20,22,24,26
pass() {   # tag counter layer-list extra-args...
  local tag=$1 ctr=$2 layers=$3; shift 3
  bash $R/tools/pmc_pass.sh r06t_${tag} $ctr -- --layers $layers --iters 3 --warmup 1 "$@" > $O/pmc_${tag}.log 2>&1
  cp $(ls $R/gpurun_out/pmc_r06t_${tag}/*/*counter_collection.csv | head -1) $O/${tag}_counter_collection.csv
  echo "pass $tag done"
}
for cfg in "f32|--batch 256" "bf16_1x224|--batch 512 --dtype bf16" "bf16_0.5x160|--batch 512 --dtype bf16 --alpha 0.5 --res 160"; do
  key=${cfg%%|*}; args=${cfg#*|}
  pass ${key}_pwF FETCH_SIZE $PW $args
  pass ${key}_pwW WRITE_SIZE $PW $args
  pass ${key}_dwF FETCH_SIZE $DW $args
  pass ${key}_dwW WRITE_SIZE $DW $args
  python3 $R/tools/make_traffic.py $R/gpurun_out/pmc_r06t_${key}_pwF $R/gpurun_out/pmc_r06t_${key}_pwW $PW 4 > $O/${key}_pw.json
  python3 $R/tools/make_traffic.py $R/gpurun_out/pmc_r06t_${key}_dwF $R/gpurun_out/pmc_r06t_${key}_dwW $DW 4 > $O/${key}_dw.json
done
python3 $R/tools/make_traffic.py assemble $O $SHA > $O/traffic.json
rm -rf $R/gpurun_out/pmc_r06t_*
echo "traffic.json written"; head -c 600 $O/traffic.json
