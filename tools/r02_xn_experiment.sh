set -x
mkdir -p gpurun_out/r02j
python -m pytest tests -m gpu -x -q -k "pointwise or headline_fp32 or per_layer" > gpurun_out/r02j/pytest.log 2>&1; tail -3 gpurun_out/r02j/pytest.log
python tools/layer_bench.py --layers 13,15,25,27 --iters 20 --tune pw_xn=1,2,4 > gpurun_out/r02j/ab_f32.txt 2>&1
python tools/layer_bench.py --layers 13,15,25,27 --iters 20 --batch 512 --dtype bf16 --tune pw_xn=1,2,4 > gpurun_out/r02j/ab_bf16.txt 2>&1
cat gpurun_out/r02j/ab_f32.txt gpurun_out/r02j/ab_bf16.txt
for xn in 1 2; do
  bash tools/pmc_pass.sh r02j_f32_xn${xn}_F FETCH_SIZE -- --layers 25,27 --iters 3 --warmup 1 --tune pw_xn=$xn
  bash tools/pmc_pass.sh r02j_f32_xn${xn}_W WRITE_SIZE -- --layers 25,27 --iters 3 --warmup 1 --tune pw_xn=$xn
  bash tools/pmc_pass.sh r02j_bf16_xn${xn}_F FETCH_SIZE -- --layers 25,27 --iters 3 --warmup 1 --batch 512 --dtype bf16 --tune pw_xn=$xn
done
python tools/pmc_summary.py gpurun_out/pmc_r02j_f32_xn1_F gpurun_out/pmc_r02j_f32_xn2_F gpurun_out/pmc_r02j_f32_xn1_W gpurun_out/pmc_r02j_f32_xn2_W gpurun_out/pmc_r02j_bf16_xn1_F gpurun_out/pmc_r02j_bf16_xn2_F > gpurun_out/r02j/pmc_summary.txt 2>&1
cat gpurun_out/r02j/pmc_summary.txt
