#!/bin/bash
# tools/r02_final.sh — the evidence set of round 2 (run on the GPU box from the repo root); outputs under gpurun_out/r02_final/
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02_final
mkdir -p $O
cd $R
python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
python bench.py > $O/bench_f32.json 2> $O/bench_f32.err; echo "f32 rc=$?"
python bench.py --dtype bf16 --batch 512 --no-cpu-variants > $O/bench_bf16_1.0x224_b512.json 2>> $O/bench_f32.err; echo "bf16 1.0 rc=$?"
python bench.py --dtype bf16 --batch 512 --alpha 0.5 --res 160 --no-cpu-variants > $O/bench_bf16_0.5x160_b512.json 2>> $O/bench_f32.err; echo "bf16 0.5 rc=$?"
python bench.py --batch 512 --alpha 0.5 --res 160 --no-cpu-variants > $O/bench_f32_0.5x160_b512.json 2>> $O/bench_f32.err
python bench.py --streams 1 --no-cpu-baseline --no-unfused-stages > $O/bench_f32_streams1.json 2>> $O/bench_f32.err
python bench.py --pw-emul 6 --no-cpu-variants > $O/bench_f32_pw_emul6.json 2>> $O/bench_f32.err; echo "f32 pw_emul 6 rc=$?"
python bench.py --streams 2 --no-cpu-baseline --no-unfused-stages --no-profile > $O/bench_f32_streams2_noprofile.json 2>> $O/bench_f32.err
for b in 1 8 32 64 128 256 512 1024; do
  python bench.py --batch $b --steps 50 --warmup 10 --streams 1 --no-cpu-baseline --no-unfused-stages --no-profile > /tmp/bs.json 2>/dev/null
  python - $b <<'PY'
import json,sys
d=json.load(open('/tmp/bs.json'))
print("batch %4s  %9.0f img/s  %.4f ms/step  median %.4f" % (sys.argv[1], d["value"], d["ms_per_step"], d["step_ms"]["median"]))
PY
done > $O/batch_sweep.txt
cat $O/batch_sweep.txt
$R/cnn-mobilenet-v1-implementation-on-aws-fpga-using-opencl_amd/mobilenet --gpus 1 --batch 256 --synthetic 1 --steps 20 --warmup 5 > $O/c_host_gpus1.txt 2>&1; $R/cnn-mobilenet-v1-implementation-on-aws-fpga-using-opencl_amd/mobilenet --gpus 1 --batch 256 --synthetic 1 --steps 20 --warmup 5 --streams 2 >> $O/c_host_gpus1.txt 2>&1; $R/cnn-mobilenet-v1-implementation-on-aws-fpga-using-opencl_amd/mobilenet --gpus 1 --batch 256 --synthetic 1 --steps 20 --warmup 5 --streams 2 --pw-emul 6 >> $O/c_host_gpus1.txt 2>&1; tail -6 $O/c_host_gpus1.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --streams 1 --no-cpu-baseline --no-unfused-stages > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_bf16 -- python3 $R/bench.py --dtype bf16 --batch 512 --no-cpu-baseline --no-unfused-stages > $O/bench_bf16_under_rocprof.json 2>> $O/bench_under_rocprof.err
cp $(ls $O/stats_bf16/*/*kernel_stats.csv | head -1) $O/kernel_stats_bf16.csv
rm -rf $O/stats $O/stats_bf16
echo done
