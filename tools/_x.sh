set -e
mkdir -p gpurun_out/x6
python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "emul" > gpurun_out/x6/pytest_emul.log 2>&1 || { tail -30 gpurun_out/x6/pytest_emul.log; exit 1; }
tail -3 gpurun_out/x6/pytest_emul.log
python bench.py > gpurun_out/x6/bench_default.json 2> gpurun_out/x6/bench_default.err
python bench.py --pw-emul 6 > gpurun_out/x6/bench_emul6.json 2> gpurun_out/x6/bench_emul6.err
python bench.py --graph --pw-emul 6 --no-cpu-baseline > gpurun_out/x6/bench_emul6_graph.json 2> gpurun_out/x6/bench_emul6_graph.err
