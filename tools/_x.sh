set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/x6
cd $R
python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "emul_split" > gpurun_out/x6/pytest_emul3.log 2>&1 || { tail -30 gpurun_out/x6/pytest_emul3.log; exit 1; }
tail -2 gpurun_out/x6/pytest_emul3.log
python tools/layer_bench.py --layers 13,15,25,27 --iters 20 --tune pw_emul=6 --tune pw_tile=11,16,17 > gpurun_out/x6/tiles_e.txt 2>&1
PW=3,5,7,9,11,13,15,17,19,21,23,25,27
bash tools/pmc_pass.sh x6trF FETCH_SIZE -- --layers $PW --iters 3 --warmup 1 --tune pw_emul=6
bash tools/pmc_pass.sh x6trW WRITE_SIZE -- --layers $PW --iters 3 --warmup 1 --tune pw_emul=6
python3 tools/make_traffic.py gpurun_out/pmc_x6trF gpurun_out/pmc_x6trW $PW 4 > gpurun_out/x6/traffic_pw_emul6.json
cp $(ls gpurun_out/pmc_x6trF/*/*counter_collection.csv | head -1) gpurun_out/x6/pw_emul6_FETCH_counter_collection.csv
cp $(ls gpurun_out/pmc_x6trW/*/*counter_collection.csv | head -1) gpurun_out/x6/pw_emul6_WRITE_counter_collection.csv
