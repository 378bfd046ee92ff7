set -e
mkdir -p gpurun_out/x6
python tools/layer_bench.py --iters 20 --tune pw_emul=6 --custom-pw "49152,512,512;50176,512,512;65536,512,512;12288,1024,1024;12544,1024,1024;16384,1024,1024;196608,256,256;200704,256,256" > gpurun_out/x6/tail.txt 2>&1
