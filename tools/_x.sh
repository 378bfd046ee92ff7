set -e
mkdir -p gpurun_out/x6
{ for e in 0 6 0 6; do echo "== pw_emul=$e"; python tools/block_bench.py --blocks 4,6,8,10 --tune pw_emul=$e; done; echo "== pw_emul=6 pw_tile=1 (128-column tiles only)"; python tools/block_bench.py --blocks 8,10 --tune pw_emul=6 --tune pw_tile=1; } > gpurun_out/x6/blocks.txt 2>&1
rm -f gpurun_out/x6/bench2.txt
for e in 0 6 0 6; do
python bench.py --no-cpu-baseline --no-unfused-stages --no-pw-emul-alt --pw-emul $e 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('emul $e', round(d['value']), round(d['ms_per_step'],4), {k:round(v['ms'],3) for k,v in d.get('stages',{}).items()})" >> gpurun_out/x6/bench2.txt
done
