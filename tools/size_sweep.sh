set -e
mkdir -p gpurun_out/r4/sw
for n in 65536 131072 262144 524288 1048576 2097152 4194304 8388608 16777216 33554432; do
  python bench.py --particles-per-gpu $n --no-cpu-baseline --no-parity-check --steps 8 > gpurun_out/r4/sw/$n.json 2>/dev/null
done
python tools/small_configs.py > gpurun_out/r4/sw/small.txt 2>&1
