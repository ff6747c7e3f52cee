#!/usr/bin/env bash
# Particle-count sweep on one GPU (65 536 .. 33 554 432 x 1081 beams) and the small configurations.
# usage: tools/size_sweep.sh [out dir, default gpurun_out/sw]; every size keeps its own stderr log, a failed size is named and
# the sweep goes on (a following size may still be of interest), the script's exit code says whether any failed
OUT=${1:-gpurun_out/sw}
mkdir -p "$OUT"
failed=0
for n in 65536 131072 262144 524288 1048576 2097152 4194304 8388608 16777216 33554432; do
  if ! python bench.py --particles-per-gpu $n --no-cpu-baseline --no-parity-check --steps 8 > "$OUT/$n.json" 2> "$OUT/$n.err"; then
    echo "size_sweep: $n particles FAILED (see $OUT/$n.err)" >&2
    tail -n 3 "$OUT/$n.err" >&2
    failed=1
  fi
done
python tools/small_configs.py > "$OUT/small.txt" 2> "$OUT/small.err" || { echo "size_sweep: small_configs FAILED (see $OUT/small.err)" >&2; failed=1; }
exit $failed
