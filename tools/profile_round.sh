#!/usr/bin/env bash
# Regenerates the artefacts under profiles/ on a GPU box (run from the repository root, e.g. through gpurun):
#   bench lines of the five configurations, a rocprofv3 kernel trace of the default bench command and the PMC
#   passes (each counter group in its own run, as the MI355X guide prescribes), then the summaries.
# usage: tools/profile_round.sh <round-tag, e.g. r01> [scratch dir, default gpurun_out]
set -euo pipefail
TAG=${1:?round tag}
OUT=${2:-gpurun_out}
R=$(pwd)
mkdir -p "$OUT"
python bench.py > "$OUT/b_4m.json"
python bench.py --no-cpu-baseline --particles-per-gpu 262144 > "$OUT/b_256k.json"
python bench.py --no-cpu-baseline --map levine > "$OUT/b_levine.json"
python bench.py --no-cpu-baseline --regime global > "$OUT/b_global.json"
python bench.py --no-cpu-baseline --resample systematic > "$OUT/b_sys.json"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/prof" -o t -- python3 "$R/bench.py" --steps 10 --no-cpu-baseline > "$R/$OUT/prof.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$R/$OUT/pmc_fetch" -o f -- python3 "$R/bench.py" --steps 5 --no-cpu-baseline > "$R/$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$R/$OUT/pmc_write" -o w -- python3 "$R/bench.py" --steps 5 --no-cpu-baseline > "$R/$OUT/pmc_write.log" 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES --output-format csv -d "$R/$OUT/pmc_sq" -o s -- python3 "$R/bench.py" --steps 5 --no-cpu-baseline > "$R/$OUT/pmc_sq.log" 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD --output-format csv -d "$R/$OUT/pmc_sq2" -o s2 -- python3 "$R/bench.py" --steps 5 --no-cpu-baseline > "$R/$OUT/pmc_sq2.log" 2>&1
cd "$R"
for n in 4m 256k levine global sys; do tail -n 1 "$OUT/b_$n.json" > "profiles/${TAG}_bench_$n.json"; done
python tools/summarize_rocprof.py "$OUT/prof/t_kernel_trace.csv" "profiles/${TAG}_kernel_trace_summary.md" \
    --pmc-fetch "$OUT/pmc_fetch/f_counter_collection.csv" --pmc-write "$OUT/pmc_write/w_counter_collection.csv" \
    --pmc-sq "$OUT/pmc_sq/s_counter_collection.csv" --pmc-sq2 "$OUT/pmc_sq2/s2_counter_collection.csv" > /dev/null
cp "$OUT/prof/t_kernel_stats.csv" "profiles/${TAG}_kernel_stats.csv"
echo "profiles/${TAG}_* refreshed; update profiles/hbm_traffic.json from the FETCH_SIZE / WRITE_SIZE rows of the summary"
