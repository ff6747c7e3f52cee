#!/usr/bin/env bash
# Regenerates the artefacts under profiles/ on a GPU box (run from the repository root, e.g. through gpurun):
#   the VALU micro-benchmarks, a rocprofv3 kernel trace of the default bench command, the PMC passes (each counter group
#   in its own run, as the MI355X guide prescribes), the roofline inputs bench.py reads, then the bench lines.
# usage: tools/profile_round.sh <round-tag, e.g. r02> [scratch dir, default gpurun_out]
set -euo pipefail
TAG=${1:?round tag}
OUT=${2:-gpurun_out}
R=$(pwd)
# gpurun only carries gpurun_out/ back: everything is written to $OUT/profiles_new and copied to profiles/ afterwards
# (cp gpurun_out/profiles_new/* profiles/); bench.py's roofline block needs the inputs file in profiles/ while it runs
PROF="$OUT/profiles_new"
mkdir -p "$OUT" "$PROF"
[ -x tools/ubench/valu_rates ] && [ tools/ubench/valu_rates -nt tools/ubench/valu_rates.hip ] || \
    /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/ubench/valu_rates.hip -o tools/ubench/valu_rates
./tools/ubench/valu_rates > "$PROF/${TAG}_valu_rates.txt"
# the probe trip in its loop structure (issue cycles against the latency of the dependent chain; one ray per lane against two)
[ -x tools/ubench/trip_rates ] && [ tools/ubench/trip_rates -nt tools/ubench/trip_rates.hip ] || \
    /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/ubench/trip_rates.hip -o tools/ubench/trip_rates
./tools/ubench/trip_rates > "$PROF/${TAG}_trip_rates.txt"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/prof" -o t -- python3 "$R/bench.py" --steps 10 --no-cpu-baseline --no-parity-check > "$R/$OUT/prof.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$R/$OUT/pmc_fetch" -o f -- python3 "$R/bench.py" --steps 6 --no-cpu-baseline --no-parity-check > "$R/$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$R/$OUT/pmc_write" -o w -- python3 "$R/bench.py" --steps 6 --no-cpu-baseline --no-parity-check > "$R/$OUT/pmc_write.log" 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES --output-format csv -d "$R/$OUT/pmc_sq" -o s -- python3 "$R/bench.py" --steps 6 --no-cpu-baseline --no-parity-check > "$R/$OUT/pmc_sq.log" 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d "$R/$OUT/pmc_sq2" -o s2 -- python3 "$R/bench.py" --steps 6 --no-cpu-baseline --no-parity-check > "$R/$OUT/pmc_sq2.log" 2>&1
rocprofv3 --kernel-trace --output-format csv -d "$R/$OUT/prof_256k" -o t -- python3 "$R/bench.py" --steps 6 --no-cpu-baseline --no-parity-check --particles-per-gpu 262144 > "$R/$OUT/prof_256k.log" 2>&1
cd "$R"
# one steady-state update as a timeline (every launch, its start, duration and the idle time before it): 4M and 262 144 particles
python3 tools/update_timeline.py "$OUT/prof/t_kernel_trace.csv" k_resample_motion "$PROF/${TAG}_timeline_4m.md" > /dev/null
python3 tools/update_timeline.py "$OUT/prof_256k/t_kernel_trace.csv" k_resample_motion "$PROF/${TAG}_timeline_256k.md" > /dev/null
# keep only the rows of the ray-stage kernels in the committed PMC files (the full CSVs are tens of MB)
for p in fetch:f write:w sq:s sq2:s2; do
  d=${p%%:*}; o=${p##*:}
  python3 - "$OUT/pmc_$d/${o}_counter_collection.csv" "$PROF/${TAG}_pmc_$d.csv" <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
keep = [rows[0]] + [r for r in rows[1:] if any(k in r[8] for k in ("k_rays_", "k_combine", "k_resample", "k_sort", "k_weights"))]
csv.writer(open(sys.argv[2], "w")).writerows(keep)
PY
done
python3 tools/summarize_rocprof.py "$OUT/prof/t_kernel_trace.csv" "$PROF/${TAG}_kernel_trace_summary.md" \
    --pmc-fetch "$OUT/pmc_fetch/f_counter_collection.csv" --pmc-write "$OUT/pmc_write/w_counter_collection.csv" \
    --pmc-sq "$OUT/pmc_sq/s_counter_collection.csv" --pmc-sq2 "$OUT/pmc_sq2/s2_counter_collection.csv" > /dev/null
cp "$OUT/prof/t_kernel_stats.csv" "$PROF/${TAG}_kernel_stats.csv"
python3 tools/roofline_inputs.py "$TAG" "$PROF/${TAG}_pmc_sq.csv" "$PROF/${TAG}_pmc_sq2.csv" "$PROF/${TAG}_pmc_fetch.csv" \
    "$PROF/${TAG}_pmc_write.csv" "$PROF/${TAG}_valu_rates.txt" 4194304 1081 "$PROF" > "$OUT/roofline_inputs.log"
cp "$PROF/${TAG}_roofline_inputs.json" "profiles/${TAG}_roofline_inputs.json"
# per-kernel achieved bandwidth of the streaming kernels (algorithmic bytes / steady-state duration) + their VALU issue time
python3 tools/kernel_bandwidth.py "$PROF/${TAG}_kernel_trace_summary.json" 4194304 "$PROF/${TAG}_kernel_bandwidth.md" "$PROF/${TAG}_pmc_sq.csv" > /dev/null
python bench.py > "$OUT/b_4m.json"
python bench.py --no-cpu-baseline --particles-per-gpu 262144 > "$OUT/b_256k.json"
python bench.py --no-cpu-baseline --map levine > "$OUT/b_levine.json"
python bench.py --no-cpu-baseline --regime global > "$OUT/b_global.json"
python bench.py --no-cpu-baseline --regime global --map levine --steps 5 > "$OUT/b_global_levine.json"
python bench.py --no-cpu-baseline --resample systematic > "$OUT/b_sys.json"
python bench.py --no-cpu-baseline --map fine025 --steps 10 > "$OUT/b_fine025.json"
for n in 4m 256k levine global global_levine sys fine025; do tail -n 1 "$OUT/b_$n.json" > "$PROF/${TAG}_bench_$n.json"; done
echo "$PROF/${TAG}_* refreshed"
