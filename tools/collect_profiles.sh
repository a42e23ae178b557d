#!/bin/bash
# One GPU-box call that produces everything kept under profiles/: kernel-trace stats of the bench
# command, the PMC passes, and the known-byte calibration of FETCH_SIZE on k_reunit.
# usage: bash tools/collect_profiles.sh <outdir under gpurun_out>
set -u
OUT=$(realpath -m "$1"); mkdir -p "$OUT"
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
# the command the round-end bench uses (--steps 20 --warmup 5), so that the kernel averages can be held against its line
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o bench --output-format csv -- \
    python3 "$ROOT/bench.py" --steps 20 --warmup 5 --no-cpu > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.log" || { echo "stats failed"; tail -5 "$OUT/stats.log"; exit 1; }
python3 "$ROOT/tools/kernel_stats_split.py" "$(find "$OUT/stats" -name 'bench_kernel_trace.csv' | head -1)" 5 20 5 > "$OUT/kernel_stats_split.json" || exit 1
echo "stats ok"
bash "$ROOT/tools/collect_pmc.sh" "$OUT/pmc" || exit 1
python3 "$ROOT/tools/summarize_pmc.py" "$OUT/pmc" > "$OUT/pmc_summary.json" || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 120 rocprofv3 --pmc $c -d "$OUT/calib_$c" -o calib --output-format csv -- \
      python3 "$ROOT/tools/calib_reunit.py" > "$OUT/calib_$c.log" 2>&1 || { echo "calib $c failed"; exit 1; }
done
bash "$ROOT/tools/profile_on.sh" "$OUT/on" || exit 1
bash "$ROOT/tools/profile_f32.sh" "$OUT/f32" || exit 1
python3 "$ROOT/tools/summarize_pmc.py" "$OUT/calib_FETCH_SIZE" > "$OUT/calib_fetch.json" &&
python3 "$ROOT/tools/summarize_pmc.py" "$OUT/calib_WRITE_SIZE" > "$OUT/calib_write.json" && echo "calib ok"
