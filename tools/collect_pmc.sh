#!/bin/bash
# Collects rocprofv3 PMC passes for the bench workload, one counter group per run.
# usage (on the GPU box, from the repo root): bash tools/collect_pmc.sh <outdir> [bench args...]
set -u
OUT=$1; shift
mkdir -p "$OUT"
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
run() {
  name=$1; shift
  timeout -s KILL 180 rocprofv3 --pmc "$@" -d "$OUT/$name" -o "$name" --output-format csv -- \
      python3 "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu $BENCH_ARGS > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -3 "$OUT/$name.log"; return 1; }
  echo "pass $name ok"
}
BENCH_ARGS="${*:-}"
run fetch FETCH_SIZE &&
run write WRITE_SIZE &&
run insts SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_FLAT SQ_WAVE_CYCLES &&
run busy SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA &&
run cache TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE
