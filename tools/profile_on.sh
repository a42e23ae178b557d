#!/bin/bash
# rocprofv3 over the O(N) 64^4 timing run: kernel trace + three PMC passes, one counter group per run.
# usage (on the GPU box, from the repo root): bash tools/profile_on.sh <outdir>
set -u
OUT=$(realpath -m "$1"); mkdir -p "$OUT"
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o on --output-format csv -- \
    python3 "$ROOT/tools/time_on.py" 10 > "$OUT/trace.log" 2>&1 || { echo "trace failed"; tail -3 "$OUT/trace.log"; exit 1; }
run() {
  name=$1; shift
  timeout -s KILL 200 rocprofv3 --pmc "$@" -d "$OUT/$name" -o "$name" --output-format csv -- \
      python3 "$ROOT/tools/time_on.py" 3 > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -3 "$OUT/$name.log"; return 1; }
  python3 "$ROOT/tools/summarize_pmc.py" "$OUT/$name" > "$OUT/$name.json" && echo "pass $name ok"
}
run insts SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES &&
run busy SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA &&
run fetch FETCH_SIZE && run write WRITE_SIZE
