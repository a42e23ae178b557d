#!/bin/bash
# Kernel stats and HBM-side counters of the single-precision library's update pass (32^4 SU(3)), one rocprofv3 run
# per counter group.  usage: bash tools/profile_f32.sh <outdir>
set -u
OUT=$(realpath -m "$1"); mkdir -p "$OUT"
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cat > /tmp/f32.py <<PY
import sys; sys.path.insert(0, "$R")
import qcdgpu_amd as q
c = q.Context(group=3, L=(32,32,32,32), beta=6.0, precision=1); c.init_lattice(); c.sweep(60); c.synchronize()   # the first ten sweeps from the cold start run 15 % slower: sixty dilute them
PY
timeout -s KILL 120 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o p --output-format csv -- python3 /tmp/f32.py > "$OUT/trace.log" 2>&1 || { echo "trace failed"; exit 1; }
pass() {
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" -d "$OUT/$name" -o p --output-format csv -- python3 /tmp/f32.py > "$OUT/$name.log" 2>&1 || { echo "pmc pass $name failed"; tail -3 "$OUT/$name.log"; return 1; }
  python3 "$R/tools/summarize_pmc.py" "$OUT/$name" > "$OUT/$name.json"
}
pass fetch FETCH_SIZE && pass write WRITE_SIZE &&
pass insts SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES &&
pass cache TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE && echo "f32 profile ok"
