#!/bin/bash
# PMC counters of the measurement kernels of a short 32^4 SU(3) run, one rocprofv3 run per counter group.
# usage: bash tools/plaq_pmc.sh <outdir> <label> [VAR=value ...]   (the variables go into the environment)
set -u
OUT=$(realpath -m "$1"); LABEL=$2; shift 2
mkdir -p "$OUT"
for kv in "$@"; do export "$kv"; done
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cat > /tmp/one.py <<PY
import sys; sys.path.insert(0, "$R")
import qcdgpu_amd as q
c = q.Context(group=3, L=(32,32,32,32), beta=6.0); c.init_lattice(); c.sweep(1)
for _ in range(4): c.plaquette(); c.fmunu(0, 3, 8); c.wilson_loop(2, 2)
PY
pass() {
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" -d "$OUT/$LABEL.$name" -o p --output-format csv -- python3 /tmp/one.py > "$OUT/$LABEL.$name.log" 2>&1 || { echo "pmc pass $name failed"; tail -3 "$OUT/$LABEL.$name.log"; return 1; }
}
pass insts SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES &&
pass busy SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM &&
pass cache TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE || exit 1
python3 - <<PY
import csv, collections, glob
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob("$OUT/$LABEL.*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"]
        for name in ("k_plaquette", "k_fmunu", "k_wilson"):
            if name in k: acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
for name, d in acc.items():
    m = {k: sum(v) / len(v) for k, v in d.items()}
    print("$LABEL %-12s" % name, "  ".join("%s=%.4gM" % (k.replace("SQ_", "").replace("_sum", ""), v / 1e6) for k, v in sorted(m.items())))
PY
