#!/bin/bash
# L2 hits and misses of the update pass, and its duration, against the number of resident workgroups
# per launch (QCDGPU_AMD_GRID): the fewer sites are in flight on an XCD, the more of what they share
# is still in its 4 MiB L2.   usage: bash tools/grid_pmc.sh <outdir> [grids...]
set -u
OUT=$(realpath -m "$1"); shift
mkdir -p "$OUT"
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cat > /tmp/one.py <<PY
import sys; sys.path.insert(0, "$R")
import qcdgpu_amd as q
c = q.Context(group=3, L=(32,32,32,32), beta=6.0); c.init_lattice(); c.sweep(2); c.synchronize()
PY
cat > /tmp/tm.py <<PY
import sys; sys.path.insert(0, "$R")
import qcdgpu_amd as q
c = q.Context(group=3, L=(32,32,32,32), beta=6.0); c.init_lattice(); c.sweep(3)
best = 1e9
for _ in range(3):
    ms, n = c.timed_sweeps(5); best = min(best, ms / n)
print("%.1f" % (1e3 * best))
PY
for n in "${@:-256 512 1024 2048}"; do
  export QCDGPU_AMD_GRID=$n
  t=$(timeout -k 10 100 python3 /tmp/tm.py 2>/dev/null) || { echo "timing failed at grid $n"; exit 1; }
  timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d "$OUT/g$n" -o p --output-format csv -- python3 /tmp/one.py > "$OUT/g$n.log" 2>&1 || { echo "pmc failed at grid $n"; exit 1; }
  python3 - <<PY
import csv, collections, glob
acc = collections.defaultdict(list)
for path in glob.glob("$OUT/g$n/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        if "k_update<3" in row["Kernel_Name"] and "redo" not in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in acc.items()}
print("grid %5s  %7s us/pass   TCC_HIT %.2fM  TCC_MISS %.2fM" % ("$n", "$t", m.get("TCC_HIT_sum", 0) / 1e6, m.get("TCC_MISS_sum", 0) / 1e6), flush=True)
PY
done
