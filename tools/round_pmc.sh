#!/bin/bash
# L2 misses per block of 64 sites of update launches that cover `n` x-slices of a 32^4 lattice
# (8 slices = 2048 blocks = one round of the resident wavefronts, all started together), against
# the full pass (32 slices, four rounds).   usage: bash tools/round_pmc.sh <outdir> [VAR=value ...]
set -u
OUT=$(realpath -m "$1"); shift
mkdir -p "$OUT"
for kv in "$@"; do export "$kv"; done
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for n in 4 8 16 32; do
cat > /tmp/one.py <<PY
import sys; sys.path.insert(0, "$R")
import qcdgpu_amd as q
c = q.Context(group=3, L=(32,32,32,32), beta=6.0); c.init_lattice(); c.sweep(2); c.synchronize()
for rep in range(2):
    for d in range(4):
        for x0 in range(0, 32, $n):
            c.update_range(1, d, x0, $n)
c.synchronize()
PY
  timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d "$OUT/n$n" -o p --output-format csv -- python3 /tmp/one.py > "$OUT/n$n.log" 2>&1 || { echo "pmc failed at $n"; exit 1; }
  python3 - <<PY
import csv, collections, glob
acc = collections.defaultdict(list)
rows = [r for path in glob.glob("$OUT/n$n/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(path))]
rows = [r for r in rows if "k_update" in r["Kernel_Name"] and "redo" not in r["Kernel_Name"]]
# the first 16 k_update dispatches per counter are the two warm-up sweeps (full passes)
by = collections.defaultdict(list)
for r in rows: by[r["Counter_Name"]].append(float(r["Counter_Value"]))
nblk = $n * 256
for k, v in sorted(by.items()):
    part = v[16:]
    print("slices %2d (%5d blocks per launch, %3d launches): %s per block = %.0f   (warm-up full passes: %.0f)" % ($n, nblk, len(part), k, sum(part) / len(part) / nblk, sum(v[:16]) / 16 / 8192), flush=True)
PY
done
