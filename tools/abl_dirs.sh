#!/bin/bash
# Average duration of the update kernel per direction for every diagnostic build under qcdgpu_amd/lib/abl
# (rocprofv3 --kernel-trace --stats of a short 32^4 SU(3) run each).  usage: bash tools/abl_dirs.sh <outdir> [lib ...]
set -u
OUT=$(realpath -m "$1"); shift
mkdir -p "$OUT"
R=${GRAFT_REPO_ROOT:-$(pwd)}
LIBS=("$@"); [ ${#LIBS[@]} -eq 0 ] && LIBS=($R/qcdgpu_amd/lib/abl/*.so)
cd /tmp && export TMPDIR=/tmp
cat > /tmp/one.py <<PY
import sys; sys.path.insert(0, "$R")
import qcdgpu_amd as q
c = q.Context(group=3, L=(32,32,32,32), beta=6.0); c.init_lattice(); c.sweep(40); c.synchronize()
PY
for lib in "${LIBS[@]}"; do
  name=$(basename "$lib" .so)
  export QCDGPU_AMD_LIB=$lib
  timeout -s KILL 120 rocprofv3 --kernel-trace --stats -d "$OUT/$name" -o p --output-format csv -- python3 /tmp/one.py > "$OUT/$name.log" 2>&1 || { echo "$name failed"; tail -3 "$OUT/$name.log"; exit 1; }
  python3 - "$OUT/$name" "$name" <<'PY'
import csv, glob, sys, re, collections
rows = collections.defaultdict(list)
for path in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        m = re.search(r"k_update_gh<(\d)", r["Kernel_Name"])
        if m: rows[int(m.group(1))].append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
avg = {}
for d, v in rows.items():
    v.sort()
    v = [x[1] for x in v[len(v) // 2:]]     # the second half of the run: thermalised
    avg[d] = sum(v) / len(v)
tot = sum(avg.values()) / max(1, len(avg))
print("%-22s %s  mean %.1f us" % (sys.argv[2], "  ".join("d%d %.1f" % (d, avg[d]) for d in sorted(avg)), tot), flush=True)
PY
done
