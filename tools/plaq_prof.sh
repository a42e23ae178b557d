#!/bin/bash
# Kernel time and L2 hits/misses of the measurement kernels at 32^4 SU(3): tools/plaq_prof.sh [lib.so]
R=${GRAFT_REPO_ROOT:-$(pwd)}
[ -n "$1" ] && export QCDGPU_AMD_LIB=$(realpath "$1")
cd /tmp && export TMPDIR=/tmp
cat > /tmp/plaq.py <<PY
import sys; sys.path.insert(0, "$R")
import qcdgpu_amd as q
c = q.Context(group=3, L=(32,32,32,32), beta=6.0); c.init_lattice(); c.sweep(1)
for _ in range(5): c.plaquette(); c.fmunu(0, 3, 8); c.fmunu(1, 3, 8); c.wilson_loop(2, 2); c.polyakov()
PY
O=$R/gpurun_out/plaqprof; rm -rf $O
timeout -s KILL 120 rocprofv3 --kernel-trace --stats -d $O/kt -o p --output-format csv -- python3 /tmp/plaq.py > /dev/null 2>&1 || { echo "trace failed"; exit 1; }
grep -E "k_plaquette|k_fmunu|k_wilson|k_polyakov" $O/kt/p_kernel_stats.csv | awk -F'","|",|,"' '{print $1, $2, $3, $4}' | cut -c1-60,150-
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $O/tcc -o p --output-format csv -- python3 /tmp/plaq.py > /dev/null 2>&1 || { echo "pmc failed"; exit 1; }
python3 - <<PY
import csv, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open("$O/tcc/p_counter_collection.csv")):
    k = row["Kernel_Name"]
    for name in ("k_plaquette", "k_fmunu_march<256, 8, 0>", "k_fmunu_march<256, 8, 1>", "k_wilson", "k_polyakov"):
        if name in k: acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
for name, d in acc.items():
    print("%-12s" % name, "  ".join("%s=%.3fM" % (k, sum(v) / len(v) / 1e6) for k, v in sorted(d.items())))
PY
