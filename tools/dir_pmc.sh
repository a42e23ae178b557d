#!/bin/bash
# L2 hits and misses per update pass and direction for every diagnostic build under qcdgpu_amd/lib/abl
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cat > /tmp/one.py <<PY
import sys; sys.path.insert(0, "$R")
import qcdgpu_amd as q
c = q.Context(group=3, L=(32,32,32,32), beta=6.0); c.init_lattice(); c.sweep(6)
PY
for lib in $R/qcdgpu_amd/lib/abl/*.so; do
  export QCDGPU_AMD_LIB=$lib
  n=$(basename $lib .so)
  timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $R/gpurun_out/dpmc_$n -o p --output-format csv -- python3 /tmp/one.py > /dev/null 2>&1 || exit 1
  python3 - <<PY
import csv, collections, glob
acc = collections.defaultdict(list)
for path in glob.glob("$R/gpurun_out/dpmc_$n/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"]
        if "k_update_gh" in k: acc[(k.split("<")[1].split(",")[0], row["Counter_Name"])].append(float(row["Counter_Value"]))
print("== $n", "  ".join("dir %s %s %.2f M" % (d, c.replace("TCC_", "").replace("_sum", ""), sum(v) / len(v) / 1e6) for (d, c), v in sorted(acc.items())))
PY
done
