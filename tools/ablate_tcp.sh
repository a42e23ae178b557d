#!/bin/bash
# L1->L2 read requests and L2 hits/misses of k_update for the diagnostic builds (one rocprofv3 PMC run
# per build and counter set): tools/ablate_tcp.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cat > /tmp/one.py <<PY
import sys; sys.path.insert(0, "$R")
import qcdgpu_amd as q
c = q.Context(group=3, L=(32,32,32,32), beta=6.0); c.init_lattice(); c.sweep(2); c.synchronize()
PY
for lib in $R/qcdgpu_amd/lib/abl/*.so; do
  n=$(basename $lib .so)
  for set in "TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum" "TCC_HIT_sum TCC_MISS_sum"; do
    tag=$(echo $set | cut -c1-7)
    QCDGPU_AMD_LIB=$lib timeout -s KILL 120 rocprofv3 --pmc $set -d $R/gpurun_out/abltcp/$n$tag -o p --output-format csv -- python3 /tmp/one.py > /dev/null 2>&1 || { echo "$n $set failed"; exit 1; }
    python3 - <<PY
import csv, collections
acc = collections.defaultdict(list)
for row in csv.DictReader(open("$R/gpurun_out/abltcp/$n$tag/p_counter_collection.csv")):
    if "k_update<3, 0" in row["Kernel_Name"]:
        acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
print("%-14s" % "$n", "  ".join("%s=%.3fM" % (k, sum(v) / len(v) / 1e6) for k, v in sorted(acc.items())))
PY
  done
done
