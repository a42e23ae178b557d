#!/bin/bash
# dynamic VALU instruction counts of the diagnostic builds (rocprofv3 PMC, one run per build)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cat > /tmp/one.py <<PY
import sys; sys.path.insert(0, "$R")
import qcdgpu_amd as q
c = q.Context(group=3, L=(32,32,32,32), beta=6.0); c.init_lattice(); c.sweep(2); c.synchronize()
PY
for lib in $R/qcdgpu_amd/lib/abl/*.so; do
  n=$(basename $lib .so)
  QCDGPU_AMD_LIB=$lib timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_LDS -d $R/gpurun_out/ablpmc/$n -o p --output-format csv -- python3 /tmp/one.py > /dev/null 2>&1
  python3 - <<PY
import csv, collections
acc = collections.defaultdict(list)
for row in csv.DictReader(open("$R/gpurun_out/ablpmc/$n/p_counter_collection.csv")):
    if "k_update<3, 0" in row["Kernel_Name"]:
        acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
w = sum(acc["SQ_WAVES"]) / len(acc["SQ_WAVES"])
print("%-22s" % "$n", "  ".join("%s/wave=%.0f" % (k[9:] if k.startswith("SQ_INSTS_") else k, sum(v) / len(v) / w) for k, v in sorted(acc.items()) if k != "SQ_WAVES"))
PY
done
