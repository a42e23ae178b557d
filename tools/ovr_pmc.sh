#!/bin/bash
# L2 misses and fetched / written bytes of the 32^4 SU(3) overrelaxation pass beside the heat-bath pass (two rocprofv3 --pmc runs)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cat > /tmp/ovrone.py <<PY
import sys; sys.path.insert(0, "$R")
import qcdgpu_amd as q
c = q.Context(group=3, L=(32,32,32,32), beta=6.0); c.init_lattice(); c.sweep(6); c.overrelax_sweep(4)
PY
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $R/gpurun_out/ovrpmc1 -o p --output-format csv -- python3 /tmp/ovrone.py > /dev/null 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/ovrpmc2 -o p --output-format csv -- python3 /tmp/ovrone.py > /dev/null 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/ovrpmc3 -o p --output-format csv -- python3 /tmp/ovrone.py > /dev/null 2>&1 || exit 1
python3 - <<PY
import csv, collections, glob
acc = collections.defaultdict(list)
for path in glob.glob("$R/gpurun_out/ovrpmc*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"]
        if "k_update_gh" in k:
            ovr = "true>" in k.replace(" ", "")
            acc[("overrelaxation" if ovr else "heat-bath", row["Counter_Name"])].append(float(row["Counter_Value"]))
for (k, c), v in sorted(acc.items()): print("%-16s %-14s %.3f M  (%d launches)" % (k, c, sum(v) / len(v) / 1e6, len(v)))
PY
