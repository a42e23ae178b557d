#!/bin/bash
# VALU instruction counts of the SU(2) 32^4 heat-bath and overrelaxation passes (one rocprofv3 --pmc run)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cat > /tmp/su2one.py <<PY
import sys; sys.path.insert(0, "$R")
import qcdgpu_amd as q
c = q.Context(group=2, L=(32,32,32,32), beta=2.5); c.init_lattice(); c.sweep(4); c.overrelax_sweep(2)
PY
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VMEM_RD -d $R/gpurun_out/su2pmc -o p --output-format csv -- python3 /tmp/su2one.py > /dev/null 2>&1 || exit 1
python3 - <<PY
import csv, collections, glob
acc = collections.defaultdict(list)
for path in glob.glob("$R/gpurun_out/su2pmc/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"]
        if "k_update" in k or "k_overrelax" in k: acc[(k.split("(")[0][:60], row["Counter_Name"])].append(float(row["Counter_Value"]))
for (k, c), v in sorted(acc.items()): print("%-62s %-22s %.3f M  (%d launches)" % (k, c, sum(v) / len(v) / 1e6, len(v)))
PY
