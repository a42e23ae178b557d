#!/bin/bash
# Pass time per direction (rocprofv3 kernel trace, 40 thermalised-ish sweeps) for every diagnostic build under qcdgpu_amd/lib/abl
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cat > /tmp/one.py <<PY
import sys; sys.path.insert(0, "$R")
import qcdgpu_amd as q
c = q.Context(group=3, L=(32,32,32,32), beta=6.0); c.init_lattice(); c.sweep(40)
PY
for lib in $R/qcdgpu_amd/lib/abl/*.so; do
  export QCDGPU_AMD_LIB=$lib
  n=$(basename $lib .so)
  timeout -k 10 120 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/dir_$n -o p --output-format csv -- python3 /tmp/one.py > /dev/null 2>&1 || exit 1
  echo "== $n"; grep k_update_gh $R/gpurun_out/dir_$n/p_kernel_stats.csv | cut -d, -f1,4,6 | sort
done
