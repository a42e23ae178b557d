#!/bin/bash
# Diagnostic builds of the library with parts of k_update stubbed out (timing shares only; the
# results of these builds are wrong by construction).  Build here, run tools/ablate_time.py on the GPU.
set -e
cd "$(dirname "$0")/../qcdgpu_amd/csrc"
mkdir -p ../lib/abl
for v in BASE RNG HB STAPLE "RNG -DQCD_ABL_HB" "RNG -DQCD_ABL_HB -DQCD_ABL_STAPLE"; do
  name=$(echo "$v" | sed 's/ -DQCD_ABL_/_/g')
  flag=""; [ "$v" != BASE ] && flag="-DQCD_ABL_$v"
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -shared -fPIC $flag -o ../lib/abl/lib_$name.so engine.hip &
done
wait
ls -la ../lib/abl
