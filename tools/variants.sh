#!/bin/bash
# Builds experimental variants of the library into qcdgpu_amd/lib/abl: tools/variants.sh name:"flags" ...
# (time them on the GPU with tools/ablate_time.py)
set -e
cd "$(dirname "$0")/../qcdgpu_amd/csrc"
mkdir -p ../lib/abl
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -shared -fPIC $flags -o ../lib/abl/lib_$name.so engine.hip &
done
wait
ls ../lib/abl
