#!/bin/bash
# Builds experimental variants of the library into qcdgpu_amd/lib/abl: tools/variants.sh name:"flags" ...
# (time them on the GPU with tools/ablate_time.py)
set -e
cd "$(dirname "$0")/../qcdgpu_amd/csrc"
mkdir -p ../lib/abl
# the O(N) half of the library is the same in every variant
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -c on_engine.hip -o ../lib/abl/on_engine.o
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -c ring.hip -o ../lib/abl/ring.o
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  ( /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -Xarch_host -ffp-contract=off $flags -c engine.hip -o ../lib/abl/engine_$name.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 --hip-link -shared -fPIC -o ../lib/abl/lib_$name.so ../lib/abl/engine_$name.o ../lib/abl/on_engine.o ../lib/abl/ring.o -ldl &&
    rm ../lib/abl/engine_$name.o ) &
done
wait
rm -f ../lib/abl/on_engine.o ../lib/abl/ring.o
ls ../lib/abl
