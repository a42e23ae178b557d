#!/bin/bash
# tools/plaq_prof.sh for every diagnostic build under qcdgpu_amd/lib/abl
for lib in ${GRAFT_REPO_ROOT:-$(pwd)}/qcdgpu_amd/lib/abl/*.so; do echo "== $(basename $lib)"; bash tools/plaq_prof.sh "$lib" | grep -i "plaquette" || exit 1; done
