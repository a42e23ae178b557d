#!/bin/bash
# Times the update pass for different numbers of workgroups per launch (0 = one per block of sites).
for n in 0 1024 2048; do
  echo "grid $n"; QCDGPU_AMD_GRID=$n timeout -k 10 100 python3 tools/ablate_time.py 2>&1 | grep -E "lib_" || exit 1
done
