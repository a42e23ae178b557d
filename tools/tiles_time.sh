#!/bin/bash
# Times the update pass for different workgroup block shapes (tz,tt,tx) and region cuts.
for t in 0,0,0 8,1,1 4,2,1 4,1,2 2,2,2 2,4,1 2,1,4 1,2,4 1,4,2 1,8,1 1,1,8; do
  echo "tile $t"; QCDGPU_AMD_TILE=$t timeout -k 10 100 python3 tools/ablate_time.py 2>&1 | grep -E "BASE" || exit 1
done
