#!/bin/bash
# Times the update pass under different (z,t)-region shapes of the workgroup order.
for r in 2,4 1,8 4,2 1,1 ; do
  echo "regions $r"; QCDGPU_AMD_REGIONS=$r timeout -k 10 200 python3 tools/ablate_time.py 2>&1 | grep -E "BASE"
done
