#!/bin/bash
for r in 2,4 1,8 1,1 1,2 ; do
  echo "regions $r xfast"; QCDGPU_AMD_XFAST=1 QCDGPU_AMD_REGIONS=$r timeout -k 10 200 python3 tools/ablate_time.py 2>&1 | grep -E "BASE|RNG_HB.so"
done
