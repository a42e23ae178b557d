#!/bin/bash
# Times the update pass of the in-tree library under the diagnostic environment overrides:
# region shapes, workgroups per launch, block shapes.  tools/knobs_time.sh
t() { echo -n "$1: "; env $1 timeout -k 10 100 python3 tools/time_lib.py 2>&1 | grep "us/launch" || exit 1; }
t "X=0"
for r in 1,8 4,2 8,1 1,1; do t "QCDGPU_AMD_REGIONS=$r"; done
for n in 1536 1792 2304 2560 4096; do t "QCDGPU_AMD_GRID=$n"; done
for s in 4,1,1 2,2,1 2,1,2 1,2,2 1,1,4; do t "QCDGPU_AMD_TILE=$s"; done
