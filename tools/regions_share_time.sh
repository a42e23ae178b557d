for r in 2,4 4,2 8,1 1,8; do echo "regions $r"; QCDGPU_AMD_REGIONS=$r timeout -k 10 300 python tools/ablate_time.py || exit 1; done
