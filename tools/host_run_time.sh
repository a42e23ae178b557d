#!/bin/bash
# End-to-end wall time of the host program on a 32^4 SU(3) run with the default measurement set:
# tools/host_run_time.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
D=$(mktemp -d)
for niter in 1 10; do
cat > $D/init.dat <<INI
	GROUP = 3
	ND = 4
	L1 = 32
	L2 = 32
	L3 = 32
	L4 = 32
	BETA = 6.0
	NHIT = 10
	INTS = 1
	PRNG = RANLUX3
	RANDSERIES = 54
	NAV = 10
	ITER = 201
	NITER = $niter
	OUTPUTPATH = $D/
INI
s=$(date +%s.%N)
$R/qcdgpu_amd/bin/qcdgpu $D/init.dat > $D/out.txt 2>&1 || { tail -5 $D/out.txt; exit 1; }
e=$(date +%s.%N)
sweeps=$((10 + 200 * niter))
python3 -c "print('NITER=%d: %d sweeps + 201 samples in %.3f s wall = %.3f ms per sweep incl. everything' % ($niter, $sweeps, $e - $s, 1e3 * ($e - $s) / $sweeps))"
grep -i "time\|sweep" $D/out.txt | head -8
done
rm -rf $D
