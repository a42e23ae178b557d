#!/usr/bin/env python3
"""Heat-bath sweep rate on lattices below the benchmark's 32^4 (SU(3), beta = 6, NHIT = 10): median of 7 brackets of 20 sweeps."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qcdgpu_amd as q
for L in ((8, 8, 8, 8), (12, 12, 12, 12), (16, 16, 16, 16), (20, 20, 20, 20), (24, 24, 24, 24), (32, 16, 16, 16), (16, 32, 32, 16), (32, 32, 32, 32)):
    c = q.Context(group=3, L=L, beta=6.0)
    c.init_lattice()
    c.sweep(10)
    c.set_profiling(True)
    ts = []
    for _ in range(7):
        t0 = c.timing()
        c.sweep(20)
        t1 = c.timing()
        ts.append(1e3 * (t1.update_ms - t0.update_ms) / (t1.update_launches - t0.update_launches))
    ts.sort()
    n = L[0] * L[1] * L[2] * L[3]
    print("%-18s %7.1f us per pass  %.2fe9 link-updates/s" % (L, ts[3], n / 2 / ts[3] / 1e3), flush=True)
    c.close()
