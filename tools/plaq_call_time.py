#!/usr/bin/env python3
"""Device time of a plaquette call (kernel + reduce + result copy) at 32^4 SU(3), median of 9 blocks of 20 calls."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qcdgpu_amd as q
c = q.Context(group=3, L=(32, 32, 32, 32), beta=6.0)
c.init_lattice()
c.sweep(5)
c.set_profiling(True)
ts = []
for _ in range(9):
    t0 = c.timing()
    for _ in range(20):
        c.plaquette()
    t1 = c.timing()
    ts.append(1e3 * (t1.measure_ms - t0.measure_ms) / (t1.measure_launches - t0.measure_launches))
ts.sort()
print("%-14s plaquette call: min %.1f  median %.1f  max %.1f us" % (os.path.basename(os.environ.get("QCDGPU_AMD_LIB", "in-tree")), ts[0], ts[4], ts[-1]))
