#!/usr/bin/env python3
"""us per overrelaxation pass and per re-unitarisation at 32^4 SU(3) (wall clock around synchronised blocks)."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qcdgpu_amd as q

c = q.Context(group=3, L=(32, 32, 32, 32), beta=6.0)
c.init_lattice()
c.sweep(2)
for name, fn, per in (("overrelaxation pass", lambda: c.overrelax_sweep(5), 40), ("re-unitarisation", lambda: [c.reunitarize() for _ in range(10)], 10)):
    fn(); c.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); fn(); c.synchronize(); best = min(best, time.perf_counter() - t0)
    print("%-22s %7.1f us" % (name, 1e6 * best / per))
