#!/usr/bin/env python3
"""Device time of an overrelaxation pass beside a heat-bath pass (SU(3) and SU(2), thermalised): median of 7 brackets of 10 sweeps."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qcdgpu_amd as q
for group, beta in ((3, 6.0), (2, 2.4)):
    for L in ((16, 16, 16, 16), (32, 32, 32, 32)):
        c = q.Context(group=group, L=L, beta=beta)
        c.init_lattice()
        c.sweep(10)
        c.set_profiling(True)
        for name, fn in (("heat-bath", c.sweep), ("overrelaxation", c.overrelax_sweep)):
            ts = []
            for _ in range(7):
                t0 = c.timing()
                fn(10)
                t1 = c.timing()
                ts.append(1e3 * (t1.update_ms - t0.update_ms) / (t1.update_launches - t0.update_launches))
            ts.sort()
            print("SU(%d) %-18s %-15s %7.1f us per pass" % (group, L, name, ts[3]), flush=True)
        c.close()
