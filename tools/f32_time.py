#!/usr/bin/env python3
"""us per update launch of the single-precision library beside the double one (32^4, SU(3) and SU(2))."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qcdgpu_amd as q
for group, beta in ((3, 6.0), (2, 2.5)):
    for prec in (2, 1):
        c = q.Context(group=group, L=(32, 32, 32, 32), beta=beta, precision=prec)
        c.init_lattice()
        c.sweep(20)
        ts = []
        for _ in range(5):
            ms, n = c.timed_sweeps(20)
            ts.append(1e3 * ms / n)
        ts.sort()
        per = (864 if group == 3 else 288) // (1 if prec == 2 else 2)
        print("SU(%d) %s: %6.1f us/launch  %5.2f Gupd/s  %4.0f GB/s algorithmic (%d B per update)  plaquette %.6f" % (
            group, "f64" if prec == 2 else "f32", ts[2], 524288 / ts[2] / 1e3, 524288 * per / ts[2] / 1e3, per, c.plaquette()[2]), flush=True)
        c.set_profiling(True)
        for name, fn in (("plaquette", c.plaquette), ("polyakov", c.polyakov)):
            for _ in range(3):
                fn()
            t0 = c.timing()
            for _ in range(20):
                fn()
            t1 = c.timing()
            print("      %-10s %.1f us per call" % (name, 1e3 * (t1.measure_ms - t0.measure_ms) / max(1, t1.measure_launches - t0.measure_launches)))
        c.close()
