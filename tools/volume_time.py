#!/usr/bin/env python3
"""us per update launch against the number of blocks per launch (SU(3), beta 6): what a launch costs beyond its blocks."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qcdgpu_amd as q
for prec in (2, 1):
    for nx in (16, 32, 64, 128):
        c = q.Context(group=3, L=(nx, 32, 32, 32), beta=6.0, precision=prec)
        c.init_lattice()
        c.sweep(12)
        ts = []
        for _ in range(5):
            ms, n = c.timed_sweeps(10)
            ts.append(1e3 * ms / n)
        ts.sort()
        print("%s  %3d x 32^3: %7.1f us/launch  = %6.2f us per x-slice" % ("f64" if prec == 2 else "f32", nx, ts[2], ts[2] / nx), flush=True)
        c.close()
