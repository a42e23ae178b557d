#!/usr/bin/env python3
"""us per measurement call of the O(N) engine at 64^4 (wall clock around synchronised calls)."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qcdgpu_amd as q

for prec in (0, 1):
    c = q.OnContext(L=(64, 64, 64, 64), precision=prec)
    c.init_lattice()
    c.sweep(1)
    c.measure((1, 0, 0, 0))
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(10):
            c.measure((1, 0, 0, 0))
        best = min(best, (time.perf_counter() - t0) / 10)
    print("precision %d: %.1f us per measure call" % (prec, 1e6 * best))
