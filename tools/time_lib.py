#!/usr/bin/env python3
"""us per update launch of the in-tree library (32^4 SU(3), best of three timed blocks)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qcdgpu_amd as q
c = q.Context(group=3, L=(32, 32, 32, 32), beta=6.0)
c.init_lattice()
c.sweep(3)
best = 1e9
for _ in range(3):
    ms, n = c.timed_sweeps(5)
    best = min(best, ms / n)
print("%8.1f us/launch  %6.2f Gupd/s  plaquette %.12f" % (1e3 * best, 524288 / best / 1e6, c.plaquette()[2]))
c.set_profiling(True)
for name, fn in (("plaquette", c.plaquette), ("polyakov", c.polyakov)):
    for _ in range(3):
        fn()
    t0 = c.timing()
    for _ in range(20):
        fn()
    t1 = c.timing()
    print("%-10s %.1f us per call (kernel + reduce + result copy, device time)" % (
        name, 1e3 * (t1.measure_ms - t0.measure_ms) / max(1, t1.measure_launches - t0.measure_launches)))
