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
