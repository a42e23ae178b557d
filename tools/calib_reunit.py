#!/usr/bin/env python3
"""Known-byte calibration of FETCH_SIZE/WRITE_SIZE for this engine's access pattern: k_reunit
reads every stored link once and writes it once (2 x 403 MB at 32^4 SU(3)), 16 B per lane."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qcdgpu_amd as q
for prec in (2, 1):      # the double and the single precision library (16 B and 8 B per lane)
    c = q.Context(group=3, L=(32, 32, 32, 32), beta=6.0, precision=prec)
    c.init_lattice()
    for _ in range(5):
        c.reunitarize()
    print(c.plaquette())
    c.close()
