#!/usr/bin/env python3
"""Which side of the gather / heat-bath pairs waits for the other (diagnostic build -DQCD_GH_STATS):
sleeps per block of 64 sites spent by the gather wavefronts waiting for a free slot and by the
heat-bath wavefronts waiting for a staple sum."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("QCDGPU_AMD_GH", "2")
import qcdgpu_amd as q
from qcdgpu_amd import capi
c = q.Context(group=3, L=(32, 32, 32, 32), beta=6.0)
c.init_lattice()
c.sweep(2)
c.synchronize()
out = (C.c_ulonglong * 2)()
capi.lib.qcd_debug_gh_spins(out, 1)
ms, n = c.timed_sweeps(5)
c.synchronize()
capi.lib.qcd_debug_gh_spins(out, 1)
blocks = n * 8192
print("%.1f us/launch; sleeps per block: gather side %.1f, heat-bath side %.1f" % (1e3 * ms / n, out[0] / blocks, out[1] / blocks))
