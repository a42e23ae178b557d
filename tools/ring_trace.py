#!/usr/bin/env python3
"""A few sweeps of one 32^4 slab through qcd_ring (copy transport, its own neighbour), to be run under
rocprofv3 --kernel-trace: tools/ring_timeline.py prints the kernels of one pass with start and end times."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qcdgpu_amd as q

L = (32, 32, 32, 32)
c = q.Context(group=3, L=L, beta=6.0, x0=0, nx_local=32, halo=1)
c.init_lattice()
r = q.Ring([c], q.RING_RCCL if len(sys.argv) > 1 and sys.argv[1] == "rccl" else q.RING_COPY)
r.sweep(4)
r.synchronize()
r.close()
c.close()
