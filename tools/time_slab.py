#!/usr/bin/env python3
"""ms per sweep of one 32^4 slab with halos on one GPU (the per-rank work of the multi-GPU run,
messages copied device to device), with the pass split boundary-first and unsplit."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from qcdgpu_amd import biglattice as bl
for depth, concurrent in ((0, False), (8, False), (1, True), (2, True)):
    bl.BigLattice.BOUNDARY_DEPTH = depth
    bl.BigLattice.BOUNDARY_DEPTH_CONCURRENT = depth
    bl.BigLattice._concurrent = (lambda self: True) if concurrent else (lambda self: False)
    big = bl.make_gpu_slabs(3, (32, 32, 32, 32), 6.0, [32])
    big.sweep(3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    big.sweep(10)
    t_host = time.perf_counter() - t0      # host time to issue the work (everything is asynchronous)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(("concurrent " if concurrent else "consecutive ") + "boundary depth %d: %.3f ms per sweep (%.1f us per pass; host issues a pass in %.1f us), plaquette %.10f" % (
        big.boundary_depth(), 1e2 * dt, 1e5 * dt / 8, 1e5 * t_host / 8, big.plaquette()[2]), flush=True)
    del big
