#!/usr/bin/env python3
"""ms per sweep of one 32^4 slab with halos on one GPU (the per-rank work of the multi-GPU run,
messages copied device to device), with the pass split boundary-first and unsplit."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from qcdgpu_amd import biglattice as bl
for depth in (0, 2, 4, 8):
    bl.BigLattice.BOUNDARY_DEPTH = depth
    big = bl.make_gpu_slabs(3, (32, 32, 32, 32), 6.0, [32])
    big.sweep(3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    big.sweep(10)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("boundary depth %d: %.3f ms per sweep (%.1f us per pass), plaquette %.10f" % (
        big.boundary_depth(), 1e2 * dt, 1e5 * dt / 8, big.plaquette()[2]), flush=True)
    del big
