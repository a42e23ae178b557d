#!/usr/bin/env python3
"""Host issue time and device time of a slab pass issued by the C library (qcd_ring, copy transport,
two 16-slice parts of a 32^4 SU(3) lattice on one device) against the Python driver's."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import qcdgpu_amd as q
from qcdgpu_amd import biglattice as bl

L = (32, 32, 32, 32)
ctxs = []
for k in range(2):
    c = q.Context(group=3, L=L, beta=6.0, x0=16 * k, nx_local=16, halo=1)
    c.init_lattice()
    ctxs.append(c)
ring = q.Ring(ctxs, q.RING_COPY)
ring.sweep(3)
ring.synchronize()
n = 20
t0 = time.perf_counter()
ring.sweep(n)
t1 = time.perf_counter()
ring.synchronize()
t2 = time.perf_counter()
print("qcd_ring  : host issues a pass of one part in %5.1f us; device: %6.1f us per pass of the whole 32^4 lattice"
      % (1e6 * (t1 - t0) / (16 * n), 1e6 * (t2 - t0) / (8 * n)))
# the host's own cost: one sweep issued into empty queues (nothing pushes back), best of five
best = 1e9
for _ in range(5):
    ring.synchronize()
    t0 = time.perf_counter()
    ring.sweep(1)
    best = min(best, time.perf_counter() - t0)
ring.synchronize()
print("qcd_ring  : into empty queues the host issues a pass of one part in %5.1f us" % (1e6 * best / 16))
ring.close()
for c in ctxs:
    c.close()
big = bl.make_gpu_slabs(3, L, 6.0, [16, 16])
big.sweep(3)
torch.cuda.synchronize()
t0 = time.perf_counter()
big.sweep(n)
t1 = time.perf_counter()
torch.cuda.synchronize()
for e in big.engines:
    e.ctx.synchronize()
t2 = time.perf_counter()
print("Python    : host issues a pass of one part in %5.1f us; device: %6.1f us per pass of the whole 32^4 lattice"
      % (1e6 * (t1 - t0) / (16 * n), 1e6 * (t2 - t0) / (8 * n)))
best = 1e9
for _ in range(5):
    torch.cuda.synchronize()
    for e in big.engines:
        e.ctx.synchronize()
    t0 = time.perf_counter()
    big.sweep(1)
    best = min(best, time.perf_counter() - t0)
torch.cuda.synchronize()
print("Python    : into empty queues the host issues a pass of one part in %5.1f us" % (1e6 * best / 16))
