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
for e in big.engines:
    e.ctx.close()

# one 32^4 slab whose neighbours on both sides are itself: bare pass, copy transport, RCCL transport
# (what a rank of an N-GPU run does per pass, minus the link between two devices)
def slab_pass_us(transport):
    c = q.Context(group=3, L=L, beta=6.0, x0=0, nx_local=32, halo=1) if transport is not None else q.Context(group=3, L=L, beta=6.0)
    c.init_lattice()
    r = q.Ring([c], transport) if transport is not None else None
    run = r.sweep if r is not None else c.sweep
    run(3)
    c.synchronize() if r is None else r.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        run(n)
        c.synchronize() if r is None else r.synchronize()
        best = min(best, time.perf_counter() - t0)
    if r is not None:
        print("   (%s: %d of %d passes as one launch with the edge signal)" % ("RCCL" if transport == q.RING_RCCL else "copy", r.one_launch_passes(), 8 * (3 + 3 * n)))
        r.close()
    c.close()
    return 1e6 * best / (8 * n)
print("32^4 slab, halo of one slice, per pass: bare update %6.1f us, ring with copy transport %6.1f us, ring over RCCL (sends to itself) %6.1f us"
      % (slab_pass_us(None), slab_pass_us(q.RING_COPY), slab_pass_us(q.RING_RCCL)), flush=True)
