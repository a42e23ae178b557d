#!/usr/bin/env python3
"""Host issue time against device time of a ring pass over several parts driven by ONE process (qcd_ring, copy
transport, all parts on the box's one device): python tools/time_ring_parts.py [parts] [slices per part] [threads 0|1]"""
import os
import sys
import time
nparts = int(sys.argv[1]) if len(sys.argv) > 1 else 8
nx = int(sys.argv[2]) if len(sys.argv) > 2 else 4
if len(sys.argv) > 3:
    os.environ["QCDGPU_AMD_RING_THREADS"] = sys.argv[3]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qcdgpu_amd as q

L = (nparts * nx, 32, 32, 32)
share = int(os.environ.get("RING_SHARE", "0"))
ctxs = []
for k in range(nparts):
    c = q.Context(group=3, L=L, beta=6.0, x0=nx * k, nx_local=nx, halo=1, rng_share=share)
    c.init_lattice()
    ctxs.append(c)
ring = q.Ring(ctxs, q.RING_COPY)
if os.environ.get("RING_CHB2"):       # one exchange per sweep (the reference's CHB2 build) instead of one per pass
    ring.sweep = ring.sweep_chb2
ring.sweep(3)
ring.synchronize()
n = 20
t0 = time.perf_counter()
ring.sweep(n)
t1 = time.perf_counter()
ring.synchronize()
t2 = time.perf_counter()
# the host's own cost: one sweep issued into empty queues (nothing pushes back), best of five
best = 1e9
for _ in range(5):
    ring.synchronize()
    t0b = time.perf_counter()
    ring.sweep(1)
    best = min(best, time.perf_counter() - t0b)
ring.synchronize()
print("%d parts x %d slices (share %d), threads %s: host issues a pass of ALL parts in %6.1f us (%5.1f us into empty queues); "
      "device: %6.1f us per pass of all parts (all on one device)"
      % (nparts, nx, ctxs[0].rng_share, os.environ.get("QCDGPU_AMD_RING_THREADS", "1"), 1e6 * (t1 - t0) / (8 * n), 1e6 * best / 8,
         1e6 * (t2 - t0) / (8 * n)))
ring.close()
for c in ctxs:
    c.close()
