#!/usr/bin/env python3
"""Large-offset check: a 64^4 SU(3) lattice in one context against the same lattice cut into two
32-slice slabs on the same GPU (streams are keyed by global site, so the digits must agree), and
the 64^4 plaquette/Polyakov kernels against the per-plane sums."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import qcdgpu_amd as q
from qcdgpu_amd import biglattice as bl

L = (64, 64, 64, 64)
t0 = time.time()
with q.Context(group=3, L=L, beta=6.0, nhit=10, rand_series=54) as c:
    c.init_lattice()
    assert c.plaquette()[2] == 3.0
    c.sweep(2)
    one = c.plaquette_sums(), c.polyakov_sums()
    prof = c.plaquette_profile_sums(1)
    assert np.allclose(prof.sum(axis=0), one[0], rtol=1e-12)
    ms, n = c.timed_sweeps(3)
    print("one context: plaquette sums", one[0], " %.2f ms per sweep" % (ms / 3), flush=True)
big = bl.make_gpu_slabs(3, L, 6.0, [32, 32], nhit=10, rand_series=54)
big.sweep(2)
two = (np.sum([e.plaquette_sums() for e in big.engines], axis=0), np.sum([e.polyakov_sums() for e in big.engines], axis=0))
for e in big.engines:
    e.ctx.close()
print("two slabs  : plaquette sums", two[0], flush=True)
assert np.allclose(one[0], two[0], rtol=1e-13), (one[0], two[0])
assert np.allclose(one[1], two[1], rtol=1e-10, atol=1e-9), (one[1], two[1])
print("ok, %.0f s" % (time.time() - t0))
