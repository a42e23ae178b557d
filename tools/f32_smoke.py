import sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qcdgpu_amd as q
from tests import oracle_lib as ol
L = (8, 8, 8, 8)
for group, beta in ((3, 6.0), (2, 2.5)):
    # gid mode: deterministic, first trial always accepted -> smooth in the arithmetic
    lat = ol.Lattice(group, L, beta, nhit=10, ints=2, rng_mode=ol.RNG_GID)
    with q.Context(group=group, L=L, beta=beta, nhit=10, start_mode=q.START_GID, precision=1) as c:
        c.init_lattice()
        print("SU(%d) gid start  f32" % group, c.plaquette(), " f64 checker", lat.plaquette())
        for s in range(2):
            c.sweep(1); lat.sweep(1)
            print("   sweep", s + 1, c.plaquette(), lat.plaquette(), np.max(np.abs(c.plaquette() - lat.plaquette()) / lat.plaquette()))
        print("   polyakov", c.polyakov(), lat.polyakov())
    lat.close()
    with q.Context(group=group, L=L, beta=beta, nhit=10, precision=1) as c, q.Context(group=group, L=L, beta=beta, nhit=10) as d:
        c.init_lattice(); d.init_lattice()
        print("SU(%d) cold f32" % group, c.plaquette())
        c.sweep(5); d.sweep(5)
        print("   5 sweeps f32", c.plaquette(), "f64", d.plaquette(), "draws", c.rng_draws(), d.rng_draws())
        t = c.download()
        print("   table dtype", t.dtype, t.shape, "unitarity row norm", np.abs((t[0, :, :3] ** 2 + t[4, :, :3] ** 2).sum(axis=1) - 1).max() if group == 3 else "")
