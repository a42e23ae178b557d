#!/usr/bin/env python3
"""Mean plaquette of the engine's chain (per-thread streams) against the reference program's chain fixtures:
long GPU runs, errors from bins of several lengths."""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import qcdgpu_amd as q

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def binned(x, size):
    x = np.asarray(x)
    nb = len(x) // size
    b = x[:nb * size].reshape(nb, size).mean(axis=1)
    return b.std(ddof=1) / np.sqrt(nb)


for n, therm, samples in ((8, 200, 20000), (16, 200, 10000), (32, 200, 4000)):
    if not os.path.exists(os.path.join(ROOT, "tests", "golden", "stat_su3_%dx%dx%dx%d_b6.json" % (n, n, n, n))):
        continue
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "stat_su3_%dx%dx%dx%d_b6.json" % (n, n, n, n))))
    ref = np.asarray(fx["plq_total"])
    print("reference %d^4: %d samples, mean %.6f, error by bins of 10/25/50: %.2e %.2e %.2e; first/second half %.6f %.6f"
          % (n, len(ref), ref.mean(), binned(ref, 10), binned(ref, 25), binned(ref, 50), ref[:len(ref) // 2].mean(), ref[len(ref) // 2:].mean()), flush=True)
    for seed in (54, 77):
        c = q.Context(group=3, L=(n, n, n, n), beta=6.0, nhit=10, rand_series=seed)
        c.init_lattice()
        c.sweep(therm)
        mine = []
        for _ in range(samples):
            c.sweep(1)
            mine.append(c.plaquette()[2])
        mine = np.asarray(mine)
        print("engine %d^4 seed %d: %d samples, mean %.6f, error by bins of 20/100/500: %.2e %.2e %.2e; quarters %s"
              % (n, seed, samples, mine.mean(), binned(mine, 20), binned(mine, 100), binned(mine, 500),
                 " ".join("%.6f" % v.mean() for v in np.array_split(mine, 4))), flush=True)
        c.close()
