#!/usr/bin/env python3
"""F_mu_nu sums of the marching kernel against the one-site-per-thread kernel on the same configuration, and the device time of a call."""
import os
import subprocess
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    import numpy as np
    import qcdgpu_amd as q
    L = tuple(int(v) for v in sys.argv[2:6])
    c = q.Context(group=3, L=L, beta=6.0)
    c.init_lattice()
    c.sweep(3)
    out = []
    for field, i1, i2 in ((0, 3, 8), (1, 1, 5), (0, 2, 7), (1, 4, 6)):
        out.append(np.asarray(c.fmunu_sums(field, i1, i2) if hasattr(c, "fmunu_sums") else c.fmunu(field, i1, i2)))
    np.save("/tmp/fmunu_%s.npy" % sys.argv[1], np.concatenate(out))
    c.set_profiling(True)
    for field in (0, 1):
        ts = []
        for _ in range(7):
            t0 = c.timing()
            for _ in range(20):
                c.fmunu(field, 3, 8)
            t1 = c.timing()
            ts.append(1e3 * (t1.measure_ms - t0.measure_ms) / (t1.measure_launches - t0.measure_launches))
        ts.sort()
        print("march=%s %s field %d call median %.1f us (min %.1f)" % (sys.argv[1], L, field, ts[3], ts[0]), flush=True)
else:
    import numpy as np
    for L in ("32 32 32 32", "16 16 16 16", "24 16 8 12", "6 8 8 4"):
        for m in ("0", "1"):
            env = dict(os.environ, QCDGPU_AMD_FMUNU_MARCH=m)
            subprocess.run([sys.executable, __file__, m] + L.split(), env=env, check=True)
        a, b = np.load("/tmp/fmunu_0.npy"), np.load("/tmp/fmunu_1.npy")
        print("   max |difference| %.3e of max |sum| %.3e" % (np.abs(a - b).max(), np.abs(a).max()), flush=True)
