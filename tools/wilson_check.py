#!/usr/bin/env python3
"""Wilson loops of the two-row kernel against the kernel with full 3 x 3 products on the same configuration, and the device time of a call."""
import os
import subprocess
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LOOPS = ((1, 1), (2, 2), (3, 4), (4, 2), (6, 6))
if len(sys.argv) > 1:
    import numpy as np
    import qcdgpu_amd as q
    L = tuple(int(v) for v in sys.argv[2:6])
    c = q.Context(group=3, L=L, beta=6.0)
    c.init_lattice()
    c.sweep(3)
    np.save("/tmp/wilson_%s.npy" % sys.argv[1], np.array([c.wilson_loop(R, T) for R, T in LOOPS]))
    c.set_profiling(True)
    for R, T in ((2, 2), (4, 4)):
        ts = []
        for _ in range(7):
            t0 = c.timing()
            for _ in range(10):
                c.wilson_loop(R, T)
            t1 = c.timing()
            ts.append(1e3 * (t1.measure_ms - t0.measure_ms) / (t1.measure_launches - t0.measure_launches))
        ts.sort()
        print("rows=%s %s W(%d,%d) call median %.1f us (min %.1f)" % (sys.argv[1], L, R, T, ts[3], ts[0]), flush=True)
else:
    import numpy as np
    for L in ("32 32 32 32", "16 16 16 16", "8 12 6 10"):
        for m in ("0", "1"):
            env = dict(os.environ, QCDGPU_AMD_WILSON_ROWS=m)
            subprocess.run([sys.executable, __file__, m] + L.split(), env=env, check=True)
        a, b = np.load("/tmp/wilson_0.npy"), np.load("/tmp/wilson_1.npy")
        print("   ", a, "max |difference| %.3e" % np.abs(a - b).max(), flush=True)
