#!/usr/bin/env python3
"""Polyakov loop sums of the kernel that cuts a line into four segments against one thread per line, and the device time of a call."""
import os
import subprocess
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    import numpy as np
    import qcdgpu_amd as q
    group = int(sys.argv[2])
    L = tuple(int(v) for v in sys.argv[3:7])
    c = q.Context(group=group, L=L, beta=6.0 if group == 3 else 2.4)
    c.init_lattice()
    c.sweep(3)
    out = np.concatenate([np.asarray(c.polyakov_sums())] + [np.asarray(c.polyakov_profile_sums(a)).ravel() for a in range(3)])
    np.save("/tmp/pol_%s.npy" % sys.argv[1], out)
    c.set_profiling(True)
    ts = []
    for _ in range(7):
        t0 = c.timing()
        for _ in range(20):
            c.polyakov()
        t1 = c.timing()
        ts.append(1e3 * (t1.measure_ms - t0.measure_ms) / (t1.measure_launches - t0.measure_launches))
    ts.sort()
    print("split=%s SU(%d) %s call median %.1f us (min %.1f)" % (sys.argv[1], group, L, ts[3], ts[0]), flush=True)
else:
    import numpy as np
    for g, L in ((3, "32 32 32 32"), (2, "32 32 32 32"), (3, "16 16 16 16"), (3, "8 12 6 10"), (3, "6 4 4 9"), (2, "4 6 4 13")):
        for m in ("0", "1"):
            env = dict(os.environ, QCDGPU_AMD_POLYAKOV_SPLIT=m)
            subprocess.run([sys.executable, __file__, m, str(g)] + L.split(), env=env, check=True)
        a, b = np.load("/tmp/pol_0.npy"), np.load("/tmp/pol_1.npy")
        print("   max |difference| %.3e of max |sum| %.3e" % (np.abs(a - b).max(), np.abs(a).max()), flush=True)
