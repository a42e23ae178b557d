"""Wall time per call of every measurement entry point on a 32^4 SU(3) configuration."""
import os
import sys
import time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import qcdgpu_amd as q  # noqa: E402

with q.Context(group=3, L=(32, 32, 32, 32), beta=6.0) as c:
    c.init_lattice()
    c.sweep(3)
    calls = [("plaquette", lambda: c.plaquette()), ("polyakov", lambda: c.polyakov()),
             ("wilson 2x2", lambda: c.wilson_loop(2, 2)), ("fmunu", lambda: c.fmunu(0, 3, 8))]
    for a in range(3):
        calls.append(("plaquette profile %s" % "xyz"[a], lambda a=a: c.plaquette_profile_sums(a)))
        calls.append(("polyakov profile %s" % "xyz"[a], lambda a=a: c.polyakov_profile_sums(a)))
    for name, fn in calls:
        fn()
        c.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            fn()
        print("%-22s %9.1f us per call" % (name, 1e6 * (time.perf_counter() - t0) / 5), flush=True)
