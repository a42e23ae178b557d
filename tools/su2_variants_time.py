#!/usr/bin/env python3
"""SU(2) 32^4 heat-bath and overrelaxation pass times for every diagnostic build under qcdgpu_amd/lib/abl (one subprocess each)."""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys; sys.path.insert(0, %r)
import qcdgpu_amd as q
c = q.Context(group=2, L=(32,32,32,32), beta=2.5); c.init_lattice(); c.sweep(20)
c.set_profiling(True)
out = []
for fn in (c.sweep, c.overrelax_sweep):
    ts = []
    for _ in range(7):
        t0 = c.timing(); fn(10); t1 = c.timing()
        ts.append(1e3 * (t1.update_ms - t0.update_ms) / (t1.update_launches - t0.update_launches))
    ts.sort(); out.append(ts[3])
print("heat-bath %%6.1f  overrelaxation %%6.1f us per pass   plaquette %%.12f" %% (out[0], out[1], c.plaquette()[2]))
''' % ROOT
for lib in sorted(glob.glob(os.path.join(ROOT, "qcdgpu_amd", "lib", "abl", "*.so"))):
    env = dict(os.environ, QCDGPU_AMD_LIB=lib)
    r = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True, timeout=300)
    print("%-28s %s" % (os.path.basename(lib), (r.stdout.strip() or r.stderr.strip()[-300:])), flush=True)
