#!/usr/bin/env python3
"""Times the sweep for every diagnostic build under qcdgpu_amd/lib/abl (one subprocess each)."""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys; sys.path.insert(0, %r)
import qcdgpu_amd as q
c = q.Context(group=3, L=(32,32,32,32), beta=6.0); c.init_lattice(); c.sweep(20)
ts = []
for _ in range(7):
    ms, n = c.timed_sweeps(20); ts.append(1e3 * ms / n)
ts.sort()
print("min %%6.1f  median %%6.1f  max %%6.1f us/launch  %%6.2f Gupd/s" %% (ts[0], ts[3], ts[-1], 524288 / ts[3] / 1e3))
''' % ROOT
for lib in sorted(glob.glob(os.path.join(ROOT, "qcdgpu_amd", "lib", "abl", "*.so"))):
    env = dict(os.environ, QCDGPU_AMD_LIB=lib)
    r = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True, timeout=300)
    print("%-28s %s" % (os.path.basename(lib), (r.stdout.strip() or r.stderr.strip()[-300:])), flush=True)
