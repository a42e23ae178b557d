#!/usr/bin/env python3
"""Wall time of a plaquette measurement at 32^4 SU(3) for every build under qcdgpu_amd/lib/abl."""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, time; sys.path.insert(0, %r)
import qcdgpu_amd as q
c = q.Context(group=3, L=(32,32,32,32), beta=6.0); c.init_lattice(); c.sweep(2)
best = 1e9
for _ in range(5):
    c.synchronize(); t0 = time.perf_counter()
    for _ in range(20): c.plaquette()
    best = min(best, (time.perf_counter() - t0) / 20)
print("%%8.1f us per plaquette call" %% (1e6 * best))
''' % ROOT
for lib in sorted(glob.glob(os.path.join(ROOT, "qcdgpu_amd", "lib", "abl", "*.so"))):
    env = dict(os.environ, QCDGPU_AMD_LIB=lib)
    r = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True, timeout=300)
    print("%-28s %s" % (os.path.basename(lib), (r.stdout.strip() or r.stderr.strip()[-300:])), flush=True)
