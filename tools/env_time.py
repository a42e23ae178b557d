#!/usr/bin/env python3
"""Times the 32^4 SU(3) sweep under sets of environment variables (the library's diagnostic overrides), one
subprocess each: python tools/env_time.py "A=1 B=2" "A=3" ...  ("" = no override).  QCDGPU_AMD_LIB picks the library."""
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
print("min %%6.1f  median %%6.1f  max %%6.1f us/launch  %%6.2f Gupd/s  share %%d" %% (ts[0], ts[3], ts[-1], 524288 / ts[3] / 1e3, c.rng_share))
''' % ROOT
for spec in sys.argv[1:] or [""]:
    env = dict(os.environ)
    for kv in spec.split():
        k, v = kv.split("=", 1)
        env[k] = v
    r = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True, timeout=300)
    print("%-44s %s" % (spec or "(defaults)", (r.stdout.strip() or r.stderr.strip()[-300:])), flush=True)
