#!/usr/bin/env python3
"""us per update launch with and without the alternating walking direction (32^4 SU(3); one process per setting)."""
import os
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys; sys.path.insert(0, %r)
import qcdgpu_amd as q
for prec in (2, 1):
    c = q.Context(group=3, L=(32,32,32,32), beta=6.0, precision=prec); c.init_lattice(); c.sweep(20)
    ts = []
    for _ in range(7):
        ms, n = c.timed_sweeps(20); ts.append(1e3 * ms / n)
    ts.sort()
    print("  %%s  min %%6.1f  median %%6.1f  max %%6.1f us/launch" %% ("f64" if prec == 2 else "f32", ts[0], ts[3], ts[-1]))
    c.close()
''' % ROOT
for rev in ("1", "0", "1", "0"):
    r = subprocess.run([sys.executable, "-c", CODE], env=dict(os.environ, QCDGPU_AMD_REVERSE=rev), capture_output=True, text=True, timeout=300)
    print("QCDGPU_AMD_REVERSE=%s\n%s" % (rev, r.stdout.rstrip() or r.stderr[-300:]), flush=True)
