#!/usr/bin/env python3
"""Phase timeline of k_update from a -DQCD_TRACE build (tools/variants.sh TRACE:-DQCD_TRACE):
per-wave cycle-counter stamps at start / after staple / after RNG open / after heat-bath / end."""
import ctypes
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("QCDGPU_AMD_LIB", os.path.join(ROOT, "qcdgpu_amd", "lib", "abl", "lib_TRACE.so"))
import numpy as np
import qcdgpu_amd as q
from qcdgpu_amd import capi

c = q.Context(group=3, L=(32, 32, 32, 32), beta=6.0)
c.init_lattice()
c.sweep(3)
ms, n = c.timed_sweeps(1)
c.update(1, 0)
c.synchronize()
lib = capi._load()
buf = np.zeros(8192 * 8, dtype=np.uint64)
rc = lib.qcd_debug_read_trace(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes))
assert rc == 0, rc
t = buf.reshape(8192, 8).astype(np.int64)
t0 = t[:, 0].min()
dur = t[:, 4].max() - t0
print("us/launch (events, sweep avg) %.1f; stamp span %d ticks" % (1e3 * ms / n, dur))
tick_us = (1e3 * ms / n) / dur   # rough: one launch = stamp span
print("tick = %.4f us (if span = launch)" % tick_us)
names = ["staple", "rng", "heatbath", "store"]
for k in range(4):
    d = (t[:, k + 1] - t[:, k]) * tick_us
    print("%-9s mean %6.2f us  p10 %6.2f  p50 %6.2f  p90 %6.2f" % (names[k], d.mean(), *np.percentile(d, [10, 50, 90])))
life = (t[:, 4] - t[:, 0]) * tick_us
print("lifetime  mean %6.2f us  p10 %6.2f  p50 %6.2f  p90 %6.2f" % (life.mean(), *np.percentile(life, [10, 50, 90])))
start = (t[:, 0] - t0) * tick_us
print("start-time histogram (us):", np.histogram(start, bins=16)[0].tolist())
print("start percentiles:", np.percentile(start, [0, 25, 50, 75, 100]).round(1).tolist())
# concurrency over time: waves in the staple phase vs in ALU phases
grid = np.linspace(0, dur, 41)[:-1]
for g in grid[::2]:
    a = g + t0
    in_st = ((t[:, 0] <= a) & (a < t[:, 1])).sum()
    in_alu = ((t[:, 1] <= a) & (a < t[:, 3])).sum()
    print("t=%6.1f us  staple-phase waves %5d   alu-phase waves %5d" % (g * tick_us, in_st, in_alu))
hw = t[:, 6]
print("distinct (cu,simd,se) ids:", len(np.unique(hw & 0xFFFF0)), " xcc ids:", np.unique(t[:, 7]).tolist())
np.save(os.path.join(ROOT, "gpurun_out", "trace.npy"), t)
