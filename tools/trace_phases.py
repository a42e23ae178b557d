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
TICK_US = 0.01   # s_memrealtime: 100 MHz, one counter for the whole device
t0 = t[:, 0].min()
dur = t[:, 4].max() - t0
print("us/launch (events, sweep avg) %.1f; stamp span %.1f us" % (1e3 * ms / n, dur * TICK_US))
names = ["staple", "rng", "heatbath", "store"]
for k in range(4):
    d = (t[:, k + 1] - t[:, k]) * TICK_US
    print("%-9s mean %6.2f us  p10 %6.2f  p50 %6.2f  p90 %6.2f" % (names[k], d.mean(), *np.percentile(d, [10, 50, 90])))
life = (t[:, 4] - t[:, 0]) * TICK_US
print("lifetime  mean %6.2f us  p10 %6.2f  p50 %6.2f  p90 %6.2f" % (life.mean(), *np.percentile(life, [10, 50, 90])))
start = (t[:, 0] - t0) * TICK_US
print("start-time histogram (10 us bins):", np.histogram(start, bins=np.arange(0, 260, 10))[0].tolist())
for a in np.arange(0, dur, 5.0 / TICK_US):
    c = [int(((t[:, k] <= a + t0) & (a + t0 < t[:, k + 1])).sum()) for k in range(4)]
    print("t=%6.1f us  staple %5d  rng %5d  heatbath %5d  store %4d   resident %5d" % (a * TICK_US, *c, sum(c)))
clk = (t[:, 5] - t[:, 7]) / ((t[:, 4] - t[:, 0]) * TICK_US)   # shader cycles per us over the wave's life
print("in-kernel shader clock: median %.0f MHz (p10 %.0f, p90 %.0f)" % (np.median(clk), *np.percentile(clk, [10, 90])))
np.save(os.path.join(ROOT, "gpurun_out", "trace.npy"), t)
