"""Times the O(N) Metropolis sweep at 64^4 (the configuration the reference's O(N) figures are
quoted on) in single and double precision: site updates per second from HIP events."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import qcdgpu_amd as q  # noqa: E402

L = (64, 64, 64, 64)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for precision in (0, 1):
    with q.OnContext(L, precision=precision, nhit=10, rand_series=7) as c:
        c.init_lattice()
        c.sweep(5)
        ms = c.timed_sweeps(n)
        sites = c.sites
        m = c.measure()
        cnt = c.counters()
    print("precision %d: %.3f ms per sweep, %.3e site updates/s, action %.6f field %.6f, "
          "acceptance sums even/odd per site*nhit %.3f %.3f" %
          (precision, ms / n, sites * n / (ms * 1e-3), m[0], m[1], cnt[1] / (sites / 2 * 10), cnt[2] / (sites / 2 * 10)), flush=True)
