import sys; sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import qcdgpu_amd as q
for beta in (2.5,):
    c = q.Context(group=2, L=(32,32,32,32), beta=beta); c.init_lattice(); c.sweep(3)
    best = 1e9
    for _ in range(3):
        ms, n = c.timed_sweeps(5); best = min(best, ms / n)
    print("SU(2) beta %.1f: %8.1f us/launch  %6.2f Gupd/s  -> %.0f GB/s algorithmic" % (beta, 1e3 * best, 524288 / best / 1e6, 524288*288/best/1e6))
