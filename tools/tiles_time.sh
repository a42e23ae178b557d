#!/bin/bash
# Times the update pass for different workgroup block shapes (tz,tt,tx: rows of a 64-site block in
# z, t, x; 0,0,0 = consecutive half-sites) through the diagnostic switch QCDGPU_AMD_TILE.
for t in 0,0,0 4,1,1 2,2,1 2,1,2 1,2,2 1,4,1 1,1,4; do
  echo -n "tile $t: "; QCDGPU_AMD_TILE=$t timeout -k 10 100 python3 - 2>/dev/null <<'PY' || exit 1
import sys; sys.path.insert(0, ".")
import qcdgpu_amd as q
c = q.Context(group=3, L=(32,32,32,32), beta=6.0); c.init_lattice(); c.sweep(3)
best = 1e9
for _ in range(3):
    ms, n = c.timed_sweeps(5); best = min(best, ms / n)
print("%8.1f us/launch" % (1e3 * best))
PY
done
