#!/usr/bin/env python3
"""Mint golden vectors for the O(N) scalar path from the compiled reference.

TEST INFRASTRUCTURE ONLY.  Fields come from this repo's checker (cold, hot, and after a few
Metropolis sweeps, so that they look like what a run produces); the EXPECTED values come from
oracle/_ref/qcdgpu_on_check, i.e. the reference's own CPU re-measurement
(O(N)/suncl/suncpu.cpp:93-270, 1805-1838) compiled from its sources by `make -C oracle ref_on`.
Inputs (parameters + the field itself) and expected outputs go to tests/golden/on_*.json.

usage: python3 oracle/make_golden_on.py
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
from tests import oracle_lib as ol  # noqa: E402

REF = os.path.join(HERE, "_ref", "qcdgpu_on_check")
OUT = os.path.join(HERE, "..", "tests", "golden")

CASES = {
    "on_4x4x4x4_cold": dict(L=(4, 4, 4, 4), hot=False, sweeps=0, step=(1, 1, 1, 1), par={}),
    "on_4x6x4x8_hot": dict(L=(4, 6, 4, 8), hot=True, sweeps=0, step=(2, 1, 3, 1), par={}),
    "on_6x4x6x4_hot_3sweeps": dict(L=(6, 4, 6, 4), hot=True, sweeps=3, step=(4, 4, 4, 4), par={}),
    "on_4x4x6x6_cold_5sweeps_eta0": dict(L=(4, 4, 6, 6), hot=False, sweeps=5, step=(1, 2, 3, 4),
                                         par=dict(eta=0.0, zeta=0.5, skgroup=3, z=0.2, b=8.0)),
}


def main():
    if not os.path.exists(REF):
        sys.exit("build the reference checker first: make -C oracle ref_on")
    for name, c in CASES.items():
        lat = ol.OnLattice(c["L"], precision=1, nhit=10, randseries=7, **c["par"])
        lat.init(hot=c["hot"])
        lat.sweep(c["sweeps"])
        field = np.array(lat.field, dtype=np.float64)
        p = dict(ol.ON_DEFAULTS)
        p.update(c["par"])
        with tempfile.NamedTemporaryFile(suffix=".bin", dir=os.path.join(HERE, "_ref")) as f:
            field.tofile(f.name)
            args = [REF] + [str(v) for v in c["L"]] + ["1", repr(p["z"]), repr(p["lam"]), repr(p["zeta"]),
                    repr(p["eta"]), repr(p["b"]), str(p["skgroup"])] + [str(v) for v in c["step"]] + [f.name]
            out = subprocess.run(args, check=True, capture_output=True, text=True).stdout.split()
        vals = [float(v) for v in out[-5:]]
        doc = dict(source="oracle/_ref/qcdgpu_on_check (reference O(N)/suncl/suncpu.cpp, double precision)",
                   L=list(c["L"]), params=p, correlator_step=list(c["step"]),
                   field=[float(v) for v in field],
                   expected=dict(action=vals[0], field=vals[1], field2=vals[2], correlator1=vals[3],
                                 correlator2=vals[4]))
        lat.close()
        with open(os.path.join(OUT, name + ".json"), "w") as fo:
            json.dump(doc, fo)
        print(name, vals)


if __name__ == "__main__":
    main()
