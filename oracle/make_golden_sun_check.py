#!/usr/bin/env python3
"""Mint golden vectors for the SU(3) and SU(2) observables that have no CPU path in the reference's
sequential program (Polyakov loop, Wilson loop, F_mu_nu) from its CPU re-measurement code.

TEST INFRASTRUCTURE ONLY.  Configurations come from this repo's checker (a few heat-bath sweeps);
the EXPECTED values come from oracle/_ref/qcdgpu_sun_check, i.e. SU3_CPU::SU::lattice_check_cpu
(SU(N)/suncl/su3cpu.cpp:1616-1704, and its SU(2) twin su2cpu.cpp:1120-1210) compiled from the reference's sources by `make -C oracle ref_sun`.
Per case: tests/golden/chk_su<N>_<name>.npy = the link table in the reference's device layout (12 or 4
rows of double4), tests/golden/chk_su<N>_<name>.json = parameters and expected outputs.

usage: python3 oracle/make_golden_su3_check.py
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

REF = os.path.join(HERE, "_ref", "qcdgpu_sun_check")
OUT = os.path.join(HERE, "..", "tests", "golden")

# field: 1 = H (planes xy, xz, yz), 2 = E (planes xt, yt, zt)
CASES = {
    "4x4x4x4_H38": dict(L=(4, 4, 4, 4), beta=5.9, nhit=8, sweeps=3, field=1, idx=(3, 8), wilson=(1, 2)),
    "4x6x4x6_E15": dict(L=(4, 6, 4, 6), beta=5.6, nhit=6, sweeps=2, field=2, idx=(1, 5), wilson=(2, 3)),
    "6x4x4x4_H47": dict(L=(6, 4, 4, 4), beta=2.0, nhit=4, sweeps=2, field=1, idx=(4, 7), wilson=(2, 2)),
    "4x4x6x4_E26": dict(L=(4, 4, 6, 4), beta=6.2, nhit=10, sweeps=4, field=2, idx=(2, 6), wilson=(3, 1)),
    # SU(2): Pauli matrices, index1 3 or 1, index2 2 (SU(N)/suncl/suncl.cpp:2094-2101)
    "su2_4x4x4x6": dict(group=2, L=(4, 4, 4, 6), beta=2.3, nhit=6, sweeps=3, field=1, idx=(3, 2), wilson=(2, 2)),
    "su2_6x4x4x4": dict(group=2, L=(6, 4, 4, 4), beta=1.5, nhit=4, sweeps=2, field=2, idx=(1, 2), wilson=(1, 3)),
    # twisted boundary condition (PHI, OMEGA): the re-measurement multiplies U_y at x = N1 - 1 by Omega on every read
    "4x4x6x4_H38_tw": dict(L=(4, 4, 6, 4), beta=5.8, nhit=8, sweeps=3, field=1, idx=(3, 8), wilson=(2, 2), phi=0.7, omega=-0.4),
    "6x4x4x4_E38_tw": dict(L=(6, 4, 4, 4), beta=6.0, nhit=8, sweeps=2, field=2, idx=(3, 8), wilson=(3, 2), phi=0.0, omega=1.1),
    "su2_4x6x4x4_tw": dict(group=2, L=(4, 6, 4, 4), beta=2.4, nhit=6, sweeps=3, field=1, idx=(3, 2), wilson=(2, 2), phi=0.9, omega=0.0),
}


def main():
    if not os.path.exists(REF):
        sys.exit("build the reference checker first: make -C oracle ref_sun")
    for name, c in CASES.items():
        group = c.get("group", 3)
        stem = ("chk_su3_%s" % name) if group == 3 else ("chk_%s" % name)
        lat = ol.Lattice(group, c["L"], c["beta"], nhit=c["nhit"], ints=1, rng_mode=ol.RNG_PERSITE)
        lat.sweep(c["sweeps"])
        rowsize = int(np.prod(c["L"]))
        table = lat.export_table(rowsize)
        lat.close()
        with tempfile.NamedTemporaryFile(suffix=".bin", dir=os.path.join(HERE, "_ref")) as f:
            table.tofile(f.name)
            args = [REF, str(group)] + [str(v) for v in c["L"]] + [str(rowsize), repr(c["beta"]), str(c["field"]), str(c["idx"][0]),
                                                      str(c["idx"][1]), str(c["wilson"][0]), str(c["wilson"][1]), f.name]
            if "phi" in c:
                args += [repr(c["phi"]), repr(c["omega"])]
            out = subprocess.run(args, check=True, capture_output=True, text=True).stdout
        exp = {}
        for line in out.splitlines():
            k, v = line.split()
            exp[k] = float(v)
        np.save(os.path.join(OUT, stem + ".npy"), table)
        doc = dict(source="oracle/_ref/qcdgpu_sun_check (reference SU(N)/suncl/su3cpu.cpp / su2cpu.cpp lattice_check_cpu, double precision)",
                   group=group, L=list(c["L"]), beta=c["beta"], field=c["field"], index1=c["idx"][0], index2=c["idx"][1],
                   wilson_R=c["wilson"][0], wilson_T=c["wilson"][1], table=stem + ".npy", expected=exp,
                   phi=c.get("phi", 0.0), omega=c.get("omega", 0.0))
        with open(os.path.join(OUT, stem + ".json"), "w") as fo:
            json.dump(doc, fo, indent=1)
        print(name, {k: exp[k] for k in ("Plq_spat", "Polyakov_loop", "Wilson_loop", "F_a_3_re")})


if __name__ == "__main__":
    main()
