#!/usr/bin/env python3
"""Mint the 32^4 fixture of the per-thread-stream mode: this repo's CPU checker (oracle/qcd_oracle.c,
rng_mode = persite: one RANLUX stream per thread of a pass, seeded like the instances of the
reference's device generator) run for two heat-bath sweeps of SU(3) 32^4 from a cold start, with the
plaquette averages, the Polyakov loop and the count of random numbers drawn after every sweep.
TEST INFRASTRUCTURE ONLY.  The reference's own CPU program runs ONE sequential stream, a chain no
parallel pass can follow, so this fixture pins the GPU engine to the checker at the headline size,
and the checker is pinned to the reference by the ref_* and chk_* fixtures.

With an argument N (2 or 4) the streams are shared by N sites of one parity (qcd_params.rng_share, the engine's
choice at this size is 4): fixture persite_su3_32x32x32x32_b6_shareN.json.

usage: python3 oracle/make_golden_persite32.py [share]        (about two minutes on one core)
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import oracle_lib as ol   # noqa: E402


def main():
    share = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    name = "persite_su3_32x32x32x32_b6" + ("_share%d" % share if share > 1 else "")
    L = (32, 32, 32, 32)
    keys = dict(GROUP=3, ND=4, LS=32, LT=32, BETA=6.0, NHIT=10, INTS=1, PRNG="RANLUX3", RANDSERIES=54)
    lat = ol.Lattice(3, L, 6.0, nhit=10, ints=1, rng_mode=ol.RNG_PERSITE, lux=3, seed=54, share=share)
    rows = []
    for s in range(1, 3):
        lat.sweep(1)
        rows.append({"sweep": s, "plaquette": [float(v) for v in lat.plaquette()],
                     "polyakov": [float(v) for v in lat.polyakov()], "rng_draws": int(ol.rng_draws(lat))})
        print(rows[-1], flush=True)
    doc = {"name": name, "init": keys, "rng_mode": "persite", "rng_share": share, "sweeps": rows,
           "source": "oracle/qcd_oracle.c, rng_mode persite (oracle/make_golden_persite32.py)"}
    with open(os.path.join(ROOT, "tests", "golden", name + ".json"), "w") as f:
        json.dump(doc, f, indent=1)
    lat.close()


if __name__ == "__main__":
    main()
