#!/usr/bin/env python3
"""Mint the statistical fixtures: long runs of the compiled reference (oracle/_ref/qcdgpu_cpu,
its own sequential CPU path and single random stream), of which only the thermalised samples'
Plq_total series is kept.  TEST INFRASTRUCTURE ONLY.  The GPU engine uses one random stream per
thread, so its chain is a different one; what must agree is the mean plaquette, within errors.

usage: python3 oracle/make_golden_stat.py [case ...]   (needs oracle/_ref/qcdgpu_cpu; about 5 min for all)
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as mg

CASES = {
    "stat_su3_8x8x8x8_b6": dict(GROUP=3, ND=4, LS=8, LT=8, BETA=6.0, NHIT=10, INTS=1,
                                PRNG="RANLUX3", RANDSERIES=54, NAV=100, ITER=1201, NITER=1),
    "stat_su2_8x8x8x8_b2.5": dict(GROUP=2, ND=4, LS=8, LT=8, BETA=2.5, NHIT=10, INTS=1,
                                  PRNG="RANLUX3", RANDSERIES=54, NAV=100, ITER=1201, NITER=1),
    # the benchmark's 16^4 lattice: 500 samples
    "stat_su3_16x16x16x16_b6": dict(GROUP=3, ND=4, LS=16, LT=16, BETA=6.0, NHIT=10, INTS=1,
                                    PRNG="RANLUX3", RANDSERIES=54, NAV=100, ITER=501, NITER=1),
    # the benchmark's 32^4 lattice: 150 samples (more than an hour of the reference's program on one core)
    "stat_su3_32x32x32x32_b6": dict(GROUP=3, ND=4, LS=32, LT=32, BETA=6.0, NHIT=10, INTS=1,
                                    PRNG="RANLUX3", RANDSERIES=54, NAV=100, ITER=151, NITER=1),
}


def main():
    if not os.path.exists(mg.REF):
        sys.exit("build the reference first: make -C oracle ref")
    for name, keys in CASES.items():
        if len(sys.argv) > 1 and name not in sys.argv[1:]:   # case names on the command line: only those
            continue
        res = mg.run_case(name, keys)
        series = [row[2] for row in res["samples"][1:]]   # sample 0 is the cold start
        doc = {"name": name, "init": keys, "plq_total": series, "summary": res["summary"],
               "source": res["source"]}
        with open(os.path.join(mg.OUT, "%s.json" % name), "w") as f:
            json.dump(doc, f)
        m = sum(series) / len(series)
        print(name, "mean Plq_total over %d samples = %.6f" % (len(series), m))


if __name__ == "__main__":
    main()
