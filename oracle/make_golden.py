#!/usr/bin/env python3
"""Mint golden vectors from the compiled reference (oracle/_ref/qcdgpu_cpu).

TEST INFRASTRUCTURE ONLY.  Runs the reference's own sequential CPU path (built by
`make -C oracle ref` from the sources under /root/reference) on a matrix of small
init.dat configurations, parses the "% 16.13e" data table it writes
(SU(N)/suncl/suncl.cpp:1822-1850) and stores inputs + expected outputs as JSON under
tests/golden/.  The fixtures are data only; the reference binary and its sources never
enter the repository.

usage: python3 oracle/make_golden.py [case ...]   (needs oracle/_ref/qcdgpu_cpu; no names: every case)
"""
import json
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.path.join(HERE, "_ref", "qcdgpu_cpu")
OUT = os.path.join(HERE, "..", "tests", "golden")

# name -> init.dat keys (order matters: ND before LT)
CASES = {
    "su2_8x8x8x8_cold_rl3": dict(GROUP=2, ND=4, LS=8, LT=8, BETA=2.5, NHIT=10, INTS=1,
                                 PRNG="RANLUX3", RANDSERIES=54, NAV=2, ITER=4, NITER=1),
    "su3_4x4x4x4_gid": dict(GROUP=3, ND=4, LS=4, LT=4, BETA=6.0, NHIT=10, INTS=2,
                            PRNG="RANLUX3", RANDSERIES=54, NAV=0, ITER=3, NITER=1),
    "su3_8x8x8x8_cold_rl3": dict(GROUP=3, ND=4, LS=8, LT=8, BETA=6.0, NHIT=10, INTS=1,
                                 PRNG="RANLUX3", RANDSERIES=54, NAV=0, ITER=3, NITER=1),
    "su3_8x8x8x8_gid": dict(GROUP=3, ND=4, LS=8, LT=8, BETA=6.0, NHIT=10, INTS=2,
                            PRNG="RANLUX3", RANDSERIES=54, NAV=0, ITER=4, NITER=1),
    "su3_16x16x16x16_gid": dict(GROUP=3, ND=4, LS=16, LT=16, BETA=6.0, NHIT=10, INTS=2,
                                PRNG="RANLUX3", RANDSERIES=54, NAV=0, ITER=3, NITER=1),
    "su3_16x16x16x16_cold_rl3": dict(GROUP=3, ND=4, LS=16, LT=16, BETA=6.0, NHIT=10, INTS=1,
                                     PRNG="RANLUX3", RANDSERIES=54, NAV=0, ITER=3, NITER=1),
    "su3_4x6x8x10_gid": dict(GROUP=3, ND=4, L1=4, L2=6, L3=8, L4=10, BETA=5.7, NHIT=4, INTS=2,
                             PRNG="RANLUX3", RANDSERIES=7, NAV=1, ITER=3, NITER=2),
    "su3_6x4x10x8_cold_rl0": dict(GROUP=3, ND=4, L1=6, L2=4, L3=10, L4=8, BETA=5.5, NHIT=3,
                                  INTS=1, PRNG="RANLUX0", RANDSERIES=1234, NAV=1, ITER=3,
                                  NITER=1),
    "su2_4x4x4x4_gid": dict(GROUP=2, ND=4, LS=4, LT=4, BETA=2.3, NHIT=10, INTS=2,
                            PRNG="RANLUX3", RANDSERIES=54, NAV=0, ITER=4, NITER=1),
    "su2_6x8x4x10_gid": dict(GROUP=2, ND=4, L1=6, L2=8, L3=4, L4=10, BETA=2.2, NHIT=5, INTS=2,
                             PRNG="RANLUX3", RANDSERIES=54, NAV=1, ITER=3, NITER=2),
    "su2_6x6x6x12_cold_rl2": dict(GROUP=2, ND=4, LS=6, LT=12, BETA=1.2, NHIT=2, INTS=1,
                                  PRNG="RANLUX2", RANDSERIES=99, NAV=1, ITER=4, NITER=2),
    "su2_16x16x16x16_gid": dict(GROUP=2, ND=4, LS=16, LT=16, BETA=2.5, NHIT=10, INTS=2,
                                PRNG="RANLUX3", RANDSERIES=54, NAV=0, ITER=3, NITER=1),
    # odd extents (the reference needs only L2 even): the sequential order decides which links a site sees updated
    "su3_5x4x3x7_gid": dict(GROUP=3, ND=4, L1=5, L2=4, L3=3, L4=7, BETA=5.7, NHIT=4, INTS=2,
                            PRNG="RANLUX3", RANDSERIES=7, NAV=1, ITER=3, NITER=2),
    "su2_4x6x4x5_gid": dict(GROUP=2, ND=4, L1=4, L2=6, L3=4, L4=5, BETA=2.3, NHIT=5, INTS=2,
                            PRNG="RANLUX3", RANDSERIES=54, NAV=0, ITER=4, NITER=1),
    "su3_3x2x3x3_gid": dict(GROUP=3, ND=4, L1=3, L2=2, L3=3, L4=3, BETA=6.0, NHIT=10, INTS=2,
                            PRNG="RANLUX3", RANDSERIES=54, NAV=0, ITER=4, NITER=1),
    "su3_4x4x4x7_cold_rl3": dict(GROUP=3, ND=4, L1=4, L2=4, L3=4, L4=7, BETA=6.0, NHIT=10, INTS=1,
                                 PRNG="RANLUX3", RANDSERIES=54, NAV=0, ITER=3, NITER=1),
    # the benchmark's lattice: one deterministic sweep of the reference at full size
    "su3_32x32x32x32_gid": dict(GROUP=3, ND=4, LS=32, LT=32, BETA=6.0, NHIT=10, INTS=2,
                                PRNG="RANLUX3", RANDSERIES=54, NAV=0, ITER=2, NITER=1),
    "su3_32x32x32x32_cold_rl3": dict(GROUP=3, ND=4, LS=32, LT=32, BETA=6.0, NHIT=10, INTS=1,
                                     PRNG="RANLUX3", RANDSERIES=54, NAV=0, ITER=2, NITER=1),
    "su3_8x8x8x8_cold_rl4_lowbeta": dict(GROUP=3, ND=4, LS=8, LT=8, BETA=1.5, NHIT=2, INTS=1,
                                         PRNG="RANLUX4", RANDSERIES=4321, NAV=0, ITER=3,
                                         NITER=1),
}

FIELDS = ["Plq_spat", "Plq_temp", "Plq_total", "S_spat", "S_temp", "S_total"]


def run_case(name, keys):
    tmp = tempfile.mkdtemp(prefix="qcdref_", dir=os.path.join(HERE, "_ref"))
    try:
        with open(os.path.join(tmp, "init.dat"), "w") as f:
            for k, v in keys.items():
                f.write("%s = %s\n" % (k, v))
            f.write("OUTPUTPATH = %s/\n" % tmp)
        subprocess.run([REF, os.path.join(tmp, "init.dat")], check=True, cwd=tmp,
                       stdout=subprocess.DEVNULL)
        outs = [p for p in os.listdir(tmp) if p.endswith(".txt")]
        assert len(outs) == 1, outs
        rows, header = [], {}
        with open(os.path.join(tmp, outs[0])) as f:
            in_table = False
            for line in f:
                if line.startswith(" Data fields"):
                    in_table = True
                    continue
                if not in_table:
                    for key in ("Mean S_total", "Variance S_total", "Mean Plq_total",
                                "Variance Plq_total"):
                        if line.strip().startswith(key):
                            header[key] = float(line.split(":")[1])
                    continue
                parts = line.split()
                if len(parts) == 7 and parts[0].isdigit():
                    rows.append([float(x) for x in parts[1:]])
        assert len(rows) == keys["ITER"], (name, len(rows))
        return {"name": name, "init": keys, "fields": FIELDS, "samples": rows,
                "summary": header,
                "source": "reference suncpp CPU path (CPU_RUN build), g++ -O2, glibc 2.35"}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    if not os.path.exists(REF):
        sys.exit("build the reference first: make -C oracle ref")
    os.makedirs(OUT, exist_ok=True)
    for name, keys in CASES.items():
        if len(sys.argv) > 1 and name not in sys.argv[1:]:   # case names on the command line: only those
            continue
        res = run_case(name, keys)
        with open(os.path.join(OUT, "ref_%s.json" % name), "w") as f:
            json.dump(res, f, indent=1)
        print(name, "last Plq_total = %.13e" % res["samples"][-1][2])


if __name__ == "__main__":
    main()
