"""Builds libqcdgpu_amd.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib", "libqcdgpu_amd.so")
SOURCES = ["engine.hip", "ring.hip"]
# compiled apart, with contraction off on the device as well (see on_engine.hip)
ON_SOURCE = "on_engine.hip"
HEADERS = ["kernels.h", "lattice.h", "ranlux.h", "su_math.h", "host_seed.h", "fastmath.h", "on_kernels.h",
           os.path.join("..", "..", "include", "qcdgpu_amd.h")]
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC",
         "-Xarch_host", "-ffp-contract=off"]


HOST_DIR = os.path.join(CSRC, "host")
HOST_BIN = os.path.join(HERE, "bin", "qcdgpu")
HOST_SOURCES = ["model.cpp", "qcdgpu_main.cpp"]


ON_BIN = os.path.join(HERE, "bin", "qcdgpu_on")
ON_HOST_SOURCES = ["on_model.cpp", "qcdgpu_on_main.cpp", "model.cpp"]


def _build_program(target, sources, headers, force, verbose):
    srcs = [os.path.join(HOST_DIR, f) for f in sources]
    deps = srcs + [os.path.join(HOST_DIR, h) for h in headers] + [LIB]
    if (not force and os.path.exists(target)
            and all(os.path.getmtime(d) <= os.path.getmtime(target) for d in deps)):
        return target
    os.makedirs(os.path.dirname(target), exist_ok=True)
    cmd = ["g++", "-O2", "-std=c++17", "-Wall"] + srcs + ["-o", target, "-L" + os.path.dirname(LIB),
           "-lqcdgpu_amd", "-Wl,-rpath,$ORIGIN/../lib"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return target


def build_host(force=False, verbose=False):
    """The C++ host shells (run API + program entry) above the C-ABI: the SU(N) program and the
    O(N) one (the reference builds them from two source trees)."""
    _build_program(ON_BIN, ON_HOST_SOURCES, ["model.h", "on_model.h"], force, verbose)
    return _build_program(HOST_BIN, HOST_SOURCES, ["model.h"], force, verbose)


def up_to_date():
    if not os.path.exists(LIB):
        return False
    t = os.path.getmtime(LIB)
    return all(os.path.getmtime(os.path.join(CSRC, f)) <= t for f in SOURCES + [ON_SOURCE] + HEADERS)


def build(force=False, verbose=False):
    if not force and up_to_date():
        build_host(force, verbose)
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    libdir = os.path.dirname(LIB)
    common = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC"]
    objs, cmds = [], []
    for src, extra in [(SOURCES[0], ["-Xarch_host", "-ffp-contract=off"]), (SOURCES[1], []),
                       (ON_SOURCE, ["-ffp-contract=off"])]:
        obj = os.path.join(libdir, os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        cmds.append([hipcc] + common + extra + ["-c", src, "-o", obj])
    cmds.append([hipcc, "--offload-arch=gfx950", "--hip-link", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"])
    for c in cmds:
        if verbose:
            print(" ".join(c), file=sys.stderr)
        subprocess.run(c, check=True, cwd=CSRC)
    for obj in objs:
        os.remove(obj)
    build_host(force, verbose)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
