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


def _build_program(target, sources, headers, force, verbose, libname="qcdgpu_amd"):
    srcs = [os.path.join(HOST_DIR, f) for f in sources]
    deps = srcs + [os.path.join(HOST_DIR, h) for h in headers] + [os.path.join(os.path.dirname(LIB), "lib%s.so" % libname)]
    if (not force and os.path.exists(target)
            and all(os.path.getmtime(d) <= os.path.getmtime(target) for d in deps)):
        return target
    os.makedirs(os.path.dirname(target), exist_ok=True)
    cmd = ["g++", "-O2", "-std=c++17", "-Wall"] + srcs + ["-o", target, "-L" + os.path.dirname(LIB),
           "-l" + libname, "-Wl,-rpath,$ORIGIN/../lib"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return target


def build_host(force=False, verbose=False):
    """The C++ host shells (run API + program entry) above the C-ABI: the SU(N) program and the
    O(N) one (the reference builds them from two source trees)."""
    _build_program(ON_BIN, ON_HOST_SOURCES, ["model.h", "on_model.h"], force, verbose)
    # the SU(N) program once per precision (the reference sets the precision in its main, SU(N)/QCDGPU.cpp:178-179)
    _build_program(HOST_BIN_F32, HOST_SOURCES, ["model.h"], force, verbose, libname="qcdgpu_amd_f32")
    return _build_program(HOST_BIN, HOST_SOURCES, ["model.h"], force, verbose)


# The same sources built a second time for single precision: links, staples and the heat-bath in float
# (the reference chooses the precision when it builds its kernels, -D PRECISION).  Its C++ names live in
# another namespace, so that both libraries can be loaded into one process and tell their kernels apart
# in a profile.
LIB_F32 = os.path.join(HERE, "lib", "libqcdgpu_amd_f32.so")
F32_FLAGS = ["-DQCD_REAL=float", "-DQCD_NS=qcd_f32"]
HOST_BIN_F32 = os.path.join(HERE, "bin", "qcdgpu_f32")


def _stale(target):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return not all(os.path.getmtime(os.path.join(CSRC, f)) <= t for f in SOURCES + [ON_SOURCE] + HEADERS)


def up_to_date():
    return not _stale(LIB) and not _stale(LIB_F32)


def _build_lib(target, extra_flags, verbose):
    os.makedirs(os.path.dirname(target), exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    libdir = os.path.dirname(target)
    tag = os.path.splitext(os.path.basename(target))[0]
    common = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC"] + extra_flags
    objs, procs = [], []
    for src, extra in [(SOURCES[0], ["-Xarch_host", "-ffp-contract=off"]), (SOURCES[1], []),
                       (ON_SOURCE, ["-ffp-contract=off"])]:
        obj = os.path.join(libdir, tag + "." + os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        cmd = [hipcc] + common + extra + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((cmd, subprocess.Popen(cmd, cwd=CSRC)))
    for cmd, pr in procs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    link = [hipcc, "--offload-arch=gfx950", "--hip-link", "-shared", "-fPIC", "-o", target] + objs + ["-ldl"]
    if verbose:
        print(" ".join(link), file=sys.stderr)
    subprocess.run(link, check=True, cwd=CSRC)
    for obj in objs:
        os.remove(obj)


def build(force=False, verbose=False):
    if force or _stale(LIB):
        _build_lib(LIB, [], verbose)
    if force or _stale(LIB_F32):
        _build_lib(LIB_F32, F32_FLAGS, verbose)
    build_host(force, verbose)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
