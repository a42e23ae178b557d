"""ctypes binding of the CPU checker (oracle/libqcd_oracle.so).  Test infrastructure only."""
import ctypes as C
import json
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")

RNG_SEQ, RNG_GID, RNG_PERSITE = 0, 1, 2


def _load():
    so = os.path.join(ORACLE_DIR, "libqcd_oracle.so")
    src = os.path.join(ORACLE_DIR, "qcd_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.run(["make", "-C", ORACLE_DIR, "libqcd_oracle.so", "qcd_oracle_cli"], check=True,
                       stdout=subprocess.DEVNULL)
    lib = C.CDLL(so)
    I, D, U, P = C.c_int, C.c_double, C.c_uint, C.c_void_p
    pd = C.POINTER(C.c_double)
    pi = C.POINTER(C.c_int)
    lib.qo_create.restype = P
    lib.qo_create.argtypes = [I, pi, D, I, I, I, I, U]
    lib.qo_set_twist.argtypes = [P, D, D]
    lib.qo_set_twist.restype = None
    lib.qo_update_pass_range.argtypes = [P, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.qo_update_pass_range.restype = None
    lib.qo_set_rng_share.argtypes = [P, C.c_int]
    lib.qo_set_rng_share.restype = C.c_int
    lib.qo_destroy.argtypes = [P]
    lib.qo_init_links.argtypes = [P]
    lib.qo_sweep.argtypes = [P]
    lib.qo_update_pass.argtypes = [P, I, I]
    lib.qo_overrelax_pass.argtypes = [P, I, I]
    lib.qo_overrelax_sweep.argtypes = [P]
    lib.qo_wilson_loop.argtypes = [P, I, I]
    lib.qo_wilson_loop.restype = C.c_double
    lib.qo_plaquette.argtypes = [P, pd]
    lib.qo_action.argtypes = [P, pd]
    lib.qo_polyakov.argtypes = [P, pd]
    lib.qo_fmunu_sums.argtypes = [P, I, I, I, pd]
    lib.qo_plaquette_profile_sums.argtypes = [P, I, pd]
    lib.qo_polyakov_profile_sums.argtypes = [P, I, pd]
    lib.qo_simulate.argtypes = [P, I, I, I, pd, pd]
    lib.qo_estimates.argtypes = [pd, I, pd, pd]
    lib.qo_export_device_layout.argtypes = [P, pd, I]
    lib.qo_import_device_layout.argtypes = [P, pd, I]
    lib.qo_stream_jseed.restype = I
    lib.qo_stream_jseed.argtypes = [U, I]
    lib.qo_odd_gid.restype = U
    lib.qo_odd_gid.argtypes = [pi, U]
    lib.qo_even_gid.restype = U
    lib.qo_even_gid.argtypes = [pi, U]
    lib.qo_neighbour.restype = I
    lib.qo_neighbour.argtypes = [pi, I, I]
    lib.qo_neighbour_back.restype = I
    lib.qo_neighbour_back.argtypes = [pi, I, I]
    return lib


lib = _load()


class RanluxState(C.Structure):
    _fields_ = [("seeds", C.c_float * 24), ("i24", C.c_int), ("j24", C.c_int),
                ("in24", C.c_int), ("carry", C.c_float), ("nskip", C.c_int)]


lib.qo_ranlux_init.argtypes = [C.POINTER(RanluxState), C.c_int, C.c_int]
lib.qo_ranlux_next.restype = C.c_float
lib.qo_ranlux_next.argtypes = [C.POINTER(RanluxState)]


class GlibcRand(C.Structure):
    _fields_ = [("r", C.c_int32 * 34), ("idx", C.c_int), ("ring", C.c_uint32 * 31),
                ("f", C.c_int), ("b", C.c_int)]


lib.qo_srand.argtypes = [C.POINTER(GlibcRand), C.c_uint]
lib.qo_rand.restype = C.c_int
lib.qo_rand.argtypes = [C.POINTER(GlibcRand)]


def _L(L):
    return (C.c_int * 4)(*L)


class Lattice:
    """One oracle lattice (links AoS [gid*nd+dir], full matrices)."""

    def __init__(self, group, L, beta, nhit=10, ints=1, rng_mode=RNG_SEQ, lux=3, seed=54, phi=0.0, omega=0.0, share=1):
        self.group, self.L = group, tuple(L)
        self.sites = int(np.prod(L))
        self.h = lib.qo_create(group, _L(L), beta, nhit, ints, rng_mode, lux, seed)
        if share != 1:   # sites of one parity per stream in the per-thread-stream mode (qcd_params.rng_share)
            if lib.qo_set_rng_share(self.h, share):
                raise ValueError("share = %d does not divide Nt = %d" % (share, L[3]))
        lib.qo_init_links(self.h)
        if phi or omega:
            lib.qo_set_twist(self.h, phi, omega)   # twisted boundary condition (PHI, OMEGA)

    def close(self):
        if self.h:
            lib.qo_destroy(self.h)
            self.h = None

    __del__ = close

    def sweep(self, n=1):
        for _ in range(n):
            lib.qo_sweep(self.h)

    def update(self, parity, direction):
        lib.qo_update_pass(self.h, parity, direction)

    def update_range(self, parity, direction, x_begin, x_end):
        """The pass restricted to the sites with x in [x_begin, x_end): what one slab of a cut lattice does."""
        lib.qo_update_pass_range(self.h, parity, direction, x_begin, x_end)

    def overrelax(self, parity, direction):
        lib.qo_overrelax_pass(self.h, parity, direction)

    def overrelax_sweep(self, n=1):
        for _ in range(n):
            lib.qo_overrelax_sweep(self.h)

    def plaquette(self):
        out = (C.c_double * 3)()
        lib.qo_plaquette(self.h, out)
        return np.array(out[:])

    def action(self):
        out = (C.c_double * 3)()
        lib.qo_action(self.h, out)
        return np.array(out[:])

    def wilson_loop(self, R, T):
        return lib.qo_wilson_loop(self.h, R, T)

    def polyakov(self):
        out = (C.c_double * 4)()
        lib.qo_polyakov(self.h, out)
        return np.array(out[:])

    def fmunu_sums(self, field, index1=3, index2=8):
        out = (C.c_double * 12)()
        lib.qo_fmunu_sums(self.h, field, index1, index2, out)
        return np.array(out[:])

    def plaquette_profile_sums(self, axis):
        out = (C.c_double * (2 * self.L[axis]))()
        lib.qo_plaquette_profile_sums(self.h, axis, out)
        return np.array(out[:]).reshape(self.L[axis], 2)

    def polyakov_profile_sums(self, axis):
        out = (C.c_double * (2 * self.L[axis]))()
        lib.qo_polyakov_profile_sums(self.h, axis, out)
        return np.array(out[:]).reshape(self.L[axis], 2)

    def rows(self):
        return 12 if self.group == 3 else 4

    def export_table(self, rowsize=None):
        rowsize = rowsize or self.sites
        t = np.zeros((self.rows(), rowsize, 4), dtype=np.float64)
        lib.qo_export_device_layout(self.h, t.ctypes.data_as(C.POINTER(C.c_double)), rowsize)
        return t

    def import_table(self, t):
        t = np.ascontiguousarray(t, dtype=np.float64)
        lib.qo_import_device_layout(self.h, t.ctypes.data_as(C.POINTER(C.c_double)), t.shape[1])


class _LatticeHead(C.Structure):
    # leading fields of qo_lattice up to rng_draws (oracle/qcd_oracle.h)
    _fields_ = [("group", C.c_int), ("nd", C.c_int), ("L", C.c_int * 4), ("sites", C.c_int),
                ("nhit", C.c_int), ("beta", C.c_double), ("ints", C.c_int), ("rng_mode", C.c_int),
                ("nskip", C.c_int), ("randseries", C.c_uint), ("links", C.c_void_p),
                ("seq", RanluxState), ("streams", C.c_void_p), ("rng_draws", C.c_ulonglong)]


def rng_draws(lat):
    return C.cast(lat.h, C.POINTER(_LatticeHead)).contents.rng_draws


def simulate(group, L, beta, nhit, ints, rng_mode, lux, seed, nav, iter_, niter):
    h = lib.qo_create(group, _L(L), beta, nhit, ints, rng_mode, lux, seed)
    plq = np.zeros((iter_, 3))
    act = np.zeros((iter_, 3))
    lib.qo_simulate(h, nav, iter_, niter, plq.ctypes.data_as(C.POINTER(C.c_double)),
                    act.ctypes.data_as(C.POINTER(C.c_double)))
    lib.qo_destroy(h)
    return plq, act


def estimates(data):
    data = np.ascontiguousarray(data, dtype=np.float64)
    m, v = C.c_double(), C.c_double()
    lib.qo_estimates(data.ctypes.data_as(C.POINTER(C.c_double)), len(data), C.byref(m), C.byref(v))
    return m.value, v.value


def golden(name):
    with open(os.path.join(GOLDEN_DIR, name)) as f:
        return json.load(f)


def golden_persite32(share):
    """The checker's per-thread-stream run of SU(3) 32^4 (oracle/make_golden_persite32.py) with `share` sites of one
    parity per stream (qcd_params.rng_share; the engine chooses 4 at this size)."""
    return golden("persite_su3_32x32x32x32_b6%s.json" % ("_share%d" % share if share > 1 else ""))


def golden_cases(prefix="ref_"):
    return sorted(p for p in os.listdir(GOLDEN_DIR) if p.startswith(prefix) and p.endswith(".json"))


def init_to_args(init):
    """init.dat keys of a fixture -> (group, L, beta, nhit, ints, lux, seed, nav, iter, niter)."""
    L = [32, 32, 32, 32]
    for k, v in init.items():
        if k == "LS":
            L[0] = L[1] = L[2] = v
        elif k == "LT":
            L[3] = v
        elif k in ("L1", "L2", "L3", "L4"):
            L[int(k[1]) - 1] = v
    lux = int(init.get("PRNG", "RANLUX3")[-1])
    return dict(group=init["GROUP"], L=L, beta=init["BETA"], nhit=init["NHIT"], ints=init["INTS"],
                lux=lux, seed=init["RANDSERIES"], nav=init["NAV"], iter_=init["ITER"],
                niter=init["NITER"])


def gid_to_coords(L, gid):
    """(x, y, z, t) of site gid = y + z*Ny + t*Ny*Nz + x*Ny*Nz*Nt."""
    c = (C.c_int * 4)()
    lib.qo_gid_to_coords(_L(L), gid, c)
    return tuple(c[:])


# ---------------------------------------------------------------------------------------------
# O(N) scalar path (oracle/on_oracle.c)

def _bind_on():
    I, D, U, P = C.c_int, C.c_double, C.c_uint, C.c_void_p
    pd = C.POINTER(C.c_double)
    pi = C.POINTER(C.c_int)
    lib.qon_create.restype = P
    lib.qon_create.argtypes = [pi, I, I, I, D, D, D, D, D, I, U]
    lib.qon_destroy.argtypes = [P]
    lib.qon_init.argtypes = [P, I]
    lib.qon_update_pass.argtypes = [P, I]
    lib.qon_sweep.argtypes = [P]
    lib.qon_measure_action.restype = D
    lib.qon_measure_action.argtypes = [P]
    lib.qon_measure_field.argtypes = [P, pd]
    lib.qon_measure_correlators.argtypes = [P, pi, pd]
    lib.qon_get.restype = D
    lib.qon_get.argtypes = [P, I]
    lib.qon_set.argtypes = [P, I, D]


_bind_on()

# defaults of the reference's O(N) main (O(N)/QCDGPU.cpp:121-128)
ON_DEFAULTS = dict(skgroup=4, z=0.1561573990981998, lam=0.25, zeta=0.25, eta=0.1, b=9.661840260273483)


class QonStruct(C.Structure):
    _fields_ = [("L", C.c_int * 4), ("sites", C.c_int), ("precision", C.c_int), ("nhit", C.c_int),
                ("skgroup", C.c_int), ("z", C.c_double), ("lam", C.c_double), ("zeta", C.c_double),
                ("eta", C.c_double), ("b", C.c_double), ("sqrt_z_lambda_zeta", C.c_double),
                ("max_U", C.c_double), ("field", C.c_void_p), ("nskip", C.c_int),
                ("streams", C.c_void_p), ("rng_draws", C.c_ulonglong), ("acc", C.c_double * 2)]


class OnLattice:
    """The O(N) checker: field in site order gid = y + z*Ny + t*Ny*Nz + x*Ny*Nz*Nt."""

    def __init__(self, L, precision=1, nhit=10, lux_level=3, randseries=7, **par):
        p = dict(ON_DEFAULTS)
        p.update(par)
        self.L = tuple(L)
        self.precision = precision
        self.h = lib.qon_create(_L(L), precision, nhit, p["skgroup"], p["z"], p["lam"], p["zeta"],
                                p["eta"], p["b"], lux_level, randseries)
        self.s = C.cast(self.h, C.POINTER(QonStruct)).contents

    def close(self):
        if self.h:
            lib.qon_destroy(self.h)
            self.h = None

    def init(self, hot=False):
        lib.qon_init(self.h, int(hot))

    def update_pass(self, parity):
        lib.qon_update_pass(self.h, parity)

    def sweep(self, n=1):
        for _ in range(n):
            lib.qon_sweep(self.h)

    def action(self):
        return lib.qon_measure_action(self.h)

    def field_moments(self):
        out = (C.c_double * 2)()
        lib.qon_measure_field(self.h, out)
        return np.array(out[:])

    def correlators(self, step=(1, 1, 1, 1)):
        out = (C.c_double * 2)()
        lib.qon_measure_correlators(self.h, _L(step), out)
        return np.array(out[:])

    @property
    def field(self):
        dt = np.float64 if self.precision else np.float32
        n = self.s.sites
        buf = (C.c_char * (n * np.dtype(dt).itemsize)).from_address(self.s.field)
        return np.frombuffer(buf, dtype=dt)

    @property
    def rng_draws(self):
        return int(self.s.rng_draws)

    @property
    def acceptance(self):
        return np.array(self.s.acc[:])

    @property
    def max_U(self):
        return float(self.s.max_U)
