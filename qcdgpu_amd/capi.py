"""ctypes binding of libqcdgpu_amd.so (the C-ABI declared in include/qcdgpu_amd.h).

There is no CPU fallback: importing this module without the built library raises, and every
entry point that needs a device reports the HIP error it got.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("QCDGPU_AMD_LIB") or os.path.join(_HERE, "lib", "libqcdgpu_amd.so")
# the same entry points built for single precision (qcd_precision() = 1)
LIB_F32_PATH = os.environ.get("QCDGPU_AMD_LIB_F32") or os.path.join(_HERE, "lib", "libqcdgpu_amd_f32.so")

START_HOT, START_COLD, START_GID = 0, 1, 2


class QcdError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("qcdgpu_amd error %d: %s" % (code, msg))
        self.code = code


class Params(C.Structure):
    _fields_ = [("group", C.c_int), ("nd", C.c_int), ("L", C.c_int * 4), ("precision", C.c_int),
                ("beta", C.c_double), ("nhit", C.c_int), ("start_mode", C.c_int),
                ("lux_level", C.c_int), ("rand_series", C.c_uint), ("device", C.c_int),
                ("x0", C.c_int), ("nx_local", C.c_int), ("halo", C.c_int), ("phi", C.c_double), ("omega", C.c_double),
                ("rng_share", C.c_int)]


class OnParams(C.Structure):
    _fields_ = [("L", C.c_int * 4), ("precision", C.c_int), ("nhit", C.c_int), ("sk_group", C.c_int),
                ("z", C.c_double), ("lam", C.c_double), ("zeta", C.c_double), ("eta", C.c_double),
                ("b", C.c_double), ("start_mode", C.c_int), ("lux_level", C.c_int),
                ("rand_series", C.c_uint), ("device", C.c_int)]


class Timing(C.Structure):
    _fields_ = [("update_launches", C.c_ulonglong), ("update_ms", C.c_double),
                ("measure_launches", C.c_ulonglong), ("measure_ms", C.c_double),
                ("halo_launches", C.c_ulonglong), ("halo_ms", C.c_double)]


# every symbol include/qcdgpu_amd.h declares
EXPORTS = ["qcd_create", "qcd_destroy", "qcd_last_error", "qcd_precision", "qcd_set_stream", "qcd_reset_stream",
           "qcd_synchronize",
           "qcd_init_lattice", "qcd_update", "qcd_sweep", "qcd_reunitarize",
           "qcd_measure_plaquette", "qcd_measure_action", "qcd_measure_polyakov",
           "qcd_measure_plaquette_sums", "qcd_measure_polyakov_sums", "qcd_upload", "qcd_download",
           "qcd_halo_bytes", "qcd_halo_pack", "qcd_halo_unpack", "qcd_update_signal_edges", "qcd_halo_region", "qcd_halo_copy",
           "qcd_device_count", "qcd_device_name", "qcd_set_profiling",
           "qcd_get_timing", "qcd_timed_sweeps", "qcd_timed_overrelax_sweeps", "qcd_rng_draws", "qcd_prng_check",
           "qcd_rng_state_bytes", "qcd_rng_state_download", "qcd_rng_state_upload", "qcd_math_check", "qcd_trig_check",
           "qcd_overrelax", "qcd_overrelax_sweep", "qcd_measure_wilson", "qcd_update_range",
           "qcd_update_split", "qcd_measure_fmunu", "qcd_measure_fmunu_sums", "qcd_measure_plaquette_profile_sums", "qcd_measure_polyakov_profile_sums",
           "qcd_on_create", "qcd_on_destroy", "qcd_on_init_lattice", "qcd_on_update", "qcd_on_sweep",
           "qcd_on_timed_sweeps", "qcd_on_measure", "qcd_on_counters", "qcd_on_max_u", "qcd_on_upload",
           "qcd_on_download", "qcd_measure_plaquette_sums_static", "qcd_get_stream", "qcd_get_params", "qcd_update_kernel", "qcd_rng_share_for",
           "qcd_ring_create", "qcd_rccl_unique_id", "qcd_ring_create_rank", "qcd_ring_destroy", "qcd_ring_update",
           "qcd_ring_overrelax", "qcd_ring_sweep", "qcd_ring_sweep_chb2", "qcd_ring_fill_halos", "qcd_ring_synchronize", "qcd_ring_one_launch_passes"]


def _load(path=None):
    path = path or LIB_PATH
    # PyTorch-ROCm wheels bundle their own HIP/HSA runtime.  A process must end up with ONE copy:
    # when torch is importable it is imported first, so that this library binds to the runtime
    # torch has already loaded (torch initialised after a second runtime sees no GPUs).
    if not os.environ.get("QCDGPU_AMD_NO_TORCH"):
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    if not os.path.exists(path):
        raise ImportError("%s is missing: build it with `python -c 'import __graft_entry__ as g; "
                          "g.build()'` (hipcc --offload-arch=gfx950)" % path)
    lib = C.CDLL(path)
    P, I, D = C.c_void_p, C.c_int, C.c_double
    pd = C.POINTER(C.c_double)
    lib.qcd_create.argtypes = [C.POINTER(P), C.POINTER(Params)]
    lib.qcd_destroy.argtypes = [P]
    lib.qcd_destroy.restype = None
    lib.qcd_last_error.restype = C.c_char_p
    lib.qcd_set_stream.argtypes = [P, P]
    lib.qcd_reset_stream.argtypes = [P]
    lib.qcd_synchronize.argtypes = [P]
    lib.qcd_init_lattice.argtypes = [P]
    lib.qcd_update.argtypes = [P, I, I]
    lib.qcd_sweep.argtypes = [P, I]
    lib.qcd_reunitarize.argtypes = [P]
    for n in ("qcd_measure_plaquette", "qcd_measure_action", "qcd_measure_polyakov",
              "qcd_measure_plaquette_sums", "qcd_measure_polyakov_sums"):
        getattr(lib, n).argtypes = [P, pd]
    lib.qcd_measure_plaquette_sums_static.argtypes = [P, P]
    lib.qcd_get_stream.argtypes = [P, C.POINTER(P)]
    lib.qcd_get_params.argtypes = [P, C.POINTER(Params)]
    lib.qcd_update_kernel.argtypes = [P, I, C.POINTER(I)]
    lib.qcd_rng_share_for.argtypes = [C.POINTER(I), I]
    lib.qcd_ring_create.argtypes = [C.POINTER(P), C.POINTER(P), I, I]
    lib.qcd_rccl_unique_id.argtypes = [P, C.c_size_t]
    lib.qcd_ring_create_rank.argtypes = [C.POINTER(P), P, I, I, P, C.c_size_t]
    lib.qcd_ring_destroy.argtypes = [P]
    lib.qcd_ring_destroy.restype = None
    lib.qcd_ring_update.argtypes = [P, I, I]
    lib.qcd_ring_overrelax.argtypes = [P, I, I]
    lib.qcd_ring_sweep.argtypes = [P, I]
    lib.qcd_ring_sweep_chb2.argtypes = [P, I]
    lib.qcd_ring_fill_halos.argtypes = [P]
    lib.qcd_ring_synchronize.argtypes = [P]
    lib.qcd_ring_one_launch_passes.argtypes = [P, C.POINTER(C.c_ulonglong)]
    lib.qcd_upload.argtypes = [P, P, C.c_size_t, I]
    lib.qcd_download.argtypes = [P, P, C.c_size_t, I]
    lib.qcd_halo_bytes.argtypes = [P]
    lib.qcd_halo_bytes.restype = C.c_size_t
    lib.qcd_halo_pack.argtypes = [P, I, I, I, P]
    lib.qcd_halo_unpack.argtypes = [P, I, I, I, P]
    lib.qcd_halo_region.argtypes = [P, I, I, I, I, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(I)]
    lib.qcd_update_signal_edges.argtypes = [P, I, I, I, P, C.c_uint]
    lib.qcd_halo_copy.argtypes = [P, I, P, I, I, I]
    lib.qcd_device_count.argtypes = [C.POINTER(C.c_int)]
    lib.qcd_device_name.argtypes = [I, C.c_char_p, C.c_size_t]
    lib.qcd_set_profiling.argtypes = [P, I]
    lib.qcd_get_timing.argtypes = [P, C.POINTER(Timing)]
    lib.qcd_timed_sweeps.argtypes = [P, I, pd, C.POINTER(C.c_ulonglong)]
    lib.qcd_timed_overrelax_sweeps.argtypes = [P, I, pd, C.POINTER(C.c_ulonglong)]
    lib.qcd_rng_draws.argtypes = [P, C.POINTER(C.c_ulonglong)]
    # host-only helpers (no device needed)
    lib.qcd_host_ranlux_stream.argtypes = [C.c_uint, I, I, I, C.POINTER(C.c_float)]
    lib.qcd_prng_check.argtypes = [I, C.c_uint, I, I, I, C.POINTER(C.c_uint)]
    lib.qcd_math_check.argtypes = [I, I, C.POINTER(C.c_double)]
    lib.qcd_trig_check.argtypes = [I, I, C.POINTER(C.c_double)]
    lib.qcd_overrelax.argtypes = [P, I, I]
    lib.qcd_update_range.argtypes = [P, I, I, I, I]
    lib.qcd_update_split.argtypes = [P, I, I, I, I]
    lib.qcd_measure_wilson.argtypes = [P, I, I, C.POINTER(C.c_double)]
    lib.qcd_overrelax_sweep.argtypes = [P, I]
    lib.qcd_rng_state_bytes.restype = C.c_size_t
    lib.qcd_rng_state_bytes.argtypes = [P]
    lib.qcd_rng_state_download.argtypes = [P, C.c_void_p, C.c_size_t]
    lib.qcd_rng_state_upload.argtypes = [P, C.c_void_p, C.c_size_t]
    lib.qcd_measure_fmunu.argtypes = [P, I, I, I, pd]
    lib.qcd_measure_fmunu_sums.argtypes = [P, I, I, I, pd]
    lib.qcd_measure_plaquette_profile_sums.argtypes = [P, I, pd]
    lib.qcd_measure_polyakov_profile_sums.argtypes = [P, I, pd]
    lib.qcd_on_create.argtypes = [C.POINTER(P), C.POINTER(OnParams)]
    lib.qcd_on_destroy.argtypes = [P]
    lib.qcd_on_destroy.restype = None
    lib.qcd_on_init_lattice.argtypes = [P]
    lib.qcd_on_update.argtypes = [P, I]
    lib.qcd_on_sweep.argtypes = [P, I]
    lib.qcd_on_timed_sweeps.argtypes = [P, I, pd]
    lib.qcd_on_measure.argtypes = [P, C.POINTER(C.c_int), pd]
    lib.qcd_on_counters.argtypes = [P, C.POINTER(C.c_ulonglong)]
    lib.qcd_on_max_u.argtypes = [P, pd]
    lib.qcd_on_upload.argtypes = [P, C.c_void_p, C.c_size_t]
    lib.qcd_on_download.argtypes = [P, C.c_void_p, C.c_size_t]
    lib.qcd_host_gid_link.argtypes = [I, I, I, I, pd]
    lib.qcd_host_gid_randoms.argtypes = [I, I, I, C.POINTER(C.c_float)]
    lib.qcd_host_rlx_pack_check.argtypes = [C.c_uint, I, C.POINTER(C.c_uint)]
    return lib


lib = _load()
_lib_f32 = None


def lib_f32():
    """The single-precision build of the library, loaded on first use."""
    global _lib_f32
    if _lib_f32 is None:
        _lib_f32 = _load(LIB_F32_PATH)
    return _lib_f32


def _check(rc, l=None):
    if rc != 0:
        raise QcdError(rc, (l or lib).qcd_last_error().decode())


def rng_share_for(L, min_nx_local=None):
    """Sites per RANLUX stream the library would choose for lattice L cut into parts of at least min_nx_local slices."""
    return lib.qcd_rng_share_for((C.c_int * 4)(*L), L[0] if min_nx_local is None else min_nx_local)


class Context:
    """One x-slab of the lattice on one device (qcd_ctx)."""

    def __init__(self, group, L, beta, nhit=10, start_mode=START_COLD, lux_level=3, rand_series=54,
                 device=0, x0=0, nx_local=None, halo=0, precision=2, phi=0.0, omega=0.0, rng_share=0):
        self.params = Params()
        p = self.params
        p.phi, p.omega = phi, omega
        p.rng_share = rng_share   # sites per RANLUX stream: 0 = the library's choice from the full lattice
        p.group, p.nd, p.precision = group, 4, precision
        for i in range(4):
            p.L[i] = L[i]
        p.beta, p.nhit, p.start_mode = beta, nhit, start_mode
        p.lux_level, p.rand_series, p.device = lux_level, rand_series, device
        p.x0, p.nx_local, p.halo = x0, (L[0] if nx_local is None else nx_local), halo
        self.L = tuple(L)
        self.group = group
        self.sites = int(np.prod(L))
        self.local_sites = p.nx_local * L[1] * L[2] * L[3]
        # precision 2 = double: libqcdgpu_amd.so; 1 = single, 3 = mixed: libqcdgpu_amd_f32.so
        self.lib = lib if precision == 2 else lib_f32()
        self.dtype = np.float64 if precision == 2 else np.float32
        self.h = C.c_void_p()
        self._check(self.lib.qcd_create(C.byref(self.h), C.byref(p)))

    def _check(self, rc):
        _check(rc, self.lib)

    @property
    def rng_share(self):
        """Sites of one parity per RANLUX stream, as the library resolved it (qcd_params.rng_share)."""
        p = Params()
        self._check(self.lib.qcd_get_params(self.h, C.byref(p)))
        return p.rng_share

    def pass_uses_pair_kernel(self, xs_count=None):
        """Whether a heat-bath pass over xs_count owned slices (default: all) runs as k_update_gh."""
        v = C.c_int(0)
        self._check(self.lib.qcd_update_kernel(self.h, self.params.nx_local if xs_count is None else xs_count, C.byref(v)))
        return bool(v.value)

    def close(self):
        if getattr(self, "h", None):
            self.lib.qcd_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_stream(self, stream_handle):
        self._check(self.lib.qcd_set_stream(self.h, C.c_void_p(stream_handle)))

    def synchronize(self):
        self._check(self.lib.qcd_synchronize(self.h))

    def init_lattice(self):
        self._check(self.lib.qcd_init_lattice(self.h))

    def update(self, parity, direction):
        self._check(self.lib.qcd_update(self.h, parity, direction))

    def sweep(self, n=1):
        self._check(self.lib.qcd_sweep(self.h, n))

    def reunitarize(self):
        self._check(self.lib.qcd_reunitarize(self.h))

    def _measure(self, fn, n):
        out = (C.c_double * n)()
        self._check(fn(self.h, out))
        return np.array(out[:])

    def plaquette(self):
        return self._measure(self.lib.qcd_measure_plaquette, 3)

    def action(self):
        return self._measure(self.lib.qcd_measure_action, 3)

    def polyakov(self):
        return self._measure(self.lib.qcd_measure_polyakov, 4)

    def plaquette_sums(self):
        return self._measure(self.lib.qcd_measure_plaquette_sums, 2)

    def plaquette_sums_static(self, dev_ptr):
        """(sum_spat, sum_temp) into two doubles of device memory at dev_ptr, asynchronously, without
        any allocation (the entry point for callers that cannot allocate)."""
        self._check(self.lib.qcd_measure_plaquette_sums_static(self.h, C.c_void_p(dev_ptr)))

    def polyakov_sums(self):
        return self._measure(self.lib.qcd_measure_polyakov_sums, 4)

    def rows(self):
        return 12 if self.group == 3 else 4

    def upload(self, table):
        t = np.ascontiguousarray(table, dtype=self.dtype)
        assert t.ndim == 3 and t.shape[0] == self.rows() and t.shape[2] == 4
        self._check(self.lib.qcd_upload(self.h, t.ctypes.data_as(C.c_void_p), t.nbytes, t.shape[1]))

    def download(self, rowsize=None):
        rowsize = rowsize or self.local_sites
        t = np.zeros((self.rows(), rowsize, 4), dtype=self.dtype)
        self._check(self.lib.qcd_download(self.h, t.ctypes.data_as(C.c_void_p), t.nbytes, rowsize))
        return t

    def halo_bytes(self):
        return self.lib.qcd_halo_bytes(self.h)

    def halo_pack(self, side, direction, parity, dev_ptr):
        self._check(self.lib.qcd_halo_pack(self.h, side, direction, parity, C.c_void_p(dev_ptr)))

    def halo_unpack(self, side, direction, parity, dev_ptr):
        self._check(self.lib.qcd_halo_unpack(self.h, side, direction, parity, C.c_void_p(dev_ptr)))

    def halo_copy_to(self, pack_side, dst, recv_side, direction, parity):
        self._check(self.lib.qcd_halo_copy(self.h, pack_side, dst.h, recv_side, direction, parity))

    def set_profiling(self, on):
        self._check(self.lib.qcd_set_profiling(self.h, int(on)))

    def timing(self):
        t = Timing()
        self._check(self.lib.qcd_get_timing(self.h, C.byref(t)))
        return t

    def timed_sweeps(self, n):
        ms = C.c_double()
        launches = C.c_ulonglong()
        self._check(self.lib.qcd_timed_sweeps(self.h, n, C.byref(ms), C.byref(launches)))
        return ms.value, launches.value

    def timed_overrelax_sweeps(self, n):
        ms = C.c_double()
        launches = C.c_ulonglong()
        self._check(self.lib.qcd_timed_overrelax_sweeps(self.h, n, C.byref(ms), C.byref(launches)))
        return ms.value, launches.value

    def update_range(self, parity, direction, xs_begin, xs_count):
        self._check(self.lib.qcd_update_range(self.h, parity, direction, xs_begin, xs_count))

    def update_split(self, parity, direction, first_begin, first_count):
        """One pass: slices [first_begin, +first_count) on the context's stream, the rest beside them."""
        self._check(self.lib.qcd_update_split(self.h, parity, direction, first_begin, first_count))

    def wilson_loop(self, R, T):
        out = C.c_double()
        self._check(self.lib.qcd_measure_wilson(self.h, R, T, C.byref(out)))
        return out.value

    def fmunu(self, field, index1=3, index2=8, sums=False):
        """F_mu_nu traces: for (index1, index2) x three planes x (re, im); field 0 = xy,xz,yz, 1 = xt,yt,zt."""
        out = (C.c_double * 12)()
        fn = self.lib.qcd_measure_fmunu_sums if sums else self.lib.qcd_measure_fmunu
        self._check(fn(self.h, field, index1, index2, out))
        return np.array(out[:])

    def _profile_extent(self, axis):
        return self.params.nx_local if axis == 0 else self.L[axis]

    def plaquette_profile_sums(self, axis):
        """Raw (spat, temp) ReTr plaquette sums of the owned sites per coordinate along axis (0 x, 1 y, 2 z)."""
        n = self._profile_extent(axis)
        out = (C.c_double * (2 * n))()
        self._check(self.lib.qcd_measure_plaquette_profile_sums(self.h, axis, out))
        return np.array(out[:]).reshape(n, 2)

    def polyakov_profile_sums(self, axis):
        """Raw (Re P, Im P) Polyakov-loop sums per coordinate along axis (0 x, 1 y, 2 z)."""
        n = self._profile_extent(axis)
        out = (C.c_double * (2 * n))()
        self._check(self.lib.qcd_measure_polyakov_profile_sums(self.h, axis, out))
        return np.array(out[:]).reshape(n, 2)

    def overrelax(self, parity, direction):
        self._check(self.lib.qcd_overrelax(self.h, parity, direction))

    def overrelax_sweep(self, n=1):
        self._check(self.lib.qcd_overrelax_sweep(self.h, n))

    def rng_state(self):
        """The RANLUX streams' state as bytes (for checkpoints)."""
        n = self.lib.qcd_rng_state_bytes(self.h)
        buf = np.empty(n, dtype=np.uint8)
        self._check(self.lib.qcd_rng_state_download(self.h, buf.ctypes.data_as(C.c_void_p), n))
        return buf

    def set_rng_state(self, buf):
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        self._check(self.lib.qcd_rng_state_upload(self.h, buf.ctypes.data_as(C.c_void_p), buf.nbytes))

    def rng_draws(self):
        d = C.c_ulonglong()
        self._check(self.lib.qcd_rng_draws(self.h, C.byref(d)))
        return d.value


RING_COPY, RING_RCCL = 0, 1


class Ring:
    """A ring of slab contexts whose passes (update + halo exchange) are issued below the C-ABI (qcd_ring)."""

    def __init__(self, contexts, transport=RING_COPY, rank=None, world=None, unique_id=None):
        self.contexts = list(contexts)
        self.lib = self.contexts[0].lib   # the library (precision) of the ring's contexts
        self.h = C.c_void_p()
        if rank is None:
            arr = (C.c_void_p * len(self.contexts))(*[c.h for c in self.contexts])
            _check(self.lib.qcd_ring_create(C.byref(self.h), arr, len(self.contexts), transport), self.lib)
        else:
            buf = (C.c_char * 128).from_buffer_copy(unique_id)
            _check(self.lib.qcd_ring_create_rank(C.byref(self.h), self.contexts[0].h, rank, world, buf, 128), self.lib)

    @staticmethod
    def unique_id():
        buf = (C.c_char * 128)()
        _check(lib.qcd_rccl_unique_id(buf, 128))
        return bytes(buf)

    def close(self):
        if getattr(self, "h", None):
            self.lib.qcd_ring_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def update(self, parity, direction):
        _check(self.lib.qcd_ring_update(self.h, parity, direction), self.lib)

    def overrelax(self, parity, direction):
        _check(self.lib.qcd_ring_overrelax(self.h, parity, direction), self.lib)

    def sweep(self, n=1):
        _check(self.lib.qcd_ring_sweep(self.h, n), self.lib)

    def sweep_chb2(self, n=1):
        """n sweeps with ONE edge exchange per sweep (the reference's CHB2 build): not the uncut lattice's chain."""
        _check(self.lib.qcd_ring_sweep_chb2(self.h, n), self.lib)

    def fill_halos(self):
        _check(self.lib.qcd_ring_fill_halos(self.h), self.lib)

    def synchronize(self):
        _check(self.lib.qcd_ring_synchronize(self.h), self.lib)

    def one_launch_passes(self):
        """Passes of parts issued as one launch with the in-kernel edge signal so far."""
        n = C.c_ulonglong(0)
        _check(self.lib.qcd_ring_one_launch_passes(self.h, C.byref(n)), self.lib)
        return int(n.value)


def prng_check(device=0, rand_series=54, lux_level=3, streams=256, samples=240):
    """Device RANLUX against the host generator (the reference's PRNG::check); returns the
    number of values that differ."""
    bad = C.c_uint()
    _check(lib.qcd_prng_check(device, rand_series, lux_level, streams, samples, C.byref(bad)))
    return bad.value


def math_check(device=0, n=1 << 20):
    """Worst error (ulp) of the engine's own log(1-x), sqrt, 1/x, 1/sqrt against the host libm."""
    out = (C.c_double * 4)()
    _check(lib.qcd_math_check(device, n, out))
    return np.array(out[:])


def trig_check(device=0, n=1 << 20):
    """Worst absolute error of the engine's own fp64 cos(2 pi r), sin(2 pi r) against the host libm in long double."""
    out = (C.c_double * 2)()
    _check(lib.qcd_trig_check(device, n, out))
    return np.array(out[:])


def host_ranlux_stream(rand_series, stream, lux_level, n):
    out = (C.c_float * n)()
    _check(lib.qcd_host_ranlux_stream(rand_series, stream, lux_level, n, out))
    return np.array(out[:], dtype=np.float32)


def host_gid_link(group, gid, direction, sites):
    q = (C.c_double * 12)()
    _check(lib.qcd_host_gid_link(group, gid, direction, sites, q))
    return np.array(q[:])


def host_gid_randoms(gid, nhit, sites):
    r = (C.c_float * 6)()
    _check(lib.qcd_host_gid_randoms(gid, nhit, sites, r))
    return np.array(r[:], dtype=np.float32)


def host_rlx_pack_check(seed, n):
    """Generator states that do not survive the device's packed form (must be 0)."""
    bad = C.c_uint(0)
    _check(lib.qcd_host_rlx_pack_check(seed, n, C.byref(bad)))
    return bad.value


# defaults of the reference's O(N) main (O(N)/QCDGPU.cpp:121-128)
ON_DEFAULTS = dict(sk_group=4, z=0.1561573990981998, lam=0.25, zeta=0.25, eta=0.1, b=9.661840260273483)


class OnContext:
    """The O(N) scalar model on one device (qcd_on_ctx): one real field value per site,
    checkerboard multi-hit Metropolis (the reference's O(N) tree)."""

    def __init__(self, L, precision=0, nhit=10, start_mode=START_COLD, lux_level=3, rand_series=7,
                 device=0, **par):
        q = dict(ON_DEFAULTS)
        q.update(par)
        self.params = OnParams()
        p = self.params
        for i in range(4):
            p.L[i] = L[i]
        p.precision, p.nhit, p.sk_group = precision, nhit, q["sk_group"]
        p.z, p.lam, p.zeta, p.eta, p.b = q["z"], q["lam"], q["zeta"], q["eta"], q["b"]
        p.start_mode, p.lux_level, p.rand_series, p.device = start_mode, lux_level, rand_series, device
        self.L = tuple(L)
        self.sites = int(np.prod(L))
        self.dtype = np.float64 if precision else np.float32
        self.h = C.c_void_p()
        _check(lib.qcd_on_create(C.byref(self.h), C.byref(p)))

    def close(self):
        if getattr(self, "h", None):
            lib.qcd_on_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        self.close()

    def init_lattice(self):
        _check(lib.qcd_on_init_lattice(self.h))

    def update(self, parity):
        _check(lib.qcd_on_update(self.h, parity))

    def sweep(self, n=1):
        _check(lib.qcd_on_sweep(self.h, n))

    def timed_sweeps(self, n):
        ms = C.c_double()
        _check(lib.qcd_on_timed_sweeps(self.h, n, C.byref(ms)))
        return ms.value

    def measure(self, step=(1, 1, 1, 1)):
        """Per-site averages: action, field, field^2, correlator(+step), correlator(+1,+1,+1,+1)."""
        out = (C.c_double * 5)()
        _check(lib.qcd_on_measure(self.h, (C.c_int * 4)(*step), out))
        return np.array(out[:])

    def counters(self):
        """(random numbers consumed, acceptance sum of the last even pass, of the last odd pass)."""
        out = (C.c_ulonglong * 3)()
        _check(lib.qcd_on_counters(self.h, out))
        return tuple(int(v) for v in out)

    @property
    def max_U(self):
        v = C.c_double()
        _check(lib.qcd_on_max_u(self.h, C.byref(v)))
        return v.value

    def download(self):
        a = np.empty(self.sites, dtype=self.dtype)
        _check(lib.qcd_on_download(self.h, a.ctypes.data, a.nbytes))
        return a

    def upload(self, field):
        a = np.ascontiguousarray(field, dtype=self.dtype)
        assert a.size == self.sites
        _check(lib.qcd_on_upload(self.h, a.ctypes.data, a.nbytes))
