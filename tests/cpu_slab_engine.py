"""A slab engine on top of the CPU checker, with the interface qcdgpu_amd.biglattice expects.

Each rank keeps a full-size link array but owns only the x-slices [x0, x0 + nx); everything
outside the owned slices and the two halo slices is poisoned with NaN, so a message that is
missing, late or sent to the wrong side shows up as NaN in the plaquette.  Test infrastructure.
"""
import ctypes as C

import numpy as np
import torch

from tests import oracle_lib as ol

ol.lib.qo_update_pass_range.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
ol.lib.qo_plaquette_sums_range.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double)]
ol.lib.qo_polyakov_sums_range.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double)]


class CpuSlabEngine:
    def __init__(self, group, L, beta, x0, nx, nhit=10, lux=3, seed=54, ints=1, share=1):
        mode = ol.RNG_GID if ints == 2 else ol.RNG_PERSITE
        self.lat = ol.Lattice(group, L, beta, nhit=nhit, ints=ints, rng_mode=mode, lux=lux, seed=seed, share=share)
        self.L, self.x0, self.nx = tuple(L), x0, nx
        self.slice = L[1] * L[2] * L[3]
        reals = 18 if group == 3 else 8
        head = C.cast(self.lat.h, C.POINTER(ol._LatticeHead)).contents
        buf = (C.c_double * (self.lat.sites * 4 * reals)).from_address(head.links)
        # [x][site in slice][dir][reals]
        self.links = np.frombuffer(buf, dtype=np.float64).reshape(L[0], self.slice, 4, reals)
        owned = np.zeros(L[0], dtype=bool)
        # hot start: the halos too are unknown until BigLattice.fill_halos() has run
        pad = 0 if ints == 0 else 1
        for x in range(x0 - pad, x0 + nx + pad):
            owned[x % L[0]] = True
        self.links[~owned] = np.nan
        c = np.arange(self.slice)
        y, z, t = c % L[1], (c // L[1]) % L[2], c // (L[1] * L[2])
        self.yzt_parity = (y + z + t) & 1

    def update(self, parity, direction):
        ol.lib.qo_update_pass_range(self.lat.h, parity, direction, self.x0, self.x0 + self.nx)

    def update_range(self, parity, direction, xs_begin, xs_count):
        # slices xs_begin .. xs_begin + xs_count - 1 of this slab, wrapping around its owned range
        first = min(xs_count, self.nx - xs_begin)
        ol.lib.qo_update_pass_range(self.lat.h, parity, direction, self.x0 + xs_begin, self.x0 + xs_begin + first)
        if xs_count > first:
            ol.lib.qo_update_pass_range(self.lat.h, parity, direction, self.x0, self.x0 + xs_count - first)

    def update_split(self, parity, direction, first_begin, first_count):
        # the device engine runs the two parts as concurrent launches; the order does not matter
        self.update_range(parity, direction, first_begin, first_count)
        if first_count < self.nx:
            self.update_range(parity, direction, (first_begin + first_count) % self.nx, self.nx - first_count)

    def _sel(self, x, parity):
        return ((self.yzt_parity + x) & 1) == parity

    def halo_pack(self, side, direction, parity):
        x = self.x0 if side == 0 else self.x0 + self.nx - 1
        return torch.from_numpy(self.links[x, self._sel(x, parity), direction].copy())

    def halo_buffer_like(self, out):
        return torch.empty_like(out)

    def halo_unpack(self, side, direction, parity, buf):
        x = (self.x0 - 1) % self.L[0] if side == 0 else (self.x0 + self.nx) % self.L[0]
        self.links[x, self._sel(x, parity), direction] = buf.numpy()

    def plaquette_sums(self):
        out = (C.c_double * 2)()
        ol.lib.qo_plaquette_sums_range(self.lat.h, self.x0, self.x0 + self.nx, out)
        return np.array(out[:])

    def polyakov_sums(self):
        out = (C.c_double * 4)()
        ol.lib.qo_polyakov_sums_range(self.lat.h, self.x0, self.x0 + self.nx, out)
        return np.array(out[:])

    def reduce_device(self):
        return "cpu"
