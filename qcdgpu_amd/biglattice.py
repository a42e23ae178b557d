"""BigLattice: one lattice cut into x-slabs, one slab per rank, halo slices exchanged after
every update pass.

This is the host side of the reference's multi-device mode (SubLattice array + model::UpdateEdges,
SU(N)/suncl/suncl.cpp:4005-4061 and :5026-5138) re-designed for one process per GPU: the slab
geometry is the reference's (x is the slowest index, so a boundary slice is one contiguous run
per component), the transport is torch.distributed point-to-point (backend "nccl" = RCCL over
xGMI on the GPU box, "gloo" in the CPU tests) instead of map/memcpy/unmap through host memory.

What travels.  A (parity p, direction d) pass rewrites U_d on the sites of parity p only, so after
it exactly that half of a boundary slice is stale in the neighbour's halo:
  * last owned slice  -> RIGHT neighbour's left halo   (every d: the backward staples read
                         U_x, U_d and U_x(+d) one slice to the left);
  * first owned slice -> LEFT neighbour's right halo   (d = Y, Z, T only: nothing ever reads
                         U_x one slice to the right).
One message is NCOMP x (Ny/2 * Nz * Nt) fp64 complex numbers: 1.5 MiB for SU(3) at 32^3.

Random numbers.  RANLUX stream i belongs to the GLOBAL half-site index i, whichever rank owns it,
so a run gives the same configuration for every way of cutting the lattice.
"""
from typing import List, Optional, Sequence

import numpy as np

X, Y, Z, T = 0, 1, 2, 3


def split_even(nx: int, parts: int) -> List[int]:
    """nx slices over `parts` slabs, every slab an even number of slices (checkerboard parity
    is taken from the global x, but an even cut also keeps the two parities balanced)."""
    if nx % (2 * parts) == 0:
        return [nx // parts] * parts
    pairs, extra = divmod(nx // 2, parts)
    if nx % 2 or pairs == 0:
        raise ValueError("cannot cut Nx=%d into %d even slabs" % (nx, parts))
    return [2 * (pairs + (1 if k < extra else 0)) for k in range(parts)]


def messages_after_pass(direction: int):
    """(pack side, receiving side at the neighbour, neighbour offset) of every message that a
    pass over links of `direction` makes necessary."""
    msgs = [(1, 0, +1)]            # last owned slice -> right neighbour's left halo
    if direction != X:
        msgs.append((0, 1, -1))    # first owned slice -> left neighbour's right halo
    return msgs


class LocalComm:
    """All slabs in one process (several parts on one device: the way the reference's authors
    exercised BigLattice on a single GPU, SU(N)/QCDGPU.cpp:129-134)."""

    def __init__(self, engines: Sequence):
        self.engines = list(engines)

    def exchange(self, direction: int, parity: int):
        n = len(self.engines)
        for k, src in enumerate(self.engines):
            for pack_side, recv_side, off in messages_after_pass(direction):
                dst = self.engines[(k + off) % n]
                buf = src.halo_pack(pack_side, direction, parity)
                dst.halo_unpack(recv_side, direction, parity, buf)

    # split form (messages in flight while the interior is updated): nothing to overlap in one
    # process, so everything happens at the start; interior updates never read a halo slice
    def start_exchange(self, direction: int, parity: int):
        self.exchange(direction, parity)
        return None

    def finish_exchange(self, direction: int, parity: int, pending):
        pass


class TorchDistComm:
    """One slab per torch.distributed rank; neighbours k-1 and k+1 on a ring."""

    def __init__(self, engine, rank: int, world: int, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.engine = engine
        self.rank, self.world = rank, world
        self.group = group
        # gloo moves host memory only: device messages are staged through the host (rehearsals
        # of the GPU path without RCCL; the production backend is "nccl")
        self.stage_host = dist.get_backend(group) == "gloo"

    def exchange(self, direction: int, parity: int):
        self.finish_exchange(direction, parity, self.start_exchange(direction, parity))

    def start_exchange(self, direction: int, parity: int):
        """Pack the boundary slices this pass changed and post the sends and receives; returns
        what finish_exchange needs.  On RCCL the transfers run on the communicator's stream,
        ordered after the pack kernels, while the caller's stream goes on with the interior."""
        dist = self.dist
        ops, recvs = [], []
        for pack_side, recv_side, off in messages_after_pass(direction):
            peer_to = (self.rank + off) % self.world
            peer_from = (self.rank - off) % self.world
            out = self.engine.halo_pack(pack_side, direction, parity)
            inp = self.engine.halo_buffer_like(out)
            wire_out, wire_in = out, inp
            if self.stage_host and out.is_cuda:
                wire_out, wire_in = out.cpu(), inp.cpu()
            ops.append(dist.P2POp(dist.isend, wire_out, peer_to, group=self.group))
            ops.append(dist.P2POp(dist.irecv, wire_in, peer_from, group=self.group))
            recvs.append((recv_side, inp, wire_in))
        return dist.batch_isend_irecv(ops), recvs

    def finish_exchange(self, direction: int, parity: int, pending):
        works, recvs = pending
        for work in works:
            work.wait()
        for recv_side, inp, wire_in in recvs:
            if wire_in is not inp:
                inp.copy_(wire_in)
            self.engine.halo_unpack(recv_side, direction, parity, inp)

    def allreduce_sum(self, values: np.ndarray) -> np.ndarray:
        import torch
        dev = "cpu" if self.stage_host else self.engine.reduce_device()
        t = torch.tensor(np.asarray(values, dtype=np.float64), device=dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t.cpu().numpy()


class GpuSlabEngine:
    """A qcd_ctx slab whose halo messages live in torch CUDA tensors (plumbing only: torch owns
    the message buffers and the stream; all lattice work is the HIP library's)."""

    def __init__(self, ctx, device_index: int):
        import torch
        self.torch = torch
        self.ctx = ctx
        self.device = torch.device("cuda", device_index)
        self.nelem = ctx.halo_bytes() // 8
        # order the library's kernels with the RCCL calls torch issues
        ctx.set_stream(torch.cuda.current_stream(self.device).cuda_stream)
        self._pool = {}

    def _buf(self, key):
        if key not in self._pool:
            self._pool[key] = self.torch.empty(self.nelem, dtype=self.torch.float64, device=self.device)
        return self._pool[key]

    def update(self, parity, direction):
        self.ctx.update(parity, direction)

    def update_range(self, parity, direction, xs_begin, xs_count):
        self.ctx.update_range(parity, direction, xs_begin, xs_count)

    def update_split(self, parity, direction, first_begin, first_count):
        self.ctx.update_split(parity, direction, first_begin, first_count)

    @property
    def nx(self):
        return self.ctx.params.nx_local

    def overrelax(self, parity, direction):
        self.ctx.overrelax(parity, direction)

    def halo_pack(self, side, direction, parity):
        buf = self._buf(("out", side))
        self.ctx.halo_pack(side, direction, parity, buf.data_ptr())
        return buf

    def halo_buffer_like(self, out):
        return self._buf(("in", id(out)))

    def halo_unpack(self, side, direction, parity, buf):
        self.ctx.halo_unpack(side, direction, parity, buf.data_ptr())

    def plaquette_sums(self):
        return self.ctx.plaquette_sums()

    def polyakov_sums(self):
        return self.ctx.polyakov_sums()

    def reduce_device(self):
        return self.device


class BigLattice:
    """Sweep driver over slabs.  `engines` are this process's slabs, `comm` moves the halos."""

    def __init__(self, engines: Sequence, comm, group: int, L: Sequence[int], beta: float):
        self.engines = list(engines)
        self.comm = comm
        self.group = group
        self.L = tuple(L)
        self.beta = beta
        self.sites = int(np.prod(L))

    def fill_halos(self):
        """Every halo slice from its owner (needed once after a hot start: cold and gid starts
        fill their halos from closed forms)."""
        for d in (X, Y, Z, T):
            for parity in (0, 1):
                self.comm.exchange(d, parity)

    # A pass is split so that a slab's boundary slices are updated, packed and sent while its
    # interior is still being updated (sites of one pass do not read each other's links, so the
    # order inside a pass is free).  Engines with update_split run the two parts as concurrent
    # launches (the boundary part on the stream the halo calls follow, the rest beside it): one
    # slice per side is enough then.  Engines with update_range only run them one after the other,
    # and the first launch gets more slices (at most a quarter of the slab per side) so that it
    # keeps the device busy.  Measured on one MI355X, 32^4 slab, messages copied device to device:
    # 145 us per pass unsplit, 156 us split into consecutive launches 8 deep, 150 us as two
    # concurrent launches with one slice per side.
    BOUNDARY_DEPTH = 8
    BOUNDARY_DEPTH_CONCURRENT = 1

    def _concurrent(self) -> bool:
        return all(hasattr(e, "update_split") for e in self.engines)

    def boundary_depth(self) -> int:
        """Slices per side updated first, or 0 to update the slab in one launch."""
        if not all((hasattr(e, "update_range") or hasattr(e, "update_split")) and hasattr(e, "nx") for e in self.engines):
            return 0
        if not hasattr(self.comm, "start_exchange"):
            return 0
        nx = min(e.nx for e in self.engines)
        want = self.BOUNDARY_DEPTH_CONCURRENT if self._concurrent() else self.BOUNDARY_DEPTH
        depth = min(want, nx // 4)
        return depth if depth >= 1 else 0

    def update(self, parity: int, direction: int):
        depth = self.boundary_depth()
        if depth == 0:
            for e in self.engines:
                e.update(parity, direction)
            self.comm.exchange(direction, parity)
            return
        # the `depth` slices next to either face first (one range, wrapping around the slab), their
        # outermost slices go on the wire, the interior is updated meanwhile
        if self._concurrent():
            for e in self.engines:
                e.update_split(parity, direction, e.nx - depth, 2 * depth)
            pending = self.comm.start_exchange(direction, parity)
        else:
            for e in self.engines:
                e.update_range(parity, direction, e.nx - depth, 2 * depth)
            pending = self.comm.start_exchange(direction, parity)
            for e in self.engines:
                e.update_range(parity, direction, depth, e.nx - 2 * depth)
        self.comm.finish_exchange(direction, parity, pending)

    def overrelax_sweep(self, n: int = 1):
        """Microcanonical sweeps (no counterpart in the reference): same pass order and halo
        traffic as the heat-bath sweep."""
        for _ in range(n):
            for parity in (1, 0):
                for d in (X, Y, Z, T):
                    for e in self.engines:
                        e.overrelax(parity, d)
                    self.comm.exchange(d, parity)

    def sweep(self, n: int = 1):
        # reference order: odd X,Y,Z,T then even X,Y,Z,T (SU(N)/suncpp/suncpp.h:133-141)
        for _ in range(n):
            for d in (X, Y, Z, T):
                self.update(1, d)
            for d in (X, Y, Z, T):
                self.update(0, d)

    def _global_sum(self, local: np.ndarray) -> np.ndarray:
        if hasattr(self.comm, "allreduce_sum"):
            return self.comm.allreduce_sum(local)
        return local

    def plaquette(self) -> np.ndarray:
        s = np.sum([e.plaquette_sums() for e in self.engines], axis=0)
        s = self._global_sum(s)
        v = float(self.sites)
        # SU(N)/suncpp/Measurements/Plq.h:90-92
        return np.array([s[0] / (3 * v), s[1] / (3 * v), (s[0] + s[1]) / (6 * v)])

    def action(self) -> np.ndarray:
        # SU(N)/suncpp/Measurements/S.h:92-97
        return self.beta * (1 - self.plaquette() / self.group)

    def polyakov(self) -> np.ndarray:
        s = np.sum([e.polyakov_sums() for e in self.engines], axis=0)
        s = self._global_sum(s)
        n3 = float(self.L[0] * self.L[1] * self.L[2])
        g = float(self.group)
        # SU(N)/suncl/su3cpu.cpp:1427-1430
        return np.array([s[0] / (n3 * g), s[1] / (n3 * g), s[2] / (n3 * g * g), s[3] / (n3 * g ** 4)])


def make_gpu_slabs(group, L, beta, nx_parts: Sequence[int], device_index: int = 0, **ctx_kw):
    """Several slabs on ONE device, wired with LocalComm (single-GPU test of the halo logic)."""
    from . import capi
    engines, x0 = [], 0
    for nx in nx_parts:
        ctx = capi.Context(group=group, L=L, beta=beta, device=device_index, x0=x0, nx_local=nx,
                           halo=1, **ctx_kw)
        ctx.init_lattice()
        engines.append(LocalGpuEngine(ctx))
        x0 += nx
    lat = BigLattice(engines, LocalComm(engines), group, L, beta)
    if ctx_kw.get("start_mode") == capi.START_HOT:
        lat.fill_halos()
    return lat


class LocalGpuEngine:
    """Slab engine for LocalComm: message buffers are plain device allocations owned here."""

    def __init__(self, ctx):
        import torch
        self.ctx = ctx
        self.buf = torch.empty(ctx.halo_bytes() // 8, dtype=torch.float64,
                               device=torch.device("cuda", ctx.params.device))
        ctx.set_stream(torch.cuda.current_stream(self.buf.device).cuda_stream)

    def update(self, parity, direction):
        self.ctx.update(parity, direction)

    def update_range(self, parity, direction, xs_begin, xs_count):
        self.ctx.update_range(parity, direction, xs_begin, xs_count)

    def update_split(self, parity, direction, first_begin, first_count):
        self.ctx.update_split(parity, direction, first_begin, first_count)

    @property
    def nx(self):
        return self.ctx.params.nx_local

    def overrelax(self, parity, direction):
        self.ctx.overrelax(parity, direction)

    def halo_pack(self, side, direction, parity):
        self.ctx.halo_pack(side, direction, parity, self.buf.data_ptr())
        return self.buf

    def halo_unpack(self, side, direction, parity, buf):
        self.ctx.halo_unpack(side, direction, parity, buf.data_ptr())

    def plaquette_sums(self):
        return self.ctx.plaquette_sums()

    def polyakov_sums(self):
        return self.ctx.polyakov_sums()


def make_rank_slab(group, L, beta, rank: int, world: int, device_index: int, dist_group=None,
                   **ctx_kw) -> BigLattice:
    """This rank's slab of an L lattice cut evenly over `world` ranks (one process per GPU)."""
    from . import capi
    parts = split_even(L[0], world)
    x0 = sum(parts[:rank])
    ctx = capi.Context(group=group, L=L, beta=beta, device=device_index, x0=x0,
                       nx_local=parts[rank], halo=1, **ctx_kw)
    ctx.init_lattice()
    eng = GpuSlabEngine(ctx, device_index)
    lat = BigLattice([eng], TorchDistComm(eng, rank, world, dist_group), group, L, beta)
    if ctx_kw.get("start_mode") == capi.START_HOT:
        lat.fill_halos()
    return lat


class RingLattice:
    """One slab per process, passes and halo exchange issued below the C-ABI (qcd_ring over RCCL);
    torch.distributed (any backend, gloo will do) only hands the RCCL identifier round and adds up
    the few doubles of a measurement.  The host loop of an N-GPU run is then the C library's, as the
    reference's is its C++ model's (SU(N)/suncl/suncl.cpp:5026-5138)."""

    def __init__(self, ctx, rank: int, world: int, group: int, L: Sequence[int], beta: float, dist_group=None):
        import torch
        import torch.distributed as dist
        from . import capi
        self.ctx, self.rank, self.world = ctx, rank, world
        self.group, self.L, self.beta = group, tuple(L), beta
        self.sites = int(np.prod(L))
        self.dist, self.dist_group, self.torch = dist, dist_group, torch
        self.on_host = dist.get_backend(dist_group) == "gloo"
        dev = "cpu" if self.on_host else torch.device("cuda", ctx.params.device)
        ident = torch.zeros(128, dtype=torch.uint8, device=dev)
        if rank == 0:
            ident = torch.frombuffer(bytearray(capi.Ring.unique_id()), dtype=torch.uint8).to(dev)
        dist.broadcast(ident, 0, group=dist_group)
        self.ring = capi.Ring([ctx], rank=rank, world=world, unique_id=bytes(ident.cpu().numpy().tobytes()))
        self.engines = [self]     # the shape BigLattice has, for callers that look at engines[0].ctx

    def sweep(self, n: int = 1):
        self.ring.sweep(n)

    def update(self, parity: int, direction: int):
        self.ring.update(parity, direction)

    def _global_sum(self, local: np.ndarray) -> np.ndarray:
        dev = "cpu" if self.on_host else self.torch.device("cuda", self.ctx.params.device)
        t = self.torch.tensor(np.asarray(local, dtype=np.float64), device=dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.dist_group)
        return t.cpu().numpy()

    def plaquette(self) -> np.ndarray:
        s = self._global_sum(self.ctx.plaquette_sums())
        v = float(self.sites)
        return np.array([s[0] / (3 * v), s[1] / (3 * v), (s[0] + s[1]) / (6 * v)])   # SU(N)/suncpp/Measurements/Plq.h:90-92

    def polyakov(self) -> np.ndarray:
        s = self._global_sum(self.ctx.polyakov_sums())
        n3 = float(self.L[0] * self.L[1] * self.L[2])
        g = float(self.group)
        return np.array([s[0] / (n3 * g), s[1] / (n3 * g), s[2] / (n3 * g * g), s[3] / (n3 * g ** 4)])

    def close(self):
        self.ring.close()


def make_rank_ring(group, L, beta, rank: int, world: int, device_index: int, dist_group=None, **ctx_kw) -> RingLattice:
    """This rank's slab of an L lattice cut evenly over `world` ranks, driven through qcd_ring (RCCL)."""
    from . import capi
    parts = split_even(L[0], world)
    ctx = capi.Context(group=group, L=L, beta=beta, device=device_index, x0=sum(parts[:rank]),
                       nx_local=parts[rank], halo=1, **ctx_kw)
    ctx.init_lattice()
    lat = RingLattice(ctx, rank, world, group, L, beta, dist_group)
    if ctx_kw.get("start_mode") == capi.START_HOT:
        lat.ring.fill_halos()   # a hot start writes the owned slices only (SU(N)/suncl/suncl.cpp:4835-4903)
    return lat
