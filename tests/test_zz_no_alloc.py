"""The entry point that allocates nothing (qcd_measure_plaquette_sums_static) against the entry
point it is a variant of (qcd_measure_plaquette_sums): same kernels, partial sums in the library's
static array instead of allocated memory, so the two doubles must be bit-identical.  Two calls in a
row, the second on a smaller lattice (the array then still holds the first call's partial sums)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _thermalised(q, group, L, beta, sweeps):
    c = q.Context(group=group, L=L, beta=beta, nhit=10, rand_series=54)
    c.init_lattice()
    c.sweep(sweeps)
    return c


def test_static_plaquette_sums_equal_allocating_entry_point():
    import torch
    import qcdgpu_amd as q
    out = torch.full((2,), float("nan"), dtype=torch.float64, device="cuda:0")
    big = _thermalised(q, 3, (16, 16, 16, 16), 6.0, 2)
    small = _thermalised(q, 3, (8, 6, 4, 6), 5.5, 1)
    try:
        for ctx in (big, small):
            want = ctx.plaquette_sums()
            out.fill_(float("nan"))
            torch.cuda.synchronize()
            ctx.plaquette_sums_static(out.data_ptr())
            ctx.synchronize()
            got = out.cpu().numpy()
            assert np.array_equal(got, want), (got, want)
        # SU(2) through the same entry point, after the SU(3) calls
        with _thermalised(q, 2, (8, 8, 8, 8), 2.5, 1) as s2:
            want = s2.plaquette_sums()
            s2.plaquette_sums_static(out.data_ptr())
            s2.synchronize()
            assert np.array_equal(out.cpu().numpy(), want)
    finally:
        big.close()
        small.close()


def test_static_entry_point_rejects_what_it_cannot_hold():
    import qcdgpu_amd as q
    with q.Context(group=2, L=(4, 4, 4, 4), beta=2.5) as c:
        c.init_lattice()
        with pytest.raises(q.QcdError):
            c.plaquette_sums_static(0)          # null output
