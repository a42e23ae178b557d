"""GPU parity: the HIP path (through the C-ABI) against the CPU checker and the golden vectors.

Tolerances (fp64): the device fuses multiply-adds and uses its own libm, so single links agree
with the checker to rounding (1e-13 after one pass); lattice averages are required to agree to
1e-12 relative -- the bar the north star states for the plaquette -- over the few sweeps for
which a chaotic Markov chain can be compared at all.
"""
import math
import os

import numpy as np
import pytest

from tests import oracle_lib as ol

pytestmark = pytest.mark.gpu

REL = 1e-12


def _ctx(**kw):
    import qcdgpu_amd as q
    return q.Context(**kw)


def _rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300))


@pytest.mark.parametrize("group", [2, 3])
def test_cold_start_known_answers(group):
    with _ctx(group=group, L=(4, 6, 4, 8), beta=6.0) as c:
        c.init_lattice()
        assert np.all(c.plaquette() == float(group))
        assert np.all(c.action() == 0.0)
        p = c.polyakov()
        assert p[0] == 1.0 and p[1] == 0.0 and p[2] == 1.0 and p[3] == 1.0


@pytest.mark.parametrize("group", [2, 3])
def test_gid_start_links_bit_exact_and_layout_roundtrip(group):
    import qcdgpu_amd as q
    L = (4, 6, 8, 4)
    with _ctx(group=group, L=L, beta=6.0, start_mode=q.START_GID) as c:
        c.init_lattice()
        lat = ol.Lattice(group, L, 6.0, ints=2, rng_mode=ol.RNG_GID)
        want = lat.export_table()
        got = c.download()
        # closed-form start evaluated with the host libm on both sides; compilers differ in
        # whether sin(a), cos(a) of one argument become one sincos() call, which moves a few
        # entries by one ulp
        assert np.max(np.abs(got - want)) < 4e-16
        assert _rel(c.plaquette(), lat.plaquette()) < 1e-13
        assert np.allclose(c.polyakov(), lat.polyakov(), rtol=1e-12, atol=1e-15)
        # upload of a padded table (ROWSIZE > sites) and download again
        rowsize = lat.sites + 192
        t = lat.export_table(rowsize)
    with _ctx(group=group, L=L, beta=6.0) as c2:
        c2.upload(t)
        assert np.array_equal(c2.download(rowsize)[:, :lat.sites], want)   # pure data movement
    lat.close()


@pytest.mark.parametrize("group,L", [(2, (4, 4, 6, 4)), (3, (4, 4, 6, 4)),
                                     # odd extents: a pass runs in phases that reproduce the sequential order across the wrap
                                     (2, (5, 4, 3, 7)), (3, (5, 4, 3, 7)), (3, (4, 2, 4, 3)), (3, (3, 2, 3, 3))])
def test_every_pass_matches_checker_link_by_link(group, L):
    """gid mode, one (parity, direction) pass at a time: every link against the checker."""
    import qcdgpu_amd as q
    beta = 6.0 if group == 3 else 2.4
    lat = ol.Lattice(group, L, beta, nhit=10, ints=2, rng_mode=ol.RNG_GID)
    with _ctx(group=group, L=L, beta=beta, nhit=10, start_mode=q.START_GID) as c:
        c.init_lattice()
        for parity in (1, 0):
            for d in range(4):
                c.update(parity, d)
                lat.update(parity, d)
                got, want = c.download(), lat.export_table()
                assert np.max(np.abs(got - want)) < 1e-12, (parity, d)
    lat.close()


GID_CASES = [c for c in ol.golden_cases() if c.endswith("_gid.json")]


@pytest.mark.parametrize("case", GID_CASES)
def test_gid_runs_match_reference_vectors(case):
    """Deterministic (INTS = 2) runs against the vectors minted from the compiled reference."""
    import qcdgpu_amd as q
    fx = ol.golden(case)
    a = ol.init_to_args(fx["init"])
    with _ctx(group=a["group"], L=a["L"], beta=a["beta"], nhit=a["nhit"], start_mode=q.START_GID,
              lux_level=a["lux"], rand_series=a["seed"]) as c:
        c.init_lattice()
        rows = [np.concatenate([c.plaquette(), c.action()])]
        c.sweep(a["nav"])
        for _ in range(1, a["iter_"]):
            c.sweep(a["niter"])
            rows.append(np.concatenate([c.plaquette(), c.action()]))
    for got, want in zip(rows, fx["samples"]):
        for g, w in zip(got, want):
            if w is None or (isinstance(w, float) and math.isnan(w)):
                assert math.isnan(g)
            else:
                # fixtures carry 14 significant digits
                assert abs(g - w) <= 2e-13 * abs(w) + 1e-13, (case, g, w)


@pytest.mark.parametrize("group,L,beta,nhit,lux,seed", [
    (3, (8, 8, 8, 8), 6.0, 10, 3, 54),
    (3, (4, 6, 8, 10), 5.5, 3, 0, 1234),
    (3, (6, 4, 4, 6), 1.5, 2, 4, 99),      # low beta: many rejected trials, batch refills
    (2, (8, 8, 8, 8), 2.5, 10, 3, 54),
    (2, (6, 8, 4, 10), 1.2, 2, 2, 7),
    (3, (4, 4, 4, 6), 1.0, 12, 3, 5),      # more than 48 numbers per link for many links: the redo pass
    (2, (4, 4, 6, 4), 0.5, 14, 1, 8),      # the same for SU(2)
    (3, (5, 4, 4, 7), 5.5, 10, 3, 54),     # odd Nx and Nt
    (2, (4, 6, 3, 4), 2.3, 5, 2, 11),      # odd Nz
    (3, (3, 4, 5, 3), 3.0, 3, 3, 3),       # all three odd, rejected trials
])
def test_ranlux_runs_match_checker(group, L, beta, nhit, lux, seed):
    """Production RNG mode: one RANLUX stream per thread.  Same streams on both sides, so the
    accept/reject history (count of numbers drawn) is identical and the averages agree."""
    lat = ol.Lattice(group, L, beta, nhit=nhit, ints=1, rng_mode=ol.RNG_PERSITE, lux=lux, seed=seed)
    with _ctx(group=group, L=L, beta=beta, nhit=nhit, lux_level=lux, rand_series=seed) as c:
        c.init_lattice()
        for sweep in range(3):
            c.sweep(1)
            lat.sweep(1)
            assert c.rng_draws() == ol.rng_draws(lat), sweep
            assert _rel(c.plaquette(), lat.plaquette()) < REL, sweep
            assert _rel(c.action(), lat.action()) < 1e-10   # S = beta(1 - Plq/N): cancellation
        assert np.allclose(c.polyakov(), lat.polyakov(), rtol=1e-10, atol=1e-13)
        got, want = c.download(), lat.export_table()
        # rounding differences between the two libms grow from sweep to sweep, fastest at small beta
        assert np.max(np.abs(got - want)) < (1e-10 if beta > 1.0 else 1e-9)
    lat.close()


@pytest.mark.parametrize("group,L,beta,nhit,lux,seed,share", [
    (3, (8, 8, 8, 8), 6.0, 10, 3, 54, 2),
    (3, (4, 6, 8, 12), 5.5, 3, 0, 1234, 4),
    (3, (4, 4, 4, 8), 1.0, 12, 3, 5, 4),     # low beta: most links run past the window, the state goes on from the slow form
    (3, (6, 4, 4, 6), 1.5, 2, 4, 99, 2),     # many second batches
    (2, (8, 8, 8, 8), 2.5, 10, 3, 54, 4),
    (2, (4, 4, 6, 4), 0.5, 14, 1, 8, 2),
    (3, (5, 4, 3, 6), 5.5, 10, 3, 54, 2),    # odd Nx and Nz: the general kernels, passes in phases
    (2, (6, 2, 2, 4), 2.3, 5, 2, 11, 4),     # a launch of partly filled blocks
])
def test_shared_streams_match_checker(group, L, beta, nhit, lux, seed, share):
    """rng_share sites of one parity on one RANLUX stream (the sites Nt / rng_share apart in t, updated one after the
    other by one lane, the generator state passed on in registers or LDS): the checker's per-thread-stream mode
    with the same map must see the same draw counts and plaquettes; the state left in memory must be the one a
    fresh context continues from."""
    import qcdgpu_amd as q
    lat = ol.Lattice(group, L, beta, nhit=nhit, ints=1, rng_mode=ol.RNG_PERSITE, lux=lux, seed=seed, share=share)
    with _ctx(group=group, L=L, beta=beta, nhit=nhit, lux_level=lux, rand_series=seed, rng_share=share) as c:
        assert c.rng_share == share
        c.init_lattice()
        for sweep in range(3):
            c.sweep(1)
            lat.sweep(1)
            assert c.rng_draws() == ol.rng_draws(lat), sweep
            # (rounding differences between the two libms grow fastest in the disordered phase)
            assert _rel(c.plaquette(), lat.plaquette()) < (REL if beta > 2.0 else 1e-11), sweep
        got, want = c.download(), lat.export_table()
        assert np.max(np.abs(got - want)) < (1e-10 if beta > 1.0 else 1e-9)
        # state round trip: a second context continues bit for bit
        with q.Context(group=group, L=L, beta=beta, nhit=nhit, lux_level=lux, rand_series=seed + 1, rng_share=share) as b:
            b.upload(got)
            b.set_rng_state(c.rng_state())
            c.sweep(1)
            b.sweep(1)
            assert np.array_equal(c.download(), b.download())
    lat.close()
    with pytest.raises(q.QcdError):
        _ctx(group=group, L=(4, 4, 4, 6), beta=beta, rng_share=4)     # must divide Nt
    with pytest.raises(q.QcdError):
        _ctx(group=group, L=L, beta=beta, rng_share=3)


@pytest.mark.parametrize("share", [2, 4])
def test_shared_streams_in_the_pair_kernel_and_in_slabs(share, monkeypatch):
    """The same through the gather / heat-bath pair kernel (forced for a 16^4 lattice) and for the lattice cut into
    slabs: the stream map is a function of global coordinates."""
    import qcdgpu_amd as q
    L = (16, 16, 16, 16)
    lat = ol.Lattice(3, L, 5.7, nhit=10, ints=1, rng_mode=ol.RNG_PERSITE, share=share)
    monkeypatch.setenv("QCDGPU_AMD_GH", "2")
    with _ctx(group=3, L=L, beta=5.7, nhit=10, rng_share=share) as c:
        monkeypatch.delenv("QCDGPU_AMD_GH")
        c.init_lattice()
        parts, x0 = [], 0
        for nx in (6, 10):
            parts.append(q.Context(group=3, L=L, beta=5.7, nhit=10, rng_share=share, x0=x0, nx_local=nx, halo=1))
            parts[-1].init_lattice()
            x0 += nx
        ring = q.Ring(parts, q.RING_COPY)
        try:
            for sweep in range(2):
                c.sweep(1)
                lat.sweep(1)
                ring.sweep(1)
                assert c.rng_draws() == ol.rng_draws(lat) == sum(p.rng_draws() for p in parts), sweep
                assert _rel(c.plaquette(), lat.plaquette()) < REL, sweep
                sums = np.sum([p.plaquette_sums() for p in parts], axis=0)
                v = float(np.prod(L))
                got = np.array([sums[0] / (3 * v), sums[1] / (3 * v), (sums[0] + sums[1]) / (6 * v)])
                assert _rel(got, lat.plaquette()) < REL, sweep
        finally:
            ring.close()
            for p in parts:
                p.close()
    lat.close()


@pytest.mark.parametrize("group,L,beta,share", [(3, (8, 6, 4, 8), 5.8, 1), (2, (6, 8, 6, 4), 2.4, 1), (3, (4, 6, 4, 8), 5.8, 4), (2, (6, 4, 6, 4), 2.4, 2)])
def test_hot_start_matches_checker(group, L, beta, share):
    """Hot start (the reference has it on the device path only: parity unpinned).  The checker
    restates lattice_random3 / lattice_random2 with the same per-thread streams; sin/cos come
    from different libms, so links agree to rounding, not bit for bit."""
    import qcdgpu_amd as q
    lat = ol.Lattice(group, L, beta, nhit=10, ints=0, rng_mode=ol.RNG_PERSITE, seed=11, share=share)
    with _ctx(group=group, L=L, beta=beta, nhit=10, start_mode=q.START_HOT, rand_series=11, rng_share=share) as c:
        c.init_lattice()
        assert c.rng_draws() == ol.rng_draws(lat)
        got, want = c.download(), lat.export_table()
        assert np.max(np.abs(got - want)) < 1e-14
        assert abs(c.plaquette()[2]) < 0.1                 # random links: plaquette ~ 0
        assert np.max(np.abs(c.plaquette() - lat.plaquette())) < 1e-13
        for sweep in range(2):                             # streams resume in front of a fresh batch
            c.sweep(1)
            lat.sweep(1)
            assert c.rng_draws() == ol.rng_draws(lat), sweep
            assert _rel(c.plaquette(), lat.plaquette()) < 1e-11, sweep
    lat.close()


@pytest.mark.parametrize("lux", [0, 1, 2, 3, 4])
def test_device_generator_reproduces_cpu_generator(lux):
    """The reference's own PRNG known-answer test (PRNG::check): device stream against the CPU
    generator, value by value -- here for many streams and against the checker as well."""
    import ctypes as C
    import qcdgpu_amd as q
    assert q.prng_check(device=0, rand_series=54, lux_level=lux, streams=300, samples=250) == 0
    # and the host generator the check compares with is the checker's (stream 0 and a later one)
    for stream in (0, 17):
        st = ol.RanluxState()
        ol.lib.qo_ranlux_init(C.byref(st), ol.lib.qo_stream_jseed(54, stream), ol.lib.qo_nskip_for_level(lux))
        want = np.array([ol.lib.qo_ranlux_next(C.byref(st)) for _ in range(100)], dtype=np.float32)
        assert np.array_equal(q.host_ranlux_stream(54, stream, lux, 100), want)


def test_heatbath_math_is_accurate_to_the_last_places():
    """The update kernel's own log / sqrt / reciprocal (hardware seed + refinement, table log)
    against the host libm in long double, on the argument forms the heat-bath feeds them."""
    import qcdgpu_amd as q
    log_ulp, sqrt_ulp, rcp_ulp, rsqrt_ulp = q.math_check(device=0, n=1 << 20)
    assert log_ulp <= 2.0 and sqrt_ulp <= 1.0 and rcp_ulp <= 1.5 and rsqrt_ulp <= 2.5, (log_ulp, sqrt_ulp, rcp_ulp, rsqrt_ulp)


def test_accept_step_cosine_and_sine_are_accurate():
    """cos(2 pi r), sin(2 pi r) of the fp64 accept step (table of 256 angles + Taylor series) against the host libm in
    long double, on RANLUX-shaped arguments including those below 2^-12 with their correction bits: within 1.5e-16 of
    the exact value (the reference's own cos(PI2 * r) is 6e-16 from it: the product is rounded first)."""
    import qcdgpu_amd as q
    err_cos, err_sin = q.trig_check(device=0, n=1 << 21)
    assert err_cos <= 1.5e-16 and err_sin <= 1.5e-16, (err_cos, err_sin)


@pytest.mark.parametrize("group,L", [(3, (2, 2, 2, 2)), (3, (2, 6, 2, 2)), (3, (6, 10, 2, 4)), (3, (4, 34, 4, 4)),
                                     (2, (2, 2, 2, 2)), (2, (10, 6, 6, 2)), (3, (3, 2, 3, 3)), (2, (3, 2, 5, 3)), (3, (7, 18, 4, 5))])
def test_small_and_ragged_lattices(group, L):
    """Smallest lattice, extents that leave workgroups partly empty and that no (z,t) region
    cut divides, Ny/2 that does not divide the workgroup size: link by link against the checker."""
    lat = ol.Lattice(group, L, 5.0 if group == 3 else 2.0, nhit=3, ints=1, rng_mode=ol.RNG_PERSITE, seed=13)
    with _ctx(group=group, L=L, beta=5.0 if group == 3 else 2.0, nhit=3, rand_series=13) as c:
        c.init_lattice()
        for sweep in range(2):
            c.sweep(1)
            lat.sweep(1)
            assert c.rng_draws() == ol.rng_draws(lat), sweep
            assert np.max(np.abs(c.download() - lat.export_table())) < 1e-11, sweep
        assert np.allclose(c.polyakov(), lat.polyakov(), rtol=1e-10, atol=1e-13)
    lat.close()


@pytest.mark.parametrize("group,L,beta", [(3, (8, 8, 4, 6), 6.0), (2, (6, 8, 6, 4), 2.4), (3, (5, 4, 3, 4), 6.0), (2, (4, 4, 5, 7), 2.4)])
def test_overrelaxation_matches_checker_and_keeps_the_action(group, L, beta):
    """Heat-bath + overrelaxation (the benchmark's 16^4 configuration names both; the reference has only the
    heat-bath, so overrelaxation is parity unpinned): link by link against the checker's
    restatement, and the plaquette average unchanged by microcanonical sweeps."""
    lat = ol.Lattice(group, L, beta, nhit=10, ints=1, rng_mode=ol.RNG_PERSITE, seed=19)
    with _ctx(group=group, L=L, beta=beta, nhit=10, rand_series=19) as c:
        c.init_lattice()
        for sweep in range(2):
            c.sweep(1)
            lat.sweep(1)
            p0 = c.plaquette()
            c.overrelax_sweep(2)
            lat.overrelax_sweep(2)
            assert abs(c.plaquette()[2] - p0[2]) < 1e-13, sweep     # the total action
            assert np.max(np.abs(c.download() - lat.export_table())) < 1e-10, sweep
            assert _rel(c.plaquette(), lat.plaquette()) < REL
        assert c.rng_draws() == ol.rng_draws(lat)       # no random numbers on that path
    lat.close()


@pytest.mark.parametrize("share", [1, 4])
def test_overrelaxation_through_the_pair_kernel_matches_checker(share, monkeypatch):
    """The overrelaxation pass of large SU(3) lattices runs through the gather / second-wavefront pairs of the heat-bath
    pass; forced here for a 16^4 lattice and compared link by link with the checker, which knows nothing of kernels."""
    L = (16, 16, 16, 16)
    lat = ol.Lattice(3, L, 5.9, nhit=10, ints=1, rng_mode=ol.RNG_PERSITE, seed=31, share=share)
    monkeypatch.setenv("QCDGPU_AMD_GH", "2")
    with _ctx(group=3, L=L, beta=5.9, nhit=10, rand_series=31, rng_share=share) as c:
        monkeypatch.delenv("QCDGPU_AMD_GH")
        assert c.pass_uses_pair_kernel()
        c.init_lattice()
        c.sweep(1)
        lat.sweep(1)
        p0 = c.plaquette()
        c.overrelax_sweep(2)
        lat.overrelax_sweep(2)
        assert abs(c.plaquette()[2] - p0[2]) < 1e-13
        assert np.max(np.abs(c.download() - lat.export_table())) < 1e-10
        c.sweep(1)                                         # the generator was left alone
        lat.sweep(1)
        assert c.rng_draws() == ol.rng_draws(lat)
        assert _rel(c.plaquette(), lat.plaquette()) < REL
    lat.close()


def test_overrelaxation_32x4_both_kernels_agree(monkeypatch):
    """At the benchmark's 32^4 the pass runs as the pair kernel by itself: same links as the one-wavefront-per-block
    kernel (chosen by switching the pair kernel off for a second context), action unchanged, every link moved."""
    L = (32, 32, 32, 32)
    with _ctx(group=3, L=L, beta=6.0, nhit=10) as a:
        assert a.pass_uses_pair_kernel()
        a.init_lattice()
        a.sweep(2)
        start = a.download()
        monkeypatch.setenv("QCDGPU_AMD_GH", "0")
        with _ctx(group=3, L=L, beta=6.0, nhit=10) as b:
            monkeypatch.delenv("QCDGPU_AMD_GH")
            assert not b.pass_uses_pair_kernel()
            b.upload(start)
            p0 = a.plaquette()
            a.overrelax_sweep(1)
            b.overrelax_sweep(1)
            got = a.download()
            assert np.max(np.abs(got - b.download())) < 1e-13
            assert abs(a.plaquette()[2] - p0[2]) < 1e-13
            assert np.mean(np.abs(got - start)) > 1e-2


@pytest.mark.parametrize("group,L,beta", [(3, (6, 4, 8, 6), 6.0), (2, (4, 8, 6, 6), 2.4), (3, (5, 4, 6, 7), 6.0)])
def test_wilson_loop_matches_checker(group, L, beta):
    import qcdgpu_amd as q
    from qcdgpu_amd import biglattice as bl
    lat = ol.Lattice(group, L, beta, nhit=5, ints=1, rng_mode=ol.RNG_PERSITE, seed=23)
    with _ctx(group=group, L=L, beta=beta, nhit=5, rand_series=23) as c:
        c.init_lattice()
        assert c.wilson_loop(3, 2) == float(group)
        c.sweep(2)
        lat.sweep(2)
        assert abs(c.wilson_loop(1, 1) - c.plaquette()[1]) < 1e-13
        for R, T in ((1, 1), (2, 1), (1, 3), (2, 2), (3, 4), (5, 7)):     # the last ones wrap around
            assert abs(c.wilson_loop(R, T) - lat.wilson_loop(R, T)) < 1e-12, (R, T)
        with pytest.raises(q.QcdError):
            c.wilson_loop(0, 1)
    lat.close()
    if any(n & 1 for n in L):
        with pytest.raises(q.QcdError, match="odd"):
            bl.make_gpu_slabs(group, L, beta, [2, L[0] - 2], nhit=5, rand_series=23)   # odd extents: one uncut lattice only
        return
    big = bl.make_gpu_slabs(group, L, beta, [2, L[0] - 2], nhit=5, rand_series=23)
    with pytest.raises(q.QcdError):
        big.engines[0].ctx.wilson_loop(1, 1)                               # not on a slab


@pytest.mark.parametrize("L,parts", [((16, 32, 32, 20), [8, 8]), ((20, 20, 20, 20), [16, 4])])
def test_fmunu_marching_kernel_matches_checker(L, parts):
    """F_mu_nu sums on lattices large enough for the kernel that marches along x (eight slices per thread on the first lattice,
    four on the second; the reference's fixtures are all on small lattices, which take the one-site-per-thread kernel):
    both fields against the checker on the same configuration after one sweep, and the same lattice cut into two
    slabs, whose sums add up to the uncut lattice's."""
    from qcdgpu_amd import biglattice as bl
    lat = ol.Lattice(3, L, 5.7, nhit=3, ints=1, rng_mode=ol.RNG_PERSITE, seed=17)
    lat.sweep(1)
    want = {(f, a, b): lat.fmunu_sums(f, a, b) for f, a, b in ((0, 3, 8), (1, 3, 8), (0, 1, 6), (1, 2, 5), (1, 4, 7))}
    plq = lat.plaquette()
    lat.close()
    with _ctx(group=3, L=L, beta=5.7, nhit=3, rand_series=17) as c:
        c.init_lattice()
        c.sweep(1)
        assert _rel(c.plaquette(), plq) < REL
        for key, w in want.items():
            assert np.allclose(c.fmunu(*key, sums=True), w, rtol=1e-9, atol=1e-10), key
    big = bl.make_gpu_slabs(3, L, 5.7, parts, nhit=3, rand_series=17)
    try:
        big.sweep(1)
        for key, w in want.items():
            got = np.sum([e.ctx.fmunu(*key, sums=True) for e in big.engines], axis=0)
            assert np.allclose(got, w, rtol=1e-9, atol=1e-10), key
    finally:
        for e in big.engines:
            e.ctx.close()


def test_config1_su3_16_plaquette_vs_cpu():
    """The benchmark's SU(3) 16^4 configuration: fp64 on one GPU, plaquette against the CPU path to 1e-12."""
    L = (16, 16, 16, 16)
    lat = ol.Lattice(3, L, 6.0, nhit=10, ints=1, rng_mode=ol.RNG_PERSITE)
    with _ctx(group=3, L=L, beta=6.0, nhit=10) as c:
        c.init_lattice()
        for sweep in range(2):
            c.sweep(1)
            lat.sweep(1)
            assert c.rng_draws() == ol.rng_draws(lat)
            assert _rel(c.plaquette(), lat.plaquette()) < REL
    lat.close()


def test_reunitarize_restores_unitarity():
    import qcdgpu_amd as q
    L = (4, 4, 4, 4)
    lat = ol.Lattice(3, L, 6.0, ints=2, rng_mode=ol.RNG_GID)
    t = lat.export_table()
    rng = np.random.default_rng(1)
    noisy = t + 1e-3 * rng.standard_normal(t.shape)
    with _ctx(group=3, L=L, beta=6.0) as c:
        c.upload(noisy)
        c.reunitarize()
        lat.import_table(c.download())
        # after Gram-Schmidt every plaquette is a product of unitary matrices: |ReTr| <= 3
        p = lat.plaquette()
        assert np.all(np.abs(p) <= 3.0)
        c.reunitarize()
        again = c.download()
        assert np.max(np.abs(again - lat.export_table())) < 1e-14   # idempotent
    lat.close()


def test_errors_are_status_codes_not_exits():
    import qcdgpu_amd as q
    with pytest.raises(q.QcdError):
        _ctx(group=4, L=(4, 4, 4, 4), beta=1.0)
    with pytest.raises(q.QcdError):
        _ctx(group=3, L=(4, 5, 4, 4), beta=1.0)          # odd Ny
    with pytest.raises(q.QcdError):
        _ctx(group=3, L=(4, 4, 4, 4), beta=1.0, start_mode=7)
    with _ctx(group=3, L=(4, 4, 4, 4), beta=1.0) as c:
        with pytest.raises(q.QcdError):
            c.sweep(1)                                    # before init
        c.init_lattice()
        with pytest.raises(q.QcdError):
            c.update(2, 0)
        with pytest.raises(q.QcdError):
            c.upload(np.zeros((12, 100, 4)))              # rowsize < sites


@pytest.mark.parametrize("group,L", [(3, (6, 4, 8, 4)), (2, (4, 8, 6, 6)), (3, (5, 4, 3, 6))])
def test_plane_profiles_match_checker(group, L):
    """Plane-resolved plaquette and Polyakov-loop sums (the reference's lattice_action_diff_* and
    lattice_polyakov_diff_* kernels; device path only there, so no reference output to pin them):
    against the checker plane by plane to 1e-12, and their totals against the lattice sums."""
    import qcdgpu_amd as q
    beta = 5.9 if group == 3 else 2.4
    lat = ol.Lattice(group, L, beta, nhit=6, ints=1, rng_mode=ol.RNG_PERSITE)
    with q.Context(group=group, L=L, beta=beta, nhit=6) as c:
        c.init_lattice()
        c.sweep(2)
        lat.sweep(2)
        sums = c.plaquette_sums()
        psums = c.polyakov_sums()
        for axis in range(3):
            got, want = c.plaquette_profile_sums(axis), lat.plaquette_profile_sums(axis)
            assert got.shape == (L[axis], 2)
            assert np.max(np.abs(got - want) / np.abs(want)) < 1e-12
            assert np.allclose(got.sum(axis=0), sums, rtol=1e-12)
            gp, wp = c.polyakov_profile_sums(axis), lat.polyakov_profile_sums(axis)
            assert np.allclose(gp, wp, rtol=1e-10, atol=1e-12)
            assert np.allclose(gp.sum(axis=0), psums[:2], rtol=1e-11, atol=1e-12)
        with pytest.raises(q.QcdError):
            c.plaquette_profile_sums(3)
        # the profiles reuse per-site values of the last global measurement: a sweep in between
        # must make them stale (profiles first, then once more after the global measurements)
        c.sweep(1)
        lat.sweep(1)
        for again in range(2):
            for axis in (1, 0, 2):
                got, want = c.plaquette_profile_sums(axis), lat.plaquette_profile_sums(axis)
                assert np.max(np.abs(got - want) / np.abs(want)) < 1e-12
                gp, wp = c.polyakov_profile_sums(axis), lat.polyakov_profile_sums(axis)
                assert np.allclose(gp, wp, rtol=1e-10, atol=1e-12)
            c.plaquette_sums()
            c.polyakov_sums()
    lat.close()


def _check_cases():
    import os
    return sorted(f for f in os.listdir(ol.GOLDEN_DIR) if f.startswith("chk_su3_") and f.endswith(".json"))


@pytest.mark.parametrize("fname", _check_cases())
def test_observables_match_reference_cpu_remeasurement(fname):
    """Plaquette, action, Polyakov loop, Wilson loop and the F_mu_nu traces of the fixture's
    configuration against what the reference's own CPU code (su3cpu.cpp lattice_check_cpu) computed
    for it; 1e-11 relative (different order of the sums)."""
    import json
    import os
    import qcdgpu_amd as q
    with open(os.path.join(ol.GOLDEN_DIR, fname)) as f:
        doc = json.load(f)
    table = np.load(os.path.join(ol.GOLDEN_DIR, doc["table"]), allow_pickle=False)
    e = doc["expected"]
    with q.Context(group=3, L=doc["L"], beta=doc["beta"], phi=doc.get("phi", 0.0), omega=doc.get("omega", 0.0)) as c:
        c.upload(table)
        plq, act, pol = c.plaquette(), c.action(), c.polyakov()
        w = c.wilson_loop(doc["wilson_R"], doc["wilson_T"])
        f = c.fmunu(doc["field"] - 1, doc["index1"], doc["index2"])
        fs = c.fmunu(doc["field"] - 1, doc["index1"], doc["index2"], sums=True)
        with pytest.raises(q.QcdError):
            c.fmunu(2, 3, 8)
    assert np.allclose(plq[:2], [e["Plq_spat"], e["Plq_temp"]], rtol=1e-12)
    assert np.allclose(act[:2], [e["S_spat"], e["S_temp"]], rtol=1e-12)
    assert np.allclose(pol, [e["Polyakov_loop"], e["Polyakov_loop_im"], e["Polyakov_loop_P2"], e["Polyakov_loop_P4"]],
                       rtol=1e-11, atol=1e-14)
    assert w == pytest.approx(e["Wilson_loop"], rel=1e-11)
    want = np.array([e["F_%s_%s_%s" % (p, a, k)] for a in ("3", "8") for p in "abc" for k in ("re", "im")])
    assert np.allclose(f, want, rtol=1e-10, atol=1e-14)
    assert np.allclose(fs / np.prod(doc["L"]), f, rtol=1e-14)


def _su2_check_cases():
    import os
    return sorted(f for f in os.listdir(ol.GOLDEN_DIR) if f.startswith("chk_su2_") and f.endswith(".json"))


@pytest.mark.parametrize("fname", _su2_check_cases())
def test_su2_observables_match_reference_cpu_remeasurement(fname):
    """SU(2): the same against the reference's su2cpu.cpp (F_mu_nu with Pauli matrices)."""
    import json
    import os
    import qcdgpu_amd as q
    with open(os.path.join(ol.GOLDEN_DIR, fname)) as f:
        doc = json.load(f)
    table = np.load(os.path.join(ol.GOLDEN_DIR, doc["table"]), allow_pickle=False)
    e = doc["expected"]
    with q.Context(group=2, L=doc["L"], beta=doc["beta"], phi=doc.get("phi", 0.0), omega=doc.get("omega", 0.0)) as c:
        c.upload(table)
        plq, act, pol = c.plaquette(), c.action(), c.polyakov()
        w = c.wilson_loop(doc["wilson_R"], doc["wilson_T"])
        f = c.fmunu(doc["field"] - 1, doc["index1"], doc["index2"])
        with pytest.raises(q.QcdError):
            c.fmunu(0, 3, 8)          # SU(2) has three generators
    assert np.allclose(plq[:2], [e["Plq_spat"], e["Plq_temp"]], rtol=1e-12)
    assert np.allclose(act[:2], [e["S_spat"], e["S_temp"]], rtol=1e-12)
    assert np.allclose(pol, [e["Polyakov_loop"], e["Polyakov_loop_im"], e["Polyakov_loop_P2"], e["Polyakov_loop_P4"]],
                       rtol=1e-11, atol=1e-14)
    assert w == pytest.approx(e["Wilson_loop"], rel=1e-11)
    want = np.array([e["F_%s_%s_%s" % (p, a, k)] for a in ("3", "8") for p in "abc" for k in ("re", "im")])
    assert np.allclose(f, want, rtol=1e-10, atol=1e-14)


def _binned(x, nbins):
    x = np.asarray(x, dtype=float)
    b = x[: len(x) // nbins * nbins].reshape(nbins, -1).mean(axis=1)
    return x.mean(), b.std(ddof=1) / np.sqrt(nbins)


STAT_CASES = [(f, g) for f, g in (("stat_su3_8x8x8x8_b6.json", 3), ("stat_su2_8x8x8x8_b2.5.json", 2), ("stat_su3_16x16x16x16_b6.json", 3),
                                  ("stat_su3_32x32x32x32_b6.json", 3)) if os.path.exists(os.path.join(ol.GOLDEN_DIR, f))]


@pytest.mark.parametrize("fname,group", STAT_CASES)
def test_mean_plaquette_agrees_with_reference_chain(fname, group):
    """Statistical leg: the reference's own program (one random stream) and this engine (one stream per thread) run
    different realisations of the same Markov chain from the same cold start, so the mean plaquette over the same sweeps
    must agree within errors: the fixture's thermalisation sweeps, then one sample per sweep, as many as the fixture holds
    (1200 at 8^4, 500 at the benchmark's 16^4, 150 at its 32^4 -- ten minutes and more than an hour of the reference's
    program).  The same window on both sides, because from a cold start the plaquette is still settling after hundreds
    of sweeps.  Errors from 20 bins (10 for the short series); tolerance 4 sigma."""
    import qcdgpu_amd as q
    fx = ol.golden(fname)
    a = fx["init"]
    ref = fx["plq_total"]
    with q.Context(group=group, L=(a["LS"], a["LS"], a["LS"], a["LT"]), beta=a["BETA"], nhit=a["NHIT"], rand_series=a["RANDSERIES"]) as c:
        c.init_lattice()
        c.sweep(a["NAV"])
        mine = []
        for _ in range(len(ref)):
            c.sweep(1)
            mine.append(c.plaquette()[2])
    nbins = 20 if len(ref) >= 400 else 10
    m_ref, e_ref = _binned(ref, nbins)
    m_gpu, e_gpu = _binned(mine, nbins)
    assert abs(m_ref - m_gpu) < 4.0 * np.hypot(e_ref, e_gpu), (m_ref, e_ref, m_gpu, e_gpu)
    assert e_gpu < 1e-3 and e_ref < 1e-3


@pytest.mark.parametrize("fname,group", [("stat_su3_8x8x8x8_b6.json", 3), ("stat_su2_8x8x8x8_b2.5.json", 2)])
def test_overrelaxation_leaves_the_measure_alone(fname, group):
    """Overrelaxation has no counterpart in the reference (parity unpinned).  Beyond its defining
    property (action unchanged link by link, tested elsewhere) a chain of one heat-bath sweep and two
    overrelaxation sweeps per sample must give the reference chain's mean plaquette."""
    import qcdgpu_amd as q
    fx = ol.golden(fname)
    a = fx["init"]
    with q.Context(group=group, L=(8, 8, 8, 8), beta=a["BETA"], nhit=a["NHIT"], rand_series=77) as c:
        c.init_lattice()
        for _ in range(50):
            c.sweep(1)
            c.overrelax_sweep(2)
        mine = []
        for _ in range(1500):
            c.sweep(1)
            c.overrelax_sweep(2)
            mine.append(c.plaquette()[2])
    m_ref, e_ref = _binned(fx["plq_total"], 20)
    m_gpu, e_gpu = _binned(mine, 25)
    assert abs(m_ref - m_gpu) < 4.0 * np.hypot(e_ref, e_gpu), (m_ref, e_ref, m_gpu, e_gpu)
    assert e_gpu < 1e-3


def test_partial_passes_must_tile_the_slab():
    """All random streams of a launch are read from the same state slot, and a pass flips it: a range
    whose slices have been through different numbers of passes is refused, not silently mis-seeded."""
    import qcdgpu_amd as q
    with q.Context(group=3, L=(8, 4, 4, 4), beta=6.0, x0=0, nx_local=8, halo=1) as c:
        c.init_lattice()
        c.update_range(1, 0, 0, 2)          # slices 0, 1 of the pass ...
        with pytest.raises(q.QcdError, match="pass"):
            c.update_range(1, 0, 0, 8)      # ... so the whole slab is no longer in step
        c.update_range(1, 0, 2, 6)          # the rest of the pass: in step again
        c.update_range(1, 1, 6, 4)          # wraps around: slices 6, 7, 0, 1
        c.update_range(1, 1, 2, 4)
        c.update(0, 0)


@pytest.mark.parametrize("beta,nhit", [(6.0, 10), (1.0, 12)])
def test_generator_state_round_trip_mid_sweep(beta, nhit):
    """Links and generator state saved after an ODD number of passes (the states then sit in the
    second of their two slots; at beta = 1 many went through the redo pass) and loaded into a fresh
    context: both contexts continue to the same links, bit for bit."""
    import qcdgpu_amd as q
    L = (4, 6, 4, 4)
    with q.Context(group=3, L=L, beta=beta, nhit=nhit, rand_series=21) as a, \
            q.Context(group=3, L=L, beta=beta, nhit=nhit, rand_series=99) as b:
        a.init_lattice()
        a.sweep(1)
        a.update(1, 0)
        table, state = a.download(), a.rng_state()
        b.upload(table)
        b.set_rng_state(state)
        for c in (a, b):
            for d in (1, 2, 3):
                c.update(1, d)
            for d in range(4):
                c.update(0, d)
            c.sweep(1)
        assert a.rng_draws() == b.rng_draws()
        assert np.array_equal(a.download(), b.download())


def test_config2_su3_16_heatbath_and_overrelaxation_vs_cpu():
    """The benchmark's heat-bath + overrelaxation configuration at its stated size: SU(3) 16^4 fp64, heat-bath AND overrelaxation
    sweeps alternating, plaquette against the CPU path to 1e-12 after every sweep (the
    overrelaxation half has no counterpart in the reference: checked against the checker's
    restatement and by the action staying put)."""
    L = (16, 16, 16, 16)
    lat = ol.Lattice(3, L, 6.0, nhit=10, ints=1, rng_mode=ol.RNG_PERSITE)
    with _ctx(group=3, L=L, beta=6.0, nhit=10) as c:
        c.init_lattice()
        for sweep in range(2):
            c.sweep(1)
            lat.sweep(1)
            assert _rel(c.plaquette(), lat.plaquette()) < REL, sweep
            before = c.plaquette()[2]
            c.overrelax_sweep(1)
            lat.overrelax_sweep(1)
            assert abs(c.plaquette()[2] - before) < 1e-13, sweep
            assert _rel(c.plaquette(), lat.plaquette()) < REL, sweep
            assert c.rng_draws() == ol.rng_draws(lat)
        assert np.allclose(c.polyakov(), lat.polyakov(), rtol=1e-10, atol=1e-13)
    lat.close()


@pytest.mark.parametrize("share", [0, 1, 2])
def test_config3_su3_32_against_checker_fixture(share):
    """The benchmark's headline lattice at its stated size: SU(3) 32^4, beta 6, cold start, two heat-bath
    sweeps against the CPU checker's per-thread-stream run of the same lattice (committed fixture,
    oracle/make_golden_persite32.py): plaquette to 1e-12, equal counts of random numbers drawn
    (the same accept/reject history at every one of 8.4 million link updates), Polyakov loop.
    share = 0: the library's own choice of sites per stream at this size (4); 1 and 2: asked for."""
    L = (32, 32, 32, 32)
    with _ctx(group=3, L=L, beta=6.0, nhit=10, rand_series=54, rng_share=share) as c:
        assert c.rng_share == (share or 4)
        fx = ol.golden_persite32(c.rng_share)
        c.init_lattice()
        for row in fx["sweeps"]:
            c.sweep(1)
            assert c.rng_draws() == row["rng_draws"], row["sweep"]
            assert _rel(c.plaquette(), row["plaquette"]) < REL, row["sweep"]
            assert np.allclose(c.polyakov(), row["polyakov"], rtol=1e-9, atol=1e-13), row["sweep"]


@pytest.mark.parametrize("group,L,phi,omega", [(3, (4, 4, 6, 4), 0.7, -0.4), (3, (6, 4, 4, 4), 0.0, 1.1), (2, (4, 6, 4, 4), 0.9, 0.0)])
def test_twisted_boundary_condition_passes_match_checker(group, L, phi, omega):
    """PHI / OMEGA: every read of U_y at x = Nx-1 is Omega U_y, an updated link is stored as the heat-bath returns it
    (the reference's device path, SU(N)/suncl/su3_matrix_memory.cl:57-98, suncl.cl:525-536).  The reference has this on
    its device path and in its CPU re-measurement only: measurements under the twist are pinned by the chk_*_tw
    fixtures above, the update pass is **parity unpinned** and checked against the checker's restatement -- gid mode
    link by link after every pass, then a RANLUX run with equal draw counts."""
    import qcdgpu_amd as q
    beta = 5.9 if group == 3 else 2.4
    lat = ol.Lattice(group, L, beta, nhit=6, ints=2, rng_mode=ol.RNG_GID, phi=phi, omega=omega)
    with _ctx(group=group, L=L, beta=beta, nhit=6, start_mode=q.START_GID, phi=phi, omega=omega) as c, \
            _ctx(group=group, L=L, beta=beta, nhit=6, start_mode=q.START_GID) as plain:
        c.init_lattice()
        plain.init_lattice()
        assert np.array_equal(c.download(), plain.download())                 # the stored start is the untwisted one
        assert _rel(c.plaquette(), lat.plaquette()) < 1e-12
        assert abs(c.plaquette()[2] - plain.plaquette()[2]) > 1e-4            # and the twist is seen by the measurement
        for parity in (1, 0):
            for d in range(4):
                c.update(parity, d)
                lat.update(parity, d)
                assert np.max(np.abs(c.download() - lat.export_table())) < 1e-12, (parity, d)
        assert _rel(c.plaquette(), lat.plaquette()) < 1e-12
        assert abs(c.wilson_loop(2, 1) - lat.wilson_loop(2, 1)) < 1e-12
        with pytest.raises(q.QcdError):
            c.overrelax_sweep(1)                                               # not with a twist
    lat.close()
    lat = ol.Lattice(group, L, beta, nhit=6, ints=1, rng_mode=ol.RNG_PERSITE, seed=17, phi=phi, omega=omega)
    with _ctx(group=group, L=L, beta=beta, nhit=6, rand_series=17, phi=phi, omega=omega) as c:
        c.init_lattice()
        for sweep in range(2):
            c.sweep(1)
            lat.sweep(1)
            assert c.rng_draws() == ol.rng_draws(lat), sweep
            assert _rel(c.plaquette(), lat.plaquette()) < REL, sweep
        if group == 3:
            for field in (0, 1):
                assert np.allclose(c.fmunu(field, 3, 8, sums=True), lat.fmunu_sums(field, 3, 8), rtol=1e-9, atol=1e-11)
        for axis in range(3):
            assert np.allclose(c.plaquette_profile_sums(axis), lat.plaquette_profile_sums(axis), rtol=1e-11, atol=1e-11)
    lat.close()
    with pytest.raises(q.QcdError, match="twist"):
        _ctx(group=group, L=L, beta=beta, x0=0, nx_local=2, halo=1, phi=phi, omega=omega)   # one uncut lattice only
    with pytest.raises(q.QcdError, match="twist"):
        _ctx(group=group, L=(L[0], L[1], L[2], 5), beta=beta, phi=phi, omega=omega)         # and even extents

