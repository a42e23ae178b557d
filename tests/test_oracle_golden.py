"""The CPU restatement (oracle/) against the vectors minted from the compiled reference.

Bar: the reference prints 14 significant digits ("% 16.13e"); the restatement must reproduce
every printed digit, i.e. agree to 5e-14 relative (NaN where the reference gives NaN).
"""
import ctypes as C
import json
import math
import os

import numpy as np
import pytest

from tests import oracle_lib as ol

REL = 5e-14


def _check(got, want):
    for g, w in zip(np.ravel(got), np.ravel(want)):
        if w is None or (isinstance(w, float) and math.isnan(w)):
            assert math.isnan(g)
        else:
            assert abs(g - w) <= REL * max(abs(w), 1e-300) + 1e-300, (g, w)


@pytest.mark.parametrize("case", ol.golden_cases())
def test_restatement_reproduces_reference(case):
    fx = ol.golden(case)
    a = ol.init_to_args(fx["init"])
    mode = ol.RNG_GID if a["ints"] == 2 else ol.RNG_SEQ
    plq, act = ol.simulate(a["group"], a["L"], a["beta"], a["nhit"], a["ints"], mode, a["lux"],
                           a["seed"], a["nav"], a["iter_"], a["niter"])
    want = np.array([[np.nan if v is None else v for v in row] for row in fx["samples"]], dtype=float)
    _check(plq, want[:, 0:3])
    _check(act, want[:, 3:6])
    if fx["summary"] and not any(math.isnan(v) for v in fx["summary"].values()):
        m, v = ol.estimates(plq[:, 2])
        _check([m, v], [fx["summary"]["Mean Plq_total"], fx["summary"]["Variance Plq_total"]])
        m, v = ol.estimates(act[:, 2])
        _check([m, v], [fx["summary"]["Mean S_total"], fx["summary"]["Variance S_total"]])


def test_glibc_rand_restatement_matches_libc():
    libc = C.CDLL("libc.so.6")
    for seed in (0, 1, 54, 4321, 2**31 + 5):
        g = ol.GlibcRand()
        ol.lib.qo_srand(C.byref(g), seed)
        libc.srand(seed)
        for _ in range(1000):
            assert ol.lib.qo_rand(C.byref(g)) == libc.rand()


def test_ranlux_known_values():
    # RANDSERIES = 54 -> first rand() = 707676680; seed table values printed by the reference
    assert ol.lib.qo_stream_jseed(54, 0) == 707676680
    st = ol.RanluxState()
    ol.lib.qo_ranlux_init(C.byref(st), 707676680, 223)
    assert abs(st.seeds[0] - 0.283334374) < 5e-9
    assert abs(st.seeds[1] - 0.365860522) < 5e-9
    assert abs(st.seeds[2] - 0.687620342) < 5e-9
    assert abs(st.seeds[23] - 0.599847376) < 5e-9
    xs = [ol.lib.qo_ranlux_next(C.byref(st)) for _ in range(100)]
    assert all(0.0 < x < 1.0 for x in xs)
    assert all(x * 2**24 == int(x * 2**24) for x in xs if x >= 2**-12)


def test_cold_start_known_answers():
    for group in (2, 3):
        lat = ol.Lattice(group, (4, 4, 4, 4), 6.0, ints=1)
        assert np.all(lat.plaquette() == float(group))
        assert np.all(lat.action() == 0.0)
        p = lat.polyakov()
        assert p[0] == 1.0 and p[1] == 0.0 and p[2] == 1.0 and p[3] == 1.0
        lat.close()


def test_hot_start_links_are_group_elements():
    """Hot start (restated from the reference's device kernels; no runnable reference: parity
    unpinned): unitary, unit determinant, plaquette near zero, reproducible from the seed."""
    for group in (2, 3):
        a = ol.Lattice(group, (4, 4, 4, 6), 5.0, ints=0, rng_mode=ol.RNG_PERSITE, seed=5)
        b = ol.Lattice(group, (4, 4, 4, 6), 5.0, ints=0, rng_mode=ol.RNG_PERSITE, seed=5)
        c = ol.Lattice(group, (4, 4, 4, 6), 5.0, ints=0, rng_mode=ol.RNG_PERSITE, seed=6)
        ta, tb, tc = a.export_table(), b.export_table(), c.export_table()
        assert np.array_equal(ta, tb) and not np.array_equal(ta, tc)
        assert ol.rng_draws(a) == a.sites * 4 * (12 if group == 3 else 4)
        assert abs(a.plaquette()[2]) < 0.15
        if group == 3:
            # rows of the 12-real storage: u = (uv1.xyz + i uv2.xyz), v = (uv3.xy, uv1.w) + i (uv3.zw, uv2.w)
            for d in range(4):
                u = ta[0 * 4 + d, :, :3] + 1j * ta[1 * 4 + d, :, :3]
                v = np.stack([ta[2 * 4 + d, :, 0] + 1j * ta[2 * 4 + d, :, 2],
                              ta[2 * 4 + d, :, 1] + 1j * ta[2 * 4 + d, :, 3],
                              ta[0 * 4 + d, :, 3] + 1j * ta[1 * 4 + d, :, 3]], axis=1)
                assert np.max(np.abs(np.sum(np.abs(u) ** 2, axis=1) - 1)) < 1e-14
                assert np.max(np.abs(np.sum(np.abs(v) ** 2, axis=1) - 1)) < 1e-14
                assert np.max(np.abs(np.sum(u * np.conj(v), axis=1))) < 1e-14
        else:
            assert np.max(np.abs(np.sum(ta ** 2, axis=2) - 1)) < 1e-14
        for l in (a, b, c):
            l.close()


def test_overrelaxation_keeps_the_action():
    """Overrelaxation is not in the reference (parity unpinned): its defining property is that
    every link move leaves Re Tr(U A), hence the plaquette average, unchanged."""
    for group, beta in ((3, 6.0), (2, 2.5)):
        lat = ol.Lattice(group, (4, 4, 6, 4), beta, ints=1, rng_mode=ol.RNG_PERSITE, seed=8)
        lat.sweep(4)
        p0, t0 = lat.plaquette(), lat.export_table().copy()
        lat.overrelax_sweep(2)
        p1, t1 = lat.plaquette(), lat.export_table()
        assert abs(p1[2] - p0[2]) < 1e-13               # the total; space- and time-like parts trade
        assert np.max(np.abs(t1 - t0)) > 0.5          # but the configuration moved, a lot
        lat.close()


def test_wilson_loop_known_answers():
    """W(R,T) (device path of the reference only: parity unpinned): cold start gives N, the 1x1
    loop is the time-like plaquette, and larger loops are smaller on a thermalised lattice."""
    for group, beta in ((3, 6.0), (2, 2.5)):
        lat = ol.Lattice(group, (4, 6, 4, 6), beta, ints=1, rng_mode=ol.RNG_PERSITE, seed=4)
        assert lat.wilson_loop(2, 3) == float(group)
        lat.sweep(3)
        assert abs(lat.wilson_loop(1, 1) - lat.plaquette()[1]) < 1e-14
        assert lat.wilson_loop(1, 1) > lat.wilson_loop(2, 1) > lat.wilson_loop(2, 2) > lat.wilson_loop(3, 3) > 0
        lat.close()


def test_device_layout_roundtrip():
    for group in (2, 3):
        lat = ol.Lattice(group, (4, 4, 6, 4), 5.0, ints=2, rng_mode=ol.RNG_GID)
        lat.sweep(1)
        p0 = lat.plaquette()
        t = lat.export_table(rowsize=lat.sites + 64)
        lat2 = ol.Lattice(group, (4, 4, 6, 4), 5.0, ints=1)
        lat2.import_table(t)
        # third row / second row are rebuilt from the stored reals: ulp-level agreement only
        assert np.allclose(lat2.plaquette(), p0, rtol=1e-13)
        lat.close(); lat2.close()


def test_parity_maps_partition_the_lattice():
    L = (4, 6, 2, 4)
    sites = int(np.prod(L))
    Lc = (C.c_int * 4)(*L)
    odd = [ol.lib.qo_odd_gid(Lc, i) for i in range(sites // 2)]
    even = [ol.lib.qo_even_gid(Lc, i) for i in range(sites // 2)]
    assert sorted(odd + even) == list(range(sites))
    for i in range(sites // 2):
        assert {odd[i], even[i]} == {2 * i, 2 * i + 1}
    for gid in range(sites):
        for d in range(4):
            assert ol.lib.qo_neighbour_back(Lc, ol.lib.qo_neighbour(Lc, gid, d), d) == gid


# ---------------------------------------------------------------------------------------------
# O(N) scalar path: the checker's measurements against the reference's CPU re-measurement
# (fixtures minted by oracle/make_golden_on.py from oracle/_ref/qcdgpu_on_check)

ON_CASES = sorted(f for f in os.listdir(ol.GOLDEN_DIR) if f.startswith("on_") and f.endswith(".json"))


@pytest.mark.parametrize("fname", ON_CASES)
def test_on_measurements_match_reference(fname):
    with open(os.path.join(ol.GOLDEN_DIR, fname)) as f:
        doc = json.load(f)
    p = doc["params"]
    lat = ol.OnLattice(doc["L"], precision=1, **p)
    lat.field[:] = np.array(doc["field"])
    e = doc["expected"]
    # the kernel's expression and the reference's CPU one differ in summation order only
    assert lat.action() == pytest.approx(e["action"], rel=1e-12, abs=1e-12)
    f1, f2 = lat.field_moments()
    assert f1 == pytest.approx(e["field"], rel=1e-13)
    assert f2 == pytest.approx(e["field2"], rel=1e-13)
    c1, c2 = lat.correlators(doc["correlator_step"])
    assert c1 == pytest.approx(e["correlator1"], rel=1e-13)
    assert c2 == pytest.approx(e["correlator2"], rel=1e-13)
    lat.close()


def test_on_single_precision_tracks_double():
    """The float build of the checker against its double build on the same field."""
    a = ol.OnLattice((4, 4, 4, 6), precision=1)
    b = ol.OnLattice((4, 4, 4, 6), precision=0)
    a.init(hot=True)
    b.init(hot=True)
    assert np.allclose(a.field, b.field, rtol=2e-7)
    assert b.action() == pytest.approx(a.action(), rel=2e-5)
    assert np.allclose(b.field_moments(), a.field_moments(), rtol=2e-6)
    a.close()
    b.close()


def test_on_update_properties():
    """Metropolis pass: only sites of the pass's parity change, proposals stay inside [0, max_U),
    the stream of every thread starts at a fresh batch, and the acceptance sums are consistent."""
    L = (4, 4, 6, 4)
    lat = ol.OnLattice(L, precision=1, nhit=5)
    lat.init(hot=True)
    before = np.array(lat.field)
    d0 = lat.rng_draws
    lat.update_pass(1)
    after = np.array(lat.field)
    changed = np.nonzero(before != after)[0]
    assert len(changed) > 0
    for gid in changed:
        c = ol.gid_to_coords(L, int(gid))
        assert sum(c) % 2 == 1
    assert after.min() >= 0.0 and after.max() < lat.max_U
    draws = lat.rng_draws - d0
    half = lat.s.sites // 2
    assert 4 * half <= draws <= 4 * 3 * half      # NHIT = 5: one to three quads per site
    acc = lat.acceptance[1]
    assert 0 <= acc <= 5 * half
    lat.close()


def test_profiles_add_up_to_the_lattice_sums():
    """Plane-resolved plaquette and Polyakov sums: every axis' planes add up to the lattice sums,
    and a cold lattice gives the same value in every plane."""
    L = (4, 6, 4, 6)
    lat = ol.Lattice(3, L, 5.8, nhit=5, ints=1, rng_mode=ol.RNG_PERSITE)
    for axis in range(3):
        p = lat.plaquette_profile_sums(axis)
        others = np.prod(L) // L[axis]
        assert np.allclose(p, 9.0 * others)
        q = lat.polyakov_profile_sums(axis)
        assert np.allclose(q[:, 0], 3.0 * others / L[3]) and np.allclose(q[:, 1], 0.0)
    lat.sweep(2)
    sites = np.prod(L)
    plq = lat.plaquette()
    pol = lat.polyakov()
    for axis in range(3):
        p = lat.plaquette_profile_sums(axis).sum(axis=0)
        assert p[0] / (3 * sites) == pytest.approx(plq[0], rel=1e-12)
        assert p[1] / (3 * sites) == pytest.approx(plq[1], rel=1e-12)
        q = lat.polyakov_profile_sums(axis).sum(axis=0)
        n123 = L[0] * L[1] * L[2]
        assert q[0] / (3 * n123) == pytest.approx(pol[0], rel=1e-11, abs=1e-14)
        assert q[1] / (3 * n123) == pytest.approx(pol[1], rel=1e-11, abs=1e-14)
    lat.close()


# ---------------------------------------------------------------------------------------------
# SU(3) observables that the reference's sequential CPU program does not have, pinned against its
# CPU re-measurement code (fixtures minted by oracle/make_golden_su3_check.py)

CHK_CASES = sorted(f for f in os.listdir(ol.GOLDEN_DIR) if f.startswith("chk_su3_") and f.endswith(".json"))


def load_check_case(fname):
    with open(os.path.join(ol.GOLDEN_DIR, fname)) as f:
        doc = json.load(f)
    table = np.load(os.path.join(ol.GOLDEN_DIR, doc["table"]), allow_pickle=False)
    return doc, table


def expected_fmunu(doc):
    e = doc["expected"]
    return np.array([e["F_%s_%s_%s" % (p, a, c)] for a in ("3", "8") for p in "abc" for c in ("re", "im")])


@pytest.mark.parametrize("fname", CHK_CASES)
def test_polyakov_wilson_fmunu_match_reference_cpu_code(fname):
    doc, table = load_check_case(fname)
    e = doc["expected"]
    lat = ol.Lattice(3, doc["L"], doc["beta"], nhit=1, ints=1, rng_mode=ol.RNG_PERSITE, phi=doc.get("phi", 0.0), omega=doc.get("omega", 0.0))
    lat.import_table(table)
    plq = lat.plaquette()
    act = lat.action()
    assert plq[0] == pytest.approx(e["Plq_spat"], rel=1e-13) and plq[1] == pytest.approx(e["Plq_temp"], rel=1e-13)
    assert act[0] == pytest.approx(e["S_spat"], rel=1e-12) and act[1] == pytest.approx(e["S_temp"], rel=1e-12)
    pol = lat.polyakov()
    want = [e["Polyakov_loop"], e["Polyakov_loop_im"], e["Polyakov_loop_P2"], e["Polyakov_loop_P4"]]
    assert np.allclose(pol, want, rtol=1e-12, atol=1e-15)
    assert lat.wilson_loop(doc["wilson_R"], doc["wilson_T"]) == pytest.approx(e["Wilson_loop"], rel=1e-12)
    f = lat.fmunu_sums(doc["field"] - 1, doc["index1"], doc["index2"]) / np.prod(doc["L"])
    assert np.allclose(f, expected_fmunu(doc), rtol=1e-11, atol=1e-15)
    lat.close()


SU2_CHK_CASES = sorted(f for f in os.listdir(ol.GOLDEN_DIR) if f.startswith("chk_su2_") and f.endswith(".json"))


@pytest.mark.parametrize("fname", SU2_CHK_CASES)
def test_su2_polyakov_wilson_match_reference_cpu_code(fname):
    """SU(2): plaquettes, action, Polyakov loop, Wilson loop and the F_mu_nu traces (Pauli matrices)
    against the reference's SU(2) CPU re-measurement (SU(N)/suncl/su2cpu.cpp lattice_check_cpu)."""
    doc, table = load_check_case(fname)
    e = doc["expected"]
    lat = ol.Lattice(2, doc["L"], doc["beta"], nhit=1, ints=1, rng_mode=ol.RNG_PERSITE, phi=doc.get("phi", 0.0), omega=doc.get("omega", 0.0))
    lat.import_table(table)
    plq, act, pol = lat.plaquette(), lat.action(), lat.polyakov()
    assert np.allclose(plq[:2], [e["Plq_spat"], e["Plq_temp"]], rtol=1e-13)
    assert np.allclose(act[:2], [e["S_spat"], e["S_temp"]], rtol=1e-12)
    want = [e["Polyakov_loop"], e["Polyakov_loop_im"], e["Polyakov_loop_P2"], e["Polyakov_loop_P4"]]
    assert np.allclose(pol, want, rtol=1e-12, atol=1e-15)
    assert lat.wilson_loop(doc["wilson_R"], doc["wilson_T"]) == pytest.approx(e["Wilson_loop"], rel=1e-12)
    f = lat.fmunu_sums(doc["field"] - 1, doc["index1"], doc["index2"]) / np.prod(doc["L"])
    assert np.allclose(f, expected_fmunu(doc), rtol=1e-11, atol=1e-15)
    lat.close()


def test_persite_streams_sample_the_same_measure_as_the_reference_chain():
    """The checker's per-thread-stream mode (what the GPU engine does) against the reference
    program's single-stream chain: SU(2) 8^4 beta = 2.5, mean plaquette after 100 thermalisation
    sweeps, 300 samples here against the fixture's 1200, binned errors, 4 sigma."""
    fx = ol.golden("stat_su2_8x8x8x8_b2.5.json")
    a = fx["init"]
    lat = ol.Lattice(2, (8, 8, 8, 8), a["BETA"], nhit=a["NHIT"], ints=1, rng_mode=ol.RNG_PERSITE)
    lat.sweep(a["NAV"])
    mine = []
    for _ in range(300):
        lat.sweep(1)
        mine.append(lat.plaquette()[2])
    lat.close()

    def binned(x, nbins):
        x = np.asarray(x, dtype=float)
        b = x[: len(x) // nbins * nbins].reshape(nbins, -1).mean(axis=1)
        return x.mean(), b.std(ddof=1) / np.sqrt(nbins)
    m_ref, e_ref = binned(fx["plq_total"], 20)
    m_own, e_own = binned(mine, 15)
    assert abs(m_ref - m_own) < 4.0 * np.hypot(e_ref, e_own), (m_ref, e_ref, m_own, e_own)
