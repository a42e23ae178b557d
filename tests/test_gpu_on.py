"""GPU parity tests of the O(N) scalar path (qcd_on_* of the C-ABI) against the CPU checker and
the golden fixtures minted from the reference's own CPU re-measurement."""
import json
import os

import numpy as np
import pytest

from tests import oracle_lib as ol

pytestmark = pytest.mark.gpu

ON_CASES = sorted(f for f in os.listdir(ol.GOLDEN_DIR) if f.startswith("on_") and f.endswith(".json"))


def _q():
    import qcdgpu_amd as q
    return q


def _par(p):
    return dict(sk_group=p["skgroup"], z=p["z"], lam=p["lam"], zeta=p["zeta"], eta=p["eta"], b=p["b"])


@pytest.mark.parametrize("fname", ON_CASES)
def test_measurements_match_reference_fixture(fname):
    """Double precision: action, field moments and correlators of the fixture's field against what
    the reference's CPU code (O(N)/suncl/suncpu.cpp) computed for it.  Tolerance 1e-12 relative:
    the expressions differ in summation order only."""
    q = _q()
    with open(os.path.join(ol.GOLDEN_DIR, fname)) as f:
        doc = json.load(f)
    with q.OnContext(doc["L"], precision=1, **_par(doc["params"])) as c:
        c.upload(np.array(doc["field"]))
        got = c.measure(doc["correlator_step"])
    e = doc["expected"]
    want = np.array([e["action"], e["field"], e["field2"], e["correlator1"], e["correlator2"]])
    assert np.allclose(got, want, rtol=1e-12, atol=1e-12), (got, want)


@pytest.mark.parametrize("precision", [0, 1])
def test_cold_and_hot_start(precision):
    q = _q()
    L = (4, 6, 4, 8)
    with q.OnContext(L, precision=precision, start_mode=q.START_COLD) as c:
        c.init_lattice()
        assert np.all(c.download() == 0.25)
        m = c.measure()
    lat = ol.OnLattice(L, precision=precision)
    lat.init(hot=False)
    assert m[0] == pytest.approx(lat.action(), rel=1e-6 if precision == 0 else 1e-13)
    with q.OnContext(L, precision=precision, start_mode=q.START_HOT) as c:
        c.init_lattice()
        got = c.download()
        draws = c.counters()[0]
        assert c.max_U == lat.max_U
    lat.init(hot=True)
    assert np.array_equal(got, lat.field)          # r * max_U: one rounding, the same on both sides
    assert draws == lat.rng_draws
    lat.close()


@pytest.mark.parametrize("L,nhit,lux,par", [
    ((8, 8, 8, 8), 10, 3, {}),
    ((4, 6, 10, 8), 5, 2, {}),                                # odd NHIT: the single closing proposal
    ((6, 4, 4, 6), 14, 3, {}),                                # more than 24 numbers: a second batch
    ((4, 4, 6, 6), 10, 0, dict(eta=0.0, zeta=0.5, skgroup=3, z=0.2, b=8.0)),
])
def test_update_double_precision_matches_checker(L, nhit, lux, par):
    """Double precision: same accept/reject history (equal random-number and acceptance counters)
    and the same field to 1e-13 after every sweep."""
    q = _q()
    p = dict(ol.ON_DEFAULTS)
    p.update(par)
    lat = ol.OnLattice(L, precision=1, nhit=nhit, lux_level=lux, randseries=11, **par)
    lat.init(hot=True)
    with q.OnContext(L, precision=1, nhit=nhit, lux_level=lux, rand_series=11, start_mode=q.START_HOT, **_par(p)) as c:
        c.init_lattice()
        for sweep in range(3):
            c.update(1)
            lat.update_pass(1)
            assert c.counters()[2] == int(lat.acceptance[1])
            c.update(0)
            lat.update_pass(0)
            cnt = c.counters()
            assert cnt[0] == lat.rng_draws, "accept/reject history differs from the checker"
            assert cnt[1] == int(lat.acceptance[0])
            got = c.download()
            assert np.max(np.abs(got - lat.field)) < 1e-13, sweep
        m = c.measure((2, 1, 1, 3))
    want = [lat.action(), *lat.field_moments(), *lat.correlators((2, 1, 1, 3))]
    assert np.allclose(m, want, rtol=1e-12)
    lat.close()


def test_update_single_precision_tracks_checker():
    """Single precision: the device's logf/expf and glibc's differ in the last place now and then,
    and a proposal within that distance of its threshold is decided differently (about one in 1e6
    proposals).  One sweep from the same start must therefore agree bit for bit on all but a few
    sites, and the lattice averages to 2e-5."""
    q = _q()
    L = (8, 8, 8, 16)
    lat = ol.OnLattice(L, precision=0, nhit=10, randseries=5)
    lat.init(hot=True)
    with q.OnContext(L, precision=0, nhit=10, rand_series=5, start_mode=q.START_HOT) as c:
        c.init_lattice()
        c.sweep(1)
        lat.sweep(1)
        got = c.download()
        m = c.measure()
        cnt = c.counters()
    same = np.count_nonzero(got == lat.field)
    assert same >= got.size - 8, (same, got.size)
    assert abs(cnt[0] - lat.rng_draws) <= 64
    want = np.array([lat.action(), *lat.field_moments(), *lat.correlators()])
    assert np.allclose(m, want, rtol=2e-5), (m, want)
    lat.close()


def test_thermalised_averages_agree_between_precisions():
    """Statistical leg: 40 sweeps from a hot start in single and in double precision give the same
    mean field within 3 sigma of the sweep-to-sweep scatter."""
    q = _q()
    L = (8, 8, 8, 8)
    means = []
    for precision in (0, 1):
        with q.OnContext(L, precision=precision, start_mode=q.START_HOT, rand_series=3 + precision) as c:
            c.init_lattice()
            c.sweep(20)
            vals = []
            for _ in range(20):
                c.sweep(1)
                vals.append(c.measure()[1])
            means.append((np.mean(vals), np.std(vals)))
    (m0, s0), (m1, s1) = means
    assert abs(m0 - m1) < 3.0 * max(s0, s1) + 1e-3, means


def test_upload_download_and_errors():
    q = _q()
    L = (4, 4, 6, 4)
    rng = np.random.default_rng(0)
    f = (rng.random(int(np.prod(L))) * 0.3).astype(np.float32)
    with q.OnContext(L, precision=0) as c:
        with pytest.raises(q.QcdError):
            c.sweep(1)                      # nothing initialised yet
        c.upload(f)
        assert np.array_equal(c.download(), f)
        with pytest.raises(q.QcdError):
            c.measure((9, 0, 0, 0))         # step beyond the lattice
    with pytest.raises(q.QcdError):
        q.OnContext((4, 5, 4, 4))           # odd Ny
    with pytest.raises(q.QcdError):
        q.OnContext(L, precision=2)


def test_full_size_single_precision_properties():
    """The configuration the reference's O(N) figures are quoted on, 64^4 in single precision:
    size-independent properties.  Two contexts with the same seed produce the same field bit for
    bit; every value is a proposal r * max_U in [0, max_U) or the cold value; the random-number
    count is between one and five quads per site and pass; the action per site falls from its
    cold value."""
    q = _q()
    L = (64, 64, 64, 64)
    fields = []
    for _ in range(2):
        with q.OnContext(L, precision=0, nhit=10, rand_series=7) as c:
            c.init_lattice()
            s0 = c.measure()[0]
            c.sweep(2)
            cnt = c.counters()
            m = c.measure()
            fields.append(c.download())
            max_u = c.max_U
    assert np.array_equal(fields[0], fields[1])
    f = fields[0]
    assert f.min() >= 0.0 and f.max() < max_u
    sites = f.size
    assert 4 * 2 * sites <= cnt[0] <= 4 * 5 * 2 * sites
    assert 0 < cnt[1] <= 10 * sites // 2 and 0 < cnt[2] <= 10 * sites // 2
    assert np.isfinite(m).all() and m[0] != s0


QCDGPU_ON = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "qcdgpu_amd", "bin", "qcdgpu_on")


def test_host_program_reproduces_checker_run(tmp_path):
    """qcdgpu_on (init file -> run -> result file) in double precision against the same run of the
    checker: every column of the result table."""
    import re
    import subprocess
    L, nav, it, niter, nhit = (6, 4, 6, 4), 2, 4, 2, 10
    with open(tmp_path / "init.dat", "w") as f:
        f.write("\tL1 = 6\n\tL2 = 4\n\tL3 = 6\n\tL4 = 4\n\tNAV = %d\n\tITER = %d\n\tNITER = %d\n\tNHIT = %d\n" % (nav, it, niter, nhit))
        f.write("\tINTS = 0\n\tPRNG = RANLUX3\n\tRANDSERIES = 21\n\tPRECISION = 2\n\tOUTPUTPATH = %s/\n" % tmp_path)
    r = subprocess.run([QCDGPU_ON, str(tmp_path / "init.dat"), "-ON_ETA=0.2"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr + r.stdout[-2000:]
    text = open(re.search(r"result file: (\S+)", r.stdout).group(1)).read()
    assert " eta                         : 2.0000000000000e-01" in text
    rows = np.array([[float(x) for x in ln.split()[1:]] for ln in text.splitlines() if re.match(r"^\s+\d+\s+[- ]\d\.\d+e", ln)])
    assert rows.shape == (it, 11)

    lat = ol.OnLattice(L, precision=1, nhit=nhit, randseries=21, eta=0.2)
    lat.init(hot=True)
    step = (4, 4, 4, 4)
    sites = lat.s.sites

    def sample(updated):
        f1, f2 = lat.field_moments()
        c1, c2 = lat.correlators(step)
        s = lat.action()
        are = lat.acceptance[0] / (sites * nhit * niter) if updated else 0.0
        aro = lat.acceptance[1] / (sites * nhit * niter) if updated else 0.0
        return [f1, f2, f2 - f1 * f1, s, 0.0, s, c1 - f1 * f1, c2 - f1 * f1, are, aro, 0.5 * (are + aro)]

    want = [sample(False)]
    lat.sweep(nav)
    for _ in range(it - 1):
        lat.sweep(niter)
        want.append(sample(True))
    want = np.array(want)
    assert np.allclose(rows, want, rtol=1e-11, atol=1e-12), np.max(np.abs(rows - want))
    lat.close()


def test_host_program_rejects_what_it_does_not_run(tmp_path):
    import subprocess
    r = subprocess.run([QCDGPU_ON, "-GROUP=3"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "GROUP = 1" in r.stderr
