"""GPU tests of the single-precision build of the library (libqcdgpu_amd_f32.so, qcd_precision() = 1).

What the reference offers for single precision: its OpenCL kernels built with -D PRECISION=0 (float
links and arithmetic, fp64 sums; SU(N)/kernel/complex.h:57-92) and, as the only check, the CPU
re-measurement of the final configuration in double precision with a relative tolerance of 5e-5
(SU(N)/data_analysis/data_analysis.cpp:40-43, 76-101).  Its sequential CPU program is double precision
only, so the update pass in single precision is **parity unpinned**; here it is tied to the pinned
fp64 path instead:
  * measurements of one and the same configuration, float library against double library and against the
    reference-minted chk_* fixtures, to the reference's own 5e-5;
  * the deterministic gid mode (no accept/reject decisions: results depend smoothly on the arithmetic)
    against the fp64 checker over the first sweeps: plaquettes to 5e-6, links to 1e-6 on average;
  * heat-bath chains: same mean plaquette as the reference's chain within 4 sigma.
"""
import json
import os

import numpy as np
import pytest

from tests import oracle_lib as ol

pytestmark = pytest.mark.gpu

CHECK_REL = 5e-5      # CHECKING_PRECISION of the reference for single precision
CHECK_FLOOR = 1e-6    # values below this are not compared (data_analysis.cpp:42, 78)


def _ctx(**kw):
    import qcdgpu_amd as q
    return q.Context(**kw)


def _check(a, b, rel=CHECK_REL):
    a, b = np.atleast_1d(np.asarray(a, float)), np.atleast_1d(np.asarray(b, float))
    big = (np.abs(a) > CHECK_FLOOR) & (np.abs(b) > CHECK_FLOOR)
    # + 2e-7: averages of O(1) terms of both signs (Im P, Re P of a small lattice) carry float rounding of that size whatever they add up to
    assert np.all(np.abs(a[big] - b[big]) <= rel * np.abs(b[big]) + 2e-7), (a, b)
    assert np.all(np.abs(a[~big] - b[~big]) <= 10 * CHECK_FLOOR), (a, b)


def test_library_reports_its_precision_and_refuses_the_other():
    import qcdgpu_amd as q
    from qcdgpu_amd import capi
    assert capi.lib.qcd_precision() == 2 and capi.lib_f32().qcd_precision() == 1
    p = capi.Params()
    p.group, p.nd, p.precision, p.beta, p.nhit, p.start_mode, p.lux_level, p.rand_series = 3, 4, 1, 6.0, 10, 1, 3, 54
    for i in range(4):
        p.L[i] = 4
    p.nx_local = 4
    import ctypes as C
    h = C.c_void_p()
    assert capi.lib.qcd_create(C.byref(h), C.byref(p)) == 4            # QCD_ERR_UNSUPPORTED: the double library, single asked
    p.precision = 2
    assert capi.lib_f32().qcd_create(C.byref(h), C.byref(p)) == 4
    with q.Context(group=3, L=(4, 4, 4, 4), beta=6.0, precision=3) as c:   # mixed = single in the reference's kernels
        c.init_lattice()
        assert np.all(c.plaquette() == 3.0)


@pytest.mark.parametrize("group", [2, 3])
def test_cold_start_known_answers(group):
    with _ctx(group=group, L=(4, 6, 4, 8), beta=6.0, precision=1) as c:
        c.init_lattice()
        assert np.all(c.plaquette() == float(group))
        assert np.all(c.action() == 0.0)
        p = c.polyakov()
        assert p[0] == 1.0 and p[1] == 0.0 and p[2] == 1.0 and p[3] == 1.0
        t = c.download()
        assert t.dtype == np.float32


@pytest.mark.parametrize("group,L", [(3, (8, 8, 8, 8)), (2, (8, 8, 8, 8)), (3, (4, 6, 8, 4)), (2, (6, 4, 2, 8)), (3, (2, 34, 4, 4)),
                                     (3, (5, 4, 3, 4))])   # the last one: odd extents, passes in phases
def test_gid_mode_follows_the_fp64_checker(group, L):
    """Closed-form start and closed-form 'random' numbers: float links are the rounded double links, and
    two sweeps later plaquettes and links are still those of the fp64 checker to float rounding."""
    import qcdgpu_amd as q
    beta = 6.0 if group == 3 else 2.5
    lat = ol.Lattice(group, L, beta, nhit=10, ints=2, rng_mode=ol.RNG_GID)
    with _ctx(group=group, L=L, beta=beta, nhit=10, start_mode=q.START_GID, precision=1) as c:
        c.init_lattice()
        assert np.array_equal(c.download(), lat.export_table().astype(np.float32))
        for _ in range(2):
            c.sweep(1)
            lat.sweep(1)
            assert np.max(np.abs(c.plaquette() - lat.plaquette()) / lat.plaquette()) < 5e-6
            dl = np.abs(c.download() - lat.export_table())
            # float rounding, amplified where the heat-bath forms sin = sqrt(1 - cos^2) near cos = +-1
            assert np.mean(dl) < 1e-6 and np.max(dl) < 2e-3
        _check(c.polyakov(), lat.polyakov())
    lat.close()


@pytest.mark.parametrize("group,beta", [(3, 6.0), (2, 2.5)])
def test_measurements_of_one_configuration_in_both_precisions(group, beta):
    """The reference's check: the configuration a single-precision run ends with, measured again in double
    precision (there on the CPU, here by the double library), 5e-5 relative."""
    L = (8, 6, 8, 4)
    with _ctx(group=group, L=L, beta=beta, precision=1) as c, _ctx(group=group, L=L, beta=beta) as d:
        c.init_lattice()
        c.sweep(8)
        table = c.download()
        # links leave the update re-unitarised in float: row norms and orthogonality to float rounding
        if group == 3:
            re_u, im_u = table[0][:, :3], table[4][:, :3]      # rows uv1, uv2 of direction x: Re, Im of u1..u3
            assert np.max(np.abs((re_u ** 2 + im_u ** 2).sum(axis=1) - 1.0)) < 1e-6
        d.upload(table.astype(np.float64))
        _check(c.plaquette(), d.plaquette())
        _check(c.action(), d.action())
        _check(c.polyakov(), d.polyakov())
        _check(c.wilson_loop(2, 2), d.wilson_loop(2, 2))
        i1, i2 = (3, 8) if group == 3 else (3, 2)
        for field in (0, 1):
            _check(c.fmunu(field, i1, i2), d.fmunu(field, i1, i2), rel=2e-4)   # sums of signed terms that nearly cancel
        for axis in range(3):
            _check(c.plaquette_profile_sums(axis), d.plaquette_profile_sums(axis))
            _check(c.polyakov_profile_sums(axis), d.polyakov_profile_sums(axis), rel=5e-4)


def test_reference_fixture_measured_in_single_precision():
    """A configuration and the numbers the reference's CPU code computed for it (chk_su3 fixture): uploaded as
    floats and measured by the float library, within the reference's single-precision tolerance."""
    fname = sorted(f for f in os.listdir(ol.GOLDEN_DIR) if f.startswith("chk_su3_") and f.endswith(".json"))[0]
    with open(os.path.join(ol.GOLDEN_DIR, fname)) as f:
        doc = json.load(f)
    table = np.load(os.path.join(ol.GOLDEN_DIR, doc["table"]), allow_pickle=False)
    e = doc["expected"]
    with _ctx(group=3, L=doc["L"], beta=doc["beta"], precision=1) as c:
        c.upload(table)
        _check(c.plaquette()[:2], [e["Plq_spat"], e["Plq_temp"]])
        _check(c.action()[:2], [e["S_spat"], e["S_temp"]])
        _check(c.polyakov(), [e["Polyakov_loop"], e["Polyakov_loop_im"], e["Polyakov_loop_P2"], e["Polyakov_loop_P4"]], rel=2e-4)
        _check(c.wilson_loop(doc["wilson_R"], doc["wilson_T"]), e["Wilson_loop"])


def _binned(x, nbins):
    x = np.asarray(x, dtype=float)
    b = x[: len(x) // nbins * nbins].reshape(nbins, -1).mean(axis=1)
    return x.mean(), b.std(ddof=1) / np.sqrt(nbins)


@pytest.mark.parametrize("fname,group", [("stat_su3_8x8x8x8_b6.json", 3), ("stat_su2_8x8x8x8_b2.5.json", 2)])
def test_mean_plaquette_agrees_with_reference_chain(fname, group):
    """Statistical leg in single precision: 8^4, the reference's chain (double precision, sequential stream)
    against 2000 samples of the float library; 4 sigma of the binned errors."""
    fx = ol.golden(fname)
    a = fx["init"]
    with _ctx(group=group, L=(8, 8, 8, 8), beta=a["BETA"], nhit=a["NHIT"], rand_series=a["RANDSERIES"], precision=1) as c:
        c.init_lattice()
        c.sweep(a["NAV"])
        mine = []
        for _ in range(2000):
            c.sweep(1)
            mine.append(c.plaquette()[2])
    m_ref, e_ref = _binned(fx["plq_total"], 20)
    m_gpu, e_gpu = _binned(mine, 25)
    assert abs(m_ref - m_gpu) < 4.0 * np.hypot(e_ref, e_gpu), (m_ref, e_ref, m_gpu, e_gpu)


@pytest.mark.parametrize("group,beta", [(3, 6.0), (2, 2.4)])
def test_overrelaxation_keeps_the_action(group, beta):
    L = (8, 8, 4, 6)
    with _ctx(group=group, L=L, beta=beta, precision=1) as c:
        c.init_lattice()
        c.sweep(3)
        before, links = c.plaquette()[2], c.download()
        c.overrelax_sweep(1)
        assert abs(c.plaquette()[2] - before) < 2e-6 * before          # float rounding of 4 V reflections
        assert np.max(np.abs(c.download() - links)) > 0.1              # and every link moved


def test_slabs_equal_the_uncut_lattice():
    """x-slabs with halo exchange through the ring (device copies): the same plaquette sums and draw counts as
    one context, in single precision as in double (the per-site arithmetic does not depend on the cut)."""
    import qcdgpu_amd as q
    from qcdgpu_amd import capi
    L, beta = (12, 8, 8, 8), 6.0
    with _ctx(group=3, L=L, beta=beta, precision=1) as one:
        one.init_lattice()
        one.sweep(2)
        want, draws = one.plaquette_sums(), one.rng_draws()
    parts, x0 = [], 0
    for nx in (4, 6, 2):
        parts.append(q.Context(group=3, L=L, beta=beta, x0=x0, nx_local=nx, halo=1, precision=1))
        x0 += nx
    ring = capi.Ring(parts)
    for p in parts:
        p.init_lattice()
    ring.fill_halos()
    ring.sweep(2)
    ring.synchronize()
    got = sum(p.plaquette_sums() for p in parts)
    assert np.max(np.abs(got - want) / np.abs(want)) < 1e-13
    assert sum(p.rng_draws() for p in parts) == draws
    ring.close()
    for p in parts:
        p.close()


def test_su3_32_pair_kernel_path():
    """32^4 SU(3): large launches run as gather / heat-bath wavefront pairs.  Two sweeps from a cold start stay
    within float rounding of the fp64 checker's run of the same lattice (committed fixture; at beta 6 the
    accept/reject decisions of two sweeps are far from their thresholds for all but a handful of links)."""
    with _ctx(group=3, L=(32, 32, 32, 32), beta=6.0, nhit=10, rand_series=54, precision=1) as c:
        fx = ol.golden_persite32(c.rng_share)
        c.init_lattice()
        for row in fx["sweeps"]:
            c.sweep(1)
            assert abs(c.rng_draws() - row["rng_draws"]) < 1e-4 * row["rng_draws"]
            assert np.max(np.abs(c.plaquette() - row["plaquette"]) / np.asarray(row["plaquette"])) < 1e-5
        t = c.download()
        re_u, im_u = t[0][:, :3], t[4][:, :3]
        assert np.max(np.abs((re_u ** 2 + im_u ** 2).sum(axis=1) - 1.0)) < 1e-6
        # at this size the plaquette and F_mu_nu kernels march along x, eight slices per thread: the same configuration
        # measured by the double library
        with _ctx(group=3, L=(32, 32, 32, 32), beta=6.0) as d:
            d.upload(t.astype(np.float64))
            _check(c.plaquette(), d.plaquette())
            _check(c.wilson_loop(2, 3), d.wilson_loop(2, 3))
            for field in (0, 1):
                _check(c.fmunu(field, 3, 8), d.fmunu(field, 3, 8), rel=2e-4)
            # the overrelaxation pass of this size runs through the same wavefront pairs: one sweep in both precisions
            # from the same links (reflections have no accept step, so float and double stay within rounding)
            before = c.plaquette()[2]
            c.overrelax_sweep(1)
            d.overrelax_sweep(1)
            assert abs(c.plaquette()[2] - before) < 2e-6 * before
            moved = c.download()
            assert np.max(np.abs(moved - d.download())) < 2e-5
            assert np.mean(np.abs(moved - t)) > 1e-2


def test_hot_start_and_state_round_trip():
    import qcdgpu_amd as q
    L = (6, 8, 4, 6)
    with _ctx(group=3, L=L, beta=5.8, start_mode=q.START_HOT, precision=1) as c:
        c.init_lattice()
        p0 = c.plaquette()[2]
        assert abs(p0) < 0.2                       # random links: plaquette near zero
        c.sweep(2)
        table, state = c.download(), c.rng_state()
        c.sweep(1)
        want = c.plaquette()
    with _ctx(group=3, L=L, beta=5.8, precision=1) as d:
        d.upload(table)
        d.set_rng_state(state)
        d.sweep(1)
        assert np.array_equal(d.plaquette(), want)   # links and generator state saved: the run continues bit for bit


def test_cli_single_precision_program(tmp_path):
    """qcdgpu_f32, the host program linked with the single-precision library: result file says so, the state file
    carries the precision word 1 and a float4 link table (as the reference's single-precision build writes its
    lattice_table buffer), a continued run reproduces the uninterrupted one digit for digit, and the double
    program refuses the file."""
    import re
    import shutil
    import struct
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe32 = os.path.join(root, "qcdgpu_amd", "bin", "qcdgpu_f32")
    exe64 = os.path.join(root, "qcdgpu_amd", "bin", "qcdgpu")

    def run(exe, d, args=(), ok=True):
        d.mkdir(exist_ok=True)
        with open(d / "init.dat", "w") as f:
            for k, v in dict(GROUP=3, ND=4, L1=8, L2=4, L3=6, L4=4, BETA=5.7, NHIT=4, INTS=1, PRNG="RANLUX2",
                             RANDSERIES=77, NAV=2, ITER=6, NITER=2).items():
                f.write("\t%s = %s\n" % (k, v))
            f.write("\tOUTPUTPATH = %s/\n" % d)
        r = subprocess.run([exe, str(d / "init.dat")] + list(args), capture_output=True, text=True, timeout=300)
        assert (r.returncode == 0) == ok, r.stderr + r.stdout[-2000:]
        return r.stdout + r.stderr

    def table_of(out):
        text = open(re.search(r"result file: (\S+)", out).group(1)).read()
        rows = [[float(x) for x in line.split()[1:]] for line in text.split(" Data fields")[1].splitlines()
                if line.split() and line.split()[0].isdigit()]
        return text, np.array(rows)

    text, full = table_of(run(exe32, tmp_path / "a"))
    assert " precision                   : single" in text
    text64, full64 = table_of(run(exe64, tmp_path / "a64"))
    assert " precision                   : double" in text64
    # same seeds, same accept/reject history on this small lattice for the first samples: columns agree to float rounding
    assert np.max(np.abs(full[:3, :3] - full64[:3, :3])) < 2e-5
    out = run(exe32, tmp_path / "b", ["--save-state", "--save-every", "1e9", "--stop-at", "3"])
    qcg = re.findall(r"state file: (\S+)", out)[-1]
    raw = open(qcg, "rb").read()
    head = struct.unpack("<64I", raw[:256])
    assert head[18] == 1                          # the precision word (convert_precision_to_uint: single = 1)
    sites, rs = 8 * 4 * 6 * 4, 768
    off = 256 + 3 * 256 * 16
    table = np.frombuffer(raw[off:off + 12 * rs * 16], dtype=np.float32).reshape(12, rs, 4)
    u = table[0:4, :sites, :3].astype(np.float64) + 1j * table[4:8, :sites, :3]
    assert np.max(np.abs(np.sum(np.abs(u) ** 2, axis=2) - 1)) < 1e-6
    assert raw[off + 12 * rs * 16:off + 12 * rs * 16 + 7] == b"QCDRLX1"
    (tmp_path / "c").mkdir()
    shutil.copy(qcg, tmp_path / "c" / "state.qcg")
    _, resumed = table_of(run(exe32, tmp_path / "c", ["--load", "state.qcg"]))
    assert np.array_equal(resumed, full)
    (tmp_path / "d").mkdir()
    shutil.copy(qcg, tmp_path / "d" / "state.qcg")
    assert "ERROR" in run(exe64, tmp_path / "d", ["--load", "state.qcg"], ok=False).upper()
