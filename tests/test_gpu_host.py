"""The C++ host shell (qcdgpu binary: init.dat -> run -> result file) and BigLattice on a GPU."""
import json
import math
import os
import re
import subprocess

import numpy as np
import pytest

from tests import oracle_lib as ol

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
QCDGPU = os.path.join(ROOT, "qcdgpu_amd", "bin", "qcdgpu")


def _run(tmp_path, keys, extra="", args=()):
    tmp_path.mkdir(exist_ok=True)
    with open(tmp_path / "init.dat", "w") as f:
        for k, v in keys.items():
            f.write("\t%s = %s\n" % (k, v))
        f.write(extra)
        f.write("\tOUTPUTPATH = %s/\n" % tmp_path)
    r = subprocess.run([QCDGPU, str(tmp_path / "init.dat")] + list(args), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr + r.stdout[-2000:]
    m = re.search(r"result file: (\S+)", r.stdout)
    if m is None:
        return None, None, None, r.stdout
    rows, fields, in_table = [], None, False
    text = open(m.group(1)).read()
    for line in text.splitlines():
        if line.startswith(" Data fields"):
            in_table = True
            continue
        if in_table and line.strip().startswith("#,"):
            fields = line.replace("#,", "").split()
        parts = line.split()
        if in_table and parts and parts[0].isdigit():
            rows.append([float(x) for x in parts[1:]])
    return text, fields, np.array(rows), r.stdout


@pytest.mark.parametrize("case", ["ref_su3_8x8x8x8_gid.json", "ref_su2_6x8x4x10_gid.json", "ref_su3_5x4x3x7_gid.json"])
def test_cli_reproduces_reference_result_table(tmp_path, case):
    fx = ol.golden(case)
    text, fields, rows, _ = _run(tmp_path, fx["init"])
    assert fields[:6] == ["Plq_spat", "Plq_temp", "Plq_total", "S_spat", "S_temp", "S_total"]
    want = np.array(fx["samples"], dtype=float)
    assert rows.shape[0] == fx["init"]["ITER"]
    assert np.max(np.abs(rows[:, :6] - want) / np.maximum(np.abs(want), 1e-300)) < 1e-12
    # header lines a parser of the reference's files keys on
    for label in (" lattice size                :", " ints (0=hot, 1=cold, 2=gid) : 2", " BETA                        :",
                  " Mean Plq_total", " Variance Plq_total", " Data fields:"):
        assert label in text
    mean = float(re.search(r"Mean Plq_total\s*:\s*(\S+)", text).group(1))
    assert abs(mean - fx["summary"]["Mean Plq_total"]) < 1e-12 * abs(mean)


def test_cli_twisted_boundary_condition(tmp_path):
    """PHI / OMEGA in init.dat (the reference's keys, SU(N)/suncl/suncl.cpp:372-373): the host passes them on, the table is the
    checker's twisted gid run, the header carries both angles, and a cut lattice or overrelaxation sweeps are refused."""
    keys = dict(GROUP=3, ND=4, L1=4, L2=4, L3=6, L4=4, BETA=5.9, NHIT=6, INTS=2, PRNG="RANLUX3", RANDSERIES=54,
                NAV=1, ITER=3, NITER=1, PHI=0.7, OMEGA=-0.4)
    text, fields, rows, _ = _run(tmp_path, keys)
    lat = ol.Lattice(3, (4, 4, 6, 4), 5.9, nhit=6, ints=2, rng_mode=ol.RNG_GID, phi=0.7, omega=-0.4)
    want = [lat.plaquette()]
    lat.sweep(1)
    for _ in range(2):
        lat.sweep(1)
        want.append(lat.plaquette())
    lat.close()
    assert np.max(np.abs(rows[:, :3] - np.array(want)) / np.array(want)) < 1e-12
    assert " PHI   (lambda_3)            : 7.0000000000000e-01" in text and " OMEGA (lambda_8)            : -4.0000000000000e-01" in text
    for extra in ("\tN1 = {2, 2}\n\tDEVICE = {0, 0}\n", "\tNOVERRELAX = 1\n"):
        d = tmp_path / ("x%d" % len(extra))
        d.mkdir()
        with open(d / "init.dat", "w") as f:
            for k, v in keys.items():
                f.write("\t%s = %s\n" % (k, v))
            f.write(extra + "\tOUTPUTPATH = %s/\n" % d)
        r = subprocess.run([QCDGPU, str(d / "init.dat")], capture_output=True, text=True, timeout=120)
        assert r.returncode != 0 and "twisted" in (r.stdout + r.stderr)


def test_cli_biglattice_equals_single_lattice(tmp_path):
    """N1 = {..}: the lattice cut into x-slabs (two and three parts on one device, the way the
    reference exercises BigLattice on one GPU) gives the same run as the uncut lattice."""
    keys = dict(GROUP=3, ND=4, L1=12, L2=6, L3=6, L4=4, BETA=6.0, NHIT=10, INTS=1, PRNG="RANLUX3",
                RANDSERIES=54, NAV=1, ITER=3, NITER=1)
    _, _, single, _ = _run(tmp_path / "a", keys)
    _, _, two, _ = _run(tmp_path / "b", keys, "\tN1 = {8, 4}\n\tDEVICE = {0, 0}\n")
    _, _, three, _ = _run(tmp_path / "c", keys, "\tN1 = {2, 6, 4}\n\tDEVICE = {0, 0, 0}\n")
    # same RANLUX streams (keyed by global site), same arithmetic: identical up to the order of
    # the final sums of the measurement
    assert np.max(np.abs(two - single)) < 1e-13
    assert np.max(np.abs(three - single)) < 1e-13
    lat = ol.Lattice(3, (12, 6, 6, 4), 6.0, nhit=10, ints=1, rng_mode=ol.RNG_PERSITE)
    lat.sweep(2)
    want = np.concatenate([lat.plaquette(), lat.action()])
    assert np.max(np.abs(single[1, :6] - want) / np.abs(want)) < 1e-11
    lat.close()


def test_cli_hot_start_single_and_biglattice(tmp_path):
    """INTS = 0 through the host shell: sample 0 is a random configuration (plaquette ~ 0), and
    the run is the same whether the lattice is cut into parts or not."""
    keys = dict(GROUP=3, ND=4, L1=8, L2=6, L3=4, L4=6, BETA=5.9, NHIT=6, INTS=0, PRNG="RANLUX3",
                RANDSERIES=31, NAV=1, ITER=3, NITER=1)
    text, _, single, _ = _run(tmp_path / "a", keys)
    _, _, two, _ = _run(tmp_path / "b", keys, "\tN1 = {2, 6}\n\tDEVICE = {0, 0}\n")
    assert " ints (0=hot, 1=cold, 2=gid) : 0" in text
    assert abs(single[0, 2]) < 0.1 and single[2, 2] > 0.5
    assert np.max(np.abs(two - single)) < 1e-13
    lat = ol.Lattice(3, (8, 6, 4, 6), 5.9, nhit=6, ints=0, rng_mode=ol.RNG_PERSITE, seed=31)
    assert np.max(np.abs(single[0, :3] - lat.plaquette())) < 1e-13
    lat.sweep(2)
    assert np.max(np.abs(single[1, :3] - lat.plaquette()) / np.abs(lat.plaquette())) < 1e-11
    lat.close()


def test_cli_heatbath_plus_overrelaxation(tmp_path):
    """NOVERRELAX (an init.dat key the reference does not have): k microcanonical sweeps after
    every heat-bath sweep, single lattice = lattice in parts = the checker."""
    keys = dict(GROUP=3, ND=4, L1=8, L2=4, L3=4, L4=6, BETA=6.0, NHIT=5, INTS=1, PRNG="RANLUX3",
                RANDSERIES=9, NAV=1, ITER=3, NITER=1, NOVERRELAX=2)
    _, _, single, _ = _run(tmp_path / "a", keys)
    _, _, two, _ = _run(tmp_path / "b", keys, "\tN1 = {4, 4}\n\tDEVICE = {0, 0}\n")
    assert np.max(np.abs(two - single)) < 1e-13
    lat = ol.Lattice(3, (8, 4, 4, 6), 6.0, nhit=5, ints=1, rng_mode=ol.RNG_PERSITE, seed=9)
    lat.sweep(1)                      # NAV = 1
    lat.overrelax_sweep(2)
    for sample in (1, 2):
        lat.sweep(1)
        lat.overrelax_sweep(2)
        assert np.max(np.abs(single[sample, :3] - lat.plaquette()) / lat.plaquette()) < 1e-11
    lat.close()


def test_cli_wilson_loop_column(tmp_path):
    """GETWILSON adds the WILSONR x WILSONT loop as the column behind the plaquettes, as in the
    reference's result file; it survives a stop and --load."""
    keys = dict(GROUP=2, ND=4, L1=6, L2=4, L3=4, L4=6, BETA=2.3, NHIT=5, INTS=1, PRNG="RANLUX3",
                RANDSERIES=17, NAV=1, ITER=4, NITER=1, GETWILSON=1, WILSONR=2, WILSONT=3)
    text, fields, rows, _ = _run(tmp_path / "a", keys)
    assert fields[:4] == ["Plq_spat", "Plq_temp", "Plq_total", "Wilson_loop"] and " Mean Wilson_loop" in text
    lat = ol.Lattice(2, (6, 4, 4, 6), 2.3, nhit=5, ints=1, rng_mode=ol.RNG_PERSITE, seed=17)
    assert rows[0, 3] == 2.0
    lat.sweep(2)
    assert abs(rows[1, 3] - lat.wilson_loop(2, 3)) < 1e-12
    lat.close()
    _, _, none, out = _run(tmp_path / "b", keys, args=["--save-state", "--save-every", "1e9", "--stop-at", "2"])
    qcg = re.findall(r"state file: (\S+)", out)[-1]
    import shutil
    (tmp_path / "c").mkdir()
    shutil.copy(qcg, tmp_path / "c" / "s.qcg")
    _, _, resumed, _ = _run(tmp_path / "c", keys, args=["--load", "s.qcg"])
    assert np.array_equal(resumed, rows)


def test_cli_state_file_resume_is_exact(tmp_path):
    """lattice_save_state / lattice_load_state: a run stopped after sample 2 and continued from
    its .qcg file gives the same table, digit for digit, as the uninterrupted run -- also when
    the continuation cuts the lattice into parts.  The file starts with the reference's header."""
    import shutil
    import struct
    keys = dict(GROUP=3, ND=4, L1=8, L2=4, L3=6, L4=4, BETA=5.7, NHIT=4, INTS=1, PRNG="RANLUX2",
                RANDSERIES=77, NAV=2, ITER=6, NITER=2)
    _, _, full, _ = _run(tmp_path / "a", keys)
    _, _, none, out = _run(tmp_path / "b", keys, args=["--save-state", "--save-every", "1e9", "--stop-at", "3"])
    assert none is None and "stopped after sample 2" in out
    qcg = re.findall(r"state file: (\S+)", out)[-1]
    raw = open(qcg, "rb").read()
    head = struct.unpack("<64I", raw[:256])
    assert raw[:6] == b"QCDGPU" and raw[8:11] == b"1.5"
    assert head[5] == 1 and head[6] == 77 and head[7] == 4 and head[8] == 97      # cold, seed, RANLUX2, nskip
    assert head[9] == 8 * (2 + 2 * 2) and head[10] == 2 and head[11] == 2         # passes, NAV, NAV_counter
    assert head[12] == 2 and head[13] == 6 and head[14] == 3 and head[15] == 4    # NITER, ITER, ITER_counter, NHIT
    assert struct.unpack("<d", raw[19 * 4:21 * 4])[0] == 5.7
    assert head[36:46] == (3, 4, 8, 4, 6, 4, 8, 4, 6, 4)
    # the link table sits where the reference's reader expects it: header, three sample sections of
    # align(ITER) double2, then 12 rows of align(sites) double4 in the reference device layout
    sites, rs = 8 * 4 * 6 * 4, 768
    off = 256 + 3 * 256 * 16
    table = np.frombuffer(raw[off:off + 12 * rs * 32], dtype=np.float64).reshape(12, rs, 4)
    u = table[0:4, :sites, :3] + 1j * table[4:8, :sites, :3]
    assert np.max(np.abs(np.sum(np.abs(u) ** 2, axis=2) - 1)) < 1e-14
    assert raw[off + 12 * rs * 32:off + 12 * rs * 32 + 7] == b"QCDRLX1"
    for name, extra in (("c", ""), ("d", "\tN1 = {2, 6}\n\tDEVICE = {0, 0}\n")):
        (tmp_path / name).mkdir()
        shutil.copy(qcg, tmp_path / name / "state.qcg")
        _, _, resumed, out2 = _run(tmp_path / name, keys, extra, args=["--load", "state.qcg"])
        assert "NAV_start=2, ITER_start=3" in out2
        assert np.array_equal(resumed, full), name


def test_rng_state_roundtrip_through_the_c_abi():
    import qcdgpu_amd as q
    kw = dict(group=2, L=(4, 4, 4, 4), beta=2.2, nhit=3, rand_series=5)
    with q.Context(**kw) as a, q.Context(**kw) as b:
        a.init_lattice(); b.init_lattice()
        a.sweep(2)
        st, links = a.rng_state(), a.download()
        a.sweep(1)
        b.upload(links)
        b.set_rng_state(st)
        b.sweep(1)
        assert np.array_equal(a.download(), b.download())
        assert a.rng_draws() == b.rng_draws()
        with pytest.raises(q.QcdError):
            b.set_rng_state(st[:-4])


def test_biglattice_python_driver_on_one_gpu():
    """Several slabs on one device through the pack/unpack halves of UpdateEdges."""
    import qcdgpu_amd as q
    from qcdgpu_amd import biglattice as bl
    L = (8, 4, 6, 4)
    for group, beta in ((3, 5.7), (2, 2.3)):
        lat = ol.Lattice(group, L, beta, nhit=5, ints=1, rng_mode=ol.RNG_PERSITE, seed=7)
        big = bl.make_gpu_slabs(group, L, beta, [2, 4, 2], nhit=5, rand_series=7)
        for s in range(2):
            big.sweep(1)
            lat.sweep(1)
            assert np.max(np.abs(big.plaquette() - lat.plaquette()) / np.abs(lat.plaquette())) < 1e-12
        assert np.allclose(big.polyakov(), lat.polyakov(), rtol=1e-10, atol=1e-13)
        big.overrelax_sweep(1)          # microcanonical sweep across the slab boundaries
        lat.overrelax_sweep(1)
        assert np.max(np.abs(big.plaquette() - lat.plaquette()) / np.abs(lat.plaquette())) < 1e-12
        draws = sum(e.ctx.rng_draws() for e in big.engines)
        assert draws == ol.rng_draws(lat)
        lat.close()


@pytest.mark.parametrize("beta,nhit,concurrent", [(5.8, 4, True), (1.0, 12, True), (5.8, 4, False)])
def test_biglattice_boundary_first_update_on_one_gpu(beta, nhit, concurrent, monkeypatch):
    """Slabs wide enough for the split pass: boundary slices, halo messages, interior -- as two
    concurrent launches (qcd_update_split; the low-beta case keeps both lanes' redo lists busy) or
    one after the other (qcd_update_range).  Same configuration as the uncut lattice; the entry
    points reject ranges outside the slab."""
    import qcdgpu_amd as q
    from qcdgpu_amd import biglattice as bl
    L = (16, 4, 4, 6)
    lat = ol.Lattice(3, L, beta, nhit=nhit, ints=1, rng_mode=ol.RNG_PERSITE, seed=29)
    big = bl.make_gpu_slabs(3, L, beta, [8, 8], nhit=nhit, rand_series=29)
    if not concurrent:
        monkeypatch.setattr(bl.BigLattice, "_concurrent", lambda self: False)
    assert big.boundary_depth() == (1 if concurrent else 2)     # one slice per side / a quarter of the narrowest slab
    for s in range(2):
        big.sweep(1)
        lat.sweep(1)
        assert np.max(np.abs(big.plaquette() - lat.plaquette()) / np.abs(lat.plaquette())) < 1e-12
    assert sum(e.ctx.rng_draws() for e in big.engines) == ol.rng_draws(lat)
    got = np.concatenate([e.ctx.download() for e in big.engines], axis=1)
    want = lat.export_table()
    assert np.max(np.abs(got - want)) < 1e-11
    with pytest.raises(q.QcdError):
        big.engines[0].ctx.update_range(0, 0, 8, 1)
    with pytest.raises(q.QcdError):
        big.engines[0].ctx.update_range(0, 0, 0, 9)
    with pytest.raises(q.QcdError):
        big.engines[0].ctx.update_split(0, 0, 0, 0)
    lat.close()


def test_hot_start_biglattice_equals_single_lattice():
    """Hot start cut into slabs: halos come from the neighbours' streams through one exchange."""
    import qcdgpu_amd as q
    from qcdgpu_amd import biglattice as bl
    L = (8, 4, 4, 6)
    lat = ol.Lattice(3, L, 5.9, nhit=4, ints=0, rng_mode=ol.RNG_PERSITE, seed=3)
    big = bl.make_gpu_slabs(3, L, 5.9, [4, 2, 2], nhit=4, rand_series=3, start_mode=q.START_HOT)
    assert np.max(np.abs(big.plaquette() - lat.plaquette())) < 1e-13
    for s in range(2):
        big.sweep(1)
        lat.sweep(1)
        assert np.max(np.abs(big.plaquette() - lat.plaquette()) / np.abs(lat.plaquette())) < 1e-11
    assert sum(e.ctx.rng_draws() for e in big.engines) == ol.rng_draws(lat)
    lat.close()


def test_gid_biglattice_matches_reference_vector():
    import qcdgpu_amd as q
    from qcdgpu_amd import biglattice as bl
    fx = ol.golden("ref_su3_4x6x8x10_gid.json")
    a = ol.init_to_args(fx["init"])
    big = bl.make_gpu_slabs(a["group"], a["L"], a["beta"], [2, 2], nhit=a["nhit"], start_mode=q.START_GID,
                            rand_series=a["seed"])
    rows = [np.concatenate([big.plaquette(), big.action()])]
    big.sweep(a["nav"])
    for _ in range(1, a["iter_"]):
        big.sweep(a["niter"])
        rows.append(np.concatenate([big.plaquette(), big.action()]))
    want = np.array(fx["samples"], dtype=float)
    assert np.max(np.abs(np.array(rows) - want) / np.abs(want)) < 1e-12


def _profile_rows(text, title):
    """{axis: {label: values}} of one "Differentiated ..." section of a result file."""
    sec = text[text.index(title):]
    sec = sec[:sec.index(" ***************************************************", 10)]
    out, axis = {}, None
    for line in sec.splitlines()[1:]:
        parts = line.split()
        if not parts:
            continue
        if parts[0] in ("X", "Y", "Z"):
            axis = parts[0]
            out[axis] = {}
        elif axis:
            out[axis][parts[0]] = np.array([float(v) for v in parts[1:]])
    return out


def test_cli_plane_resolved_observables(tmp_path):
    """PL_LEVEL = 3 and GETSDIFF: the two "Differentiated ..." sections of the result file, on the
    uncut lattice and cut into parts, against the checker's plane sums."""
    L = (8, 4, 6, 4)
    keys = dict(GROUP=3, ND=4, L1=L[0], L2=L[1], L3=L[2], L4=L[3], BETA=5.9, NHIT=10, INTS=1, PRNG="RANLUX3",
                RANDSERIES=54, NAV=0, ITER=3, NITER=1, PL_LEVEL=3)
    text, _, rows, _ = _run(tmp_path / "a", keys, "\tGETSDIFF = 1\n")
    text2, _, rows2, _ = _run(tmp_path / "b", keys, "\tGETSDIFF = 1\n\tN1 = {2, 6}\n\tDEVICE = {0, 0}\n")
    pl = _profile_rows(text, "Differentiated Polyakov loop data")
    sd = _profile_rows(text, "Differentiated S data")
    pl2 = _profile_rows(text2, "Differentiated Polyakov loop data")
    sd2 = _profile_rows(text2, "Differentiated S data")
    lat = ol.Lattice(3, L, 5.9, nhit=10, ints=1, rng_mode=ol.RNG_PERSITE)
    want_s = {a: [] for a in range(3)}
    want_p = {a: [] for a in range(3)}
    for _ in range(2):
        lat.sweep(1)
        for a in range(3):
            plane = np.prod(L) // L[a]
            p = lat.plaquette_profile_sums(a) / (3.0 * plane)
            want_s[a].append(0.5 * (5.9 * (1 - p[:, 0] / 3) + 5.9 * (1 - p[:, 1] / 3)))
            want_p[a].append(lat.polyakov_profile_sums(a) / (plane / L[3] * 3.0))
    for a, name in enumerate("XYZ"):
        assert len(sd[name]["S"]) == L[a]
        assert np.allclose(sd[name]["S"], np.mean(want_s[a], axis=0), rtol=1e-11)
        assert np.allclose(sd[name]["S_variance"], np.var(want_s[a], axis=0), rtol=1e-6, atol=1e-18)
        wp = np.mean(want_p[a], axis=0)
        assert np.allclose(pl[name]["PL"], wp[:, 0], rtol=1e-9, atol=1e-13)
        assert np.allclose(pl[name]["PL_im"], wp[:, 1], rtol=1e-9, atol=1e-13)
        for key in sd[name]:
            assert np.allclose(sd[name][key], sd2[name][key], rtol=1e-11, atol=1e-18)
        for key in pl[name]:
            assert np.allclose(pl[name][key], pl2[name][key], rtol=1e-9, atol=1e-13)
    # the planes of an axis average to the lattice value
    assert np.mean(sd["X"]["S"]) == pytest.approx(np.mean(rows[1:, 5]), rel=1e-11)
    lat.close()
    # the series travel in the state file: stopped after sample 2 and continued
    import shutil
    _, _, _, out = _run(tmp_path / "c", keys, "\tGETSDIFF = 1\n", args=["--save-state", "--save-every", "1e9", "--stop-at", "2"])
    qcg = re.findall(r"state file: (\S+)", out)[-1]
    (tmp_path / "d").mkdir()
    shutil.copy(qcg, tmp_path / "d" / "s.qcg")
    text3, _, rows3, _ = _run(tmp_path / "d", keys, "\tGETSDIFF = 1\n", args=["--load", "s.qcg"])
    assert np.array_equal(rows3, rows)
    sd3 = _profile_rows(text3, "Differentiated S data")
    pl3 = _profile_rows(text3, "Differentiated Polyakov loop data")
    for name in "XYZ":
        for key in sd[name]:
            assert np.array_equal(sd3[name][key], sd[name][key])
        for key in pl[name]:
            assert np.array_equal(pl3[name][key], pl[name][key])


def test_cli_fmunu_columns(tmp_path):
    """FMUNU / F0MU with the FMUNU<k> index keys: the sixteen columns behind the Polyakov loop, on
    the uncut lattice and cut into parts, against the checker (whose traces are pinned to the
    reference's CPU code by the chk_su3 fixtures)."""
    L = (6, 4, 4, 6)
    keys = dict(GROUP=3, ND=4, L1=L[0], L2=L[1], L3=L[2], L4=L[3], BETA=5.8, NHIT=10, INTS=1, PRNG="RANLUX3",
                RANDSERIES=54, NAV=0, ITER=3, NITER=1, PL_LEVEL=2, F0MU=1, FMUNU4=1, FMUNU7=1)
    text, fields, rows, _ = _run(tmp_path / "a", keys)
    _, _, rows2, _ = _run(tmp_path / "b", keys, "\tN1 = {2, 4}\n\tDEVICE = {0, 0}\n")
    assert fields[10:14] == ["Fmunu_xt_4_re", "Fmunu_xt_4_im", "Fmunu_yt_4_re", "Fmunu_yt_4_im"]
    assert fields[16:18] == ["Fmunu_xt_7_re", "Fmunu_xt_7_im"] and fields[22:26] == ["F0mu_abs_4_re", "F0mu_abs_4_im", "F0mu_abs_7_re", "F0mu_abs_7_im"]
    assert " F0MU(4, 7)" in text and " Mean Fmunu_zt_7_im" in text
    lat = ol.Lattice(3, L, 5.8, nhit=10, ints=1, rng_mode=ol.RNG_PERSITE)
    assert np.allclose(rows[0, 10:22], 0.0, atol=1e-15)            # cold start: every plaquette is the unit matrix
    for i in (1, 2):
        lat.sweep(1)
        f = lat.fmunu_sums(1, 4, 7) / np.prod(L)
        assert np.allclose(rows[i, 10:22], f, rtol=1e-10, atol=1e-14)
        assert rows[i, 22] == pytest.approx(math.sqrt(f[0] ** 2 + f[2] ** 2 + f[4] ** 2), rel=1e-10)
        assert rows[i, 25] == pytest.approx(math.sqrt(f[7] ** 2 + f[9] ** 2 + f[11] ** 2), rel=1e-10)
    lat.close()
    assert np.allclose(rows2, rows, rtol=1e-10, atol=1e-14)


def test_64x4_lattice_one_context_equals_two_slabs():
    """Size-independent property at a large size (1.6 GB of links per direction: exercises the 32-bit
    buffer offsets): the random streams are keyed by the global site, so a 64^4 SU(3) lattice in one
    context and the same lattice cut into two slabs give the same plaquette sums digit for digit;
    the y-profile adds up to the lattice sum; cold start gives exactly N."""
    import qcdgpu_amd as q
    from qcdgpu_amd import biglattice as bl
    L = (64, 64, 64, 64)
    with q.Context(group=3, L=L, beta=6.0, nhit=10, rand_series=54) as c:
        c.init_lattice()
        assert c.plaquette()[2] == 3.0
        c.sweep(1)
        one_p, one_l = c.plaquette_sums(), c.polyakov_sums()
        assert np.allclose(c.plaquette_profile_sums(1).sum(axis=0), one_p, rtol=1e-12)
        assert 2.0 < one_p.sum() / (6 * 64 ** 4) < 2.2   # first sweep from cold at beta = 6
    big = bl.make_gpu_slabs(3, L, 6.0, [32, 32], nhit=10, rand_series=54)
    try:
        big.sweep(1)
        two_p = np.sum([e.plaquette_sums() for e in big.engines], axis=0)
        two_l = np.sum([e.polyakov_sums() for e in big.engines], axis=0)
    finally:
        for e in big.engines:
            e.ctx.close()
    assert np.allclose(one_p, two_p, rtol=1e-13)
    assert np.allclose(one_l, two_l, rtol=1e-10, atol=1e-8)


def test_32x4_lattice_one_context_equals_eight_slabs():
    """The benchmark's single-GPU and eight-GPU lattices at their stated size: SU(3) 32^4 in one context, and the same
    lattice cut the way an 8-GPU run cuts it, N1 = {4,4,4,4,4,4,4,4} (all parts on the one device
    of the box).  Nx = 4 per part: the split pass updates one slice next to either face first and
    leaves an interior of two.  Same plaquette sums (1e-13), Polyakov sums and counts of random
    numbers after two sweeps, and both equal the CPU checker's fixture for this lattice."""
    import qcdgpu_amd as q
    from qcdgpu_amd import biglattice as bl
    L = (32, 32, 32, 32)
    with q.Context(group=3, L=L, beta=6.0, nhit=10, rand_series=54) as c:
        fx = ol.golden_persite32(c.rng_share)
        c.init_lattice()
        c.sweep(2)
        one_p, one_l, one_d = c.plaquette_sums(), c.polyakov_sums(), c.rng_draws()
        assert np.max(np.abs(c.plaquette() - fx["sweeps"][1]["plaquette"]) / np.abs(fx["sweeps"][1]["plaquette"])) < 1e-12
    big = bl.make_gpu_slabs(3, L, 6.0, [4] * 8, nhit=10, rand_series=54)
    try:
        assert big.boundary_depth() == 1
        big.sweep(2)
        many_p = np.sum([e.plaquette_sums() for e in big.engines], axis=0)
        many_l = np.sum([e.polyakov_sums() for e in big.engines], axis=0)
        many_d = sum(e.ctx.rng_draws() for e in big.engines)
        plq = big.plaquette()
    finally:
        for e in big.engines:
            e.ctx.close()
    assert many_d == one_d == fx["sweeps"][1]["rng_draws"]
    assert np.allclose(one_p, many_p, rtol=1e-13)
    assert np.allclose(one_l, many_l, rtol=1e-10, atol=1e-8)
    assert np.max(np.abs(plq - fx["sweeps"][1]["plaquette"]) / np.abs(fx["sweeps"][1]["plaquette"])) < 1e-12


@pytest.mark.parametrize("group,L,parts,beta,nhit", [(3, (16, 4, 4, 6), [8, 4, 4], 5.8, 4), (3, (12, 4, 4, 4), [4, 8], 1.0, 12),
                                                    (2, (8, 6, 4, 4), [2, 4, 2], 2.3, 5), (3, (8, 4, 4, 4), [3, 5], 5.8, 4),
                                                    (3, (8, 4, 6, 4), [1, 7], 5.6, 3)])
def test_ring_pass_below_the_c_abi_equals_uncut_lattice(group, L, parts, beta, nhit):
    """qcd_ring: update of every part + edge exchange as one call (copy transport, all parts on the
    one device): same configuration as the uncut lattice of the checker, link by link; parts of four
    or more slices take the split pass (boundary slices first, interior on the side lane).  Parts may
    be of odd width, one slice included: only the uncut extents have to be even."""
    import qcdgpu_amd as q
    lat = ol.Lattice(group, L, beta, nhit=nhit, ints=1, rng_mode=ol.RNG_PERSITE, seed=31)
    ctxs, x0 = [], 0
    for nx in parts:
        c = q.Context(group=group, L=L, beta=beta, nhit=nhit, rand_series=31, x0=x0, nx_local=nx, halo=1)
        c.init_lattice()
        ctxs.append(c)
        x0 += nx
    ring = q.Ring(ctxs, q.RING_COPY)
    try:
        for s in range(2):
            ring.sweep(1)
            lat.sweep(1)
            sums = np.sum([c.plaquette_sums() for c in ctxs], axis=0)
            v = float(np.prod(L))
            got = np.array([sums[0] / (3 * v), sums[1] / (3 * v), (sums[0] + sums[1]) / (6 * v)])
            assert np.max(np.abs(got - lat.plaquette()) / np.abs(lat.plaquette())) < 1e-12
        for parity in (1, 0):
            for d in range(4):
                ring.overrelax(parity, d)
        lat.overrelax_sweep(1)
        assert sum(c.rng_draws() for c in ctxs) == ol.rng_draws(lat)
        got = np.concatenate([c.download() for c in ctxs], axis=1)
        # three sweeps of a chaotic map: rounding differences between device and host arithmetic grow
        # from 1e-16 to 1e-12 at beta = 6 and to 1e-9 in the disordered phase at beta = 1
        assert np.max(np.abs(got - lat.export_table())) < (1e-10 if beta > 3 else 1e-8)
        with pytest.raises(q.QcdError):
            ring.update(2, 0)
    finally:
        ring.close()
        for c in ctxs:
            c.close()
        lat.close()


@pytest.mark.parametrize("transport", ["copy", "rccl"])
def test_ring_one_launch_pass_32_against_checker_fixture(transport):
    """A 32^4 slab that is its own neighbour on both sides: every pass goes out as ONE launch that
    takes the boundary slices first and signals from inside the kernel when they are in memory, the
    messages travel beside the rest of the launch.  Two sweeps against the CPU checker's numbers for
    the periodic 32^4 lattice (plaquette 1e-12, equal count of random numbers)."""
    import json
    import qcdgpu_amd as q
    L = (32, 32, 32, 32)
    c = q.Context(group=3, L=L, beta=6.0, nhit=10, rand_series=54, x0=0, nx_local=32, halo=1)
    fx = ol.golden_persite32(c.rng_share)
    c.init_lattice()
    ring = q.Ring([c], q.RING_COPY if transport == "copy" else q.RING_RCCL)
    try:
        for row in fx["sweeps"]:
            ring.sweep(1)
            sums = c.plaquette_sums()
            v = float(np.prod(L))
            got = np.array([sums[0] / (3 * v), sums[1] / (3 * v), (sums[0] + sums[1]) / (6 * v)])
            assert np.max(np.abs(got - row["plaquette"]) / np.abs(row["plaquette"])) < 1e-12
            assert c.rng_draws() == row["rng_draws"]
        assert ring.one_launch_passes() == 8 * len(fx["sweeps"])
    finally:
        ring.close()
        c.close()


def test_ring_one_launch_passes_two_parts_equal_uncut_lattice(monkeypatch):
    """Two slabs of 8 slices of a 16 x 32^3 lattice on the one device, passes as one launch with
    the in-kernel edge signal, against the uncut lattice in one context: same sums and draw counts
    after every sweep, same links at the end; and the same again with the signal switched off
    (two launches per pass, boundary slices first)."""
    import qcdgpu_amd as q
    L = (16, 32, 32, 32)
    kw = dict(group=3, L=L, beta=5.9, nhit=6, rand_series=77)
    one = q.Context(**kw)
    one.init_lattice()
    want = []
    for s in range(2):
        one.sweep(1)
        want.append((one.plaquette_sums(), one.polyakov_sums(), one.rng_draws()))
    want_links = one.download()
    one.close()
    for signal in (True, False):
        monkeypatch.setenv("QCDGPU_AMD_RING_ONE_LAUNCH", "1" if signal else "0")
        ctxs = [q.Context(x0=8 * k, nx_local=8, halo=1, **kw) for k in range(2)]
        for c in ctxs:
            c.init_lattice()
        ring = q.Ring(ctxs, q.RING_COPY)
        try:
            for s in range(2):
                ring.sweep(1)
                plq = np.sum([c.plaquette_sums() for c in ctxs], axis=0)
                pol = np.sum([c.polyakov_sums() for c in ctxs], axis=0)
                assert np.allclose(plq, want[s][0], rtol=1e-13)
                assert np.allclose(pol, want[s][1], rtol=1e-12, atol=1e-9)
                assert sum(c.rng_draws() for c in ctxs) == want[s][2]
            assert ring.one_launch_passes() == (32 if signal else 0)
            got = np.concatenate([c.download() for c in ctxs], axis=1)
            assert np.max(np.abs(got - want_links)) < 1e-11
        finally:
            ring.close()
            for c in ctxs:
                c.close()


@pytest.mark.parametrize("by_rank", [False, True])
def test_ring_over_rccl_on_one_gpu(by_rank):
    """The RCCL transport where a one-GPU box can run it: a ring of ONE part whose neighbours on
    both sides are itself (ncclSend / ncclRecv to its own rank inside one group), made from all
    parts of a process (ncclCommInitAll) or as rank 0 of a world of one (ncclCommInitRank with an
    identifier).  The slab with its halos must reproduce the periodic lattice of the checker."""
    import qcdgpu_amd as q
    L = (8, 4, 4, 6)
    lat = ol.Lattice(3, L, 5.7, nhit=5, ints=1, rng_mode=ol.RNG_PERSITE, seed=17)
    c = q.Context(group=3, L=L, beta=5.7, nhit=5, rand_series=17, x0=0, nx_local=L[0], halo=1)
    c.init_lattice()
    ring = q.Ring([c], rank=0, world=1, unique_id=q.Ring.unique_id()) if by_rank else q.Ring([c], q.RING_RCCL)
    try:
        for s in range(2):
            ring.sweep(1)
            lat.sweep(1)
            sums = c.plaquette_sums()
            v = float(np.prod(L))
            got = np.array([sums[0] / (3 * v), sums[1] / (3 * v), (sums[0] + sums[1]) / (6 * v)])
            assert np.max(np.abs(got - lat.plaquette()) / np.abs(lat.plaquette())) < 1e-12
        assert c.rng_draws() == ol.rng_draws(lat)
        assert np.max(np.abs(c.download() - lat.export_table())) < 1e-11
    finally:
        ring.close()
        c.close()
        lat.close()


@pytest.mark.parametrize("group,L,parts,beta", [(3, (8, 4, 4, 6), [4, 4], 5.8), (2, (12, 4, 6, 4), [4, 6, 2], 2.4)])
def test_ring_sweep_with_one_exchange_per_sweep(group, L, parts, beta):
    """qcd_ring_sweep_chb2, the reference's CHB2 build (SU(N)/suncl/suncl.cpp:4683-4729): eight passes per part on halos
    that are as old as the last sweep, then one exchange of all edges.  Emulated with the checker: one full-size
    lattice per part in which only the part's slices are updated (qo_update_pass_range reads the part's own, stale copies
    of its neighbours' boundary slices), the boundary slices copied between the lattices behind every sweep.  Not the
    uncut lattice's chain -- which the test checks as well."""
    import qcdgpu_amd as q
    nparts = len(parts)
    x0s = [sum(parts[:k]) for k in range(nparts)]
    lats = [ol.Lattice(group, L, beta, nhit=6, ints=1, rng_mode=ol.RNG_PERSITE, seed=37) for _ in parts]
    whole = ol.Lattice(group, L, beta, nhit=6, ints=1, rng_mode=ol.RNG_PERSITE, seed=37)
    ctxs = [q.Context(group=group, L=L, beta=beta, nhit=6, rand_series=37, x0=x0s[k], nx_local=parts[k], halo=1) for k in range(nparts)]
    for c in ctxs:
        c.init_lattice()
    ring = q.Ring(ctxs, q.RING_COPY)
    gslice = L[1] * L[2] * L[3]
    try:
        for s in range(3):
            ring.sweep_chb2(1)
            whole.sweep(1)
            for k, lat in enumerate(lats):
                for parity in (1, 0):
                    for d in range(4):
                        lat.update_range(parity, d, x0s[k], x0s[k] + parts[k])
            tabs = [lat.export_table() for lat in lats]
            for k in range(nparts):
                left, right = (k - 1) % nparts, (k + 1) % nparts
                xl, xr = (x0s[k] - 1) % L[0], (x0s[k] + parts[k]) % L[0]
                tabs[k][:, xl * gslice:(xl + 1) * gslice] = lats[left].export_table()[:, xl * gslice:(xl + 1) * gslice]
                tabs[k][:, xr * gslice:(xr + 1) * gslice] = lats[right].export_table()[:, xr * gslice:(xr + 1) * gslice]
            for k, lat in enumerate(lats):
                lat.import_table(tabs[k])
            got = np.concatenate([c.download() for c in ctxs], axis=1)
            want = np.concatenate([lats[k].export_table()[:, x0s[k] * gslice:(x0s[k] + parts[k]) * gslice] for k in range(nparts)], axis=1)
            assert np.max(np.abs(got - want)) < 1e-11, s
            assert sum(c.rng_draws() for c in ctxs) == sum(ol.rng_draws(lat) for lat in lats), s
        assert np.max(np.abs(got - whole.export_table())) > 1e-3      # a different chain from the first sweep's faces on
    finally:
        ring.close()
        for c in ctxs:
            c.close()
        for lat in lats + [whole]:
            lat.close()


def test_cli_chb2_key_switches_the_working_cycles(tmp_path):
    """The reference's build flag -D CHB2 as an init key of the host: thermalisation sweeps exchange edges per pass as
    always, the working cycles once per sweep (SU(N)/suncl/suncl.cpp:4683-4729) -- so the sample behind the
    thermalisation is the same with and without the key and the later ones are not."""
    keys = dict(GROUP=3, ND=4, L1=8, L2=4, L3=4, L4=6, BETA=5.8, NHIT=4, INTS=1, PRNG="RANLUX3", RANDSERIES=41, NAV=2, ITER=5, NITER=1,
                N1="{4, 4}", DEVICE="{0, 0}")
    _, fields, plain, _ = _run(tmp_path / "a", keys)
    _, _, lazy, _ = _run(tmp_path / "b", dict(keys, CHB2=1))
    i = fields.index("Plq_total")
    assert plain[0, i] == lazy[0, i] == 3.0
    assert plain[1, i] != lazy[1, i] and abs(plain[4, i] - lazy[4, i]) < 0.1      # another chain of the same physics
    _, _, uncut, _ = _run(tmp_path / "c", dict({k: v for k, v in keys.items() if k not in ("N1", "DEVICE")}, CHB2=1))
    assert np.array_equal(uncut, _run(tmp_path / "d", {k: v for k, v in keys.items() if k not in ("N1", "DEVICE")})[2])   # no parts: no effect


@pytest.mark.parametrize("threads", ["1", "0"])
def test_ring_of_eight_parts_one_host_thread_per_part(threads, monkeypatch):
    """Eight parts driven by one process, as the reference's BigLattice on an 8-GPU node (one OpenMP thread per device,
    SU(N)/suncl/suncl.cpp:639-658): the ring issues every step of a pass on one host thread per part (or, switched off,
    part after part from the caller's thread).  Same links as the uncut lattice of the checker either way."""
    import qcdgpu_amd as q
    L = (16, 8, 8, 8)
    monkeypatch.setenv("QCDGPU_AMD_RING_THREADS", threads)
    lat = ol.Lattice(3, L, 5.8, nhit=6, ints=1, rng_mode=ol.RNG_PERSITE, seed=41)
    ctxs = [q.Context(group=3, L=L, beta=5.8, nhit=6, rand_series=41, x0=2 * k, nx_local=2, halo=1) for k in range(8)]
    for c in ctxs:
        c.init_lattice()
    ring = q.Ring(ctxs, q.RING_COPY)
    try:
        for s in range(3):
            ring.sweep(1)
            lat.sweep(1)
        for parity in (1, 0):
            for d in range(4):
                ring.overrelax(parity, d)
        lat.overrelax_sweep(1)
        assert sum(c.rng_draws() for c in ctxs) == ol.rng_draws(lat)
        got = np.concatenate([c.download() for c in ctxs], axis=1)
        assert np.max(np.abs(got - lat.export_table())) < 1e-10
        with pytest.raises(q.QcdError):      # an error on a part's thread comes back as the caller's
            ring.update(0, 7)
    finally:
        ring.close()
        for c in ctxs:
            c.close()
        lat.close()


@pytest.mark.parametrize("transport,group", [("rccl", 3), ("rank", 3), ("copy", 2), ("rccl", 2)])
def test_hot_start_fills_halos_over_the_ring_transport(transport, group):
    """Hot start of a slab: the start kernel writes the owned slices, the halos come from the neighbours through
    qcd_ring_fill_halos -- the reference's UpdateEdges x 4 behind its start kernels (SU(N)/suncl/suncl.cpp:4835-4903) --
    over the ring's own transport, RCCL included (a world of one sends to itself).  Halos poisoned first; then the
    hot lattice and two sweeps against the checker's hot start of the periodic lattice."""
    import qcdgpu_amd as q
    L = (8, 4, 6, 4)
    beta = 5.7 if group == 3 else 2.4
    lat = ol.Lattice(group, L, beta, nhit=5, ints=0, rng_mode=ol.RNG_PERSITE, seed=23)
    c = q.Context(group=group, L=L, beta=beta, nhit=5, rand_series=23, x0=0, nx_local=L[0], halo=1, start_mode=q.START_HOT)
    c.init_lattice()
    if transport == "rank":
        ring = q.Ring([c], rank=0, world=1, unique_id=q.Ring.unique_id())
    else:
        ring = q.Ring([c], q.RING_RCCL if transport == "rccl" else q.RING_COPY)
    try:
        ring.fill_halos()
        assert np.max(np.abs(c.download() - lat.export_table())) < 1e-14
        sums = c.plaquette_sums()      # reads the right halo
        v = float(np.prod(L))
        got = np.array([sums[0] / (3 * v), sums[1] / (3 * v), (sums[0] + sums[1]) / (6 * v)])
        assert np.max(np.abs(got - lat.plaquette())) < 1e-13
        for s in range(2):
            ring.sweep(1)
            lat.sweep(1)
            sums = c.plaquette_sums()
            got = np.array([sums[0] / (3 * v), sums[1] / (3 * v), (sums[0] + sums[1]) / (6 * v)])
            assert np.max(np.abs(got - lat.plaquette()) / np.abs(lat.plaquette())) < 1e-11
        assert c.rng_draws() == ol.rng_draws(lat)
    finally:
        ring.close()
        c.close()
        lat.close()


QCG_CHECK = os.path.join(ROOT, "oracle", "_ref", "qcdgpu_qcg_check")


def read_with_reference_reader(qcg, f0mu=0):
    """A state file through the reference's own reader and CPU re-measurement (oracle/ref_qcg_driver.cpp:
    model::lattice_load_state twice, then lattice_check_cpu on the table it loaded): header fields, stored samples,
    re-measured values."""
    r = subprocess.run([QCG_CHECK, str(qcg), str(f0mu)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:] + r.stdout[-2000:]
    head, samples, check = {}, [], {}
    for line in r.stdout.splitlines():
        w = line.split()
        if not w:
            continue
        if w[0] == "sample":
            row, i = {}, 2
            for name, n in (("energies", 2), ("plq", 2), ("wilson", 1), ("polyakov", 2)):
                if i < len(w) and w[i] == name:
                    row[name] = [float(x) for x in w[i + 1:i + 1 + n]]
                    i += 1 + n
            samples.append(row)
        elif w[0].startswith("check_"):
            check[w[0][6:]] = float(w[1])
        elif w[0] in ("L", "domain"):
            head[w[0]] = [int(x) for x in w[1:]]
        elif len(w) == 2:
            try:
                head[w[0]] = float(w[1]) if w[0] in ("BETA", "PHI", "OMEGA") else int(w[1])
            except ValueError:
                pass
    return head, samples, check


QCG_CASES = {
    "su3_plain": dict(GROUP=3, ND=4, L1=4, L2=4, L3=4, L4=6, BETA=5.8, NHIT=4, INTS=1, PRNG="RANLUX3", RANDSERIES=41, NAV=1, ITER=5, NITER=1),
    "su3_fmunu_wilson": dict(GROUP=3, ND=4, L1=4, L2=6, L3=4, L4=4, BETA=5.6, NHIT=5, INTS=1, PRNG="RANLUX2", RANDSERIES=7, NAV=2, ITER=4, NITER=2,
                             FMUNU=1, FMUNU1=1, FMUNU6=1, GETWILSON=1, WILSONR=2, WILSONT=2, GETRETRACE=1),
    "su3_f0mu_twist_hot": dict(GROUP=3, ND=4, L1=4, L2=4, L3=6, L4=4, BETA=5.9, NHIT=4, INTS=0, PRNG="RANLUX3", RANDSERIES=19, NAV=1, ITER=4, NITER=1,
                               F0MU=1, FMUNU2=1, FMUNU6=1, PHI=0.4, OMEGA=-0.3),
    "su2_plain": dict(GROUP=2, ND=4, L1=6, L2=4, L3=4, L4=4, BETA=2.4, NHIT=6, INTS=1, PRNG="RANLUX3", RANDSERIES=23, NAV=1, ITER=5, NITER=1),
    "su2_wilson_nopl": dict(GROUP=2, ND=4, L1=4, L2=4, L3=4, L4=8, BETA=2.6, NHIT=6, INTS=2, PRNG="RANLUX1", RANDSERIES=5, NAV=0, ITER=4, NITER=1,
                            GETWILSON=1, WILSONR=1, WILSONT=3, PL_LEVEL=0, GETRETRACE=1),
}


@pytest.mark.parametrize("case", sorted(QCG_CASES))
def test_state_file_is_read_by_the_reference_reader(tmp_path, case):
    """A .qcg file written by `qcdgpu --save-state` through the REFERENCE's reader (model::lattice_load_bin_header and
    model::lattice_load_state, SU(N)/suncl/suncl.cpp:1990-2079, compiled from the reference's sources by
    `make -C oracle ref_qcg`): every header field comes back as the init file set it, the sample sections hold the
    numbers of the result table in the reference's normalisation, and the reference's CPU re-measurement of the link
    table it loaded (lattice_check_cpu) gives the sample measured on that configuration."""
    if not os.path.exists(QCG_CHECK):
        pytest.skip("oracle/_ref/qcdgpu_qcg_check is not built (make -C oracle ref_qcg needs /root/reference)")
    keys = QCG_CASES[case]
    _, fields, full, _ = _run(tmp_path / "a", keys)
    stop = keys["ITER"] - 1
    _, _, _, out = _run(tmp_path / "b", keys, args=["--save-state", "--save-every", "1e9", "--stop-at", str(stop)])
    qcg = re.findall(r"state file: (\S+)", out)[-1]
    head, samples, check = read_with_reference_reader(qcg, keys.get("F0MU", 0))
    # header
    L = [keys["L1"], keys["L2"], keys["L3"], keys["L4"]]
    assert head["L"] == L and head["domain"] == L and head["group"] == keys["GROUP"] and head["nd"] == 4
    assert head["ints"] == keys["INTS"] and head["randseries"] == keys["RANDSERIES"]
    lux = int(keys["PRNG"][-1])
    assert head["nskip"] == [24, 48, 97, 223, 389][lux] and head["generator"] == 2 + lux
    for k in ("NAV", "NITER", "ITER", "NHIT"):
        assert head[k] == keys[k], k
    assert head["BETA"] == keys["BETA"] and head["PHI"] == keys.get("PHI", 0.0) and head["OMEGA"] == keys.get("OMEGA", 0.0)
    assert head["precision"] == 2 and head["PL_level"] == keys.get("PL_LEVEL", 3)
    assert head["get_Fmunu"] == (1 if "FMUNU" in keys else 0) and head["get_wilson_loop"] == (1 if "GETWILSON" in keys else 0)
    if "GETWILSON" in keys:
        assert (head["wilson_R"], head["wilson_T"]) == (keys["WILSONR"], keys["WILSONT"])
    if "FMUNU" in keys or "F0MU" in keys:
        # FMUNU1/2/4 choose the first Gell-Mann index (3 otherwise), FMUNU5/6/7 the second (8 otherwise): suncl.cpp:424-453
        idx = [int(k[5:]) for k in keys if k.startswith("FMUNU") and k[5:].isdigit()]
        want_idx = [([i for i in idx if i in (1, 2, 4)] or [3])[-1], ([i for i in idx if i in (5, 6, 7)] or [8])[-1]]
        assert [head["Fmunu_index1"], head["Fmunu_index2"]] == want_idx
        assert head["energies_size_F"] == 7 * head["energies_size"]
    else:
        assert head["energies_size_F"] == head["energies_size"]
    assert head["ITER_counter"] == stop == len(samples) and head["NAV_counter"] == keys["NAV"]   # stopped with `stop` samples taken
    # stored samples against the uninterrupted run's result table (normalisations: SU(N)/suncl/suncl.cpp:1340-1348, 1486-1492)
    col = {f: i for i, f in enumerate(fields)}
    v = float(np.prod(L))
    n3, g = float(L[0] * L[1] * L[2]), float(keys["GROUP"])
    for i, row in enumerate(samples):
        assert np.allclose(np.array(row["energies"]) / (3 * v), full[i, [col["S_spat"], col["S_temp"]]], rtol=1e-12, atol=1e-15)
        if "plq" in row:
            assert np.allclose(np.array(row["plq"]) / (3 * v), full[i, [col["Plq_spat"], col["Plq_temp"]]], rtol=1e-12)
        if "wilson" in row:
            assert abs(row["wilson"][0] / (3 * v) - full[i, col["Wilson_loop"]]) < 1e-12
        if "polyakov" in row:
            assert np.allclose(np.array(row["polyakov"]) / (n3 * g), full[i, [col["Polyakov_loop"], col["Polyakov_loop_im"]]], rtol=1e-11, atol=1e-14)
    # the link table the reference loaded, measured by the reference: the sample taken on that configuration
    last = full[stop - 1]
    assert abs(check["Plq_spat"] - last[col["Plq_spat"]]) < 1e-11 and abs(check["Plq_temp"] - last[col["Plq_temp"]]) < 1e-11
    assert abs(check["S_spat"] - last[col["S_spat"]]) < 1e-10 and abs(check["S_temp"] - last[col["S_temp"]]) < 1e-10
    if head["PL_level"] > 0:
        assert abs(check["Polyakov_loop"] - last[col["Polyakov_loop"]]) < 1e-11
        assert abs(check["Polyakov_loop_im"] - last[col["Polyakov_loop_im"]]) < 1e-11
    if "GETWILSON" in keys:
        assert abs(check["Wilson_loop"] - last[col["Wilson_loop"]]) < 1e-11
    # kept for the record under gpurun_out/ (scratch): the files and what the reference's reader said about them
    keep = os.path.join(ROOT, "gpurun_out", "qcg")
    os.makedirs(keep, exist_ok=True)
    import shutil
    shutil.copy(qcg, os.path.join(keep, case + ".qcg"))
    with open(os.path.join(keep, case + ".json"), "w") as f:
        json.dump({"init": keys, "header": head, "samples": samples, "reference_remeasurement": check,
                   "result_table_fields": fields, "result_table_row_at_save": [float(x) for x in last]}, f, indent=1)


def test_cli_state_file_with_fmunu_sections(tmp_path):
    """The state file with F0MU = 1 and no plaquette average: the reference writes its
    lattice_energies_plq section whenever plaquettes OR either field tensor is measured
    (SU(N)/suncl/suncl.cpp:1892, 2063), seven blocks long with the tensor (MODEL_energies_size),
    and the link table behind it.  Header words, the table's offset, an exact resume, and a file
    that is too short for the switches of the run that loads it."""
    import shutil
    import struct
    keys = dict(GROUP=3, ND=4, L1=4, L2=4, L3=4, L4=6, BETA=5.8, NHIT=4, INTS=1, PRNG="RANLUX3",
                RANDSERIES=41, NAV=1, ITER=5, NITER=1, F0MU=1, FMUNU2=1, FMUNU6=1)
    _, fields, full, _ = _run(tmp_path / "a", keys)
    assert any(f.startswith("Fmunu_xt_2") for f in fields) and any(f.startswith("Fmunu_xt_6") for f in fields)
    _, _, none, out = _run(tmp_path / "b", keys, args=["--save-state", "--save-every", "1e9", "--stop-at", "3"])
    qcg = re.findall(r"state file: (\S+)", out)[-1]
    raw = open(qcg, "rb").read()
    head = struct.unpack("<64I", raw[:256])
    assert head[26:29] == (2, 6, 0)            # Fmunu_index1, Fmunu_index2, get_Fmunu (the E field is F0MU: not a header word)
    es = head[33]
    assert head[32:36] == (0, es, 7 * es, es)  # measurement, energies, energies_plq (7 blocks), Polyakov loop
    sites, rs = 4 * 4 * 4 * 6, 512
    off = 256 + 16 * es + 16 * 7 * es + 16 * es          # header, energies, energies_plq, Polyakov loop
    table = np.frombuffer(raw[off:off + 12 * rs * 32], dtype=np.float64).reshape(12, rs, 4)
    u = table[0:4, :sites, :3] + 1j * table[4:8, :sites, :3]
    assert np.max(np.abs(np.sum(np.abs(u) ** 2, axis=2) - 1)) < 1e-14
    assert raw[off + 12 * rs * 32:off + 12 * rs * 32 + 7] == b"QCDRLX1"
    (tmp_path / "c").mkdir()
    shutil.copy(qcg, tmp_path / "c" / "s.qcg")
    _, _, resumed, _ = _run(tmp_path / "c", keys, args=["--load", "s.qcg"])
    assert np.array_equal(resumed, full)
    # a file cut inside its link table is refused, not read at a wrong offset
    (tmp_path / "d").mkdir()
    with open(tmp_path / "d" / "s.qcg", "wb") as f:
        f.write(raw[:off + 1000])
    with pytest.raises(AssertionError):
        _run(tmp_path / "d", keys, args=["--load", "s.qcg"])
