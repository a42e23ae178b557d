"""Host logic without a GPU: the init.dat reader and key table of the C++ run API, its
statistics, and that the C-ABI library loads and exports every declared symbol."""
import ctypes as C
import json
import os
import re
import subprocess

import numpy as np
import pytest

from tests import oracle_lib as ol

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
QCDGPU = os.path.join(ROOT, "qcdgpu_amd", "bin", "qcdgpu")

BIGLAT_INIT = ("\tPLATFORM = {0, 0}\n\tDEVICE= {0, 0}\n\tFINISHPATH=./\n\tOUTPUTPATH=/home/science/scalar/\n"
               "\tGROUP = 3\n\tND = 4\n\tLT = 4\n\tLS = 6\n\tL1 = 12\n\tN1 = {8, 4}\n\tITER = 2\n\tNITER = 1\n"
               "\tNHIT = 10\n\tBETA = 6.3690000000000e+00\n\tNAV = 0\n\tINTS = 2\n\tPRNG = RANLUX4\n"
               "\tRANDSERIES = {1420193282, 45}\n\tWILSONR = 10\n\tWILSONT = 3\n"
               "\tOMEGA = 0.0000000000000e+00\n\tPHI = 0.0000000000000e+00\n\tFst = 0\n\tFMUNU = 1\n")

SINGLE_INIT = ("\tPLATFORM = 0\n\tDEVICE= 0\n\tFINISHPATH=./\n\tOUTPUTPATH=/home/science/scalar/\n\tGROUP = 2\n"
               "\tND = 4\n\tLT = 16\n\tLS = 16\n\tL1 = 16\n\tN1 = 16\n\tITER = 50   # samples\n\tNITER = 10\n"
               "\tNHIT = 10\n\tBETA = 3.5000000000000e+00\n\tNAV = 50\n\tINTS = 1\n\tPRNG = RANLUX3\n"
               "\tRANDSERIES = 0\n\tGETWILSON = 1\n\tWILSONR = 10\n\tWILSONT = 3\n\tPL_LEVEL = 3\n"
               "\tOMEGA = 0.0000000000000e+00\n\tPHI = 3.000000000000e+00\n\tFMUNU = 1\n\tFst = 0\n")


def _parse(tmp_path, text):
    f = tmp_path / "init.dat"
    f.write_text(text)
    out = subprocess.run([QCDGPU, "--parse-only", str(f)], check=True, capture_output=True, text=True)
    return json.loads(out.stdout)


def test_host_binary_exists():
    assert os.path.exists(QCDGPU), "build it: python -c 'import __graft_entry__ as g; g.build()'"


def test_init_file_single_device_keys(tmp_path):
    p = _parse(tmp_path, SINGLE_INIT)
    assert p["GROUP"] == 2 and p["PREFIX"] == "su2-" and p["ND"] == 4
    assert p["L"] == [16, 16, 16, 16]
    assert (p["ITER"], p["NITER"], p["NHIT"], p["NAV"], p["INTS"]) == (50, 10, 10, 50, 1)
    assert p["BETA"] == 3.5 and p["PHI"] == 3.0 and p["OMEGA"] == 0.0
    assert p["LUX"] == 3 and p["RANDSERIES"] == 0
    assert p["OUTPUTPATH"] == "/home/science/scalar/" and p["FINISHPATH"] == "./"
    assert p["GETWILSON"] == 1 and p["WILSONR"] == 10 and p["WILSONT"] == 3 and p["PL_LEVEL"] == 3
    assert p["NPARTS"] == 1        # "N1 = 16" is a scalar: one part


def test_init_file_biglattice_vectors(tmp_path):
    p = _parse(tmp_path, BIGLAT_INIT)
    assert p["GROUP"] == 3 and p["L"] == [12, 6, 6, 4]      # LS then L1 overrides Nx; LT -> L[ND-1]
    assert p["NPARTS"] == 2 and p["N1"] == [8, 4] and p["DEVICES"] == [0, 0]
    assert p["LUX"] == 4 and p["RANDSERIES"] == 1420193282   # first entry of the vector
    assert p["INTS"] == 2 and abs(p["BETA"] - 6.369) < 1e-15


def test_init_file_order_and_comments(tmp_path):
    # LT writes index ND-1, so ND must come first (reference behaviour); '#' starts a comment;
    # unknown keys and lines without '=' are ignored
    p = _parse(tmp_path, "ND = 4\nLS = 8\nLT = 6 # time extent\nnonsense line\nFOO = 7\n  BETA   =   5.75  \n")
    assert p["L"] == [8, 8, 8, 6] and p["BETA"] == 5.75


@pytest.mark.skipif(not os.path.isdir("/root/reference/SU(N)"), reason="reference not mounted")
def test_reads_the_reference_sample_files_unchanged(tmp_path):
    a = json.loads(subprocess.run([QCDGPU, "--parse-only", "/root/reference/SU(N)/init.dat"], check=True,
                                  capture_output=True, text=True).stdout)
    assert a["GROUP"] == 2 and a["L"] == [16, 16, 16, 16] and a["ITER"] == 50 and a["BETA"] == 3.5
    b = json.loads(subprocess.run([QCDGPU, "--parse-only", "/root/reference/SU(N)/init_2.dat"], check=True,
                                  capture_output=True, text=True).stdout)
    assert b["N1"] == [8, 4] and b["L"] == [12, 6, 6, 4] and b["LUX"] == 4


def test_statistics_convention_matches_checker():
    rng = np.random.default_rng(3)
    data = 1.3 + 1e-3 * rng.standard_normal(40)
    out = subprocess.run([QCDGPU, "--stats"] + ["%.17g" % v for v in data], check=True,
                         capture_output=True, text=True)
    got = json.loads(out.stdout)
    m, v = ol.estimates(data)            # sample 0 excluded, biased variance
    assert abs(got["mean"] - m) < 1e-15 and abs(got["variance"] - v) < 1e-18
    assert got["last"] == data[-1]


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "qcdgpu_amd.h")).read()
    declared = set(re.findall(r"\b(qcd_[a-z_0-9]+)\s*\(", hdr))
    declared -= {"qcd_ctx", "qcd_params", "qcd_timing"}
    for so in ("libqcdgpu_amd.so", "libqcdgpu_amd_f32.so"):      # the double and the single precision build
        lib = C.CDLL(os.path.join(ROOT, "qcdgpu_amd", "lib", so))
        for name in sorted(declared):
            assert hasattr(lib, name), (so, name)
        assert lib.qcd_precision() == (2 if so == "libqcdgpu_amd.so" else 1)
    from qcdgpu_amd import capi
    assert declared == set(capi.EXPORTS)


def test_no_device_is_an_error_code_not_a_crash():
    import qcdgpu_amd as q
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(q.QcdError):
        q.Context(group=3, L=(4, 4, 4, 4), beta=6.0)


def test_odd_extents_where_they_are_refused():
    """Ny must be even (the reference pairs sites 2h, 2h+1 in y).  Odd Nx, Nz, Nt: one uncut SU(N) lattice is served
    (tests/test_gpu_parity.py), x-slabs and the scalar model's engine say that they are not instead of computing
    nonsense -- before any device work."""
    import qcdgpu_amd as q
    with pytest.raises(q.QcdError, match="even"):
        q.Context(group=3, L=(4, 5, 4, 4), beta=6.0)
    for L in ((5, 4, 4, 4), (4, 4, 3, 4), (4, 4, 4, 7)):
        with pytest.raises(q.QcdError, match="odd"):
            q.Context(group=3, L=L, beta=6.0, x0=0, nx_local=2, halo=1)
        with pytest.raises(q.QcdError, match="odd|even"):
            q.OnContext(L=L)


def test_packed_generator_state_round_trips():
    """The device keeps a RANLUX state as 19 words (24-bit words four in three); the checkpoint
    format has 25.  All-ones, all-zero and 100000 pseudo-random states survive pack + unpack."""
    import qcdgpu_amd as q
    assert q.capi.host_rlx_pack_check(12345, 100000) == 0


def test_ranlux_batch_formulation_equals_reference_generator():
    import qcdgpu_amd as q
    for lux, nskip in enumerate((24, 48, 97, 223, 389)):
        for stream in (0, 1, 11):
            n = 20000
            got = q.host_ranlux_stream(54, stream, lux, n)
            st = ol.RanluxState()
            ol.lib.qo_ranlux_init(C.byref(st), ol.lib.qo_stream_jseed(54, stream), nskip)
            want = np.array([ol.lib.qo_ranlux_next(C.byref(st)) for _ in range(n)], dtype=np.float32)
            assert np.array_equal(got, want), (lux, stream)
            assert (want < 2.0 ** -12).sum() > 0      # the tiny-value correction was exercised


def test_gid_tables_match_checker():
    import qcdgpu_amd as q
    sites = 4 * 6 * 8 * 4
    lat = ol.Lattice(3, (4, 6, 8, 4), 6.0, ints=2, rng_mode=ol.RNG_GID)
    t = lat.export_table()
    for gid in (0, 1, 77, sites - 1):
        for d in range(4):
            got = q.host_gid_link(3, gid, d, sites)
            want = np.concatenate([t[0 * 4 + d, gid], t[1 * 4 + d, gid], t[2 * 4 + d, gid]])
            assert np.max(np.abs(got - want)) < 4e-16
    r = q.host_gid_randoms(1234, 10, 65536)
    assert np.all((r >= 0) & (r <= 1))
    lat.close()


QCG_DIR = os.path.join(ROOT, "tests", "golden", "qcg")
QCG_CHECK = os.path.join(ROOT, "oracle", "_ref", "qcdgpu_qcg_check")


@pytest.mark.parametrize("case", ["su3_plain", "su3_fmunu_wilson", "su3_f0mu_twist_hot", "su2_plain", "su2_wilson_nopl"])
def test_committed_state_files_against_the_reference_reader(case):
    """State files this engine's host wrote on the GPU box (tests/test_gpu_host.py keeps them), committed with what
    the reference's own reader and CPU re-measurement said about them (oracle/ref_qcg_driver.cpp over
    model::lattice_load_state, SU(N)/suncl/suncl.cpp:1990-2079).  Without a GPU: the header words are where the
    reference's lattice_make_bin_header puts them (:1939-1988), the sections have the lengths the header announces, and
    -- where the reference reader is built -- it reads the committed file to the committed numbers again."""
    import json
    import struct
    import subprocess
    doc = json.load(open(os.path.join(QCG_DIR, case + ".json")))
    raw = open(os.path.join(QCG_DIR, case + ".qcg"), "rb").read()
    head = struct.unpack("<64I", raw[:256])
    want, keys = doc["header"], doc["init"]
    assert raw[:6] == b"QCDGPU" and raw[8:11] == b"1.5"
    assert head[5] == want["ints"] and head[6] == want["randseries"] and head[7] == want["generator"] and head[8] == want["nskip"]
    assert head[10:16] == (want["NAV"], want["NAV_counter"], want["NITER"], want["ITER"], want["ITER_counter"], want["NHIT"])
    assert struct.unpack("<d", raw[19 * 4:21 * 4])[0] == keys["BETA"]
    assert head[36:38] == (want["group"], 4) and list(head[38:42]) == want["L"] == list(head[42:46])
    assert head[63] in (1, 2, 4)          # sites per random stream, in a word the reference leaves zero and does not read
    es, esf, pls = head[33], head[34], head[35]
    sites = int(np.prod(want["L"]))
    rows, rs = (12 if want["group"] == 3 else 4), (1 + (sites - 1) // 256) * 256
    plq = want["get_plaquettes_avr"] or want["get_Fmunu"] or keys.get("F0MU", 0)
    off = 256 + 16 * es + (16 * esf if plq else 0) + (8 * es if want["get_wilson_loop"] else 0) + (16 * pls if want["PL_level"] > 0 else 0)
    assert raw[off + rows * rs * 32:off + rows * rs * 32 + 7] == b"QCDRLX1"      # this engine's generator states follow the reference's file
    table = np.frombuffer(raw[off:off + rows * rs * 32], dtype=np.float64).reshape(rows, rs, 4)
    if want["group"] == 3:
        u = table[0:4, :sites, :3] + 1j * table[4:8, :sites, :3]       # first rows of the links (su3_matrix_memory.cl:42-176)
        assert np.max(np.abs(np.sum(np.abs(u) ** 2, axis=2) - 1)) < 1e-13
    if not os.path.exists(QCG_CHECK):
        pytest.skip("oracle/_ref/qcdgpu_qcg_check is not built here")
    r = subprocess.run([QCG_CHECK, os.path.join(QCG_DIR, case + ".qcg"), str(keys.get("F0MU", 0))], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    got = {ln.split()[0]: ln.split()[1:] for ln in r.stdout.splitlines() if ln.startswith("check_")}
    for name, value in doc["reference_remeasurement"].items():
        assert abs(float(got["check_" + name][0]) - value) <= 1e-14 * max(1.0, abs(value)), name
    # ... and the re-measured plaquette is the result table's row of the configuration the file holds
    f = doc["result_table_fields"]
    row = doc["result_table_row_at_save"]
    assert abs(doc["reference_remeasurement"]["Plq_spat"] - row[f.index("Plq_spat")]) < 1e-11
    assert abs(doc["reference_remeasurement"]["Plq_temp"] - row[f.index("Plq_temp")]) < 1e-11
