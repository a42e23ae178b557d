"""Host code under AddressSanitizer + UndefinedBehaviorSanitizer, on the CPU (never on a GPU box: `-m "not gpu"`).

  * the CPU checker: `make -C oracle asan` builds it with both sanitizers and walks through every entry point;
  * the C++ host (init-file tokenizer, run loop, result writer, state-file writer and reader of
    qcdgpu_amd/csrc/host/model.cpp) linked against a host-memory stand-in for the C-ABI (tests/host_stub/stub_capi.cpp),
    fed with well-formed and malformed init files and with truncated and corrupted state files.
A sanitizer finding ends the process with a report and a non-zero status ("runtime error", "ERROR: AddressSanitizer")."""
import os
import shutil
import struct
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = ["-O1", "-g", "-std=c++17", "-Wall", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
       "-fno-omit-frame-pointer", "-static-libasan", "-static-libubsan"]


def _findings(text):
    return [ln for ln in text.splitlines() if "runtime error" in ln or "AddressSanitizer" in ln or "LeakSanitizer" in ln]


def test_checker_under_sanitizers():
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and not _findings(r.stdout + r.stderr), (r.stdout + r.stderr)[-3000:]
    assert "O(N) precision 1" in r.stdout
    # the command-line driver on a reference-format init file
    init = os.path.join(ROOT, "oracle", "_asan_init.dat")
    with open(init, "w") as f:
        f.write("\tGROUP = 2\n\tLS = 4\n\tLT = 4\n\tBETA = 2.5\n\tNHIT = 10\n\tINTS = 1\n\tNAV = 1\n\tITER = 3\n\tNITER = 1\n"
                "\tPRNG = RANLUX3\n\tRANDSERIES = 54\n# a comment line = with an equals sign\n\n\tFst=0\n")
    try:
        r = subprocess.run([os.path.join(ROOT, "oracle", "qcd_oracle_cli_asan"), init], capture_output=True, text=True, timeout=300)
    finally:
        os.remove(init)
    assert r.returncode == 0 and not _findings(r.stdout + r.stderr), (r.stdout + r.stderr)[-3000:]


@pytest.fixture(scope="module")
def host(tmp_path_factory):
    d = tmp_path_factory.mktemp("host_asan")
    exe = str(d / "qcdgpu_asan")
    src = [os.path.join(ROOT, "qcdgpu_amd", "csrc", "host", f) for f in ("model.cpp", "qcdgpu_main.cpp")]
    r = subprocess.run(["g++"] + SAN + src + [os.path.join(ROOT, "tests", "host_stub", "stub_capi.cpp"), "-o", exe],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


BASE = dict(GROUP=3, ND=4, LS=4, LT=6, BETA=5.8, NHIT=4, INTS=1, NAV=1, ITER=5, NITER=1, PRNG="RANLUX3", RANDSERIES=41)


def _run(exe, d, text, args=()):
    d.mkdir(exist_ok=True)
    with open(d / "init.dat", "w") as f:
        f.write(text)
        f.write("\tOUTPUTPATH = %s/\n" % d)
    r = subprocess.run([exe, str(d / "init.dat")] + list(args), capture_output=True, text=True, timeout=300)
    out = r.stdout + r.stderr
    assert not _findings(out), out[-3000:]
    assert r.returncode in (0, 1, 2), (r.returncode, out[-2000:])     # a refusal is a status, not a signal
    return r.returncode, out


def _keys(d):
    return "".join("\t%s = %s\n" % kv for kv in d.items())


def test_host_runs_and_resumes_clean(host, tmp_path):
    for extra in ({}, dict(GETWILSON=1, WILSONR=2, WILSONT=2, FMUNU=1, FMUNU2=1), dict(F0MU=1, PL_LEVEL=0), dict(GROUP=2, GETSDIFF=1),
                  dict(L1=8, N1="{4, 2, 2}", DEVICE="{0, 0, 0}"), dict(L1=8, N1="{4, 4}", DEVICE="{0, 0}", CHB2=1, NOVERRELAX=0)):
        keys = dict(BASE, **extra)
        name = "run%d" % len(extra)
        rc, out = _run(host, tmp_path / name, _keys(keys))
        assert rc == 0 and "result file:" in out, out[-2000:]
        rc, out = _run(host, tmp_path / (name + "s"), _keys(keys), ["--save-state", "--save-every", "1e9", "--stop-at", "3"])
        assert rc == 0, out[-2000:]
        qcg = [w for w in out.split() if w.endswith(".qcg")][-1]
        shutil.copy(qcg, tmp_path / (name + "s") / "s.qcg")
        rc, out = _run(host, tmp_path / (name + "s"), _keys(keys), ["--load", "s.qcg"])
        assert rc == 0 and "result file:" in out, out[-2000:]


def test_host_malformed_init_files(host, tmp_path):
    long_line = "\tBETA = " + "9" * 700 + "\n"
    cases = [
        "",                                                     # empty file: the defaults (a 32^4 run is too big for the stub: refused)
        "GROUP\nND\n=\n= =\n==\n\t\t\n#\n# = #\n",               # no keys at all
        _keys(BASE) + long_line,                                # a line longer than the tokenizer's buffer
        _keys(BASE) + "\t" + "K" * 400 + " = 1\n",               # a key longer than the buffer
        _keys(BASE) + "\tN1 = {2, 2\n",                          # unterminated vector
        _keys(BASE) + "\tN1 = {" + "1, " * 200 + "}\n",          # a vector longer than the line buffer
        _keys(BASE) + "\tN1 = {}\n\tDEVICE = {,,,}\n",
        _keys(dict(BASE, ITER=0)), _keys(dict(BASE, ITER=-3)), _keys(dict(BASE, NITER=-1, NAV=-5)),
        _keys(dict(BASE, LS=-4)), _keys(dict(BASE, LS=0)), _keys(dict(BASE, L2=3)), _keys(dict(BASE, GROUP=7)), _keys(dict(BASE, ND=9, LT=4)),
        _keys(dict(BASE, NHIT=0)), _keys(dict(BASE, BETA="abc")), _keys(dict(BASE, PRNG="NOSUCH", RANDSERIES=-1)),
        _keys(dict(BASE, WILSONR=-2, GETWILSON=1)), _keys(dict(BASE, PL_LEVEL=99)), _keys(dict(BASE, RNGSHARE=3)), _keys(dict(BASE, RNGSHARE=4)),
        _keys(BASE).replace("\n", "\r\n"),                      # DOS line ends
        _keys(BASE) + "\x00\x01\xff binary = \x7f\n",
    ]
    for i, text in enumerate(cases):
        _run(host, tmp_path / ("m%d" % i), text)
        _run(host, tmp_path / ("m%d" % i), text, ["--parse-only"])


def test_host_truncated_and_corrupted_state_files(host, tmp_path):
    keys = dict(BASE, GETWILSON=1, WILSONR=2, WILSONT=2, GETSDIFF=1)
    rc, out = _run(host, tmp_path / "w", _keys(keys), ["--save-state", "--save-every", "1e9", "--stop-at", "3"])
    assert rc == 0
    qcg = [w for w in out.split() if w.endswith(".qcg")][-1]
    raw = open(qcg, "rb").read()
    d = tmp_path / "t"
    d.mkdir()

    def load(blob):
        with open(d / "s.qcg", "wb") as f:
            f.write(blob)
        return _run(host, d, _keys(keys), ["--load", "s.qcg"])

    assert load(raw)[0] == 0
    for cut in (0, 3, 100, 255, 256, 257, 5000, len(raw) // 2, len(raw) - 4000, len(raw) - 9, len(raw) - 1):
        load(raw[:cut])                                         # any outcome but a sanitizer finding or a signal
    head = list(struct.unpack("<64I", raw[:256]))
    for word, value in ((13, 0), (13, 0x7fffffff), (14, 0xffffffff), (33, 0xfffffff0), (34, 0), (35, 0x40000000), (36, 9), (37, 0),
                        (38, 0), (38, 0x7fffffff), (41, 0xffffffff), (7, 99), (63, 3), (10, 0xffffffff)):
        h = list(head)
        h[word] = value
        load(struct.pack("<64I", *h) + raw[256:])
    # garbage behind the link table where this engine's generator states start
    at = raw.index(b"QCDRLX1")
    load(raw[:at + 8] + b"\xff" * 16 + raw[at + 24:])
    load(raw[:at] + b"QCDRLX1\x00" + raw[at + 8:at + 40])
    at2 = raw.index(b"QCDPRF1")
    load(raw[:at2 + 16] + b"\xff" * 8 + raw[at2 + 24:])         # an absurd series length
