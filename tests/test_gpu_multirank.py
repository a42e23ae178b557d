"""The N > 1 path on the GPU: two (and four) ranks as child processes on the one device of the
box, through make_rank_slab -> GpuSlabEngine + TorchDistComm -- the code an N-GPU bench run takes,
with gloo staging in place of RCCL -- against the uncut lattice of the CPU checker: plaquette and
Polyakov loop per sweep, the count of random numbers drawn, and the concatenated link table."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from tests import oracle_lib as ol

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(world, cfg, tmp_path):
    cfg = dict(cfg, dir=str(tmp_path))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "gpu_slab_worker.py"), json.dumps(cfg)]
    env = dict(os.environ, OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    with open(tmp_path / "rows.json") as f:
        doc = json.load(f)
    links = np.concatenate([np.load(tmp_path / ("links_%d.npy" % k)) for k in range(world)], axis=1)
    return doc, links


@pytest.mark.parametrize("world,cfg", [
    (2, dict(group=3, L=[16, 4, 4, 6], beta=5.8, nhit=4, seed=29, ints=1, sweeps=2)),
    (2, dict(group=3, L=[16, 4, 4, 6], beta=1.0, nhit=12, seed=29, ints=1, sweeps=2)),   # low beta: most links run past their window of random numbers
    (4, dict(group=2, L=[16, 4, 6, 4], beta=2.3, nhit=5, seed=7, ints=1, sweeps=2)),     # Nx = 4 per rank
    (2, dict(group=3, L=[8, 4, 4, 4], beta=5.9, nhit=4, seed=3, ints=0, sweeps=1)),      # hot start: halos by exchange
])
def test_gpu_ranks_reproduce_uncut_lattice(world, cfg, tmp_path):
    doc, links = _run(world, cfg, tmp_path)
    mode = ol.RNG_PERSITE
    lat = ol.Lattice(cfg["group"], cfg["L"], cfg["beta"], nhit=cfg["nhit"], ints=cfg["ints"], rng_mode=mode,
                     seed=cfg["seed"])
    for row in doc["rows"]:
        lat.sweep(1)
        want = np.concatenate([lat.plaquette(), lat.polyakov()])
        got = np.array(row)
        assert np.max(np.abs(got[:3] - want[:3]) / np.abs(want[:3])) < 1e-12
        assert np.allclose(got[3:], want[3:], rtol=1e-10, atol=1e-13)
    assert doc["draws"] == ol.rng_draws(lat)
    assert np.max(np.abs(links - lat.export_table())) < 1e-11
    assert doc["split"] and doc["boundary_depth"] == (1 if cfg["L"][0] // world >= 4 else 0)
    lat.close()


def test_rank_ring_world_of_one(tmp_path):
    """The path an N-GPU bench run takes by default (make_rank_ring: torch.distributed over gloo hands
    the RCCL identifier round, the C library issues passes and RCCL send/receive) with the one rank a
    one-GPU box allows: the slab's neighbours on both sides are itself."""
    cfg = dict(group=3, L=[8, 4, 4, 6], beta=5.8, nhit=4, seed=29, ints=1, sweeps=2, ring=True)
    doc, links = _run(1, cfg, tmp_path)
    lat = ol.Lattice(3, cfg["L"], cfg["beta"], nhit=4, ints=1, rng_mode=ol.RNG_PERSITE, seed=29)
    for row in doc["rows"]:
        lat.sweep(1)
        want = np.concatenate([lat.plaquette(), lat.polyakov()])
        assert np.max(np.abs(np.array(row)[:3] - want[:3]) / np.abs(want[:3])) < 1e-12
        assert np.allclose(np.array(row)[3:], want[3:], rtol=1e-10, atol=1e-13)
    assert doc["draws"] == ol.rng_draws(lat)
    assert np.max(np.abs(links - lat.export_table())) < 1e-11
    lat.close()


def test_bench_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` without a launcher: the process starts two ranks as children and
    prints rank 0's JSON line (both ranks on device 0, gloo staging: the one-GPU rehearsal switches)."""
    env = dict(os.environ, QCDGPU_BENCH_SHARE_DEVICE="1", QCDGPU_BENCH_BACKEND="gloo", OMP_NUM_THREADS="1",
               QCDGPU_BENCH_FORCE_CROSSCHECK="1")   # ... and the cross-check of the first sweeps that a multi-GPU run makes
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "2",
                        "--brackets", "2", "--L", "16", "--no-cpu"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    doc = json.loads(line)
    assert doc["n_gpus"] == 2 and doc["config"]["lattice"] == [32, 16, 16, 16] and doc["scaling"] == "weak"
    assert doc["mean_plaquette"]["parity_check"]["ok"], doc["mean_plaquette"]["parity_check"]
    assert doc["value"] > 0 and doc["roofline"]["launches"] >= 16
    # 6 sweeps of 32 x 16^3 from a cold start at beta = 6: the plaquette has left 3 and is still above its equilibrium value
    assert 1.7 < doc["mean_plaquette"]["Plq_total"] < 2.2


def test_bench_strong_scaling_cuts_one_lattice(tmp_path):
    """`bench.py --gpus 2 --scaling strong`: ONE 32^4 lattice cut into two slabs of 16 slices (the reference's
    N1 = {16, 16}), both ranks on the box's one device.  The line says "strong", counts the link updates of one
    lattice, and its first two sweeps equal the CPU checker's fixture for the uncut 32^4 lattice."""
    env = dict(os.environ, QCDGPU_BENCH_SHARE_DEVICE="1", QCDGPU_BENCH_BACKEND="gloo", OMP_NUM_THREADS="1")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "2",
                        "--brackets", "1", "--L", "32", "--scaling", "strong", "--no-cpu"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    doc = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert doc["scaling"] == "strong" and doc["n_gpus"] == 2 and doc["config"]["lattice"] == [32, 32, 32, 32]
    assert doc["config"]["rng_share"] == 2
    assert doc["mean_plaquette"]["parity_check"]["ok"], doc["mean_plaquette"]["parity_check"]
    assert "share2" in doc["mean_plaquette"]["parity_check"]["against"]
    assert abs(doc["value"] * doc["ms_per_step"] * 1e-3 - 4 * 32 ** 4) < 1.0      # one lattice's link updates per sweep
    assert doc["roofline"]["algorithmic_bytes_per_launch"] == 864 * 32 ** 4 / 4   # half the lattice per GPU, half its sites per pass


def test_bench_falls_back_when_the_rccl_ring_does_not_come_up(tmp_path):
    """Two ranks told to use the C library's RCCL ring on ONE device: RCCL refuses two ranks on a
    device, every rank reports that, the ranks agree (over gloo) and the run goes on with the Python
    driver's exchange -- the same plaquette as the run that used that exchange from the start."""
    docs = []
    for transport in ("ring", "torch"):
        env = dict(os.environ, QCDGPU_BENCH_SHARE_DEVICE="1", QCDGPU_BENCH_BACKEND="gloo", QCDGPU_BENCH_TRANSPORT=transport,
                   OMP_NUM_THREADS="1")
        env.pop("WORLD_SIZE", None)
        env.pop("RANK", None)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
                            "--brackets", "1", "--L", "8", "--no-cpu"], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, r.stderr[-3000:]
        if transport == "ring":
            assert "did not come up on every rank" in r.stderr
        docs.append(json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]))
    assert "torch.distributed" in docs[0]["config"]["halo_transport"]
    assert docs[0]["mean_plaquette"]["Plq_total"] == docs[1]["mean_plaquette"]["Plq_total"]
