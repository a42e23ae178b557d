"""The N > 1 path on CPU: world_size-2 (and 3) gloo runs of the BigLattice driver must reproduce
the single-process run of the checker -- same RANLUX streams (keyed by global site), same
arithmetic, so the only difference allowed is the order of the final sums."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from tests import oracle_lib as ol

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(world, cfg, tmp_path):
    cfg = dict(cfg, out=str(tmp_path / "rows.json"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "mp_slab_worker.py"), json.dumps(cfg)]
    env = dict(os.environ, OMP_NUM_THREADS="1", QCDGPU_AMD_NO_TORCH="")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    return np.array(json.load(open(cfg["out"])))


def _single(cfg):
    mode = ol.RNG_GID if cfg["ints"] == 2 else ol.RNG_PERSITE
    lat = ol.Lattice(cfg["group"], cfg["L"], cfg["beta"], nhit=cfg["nhit"], ints=cfg["ints"], rng_mode=mode,
                     seed=cfg["seed"], share=cfg.get("share", 1))
    rows = [np.concatenate([lat.plaquette(), lat.action(), lat.polyakov()])]
    for _ in range(cfg["sweeps"]):
        lat.sweep(1)
        rows.append(np.concatenate([lat.plaquette(), lat.action(), lat.polyakov()]))
    lat.close()
    return np.array(rows)


@pytest.mark.parametrize("world,cfg", [
    (2, dict(group=3, L=[8, 4, 4, 4], beta=6.0, nhit=10, seed=54, ints=1, sweeps=2)),
    (2, dict(group=2, L=[4, 6, 4, 6], beta=2.3, nhit=5, seed=7, ints=1, sweeps=2)),
    (2, dict(group=3, L=[4, 4, 4, 4], beta=6.0, nhit=10, seed=54, ints=2, sweeps=2)),
    (3, dict(group=3, L=[8, 4, 4, 4], beta=5.5, nhit=3, seed=9, ints=1, sweeps=1, parts=[2, 4, 2])),
    (2, dict(group=3, L=[8, 4, 4, 4], beta=5.8, nhit=4, seed=21, ints=0, sweeps=1)),   # hot start
    (2, dict(group=2, L=[16, 4, 4, 4], beta=2.3, nhit=5, seed=3, ints=1, sweeps=2)),   # boundary slices two deep
    # random streams shared by 2 and by 4 sites of one parity: the stream of a site does not depend on the cut
    (2, dict(group=3, L=[8, 4, 4, 4], beta=6.0, nhit=10, seed=54, ints=1, sweeps=2, share=2)),
    (3, dict(group=2, L=[8, 4, 4, 8], beta=2.3, nhit=5, seed=11, ints=0, sweeps=1, parts=[2, 4, 2], share=4)),
])
def test_gloo_ranks_reproduce_single_process_run(world, cfg, tmp_path):
    got = _run(world, cfg, tmp_path)
    want = _single(cfg)
    assert not np.isnan(got).any()
    assert np.max(np.abs(got - want)) < 1e-13


def test_split_even():
    from qcdgpu_amd.biglattice import messages_after_pass, split_even
    assert split_even(32, 8) == [4] * 8
    assert split_even(12, 4) == [4, 4, 2, 2]
    assert sum(split_even(20, 3)) == 20 and all(n % 2 == 0 for n in split_even(20, 3))
    with pytest.raises(ValueError):
        split_even(4, 4)
    # nothing ever reads U_x one slice to the right: a pass over X links sends one message
    assert messages_after_pass(0) == [(1, 0, +1)]
    assert messages_after_pass(2) == [(1, 0, +1), (0, 1, -1)]
