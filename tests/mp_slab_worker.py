"""One rank of the multi-process BigLattice test (gloo, CPU).  Launched by test_multirank.py."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch.distributed as dist
    from qcdgpu_amd import biglattice as bl
    from tests.cpu_slab_engine import CpuSlabEngine
    cfg = json.loads(sys.argv[1])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group(backend="gloo")
    L = cfg["L"]
    parts = cfg.get("parts") or bl.split_even(L[0], world)
    x0 = sum(parts[:rank])
    eng = CpuSlabEngine(cfg["group"], L, cfg["beta"], x0, parts[rank], nhit=cfg["nhit"], seed=cfg["seed"],
                        ints=cfg["ints"], share=cfg.get("share", 1))
    lat = bl.BigLattice([eng], bl.TorchDistComm(eng, rank, world), cfg["group"], L, cfg["beta"])
    if cfg["ints"] == 0:
        lat.fill_halos()
    rows = [np.concatenate([lat.plaquette(), lat.action(), lat.polyakov()]).tolist()]
    for _ in range(cfg["sweeps"]):
        lat.sweep(1)
        rows.append(np.concatenate([lat.plaquette(), lat.action(), lat.polyakov()]).tolist())
    if rank == 0:
        with open(cfg["out"], "w") as f:
            json.dump(rows, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
