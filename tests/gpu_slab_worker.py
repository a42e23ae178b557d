"""One rank of the multi-process GPU BigLattice test: the production path of an N-GPU run
(make_rank_slab -> GpuSlabEngine + TorchDistComm, split passes with the interior on the side lane),
with every rank on device 0 of a one-GPU box and the messages staged through gloo instead of RCCL.
Launched by test_gpu_multirank.py."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    import qcdgpu_amd as q
    from qcdgpu_amd import biglattice as bl
    cfg = json.loads(sys.argv[1])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group(backend="gloo")
    L = cfg["L"]
    start = {0: q.START_HOT, 1: q.START_COLD, 2: q.START_GID}[cfg["ints"]]
    if cfg.get("ring"):
        # the C library's own pass over RCCL (one part per process); gloo only carries the identifier and the sums
        lat = bl.make_rank_ring(cfg["group"], L, cfg["beta"], rank, world, 0, nhit=cfg["nhit"],
                                rand_series=cfg["seed"], start_mode=start)
        eng, depth = lat, 1
        lat.comm = lat
        lat.allreduce_sum = lat._global_sum
    else:
        lat = bl.make_rank_slab(cfg["group"], L, cfg["beta"], rank, world, 0, nhit=cfg["nhit"],
                                rand_series=cfg["seed"], start_mode=start)
        eng = lat.engines[0]
        assert isinstance(eng, bl.GpuSlabEngine) and isinstance(lat.comm, bl.TorchDistComm)
        depth = lat.boundary_depth()
    rows = []
    for _ in range(cfg["sweeps"]):
        lat.sweep(1)
        rows.append(np.concatenate([lat.plaquette(), lat.polyakov()]).tolist())
    draws = lat.comm.allreduce_sum(np.array([float(eng.ctx.rng_draws())]))
    np.save(os.path.join(cfg["dir"], "links_%d.npy" % rank), eng.ctx.download())
    if rank == 0:
        with open(os.path.join(cfg["dir"], "rows.json"), "w") as f:
            json.dump({"rows": rows, "draws": int(draws[0]), "boundary_depth": depth,
                       "split": hasattr(eng, "update_split") or bool(cfg.get("ring"))}, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
