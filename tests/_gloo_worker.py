"""Worker for the world_size-2 gloo test: the host-side sharding logic of the multi-GPU path
(bench.py / INTEGRATION.md) without touching a GPU."""
import importlib
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_path = sys.argv[1]
    rank = int(os.environ["RANK"])
    world = int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mvs = importlib.import_module("multiview-simulation_amd")
    synth = importlib.import_module("multiview-simulation_amd.synthetic")

    n_views = 8 * world
    mine = mvs.shard_views(n_views, world, rank)
    angles = [15 + (360 * v) // n_views for v in range(n_views)]

    # rank 0 owns the ground truth and the communicator id; everybody receives both
    gt = torch.from_numpy(synth.sphere_phantom(16).reshape(-1).copy()) if rank == 0 else torch.empty(16 ** 3)
    dist.broadcast(gt, src=0)
    uid = [bytes(range(128))] if rank == 0 else [None]
    dist.broadcast_object_list(uid, src=0)

    # z-slab tiling of one view (mvsim_slab_range is host-only): every rank's slab, and the one-double exchange that
    # adjustImage needs -- here the "slab sum" is the sum of the rank's planes of the broadcast volume
    import ctypes
    L = importlib.import_module("multiview-simulation_amd._lib").load()
    z0, z1 = ctypes.c_int64(), ctypes.c_int64()
    assert L.mvsim_slab_range(16, world, rank, ctypes.byref(z0), ctypes.byref(z1)) == 0
    slab_sum = torch.tensor([float(gt.reshape(16, 16, 16)[z0.value:z1.value].double().sum())], dtype=torch.float64)
    dist.all_reduce(slab_sum)

    # every rank reports its shard; the union must be a partition of the views
    gathered = [None] * world
    dist.all_gather_object(gathered, {"rank": rank, "views": mine, "angles": [angles[v] for v in mine],
                                      "gt_sum": float(gt.double().sum()), "uid_len": len(uid[0]),
                                      "slab": [z0.value, z1.value], "slab_total": float(slab_sum.item())})
    # max-over-ranks timing reduction as bench.py does it
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.barrier()
    if rank == 0:
        json.dump({"gathered": gathered, "tmax": float(t.item()), "n_views": n_views}, open(out_path, "w"))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
