#!/usr/bin/env python3
"""One view tiled into z slabs over the ranks of a torch.distributed job (BASELINE configs[3]: "1024^3 volume ...
z-slab tiling ... on 8 GPUs").  One process per GPU:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29544 \
        examples/tiled_view.py --size 1024 --psf 15 15 41 --inc 4

Rank 0 owns the ground truth and broadcasts it (RCCL); every rank rotates+attenuates its slab and the halo planes the
PSF reaches, convolves the slab, the ranks all-reduce ONE double (the sum adjustImage divides by) through the C ABI's own
communicator, and every rank extracts and noises its acquired planes (multiview-simulation_amd/tiling.py: TiledView).  `--check` gathers the slabs on rank 0 and compares them with the untiled
view computed there.  `--backend gloo` lets several ranks share one GPU to rehearse the control flow.
"""
import argparse
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--psf", type=int, nargs=3, default=[15, 15, 41], metavar=("KX", "KY", "KZ"))
    ap.add_argument("--inc", type=int, default=4)
    ap.add_argument("--degrees", type=int, default=60)
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--check", action="store_true")
    a = ap.parse_args()

    import torch
    import torch.distributed as dist
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29544")
    if a.backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group(a.backend, rank=rank, world_size=world)

    mvs = importlib.import_module("multiview-simulation_amd")
    synth = importlib.import_module("multiview-simulation_amd.synthetic")
    tiling = importlib.import_module("multiview-simulation_amd.tiling")
    n = a.size
    dims = (n, n, n)
    gt = torch.empty(n ** 3, dtype=torch.float32, device=dev)
    gt_host = None
    if rank == 0:
        gt_host = synth.sphere_phantom(n)
        gt.copy_(torch.from_numpy(gt_host.reshape(-1)))
    dist.broadcast(gt, src=0)                                    # the only volume-sized exchange
    torch.cuda.synchronize()
    kx, ky, kz = a.psf
    psf = synth.gaussian_psf(kx, ky, kz, sigma=(1.5, 1.6, max(1.0, kz / 6)))

    ctx = mvs.Context(local)
    p = ctx.view_params(degrees=a.degrees, delta=0.01, inc=a.inc, snr=25.0, seed=464232194, stream=0, conv_method=1)
    # the one double of adjustImage's sum travels through the C ABI's own communicator (mvsim_comm_allreduce_sum_f64) -- except where
    # RCCL cannot run: gloo ranks that share a GPU reduce it through torch.distributed
    reduce_fn = None
    if a.backend == "nccl":
        box = [mvs.Context.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        ctx.comm_init(world, rank, box[0])
    else:
        def reduce_fn(x):
            t = torch.tensor([x], dtype=torch.float64)
            dist.all_reduce(t)
            return float(t.item())
    tv = tiling.TiledView(ctx, rank, world, allreduce_f64=reduce_fn)
    got = tv.acq_planes(n, a.inc)
    acq = torch.empty(max(1, got) * n * n, dtype=torch.float32, device=dev)
    info = tv.run(gt.data_ptr(), dims, psf.copy(), p, acq.data_ptr())
    ctx.synchronize()
    z0, z1, k0, k1 = info["z0"], info["z1"], info["k0"], info["k1"]
    print(f"rank {rank}: planes [{z0},{z1}) -> acquired planes [{k0},{k1}), rotated {info['planes_rotated']} planes for {info['planes_owned']} owned, "
          f"slab sum {info['slab_sum']:.6g}, mean count {float(acq[: got * n * n].mean()) if got else float('nan'):.3f}", flush=True)

    if a.check:
        parts = [None] * world
        dist.all_gather_object(parts, acq[: got * n * n].cpu().numpy().reshape(got, n, n))
        if rank == 0:
            tiled = np.concatenate(parts, axis=0)
            whole = ctx.simulate_view(gt_host, psf.copy(), p, want=("acq",))["acq"]
            differing = float((tiled != whole).mean())
            print(f"rank 0: tiled {tiled.shape} vs untiled {whole.shape}: {differing:.2e} of the counts differ "
                  f"(only the order of the global sum differs)", flush=True)
            assert tiled.shape == whole.shape and differing < 5e-3
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
