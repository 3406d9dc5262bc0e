"""One rank of the two-process RCCL test (tests/test_gpu_parity.py): python _rccl_worker.py RANK NRANKS UID_FILE.
Rank 0 creates the RCCL unique id and hands it over through a file (any host-side channel will do: the C ABI only needs
the 128 bytes); every rank binds GPU `rank`, then runs the collectives of include/mvsim.h and checks the data."""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, nranks, uid_file = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    mvs = importlib.import_module("multiview-simulation_amd")
    if rank == 0:
        uid = mvs.Context.comm_unique_id()
        with open(uid_file + ".tmp", "wb") as f:
            f.write(uid)
        os.replace(uid_file + ".tmp", uid_file)
    else:
        t0 = time.time()
        while not os.path.exists(uid_file):
            if time.time() - t0 > 120:
                raise SystemExit("no unique id after 120 s")
            time.sleep(0.05)
        uid = open(uid_file, "rb").read()
    with mvs.Context(rank) as c:
        c.comm_init(nranks, rank, uid)
        n = 1_000_003
        truth = np.random.default_rng(17).random(n, dtype=np.float32)
        d = c.dev_alloc(n * 4)
        for form in ("scatter_allgather", "ring", "peer_copy", "peer_copy", "pipelined"):
            c.set_option("broadcast", form)
            if form == "peer_copy":
                c.comm_register_volume(d, n)            # collective; the second round replaces the first registration
            c.upload(d, truth if rank == 0 else np.full(n, -1.0, np.float32))
            c.comm_broadcast_volume(d, n, 0)
            c.synchronize()
            assert np.array_equal(c.download(d, (n,)), truth), f"rank {rank}: {form} broadcast differs"
        c.upload(d, np.full(n, float(rank + 1), np.float32))
        c.comm_allreduce_sum(d, n)
        c.synchronize()
        assert np.all(c.download(d, (n,)) == nranks * (nranks + 1) / 2)
        assert c.comm_allreduce_sum_f64(0.5 + rank) == sum(0.5 + r for r in range(nranks))
        assert mvs.shard_views(8, nranks, rank) == list(range(rank, 8, nranks))
        c.dev_free(d)
        # BASELINE configs[3]'s rank-level path through the C ABI alone: every rank its z slab of ONE view, the one double of
        # adjustImage's sum through mvsim_comm_allreduce_sum_f64, the stitched acquisition against the untiled view (rank 0)
        import importlib
        tiling = importlib.import_module("multiview-simulation_amd.tiling")
        synth = importlib.import_module("multiview-simulation_amd.synthetic")
        m, inc = 96, 4
        gt = synth.sphere_phantom(m)
        psf = synth.gaussian_psf(7, 9, 21, sigma=(1.3, 1.5, 4.0))
        d_gt = c.dev_alloc(gt.nbytes)
        c.upload(d_gt, gt if rank == 0 else np.zeros_like(gt))
        c.set_option("broadcast", "scatter_allgather")
        c.comm_broadcast_volume(d_gt, gt.size, 0)
        p = c.view_params(degrees=50, delta=0.01, inc=inc, snr=25.0, seed=464232194, stream=5, conv_method=1)
        tv = tiling.TiledView(c, rank, nranks)
        planes = tv.acq_planes(m, inc)
        d_acq = c.dev_alloc(max(1, planes) * m * m * 4)
        info = tv.run(d_gt, (m, m, m), psf.copy(), p, d_acq)
        part = c.download(d_acq, (planes, m, m))
        np.save(uid_file + f".slab{rank}.npy", part)
        c.comm_allreduce_sum_f64(0.0)                       # barrier: every rank's slab is on disk
        if rank == 0:
            tiled = np.concatenate([np.load(uid_file + f".slab{r}.npy") for r in range(nranks)], axis=0)
            whole = c.simulate_view(gt, psf.copy(), p, want=("acq",))["acq"]
            assert tiled.shape == whole.shape and (tiled != whole).mean() < 5e-3, (tiled.shape, whole.shape)
        assert info["planes_rotated"] > info["planes_owned"]
        c.dev_free(d_acq); c.dev_free(d_gt)
        c.comm_destroy()
    print(f"rank {rank} ok", flush=True)


if __name__ == "__main__":
    main()
