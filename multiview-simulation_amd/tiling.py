"""Rank-level z-slab tiling of ONE view over the GPUs of a job (BASELINE configs[3]: "1024^3 volume, 6 views ... z-slab tiling
... on 8 GPUs"; the loop body being tiled is SimulateMultiViewDataset.java:570-585).

One process per GPU.  Every rank holds the whole ground truth (the broadcast of the sharded path, `mvsim_comm_broadcast_volume`);
rank r owns the planes [z0, z1) of the view (`mvsim_slab_range`).  Per view and rank:

    slab_sum = mvsim_view_slab_convolve_dev(...)        rotate + attenuate the slab AND the Kz/2 halo planes the PSF reaches
                                                        (attenuation runs along y inside a plane, :335-359, so the halo is
                                                        recomputed locally, never exchanged), convolve the slab
    total    = mvsim_comm_allreduce_sum_f64(slab_sum)   the ONE exchange of the view: the sum Tools.adjustImage divides by
    mvsim_view_slab_finish_dev(..., total, acq)         adjust with the global mean, extract, Poisson (global RNG counters)

The reduction goes through the C ABI's own RCCL communicator (`Context.comm_init`); `allreduce_f64` replaces it only where RCCL
cannot run -- two gloo ranks sharing one GPU in the rehearsal tests.  Nothing here touches torch.
"""
from __future__ import annotations

import time
from typing import Callable, Optional

__all__ = ["acquired_planes", "halo_planes", "TiledView"]


def acquired_planes(z0: int, z1: int, inc: int):
    """Acquired planes k with z0 <= k * inc < z1 (what mvsim_view_slab_finish_dev writes, in order): [k0, k1)."""
    return (z0 + inc - 1) // inc, (z1 + inc - 1) // inc


def halo_planes(nz: int, kz: int, z0: int, z1: int):
    """Input planes [za, zb) the taps of the slab's outputs reach, folded back at the global faces (mirror boundary) -- the planes a
    rank rotates and attenuates for its slab (host restatement of mvsim_view_slab_convolve_dev's range, for the diagnostics)."""
    c = kz // 2
    hl = kz - 1 - c
    za, zb = z0 - hl, z1 + c
    if za < 0:
        zb = max(zb, min(nz, -za + 1))
        za = 0
    if zb > nz:
        za = min(za, max(0, 2 * nz - 1 - zb))
        zb = nz
    if kz >= nz:
        za, zb = 0, nz
    return za, zb


class TiledView:
    """The slab of one rank.  `ctx` computes; `comm_ctx` (default: `ctx`) holds the C ABI's communicator the one double is reduced on."""

    def __init__(self, ctx, rank: int, world: int, comm_ctx=None, allreduce_f64: Optional[Callable[[float], float]] = None):
        if not 0 <= rank < world:
            raise ValueError("rank out of range")
        self.ctx, self.rank, self.world = ctx, rank, world
        self.comm_ctx = comm_ctx if comm_ctx is not None else ctx
        self._allreduce = allreduce_f64

    def slab(self, nz: int):
        return self.ctx.slab_range(nz, self.world, self.rank)

    def acq_planes(self, nz: int, inc: int) -> int:
        z0, z1 = self.slab(nz)
        k0, k1 = acquired_planes(z0, z1, inc)
        return k1 - k0

    def run_on_device(self, gt_dptr: int, dim_xyz, psf, params, acq_dptr: int) -> dict:
        """One tiled view through `mvsim_view_slab_dev` (round 6): the same three steps as ONE asynchronous call -- the slab's share of
        the sum never leaves the device, the reduction runs on the compute context's stream through `comm_ctx`'s communicator, nothing
        waits for the host.  Needs the C ABI's communicator (or a job of one rank); `run` remains for reductions the caller brings."""
        if self._allreduce is not None and self.world > 1:
            raise RuntimeError("run_on_device reduces through the C ABI's communicator; use run() with a caller-supplied reduction")
        nz = int(dim_xyz[2])
        z0, z1 = self.slab(nz)
        got = self.ctx.view_slab_dev(gt_dptr, dim_xyz, psf, params, z0, z1, acq_dptr,
                                     comm_ctx=self.comm_ctx if self.comm_ctx is not self.ctx else None)
        k0, k1 = acquired_planes(z0, z1, int(params.inc))
        if got != k1 - k0:
            raise RuntimeError(f"rank {self.rank}: slab [{z0},{z1}) produced {got} planes, expected {k1 - k0}")
        za, zb = halo_planes(nz, int(psf.shape[0]), z0, z1)
        return {"z0": z0, "z1": z1, "k0": k0, "k1": k1, "planes_rotated": zb - za, "planes_owned": z1 - z0,
                "convolve_ms": 0.0, "allreduce_ms": 0.0}

    def run(self, gt_dptr: int, dim_xyz, psf, params, acq_dptr: int) -> dict:
        """One tiled view.  `acq_dptr` receives this rank's acquired planes (acq_planes() of them, Nx * Ny floats each).
        Returns the slab's geometry and where the time went (host clocks; the first two steps end in a synchronisation by
        construction -- the slab sum travels through the host --, the third is asynchronous on the context's stream)."""
        nz = int(dim_xyz[2])
        z0, z1 = self.slab(nz)
        t0 = time.perf_counter()
        slab_sum = self.ctx.view_slab_convolve_dev(gt_dptr, dim_xyz, psf, params, z0, z1)
        t1 = time.perf_counter()
        total = self._allreduce(slab_sum) if self._allreduce is not None else self.comm_ctx.comm_allreduce_sum_f64(slab_sum)
        t2 = time.perf_counter()
        got = self.ctx.view_slab_finish_dev(dim_xyz, params, z0, z1, total, acq_dptr)
        k0, k1 = acquired_planes(z0, z1, int(params.inc))
        if got != k1 - k0:
            raise RuntimeError(f"rank {self.rank}: slab [{z0},{z1}) produced {got} planes, expected {k1 - k0}")
        za, zb = halo_planes(nz, int(psf.shape[0]), z0, z1)
        return {"z0": z0, "z1": z1, "k0": k0, "k1": k1, "slab_sum": slab_sum, "total": total,
                "planes_rotated": zb - za, "planes_owned": z1 - z0,
                "convolve_ms": (t1 - t0) * 1e3, "allreduce_ms": (t2 - t1) * 1e3}
