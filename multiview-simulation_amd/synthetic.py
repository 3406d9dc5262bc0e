"""Synthetic workloads for tests and bench.py (host-side numpy; not part of the hot path).

The reference's ground truth is a cloud of small spheres inside a large sphere
(SimulateMultiViewDataset.drawSpheres, :436-522) and its PSFs are measured 51^3 stacks with
sigma ~ (2.0, 2.2, 6.7) px.  The GPL data files are not copied; these generators produce inputs of
the same character.
"""
from __future__ import annotations

import numpy as np


def sphere_phantom(nx: int, ny: int | None = None, nz: int | None = None, seed: int = 464232194) -> np.ndarray:
    """Zero background; S = round(12000 (N/578)^3) spheres, centres uniform in a ball of radius
    0.335*Nmin, integer radius in [1, max(1, 20 N/578)], value U[0,1), max-composited."""
    ny = ny or nx
    nz = nz or nx
    nmin = min(nx, ny, nz)
    rng = np.random.Generator(np.random.Philox(key=seed))
    vol = np.zeros((nz, ny, nx), dtype=np.float32)
    count = max(1, int(round(12000 * (nmin / 578.0) ** 3)))
    rmax = max(1, int(20 * nmin / 578))
    big = 0.335 * nmin
    c = np.array([(nz - 1) / 2.0, (ny - 1) / 2.0, (nx - 1) / 2.0])
    for _ in range(count):
        while True:
            p = rng.uniform(-1.0, 1.0, 3)
            if p @ p <= 1.0:
                break
        cz, cy, cx = np.rint(c + p * big).astype(int)
        r = int(rng.integers(1, rmax + 1))
        val = np.float32(rng.random())
        z0, z1 = max(cz - r, 0), min(cz + r + 1, nz)
        y0, y1 = max(cy - r, 0), min(cy + r + 1, ny)
        x0, x1 = max(cx - r, 0), min(cx + r + 1, nx)
        if z0 >= z1 or y0 >= y1 or x0 >= x1:
            continue
        zz, yy, xx = np.ogrid[z0:z1, y0:y1, x0:x1]
        mask = (zz - cz) ** 2 + (yy - cy) ** 2 + (xx - cx) ** 2 <= r * r
        sub = vol[z0:z1, y0:y1, x0:x1]
        np.maximum(sub, np.where(mask, val, np.float32(0)), out=sub)
    return vol


def gaussian_psf(kx: int, ky: int | None = None, kz: int | None = None, sigma=(2.0, 2.2, 6.0)) -> np.ndarray:
    """Un-normalised anisotropic Gaussian, peak 1 at index K/2; shape (Kz, Ky, Kx), sigma = (sx, sy, sz)."""
    ky = ky or kx
    kz = kz or kx
    sx, sy, sz = sigma
    x = np.arange(kx) - kx // 2
    y = np.arange(ky) - ky // 2
    z = np.arange(kz) - kz // 2
    g = np.exp(-0.5 * ((z[:, None, None] / sz) ** 2 + (y[None, :, None] / sy) ** 2 + (x[None, None, :] / sx) ** 2))
    return np.ascontiguousarray(g, dtype=np.float32)


def hourglass_psf(k: int = 63, sigma=(2.5, 2.5, 10.0), tilt_deg: float = 20.0) -> np.ndarray:
    """Non-separable PSF: Gaussian x (1 + 0.5 cos(2 phi) r_perp / 8), tilted in the y-z plane, clamped >= 0."""
    c = k // 2
    z, y, x = np.meshgrid(np.arange(k) - c, np.arange(k) - c, np.arange(k) - c, indexing="ij")
    t = np.deg2rad(tilt_deg)
    yr = np.cos(t) * y + np.sin(t) * z
    zr = -np.sin(t) * y + np.cos(t) * z
    sx, sy, sz = sigma
    g = np.exp(-0.5 * ((x / sx) ** 2 + (yr / sy) ** 2 + (zr / sz) ** 2))
    rp = np.sqrt(x ** 2 + yr ** 2)
    phi = np.arctan2(yr, x)
    g = g * (1.0 + 0.5 * np.cos(2 * phi) * rp / 8.0)
    return np.ascontiguousarray(np.maximum(g, 0.0), dtype=np.float32)


def measured_like_psf(k: int = 51) -> np.ndarray:
    """Stand-in for the reference's shipped PSF stacks (`src/main/resources/Angle<k>.tif`, loaded at
    SimulateMultiViewDataset.java:579).  What those 18 files are is recorded -- as facts, not pixels -- in
    `tests/golden/psf_tiff_facts.json` (8 distinct 51^3 float32 stacks, peak 0.99 at (25, 25, 25), minimum 0, fp64 sums 205 .. 303,
    3.1 .. 4.6 % of the voxels non-zero, second-moment widths sigma_x 1.85 .. 2.15, sigma_y 2.02 .. 2.34, sigma_z 6.41 .. 7.22 px,
    0.900 .. 0.954 of their energy in the best separable approximation).  A single Gaussian of those widths sums to ~450: the measured
    stacks have a sharp core on a wide, tilted, not separable skirt.  Hence two tilted hour-glass terms -- a core of sigma (1.2, 1.3,
    3.5) carrying 77 % of the peak and a skirt of (2.4, 2.6, 8.0), both tilted 10 degrees in the y-z plane --, scaled to peak 0.99 and
    clipped to zero below 0.5 % of the peak like a background-subtracted measurement: sigma (2.15, 2.28, 6.53), sum 235, 3.3 % non-zero,
    rank-1 energy 0.940 at k = 51 -- every figure inside the range of the real stacks (checked by
    tests/test_host_logic.py::test_reference_psf_stacks_through_tiffio).  The GPL data files themselves are not part of this repository."""
    core = hourglass_psf(k, sigma=(1.2, 1.3, 3.5), tilt_deg=10.0).astype(np.float64)
    skirt = hourglass_psf(k, sigma=(2.4, 2.6, 8.0), tilt_deg=10.0).astype(np.float64)
    g = 0.766 * core + (1.0 - 0.766) * skirt
    g = (g * (0.99 / g.max())).astype(np.float32)
    g[g < np.float32(0.005 * 0.99)] = 0.0
    return np.ascontiguousarray(g, dtype=np.float32)


def view_angles(n_views: int, offset: int = 15) -> list[int]:
    """angleOffset + k * (360 / n_views) (SimulateMultiViewDataset.java:540-548,567-570)."""
    step = 360 // n_views
    return [offset + k * step for k in range(n_views)]
