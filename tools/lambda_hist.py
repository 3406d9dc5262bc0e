"""lambda histogram of the bench workload (512^3 sphere phantom, 31^3 PSF, SNR 25): which sampler regime the voxels of
a view fall into, and how many end up as work items.  python tools/lambda_hist.py [n] [degrees]"""
import importlib, sys
import numpy as np
sys.path.insert(0, "/root/repo")
mvs = importlib.import_module("multiview-simulation_amd")
synth = importlib.import_module("multiview-simulation_amd.synthetic")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
deg = int(sys.argv[2]) if len(sys.argv) > 2 else 15
gt = synth.sphere_phantom(n)
psf = synth.gaussian_psf(31, sigma=(2.0, 2.2, 6.0))
with mvs.Context(0) as ctx:
    p = ctx.view_params(degrees=deg, inc=1, snr=25.0, seed=464232194, stream=0, conv_method=1)
    r = ctx.simulate_view(gt, psf, p, want=("con", "acq"))
mul = (25.0 / np.sqrt(5.0)) ** 2
lam = r["con"].astype(np.float64).ravel() * mul
print(f"n={n} deg={deg} mul={mul:.3f} mean lambda {lam.mean():.3f} max {lam.max():.1f}")
edges = [0, 1e-3, 0.01, 0.05, 0.1, 0.3, 1, 3, 10, 30, 100, 300, 1000, 1e9]
h, _ = np.histogram(lam, bins=edges)
for a, b, c in zip(edges[:-1], edges[1:], h):
    print(f"  [{a:g}, {b:g}): {c / lam.size * 100:7.3f} %")
print(f"  lambda <= 0: {(lam <= 0).mean() * 100:.3f} %")
small = (lam > 0) & (lam < 10)
# P(count >= 1) = 1 - exp(-lambda): the share of inversion voxels the shortcut cannot settle is at least that
print(f"  inversion regime {small.mean() * 100:.2f} %, of which expected count>=1: {(1 - np.exp(-lam[small])).mean() * 100:.2f} %;"
      f" shortcut misses (u >= 1-lambda-1e-6 or lambda >= 1): {np.where(lam[small] < 1, np.minimum(1, lam[small] + 1e-6), 1).mean() * 100:.2f} %")
print(f"  PTRS regime {(lam >= 10).mean() * 100:.2f} %")
# run-length structure along x: how often are all 4 voxels of a group / all 256 of a wave slot in the same regime
g4 = (lam >= 10).reshape(-1, 4)
print(f"  groups of 4: all bright {g4.all(1).mean() * 100:.2f} %, none bright {(~g4.any(1)).mean() * 100:.2f} %, mixed {(g4.any(1) & ~g4.all(1)).mean() * 100:.2f} %")
g256 = (lam >= 10).reshape(-1, 256)
print(f"  wave slots of 256: none bright {(~g256.any(1)).mean() * 100:.2f} %, all bright {g256.all(1).mean() * 100:.2f} %, mean bright in mixed {g256[g256.any(1) & ~g256.all(1)].sum(1).mean():.1f}")
t4 = (lam < 1e-3).reshape(-1, 4)
print(f"  groups of 4 with all lambda < 1e-3: {t4.all(1).mean() * 100:.2f} %")
