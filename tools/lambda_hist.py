import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mvs = importlib.import_module("multiview-simulation_amd")
synth = importlib.import_module("multiview-simulation_amd.synthetic")
n = 512
gt = synth.sphere_phantom(n)
psf = synth.gaussian_psf(31, sigma=(2.0, 2.2, 6.0))
ctx = mvs.Context(0)
p = ctx.view_params(degrees=60, inc=1, snr=-1.0, conv_method=1)
res = ctx.simulate_view(gt, psf, p, want=("con", "acq"))
lam = res["con"].astype(np.float64) * 124.99999999999997
edges = [0, 0.02, 1, 10, 32, 64, 125, 1000, 4000, 1e9]
h, _ = np.histogram(lam, bins=edges)
print("lambda histogram (fraction):")
for a, b, c in zip(edges[:-1], edges[1:], h):
    print(f"  [{a:8.2f}, {b:10.2f}) : {c / lam.size:8.4f}")
print("mean", lam.mean(), "median", np.median(lam), "p99", np.percentile(lam, 99), "max", lam.max())
