import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
mvs = importlib.import_module("multiview-simulation_amd")
synth = importlib.import_module("multiview-simulation_amd.synthetic")
n = 512
ctx = mvs.Context(0)
dims = (n, n, n)
d_in = ctx.dev_alloc(n**3*4); d_out = ctx.dev_alloc(n**3*4)
ctx.upload(d_in, np.random.default_rng(0).random(n**3, dtype=np.float32))
psf = synth.gaussian_psf(31, sigma=(2.0, 2.2, 6.0))
def f(): ctx.convolve_dev(d_in, dims, psf.copy(), d_out, method=1)
ctx.enable_timing(True); f(); ctx.synchronize()
acc = {}
for _ in range(5):
    f()
    for k, v in ctx.timings().items(): acc[k] = acc.get(k, 0) + v / 5
print({k: round(v, 4) for k, v in acc.items() if v})
