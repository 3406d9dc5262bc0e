import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
mvs = importlib.import_module("multiview-simulation_amd")
synth = importlib.import_module("multiview-simulation_amd.synthetic")
L = mvs._lib.load()
n = 512
ctx = mvs.Context(0)
dims = (n, n, n)
d_in = ctx.dev_alloc(n**3*4); d_out = ctx.dev_alloc(n**3*4)
ctx.upload(d_in, np.random.default_rng(0).random(n**3, dtype=np.float32))
psf = synth.gaussian_psf(31, sigma=(2.0, 2.2, 6.0))
for dbg in [int(a) for a in sys.argv[1:]]:
    L.mvsim_debug_set(dbg)
    def f(): ctx.convolve_dev(d_in, dims, psf.copy(), d_out, method=1)
    ctx.enable_timing(True); f(); ctx.synchronize()
    acc = 0
    for _ in range(4):
        f(); acc += ctx.timings()["convolve_ms"] / 4
    print("dbg", dbg, "convolve_ms", round(acc, 4))
