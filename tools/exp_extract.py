import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mvs = importlib.import_module("multiview-simulation_amd")
synth = importlib.import_module("multiview-simulation_amd.synthetic")
n = 512
gt = synth.sphere_phantom(n)
psf = synth.gaussian_psf(31, sigma=(2.0, 2.2, 6.0))
ctx = mvs.Context(0)
dims = (n, n, n)
d_gt = ctx.dev_alloc(gt.nbytes); d_con = ctx.dev_alloc(gt.nbytes); d_acq = ctx.dev_alloc(gt.nbytes)
ctx.upload(d_gt, gt)
p = ctx.view_params(degrees=60, inc=1, snr=-1.0, conv_method=1)
ctx.simulate_view_dev(d_gt, dims, psf, p, d_acq, con_dptr=d_con)
ctx.synchronize()
ctx.enable_timing(True)
for _ in range(2):
    ctx.extract_slices_dev(d_con, dims, 1, 25.0, 1234, 0, d_acq)
ctx.timings()
acc = 0
for _ in range(5):
    ctx.extract_slices_dev(d_con, dims, 1, 25.0, 1234, 0, d_acq)
print("MVSIM_DBG", os.environ.get("MVSIM_DBG", "0"), "extract_ms", round(ctx.timings()["extract_ms"], 4))
