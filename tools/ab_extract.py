"""A/B harness: HIP-event stage times of one 512^3 view (serial) for the library under ROOT (a checkout / worktree / copy of the package with its own
libmvsim.so), so that several builds run on ONE box in one gpurun call:   python tools/ab_extract.py ROOT [dense]"""
import importlib, os, sys, numpy as np
root = sys.argv[1]
sys.path.insert(0, root)
mvs = importlib.import_module("multiview-simulation_amd")
synth = importlib.import_module("multiview-simulation_amd.synthetic")
n = 512
ctx = mvs.Context(0)
ctx.set_option("tail_overlap", 0); ctx.set_option("psf_overlap", 0)
gt = synth.sphere_phantom(n)
if len(sys.argv) > 2 and sys.argv[2] == "dense":
    gt = gt + np.float32(1e-6)          # no empty rows: nothing for the zero fast paths of the fused rotate kernel to skip
psf = synth.gaussian_psf(31, sigma=(2.0, 2.2, 6.0))
d_gt = ctx.dev_alloc(gt.nbytes); ctx.upload(d_gt, gt)
d_acq = ctx.dev_alloc(gt.nbytes)
p = ctx.view_params(degrees=60, delta=0.01, inc=1, snr=25.0, seed=464232194, stream=0, conv_method=1)
for _ in range(3): ctx.simulate_view_dev(d_gt, (n, n, n), psf.copy(), p, d_acq)
ctx.synchronize(); ctx.enable_timing(True)
acc = {}
for _ in range(16):
    ctx.simulate_view_dev(d_gt, (n, n, n), psf.copy(), p, d_acq)
t = ctx.timings()
print(os.path.basename(root) or root, (sys.argv[2] if len(sys.argv) > 2 else "phantom"), {k: round(v, 4) for k, v in t.items() if k in ("rotate_ms", "convolve_ms", "extract_ms", "total_ms", "pass_b_ms", "pass_c_ms", "pass_d_ms", "pass_e_ms")}, flush=True)
