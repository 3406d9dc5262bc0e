import importlib, os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
mvs = importlib.import_module("multiview-simulation_amd")
synth = importlib.import_module("multiview-simulation_amd.synthetic")
# python tools/small_view_overhead.py [n k graph]  (one case only, e.g. under rocprofv3 --kernel-trace --stats)
CASES = ((128, 15, 0), (128, 15, 1), (64, 9, 0), (64, 9, 1))
if len(sys.argv) == 4:
    CASES = (tuple(int(a) for a in sys.argv[1:4]),)
for n, k, graph in CASES:
    gt = synth.sphere_phantom(n)
    ctx = mvs.Context(0)
    ctx.set_option("graph", graph)
    d_gt = ctx.dev_alloc(gt.nbytes); ctx.upload(d_gt, gt)
    d_acq = ctx.dev_alloc(gt.nbytes)
    psf = synth.gaussian_psf(k, sigma=(2.0, 2.0, 2.0))
    p = ctx.view_params(degrees=60, inc=1, snr=25.0, seed=1, stream=0, conv_method=1)
    for _ in range(5):
        ctx.simulate_view_dev(d_gt, (n, n, n), psf.copy(), p, d_acq)
    ctx.synchronize()
    reps = 200
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.simulate_view_dev(d_gt, (n, n, n), psf.copy(), p, d_acq)
    t_issue = time.perf_counter() - t0
    ctx.synchronize()
    t_all = time.perf_counter() - t0
    print(f"{n}^3/{k}^3 graph={graph}: host issue {t_issue/reps*1e6:.0f} us/view, wall {t_all/reps*1e6:.0f} us/view")
    ctx.close()
