"""PCIe-inclusive rate of the host-buffer entry point (what the JNI boundary calls); not the headline metric."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mvs = importlib.import_module("multiview-simulation_amd")
synth = importlib.import_module("multiview-simulation_amd.synthetic")
n = 512
gt = synth.sphere_phantom(n)
psf = synth.gaussian_psf(31, sigma=(2.0, 2.2, 6.0))
ctx = mvs.Context(0)
p = ctx.view_params(degrees=60, inc=1, snr=25.0)
gt_pinned = ctx.pinned_empty(gt.shape)
gt_pinned[...] = gt
acq_pinned = ctx.pinned_empty(gt.shape)
for label, g, dst in (("pageable", gt, None), ("page-locked (mvsim_host_alloc)", gt_pinned, {"acq": acq_pinned})):
    ref = ctx.simulate_view(g, psf.copy(), p, out=dst)["acq"].copy()
    t0 = time.perf_counter()
    for _ in range(3):
        out = ctx.simulate_view(g, psf.copy(), p, out=dst)
    dt = (time.perf_counter() - t0) / 3
    assert np.array_equal(out["acq"], ref)
    print(f"host-buffer simulate_view 512^3, {label} buffers: {dt*1e3:.1f} ms/view = {n**3/dt/1e6:.0f} Mvoxel/s "
          f"(H2D 0.54 GB + D2H 0.54 GB included)")
