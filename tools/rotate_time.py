"""HIP-event time of the device-resident rotation operator (rotateAroundAxis, SimulateMultiViewDataset.java:104-135) on a 512^3 volume:
    python tools/rotate_time.py [size] [degrees]"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mvs = importlib.import_module("multiview-simulation_amd")
synth = importlib.import_module("multiview-simulation_amd.synthetic")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
deg = int(sys.argv[2]) if len(sys.argv) > 2 else -60
ctx = mvs.Context(0)
gt = synth.sphere_phantom(n)
d_in = ctx.dev_alloc(gt.nbytes); ctx.upload(d_in, gt)
d_out = ctx.dev_alloc(gt.nbytes)
for _ in range(3):
    ctx.rotate_around_axis_dev(d_in, (n, n, n), 0, deg, d_out)
ctx.synchronize()
reps = 50
t0 = time.perf_counter()
for _ in range(reps):
    ctx.rotate_around_axis_dev(d_in, (n, n, n), 0, deg, d_out)
ctx.synchronize()
dt = (time.perf_counter() - t0) / reps
out = ctx.download(d_out, (n, n, n))
print(f"rotateAroundAxis {n}^3, {deg} degrees about x: {dt * 1e3:.4f} ms = {n ** 3 / dt / 1e9:.1f} Gvoxel/s; checksum {float(out.astype(np.float64).sum()):.6f}")
