"""Guest tail (extract + Poisson of view v as guest waves in passes B and D of view v + 1): bit-identity against the
stand-alone kernels and time per 8-view dataset.    python tools/guest_tail_check.py [size] [reps]"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mvs = importlib.import_module("multiview-simulation_amd")
synth = importlib.import_module("multiview-simulation_amd.synthetic")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
nv = 8
gt = synth.sphere_phantom(n)
psf = synth.gaussian_psf(31, sigma=(2.0, 2.2, 6.0))
ctx = mvs.Context(0)
d_gt = ctx.dev_alloc(gt.nbytes); ctx.upload(d_gt, gt)
d_acq = [ctx.dev_alloc(gt.nbytes) for _ in range(nv)]
params = [ctx.view_params(degrees=15 + 45 * v, delta=0.01, inc=1, snr=25.0, seed=464232194, stream=v, conv_method=1) for v in range(nv)]

def dataset():
    for v in range(nv):
        ctx.simulate_view_dev(d_gt, (n, n, n), psf.copy(), params[v], d_acq[v])

res = {}
# configurations: "name=value,name=value;..." (context options); the first one is the reference the others are compared with
configs = sys.argv[3].split(";") if len(sys.argv) > 3 else ["guest_tail=0", "guest_tail=1"]
for ci, cfg in enumerate(configs):
    for kv in cfg.split(","):
        k, v = kv.split("=")
        ctx.set_option(k, v.replace(":", ","))
    dataset(); ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): dataset()
    ctx.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    print(f"{cfg:50s}: {ms:.3f} ms per {nv} views = {nv * n**3 / ms / 1e6:.1f} Gvoxel/s", flush=True)
    res["0" if ci == 0 else "1"] = [ctx.download(d, gt.shape) for d in d_acq]
for v in range(nv):
    same = np.array_equal(res["0"][v], res["1"][v])
    print(f"view {v}: identical={same} mean={res['1'][v].mean():.4f}" + ("" if same else f" ndiff={(res['0'][v] != res['1'][v]).sum()}"))
