"""How much vector work can ride inside passes B and D for free?  Two guest waves per block run a chain of Philox blocks
(option exp_guest = blocks per guest wave; -1 = the plain kernels) while the block's own waves transform their tile.
Prints pass B / D times (HIP events) per setting.    python tools/guest_probe.py"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mvs = importlib.import_module("multiview-simulation_amd")
synth = importlib.import_module("multiview-simulation_amd.synthetic")
n = 512
ctx = mvs.Context(0)
ctx.set_option("psf_overlap", 0)
gt = synth.sphere_phantom(n)
d_in, d_out = ctx.dev_alloc(gt.nbytes), ctx.dev_alloc(gt.nbytes)
ctx.upload(d_in, gt)
psf = synth.gaussian_psf(31, sigma=(2.0, 2.2, 6.0))
def f(): ctx.convolve_dev(d_in, (n, n, n), psf.copy(), d_out, method=1)
ref = None
print("# guest Philox blocks per wave (2 guest waves per block, 9216 blocks per pass): pass B ms, pass D ms, convolve ms, max|d| vs plain")
for it in [int(a) for a in (sys.argv[1] if len(sys.argv) > 1 else "-1,0,20,40,60,80,120,160,240").split(",")]:
    ctx.set_option("exp_guest", str(it))
    ctx.enable_timing(True); f(); ctx.synchronize()
    acc = {}
    for _ in range(8):
        f()
        for k, v in ctx.timings().items(): acc[k] = acc.get(k, 0) + v / 8
    ctx.enable_timing(False)
    out = ctx.download(d_out, gt.shape)
    if ref is None: ref = out
    # guest work = 2 waves x it blocks x ~62 VALU (20 of them quarter rate) per block of B and D
    print(f"{it:4d}  B {acc['pass_b_ms']:.4f}  D {acc['pass_d_ms']:.4f}  conv {acc['convolve_ms']:.4f}  dev {float(np.abs(out - ref).max()):.1e}", flush=True)
