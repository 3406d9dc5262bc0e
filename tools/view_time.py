"""HIP-event time of one fused view at an arbitrary geometry (device-resident), e.g. the BASELINE configs:
    python tools/view_time.py 1024 1024 1024  15 15 41  4      # configs[3]
    python tools/view_time.py 2048 2048 512   63 63 63  1      # configs[4]
Further arguments are context options: zconv_strided=0 fft_zpass=inline ...
"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mvs = importlib.import_module("multiview-simulation_amd")
synth = importlib.import_module("multiview-simulation_amd.synthetic")
nx, ny, nz, kx, ky, kz, inc = (int(a) for a in sys.argv[1:8])
rng = np.random.default_rng(1)
ctx = mvs.Context(0)
ctx.set_option("tail_overlap", 0); ctx.set_option("psf_overlap", 0)   # one kernel at a time: stage times are the kernels' own
gt_kind = "blob"
for kv in sys.argv[8:]:
    if kv.startswith("gt="):
        gt_kind = kv[3:]                 # gt=phantom2x: the 512^3 sphere phantom up-sampled 2x (what bench.py's size_1024 record runs on)
    else:
        ctx.set_option(*kv.split("=", 1))
# cheap compactly supported volume built on the host plane by plane
w = lambda n: np.clip(1 - ((np.arange(n, dtype=np.float32) - (n - 1) / 2) / (0.3 * n)) ** 2, 0, None) ** 2
if gt_kind in ("phantom", "dense"):          # the bench's 512^3 workload: the sphere phantom; dense = the same + 1e-6 in every voxel
    assert nx == ny == nz
    gt = synth.sphere_phantom(nx)
    if gt_kind == "dense":
        gt = gt + np.float32(1e-6)
elif gt_kind == "phantom2x":
    assert nx == ny == nz and nx % 2 == 0
    gt = np.ascontiguousarray(synth.sphere_phantom(nx // 2).repeat(2, axis=0).repeat(2, axis=1).repeat(2, axis=2))
else:
    gt = (w(nz)[:, None, None] * w(ny)[None, :, None]).astype(np.float32) * w(nx)[None, None, :]
d_gt = ctx.dev_alloc(gt.nbytes); ctx.upload(d_gt, gt)
nzo = (nz - 1) // inc + 1
d_acq = ctx.dev_alloc(nzo * ny * nx * 4)
psf = synth.gaussian_psf(kx, ky, kz, sigma=(kx / 6, ky / 6, kz / 6))
p = ctx.view_params(degrees=60, delta=0.01, inc=inc, snr=25.0, seed=1, stream=0, conv_method=1)
for _ in range(2):
    ctx.simulate_view_dev(d_gt, (nx, ny, nz), psf.copy(), p, d_acq)
ctx.synchronize()
ctx.enable_timing(True)
for _ in range(4):
    ctx.simulate_view_dev(d_gt, (nx, ny, nz), psf.copy(), p, d_acq)
t = ctx.timings()
n = nx * ny * nz
print(f"{nx}x{ny}x{nz}, PSF {kx}x{ky}x{kz}, inc {inc}{' ' + ' '.join(sys.argv[8:]) if sys.argv[8:] else ''}: {t['total_ms']:.2f} ms/view = {n / t['total_ms'] / 1e6:.1f} Gvoxel/s  "
      + " ".join(f"{k}={v:.2f}" for k, v in t.items() if k != "total_ms"))
