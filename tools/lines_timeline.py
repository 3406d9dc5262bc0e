"""Where a block of the image's y passes (k_fft_lines FWD = pass B, INV = pass D) spends its time: shader-clock stamps written by an
ATTRIBUTION build (python tools/build_variant.py stamps "-DMVSIM_DEV_ATTRIBUTION -DMVSIM_EXP_LINES_STAMPS" fft_kernels.hip), read back
after one view:
    cp multiview-simulation_amd/libmvsim_stamps.so multiview-simulation_amd/libmvsim.so   (on the GPU box: the copy there is scratch)
    python tools/lines_timeline.py 1024 1024 1024 31 31 31 1 [gt=phantom2x]
Prints, per pass, the mean cycles between the stamps and how many blocks were alive at once."""
import ctypes, importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mvs = importlib.import_module("multiview-simulation_amd")
synth = importlib.import_module("multiview-simulation_amd.synthetic")
nx, ny, nz, kx, ky, kz, inc = (int(a) for a in sys.argv[1:8])
ctx = mvs.Context(0)
ctx.set_option("tail_overlap", 0); ctx.set_option("psf_overlap", 0)
gt_kind = "blob"
for kv in sys.argv[8:]:
    if kv.startswith("gt="):
        gt_kind = kv[3:]
    else:
        ctx.set_option(*kv.split("=", 1))
w = lambda n: np.clip(1 - ((np.arange(n, dtype=np.float32) - (n - 1) / 2) / (0.3 * n)) ** 2, 0, None) ** 2
if gt_kind == "phantom":
    gt = synth.sphere_phantom(nx)
elif gt_kind == "phantom2x":
    gt = np.ascontiguousarray(synth.sphere_phantom(nx // 2).repeat(2, axis=0).repeat(2, axis=1).repeat(2, axis=2))
else:
    gt = (w(nz)[:, None, None] * w(ny)[None, :, None]).astype(np.float32) * w(nx)[None, None, :]
d_gt = ctx.dev_alloc(gt.nbytes); ctx.upload(d_gt, gt)
nzo = (nz - 1) // inc + 1
d_acq = ctx.dev_alloc(nzo * ny * nx * 4)
psf = synth.gaussian_psf(kx, ky, kz, sigma=(kx / 6, ky / 6, kz / 6))
p = ctx.view_params(degrees=60, delta=0.01, inc=inc, snr=25.0, seed=1, stream=0, conv_method=1)
for _ in range(3):
    ctx.simulate_view_dev(d_gt, (nx, ny, nz), psf.copy(), p, d_acq)
ctx.synchronize()
ctx.enable_timing(True)
ctx.simulate_view_dev(d_gt, (nx, ny, nz), psf.copy(), p, d_acq)
t = ctx.timings()
print(f"{nx}x{ny}x{nz} K {kx}x{ky}x{kz} inc {inc}: pass B {t['pass_b_ms']:.3f} ms, pass D {t['pass_d_ms']:.3f} ms")
lib = mvs._lib.load()
fn = lib.mvsim_dev_read_line_stamps
fn.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t]
NB = 1 << 20
names = ["start->loads requested", "->loads arrived (twiddles staged)", "->tile in LDS (barrier)", "->wave 0 transformed", "->barrier", "->stores requested"]
for inv, label in ((0, "pass B (FWD)"), (1, "pass D (INV)")):
    buf = np.zeros((NB, 8), dtype=np.uint64)
    assert fn(inv, buf.ctypes.data, NB) == 0
    live = buf[(buf[:, 0] != 0) & (buf[:, 6] > buf[:, 0])]          # blocks that ran to the end in the last launch (empty planes return early)
    if not len(live):
        print(label, ": no stamps"); continue
    t0 = live[:, 0].astype(np.int64)
    d = np.diff(live[:, :7].astype(np.int64), axis=1)
    span = int(live[:, 6].max() - live[:, 0].min())
    tot = (live[:, 6].astype(np.int64) - t0)
    print(f"{label}: {len(live)} blocks, launch spans {span} clocks; a block lives {tot.mean():.0f} clocks (p10 {np.percentile(tot, 10):.0f}, p90 {np.percentile(tot, 90):.0f}); "
          f"blocks alive at once (sum of lives / span): {tot.sum() / span:.1f}")
    for k, nme in enumerate(names):
        print(f"    {nme:36s} mean {d[:, k].mean():9.0f}   p10 {np.percentile(d[:, k], 10):9.0f}   p90 {np.percentile(d[:, k], 90):9.0f}   share {d[:, k].mean() / tot.mean():.2f}")
