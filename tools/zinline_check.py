"""fft_zpass=inline (FFT z pass on the unpadded spectrum, PSF z spectrum per tile) against the direct Kz-tap z pass: deviation of the
convolved volume and of a fused view's acquisition.   python tools/zinline_check.py"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mvs = importlib.import_module("multiview-simulation_amd")
synth = importlib.import_module("multiview-simulation_amd.synthetic")
rng = np.random.default_rng(3)
for (nz, ny, nx), k in (((64, 80, 72), (9, 7, 5)), ((100, 128, 96), (31, 15, 11)), ((256, 256, 256), (63, 31, 31)), ((512, 512, 512), (41, 31, 31))):
    v = synth.sphere_phantom(nx, ny, nz) + 0.01 * rng.random((nz, ny, nx), dtype=np.float32)
    psf = synth.gaussian_psf(k[2], k[1], k[0], sigma=(k[2] / 6, k[1] / 6, k[0] / 5)) * (1 + 0.3 * rng.random(k, dtype=np.float32))
    outs = {}
    for zp in ("direct", "inline"):
        with mvs.Context(0) as c:
            c.set_option("fft_zpass", zp)
            con = c.convolve(v, psf.copy(), method=1)
            p = c.view_params(degrees=40, inc=3, snr=-1.0, conv_method=1)
            acq = c.simulate_view(v, psf.copy(), p, want=("acq",))
            outs[zp] = (con, acq["acq"], acq["corr"])
    a, b = outs["direct"], outs["inline"]
    print(f"{nx}x{ny}x{nz} psf {k[2]}x{k[1]}x{k[0]}: con max|d|/max = {np.abs(a[0] - b[0]).max() / np.abs(a[0]).max():.2e}, "
          f"view acq = {np.abs(a[1] - b[1]).max() / np.abs(a[1]).max():.2e}, corr rel = {abs(a[2] / b[2] - 1):.2e}", flush=True)
