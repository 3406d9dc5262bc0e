"""Pass C of the convolution as a direct Kz-tap convolution along z (k_zconv) against the FFT formulation (k_fft_lines<CONV>, z padded to
Nz + Kz - 1), by PSF depth, at the sizes of BASELINE configs[3] and configs[4]; HIP-event stage times of whole fused views, serial.
    python tools/zpass_sweep.py > profiles/r04_zpass_sweep.txt"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mvs = importlib.import_module("multiview-simulation_amd")
synth = importlib.import_module("multiview-simulation_amd.synthetic")
cases = [(1024, 1024, 1024, 4), (2048, 2048, 512, 3)] if len(sys.argv) < 2 else [tuple(int(a) for a in s.split("x")) for s in sys.argv[1:]]
depths = (15, 31, 41, 51, 63)
w = lambda n: np.clip(1 - ((np.arange(n, dtype=np.float32) - (n - 1) / 2) / (0.3 * n)) ** 2, 0, None) ** 2
print("# volume, PSF 31 x 31 x Kz, inc | z pass | pass C ms | B + C + D ms | convolve ms | view ms     (HIP events, one view, overlaps off)")
for nx, ny, nz, inc in cases:
    gt = (w(nz)[:, None, None] * w(ny)[None, :, None]).astype(np.float32) * w(nx)[None, None, :]
    for kz in depths:
        psf = synth.gaussian_psf(31, 31, kz, sigma=(2.0, 2.2, kz / 5.0))
        row = {}
        for zp in ("direct", "inline", "fft"):
            ctx = mvs.Context(0)
            ctx.set_option("tail_overlap", 0); ctx.set_option("psf_overlap", 0); ctx.set_option("fft_zpass", zp)
            d_gt = ctx.dev_alloc(gt.nbytes); ctx.upload(d_gt, gt)
            nzo = (nz - 1) // inc + 1
            d_acq = ctx.dev_alloc(nzo * ny * nx * 4)
            p = ctx.view_params(degrees=60, delta=0.01, inc=inc, snr=25.0, seed=1, stream=0, conv_method=1)
            for _ in range(2):
                ctx.simulate_view_dev(d_gt, (nx, ny, nz), psf.copy(), p, d_acq)
            ctx.synchronize()
            ctx.enable_timing(True)
            for _ in range(3):
                ctx.simulate_view_dev(d_gt, (nx, ny, nz), psf.copy(), p, d_acq)
            t = ctx.timings()
            row[zp] = t
            print(f"{nx}x{ny}x{nz} Kz={kz:2d} inc={inc} | {zp:6s} | {t['pass_c_ms']:8.3f} | {t['pass_b_ms'] + t['pass_c_ms'] + t['pass_d_ms']:8.3f} | "
                  f"{t['convolve_ms']:8.3f} | {t['total_ms']:8.3f}", flush=True)
            ctx.dev_free(d_gt); ctx.dev_free(d_acq); ctx.close()
        d, f, i = row["direct"], row["fft"], row["inline"]
        print(f"#   Kz={kz}: convolve time direct / fft = {d['convolve_ms'] / f['convolve_ms']:.3f}, inline / direct = {i['convolve_ms'] / d['convolve_ms']:.3f}")
