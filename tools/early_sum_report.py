"""What taking adjustImage's sum from the spectrum side (epilogue of the z pass) instead of from pass E's voxels does to
a 512^3 view: relative deviation of the two sums / corrections, float-rounding flips of the adjusted volume, count flips.
    python tools/early_sum_report.py > profiles/r02_early_sum.txt
"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mvs = importlib.import_module("multiview-simulation_amd")
synth = importlib.import_module("multiview-simulation_amd.synthetic")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
gt = synth.sphere_phantom(n)
psf = synth.gaussian_psf(31, sigma=(2.0, 2.2, 6.0))
with mvs.Context(0) as c:
    for deg in (15, 60):
        p = c.view_params(degrees=deg, inc=1, snr=25.0, seed=464232194, stream=0, conv_method=1)
        c.set_option("early_sum", 1)
        a = c.simulate_view(gt, psf.copy(), p, want=("con", "acq"))
        c.set_option("early_sum", 0)
        b = c.simulate_view(gt, psf.copy(), p, want=("con", "acq"))
        rel = abs(a["corr"] - b["corr"]) / b["corr"]
        flips = float((a["con"] != b["con"]).mean())
        ulp = float(np.max(np.abs(a["con"] - b["con"]) / b["con"]))
        cnt = float((a["acq"] != b["acq"]).mean())
        dmax = float(np.abs(a["acq"] - b["acq"]).max())
        print(f"{n}^3, 31^3 PSF, {deg} deg: corr(early) = {a['corr']:.12g}, corr(pass E) = {b['corr']:.12g}, relative deviation {rel:.2e}; "
              f"adjusted voxels that differ: {flips:.4f} (max relative {ulp:.2e} = one float ulp); counts that differ: {cnt:.2e} (max |d| {dmax:g})")
