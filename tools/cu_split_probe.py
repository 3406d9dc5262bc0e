"""Two probes of how the view's kernels could share the chip.

1. CU split.  Context A runs the FFT convolution of a 512^3 volume back to back, context B extract + Poisson back to back
   (as tools/overlap_probe.py), but each context's stream is confined to a CU set (option cu_range, hipExtStreamCreateWithCUMask):
   B on the first k CUs of the mask order, A on the others.  Reported per k: each alone on its CU set, both together, and
   what a split pipeline of `conv || sampler` would take per view against the serial sum on the whole chip.
2. kx panels.  Passes B, C', D panel by panel over kx (option kx_panel): the convolution's time and its deviation from the
   unpanelled result.

    python tools/cu_split_probe.py [--reps 20] > profiles/r04_cu_split.txt"""
import argparse, importlib, os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mvs = importlib.import_module("multiview-simulation_amd")
synth = importlib.import_module("multiview-simulation_amd.synthetic")

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--size", type=int, default=512)
ap.add_argument("--splits", default="0,32,48,64,80,96,128")
ap.add_argument("--panels", default="0,32,48,64,96,144")
args = ap.parse_args()
n, reps = args.size, args.reps
gt = synth.sphere_phantom(n)
psf = synth.gaussian_psf(31, sigma=(2.0, 2.2, 6.0))
a, b = mvs.Context(0), mvs.Context(0)
for c in (a, b):
    c.set_option("psf_overlap", 0)
da_in, da_out = a.dev_alloc(gt.nbytes), a.dev_alloc(gt.nbytes)
db_in, db_out = b.dev_alloc(gt.nbytes), b.dev_alloc(gt.nbytes)
a.upload(da_in, gt)
con = a.convolve(gt, psf.copy(), method=1)
con *= np.float32(1.0 / con.mean())
b.upload(db_in, con)


def run_conv():
    for _ in range(reps):
        a.convolve_dev(da_in, (n, n, n), psf.copy(), da_out, method=1)
    a.synchronize()


def run_noise():
    for _ in range(reps):
        b.extract_slices_dev(db_in, (n, n, n), 1, 25.0, 464232194, 1, db_out)
    b.synchronize()


def timed(*fns):
    ts = [threading.Thread(target=f) for f in fns]
    t0 = time.perf_counter()
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    return (time.perf_counter() - t0) / reps * 1e3


run_conv(); run_noise()
print(f"# CU split probe, {n}^3, {reps} iterations each; convolve = PSF spectrum + passes A-E of mvsim_convolve_dev, sampler = "
      f"extract + Poisson (k_extract4_noise2 + k_poisson_resolve)")
print("# k = CUs of the sampler's stream (0: no masks at all); times in ms per iteration")
print("#   k  conv_alone  sampler_alone  together   sum_unmasked  gain_vs_serial")
base = None
for k in [int(x) for x in args.splits.split(",")]:
    if k == 0:
        a.set_option("cu_range", "0:0"); b.set_option("cu_range", "0:0")
    else:
        a.set_option("cu_range", f"{k}:256"); b.set_option("cu_range", f"0:{k}")
    run_conv(); run_noise()
    tc, tn, tb = timed(run_conv), timed(run_noise), timed(run_conv, run_noise)
    if base is None:
        base = tc + tn
    print(f"  {k:3d}  {tc:9.3f}  {tn:12.3f}  {tb:9.3f}  {base:12.3f}  {(base - tb) / base:+.1%}", flush=True)
# the unmasked convolution beside a masked sampler: the convolution's kernels may use every CU the sampler leaves idle
print("# sampler confined to k CUs, convolution unmasked")
for k in (32, 64, 96):
    a.set_option("cu_range", "0:0"); b.set_option("cu_range", f"0:{k}")
    run_conv(); run_noise()
    tc, tn, tb = timed(run_conv), timed(run_noise), timed(run_conv, run_noise)
    print(f"  {k:3d}  {tc:9.3f}  {tn:12.3f}  {tb:9.3f}  {base:12.3f}  {(base - tb) / base:+.1%}", flush=True)
a.set_option("cu_range", "0:0"); b.set_option("cu_range", "0:0")

print("# kx panels: passes B, C', D of the convolution panel by panel (columns per panel; 0 = whole spectrum per pass)")
print("#  cols  conv_ms   max|d|/max vs unpanelled")
ref = None
for p in [int(x) for x in args.panels.split(",")]:
    a.set_option("kx_panel", str(p))
    run_conv()
    tc = timed(run_conv)
    out = a.download(da_out, gt.shape, np.float32) if hasattr(a, "download") else None
    dev = ""
    if out is not None:
        if ref is None:
            ref = out
        dev = f"{float(np.max(np.abs(out - ref)) / np.max(np.abs(ref))):.2e}"
    print(f"  {p:4d}  {tc:7.3f}   {dev}", flush=True)
a.set_option("kx_panel", "0")
