"""How well do the vector-issue-bound sampler kernels and the HBM-bound convolution passes share the chip?  Two contexts on one
GPU: context A runs the FFT convolution of a 512^3 volume back to back, context B runs extract + Poisson back to back; time of
N iterations of each ALONE and of both TOGETHER (two host threads).  together ~ max(alone) = the kernels co-schedule;
together ~ sum = they serialise.    python tools/overlap_probe.py"""
import importlib, os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mvs = importlib.import_module("multiview-simulation_amd")
synth = importlib.import_module("multiview-simulation_amd.synthetic")

n, reps = 512, 20
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
tc, tn, tb = timed(run_conv), timed(run_noise), timed(run_conv, run_noise)
print(f"convolve alone {tc:.3f} ms/iter, extract+Poisson alone {tn:.3f} ms/iter, both together {tb:.3f} ms/iter "
      f"(sum {tc + tn:.3f}, max {max(tc, tn):.3f}): overlap recovers {(tc + tn - tb) / min(tc, tn):.0%} of the shorter one")
