"""Micro-benchmarks of single stages on the GPU (development aid; prints ms per call)."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
mvs = importlib.import_module("multiview-simulation_amd")
synth = importlib.import_module("multiview-simulation_amd.synthetic")

def timeit(ctx, fn, key, reps=5):
    ctx.enable_timing(True)
    fn(); ctx.synchronize()
    acc = 0.0
    for _ in range(reps):
        fn()
        acc += ctx.timings()[key]
    ctx.enable_timing(False)
    return acc / reps

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    which = sys.argv[2] if len(sys.argv) > 2 else "poisson"
    ctx = mvs.Context(0)
    dims = (n, n, n)
    nbytes = n ** 3 * 4
    d_in = ctx.dev_alloc(nbytes); d_out = ctx.dev_alloc(nbytes)
    if which == "poisson":
        for name, val in [("zero", 0.0), ("bg 1e-4 (lam .0125)", 1e-4), ("lam 1", 1 / 125), ("lam 5", 5 / 125), ("lam 12", 12 / 125.),
                          ("lam 125", 1.0), ("lam 4000", 32.0), ("lam 22000", 176.0)]:
            ctx.upload(d_in, np.full(n ** 3, val, np.float32))
            ms = timeit(ctx, lambda: ctx.extract_slices_dev(d_in, dims, 1, 25.0, 1234, 0, d_out), "extract_ms")
            print(f"poisson {name:22s}: {ms:8.3f} ms  ({n**3/ms/1e6:8.1f} Gvox/s)")
        ms = timeit(ctx, lambda: ctx.extract_slices_dev(d_in, dims, 1, -1.0, 1234, 0, d_out), "extract_ms")
        print(f"copy only              : {ms:8.3f} ms  ({2*nbytes/ms/1e6:8.1f} GB/s)")
        rng = np.random.default_rng(0)
        mix = np.where(rng.random(n ** 3) < 0.2, 1.0, 1e-4).astype(np.float32)
        ctx.upload(d_in, mix)
        ms = timeit(ctx, lambda: ctx.extract_slices_dev(d_in, dims, 1, 25.0, 1234, 0, d_out), "extract_ms")
        print(f"random 20% bright mix  : {ms:8.3f} ms")
    elif which == "poisson1":
        val = float(sys.argv[3])
        ctx.upload(d_in, np.full(n ** 3, val, np.float32))
        ms = timeit(ctx, lambda: ctx.extract_slices_dev(d_in, dims, 1, 25.0, 1234, 0, d_out), "extract_ms", reps=2)
        print(f"poisson value {val}: {ms:8.3f} ms")
    elif which == "rotate":
        ctx.upload(d_in, synth.sphere_phantom(n))
        for deg in (0, 15, 60, 90):
            ms = timeit(ctx, lambda: ctx.rotate_around_axis_dev(d_in, dims, 0, deg, d_out), "rotate_ms")
            print(f"rotate axis0 {deg:3d} deg: {ms:8.3f} ms ({2*nbytes/ms/1e6:8.1f} GB/s algorithmic)")
        ms = timeit(ctx, lambda: ctx.rotate_around_axis_dev(d_in, dims, 1, 30, d_out), "rotate_ms")
        print(f"rotate axis1 30 deg (generic): {ms:8.3f} ms")
        ms = timeit(ctx, lambda: ctx.attenuate3d_dev(d_in, dims, 0.01, d_out), "attenuate_ms")
        print(f"attenuate: {ms:8.3f} ms ({2*nbytes/ms/1e6:8.1f} GB/s algorithmic)")
    elif which == "conv":
        k = int(sys.argv[3]) if len(sys.argv) > 3 else 31
        ctx.upload(d_in, synth.sphere_phantom(n))
        psf = synth.gaussian_psf(k, sigma=(2.0, 2.2, 6.0))
        def f(): ctx.convolve_dev(d_in, dims, psf.copy(), d_out, method=1)
        ctx.enable_timing(True); f(); ctx.synchronize()
        acc = {}
        for _ in range(5):
            f()
            for kk, v in ctx.timings().items(): acc[kk] = acc.get(kk, 0) + v / 5
        print({kk: round(v, 4) for kk, v in acc.items()})
    ctx.close()

if __name__ == "__main__":
    main()
