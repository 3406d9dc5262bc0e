"""One-off stress of the direct stencil: random volumes and PSFs (1..64 taps per axis, volumes smaller and larger than the
PSF) against the FFT passes (1e-5 of the range) and, for the small ones, the oracle's exact direct sum.
python tools/fuzz_stencil.py [count] [seed]"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mvs = importlib.import_module("multiview-simulation_amd")
import oracle


def run(count=40, seed=99, verbose=True, oracle_limit=4e9):
    rng = np.random.default_rng(seed)
    bad = 0
    with mvs.Context(0) as c:
        for it in range(count):
            shape = tuple(int(rng.integers(1, 90)) for _ in range(3))
            kshape = tuple(int(rng.integers(1, 65)) for _ in range(3))
            v = rng.random(shape, dtype=np.float32)
            psf = rng.random(kshape, dtype=np.float32) + 0.01
            got2 = c.convolve(v, psf.copy(), method=2)
            small = v.size * psf.size < oracle_limit
            ref = oracle.convolve_direct(v, psf.copy()) if small else c.convolve(v, psf.copy(), method=1)
            err = float(np.abs(got2 - ref).max() / np.abs(ref).max())
            ok = err <= 1e-5
            bad += not ok
            if verbose or not ok:
                print(f"{it:3d} vol {shape} psf {kshape} vs {'oracle' if small else 'fft'}: {err:.2e} {'ok' if ok else 'FAIL'}", flush=True)
    if verbose:
        print("failures:", bad)
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 99) else 0)
