"""One-off stress of the fused rotate + attenuate + x-transform kernel: random geometries, fused_fftx = 1 against 0, every
output bit-identical.    python tools/fuzz_fused_rotate.py [count] [seed]"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mvs = importlib.import_module("multiview-simulation_amd")


def run(count=40, seed=2024, verbose=True):
    rng = np.random.default_rng(seed)
    bad = 0
    for it in range(count):
        nx = int(rng.choice([64, 65, 100, 128, 191, 192, 256, 300, 448, 512, 513, 640, 777, 1000, 1024]))
        ny = nx + int(rng.integers(0, 70))
        nz = int(rng.integers(6, 28))
        kx = int(min(2 * rng.integers(0, 32) + 1, nx))
        ky = int(min(2 * rng.integers(0, 16) + 1, 2 * ((ny - 1) // 2) - 1))
        kz = int(min(2 * rng.integers(0, 12) + 1, 63))
        gt = (rng.random((nz, ny, nx), dtype=np.float32) ** 4) * (rng.random((nz, ny, nx)) < rng.choice([0.02, 0.3, 1.0])).astype(np.float32)
        if rng.random() < 0.5:
            gt[:, : ny // 3] = 0      # empty rows: the zero fast paths
        psf = rng.random((kz, ky, kx), dtype=np.float32) + 0.05
        deg, inc = int(rng.integers(-179, 180)), int(rng.integers(1, 4))
        res = []
        for mode in (0, 1):
            with mvs.Context(0) as c:
                c.set_option("fused_fftx", mode)
                p = c.view_params(degrees=deg, inc=inc, snr=25.0, seed=464232194, stream=it, conv_method=1)
                want = ("rot", "att", "con", "acq") if it % 2 == 0 else ("acq",)
                res.append(c.simulate_view(gt, psf.copy(), p, want=want))
        ok = all(np.array_equal(res[0][k], res[1][k]) for k in res[0] if isinstance(res[0][k], np.ndarray))
        bad += not ok
        if verbose or not ok:
            print(f"{it:3d} {nx}x{ny}x{nz} psf {kx}x{ky}x{kz} deg {deg} inc {inc}: {'ok' if ok else 'MISMATCH'}", flush=True)
    if verbose:
        print("mismatches:", bad)
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 2024) else 0)
