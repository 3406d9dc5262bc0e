"""Random geometries through the strided z pass of compact views (k_zconv_strided, forced with exp=2) against the same view with every plane
convolved (zconv_strided=0): planes to 3e-6 of the range, the factor to 1e-6, and skip_empty on against off bit for bit.
    python tools/fuzz_strided_zpass.py [cases] [seed]"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mvs = importlib.import_module("multiview-simulation_amd")


def run(n_cases=100, seed=1, EXP=2, verbose=True):
    """EXP: experiment bits beside 2 (= the strided kernel wherever its geometry allows).  Returns the number of failing cases."""
    rng = np.random.default_rng(seed)
    ctx = mvs.Context(0)
    ctx.set_option("fused_fftx", 1)                             # the fused rotate kernel (the source of the plane flags) wherever the geometry allows it
    bad = 0
    degenerate = 0
    for case in range(n_cases):
        inc = int(rng.integers(2, 5))
        nx = int(rng.integers(8, 80))                         # (any plane size: the sampler takes planes that are no multiple of four voxels group by group)
        ny = nx + int(rng.integers(0, 24))
        nz = int(rng.integers(2, 340))
        kx, ky = int(rng.integers(1, 12)), int(rng.integers(1, 12))
        kz = int(rng.integers(1, 65))
        if nz * ny * nx > 6_000_000:
            nz = max(2, 6_000_000 // (ny * nx))
        gt = (rng.random((nz, ny, nx), dtype=np.float32) * (rng.random((nz, ny, nx)) < 0.35)).astype(np.float32)
        if rng.random() < 0.5:                                    # a specimen in empty space: planes at both ends empty
            h = int(rng.integers(1, nz // 2 + 1))                 # ... around the rotation centre, so that it stays inside after the rotation
            gt[:nz // 2 - h] = 0; gt[nz // 2 + h:] = 0
        psf = (rng.random((kz, ky, kx), dtype=np.float32) + 0.05).astype(np.float32)
        p = ctx.view_params(degrees=int(rng.integers(0, 360)), inc=inc, snr=-1.0, seed=7, stream=1, conv_method=1)
        d_gt = ctx.dev_alloc(gt.nbytes); ctx.upload(d_gt, gt)
        nzo = (nz - 1) // inc + 1
        d_acq = ctx.dev_alloc(nzo * ny * nx * 4)
        res = {}
        for name, opts in (("all", dict(zconv_strided=0, exp=0, skip_empty=1)), ("strided", dict(zconv_strided=1, exp=EXP, skip_empty=1)),
                           ("strided_noskip", dict(zconv_strided=1, exp=EXP, skip_empty=0))):
            for k, v in opts.items():
                ctx.set_option(k, v)
            for _ in range(2):                                    # twice: the adaptive plane flags take a view to settle
                info = ctx.simulate_view_dev(d_gt, (nx, ny, nz), psf.copy(), p, d_acq)
            res[name] = (ctx.download(d_acq, (nzo, ny, nx)), info)
        ctx.dev_free(d_gt); ctx.dev_free(d_acq)
        ref, got = res["all"][0], res["strided"][0]
        if not np.isfinite(ref).all():                            # the rotated volume is empty: adjustImage divides by a zero mean, as the reference does
            degenerate += 1
            continue
        scale = float(np.abs(ref).max()) or 1.0
        err = float(np.abs(got - ref).max()) / scale
        same = np.array_equal(res["strided"][0], res["strided_noskip"][0])
        ok = err <= 3e-6 and same and np.isfinite(got).all()
        if not ok:
            bad += 1
            fin = {k: bool(np.isfinite(v[0]).all()) for k, v in res.items()}
            nanz = sorted(set(np.argwhere(~np.isfinite(got))[:, 0].tolist()))[:12]
            print(f"FAIL case {case}: {nx}x{ny}x{nz} psf {kx}x{ky}x{kz} inc {inc}: err {err:.2e} skip_identical {same} finite {fin} "
                  f"nan planes {nanz} gt planes non-empty {int((gt.reshape(nz, -1) != 0).any(1).sum())}/{nz}", flush=True)
    ctx.close()
    if verbose:
        print(f"{n_cases} cases, {bad} failures, {degenerate} skipped (empty after the rotation)")
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 1,
                      int(sys.argv[3]) if len(sys.argv) > 3 else 2) else 0)
