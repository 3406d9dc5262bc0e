"""Views that cannot fill the chip one at a time: mvsim_simulate_views_dev (V views of one ground truth in one call, `view_lanes`
of them side by side) against V sequential mvsim_simulate_view_dev calls, at the sizes the reference itself runs.

    python tools/small_views.py [case ...] [lanes=1,2,4,8] [reps=20] [graph=0|1]

Cases: c0 = BASELINE configs[0] (128^3, 15^3 PSF, inc 1, 8 views), ref = the reference's own run (289^3, 51^3 PSF stack, inc 3, 7 views:
SimulateMultiViewDataset.java:376-380,399,531-548), 256 = 256^3 / 31^3 / inc 1 / 8 views, 512 = the headline view (control).
Prints wall-clock ms per view and Gvoxel/s per setting and checks that every acquisition is bit-identical to the sequential one."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mvs = importlib.import_module("multiview-simulation_amd")
synth = importlib.import_module("multiview-simulation_amd.synthetic")

CASES = {"c0": (128, 15, 1, 8), "ref": (289, 51, 3, 7), "256": (256, 31, 1, 8), "512": (512, 31, 1, 8), "64": (64, 9, 1, 8)}
names, lanes, reps, extra, only = [], [1, 2, 4, 8], 20, [], None
for a in sys.argv[1:]:
    if a.startswith("lanes="): lanes = [int(x) for x in a[6:].split(",")]
    elif a.startswith("only="): only = a[5:]              # only=stacked | only=sequential: one form alone (profiles)
    elif a.startswith("reps="): reps = int(a[5:])
    elif "=" in a: extra.append(a.split("=", 1))
    else: names.append(a)
names = names or ["c0", "ref", "256"]

for name in names:
    n, k, inc, nv = CASES[name]
    gt = synth.sphere_phantom(n)
    nzo = (n - 1) // inc + 1
    psfs = [synth.gaussian_psf(k, sigma=(k / 15.0, k / 14.0, k / 5.0 + 0.05 * v)) for v in range(nv)]
    with mvs.Context(0) as ctx:
        for kv in extra:
            ctx.set_option(*kv)
        d_gt = ctx.dev_alloc(gt.nbytes); ctx.upload(d_gt, gt)
        acq = [ctx.dev_alloc(nzo * n * n * 4) for _ in range(nv)]
        params = [ctx.view_params(degrees=15 + (360 * v) // nv, inc=inc, snr=25.0, seed=464232194, stream=v, conv_method=1) for v in range(nv)]
        dim = (n, n, n)

        def sequential():
            for v in range(nv):
                ctx.simulate_view_dev(d_gt, dim, psfs[v].copy(), params[v], acq[v])

        def batched():
            ctx.simulate_views_dev(d_gt, dim, [p.copy() for p in psfs], params, acq)

        def clock(fn):
            for _ in range(3):
                fn()
            ctx.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            t_issue = time.perf_counter() - t0
            ctx.synchronize()
            return (time.perf_counter() - t0) / (reps * nv) * 1e3, t_issue / (reps * nv) * 1e3

        if only == "stacked":
            ctx.set_option("view_batch", 1)
            ms, issue = clock(batched)
            print(f"{name}: stacked only: {ms:7.3f} ms/view = {n ** 3 / ms / 1e6:6.1f} Gvoxel/s")
            continue
        ms, issue = clock(sequential)
        if only == "sequential":
            print(f"{name}: sequential only: {ms:7.3f} ms/view = {n ** 3 / ms / 1e6:6.1f} Gvoxel/s")
            continue
        want = [ctx.download(a, (nzo, n, n)) for a in acq]
        print(f"{name}: {n}^3, PSF {k}^3, inc {inc}, {nv} views  {' '.join('='.join(kv) for kv in extra)}")
        print(f"  sequential simulate_view_dev : {ms:7.3f} ms/view = {n ** 3 / ms / 1e6:6.1f} Gvoxel/s   (host issue {issue * 1e3:5.0f} us/view)")
        ctx.set_option("view_batch", 1)
        ms, issue = clock(batched)
        got = [ctx.download(a, (nzo, n, n)) for a in acq]
        same = all(np.array_equal(a, b) for a, b in zip(got, want))
        print(f"  simulate_views_dev, stacked   : {ms:7.3f} ms/view = {n ** 3 / ms / 1e6:6.1f} Gvoxel/s   (host issue {issue * 1e3:5.0f} us/view)"
              f"   {'bit-identical' if same else 'DIFFERS'}")
        ctx.set_option("view_batch", 0)
        for L in lanes:
            ctx.set_option("view_lanes", L)
            ms, issue = clock(batched)
            got = [ctx.download(a, (nzo, n, n)) for a in acq]
            same = all(np.array_equal(a, b) for a, b in zip(got, want))
            print(f"  simulate_views_dev, {L} lanes   : {ms:7.3f} ms/view = {n ** 3 / ms / 1e6:6.1f} Gvoxel/s   (host issue {issue * 1e3:5.0f} us/view)"
                  f"   {'bit-identical' if same else 'DIFFERS'}")
        for a in acq + [d_gt]:
            ctx.dev_free(a)
