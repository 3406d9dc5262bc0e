"""The Poisson work queue against its size (option poisson_queue_share): bytes held, what one view queues, what full segments refuse,
and the time of the extract + Poisson stage and of the whole view, on the bench's two volumes (512^3 sphere phantom and the same without
an empty voxel, 31^3 PSF, SNR 25) and on a volume built to overflow every share (every voxel in the inversion regime with a count >= 1
possible).  One context per share, so that `bytes` is what THAT share reserves.
    python tools/queue_share.py [size]
"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mvs = importlib.import_module("multiview-simulation_amd")
synth = importlib.import_module("multiview-simulation_amd.synthetic")

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
k = 31
phantom = synth.sphere_phantom(n)
volumes = {
    "phantom": phantom,
    "dense (phantom + 0.5)": phantom + np.float32(0.5),
    "every voxel 1 < lambda < 10 (constant 0.03)": np.full_like(phantom, 0.03),
}
psf = synth.gaussian_psf(k)
print(f"{n}^3, PSF {k}^3, inc 1, SNR 25: one view, device-resident; times are HIP-event stage times of the library (mean of 6 views)")
for name, gt in volumes.items():
    print(f"-- {name}")
    ref = None
    for share in ("16", "auto", "3", "1"):
        ctx = mvs.Context(0)
        ctx.set_option("poisson_queue_share", share)
        ctx.set_option("tail_overlap", 0); ctx.set_option("psf_overlap", 0)
        d_gt = ctx.dev_alloc(gt.nbytes); ctx.upload(d_gt, gt)
        d_acq = ctx.dev_alloc(gt.nbytes)
        # adjustImage brings the mean to 1, so lambda = v / mean * 125; the constant volume keeps its regime through target_average
        target = 0.03 if name.startswith("every") else 1.0
        p = ctx.view_params(degrees=60, delta=0.01, min_value=0.0, target_average=target, inc=1, snr=25.0, seed=1, stream=0, conv_method=1)
        for _ in range(2):
            ctx.simulate_view_dev(d_gt, (n, n, n), psf.copy(), p, d_acq)
        ctx.synchronize()
        ctx.enable_timing(True)
        for _ in range(6):
            ctx.simulate_view_dev(d_gt, (n, n, n), psf.copy(), p, d_acq)
        t = ctx.timings()
        ctx.enable_timing(False)
        st = ctx.queue_stats()
        acq = ctx.download(d_acq, (n, n, n))
        if ref is None:
            ref = acq
        same = bool(np.array_equal(acq, ref))
        nv = n ** 3
        print(f"   share {share:>4}: queue {st['bytes'] / 2**30:6.3f} GiB, segment {st['segment_items']:5d} items; queued {100 * (st['bright'] + st['inversion']) / nv:5.1f} % "
              f"of the voxels ({100 * st['bright'] / nv:4.1f} bright + {100 * st['inversion'] / nv:4.1f} inversion), refused {100 * st['refused'] / nv:5.1f} %, fullest block {st['fullest_block']} pending; "
              f"extract+Poisson {t.get('extract_ms', float('nan')):.3f} ms, view {t['total_ms']:.3f} ms; counts {'identical' if same else 'DIFFER'}")
        ctx.dev_free(d_gt); ctx.dev_free(d_acq)
        ctx.close()
        assert same
