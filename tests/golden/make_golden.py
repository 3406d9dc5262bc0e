"""Generates the committed golden fixtures from the CPU oracle.

The reference (Java) cannot run in the build container and ships no golden vectors, so these are
(a) JDK-specification answers for java.util.Random (also listed in SURVEY.md Appendix B), computed
by the oracle's restatement and cross-checked against the published values, and (b) a small
end-to-end view produced by the oracle (regression pin for oracle AND HIP path).

    python tests/golden/make_golden.py
"""
import importlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import oracle as O  # noqa: E402

synth = importlib.import_module("multiview-simulation_amd.synthetic")


def jdk():
    seed = 464232194
    r = O.JRandom(seed); nd = [r.nextDouble() for _ in range(4)]
    r = O.JRandom(seed); ni = [r.nextInt(20) for _ in range(6)]
    r = O.JRandom(seed); po = [r.poisson(2.5) for _ in range(12)]
    r = O.JRandom(seed); nl = [r.nextLong() for _ in range(3)]
    d = {"seed": seed, "random0_nextInt": O.JRandom(0).nextInt(), "nextDouble4": nd, "nextInt20_6": ni,
         "poisson_mean2.5_12": po, "nextLong3": nl, "mul_snr25": O.poisson_mul(25.0)}
    # published values (from the JDK specification)
    assert d["random0_nextInt"] == -1155484576
    assert nd == [0.4143130143281428, 0.9731632560980291, 0.6356592534797139, 0.45751024578762167]
    assert ni == [14, 6, 19, 4, 11, 9]
    assert po == [4, 2, 2, 2, 2, 3, 3, 1, 5, 3, 1, 4]
    json.dump(d, open(os.path.join(HERE, "jdk_vectors.json"), "w"), indent=1)


def view():
    gt = synth.sphere_phantom(24)
    psf_raw = synth.gaussian_psf(7, 7, 9, sigma=(1.2, 1.4, 2.5))
    psf = psf_raw.copy()
    p = dict(degrees=15 + 45, delta=0.01, inc=3, snr=25.0, seed=464232194, stream=1)
    res = O.simulate_view(gt, psf, p["degrees"], delta=p["delta"], inc=p["inc"], snr=p["snr"], seed=p["seed"],
                          stream=p["stream"])
    np.savez_compressed(os.path.join(HERE, "view_24.npz"), gt=gt, psf_raw=psf_raw, psf_norm=psf, rot=res["rot"],
                        att=res["att"], con=res["con"], acq=res["acq"], corr=res["corr"], **p)


def phantom():
    """drawSpheres + downSample2x (SMVD:394-522) on a 260^3 canvas at scale 2, the reference's seed."""
    import hashlib
    canvas, scale, seed = 260, 2, 464232194
    img = np.zeros((canvas,) * 3, np.float32)
    r = O.JRandom(seed)
    n = O.draw_spheres(img, 0.0, 1.0, scale, False, r)
    ds = O.downsample2x(img)
    d = {"canvas": canvas, "scale": scale, "seed": seed, "n_spheres": n, "rnd_state_after": int(r.st.s),
         "canvas_sha256": hashlib.sha256(img.tobytes()).hexdigest(),
         "downsampled_sha256": hashlib.sha256(ds.tobytes()).hexdigest(),
         "downsampled_nonzero": int((ds > 0).sum())}
    json.dump(d, open(os.path.join(HERE, "phantom_vectors.json"), "w"), indent=1)


if __name__ == "__main__":
    jdk()
    view()
    phantom()
    print("golden fixtures written to", HERE)
