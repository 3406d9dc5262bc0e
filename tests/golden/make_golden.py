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


REF_CASES = [
    # name        nx  ny  nz  kx ky kz axis deg  delta inc  min     target snr  seed
    ("c24",       24, 24, 24, 7, 7, 9, 0,   60,  0.01, 3,   1e-4,   1.0,   25., 464232194),
    ("a404420",   40, 44, 20, 5, 7, 9, 0,  -15,  0.01, 2,   1e-4,   1.0,   25., 464232194),
    ("ax1",       20, 24, 28, 5, 5, 5, 1,   30,  0.02, 1,   1e-4,   1.0,   10., 7),
    ("ax2",       20, 24, 28, 3, 5, 7, 2,  135,  0.005, 4,  1e-4,   2.0,   40., 11),
]


def ref_case_inputs(name, nx, ny, nz, kx, ky, kz):
    """The (deterministic) inputs of a reference-dump case: a sphere cloud cropped to the shape, an anisotropic Gaussian PSF
    with a small non-separable term, un-normalised."""
    n = max(nx, ny, nz)
    gt = np.ascontiguousarray(synth.sphere_phantom(n)[(n - nz) // 2:(n - nz) // 2 + nz, (n - ny) // 2:(n - ny) // 2 + ny,
                                                      (n - nx) // 2:(n - nx) // 2 + nx])
    psf = synth.gaussian_psf(kx, ky, kz, sigma=(0.9 + kx / 8, 1.0 + ky / 8, 1.2 + kz / 6)).astype(np.float64)
    z, y, x = np.meshgrid(np.arange(kz) - kz // 2, np.arange(ky) - ky // 2, np.arange(kx) - kx // 2, indexing="ij")
    psf = (psf * (1.0 + 0.25 * np.tanh(0.5 * x * z) + 0.1 * np.tanh(y))).astype(np.float32)
    return gt.astype(np.float32), psf


def ref_inputs():
    """Raw float32 inputs + manifest for java/harness/DumpReference.java (the reference run on a host with a JDK)."""
    d = os.path.join(HERE, "ref_in")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "manifest.txt"), "w") as fh:
        fh.write("# name nx ny nz kx ky kz axis degrees delta inc minValue targetAverage snr seed\n")
        for c in REF_CASES:
            name = c[0]
            gt, psf = ref_case_inputs(*c[:7])
            gt.astype("<f4").tofile(os.path.join(d, name + ".gt.raw"))
            psf.astype("<f4").tofile(os.path.join(d, name + ".psf.raw"))
            fh.write(" ".join(repr(v) if isinstance(v, float) else str(v) for v in c) + "\n")


if __name__ == "__main__":
    jdk()
    view()
    phantom()
    ref_inputs()
    print("golden fixtures written to", HERE)
