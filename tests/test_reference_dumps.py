"""Oracle AND HIP path against outputs of the REFERENCE itself -- when somebody has produced them.

`java/harness/DumpReference.java` (run on a host with a JDK and the reference on the classpath, see its header) reads the raw
inputs of tests/golden/ref_in/ (written by tests/golden/make_golden.py, committed) and writes what
SimulateMultiViewDataset.* / Tools.* return into tests/golden/ref/.  With those files present this module is what pins the
oracle to the reference (until then parity is "unpinned"); without them every test here skips.
Nothing of the reference is needed at run time -- only its dumped outputs, which are data.
"""
import importlib
import os

import numpy as np
import pytest

import oracle

HERE = os.path.dirname(os.path.abspath(__file__))
REF_IN = os.path.join(HERE, "golden", "ref_in")
REF = os.environ.get("MVSIM_REF_DUMPS", os.path.join(HERE, "golden", "ref"))   # (override: rehearsing this module on stand-in dumps)


def _cases():
    out = []
    with open(os.path.join(REF_IN, "manifest.txt")) as fh:
        for line in fh:
            t = line.split()
            if not t or t[0].startswith("#"):
                continue
            out.append(dict(name=t[0], nx=int(t[1]), ny=int(t[2]), nz=int(t[3]), kx=int(t[4]), ky=int(t[5]), kz=int(t[6]), axis=int(t[7]),
                            degrees=int(t[8]), delta=float(t[9]), inc=int(t[10]), min_value=float(t[11]), target=float(t[12]),
                            snr=float(t[13]), seed=int(t[14])))
    return out


CASES = _cases()


def _have(c, *kinds):
    return all(os.path.exists(os.path.join(REF, f"{c['name']}.{k}")) for k in kinds)


def _ref(c, kind, shape=None, dtype="<f4"):
    a = np.fromfile(os.path.join(REF, f"{c['name']}.{kind}"), dtype=dtype)
    return a.reshape(shape) if shape is not None else a


def _inputs(c):
    gt = np.fromfile(os.path.join(REF_IN, c["name"] + ".gt.raw"), dtype="<f4").reshape(c["nz"], c["ny"], c["nx"])
    psf = np.fromfile(os.path.join(REF_IN, c["name"] + ".psf.raw"), dtype="<f4").reshape(c["kz"], c["ky"], c["kx"])
    return gt, psf


def test_reference_inputs_are_what_the_generator_writes():
    """The committed inputs are reproducible from make_golden.py (so a dump made from them belongs to this tree)."""
    mg = importlib.import_module("tests.golden.make_golden") if False else None  # the generator imports the package; load it by path
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    assert [c[0] for c in mg.REF_CASES] == [c["name"] for c in CASES]
    for rc, c in zip(mg.REF_CASES, CASES):
        gt, psf = mg.ref_case_inputs(*rc[:7])
        g, p = _inputs(c)
        assert np.array_equal(gt, g) and np.array_equal(psf, p), c["name"]


def _shape(c):
    return (c["nz"], c["ny"], c["nx"])


@pytest.mark.parametrize("c", CASES, ids=[c["name"] for c in CASES])
def test_oracle_against_reference_dumps(c):
    """Every stage of the oracle against what the reference returned for the same input (stage by stage: each stage starts from
    the REFERENCE's previous output, so that one deviation does not hide the next)."""
    if not _have(c, "rot.raw", "att.raw", "con.raw", "ext.raw"):
        pytest.skip("no reference dumps under tests/golden/ref (run java/harness/DumpReference.java on a host with a JDK)")
    gt, psf = _inputs(c)
    shp = _shape(c)
    m = oracle.axis_rotation((c["nx"], c["ny"], c["nz"]), c["axis"], c["degrees"])
    if _have(c, "affine.f64"):
        assert np.allclose(np.asarray(m).reshape(-1), _ref(c, "affine.f64", dtype="<f8"), rtol=0, atol=1e-9)
    rot_ref, att_ref = _ref(c, "rot.raw", shp), _ref(c, "att.raw", shp)
    rot = oracle.rotate_around_axis(gt, c["axis"], c["degrees"])
    assert np.array_equal(rot, rot_ref), f"rotateAroundAxis: {np.count_nonzero(rot != rot_ref)} voxels differ, max {np.abs(rot - rot_ref).max():.3e}"
    if c["nx"] <= c["ny"]:
        att = oracle.attenuate3d(rot_ref, c["delta"])
        assert np.array_equal(att, att_ref), f"attenuate3d: {np.count_nonzero(att != att_ref)} voxels differ, max {np.abs(att - att_ref).max():.3e}"
    pn = psf.copy()
    oracle.norm_image(pn)
    if _have(c, "psf_norm.raw"):
        assert np.array_equal(pn, _ref(c, "psf_norm.raw", psf.shape)), "normImage"
    if _have(c, "con_raw.raw"):
        con_raw_ref = _ref(c, "con_raw.raw", shp)
        con = oracle.convolve_direct(att_ref, pn)          # exact fp64 direct sum: the FFT result must sit within float32 FFT noise of it
        assert np.abs(con - con_raw_ref).max() <= 1e-5 * np.abs(con_raw_ref).max()
        adj = con_raw_ref.copy()
        corr = oracle.adjust_image(adj, c["min_value"], c["target"])
        assert np.array_equal(adj, _ref(c, "con.raw", shp)), "adjustImage"
        if _have(c, "corr.f64"):
            assert corr == pytest.approx(float(_ref(c, "corr.f64", dtype="<f8")[0]), rel=1e-12)
    con_ref = _ref(c, "con.raw", shp)
    nzo = oracle.extract_nz(c["nz"], c["inc"])
    ext_ref = _ref(c, "ext.raw", (nzo, c["ny"], c["nx"]))
    assert np.array_equal(oracle.extract_slices_ref(con_ref, c["inc"], -1.0), ext_ref), "extractSlices"
    if _have(c, "iso.raw"):
        iso = oracle.make_isotropic(ext_ref, c["inc"])
        iso_ref = _ref(c, "iso.raw", iso.shape)
        assert np.array_equal(iso, iso_ref), f"makeIsotropic: max {np.abs(iso - iso_ref).max():.3e}"
        if _have(c, "weight.raw"):
            w = oracle.compute_weight_image(iso.shape)
            assert np.abs(w - _ref(c, "weight.raw", iso.shape)).max() <= 6e-8
    if _have(c, "poisson.raw"):
        # the reference's own sampler on java.util.Random, restated (Tools.java:73-86, uncommons PoissonGenerator:95-109)
        noisy = oracle.extract_slices_ref(ext_ref, 1, c["snr"], oracle.JRandom(c["seed"]))
        pr = _ref(c, "poisson.raw", ext_ref.shape)
        assert np.count_nonzero(noisy != pr) <= 1, "Tools.poissonProcess (one Math.log ulp in ~1e16 draws may flip a count)"


@pytest.mark.gpu
@pytest.mark.parametrize("c", CASES, ids=[c["name"] for c in CASES])
def test_hip_path_against_reference_dumps(c):
    """The HIP path through the C ABI against the reference's outputs (the tolerance BASELINE.json states: 1e-5 relative for the
    floating-point stages, bit-exact for extractSlices indexing)."""
    if not _have(c, "rot.raw", "att.raw", "con.raw", "ext.raw"):
        pytest.skip("no reference dumps under tests/golden/ref (run java/harness/DumpReference.java on a host with a JDK)")
    mvs = importlib.import_module("multiview-simulation_amd")
    gt, psf = _inputs(c)
    shp = _shape(c)
    with mvs.Context(0) as ctx:
        rot = ctx.rotate_around_axis(gt, c["axis"], c["degrees"])
        rot_ref = _ref(c, "rot.raw", shp)
        assert np.abs(rot - rot_ref).max() <= 1e-5 * max(np.abs(rot_ref).max(), 1e-30)
        if c["nx"] <= c["ny"]:
            att = ctx.attenuate3d(rot_ref, c["delta"])
            att_ref = _ref(c, "att.raw", shp)
            assert np.abs(att - att_ref).max() <= 1e-5 * max(np.abs(att_ref).max(), 1e-30)
        if _have(c, "con_raw.raw"):
            con = ctx.convolve(_ref(c, "att.raw", shp), psf.copy(), method=1)
            cr = _ref(c, "con_raw.raw", shp)
            assert np.abs(con - cr).max() <= 1e-5 * np.abs(cr).max()
        con_ref = _ref(c, "con.raw", shp)
        ext = ctx.extract_slices(con_ref, c["inc"], -1.0, 0)
        assert np.array_equal(ext, _ref(c, "ext.raw", ext.shape))
        if c["axis"] == 0 and c["nx"] <= c["ny"]:
            p = ctx.view_params(axis=0, degrees=c["degrees"], delta=c["delta"], inc=c["inc"], snr=-1.0, seed=c["seed"], stream=0,
                                min_value=c["min_value"], target_average=c["target"], conv_method=1)
            res = ctx.simulate_view(gt, psf.copy(), p, want=("rot", "att", "con", "acq"))
            assert np.abs(res["con"] - con_ref).max() <= 1e-5 * np.abs(con_ref).max()
            assert np.abs(res["acq"] - _ref(c, "ext.raw", res["acq"].shape)).max() <= 1e-5 * np.abs(con_ref).max()
