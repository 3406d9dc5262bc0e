"""Known-answer tests that pin the CPU oracle (oracle/) -- no GPU needed.

The reference ships no tests or golden vectors, so the oracle is pinned by
(a) JDK-specified java.util.Random answers, (b) Random123 Philox answers, (c) analytic properties of
each stage that follow from the reference source, (d) an independent float64 scipy derivation.
"""
import json
import math
import os

import numpy as np
import pytest

from .conftest import rel_to_max


# ------------------------------------------------------------------------------------------------ RNG
def test_jdk_random_known_answers(orc, golden_dir):
    kat = json.load(open(os.path.join(golden_dir, "jdk_vectors.json")))
    assert orc.JRandom(0).nextInt() == kat["random0_nextInt"] == -1155484576
    r = orc.JRandom(kat["seed"])
    assert [r.nextDouble() for _ in range(4)] == kat["nextDouble4"]
    r = orc.JRandom(kat["seed"])
    assert [r.nextInt(20) for _ in range(6)] == kat["nextInt20_6"]
    r = orc.JRandom(kat["seed"])
    assert [r.poisson(2.5) for _ in range(12)] == kat["poisson_mean2.5_12"]
    assert orc.poisson_mul(25.0) == kat["mul_snr25"] == 124.99999999999997


def test_jdk_random_nextint_power_of_two_and_long(orc):
    # JDK spec: power-of-two bounds take the high bits; nextLong = (next(32) << 32) + next(32)
    r1, r2 = orc.JRandom(42), orc.JRandom(42)
    assert r1.nextInt(16) == (16 * r2.next(31)) >> 31
    r1, r2 = orc.JRandom(7), orc.JRandom(7)
    hi, lo = r2.next(32), r2.next(32)
    want = ((hi << 32) + lo) & 0xFFFFFFFFFFFFFFFF
    want = want - (1 << 64) if want >> 63 else want
    assert r1.nextLong() == want


def test_philox_random123_known_answers(orc):
    assert orc.philox4x32_10((0, 0, 0, 0), (0, 0)) == (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)
    assert orc.philox4x32_10((0xFFFFFFFF,) * 4, (0xFFFFFFFF,) * 2) == (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)
    assert orc.philox4x32_10((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0)) == \
        (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1)


def test_det_functions_close_to_libm(orc):
    rng = np.random.default_rng(1)
    for x in np.concatenate([rng.random(2000) * 1e-3, rng.random(2000), rng.random(2000) * 1e6, [1.0, 2.0, 0.5]]):
        assert abs(orc.det_log(x) - math.log(x)) <= 4e-16 * max(1.0, abs(math.log(x)))
    for x in -rng.random(2000) * 10:
        assert abs(orc.det_exp(x) - math.exp(x)) <= 4e-16 * math.exp(x)
        assert abs(orc.det_exp_neg(-x) - math.exp(x)) <= 6e-16 * math.exp(x)
    for k in list(range(0, 300)) + [1000, 10 ** 5, 10 ** 7]:
        assert abs(orc.det_lgamma_int(k) - math.lgamma(k + 1)) <= 1e-13 * max(1.0, math.lgamma(k + 1))


@pytest.mark.parametrize("lam", [0.0125, 1.0, 9.99, 10.0, 125.0, 4000.0])
def test_counter_poisson_moments(orc, lam):
    n = 60000
    s = np.array([orc.poisson_counter(lam, 99, 3, i) for i in range(n)], dtype=np.float64)
    se_mean = math.sqrt(lam / n)
    assert abs(s.mean() - lam) < 5 * se_mean
    se_var = lam * math.sqrt(2.0 / n + 1.0 / (lam * n))
    assert abs(s.var() - lam) < 5 * se_var
    assert orc.poisson_counter(0.0, 1, 0, 0) == 0 and orc.poisson_counter(-3.0, 1, 0, 0) == 0
    assert orc.poisson_counter(float("nan"), 1, 0, 0) == 0


def test_counter_poisson_matches_reference_sampler_distribution(orc):
    """Same distribution as uncommons PoissonGenerator (two-sample chi-square on the histogram)."""
    lam, n = 7.5, 40000
    a = np.array([orc.poisson_counter(lam, 5, 0, i) for i in range(n)])
    r = orc.JRandom(464232194)
    b = np.array([r.poisson(lam) for _ in range(n)])
    hi = 25
    ha = np.bincount(np.minimum(a, hi), minlength=hi + 1).astype(float)
    hb = np.bincount(np.minimum(b, hi), minlength=hi + 1).astype(float)
    m = (ha + hb) > 10
    chi2 = np.sum((ha[m] - hb[m]) ** 2 / (ha[m] + hb[m]))
    assert chi2 < 60.0, chi2   # ~ 20 dof; 60 is far in the tail


# ------------------------------------------------------------------------------------------------ rotate
def test_axis_rotation_matrix(orc):
    # even size: centre uses integer division (Q2): 512 -> 255
    m = orc.axis_rotation((512, 512, 512), 0, 0)
    assert np.array_equal(m, np.hstack([np.eye(3), np.zeros((3, 1))]))
    m = orc.axis_rotation((512, 512, 512), 0, 90)
    th = float(np.float32(math.radians(90)))   # Q3: float radians
    c, s = math.cos(th), math.sin(th)
    want = np.array([[1, 0, 0, 0], [0, c, -s, 255 - 255 * c + 255 * s], [0, s, c, 255 - 255 * s - 255 * c]])
    assert np.allclose(m, want, atol=1e-12)
    assert -4.4e-8 < c < -4.3e-8                       # cos of float(pi/2) is not 0
    inv = orc.affine_invert(m)
    full = np.vstack([m, [0, 0, 0, 1]]) @ np.vstack([inv, [0, 0, 0, 1]])
    assert np.allclose(full, np.eye(4), atol=1e-9)


def test_rotate_zero_is_identity_bit_exact(orc):
    v = np.random.default_rng(0).random((7, 9, 8), dtype=np.float32)
    for axis in (0, 1, 2):
        assert np.array_equal(orc.rotate_around_axis(v, axis, 0), v)
        assert np.array_equal(orc.rotate_around_axis(v, axis, 360), orc.rotate_around_axis(v, axis, 360))


def test_rotate_single_voxel_pins_centre_and_sign(orc):
    n = 9
    v = np.zeros((n, n, n), np.float32)
    v[4, 5, 4] = 1.0                       # (x=4, y=5, z=4): +1 in y from the centre
    out = orc.rotate_around_axis(v, 0, 90)  # forward model sends (dy=1) to (dz=+1)
    assert out[5, 4, 4] > 0.999999 and abs(out.sum() - 1.0) < 1e-5
    out = orc.rotate_around_axis(v, 2, 90)  # about z: (x, y) -> (-y, x): dy=1 -> dx=-1
    assert out[4, 4, 3] > 0.999999
    out = orc.rotate_around_axis(v, 1, 90)  # about y: (x, z) -> (z, -x); dy unaffected
    assert out[4, 5, 4] > 0.999999
    # even size: centre = (8-1)//2 = 3, so a voxel at the centre stays put under any rotation
    w = np.zeros((8, 8, 8), np.float32)
    w[3, 3, 3] = 1.0
    assert orc.rotate_around_axis(w, 0, 90)[3, 3, 3] > 0.999999


def test_rotate_quarter_turns_permute_axes(orc):
    v = np.random.default_rng(3).random((9, 9, 9), dtype=np.float32)
    out = orc.rotate_around_axis(v, 0, 90)
    # out[z, y, x] = in(y' = cy + (z - cz), z' = cz - (y - cy))
    want = np.transpose(v, (1, 0, 2))[::-1, :, :]   # want[z,y,x] = v[8-y... built explicitly below
    want = np.empty_like(v)
    for z in range(9):
        for y in range(9):
            want[z, y, :] = v[4 - (y - 4), 4 + (z - 4), :]
    assert rel_to_max(out, want) < 1e-5
    assert rel_to_max(orc.rotate_around_axis(v, 0, 180), v[::-1, ::-1, :]) < 1e-5


def test_rotate_matches_independent_scipy_derivation(orc):
    import scipy.ndimage as ndi
    v = np.random.default_rng(4).random((12, 14, 10), dtype=np.float32)
    for axis, deg in ((0, 37), (1, -20), (2, 123)):
        out = orc.rotate_around_axis(v, axis, deg)
        inv = orc.affine_invert(orc.axis_rotation((10, 14, 12), axis, deg))
        perm = [2, 1, 0]                     # (x,y,z) -> numpy (z,y,x)
        A = inv[:, :3][np.ix_(perm, perm)]
        off = inv[:, 3][perm]
        want = ndi.affine_transform(v.astype(np.float64), A, offset=off, order=1, mode="grid-constant", cval=0.0)
        assert rel_to_max(out, want) < 2e-6


def test_rotate_there_and_back_is_smooth_identity(orc, synth):
    v = synth.sphere_phantom(24)
    import scipy.ndimage as ndi
    v = ndi.gaussian_filter(v, 2.0).astype(np.float32)
    back = orc.rotate_around_axis(orc.rotate_around_axis(v, 0, 30), 0, -30)
    zz, yy = np.ogrid[:24, :24]
    inside = ((zz - 11) ** 2 + (yy - 11) ** 2) <= 9 ** 2
    err = np.abs(back - v)[inside].max()
    assert err < 0.08 * v.max()


# ------------------------------------------------------------------------------------------------ attenuate
def test_attenuate_kats(orc):
    rng = np.random.default_rng(5)
    v = rng.random((5, 8, 8), dtype=np.float32)
    assert np.array_equal(orc.attenuate3d(v, 0.0), v)                      # delta 0 -> identity
    c = np.full((3, 6, 6), 0.5, np.float32)
    out = orc.attenuate3d(c, 0.1)
    for k in range(6):                                                      # k counted from y = Ny-1 down
        want = 0.5 * (1 - 0.5 * 0.1) ** (k + 1)
        assert np.allclose(out[:, 5 - k, :], want, rtol=1e-6)
    assert np.all(orc.attenuate3d(np.full((2, 4, 4), 2.0, np.float32), 0.5) == 0)   # v*delta >= 1 -> 0
    # direction: a bright slab at high y darkens lower y, not vice versa
    s = np.full((2, 8, 8), 0.1, np.float32)
    s[:, 6, :] = 5.0
    o = orc.attenuate3d(s, 0.05)
    assert np.allclose(o[:, 7, :], 0.1 * (1 - 0.1 * 0.05))
    assert np.all(o[:, 5, :] < 0.1 * (1 - 0.1 * 0.05) ** 2)
    # Q1: steps = Nx; Nx < Ny leaves low-y rows untouched (zero), Nx > Ny is rejected
    r = orc.attenuate3d(np.ones((2, 8, 4), np.float32), 0.01)
    assert np.all(r[:, :4, :] == 0) and np.all(r[:, 4:, :] > 0)
    with pytest.raises(ValueError):
        orc.attenuate3d(np.ones((2, 4, 8), np.float32), 0.01)


# ------------------------------------------------------------------------------------------------ convolve
def _delta_psf(k, shift=(0, 0, 0)):
    p = np.zeros((k, k, k), np.float32)
    p[k // 2 + shift[0], k // 2 + shift[1], k // 2 + shift[2]] = 3.0   # un-normalised on purpose
    return p


@pytest.mark.parametrize("conv", ["direct", "fft"])
def test_convolve_delta_and_shift(orc, conv):
    f = orc.convolve_direct if conv == "direct" else orc.convolve_fft
    v = np.random.default_rng(6).random((10, 11, 12), dtype=np.float32)
    psf = _delta_psf(5)
    out = f(v, psf)
    assert abs(psf.sum() - 1.0) < 1e-6 and psf.max() == 1.0           # normalised IN PLACE (Q5)
    assert rel_to_max(out, v) < (1e-7 if conv == "direct" else 2e-6)
    # delta at K/2 + s  ->  out[x] = in_mirror[x - s]: pins centre convention, no flip, mirror-single
    s = (1, -2, 2)                                                      # (sz, sy, sx)
    out = f(v, _delta_psf(5, s))
    idx = [np.arange(n) - d for n, d in zip(v.shape, s)]
    idx = [np.where(i < 0, -i, np.where(i >= n, 2 * n - 2 - i, i)) for i, n in zip(idx, v.shape)]
    want = v[np.ix_(*idx)]
    assert rel_to_max(out, want) < (1e-7 if conv == "direct" else 2e-6)


@pytest.mark.parametrize("conv", ["direct", "fft"])
def test_convolve_constant_and_linearity(orc, synth, conv):
    f = orc.convolve_direct if conv == "direct" else orc.convolve_fft
    c = np.full((9, 9, 9), 2.5, np.float32)
    out = f(c, synth.gaussian_psf(5, sigma=(1, 1.2, 1.5)))
    assert np.allclose(out, 2.5, rtol=2e-6)
    rng = np.random.default_rng(7)
    a, b = rng.random((8, 9, 10), dtype=np.float32), rng.random((8, 9, 10), dtype=np.float32)
    psf = synth.gaussian_psf(5, 3, 7, sigma=(1.0, 0.8, 2.0))
    lhs = f(a + b, psf.copy())
    rhs = f(a, psf.copy()) + f(b, psf.copy())
    assert rel_to_max(lhs, rhs) < 3e-6


def test_convolve_direct_vs_fft_and_scipy(orc, synth):
    import scipy.ndimage as ndi
    v = synth.sphere_phantom(20)
    psf = synth.gaussian_psf(7, 5, 9, sigma=(1.5, 1.0, 2.5))
    d = orc.convolve_direct(v, psf.copy())
    f = orc.convolve_fft(v, psf.copy())
    assert rel_to_max(f, d) < 2e-6
    pn = psf.astype(np.float64) / psf.astype(np.float64).sum()
    want = ndi.convolve(v.astype(np.float64), pn, mode="mirror")
    assert rel_to_max(d, want) < 5e-7
    # even kernel sizes: centre index K/2, direct and FFT restatements agree
    pe = np.random.default_rng(8).random((4, 6, 2), dtype=np.float32)
    assert rel_to_max(orc.convolve_fft(v, pe.copy()), orc.convolve_direct(v, pe.copy())) < 2e-6


def test_jtk_fast_sizes(orc):
    assert orc.jtk_nfft_fast(542) == 546 and orc.jtk_nfft_fast(1) == 1 and orc.jtk_nfft_fast(17) == 18
    assert orc.jtk_nfft_fast(720720) == 720720


# ------------------------------------------------------------------------------------------------ adjust / norm
def test_norm_and_adjust(orc):
    rng = np.random.default_rng(9)
    p = rng.random((5, 5, 5), dtype=np.float32)
    orc.norm_image(p)
    assert abs(float(p.astype(np.float64).sum()) - 1.0) < 1e-6
    a = rng.random((6, 7, 8), dtype=np.float32) * 3
    a0 = a.copy()
    corr = orc.adjust_image(a, 1e-4, 1.0)
    mean0 = a0.astype(np.float64).mean()
    assert abs(corr - float(np.float32(1.0) - np.float32(1e-4)) / mean0) < 1e-12 * corr
    assert abs(a.astype(np.float64).mean() - 1.0) < 1e-6 and a.min() >= np.float32(1e-4)
    want = (a0.astype(np.float64) * corr).astype(np.float32) + np.float32(1e-4)   # two roundings (Q6)
    assert np.array_equal(a, want)


def test_sum_image_is_exact_to_double(orc):
    a = (np.random.default_rng(10).random(100001) * 1e3).astype(np.float32)
    assert abs(orc.sum_image(a) - math.fsum(a.astype(np.float64))) < 1e-6


# ------------------------------------------------------------------------------------------------ extract / Poisson
def test_extract_slices_copy_is_bit_exact(orc):
    v = np.random.default_rng(11).random((11, 5, 6), dtype=np.float32)
    for inc in (1, 2, 3, 4, 11, 50):
        out = orc.extract_slices_ref(v, inc, -1.0)
        assert out.shape == ((11 - 1) // inc + 1, 5, 6)
        assert np.array_equal(out, v[::inc])
        assert np.array_equal(orc.extract_slices_counter(v, inc, -1.0, 1), v[::inc])
    with pytest.raises(ValueError):
        orc.extract_slices_ref(v, 0, -1.0)


def test_extract_slices_reference_stream_order(orc):
    """Q10: one RNG stream, consumed slice by slice over extracted slices only, x fastest."""
    v = np.full((4, 2, 3), 0.02, np.float32)
    out = orc.extract_slices_ref(v, 2, 25.0, orc.JRandom(464232194))
    r = orc.JRandom(464232194)
    mul = orc.poisson_mul(25.0)
    want = np.array([r.poisson(float(np.float32(0.02)) * mul) for _ in range(2 * 2 * 3)], np.float32).reshape(2, 2, 3)
    assert np.array_equal(out, want)
    # SNR = 0 -> lambda = 0 -> zeros (Q8); raw counts are not rescaled (Q7)
    big = np.full((1, 4, 4), 2.0, np.float32)
    c = orc.extract_slices_counter(big, 1, 25.0, 7)
    assert 150 < c.mean() < 350 and np.all(c == np.round(c))
    assert np.all(orc.extract_slices_counter(big, 1, 0.0, 7) == 0)


def test_counter_stream_is_tiling_invariant(orc):
    """Counter = global source voxel index: extracting a z-slab separately gives the same counts."""
    v = np.random.default_rng(12).random((6, 4, 4), dtype=np.float32)
    full = orc.extract_slices_counter(v, 1, 25.0, 123, 2)
    again = orc.extract_slices_counter(v, 1, 25.0, 123, 2)
    assert np.array_equal(full, again)
    other = orc.extract_slices_counter(v, 1, 25.0, 123, 3)
    assert not np.array_equal(full, other)


# ------------------------------------------------------------------------------------------------ next items
def test_plane_and_window_forms_equal_the_whole_volume(orc):
    """The forms the full-size GPU parity tests use: a range of output planes of rotateAroundAxis (SMVD:119-132: independent output
    voxels), attenuate3d on those planes (SMVD:335-359: the walk stays inside a plane of constant z) and the counter sampler on a
    window of planes with the full volume's source indices -- each equal, bit for bit, to the same planes of the whole-volume call."""
    rng = np.random.default_rng(5)
    v = rng.random((21, 26, 23), dtype=np.float32)
    for axis, deg in ((0, 60), (1, -33), (2, 15)):
        full = orc.rotate_around_axis(v, axis, deg)
        assert np.array_equal(orc.rotate_around_axis_planes(v, axis, deg, 6, 9), full[6:15])
        assert np.array_equal(orc.rotate_around_axis_planes(v, axis, deg, 0, 21), full)
    rot = orc.rotate_around_axis(v, 0, 60)
    assert np.array_equal(orc.attenuate3d(rot[6:15], 0.01), orc.attenuate3d(rot, 0.01)[6:15])
    with pytest.raises(ValueError):
        orc.rotate_around_axis_planes(v, 0, 60, 15, 9)
    lam = (rng.random((20, 8, 12), dtype=np.float32) * 40).astype(np.float32)
    for inc in (1, 4):
        whole = orc.extract_slices_counter(lam, inc, 25.0, 464232194, 2)
        assert np.array_equal(orc.extract_slices_counter_window(lam[8:17], inc, 25.0, 464232194, 2, 8), whole[8 // inc:8 // inc + (8 // inc) + 1])
    with pytest.raises(ValueError):
        orc.extract_slices_counter_window(lam[3:9], 4, 25.0, 1, 0, 3)


def test_make_isotropic_and_weights(orc):
    z = np.arange(5, dtype=np.float32)[:, None, None] * np.ones((1, 3, 4), np.float32)
    iso = orc.make_isotropic(z, 3)
    assert iso.shape == ((5 - 1) * 3 + 1, 3, 4)
    assert np.allclose(iso[:, 0, 0], np.arange(13) / 3.0, atol=1e-6)
    w = orc.compute_weight_image((2, 100, 3))
    assert np.all(w[:, 51:, :] == 1.0)          # l = 99 - y < 50
    assert np.all(w[:, :9, :] == 0.0)           # l > 90
    assert abs(w[0, 29, 0] - 0.5) < 1e-6        # l = 70: halfway down the cosine


# ------------------------------------------------------------------------------------------------ golden fixtures
def test_golden_view_fixture_reproduces(orc, golden_dir):
    g = np.load(os.path.join(golden_dir, "view_24.npz"))
    psf = g["psf_raw"].copy()
    res = orc.simulate_view(g["gt"], psf, int(g["degrees"]), delta=float(g["delta"]), inc=int(g["inc"]),
                            snr=float(g["snr"]), seed=int(g["seed"]), stream=int(g["stream"]))
    for k in ("rot", "att", "con", "acq"):
        assert np.array_equal(res[k], g[k]), k
    assert res["corr"] == float(g["corr"])
    assert np.array_equal(psf, g["psf_norm"])


def test_normalize_weights_kats(orc):
    """SMVD:615-640: per-voxel float sum over views; zero sum -> zeros; else min(1, osem * w / sum)."""
    w = [np.array([0.0, 0.5, 1.0, 0.2], np.float32), np.array([0.0, 0.5, 1.0, 0.0], np.float32),
         np.array([0.0, 0.0, 1.0, 0.0], np.float32)]
    orc.normalize_weights(w, 3.0)
    assert np.array_equal(w[0], np.array([0.0, 1.0, 1.0, 1.0], np.float32))     # 3*0.5 = 1.5 -> clipped to 1
    assert np.array_equal(w[1], np.array([0.0, 1.0, 1.0, 0.0], np.float32))
    assert np.array_equal(w[2], np.array([0.0, 0.0, 1.0, 0.0], np.float32))
    rng = np.random.default_rng(20)
    ws = [rng.random(1000, dtype=np.float32) for _ in range(7)]
    ref = [x.copy() for x in ws]
    orc.normalize_weights(ws, 3.0)
    s = np.zeros(1000, np.float32)
    for x in ref:
        s = s + x
    for a, b in zip(ws, ref):
        assert np.array_equal(a, np.minimum(np.float32(1), np.float32(3.0) * (b / s)))


# ------------------------------------------------------------------------------------------------ phantom (8f rank 3)
def _nested_sphere_count(R):
    import math
    n = 0
    for dz in range(-R, R + 1):
        r1 = math.isqrt(R * R - dz * dz)
        for dy in range(-r1, r1 + 1):
            n += 2 * math.isqrt(r1 * r1 - dy * dy) + 1
    return n


def test_hypersphere_cursor_counts(orc):
    """HyperSphereCursor restated as a state machine == closed form of the nested truncated radii."""
    assert [orc.hypersphere_size(r) for r in (0, 1, 2)] == [1, 7, 25]
    for r in (3, 5, 10, 20, 33):
        assert orc.hypersphere_size(r) == _nested_sphere_count(r)


def test_draw_spheres_random_stream_and_geometry(orc, mvs):
    """SMVD:436-522: per voxel of the large sphere nextInt(10*scale) and nextDouble, a third draw only for the
    voxels that become centres; replayed with the independent pure-Python java.util.Random of the facade."""
    n, scale, seed = 128, 1, 464232194
    img = np.zeros((n, n, n), np.float32)
    r = orc.JRandom(seed)
    drawn = orc.draw_spheres(img, 0.0, 1.0, scale, False, r)
    R = n // 2 - 47 * scale - 1
    j = mvs.JavaRandom(seed)
    expect, vmax = 0, 0.0
    for _ in range(_nested_sphere_count(R)):
        j.nextInt(10 * scale)
        rv = j.nextDouble()
        if int(np.floor(rv * 10000 + 0.5)) % (7 * scale) ** 3 == 0:
            vmax = max(vmax, j.nextDouble())
            expect += 1
    assert drawn == expect and drawn > 20
    assert r.st.s == j._s                                   # both generators consumed the same number of draws
    assert img.max() == np.float32(vmax) and img.min() == 0.0
    zz, yy, xx = np.nonzero(img)
    c = n // 2
    assert max(np.abs(zz - c).max(), np.abs(yy - c).max(), np.abs(xx - c).max()) <= R + 10 * scale
    # halfPixelOffset moves every small sphere by (+1, +1, 0) and nothing else
    img2 = np.zeros_like(img)
    orc.draw_spheres(img2, 0.0, 1.0, scale, True, orc.JRandom(seed))
    assert np.array_equal(img2[:, 1:, 1:], img[:, :-1, :-1])


def test_draw_spheres_single_sphere_is_a_hypersphere(orc):
    """Exactly one small sphere of radius r covers hypersphere_size(r) voxels: use a canvas whose large sphere has
    radius 0 (one voxel, one draw) and search seeds until that voxel becomes a centre."""
    n = 2 * (47 + 1)          # radius_large = n/2 - 47 - 1 = 0
    for seed in range(5000):
        img = np.zeros((n, n, n), np.float32)
        r = orc.JRandom(seed)
        if orc.draw_spheres(img, 0.0, 1.0, 1, False, r) == 1:
            rad = orc.JRandom(seed).nextInt(10) + 1
            assert int((img > 0).sum()) == orc.hypersphere_size(rad)
            zz, yy, xx = np.nonzero(img)
            assert zz.min() == n // 2 - rad and zz.max() == n // 2 + rad
            assert len(np.unique(img[img > 0])) == 1
            return
    raise AssertionError("no seed drew the single sphere")


def test_downsample2x_kats(orc):
    """SMVD:394-424: dims N/2-1, samples at 2l+0.5 (mean of the 2x2x2 block starting at 2l)."""
    rng = np.random.default_rng(5)
    v = rng.random((10, 13, 16), dtype=np.float32)
    d = orc.downsample2x(v)
    assert d.shape == (4, 5, 7)
    blk = v[:8, :10, :14].reshape(4, 2, 5, 2, 7, 2).astype(np.float64).mean(axis=(1, 3, 5))
    assert np.allclose(d, blk, rtol=0, atol=2e-7)
    assert np.array_equal(orc.downsample2x(np.full((8, 8, 8), 3.0, np.float32)), np.full((3, 3, 3), 3.0, np.float32))
    ramp = np.broadcast_to(np.arange(12, dtype=np.float32), (6, 6, 12)).copy()
    assert np.array_equal(orc.downsample2x(ramp)[0, 0], 2 * np.arange(5, dtype=np.float32) + 0.5)


def test_golden_phantom_fixture_reproduces(orc, golden_dir):
    import hashlib
    import json
    g = json.load(open(os.path.join(golden_dir, "phantom_vectors.json")))
    img = np.zeros((g["canvas"],) * 3, np.float32)
    r = orc.JRandom(g["seed"])
    assert orc.draw_spheres(img, 0.0, 1.0, g["scale"], False, r) == g["n_spheres"]
    assert int(r.st.s) == g["rnd_state_after"]
    assert hashlib.sha256(img.tobytes()).hexdigest() == g["canvas_sha256"]
    ds = orc.downsample2x(img)
    assert hashlib.sha256(ds.tobytes()).hexdigest() == g["downsampled_sha256"]
    assert int((ds > 0).sum()) == g["downsampled_nonzero"]
