"""CPU ORACLE -- test infrastructure, NOT product code.

Python face of ``oracle/liborc.so`` (plain-C restatement of the reference's per-view
hot path, see ``mvsim_oracle.h``) plus a numpy/scipy restatement of the
imglib2-algorithm ``FFTConvolution`` recipe used by
``SimulateMultiViewDataset.convolve`` (SimulateMultiViewDataset.java:253-264).

PARITY UNPINNED: the reference has no tests / golden vectors for this path and could not be
executed here (Java, no JVM in the image).  Pinned pieces: the JDK ``java.util.Random``
stream, Random123 Philox known answers and the analytic KATs under ``tests/``.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package.  The product (``libmvsim.so`` and the ``multiview-simulation_amd``
package) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liborc.so")


def build(force: bool = False) -> str:
    """Compile the C restatement with gcc (``make -C oracle``)."""
    src_m = max(os.path.getmtime(os.path.join(_HERE, f)) for f in ("mvsim_oracle.c", "mvsim_oracle.h"))
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < src_m:
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None

_i64p = C.POINTER(C.c_int64)
_f32p = C.POINTER(C.c_float)
_f64p = C.POINTER(C.c_double)
_u32p = C.POINTER(C.c_uint32)


class JRandomState(C.Structure):
    _fields_ = [("s", C.c_uint64)]


def lib():
    global _lib
    if _lib is None:
        # On the GPU box there may be no need to rebuild; build() is a no-op when current.
        try:
            build()
        except Exception:
            if not os.path.exists(_LIB_PATH):
                raise
        L = C.CDLL(_LIB_PATH)
        L.orc_jrandom_seed.argtypes = [C.POINTER(JRandomState), C.c_int64]
        L.orc_jrandom_next.argtypes = [C.POINTER(JRandomState), C.c_int]
        L.orc_jrandom_next.restype = C.c_int32
        L.orc_jrandom_next_int.argtypes = [C.POINTER(JRandomState)]
        L.orc_jrandom_next_int.restype = C.c_int32
        L.orc_jrandom_next_int_bound.argtypes = [C.POINTER(JRandomState), C.c_int32]
        L.orc_jrandom_next_int_bound.restype = C.c_int32
        L.orc_jrandom_next_long.argtypes = [C.POINTER(JRandomState)]
        L.orc_jrandom_next_long.restype = C.c_int64
        L.orc_jrandom_next_double.argtypes = [C.POINTER(JRandomState)]
        L.orc_jrandom_next_double.restype = C.c_double
        L.orc_poisson_interarrival.argtypes = [C.POINTER(JRandomState), C.c_double]
        L.orc_poisson_interarrival.restype = C.c_int32
        L.orc_poisson_mul.argtypes = [C.c_double]
        L.orc_poisson_mul.restype = C.c_double
        L.orc_philox4x32_10.argtypes = [_u32p, _u32p, _u32p]
        L.orc_det_log.argtypes = [C.c_double]
        L.orc_det_log.restype = C.c_double
        L.orc_det_exp.argtypes = [C.c_double]
        L.orc_det_exp.restype = C.c_double
        L.orc_det_exp_neg.argtypes = [C.c_double]
        L.orc_det_exp_neg.restype = C.c_double
        L.orc_det_lgamma_int.argtypes = [C.c_int64]
        L.orc_det_lgamma_int.restype = C.c_double
        L.orc_poisson_counter.argtypes = [C.c_double, C.c_uint64, C.c_uint32, C.c_uint64]
        L.orc_poisson_counter.restype = C.c_int64
        L.orc_axis_rotation.argtypes = [_i64p, C.c_int, C.c_int, _f64p]
        L.orc_affine_invert.argtypes = [_f64p, _f64p]
        L.orc_rotate_around_axis.argtypes = [_f32p, _i64p, C.c_int, C.c_int, _f32p]
        L.orc_rotate_around_axis_planes.argtypes = [_f32p, _i64p, C.c_int, C.c_int, C.c_int64, C.c_int64, _f32p]
        L.orc_attenuate3d.argtypes = [_f32p, _i64p, C.c_double, _f32p]
        L.orc_sum_image.argtypes = [_f32p, C.c_int64]
        L.orc_sum_image.restype = C.c_double
        L.orc_norm_image.argtypes = [_f32p, C.c_int64]
        L.orc_adjust_image.argtypes = [_f32p, C.c_int64, C.c_float, C.c_float]
        L.orc_adjust_image.restype = C.c_double
        L.orc_convolve_direct.argtypes = [_f32p, _i64p, _f32p, _i64p, _f32p]
        L.orc_set_parallel.argtypes = [C.c_int]
        L.orc_max_threads.restype = C.c_int
        L.orc_convolve_direct_at.argtypes = [_f32p, _i64p, _f32p, _i64p, _i64p, C.c_int64, _f64p]
        L.orc_extract_nz.argtypes = [C.c_int64, C.c_int]
        L.orc_extract_nz.restype = C.c_int64
        L.orc_extract_slices_ref.argtypes = [_f32p, _i64p, C.c_int, C.c_float, C.POINTER(JRandomState), _f32p]
        L.orc_extract_slices_counter.argtypes = [_f32p, _i64p, C.c_int, C.c_float, C.c_uint64, C.c_uint32, _f32p]
        L.orc_extract_slices_counter_window.argtypes = [_f32p, _i64p, C.c_int, C.c_float, C.c_uint64, C.c_uint32, C.c_int64, _f32p]
        L.orc_isotropic_nz.argtypes = [C.c_int64, C.c_int]
        L.orc_isotropic_nz.restype = C.c_int64
        L.orc_make_isotropic.argtypes = [_f32p, _i64p, C.c_int, _f32p]
        L.orc_draw_spheres.argtypes = [_f32p, _i64p, C.c_double, C.c_double, C.c_int, C.c_int, C.c_void_p,
                                       C.POINTER(C.c_int64)]
        L.orc_draw_spheres.restype = C.c_int
        L.orc_downsample2x.argtypes = [_f32p, _i64p, _f32p]
        L.orc_downsample2x.restype = C.c_int
        L.orc_hypersphere_size.argtypes = [C.c_int64]
        L.orc_hypersphere_size.restype = C.c_int64
        L.orc_compute_weight_image.argtypes = [_i64p, _f32p]
        L.orc_normalize_weights.argtypes = [C.POINTER(_f32p), C.c_int, C.c_int64, C.c_float]
        _lib = L
    return _lib


# --------------------------------------------------------------------------- helpers
def _vol(a) -> np.ndarray:
    """Volumes are numpy arrays of shape (Nz, Ny, Nx), float32, C-contiguous (x fastest)."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    if a.ndim != 3:
        raise ValueError("expected a 3-D volume shaped (Nz, Ny, Nx)")
    return a


def _dim(a: np.ndarray):
    nz, ny, nx = a.shape
    return (C.c_int64 * 3)(nx, ny, nz)


def _p(a: np.ndarray):
    return a.ctypes.data_as(_f32p)


# --------------------------------------------------------------------------- java.util.Random
class JRandom:
    """java.util.Random restated from the JDK specification."""

    def __init__(self, seed: int):
        self.st = JRandomState()
        lib().orc_jrandom_seed(C.byref(self.st), seed)

    def next(self, bits: int) -> int:
        return lib().orc_jrandom_next(C.byref(self.st), bits)

    def nextInt(self, bound: int | None = None) -> int:
        if bound is None:
            return lib().orc_jrandom_next_int(C.byref(self.st))
        if bound <= 0:
            raise ValueError("bound must be positive")
        return lib().orc_jrandom_next_int_bound(C.byref(self.st), bound)

    def nextLong(self) -> int:
        return lib().orc_jrandom_next_long(C.byref(self.st))

    def nextDouble(self) -> float:
        return lib().orc_jrandom_next_double(C.byref(self.st))

    def poisson(self, mean: float) -> int:
        """uncommons/PoissonGenerator.java:95-109"""
        return lib().orc_poisson_interarrival(C.byref(self.st), mean)


def poisson_mul(snr: float) -> float:
    return lib().orc_poisson_mul(float(np.float32(snr)))


def philox4x32_10(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().orc_philox4x32_10(c, k, o)
    return tuple(int(v) for v in o)


def det_log(x: float) -> float:
    return lib().orc_det_log(x)


def det_exp(x: float) -> float:
    return lib().orc_det_exp(x)


def det_exp_neg(lam: float) -> float:
    return lib().orc_det_exp_neg(lam)


def det_lgamma_int(k: int) -> float:
    return lib().orc_det_lgamma_int(k)


def poisson_counter(lam: float, seed: int, stream: int, index: int) -> int:
    return lib().orc_poisson_counter(lam, seed & 0xFFFFFFFFFFFFFFFF, stream, index)


# --------------------------------------------------------------------------- stage ops
def axis_rotation(dim_xyz, axis: int, degrees: int) -> np.ndarray:
    """SMVD:80-102 -> 3x4 row-major forward model."""
    d = (C.c_int64 * 3)(*dim_xyz)
    m = (C.c_double * 12)()
    lib().orc_axis_rotation(d, axis, degrees, m)
    return np.array(m, dtype=np.float64).reshape(3, 4)


def affine_invert(m) -> np.ndarray:
    a = (C.c_double * 12)(*np.asarray(m, dtype=np.float64).ravel())
    o = (C.c_double * 12)()
    lib().orc_affine_invert(a, o)
    return np.array(o, dtype=np.float64).reshape(3, 4)


def rotate_around_axis(vol, axis: int, degrees: int) -> np.ndarray:
    v = _vol(vol)
    out = np.empty_like(v)
    rc = lib().orc_rotate_around_axis(_p(v), _dim(v), axis, degrees, _p(out))
    if rc:
        raise ValueError("rotate_around_axis: invalid axis")
    return out


def rotate_around_axis_planes(vol, axis: int, degrees: int, z0: int, nzp: int) -> np.ndarray:
    """Planes z0 .. z0 + nzp - 1 of rotate_around_axis(vol, axis, degrees): the reference's cursor loop (SMVD:119-132) over those
    output voxels only -- for parity at sizes where the whole volume would take the single-threaded restatement minutes."""
    v = _vol(vol)
    out = np.empty((nzp,) + v.shape[1:], np.float32)
    rc = lib().orc_rotate_around_axis_planes(_p(v), _dim(v), axis, degrees, int(z0), int(nzp), _p(out))
    if rc:
        raise ValueError("rotate_around_axis_planes: invalid axis or plane range")
    return out


def attenuate3d(vol, delta: float) -> np.ndarray:
    v = _vol(vol)
    out = np.empty_like(v)
    rc = lib().orc_attenuate3d(_p(v), _dim(v), float(delta), _p(out))
    if rc:
        raise ValueError("attenuate3d: Nx > Ny walks outside the interval in the reference (Q1)")
    return out


def sum_image(a) -> float:
    a = np.ascontiguousarray(a, dtype=np.float32)
    return lib().orc_sum_image(_p(a), a.size)


def norm_image(a: np.ndarray) -> None:
    """In place (Tools:112-118)."""
    assert a.dtype == np.float32 and a.flags.c_contiguous
    lib().orc_norm_image(_p(a), a.size)


def adjust_image(a: np.ndarray, min_value: float = 1e-4, target_average: float = 1.0) -> float:
    """In place; returns the correction (Tools:143-159)."""
    assert a.dtype == np.float32 and a.flags.c_contiguous
    return lib().orc_adjust_image(_p(a), a.size, min_value, target_average)


def convolve_direct(vol, psf: np.ndarray) -> np.ndarray:
    """Exact (double-accumulated) convolution; normalises ``psf`` in place (Q5)."""
    v = _vol(vol)
    assert psf.dtype == np.float32 and psf.flags.c_contiguous and psf.ndim == 3
    out = np.empty_like(v)
    rc = lib().orc_convolve_direct(_p(v), _dim(v), _p(psf), _dim(psf), _p(out))
    if rc:
        raise MemoryError("convolve_direct")
    return out


def set_parallel(on: bool) -> None:
    """bench.py's cpu_baseline mode "all_cores": OpenMP in rotate / attenuate / adjust_image (default off = the reference's
    single-threaded cursor loops; the parity tests never switch it on)."""
    lib().orc_set_parallel(1 if on else 0)


def max_threads() -> int:
    return int(lib().orc_max_threads())


def convolve_direct_at(vol, psf_normalised: np.ndarray, idx) -> np.ndarray:
    """The exact (double) convolution sum at the listed flat voxel indices only; ``psf_normalised`` is used as given."""
    v = _vol(vol)
    assert psf_normalised.dtype == np.float32 and psf_normalised.flags.c_contiguous and psf_normalised.ndim == 3
    ii = np.ascontiguousarray(idx, dtype=np.int64)
    assert ii.min() >= 0 and ii.max() < v.size
    out = np.empty(ii.size, dtype=np.float64)
    lib().orc_convolve_direct_at(_p(v), _dim(v), _p(psf_normalised), _dim(psf_normalised),
                                 ii.ctypes.data_as(_i64p), ii.size, out.ctypes.data_as(_f64p))
    return out


def jtk_nfft_fast(n: int) -> int:
    """Smallest product of mutually prime factors from {2,3,4,5,7,8,9,11,13,16} that is >= n
    (the Mines JTK ``FftComplex.nfftFast`` size table used by imglib2's fft2)."""
    sizes = set()
    for p2 in (1, 2, 4, 8, 16):
        for p3 in (1, 3, 9):
            for p5 in (1, 5):
                for p7 in (1, 7):
                    for p11 in (1, 11):
                        for p13 in (1, 13):
                            sizes.add(p2 * p3 * p5 * p7 * p11 * p13)
    cands = sorted(s for s in sizes if s >= n)
    if not cands:
        raise ValueError("size exceeds the JTK table (720720)")
    return cands[0]


def convolve_fft(vol, psf: np.ndarray, workers: int = -1, padded=None) -> np.ndarray:
    """FFTConvolution recipe in float32 (imglib2-algorithm 0.18.3):
    mirror-single extended image, zero-extended kernel re-centred so index K/2 sits on the
    origin, r2c, plain complex product (no conjugate, SMVD:260), c2r, crop.  Normalises
    ``psf`` in place first (SMVD:255)."""
    import scipy.fft as sfft

    v = _vol(vol)
    assert psf.dtype == np.float32 and psf.flags.c_contiguous and psf.ndim == 3
    norm_image(psf)
    n = np.array(v.shape)
    k = np.array(psf.shape)
    need = n + k - 1
    if padded is None:
        # x (last numpy axis) is the real-to-complex dimension: FftReal.nfftFast = 2*nfftFast(n/2)
        P = [jtk_nfft_fast(int(s)) for s in need]
        P[2] = 2 * jtk_nfft_fast((int(need[2]) + 1) // 2)
    else:
        P = list(padded)
    P = np.array(P)
    # centred padding of the image interval, filled by the mirror extension
    lo = (P - n) // 2
    hi = P - n - lo
    img = _mirror_pad(v, lo, hi)
    ker = np.zeros(tuple(int(p) for p in P), dtype=np.float32)
    ker[: k[0], : k[1], : k[2]] = psf
    ker = np.roll(ker, shift=tuple(int(-(kk // 2)) for kk in k), axis=(0, 1, 2))
    F = sfft.rfftn(img, workers=workers)
    G = sfft.rfftn(ker, workers=workers)
    F *= G
    full = sfft.irfftn(F, s=tuple(int(p) for p in P), workers=workers)
    out = full[lo[0]: lo[0] + n[0], lo[1]: lo[1] + n[1], lo[2]: lo[2] + n[2]]
    return np.ascontiguousarray(out, dtype=np.float32)


def _mirror_pad(v, lo, hi):
    idx = []
    for d in range(3):
        n = v.shape[d]
        i = np.arange(-int(lo[d]), n + int(hi[d]))
        if n == 1:
            i[:] = 0
        else:
            p = 2 * n - 2
            i = np.mod(i, p)
            i = np.where(i < n, i, p - i)
        idx.append(i)
    return v[np.ix_(*idx)]


def extract_nz(nz: int, inc: int) -> int:
    return lib().orc_extract_nz(nz, inc)


def extract_slices_ref(vol, inc: int, snr: float, rnd: JRandom | None = None) -> np.ndarray:
    """SMVD:195-231 with the reference's own sequential RNG consumption (Q10)."""
    v = _vol(vol)
    if inc < 1:
        raise ValueError("inc must be >= 1")
    if rnd is None:
        rnd = JRandom(464232194)  # SMVD:76
    nz, ny, nx = v.shape
    out = np.empty((extract_nz(nz, inc), ny, nx), dtype=np.float32)
    lib().orc_extract_slices_ref(_p(v), _dim(v), inc, snr, C.byref(rnd.st), _p(out))
    return out


def extract_slices_counter(vol, inc: int, snr: float, seed: int, stream: int = 0) -> np.ndarray:
    """Same indexing, counter-based RNG (the sampler the HIP path implements)."""
    v = _vol(vol)
    if inc < 1:
        raise ValueError("inc must be >= 1")
    nz, ny, nx = v.shape
    out = np.empty((extract_nz(nz, inc), ny, nx), dtype=np.float32)
    lib().orc_extract_slices_counter(_p(v), _dim(v), inc, snr, seed & 0xFFFFFFFFFFFFFFFF, stream, _p(out))
    return out


def extract_slices_counter_window(win, inc: int, snr: float, seed: int, stream: int, z0: int) -> np.ndarray:
    """extract_slices_counter on the planes z0 .. z0 + len(win) - 1 of a larger volume (z0 a multiple of inc): the counters are the
    source indices in the FULL volume, the result the planes z0 // inc ... of the whole view's acquisition."""
    v = _vol(win)
    nz, ny, nx = v.shape
    out = np.empty((extract_nz(nz, inc), ny, nx), dtype=np.float32)
    rc = lib().orc_extract_slices_counter_window(_p(v), _dim(v), inc, snr, seed & 0xFFFFFFFFFFFFFFFF, stream, int(z0), _p(out))
    if rc:
        raise ValueError("extract_slices_counter_window: inc < 1 or z0 is not a plane extractSlices reads")
    return out


def make_isotropic(vol, inc: int) -> np.ndarray:
    v = _vol(vol)
    nz, ny, nx = v.shape
    out = np.empty((lib().orc_isotropic_nz(nz, inc), ny, nx), dtype=np.float32)
    lib().orc_make_isotropic(_p(v), _dim(v), inc, _p(out))
    return out


def draw_spheres(img: np.ndarray, min_value: float, max_value: float, scale: int, half_pixel_offset: bool,
                 rnd: "JRandom") -> int:
    """SMVD:436-522, in place on a contiguous float32 (Nz,Ny,Nx) image; returns the number of small spheres."""
    assert img.dtype == np.float32 and img.flags.c_contiguous and img.ndim == 3
    n = C.c_int64(0)
    rc = lib().orc_draw_spheres(_p(img), _dim(img), float(min_value), float(max_value), int(scale),
                                int(bool(half_pixel_offset)), C.byref(rnd.st), C.byref(n))
    if rc != 0:
        raise ValueError(f"orc_draw_spheres failed ({rc})")
    return int(n.value)


def downsample2x(vol) -> np.ndarray:
    """SMVD:394-424."""
    v = _vol(vol)
    nz, ny, nx = v.shape
    out = np.empty((nz // 2 - 1, ny // 2 - 1, nx // 2 - 1), dtype=np.float32)
    if lib().orc_downsample2x(_p(v), _dim(v), _p(out)) != 0:
        raise ValueError("orc_downsample2x: image too small")
    return out


def simulate_phantom(size: int = 289, scale: int = 2, half_pixel_offset: bool = False, rnd: "JRandom" = None) -> np.ndarray:
    """SMVD:366-392 (`simulate`): spheres at `scale`x resolution, then 2x downsampling."""
    if scale == 2:
        size += 1
    img = np.zeros((size * scale,) * 3, dtype=np.float32)
    draw_spheres(img, 0.0, 1.0, scale, half_pixel_offset, rnd if rnd is not None else JRandom(464232194))
    return downsample2x(img) if scale == 2 else img


def hypersphere_size(radius: int) -> int:
    return int(lib().orc_hypersphere_size(int(radius)))


def compute_weight_image(shape_zyx) -> np.ndarray:
    nz, ny, nx = shape_zyx
    out = np.empty((nz, ny, nx), dtype=np.float32)
    lib().orc_compute_weight_image((C.c_int64 * 3)(nx, ny, nz), _p(out))
    return out


def normalize_weights(weights: list, osem: float) -> None:
    """SMVD:615-640, in place on a list of float32 arrays."""
    arr = (_f32p * len(weights))(*[_p(w) for w in weights])
    rc = lib().orc_normalize_weights(arr, len(weights), weights[0].size, osem)
    if rc:
        raise ValueError("normalize_weights")


REF_ATTENUATION = float(np.float32(0.01))     # `final float attenuation = 0.01f`, widened to double at the call (SMVD:533,573)


def simulate_view(gt, psf, angle_deg: int, *, axis: int = 0, delta: float = REF_ATTENUATION, min_value: float = 1e-4,
                  avg: float = 1.0, inc: int = 1, snr: float = 25.0, seed: int = 464232194, stream: int = 0,
                  conv: str = "direct"):
    """One iteration of the loop body SMVD:567-585 (rotate -> attenuate -> convolve -> adjust ->
    extractSlices) with the counter-based Poisson stream.  Returns dict of all stages."""
    rot = rotate_around_axis(gt, axis, angle_deg)
    att = attenuate3d(rot, delta)
    con = convolve_direct(att, psf) if conv == "direct" else convolve_fft(att, psf)
    corr = adjust_image(con, min_value, avg)
    acq = extract_slices_counter(con, inc, snr, seed, stream)
    return {"rot": rot, "att": att, "con": con, "corr": corr, "acq": acq}
