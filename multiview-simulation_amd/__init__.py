"""multiview-simulation_amd -- MI355X (gfx950) drop-in for the per-view acquisition path of
``net.preibisch.simulation.SimulateMultiViewDataset``:

    rotate -> attenuate -> 3-D PSF convolve -> adjust -> axial slice extraction -> Poisson

The arithmetic lives in hand-written HIP kernels inside ``libmvsim.so`` (C ABI: ``include/mvsim.h``).
This package is the host-side mirror of the reference's static-method interface, written in Python
because the image has no JVM; the Java facade + JNI shim with identical signatures is under
``java/``.  Images are numpy float32 arrays shaped ``(Nz, Ny, Nx)`` (x fastest),
standing in for ImgLib2 ``RandomAccessibleInterval<FloatType>`` / ``Img<FloatType>``.

The directory name carries a hyphen, so import it with::

    import importlib; mvs = importlib.import_module("multiview-simulation_amd")

There is no CPU fallback anywhere in this package.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import MvsimError, MvsimNoDeviceError, Sphere, Timings, ViewOutputs, ViewParams  # noqa: F401

__all__ = ["Context", "Group", "JavaRandom", "SimulateMultiViewDataset", "Tools", "broadcast_plan", "default_context", "MvsimError",
           "MvsimNoDeviceError", "ViewParams", "shard_views", "version"]


def broadcast_plan(nranks: int, rank: int, root: int, count: int, pieces: int = 8):
    """The schedule of the pipelined ground-truth broadcast (mvsim_comm_broadcast_plan; host only): a list of
    (stage, 'send' | 'recv', peer, first float, floats) in the order the rank issues them."""
    L = _lib.load()
    n = C.c_int(0)
    _lib.check(L.mvsim_comm_broadcast_plan(nranks, rank, root, count, pieces, None, 0, C.byref(n)))
    ops = (_lib.BcastOp * max(1, n.value))()
    _lib.check(L.mvsim_comm_broadcast_plan(nranks, rank, root, count, pieces, ops, n.value, C.byref(n)))
    return [(o.stage, "send" if o.kind == 0 else "recv", o.peer, o.first, o.count) for o in ops[: n.value]]


def version() -> str:
    return _lib.load().mvsim_version().decode()


# ------------------------------------------------------------------------------------------------
# java.util.Random (JDK specification) -- host logic: the reference's entry points take a Random
# (SimulateMultiViewDataset.java:195, Tools.java:73); the facade draws nextLong() from it as the
# counter-RNG seed so that `new Random(seed0)` on the caller side stays reproducible.
# ------------------------------------------------------------------------------------------------
class JavaRandom:
    _MULT = 0x5DEECE66D
    _MASK = (1 << 48) - 1

    def __init__(self, seed: int):
        self.setSeed(seed)

    def setSeed(self, seed: int) -> None:
        self._s = (seed ^ self._MULT) & self._MASK

    def next(self, bits: int) -> int:
        self._s = (self._s * self._MULT + 0xB) & self._MASK
        v = self._s >> (48 - bits)
        v &= 0xFFFFFFFF
        return v - (1 << 32) if v & 0x80000000 else v

    def nextInt(self, bound: int | None = None) -> int:
        if bound is None:
            return self.next(32)
        if bound <= 0:
            raise ValueError("bound must be positive")
        r = self.next(31)
        m = bound - 1
        if bound & m == 0:
            return (bound * r) >> 31
        u = r
        while True:
            r = u % bound
            t = (u - r + m) & 0xFFFFFFFF
            if not (t & 0x80000000):
                return r
            u = self.next(31)

    def nextLong(self) -> int:
        v = ((self.next(32) << 32) + self.next(32)) & 0xFFFFFFFFFFFFFFFF
        return v - (1 << 64) if v & (1 << 63) else v

    def nextDouble(self) -> float:
        return ((self.next(26) << 27) + self.next(27)) * (1.0 / (1 << 53))


def _seed_from(rnd) -> int:
    if rnd is None:
        rnd = SimulateMultiViewDataset.rnd
    if isinstance(rnd, int):
        return rnd & 0xFFFFFFFFFFFFFFFF
    return rnd.nextLong() & 0xFFFFFFFFFFFFFFFF


# ------------------------------------------------------------------------------------------------
def _as_volume(a, name="image") -> np.ndarray:
    """Any array-like / strided view -> contiguous float32 (Nz,Ny,Nx), as the Java facade copies an
    arbitrary RandomAccessibleInterval view into a direct buffer."""
    v = np.ascontiguousarray(a, dtype=np.float32)
    if v.ndim != 3:
        raise ValueError(f"{name}: expected 3 dimensions, got {v.ndim}")
    if v.size == 0:
        raise ValueError(f"{name}: empty interval")
    return v


def _dim(v: np.ndarray):
    nz, ny, nx = v.shape
    return (C.c_int64 * 3)(nx, ny, nz)


def _ptr(a: np.ndarray) -> C.c_void_p:
    return C.c_void_p(a.ctypes.data)


class Context:
    """One ``mvsim_ctx``: bound to one GPU, not thread-safe."""

    def __init__(self, device: int | None = None):
        self._h = C.c_void_p()
        self._L = _lib.load()
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0"))
        _lib.check(self._L.mvsim_create(device, C.byref(self._h)))
        self.device = device

    # -- lifecycle
    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h:
            self._L.mvsim_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_stream(self, hip_stream: int | None) -> None:
        _lib.check(self._L.mvsim_set_stream(self._h, C.c_void_p(hip_stream or 0)))

    def set_option(self, name: str, value) -> None:
        """Run-time switch of this context (mvsim_set_option): fft_zpass, fft_backend, fft_pad, fused_rotate,
        poisson_queue, poisson_queue_share, early_sum, graph, fuse_tail, psf_overlap, tail_overlap, attenuate, broadcast, view_batch,
        view_lanes, acq_transfer, host_threads (include/mvsim.h and DESIGN.md list them with their values)."""
        if isinstance(value, bool):
            value = "1" if value else "0"
        _lib.check(self._L.mvsim_set_option(self._h, name.encode(), str(value).encode()))

    def synchronize(self) -> None:
        _lib.check(self._L.mvsim_synchronize(self._h))

    def join(self) -> None:
        """mvsim_join: order the context's internal streams (the tail of the last device view) in front of whatever its
        stream gets next; only a caller with a stream of its own and tail_overlap=any needs it."""
        _lib.check(self._L.mvsim_join(self._h))

    def release_caches(self) -> None:
        _lib.check(self._L.mvsim_release_caches(self._h))

    # -- device memory
    def dev_alloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        _lib.check(self._L.mvsim_dev_alloc(self._h, nbytes, C.byref(p)))
        return p.value or 0

    def dev_free(self, dptr: int) -> None:
        _lib.check(self._L.mvsim_dev_free(self._h, C.c_void_p(dptr)))

    def pinned_empty(self, shape, dtype=np.float32) -> np.ndarray:
        """A numpy array in page-locked host memory (PCIe-speed transfers for the host-buffer entry points).  The
        block is freed when the array -- and every view of it -- is garbage collected."""
        dt = np.dtype(dtype)
        n = int(np.prod(shape))
        p = C.c_void_p()
        _lib.check(self._L.mvsim_host_alloc(self._h, max(1, n * dt.itemsize), C.byref(p)))
        L, addr = self._L, p.value

        class _Block(C.Array):
            _type_ = C.c_char
            _length_ = max(1, n * dt.itemsize)

            def __del__(self):
                try:
                    L.mvsim_host_free(None, C.c_void_p(addr))      # no context needed: arrays may outlive it
                except Exception:
                    pass

        blk = _Block.from_address(addr)
        return np.frombuffer(blk, dtype=dt, count=n).reshape(shape)

    def upload(self, dptr: int, host: np.ndarray) -> None:
        host = np.ascontiguousarray(host)
        _lib.check(self._L.mvsim_upload(self._h, C.c_void_p(dptr), _ptr(host), host.nbytes))

    def download(self, dptr: int, shape, dtype=np.float32) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        _lib.check(self._L.mvsim_download(self._h, _ptr(out), C.c_void_p(dptr), out.nbytes))
        return out

    # -- host-buffer stage operators
    def rotate_around_axis(self, img, axis: int, degrees: int) -> np.ndarray:
        v = _as_volume(img)
        out = np.empty_like(v)
        _lib.check(self._L.mvsim_rotate_around_axis(self._h, _ptr(v), _dim(v), axis, degrees, _ptr(out)))
        return out

    def attenuate3d(self, img, delta: float) -> np.ndarray:
        v = _as_volume(img)
        out = np.empty_like(v)
        _lib.check(self._L.mvsim_attenuate3d(self._h, _ptr(v), _dim(v), float(delta), _ptr(out)))
        return out

    def norm_image(self, img: np.ndarray) -> None:
        _check_inplace(img)
        _lib.check(self._L.mvsim_norm_image(self._h, _ptr(img), img.size))

    def convolve(self, img, psf: np.ndarray, method: int = 0) -> np.ndarray:
        v = _as_volume(img)
        _check_inplace(psf, "psf")
        if psf.ndim != 3:
            raise ValueError("psf: expected 3 dimensions")
        out = np.empty_like(v)
        _lib.check(self._L.mvsim_convolve(self._h, _ptr(v), _dim(v), _ptr(psf), _dim(psf), method, _ptr(out)))
        return out

    def adjust_image(self, img: np.ndarray, min_value: float, target_average: float) -> float:
        _check_inplace(img)
        corr = C.c_double()
        _lib.check(self._L.mvsim_adjust_image(self._h, _ptr(img), img.size, min_value, target_average, C.byref(corr)))
        return corr.value

    def extract_slices(self, img, inc: int, snr: float, seed: int, stream: int = 0) -> np.ndarray:
        v = _as_volume(img)
        nz, ny, nx = v.shape
        if inc < 1:
            raise ValueError("inc must be >= 1")
        out = np.empty((self._L.mvsim_extract_nz(nz, inc), ny, nx), dtype=np.float32)
        _lib.check(self._L.mvsim_extract_slices(self._h, _ptr(v), _dim(v), inc, snr, seed & 0xFFFFFFFFFFFFFFFF,
                                                stream, _ptr(out)))
        return out

    def poisson_process(self, img: np.ndarray, snr: float, seed: int, stream: int = 0, index_offset: int = 0) -> None:
        _check_inplace(img)
        _lib.check(self._L.mvsim_poisson_process(self._h, _ptr(img), img.size, float(snr),
                                                 seed & 0xFFFFFFFFFFFFFFFF, stream, index_offset))

    def make_isotropic(self, img, inc: int) -> np.ndarray:
        v = _as_volume(img)
        nz, ny, nx = v.shape
        if inc < 1:
            raise ValueError("inc must be >= 1")
        out = np.empty((self._L.mvsim_isotropic_nz(nz, inc), ny, nx), dtype=np.float32)
        _lib.check(self._L.mvsim_make_isotropic(self._h, _ptr(v), _dim(v), inc, _ptr(out)))
        return out

    # -- ground-truth phantom (SimulateMultiViewDataset.java:366-522)
    def draw_spheres(self, img: np.ndarray, min_value: float, max_value: float, scale: int, half_pixel_offset: bool,
                     rnd: "JavaRandom") -> int:
        """drawSpheres, in place on a contiguous float32 (Nz,Ny,Nx) image; ``rnd`` is advanced exactly as the
        reference advances its java.util.Random.  Returns the number of small spheres drawn."""
        _check_inplace(img)
        st = C.c_uint64(rnd._s)
        n = C.c_int64(0)
        _lib.check(self._L.mvsim_draw_spheres(self._h, _ptr(img), _dim(img), float(min_value), float(max_value),
                                              int(scale), int(bool(half_pixel_offset)), C.byref(st), C.byref(n)))
        rnd._s = int(st.value)
        return int(n.value)

    def splat_spheres(self, img: np.ndarray, spheres) -> None:
        """The GPU half of drawSpheres: max-composite (cx, cy, cz, radius, value) items into ``img`` in place."""
        _check_inplace(img)
        arr = (Sphere * len(spheres))(*[Sphere(int(cx), int(cy), int(cz), int(r), float(v)) for cx, cy, cz, r, v in spheres])
        _lib.check(self._L.mvsim_splat_spheres(self._h, _ptr(img), _dim(img), arr, len(spheres)))

    def downsample2x(self, img) -> np.ndarray:
        v = _as_volume(img)
        nz, ny, nx = v.shape
        if min(nz, ny, nx) < 4:
            raise ValueError("downSample2x needs at least 4 samples per dimension")
        out = np.empty((nz // 2 - 1, ny // 2 - 1, nx // 2 - 1), dtype=np.float32)
        _lib.check(self._L.mvsim_downsample2x(self._h, _ptr(v), _dim(v), _ptr(out)))
        return out

    def simulate_phantom(self, size: int = 289, scale: int = 2, half_pixel_offset: bool = False,
                         rnd: "JavaRandom | None" = None) -> np.ndarray:
        """`simulate` (:366-392) with the canvas resident in HBM: zero canvas of (size[+1])*scale voxels per
        dimension, drawSpheres, 2x down-sampling when scale == 2; only the result crosses PCIe."""
        if rnd is None:
            rnd = SimulateMultiViewDataset.rnd
        if scale == 2:
            size += 1
        n = size * scale
        dim = (C.c_int64 * 3)(n, n, n)
        nbytes = n ** 3 * 4
        canvas = self.dev_alloc(nbytes)
        out_d = 0
        try:
            _lib.check(self._L.mvsim_dev_memset(self._h, canvas, 0, nbytes))
            st = C.c_uint64(rnd._s)
            _lib.check(self._L.mvsim_draw_spheres_dev(self._h, canvas, dim, 0.0, 1.0, int(scale),
                                                      int(bool(half_pixel_offset)), C.byref(st), None))
            rnd._s = int(st.value)
            if scale != 2:
                return self.download(canvas, (n, n, n))
            o = n // 2 - 1
            out_d = self.dev_alloc(o ** 3 * 4)
            _lib.check(self._L.mvsim_downsample2x_dev(self._h, canvas, dim, out_d))
            return self.download(out_d, (o, o, o))
        finally:
            self.dev_free(canvas)
            if out_d:
                self.dev_free(out_d)

    def compute_weight_image(self, shape_zyx) -> np.ndarray:
        nz, ny, nx = (int(s) for s in shape_zyx)
        out = np.empty((nz, ny, nx), dtype=np.float32)
        _lib.check(self._L.mvsim_compute_weight_image(self._h, (C.c_int64 * 3)(nx, ny, nz), _ptr(out)))
        return out

    # -- cross-view weight normalisation (SimulateMultiViewDataset.java:615-640)
    def normalize_weights(self, weights: list, osem: float) -> None:
        """In place on a list of equally sized float32 arrays (one per view)."""
        for w in weights:
            _check_inplace(w, "weights")
        n = weights[0].size
        if any(w.size != n for w in weights):
            raise ValueError("weights: all views must have the same size")
        arr = (C.c_void_p * len(weights))(*[w.ctypes.data for w in weights])
        _lib.check(self._L.mvsim_normalize_weights(self._h, arr, len(weights), n, osem))

    def normalize_weights_dev(self, dptrs: list, n: int, osem: float, sum_dptr: int = 0) -> None:
        arr = (C.c_void_p * len(dptrs))(*dptrs)
        _lib.check(self._L.mvsim_normalize_weights_dev(self._h, arr, len(dptrs), n, C.c_void_p(sum_dptr or None), osem))

    def sum_views_dev(self, dptrs: list, n: int, out_dptr: int) -> None:
        arr = (C.c_void_p * len(dptrs))(*dptrs)
        _lib.check(self._L.mvsim_sum_views_dev(self._h, arr, len(dptrs), n, C.c_void_p(out_dptr)))

    # -- fused per-view pipeline
    def view_params(self, **kw) -> ViewParams:
        p = ViewParams()
        self._L.mvsim_view_params_default(C.byref(p))
        for k, v in kw.items():
            if not hasattr(p, k):
                raise TypeError(f"unknown view parameter {k!r}")
            setattr(p, k, v)
        return p

    def simulate_view(self, gt, psf: np.ndarray, params: ViewParams, want=("acq",), out: dict | None = None) -> dict:
        """Host buffers in/out.  ``want`` is a subset of {'rot','att','con','acq'}; returns a dict with the
        requested stages plus 'corr'.  ``psf`` is normalised in place (reference behaviour).  ``out`` may carry
        preallocated destination arrays by stage name -- e.g. page-locked ones from pinned_empty, which (with a
        pinned ``gt``) make the transfers run at PCIe speed; allocate them once, page-locking is slow."""
        v = _as_volume(gt, "ground truth")
        _check_inplace(psf, "psf")
        nz, ny, nx = v.shape
        res = {}
        o = ViewOutputs()

        def alloc(name, shape):
            a = (out or {}).get(name)
            if a is None:
                return np.empty(shape, dtype=np.float32)
            if a.shape != tuple(shape) or a.dtype != np.float32 or not a.flags.c_contiguous:
                raise ValueError(f"out[{name!r}] must be a contiguous float32 array of shape {tuple(shape)}")
            return a
        for name in ("rot", "att", "con"):
            if name in want:
                res[name] = alloc(name, v.shape)
                setattr(o, name, res[name].ctypes.data)
        res["acq"] = alloc("acq", (self._L.mvsim_extract_nz(nz, params.inc), ny, nx))
        o.acq = res["acq"].ctypes.data
        corr = C.c_double()
        _lib.check(self._L.mvsim_simulate_view(self._h, _ptr(v), _dim(v), _ptr(psf), _dim(psf), C.byref(params),
                                               C.byref(o), C.byref(corr)))
        res["corr"] = corr.value
        return res

    def simulate_view_async(self, gt: np.ndarray, psf: np.ndarray, params: ViewParams, out: dict, gt_generation: int = 0) -> int:
        """Pipelined host-buffer view (mvsim_simulate_view_async): returns a ticket at once; ``out`` maps stage names
        ('acq' required; 'rot', 'att', 'con' optional) to preallocated contiguous float32 arrays -- page-locked ones
        (pinned_empty) make the transfers overlap the kernels.  ``gt``, ``psf`` and the outputs must stay alive and
        untouched until ``wait(ticket)``."""
        if not (isinstance(gt, np.ndarray) and gt.dtype == np.float32 and gt.flags.c_contiguous and gt.ndim == 3):
            raise ValueError("ground truth: contiguous float32 (Nz,Ny,Nx) array required (no hidden copy on the async path)")
        _check_inplace(psf, "psf")
        nz, ny, nx = gt.shape
        o = ViewOutputs()
        shapes = {"rot": gt.shape, "att": gt.shape, "con": gt.shape, "acq": (self._L.mvsim_extract_nz(nz, params.inc), ny, nx)}
        if "acq" not in out:
            raise ValueError("out['acq'] is required")
        for name, a in out.items():
            if name not in shapes:
                raise ValueError(f"unknown output {name!r}")
            if a.shape != tuple(shapes[name]) or a.dtype != np.float32 or not a.flags.c_contiguous:
                raise ValueError(f"out[{name!r}] must be a contiguous float32 array of shape {tuple(shapes[name])}")
            setattr(o, name, a.ctypes.data)
        t = C.c_int64()
        _lib.check(self._L.mvsim_simulate_view_async(self._h, _ptr(gt), gt_generation, _dim(gt), _ptr(psf), _dim(psf),
                                                     C.byref(params), C.byref(o), C.byref(t)))
        return int(t.value)

    def wait(self, ticket: int) -> float:
        """Blocks until the view behind ``ticket`` has landed in its host buffers; returns the adjustImage factor."""
        corr = C.c_double()
        _lib.check(self._L.mvsim_wait(self._h, ticket, C.byref(corr)))
        return corr.value

    def simulate_view_zslabs(self, gt_slabs: list, psf: np.ndarray, params: ViewParams, acq_slab_nz: list, pinned: bool = False) -> tuple:
        """Host buffers as z slabs (volumes beyond 2^31-1 voxels do not fit one Java array): ``gt_slabs`` is a list of
        contiguous float32 (nz_i, Ny, Nx) arrays, ``acq_slab_nz`` the plane counts of the acquisition slabs to return
        (``pinned``: in page-locked blocks of mvsim_host_alloc, as the Java facade hands them over)."""
        _check_inplace(psf, "psf")
        gs = [np.ascontiguousarray(g, dtype=np.float32) for g in gt_slabs]
        ny, nx = gs[0].shape[1:]
        nz = sum(g.shape[0] for g in gs)
        acq = [(self.pinned_empty((int(k), ny, nx)) if pinned else np.empty((int(k), ny, nx), dtype=np.float32)) for k in acq_slab_nz]
        ga = (C.c_void_p * len(gs))(*[g.ctypes.data for g in gs])
        gn = (C.c_int64 * len(gs))(*[g.shape[0] for g in gs])
        aa = (C.c_void_p * len(acq))(*[a.ctypes.data for a in acq])
        an = (C.c_int64 * len(acq))(*[a.shape[0] for a in acq])
        corr = C.c_double()
        _lib.check(self._L.mvsim_simulate_view_zslabs(self._h, ga, gn, len(gs), (C.c_int64 * 3)(nx, ny, nz), _ptr(psf), _dim(psf),
                                                      C.byref(params), aa, an, len(acq), C.byref(corr)))
        return acq, corr.value

    def stage_zslabs(self, op: str, in_slabs: list, out_slab_nz: list, **kw) -> list:
        """The per-stage operators on host buffers given as z slabs (``mvsim_*_zslabs``): op in {"rotate", "attenuate",
        "convolve", "extract"}; keyword arguments as the single-buffer methods take them.  Returns the output slabs."""
        gs = [np.ascontiguousarray(g, dtype=np.float32) for g in in_slabs]
        ny, nx = gs[0].shape[1:]
        nz = sum(g.shape[0] for g in gs)
        outs = [np.empty((int(k), ny, nx), dtype=np.float32) for k in out_slab_nz]
        ga = (C.c_void_p * len(gs))(*[g.ctypes.data for g in gs])
        gn = (C.c_int64 * len(gs))(*[g.shape[0] for g in gs])
        oa = (C.c_void_p * len(outs))(*[a.ctypes.data for a in outs])
        on = (C.c_int64 * len(outs))(*[a.shape[0] for a in outs])
        dim = (C.c_int64 * 3)(nx, ny, nz)
        L = self._L
        if op == "rotate":
            rc = L.mvsim_rotate_around_axis_zslabs(self._h, ga, gn, len(gs), dim, kw["axis"], kw["degrees"], oa, on, len(outs))
        elif op == "attenuate":
            rc = L.mvsim_attenuate3d_zslabs(self._h, ga, gn, len(gs), dim, kw["delta"], oa, on, len(outs))
        elif op == "convolve":
            psf = kw["psf"]
            _check_inplace(psf, "psf")
            rc = L.mvsim_convolve_zslabs(self._h, ga, gn, len(gs), dim, _ptr(psf), _dim(psf), kw.get("method", 0), oa, on, len(outs))
        elif op == "extract":
            rc = L.mvsim_extract_slices_zslabs(self._h, ga, gn, len(gs), dim, kw["inc"], kw["snr"], kw["seed"], kw.get("stream", 0), oa, on, len(outs))
        else:
            raise ValueError(f"unknown stage {op!r}")
        _lib.check(rc)
        return outs

    def simulate_view_dev(self, gt_dptr: int, dim_xyz, psf: np.ndarray, params: ViewParams, acq_dptr: int,
                          rot_dptr: int = 0, att_dptr: int = 0, con_dptr: int = 0, want_corr: bool = False):
        """Device-resident buffers (raw HBM addresses); asynchronous unless ``want_corr``."""
        _check_inplace(psf, "psf")
        o = ViewOutputs(rot_dptr or None, att_dptr or None, con_dptr or None, acq_dptr or None)
        corr = C.c_double()
        _lib.check(self._L.mvsim_simulate_view_dev(
            self._h, C.c_void_p(gt_dptr), (C.c_int64 * 3)(*dim_xyz), _ptr(psf), _dim(psf), C.byref(params),
            C.byref(o), C.byref(corr) if want_corr else None))
        return corr.value if want_corr else None

    def plane_stats(self):
        """(planes of the last flagged view, its empty attenuated planes, its empty planes behind the z pass): what option skip_empty skipped."""
        st = (C.c_int64 * 3)()
        _lib.check(self._L.mvsim_get_plane_stats(self._h, st))
        return int(st[0]), int(st[1]), int(st[2])

    def queue_stats(self):
        """The Poisson work queue of the last sampled view: dict(bytes, segment_items, bright, inversion, refused, fullest_block) --
        mvsim_get_queue_stats."""
        st = (C.c_int64 * 6)()
        _lib.check(self._L.mvsim_get_queue_stats(self._h, st))
        return dict(bytes=int(st[0]), segment_items=int(st[1]), bright=int(st[2]), inversion=int(st[3]), refused=int(st[4]), fullest_block=int(st[5]))

    def transfer_stats(self):
        """(views whose acquisition crossed PCIe as uint16 counts, how many of them fell back to float32)."""
        a, b = C.c_int64(), C.c_int64()
        _lib.check(self._L.mvsim_get_transfer_stats(self._h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def simulate_views_dev(self, gt_dptr: int, dim_xyz, psfs, params, acq_dptrs, con_dptrs=None) -> None:
        """``len(psfs)`` independent views of one device-resident ground truth in one call (the view loop of `main`,
        SimulateMultiViewDataset.java:567-585): the library runs as many side by side as pays for their size.  ``psfs[v]`` is
        normalised in place; ``params[v]``, ``acq_dptrs[v]`` (and ``con_dptrs[v]``) belong to view v.  Asynchronous."""
        nv = len(psfs)
        if not (len(params) == len(acq_dptrs) == nv) or (con_dptrs is not None and len(con_dptrs) != nv):
            raise ValueError("psfs, params and output lists must have the same length")
        if nv == 0:
            return
        for p in psfs:
            _check_inplace(p, "psf")
            if p.shape != psfs[0].shape:
                raise ValueError("all PSFs of one call share their dimensions")
        pp = (C.c_void_p * nv)(*[p.ctypes.data for p in psfs])
        pa = (ViewParams * nv)(*params)
        oo = (ViewOutputs * nv)(*[ViewOutputs(None, None, (con_dptrs[v] or None) if con_dptrs is not None else None, acq_dptrs[v] or None)
                                  for v in range(nv)])
        _lib.check(self._L.mvsim_simulate_views_dev(self._h, C.c_void_p(gt_dptr), (C.c_int64 * 3)(*dim_xyz), pp, _dim(psfs[0]), pa, oo, nv))

    def simulate_views(self, gt: np.ndarray, psfs, params) -> list:
        """Host buffers: ``len(psfs)`` views of one ground truth in ONE call (mvsim_simulate_views) -- the view loop of `main`
        (SimulateMultiViewDataset.java:567-585).  ``psfs[v]`` is normalised in place.  Returns the acquisitions."""
        g = _as_volume(gt, "ground truth")
        nv = len(psfs)
        if len(params) != nv:
            raise ValueError("psfs and params must have the same length")
        if nv == 0:
            return []
        for p in psfs:
            _check_inplace(p, "psf")
            if p.shape != psfs[0].shape:
                raise ValueError("all PSFs of one call share their dimensions")
        nz, ny, nx = g.shape
        outs = [np.empty((self._L.mvsim_extract_nz(nz, pr.inc), ny, nx), dtype=np.float32) for pr in params]
        pp = (C.c_void_p * nv)(*[p.ctypes.data for p in psfs])
        oo = (C.c_void_p * nv)(*[o.ctypes.data for o in outs])
        pa = (ViewParams * nv)(*params)
        _lib.check(self._L.mvsim_simulate_views(self._h, _ptr(g), _dim(g), pp, _dim(psfs[0]), pa, oo, nv))
        return outs

    def simulate_iteration_dev(self, gt_dptr: int, dim_xyz, psf: np.ndarray, params: ViewParams, back_degrees: int, acq_dptr: int,
                               iso_dptr: int = 0, view_dptr: int = 0, view_weights_dptr: int = 0, view_psf_dptr: int = 0,
                               rot_dptr: int = 0, att_dptr: int = 0, con_dptr: int = 0) -> None:
        """One iteration of `main`'s view loop (SimulateMultiViewDataset.java:567-613), device-resident: the view, then
        makeIsotropic and the three rotate-backs (iso, the weight image, the PSF) into whichever buffers are given."""
        _check_inplace(psf, "psf")
        o = ViewOutputs(rot_dptr or None, att_dptr or None, con_dptr or None, acq_dptr or None)
        more = _lib.IterationOutputs(iso_dptr or None, view_dptr or None, view_weights_dptr or None, view_psf_dptr or None)
        _lib.check(self._L.mvsim_simulate_iteration_dev(
            self._h, C.c_void_p(gt_dptr), (C.c_int64 * 3)(*dim_xyz), _ptr(psf), _dim(psf), C.byref(params), int(back_degrees),
            C.byref(o), C.byref(more)))

    # -- z-slab tiling of one view across GPUs (BASELINE configs[3]/[4])
    def slab_range(self, nz: int, nranks: int, rank: int):
        z0, z1 = C.c_int64(), C.c_int64()
        _lib.check(self._L.mvsim_slab_range(int(nz), int(nranks), int(rank), C.byref(z0), C.byref(z1)))
        return int(z0.value), int(z1.value)

    def view_slab_convolve_dev(self, gt_dptr: int, dim_xyz, psf: np.ndarray, params: ViewParams, z0: int, z1: int) -> float:
        """Rotate, attenuate and convolve the planes [z0, z1) of the view (kept in this context); returns their sum."""
        _check_inplace(psf, "psf")
        s = C.c_double()
        _lib.check(self._L.mvsim_view_slab_convolve_dev(self._h, C.c_void_p(gt_dptr), (C.c_int64 * 3)(*dim_xyz),
                                                        _ptr(psf), _dim(psf), C.byref(params), int(z0), int(z1),
                                                        C.byref(s)))
        return float(s.value)

    def view_slab_finish_dev(self, dim_xyz, params: ViewParams, z0: int, z1: int, total_sum: float, acq_dptr: int) -> int:
        """Adjust with the global mean, extract the acquired planes of the slab, Poisson; returns their number."""
        n = C.c_int64()
        _lib.check(self._L.mvsim_view_slab_finish_dev(self._h, (C.c_int64 * 3)(*dim_xyz), C.byref(params), int(z0),
                                                      int(z1), float(total_sum), C.c_void_p(acq_dptr), C.byref(n)))
        return int(n.value)

    def view_slab_dev(self, gt_dptr: int, dim_xyz, psf: np.ndarray, params: ViewParams, z0: int, z1: int, acq_dptr: int,
                      comm_ctx: "Context | None" = None) -> int:
        """One tiled view's slab in one asynchronous call, nothing through the host (mvsim_view_slab_dev): convolve, the slab sums
        reduced in place on this context's stream by ``comm_ctx``'s communicator (default: this context's own, if it has one), adjust,
        extract, Poisson.  Returns the number of acquired planes written to ``acq_dptr``."""
        _check_inplace(psf, "psf")
        n = C.c_int64()
        _lib.check(self._L.mvsim_view_slab_dev(self._h, comm_ctx._h if comm_ctx is not None else None, C.c_void_p(gt_dptr),
                                               (C.c_int64 * 3)(*dim_xyz), _ptr(psf), _dim(psf), C.byref(params), int(z0), int(z1),
                                               C.c_void_p(acq_dptr), C.byref(n)))
        return int(n.value)

    def comm_allreduce_sum_f64_dev(self, value_dptr: int, stream: int = 0) -> None:
        """In-place sum over the ranks of one DEVICE double, asynchronous on ``stream`` (0: the context's)."""
        _lib.check(self._L.mvsim_comm_allreduce_sum_f64_dev(self._h, C.c_void_p(value_dptr), C.c_void_p(stream) if stream else None))

    def comm_allreduce_sum_f64(self, value: float) -> float:
        v = C.c_double(value)
        _lib.check(self._L.mvsim_comm_allreduce_sum_f64(self._h, C.byref(v)))
        return float(v.value)

    # -- device-resident stage operators (raw addresses)
    def rotate_around_axis_dev(self, in_dptr, dim_xyz, axis, degrees, out_dptr):
        _lib.check(self._L.mvsim_rotate_around_axis_dev(self._h, C.c_void_p(in_dptr), (C.c_int64 * 3)(*dim_xyz), axis,
                                                        degrees, C.c_void_p(out_dptr)))

    def attenuate3d_dev(self, in_dptr, dim_xyz, delta, out_dptr):
        _lib.check(self._L.mvsim_attenuate3d_dev(self._h, C.c_void_p(in_dptr), (C.c_int64 * 3)(*dim_xyz), float(delta),
                                                 C.c_void_p(out_dptr)))

    def convolve_dev(self, in_dptr, dim_xyz, psf: np.ndarray, out_dptr, method: int = 0):
        _check_inplace(psf, "psf")
        _lib.check(self._L.mvsim_convolve_dev(self._h, C.c_void_p(in_dptr), (C.c_int64 * 3)(*dim_xyz), _ptr(psf),
                                              _dim(psf), method, C.c_void_p(out_dptr)))

    def adjust_image_dev(self, dptr, n, min_value, target_average, want_corr=True):
        corr = C.c_double()
        _lib.check(self._L.mvsim_adjust_image_dev(self._h, C.c_void_p(dptr), n, min_value, target_average,
                                                  C.byref(corr) if want_corr else None))
        return corr.value if want_corr else None

    def extract_slices_dev(self, in_dptr, dim_xyz, inc, snr, seed, stream, out_dptr):
        _lib.check(self._L.mvsim_extract_slices_dev(self._h, C.c_void_p(in_dptr), (C.c_int64 * 3)(*dim_xyz), inc, snr,
                                                    seed & 0xFFFFFFFFFFFFFFFF, stream, C.c_void_p(out_dptr)))

    # -- timings
    def enable_timing(self, enable: bool = True) -> None:
        _lib.check(self._L.mvsim_enable_timing(self._h, 1 if enable else 0))

    def timings(self) -> dict:
        t = Timings()
        _lib.check(self._L.mvsim_get_timings(self._h, C.byref(t)))
        return t.as_dict()

    # -- multi-GPU
    @staticmethod
    def comm_unique_id() -> bytes:
        buf = (C.c_ubyte * _lib.UNIQUE_ID_BYTES)()
        _lib.check(_lib.load().mvsim_comm_unique_id(buf))
        return bytes(buf)

    @staticmethod
    def comm_library_info() -> dict:
        """Path and version of the RCCL shared object libmvsim.so is bound to in this process."""
        L = _lib.load()
        buf = C.create_string_buffer(1024)
        ver = C.c_int(0)
        _lib.check(L.mvsim_comm_library_info(buf, len(buf), C.byref(ver)))
        v = int(ver.value)
        return {"path": buf.value.decode(), "version_code": v, "version": f"{v // 10000}.{(v // 100) % 100}.{v % 100}"}

    def comm_init(self, nranks: int, rank: int, uid: bytes) -> None:
        if len(uid) != _lib.UNIQUE_ID_BYTES:
            raise ValueError("unique id must be 128 bytes")
        buf = (C.c_ubyte * _lib.UNIQUE_ID_BYTES).from_buffer_copy(uid)
        _lib.check(self._L.mvsim_comm_init(self._h, nranks, rank, buf))

    def comm_broadcast_volume(self, dptr: int, count: int, root: int = 0) -> None:
        _lib.check(self._L.mvsim_comm_broadcast_volume(self._h, C.c_void_p(dptr), count, root))

    def comm_register_volume(self, dptr: int, count: int) -> None:
        """Collective: every rank registers the buffer it will broadcast into with option broadcast=peer_copy."""
        _lib.check(self._L.mvsim_comm_register_volume(self._h, C.c_void_p(dptr), count))

    def comm_unregister_volume(self, dptr: int) -> None:
        _lib.check(self._L.mvsim_comm_unregister_volume(self._h, C.c_void_p(dptr)))

    def comm_allreduce_sum(self, dptr: int, count: int) -> None:
        _lib.check(self._L.mvsim_comm_allreduce_sum(self._h, C.c_void_p(dptr), count))

    def comm_destroy(self) -> None:
        _lib.check(self._L.mvsim_comm_destroy(self._h))


class Group:
    """``mvsim_group``: ONE process driving several GPUs (what a JVM host is) -- one context per device plus an RCCL
    communicator over them.  ``broadcast_volume`` puts the ground truth on every device (host -> device 0 -> scatter +
    all-gather over xGMI); ``simulate_views`` runs view v on device v % ndev and returns the acquisitions."""

    def __init__(self, ndev: int, devices=None):
        self._L = _lib.load()
        self._h = C.c_void_p()
        arr = (C.c_int * ndev)(*devices) if devices is not None else None
        _lib.check(self._L.mvsim_group_create(ndev, arr, C.byref(self._h)))

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h:
            self._L.mvsim_group_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self) -> int:
        return int(self._L.mvsim_group_size(self._h))

    def set_option(self, name: str, value) -> None:
        for i in range(len(self)):
            _lib.check(self._L.mvsim_set_option(C.c_void_p(self._L.mvsim_group_ctx(self._h, i)), name.encode(), str(value).encode()))

    def broadcast_volume(self, gt) -> None:
        v = _as_volume(gt, "ground truth")
        self._shape = v.shape
        _lib.check(self._L.mvsim_group_broadcast_volume(self._h, _ptr(v), _dim(v)))

    def simulate_views(self, psfs: list, params: list) -> list:
        """psfs[v]: float32 (Kz,Ky,Kx) arrays of one common shape, normalised in place; params[v]: ViewParams."""
        n = len(psfs)
        if n != len(params):
            raise ValueError("one ViewParams per PSF")
        for q in psfs:
            _check_inplace(q, "psf")
            if q.shape != psfs[0].shape:
                raise ValueError("all PSFs must have the same shape")
        nz, ny, nx = self._shape
        acq = [np.empty((self._L.mvsim_extract_nz(nz, p.inc), ny, nx), dtype=np.float32) for p in params]
        pa = (C.c_void_p * n)(*[q.ctypes.data for q in psfs])
        oa = (C.c_void_p * n)(*[a.ctypes.data for a in acq])
        pv = (ViewParams * n)(*params)
        _lib.check(self._L.mvsim_group_simulate_views(self._h, pa, _dim(psfs[0]), pv, n, oa))
        return acq


def _check_inplace(a, name="image") -> None:
    if not isinstance(a, np.ndarray) or a.dtype != np.float32 or not a.flags.c_contiguous or not a.flags.writeable:
        raise ValueError(f"{name}: in-place operators need a writable C-contiguous float32 numpy array")
    if a.size == 0:
        raise ValueError(f"{name}: empty image")


def shard_views(n_views: int, nranks: int, rank: int) -> list[int]:
    """view v -> rank v % nranks (host logic only; usable without a GPU)."""
    L = _lib.load()
    cap = max(n_views, 1)
    buf = (C.c_int * cap)()
    cnt = L.mvsim_shard_views(n_views, nranks, rank, buf, cap)
    if cnt < 0:
        _lib.check(cnt)
    return [buf[i] for i in range(cnt)]


def _isqrt(v: int) -> int:
    import math
    return math.isqrt(v)


def walk_large_sphere(shape_zyx, minValue: float, maxValue: float, scale: int, halfPixelOffset: bool, rnd) -> list:
    """The host half of drawSpheres (SimulateMultiViewDataset.java:436-522) with the CALLER's generator: visit the voxels
    of the large sphere in ImgLib2 HyperSphereCursor order (z outermost, nested truncated radii, x fastest), draw
    ``radius = rnd.nextInt(10*scale) + 1`` and a double per voxel, and where the rounding test on that double selects
    the voxel, a second double for the intensity.  Returns the (cx, cy, cz, radius, value) list for the GPU splat."""
    nz, ny, nx = (int(v) for v in shape_zyx)
    c = (nx // 2, ny // 2, nz // 2)
    R = min(nx, ny, nz) // 2 - 47 * scale - 1
    if R < 0:
        raise ValueError("drawSpheres: image too small for this scale")
    max_radius = 10 * scale
    modulus = (7 * scale) ** 3
    off = 1 if halfPixelOffset else 0
    out = []
    for dz in range(-R, R + 1):
        r1 = _isqrt(R * R - dz * dz)
        for dy in range(-r1, r1 + 1):
            r0 = _isqrt(r1 * r1 - dy * dy)
            for dx in range(-r0, r0 + 1):
                radius = rnd.nextInt(max_radius) + 1
                rv = rnd.nextDouble()
                if int(np.floor(rv * 10000 + 0.5)) % modulus != 0:
                    continue
                value = rnd.nextDouble() * (maxValue - minValue) + minValue
                out.append((c[0] + dx + off, c[1] + dy + off, c[2] + dz, radius, np.float32(value)))
    return out


_default_ctx: Context | None = None


def default_context() -> Context:
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context()
    return _default_ctx


# ------------------------------------------------------------------------------------------------
# Mirror of the reference's static-method interface (same names, argument meaning, in-place
# behaviour).  File:line citations are into the reference tree.
# ------------------------------------------------------------------------------------------------
class Tools:
    """net.preibisch.simulation.Tools (numeric helpers on the hot path)."""

    @staticmethod
    def poissonProcess(img: np.ndarray, SNR: float, rnd=None) -> None:
        """Tools.java:73-86 -- in place; counts are not rescaled."""
        default_context().poisson_process(img, SNR, _seed_from(rnd))

    @staticmethod
    def save(img, file: str) -> None:
        """Tools.java:88-105 -- ImageJ big-endian float32 TIFF (stack for 3-D images)."""
        from . import tiffio
        tiffio.save_tiff(img, file)

    @staticmethod
    def open(file: str, square: bool = False) -> np.ndarray:
        """Tools.java:162-232, :290-300 -- float32 TIFF stack -> (Nz,Ny,Nx); ``square`` pads to a cube (makeSquare)."""
        from . import tiffio
        img = tiffio.open_tiff(file)
        return tiffio.make_square(img) if square else img

    @staticmethod
    def makeSquare(img) -> np.ndarray:
        """Tools.java:313-349"""
        from . import tiffio
        return tiffio.make_square(img)

    @staticmethod
    def normImage(img: np.ndarray) -> None:
        """Tools.java:112-118 -- in place, sum -> 1."""
        default_context().norm_image(img)

    @staticmethod
    def adjustImage(image: np.ndarray, minValue: float, targetAverage: float) -> float:
        """Tools.java:143-159 -- in place; returns the multiplicative correction."""
        return default_context().adjust_image(image, minValue, targetAverage)


class SimulateMultiViewDataset:
    """net.preibisch.simulation.SimulateMultiViewDataset -- per-view operators."""

    rnd = JavaRandom(464232194)        # SimulateMultiViewDataset.java:76
    minValue = float(np.float32(0.0001))   # :77
    avgIntensity = 1.0                 # :78

    @staticmethod
    def axisRotation(interval_dims_xyz, axis: int, degrees: int) -> np.ndarray:
        """:80-102 -- forward model T(+c) R T(-c) as a 3x4 row-major matrix."""
        L = _lib.load()
        m = (C.c_double * 12)()
        _lib.check(L.mvsim_axis_rotation((C.c_int64 * 3)(*interval_dims_xyz), axis, degrees, m))
        return np.array(m, dtype=np.float64).reshape(3, 4)

    @staticmethod
    def rotateAroundAxis(img, axis: int, degrees: int) -> np.ndarray:
        """:104-135"""
        return default_context().rotate_around_axis(img, axis, degrees)

    @staticmethod
    def attenuate3d(img, delta: float) -> np.ndarray:
        """:318-364"""
        return default_context().attenuate3d(img, delta)

    @staticmethod
    def convolve(img, psf: np.ndarray, service=None) -> np.ndarray:
        """:253-264 -- normalises ``psf`` in place; ``service`` (ExecutorService) accepted and ignored."""
        return default_context().convolve(img, psf)

    @staticmethod
    def extractSlices(img, inc: int, poissonSNR: float, rnd=None) -> np.ndarray:
        """:181-231 -- every inc-th slice; Poisson noise iff poissonSNR >= 0."""
        seed = _seed_from(rnd) if poissonSNR >= 0.0 else 0
        return default_context().extract_slices(img, inc, poissonSNR, seed)

    @staticmethod
    def poissonProcess(img, poissonSNR: float, rnd=None) -> np.ndarray:
        """:233-251 -- copy, then Tools.poissonProcess on the copy."""
        out = np.array(img, dtype=np.float32, order="C", copy=True)
        Tools.poissonProcess(out, poissonSNR, rnd)
        return out

    @staticmethod
    def simulate(halfPixelOffset: bool = False, rnd=None) -> np.ndarray:
        """:366-392 -- the 289^3 sphere phantom (rendered at 2x on the GPU, then down-sampled)."""
        return default_context().simulate_phantom(289, 2, halfPixelOffset, rnd)

    @staticmethod
    def drawSpheres(img: np.ndarray, minValue: float, maxValue: float, scale: int, halfPixelOffset: bool,
                    rnd=None) -> None:
        """:436-522, in place.  ``rnd`` may be this package's JavaRandom (its 48-bit state goes to the library, which
        replays the walk natively) or ANY object with java.util.Random's ``nextInt(bound)`` / ``nextDouble()`` -- the
        second caller passes a plain ``new Random(seed)`` (SimulateTileStitching.java:93,108): then the walk over the
        large sphere runs here on the host with the caller's generator and only the compositing goes to the GPU."""
        rnd = rnd if rnd is not None else SimulateMultiViewDataset.rnd
        if isinstance(rnd, JavaRandom):
            default_context().draw_spheres(img, minValue, maxValue, scale, halfPixelOffset, rnd)
        else:
            default_context().splat_spheres(img, walk_large_sphere(img.shape, minValue, maxValue, scale, halfPixelOffset, rnd))

    @staticmethod
    def downSample2x(img) -> np.ndarray:
        """:394-424"""
        return default_context().downsample2x(img)

    @staticmethod
    def makeIsotropic(img, inc: int) -> np.ndarray:
        """:144-171"""
        return default_context().make_isotropic(img, inc)

    @staticmethod
    def normalizeWeights(weights: list, osem: float) -> None:
        """:615-640 (the block of main() after the view loop) -- in place on the list of aligned weight images."""
        default_context().normalize_weights(weights, osem)

    @staticmethod
    def computeWeightImage(img, delta: float = 0.0) -> np.ndarray:
        """:280-316 -- only the interval of ``img`` is used; ``delta`` is ignored as in the reference."""
        return default_context().compute_weight_image(np.shape(img))


class SimulateTileStitching:
    """Mirror of ``net.preibisch.simulation.SimulateTileStitching`` (SimulateTileStitching.java:43-189), the second
    caller of the per-view operators: two overlapping tiles cut out of one convolved phantom (optionally with a
    half-pixel shift between them), each run through ``extractSlices`` with its own seeded generator.

    Differences from the reference, all host-side: the two phantoms/convolutions of ``init`` run one after the other
    on the GPU instead of on two pool threads (in the reference both threads normalise the shared PSF in place at
    the same time); ``service`` is accepted and ignored; ``psf`` may be passed in instead of being read from
    ``dir + "Angle0.tif"``.
    """

    dir = "src/main/resources/"

    def __init__(self, rnd=None, halfPixelOffset: bool = False, overlapRatio=(0.2, 0.2, 0.2), service=None,
                 psf: np.ndarray | None = None):
        self.rnd = rnd if rnd is not None else JavaRandom(464232194)          # :66-69
        self.service = service
        self.lightsheetSpacing = 3                                              # :48
        self.attenuation = float(np.float32(0.01))                              # :49, a Java float widened to double
        self.psf = psf if psf is not None else Tools.open(self.dir + "Angle0.tif", True)
        self.init(overlapRatio, halfPixelOffset)

    def init(self, overlapRatio, halfPixelOffset: bool) -> None:
        """:77-129"""
        S = SimulateMultiViewDataset
        self.halfPixelOffset = bool(halfPixelOffset)
        seed = self.rnd.nextInt()                                               # same phantom for both variants

        def rendered(half: bool) -> np.ndarray:
            gt = S.simulate(half, JavaRandom(seed))
            att = S.attenuate3d(gt, self.attenuation)
            con = S.convolve(att, self.psf, self.service)
            Tools.adjustImage(con, S.minValue, S.avgIntensity)
            return con

        self.con = rendered(False)
        self.conHalfPixel = rendered(True)
        nz, ny, nx = self.con.shape
        dims = (nx, ny, nz)
        self.overlap = [int(np.floor(dims[d] * overlapRatio[d] / 2 + 0.5)) for d in range(3)]     # Math.round
        self.min = [0, 0, 0]
        self.max = [0, 0, 0]

    def getInterval(self, tile: int):
        """:226-246 -- inclusive (min, max) per dimension, x first.  Tile 1 starts at dimension(0)/2 - overlap in
        every dimension, as in the reference."""
        nz, ny, nx = self.con.shape
        dims = (nx, ny, nz)
        lo, hi = [0, 0, 0], [d - 1 for d in dims]
        if tile == 0:
            hi = [dims[d] // 2 + self.overlap[d] for d in range(3)]
        else:
            lo = [dims[0] // 2 - self.overlap[d] for d in range(3)]
        return lo, hi

    def _tile(self, img: np.ndarray, tile: int) -> np.ndarray:
        lo, hi = self.getInterval(tile)
        return np.ascontiguousarray(img[lo[2]:hi[2] + 1, lo[1]:hi[1] + 1, lo[0]:hi[0] + 1])

    def getNextPair(self, snr: float):
        """:133-198 -- (left, right) tiles; each draws its extractSlices seed from ``new Random(seedN)``."""
        S = SimulateMultiViewDataset
        seed0 = self.rnd.nextInt()
        seed1 = self.rnd.nextInt()
        split0 = S.extractSlices(self._tile(self.con, 0), self.lightsheetSpacing, snr, JavaRandom(seed0))
        src1 = self.conHalfPixel if self.halfPixelOffset else self.con
        split1 = S.extractSlices(self._tile(src1, 1), self.lightsheetSpacing, snr, JavaRandom(seed1))
        return split0, split1

    def getCorrectTranslation(self):
        """:200-224 -- shift of the right tile relative to the left one, (x, y, z) with z in acquired planes."""
        lo, hi = self.getInterval(1)
        self.min, self.max = lo, hi
        t = [float(v) for v in lo]
        if self.halfPixelOffset:
            t[0] -= 0.5
            t[1] -= 0.5
        t[2] /= self.lightsheetSpacing
        return t
