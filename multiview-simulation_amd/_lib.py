"""ctypes binding of libmvsim.so (include/mvsim.h).  No CPU fallback: if the HIP library is
missing or no gfx950 device is usable, every compute entry point raises."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libmvsim.so")

MVSIM_OK, MVSIM_EINVAL, MVSIM_ENOMEM, MVSIM_EHIP, MVSIM_EFFT, MVSIM_ERCCL, MVSIM_ENODEV = 0, -1, -2, -3, -4, -5, -6
UNIQUE_ID_BYTES = 128


class MvsimError(RuntimeError):
    """HIP / rocFFT / RCCL failure inside libmvsim (Java side: RuntimeException)."""

    def __init__(self, status: int, message: str):
        super().__init__(f"libmvsim status {status}: {message}")
        self.status = status


class MvsimNoDeviceError(MvsimError):
    pass


class ViewParams(C.Structure):
    _fields_ = [
        ("axis", C.c_int32), ("degrees", C.c_int32), ("delta", C.c_double), ("min_value", C.c_float),
        ("target_average", C.c_float), ("inc", C.c_int32), ("snr", C.c_float), ("seed", C.c_uint64),
        ("stream", C.c_uint32), ("conv_method", C.c_int32),
    ]


class BcastOp(C.Structure):
    _fields_ = [("stage", C.c_int32), ("kind", C.c_int32), ("peer", C.c_int32), ("pad", C.c_int32), ("first", C.c_int64), ("count", C.c_int64)]


class ViewOutputs(C.Structure):
    _fields_ = [("rot", C.c_void_p), ("att", C.c_void_p), ("con", C.c_void_p), ("acq", C.c_void_p)]


class IterationOutputs(C.Structure):
    _fields_ = [("iso", C.c_void_p), ("view", C.c_void_p), ("view_weights", C.c_void_p), ("view_psf", C.c_void_p)]


class Sphere(C.Structure):
    _fields_ = [("cx", C.c_int32), ("cy", C.c_int32), ("cz", C.c_int32), ("radius", C.c_int32), ("value", C.c_float)]


class Timings(C.Structure):
    _fields_ = [(n, C.c_float) for n in
                ("rotate_ms", "attenuate_ms", "psf_ms", "convolve_ms", "adjust_ms", "extract_ms", "total_ms",
                 "pass_a_ms", "pass_b_ms", "pass_c_ms", "pass_d_ms", "pass_e_ms")]

    def as_dict(self):
        return {n: float(getattr(self, n)) for n, _ in self._fields_}


_i64p = C.POINTER(C.c_int64)
_vp = C.c_void_p

# name -> (restype, argtypes).  Must list every symbol declared in include/mvsim.h.
SIGNATURES = {
    "mvsim_version": (C.c_char_p, []),
    "mvsim_last_error": (C.c_char_p, []),
    "mvsim_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "mvsim_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "mvsim_destroy": (C.c_int, [_vp]),
    "mvsim_set_stream": (C.c_int, [_vp, _vp]),
    "mvsim_synchronize": (C.c_int, [_vp]),
    "mvsim_join": (C.c_int, [_vp]),
    "mvsim_set_option": (C.c_int, [_vp, C.c_char_p, C.c_char_p]),
    "mvsim_release_caches": (C.c_int, [_vp]),
    "mvsim_dev_alloc": (C.c_int, [_vp, C.c_size_t, C.POINTER(_vp)]),
    "mvsim_dev_free": (C.c_int, [_vp, _vp]),
    "mvsim_upload": (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    "mvsim_dev_memset": (C.c_int, [_vp, _vp, C.c_int, C.c_size_t]),
    "mvsim_host_alloc": (C.c_int, [_vp, C.c_size_t, C.POINTER(C.c_void_p)]),
    "mvsim_host_free": (C.c_int, [_vp, _vp]),
    "mvsim_download": (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    "mvsim_axis_rotation": (C.c_int, [_i64p, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    "mvsim_extract_nz": (C.c_int64, [C.c_int64, C.c_int]),
    "mvsim_isotropic_nz": (C.c_int64, [C.c_int64, C.c_int]),
    "mvsim_poisson_mul": (C.c_double, [C.c_double]),
    "mvsim_rotate_around_axis": (C.c_int, [_vp, _vp, _i64p, C.c_int, C.c_int, _vp]),
    "mvsim_attenuate3d": (C.c_int, [_vp, _vp, _i64p, C.c_double, _vp]),
    "mvsim_norm_image": (C.c_int, [_vp, _vp, C.c_int64]),
    "mvsim_convolve": (C.c_int, [_vp, _vp, _i64p, _vp, _i64p, C.c_int, _vp]),
    "mvsim_adjust_image": (C.c_int, [_vp, _vp, C.c_int64, C.c_float, C.c_float, C.POINTER(C.c_double)]),
    "mvsim_extract_slices": (C.c_int, [_vp, _vp, _i64p, C.c_int, C.c_float, C.c_uint64, C.c_uint32, _vp]),
    "mvsim_poisson_process": (C.c_int, [_vp, _vp, C.c_int64, C.c_double, C.c_uint64, C.c_uint32, C.c_uint64]),
    "mvsim_make_isotropic": (C.c_int, [_vp, _vp, _i64p, C.c_int, _vp]),
    "mvsim_compute_weight_image": (C.c_int, [_vp, _i64p, _vp]),
    "mvsim_rotate_around_axis_dev": (C.c_int, [_vp, _vp, _i64p, C.c_int, C.c_int, _vp]),
    "mvsim_attenuate3d_dev": (C.c_int, [_vp, _vp, _i64p, C.c_double, _vp]),
    "mvsim_convolve_dev": (C.c_int, [_vp, _vp, _i64p, _vp, _i64p, C.c_int, _vp]),
    "mvsim_adjust_image_dev": (C.c_int, [_vp, _vp, C.c_int64, C.c_float, C.c_float, C.POINTER(C.c_double)]),
    "mvsim_extract_slices_dev": (C.c_int, [_vp, _vp, _i64p, C.c_int, C.c_float, C.c_uint64, C.c_uint32, _vp]),
    "mvsim_make_isotropic_dev": (C.c_int, [_vp, _vp, _i64p, C.c_int, _vp]),
    "mvsim_draw_spheres": (C.c_int, [_vp, _vp, _i64p, C.c_double, C.c_double, C.c_int, C.c_int,
                                     C.POINTER(C.c_uint64), C.POINTER(C.c_int64)]),
    "mvsim_draw_spheres_dev": (C.c_int, [_vp, _vp, _i64p, C.c_double, C.c_double, C.c_int, C.c_int,
                                         C.POINTER(C.c_uint64), C.POINTER(C.c_int64)]),
    "mvsim_downsample2x": (C.c_int, [_vp, _vp, _i64p, _vp]),
    "mvsim_downsample2x_dev": (C.c_int, [_vp, _vp, _i64p, _vp]),
    "mvsim_compute_weight_image_dev": (C.c_int, [_vp, _i64p, _vp]),
    "mvsim_sum_views_dev": (C.c_int, [_vp, C.POINTER(_vp), C.c_int, C.c_int64, _vp]),
    "mvsim_normalize_weights_dev": (C.c_int, [_vp, C.POINTER(_vp), C.c_int, C.c_int64, _vp, C.c_float]),
    "mvsim_normalize_weights": (C.c_int, [_vp, C.POINTER(_vp), C.c_int, C.c_int64, C.c_float]),
    "mvsim_view_params_default": (None, [C.POINTER(ViewParams)]),
    "mvsim_simulate_view_dev": (C.c_int, [_vp, _vp, _i64p, _vp, _i64p, C.POINTER(ViewParams),
                                          C.POINTER(ViewOutputs), C.POINTER(C.c_double)]),
    "mvsim_get_plane_stats": (C.c_int, [_vp, _i64p]),
    "mvsim_get_queue_stats": (C.c_int, [_vp, _i64p]),
    "mvsim_get_transfer_stats": (C.c_int, [_vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "mvsim_simulate_views_dev": (C.c_int, [_vp, _vp, _i64p, C.POINTER(_vp), _i64p, C.POINTER(ViewParams),
                                           C.POINTER(ViewOutputs), C.c_int]),
    "mvsim_simulate_views": (C.c_int, [_vp, _vp, _i64p, C.POINTER(_vp), _i64p, C.POINTER(ViewParams), C.POINTER(_vp), C.c_int]),
    "mvsim_simulate_iteration_dev": (C.c_int, [_vp, _vp, _i64p, _vp, _i64p, C.POINTER(ViewParams), C.c_int,
                                               C.POINTER(ViewOutputs), C.POINTER(IterationOutputs)]),
    "mvsim_simulate_view": (C.c_int, [_vp, _vp, _i64p, _vp, _i64p, C.POINTER(ViewParams),
                                      C.POINTER(ViewOutputs), C.POINTER(C.c_double)]),
    "mvsim_splat_spheres": (C.c_int, [_vp, _vp, _i64p, _vp, C.c_int64]),
    "mvsim_splat_spheres_dev": (C.c_int, [_vp, _vp, _i64p, _vp, C.c_int64]),
    "mvsim_simulate_view_async": (C.c_int, [_vp, _vp, C.c_uint64, _i64p, _vp, _i64p, C.POINTER(ViewParams),
                                            C.POINTER(ViewOutputs), C.POINTER(C.c_int64)]),
    "mvsim_wait": (C.c_int, [_vp, C.c_int64, C.POINTER(C.c_double)]),
    "mvsim_simulate_view_zslabs": (C.c_int, [_vp, C.POINTER(_vp), _i64p, C.c_int, _i64p, _vp, _i64p, C.POINTER(ViewParams),
                                             C.POINTER(_vp), _i64p, C.c_int, C.POINTER(C.c_double)]),
    "mvsim_fft_geometry": (C.c_int, [_i64p, _i64p, _i64p]),
    "mvsim_rotate_around_axis_zslabs": (C.c_int, [_vp, C.POINTER(_vp), _i64p, C.c_int, _i64p, C.c_int, C.c_int, C.POINTER(_vp), _i64p, C.c_int]),
    "mvsim_attenuate3d_zslabs": (C.c_int, [_vp, C.POINTER(_vp), _i64p, C.c_int, _i64p, C.c_double, C.POINTER(_vp), _i64p, C.c_int]),
    "mvsim_convolve_zslabs": (C.c_int, [_vp, C.POINTER(_vp), _i64p, C.c_int, _i64p, _vp, _i64p, C.c_int, C.POINTER(_vp), _i64p, C.c_int]),
    "mvsim_extract_slices_zslabs": (C.c_int, [_vp, C.POINTER(_vp), _i64p, C.c_int, _i64p, C.c_int, C.c_float, C.c_uint64, C.c_uint32,
                                              C.POINTER(_vp), _i64p, C.c_int]),
    "mvsim_stencil_geometry": (C.c_int, [_i64p, _i64p]),
    "mvsim_enable_timing": (C.c_int, [_vp, C.c_int]),
    "mvsim_get_timings": (C.c_int, [_vp, C.POINTER(Timings)]),
    "mvsim_comm_unique_id": (C.c_int, [C.POINTER(C.c_ubyte)]),
    "mvsim_comm_init": (C.c_int, [_vp, C.c_int, C.c_int, C.POINTER(C.c_ubyte)]),
    "mvsim_comm_library_info": (C.c_int, [C.c_char_p, C.c_size_t, C.POINTER(C.c_int)]),
    "mvsim_comm_broadcast_volume": (C.c_int, [_vp, _vp, C.c_int64, C.c_int]),
    "mvsim_comm_broadcast_plan": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int, _vp, C.c_int, C.POINTER(C.c_int)]),
    "mvsim_comm_register_volume": (C.c_int, [_vp, _vp, C.c_int64]),
    "mvsim_comm_unregister_volume": (C.c_int, [_vp, _vp]),
    "mvsim_comm_allreduce_sum": (C.c_int, [_vp, _vp, C.c_int64]),
    "mvsim_comm_allreduce_sum_f64": (C.c_int, [_vp, C.POINTER(C.c_double)]),
    "mvsim_slab_range": (C.c_int, [C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "mvsim_view_slab_convolve_dev": (C.c_int, [_vp, _vp, _i64p, _vp, _i64p, _vp, C.c_int64, C.c_int64,
                                               C.POINTER(C.c_double)]),
    "mvsim_view_slab_finish_dev": (C.c_int, [_vp, _i64p, _vp, C.c_int64, C.c_int64, C.c_double, _vp,
                                             C.POINTER(C.c_int64)]),
    "mvsim_view_slab_dev": (C.c_int, [_vp, _vp, _vp, _i64p, _vp, _i64p, _vp, C.c_int64, C.c_int64, _vp, C.POINTER(C.c_int64)]),
    "mvsim_comm_allreduce_sum_f64_dev": (C.c_int, [_vp, _vp, _vp]),
    "mvsim_host_copy": (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    "mvsim_comm_destroy": (C.c_int, [_vp]),
    "mvsim_group_create": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.POINTER(_vp)]),
    "mvsim_group_destroy": (C.c_int, [_vp]),
    "mvsim_group_size": (C.c_int, [_vp]),
    "mvsim_group_ctx": (_vp, [_vp, C.c_int]),
    "mvsim_group_broadcast_volume": (C.c_int, [_vp, _vp, _i64p]),
    "mvsim_group_simulate_views": (C.c_int, [_vp, C.POINTER(_vp), _i64p, C.POINTER(ViewParams), C.c_int, C.POINTER(_vp)]),
    "mvsim_shard_views": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int]),
}

_lib = None


def load():
    """Load libmvsim.so (built in-tree by ``build.py``).  Raises ImportError when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError => the .so does not match include/mvsim.h
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status: int) -> None:
    if status == MVSIM_OK:
        return
    msg = (load().mvsim_last_error() or b"").decode("utf-8", "replace")
    if status == MVSIM_EINVAL:
        raise ValueError(msg)             # Java side: IllegalArgumentException
    if status == MVSIM_ENOMEM:
        raise MemoryError(msg)            # Java side: OutOfMemoryError
    if status == MVSIM_ENODEV:
        raise MvsimNoDeviceError(status, msg)
    raise MvsimError(status, msg)
