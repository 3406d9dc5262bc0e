"""java/jni/mvsim_jni.cpp EXECUTED against a fake JNIEnv (tests/jni_fake/fake_jni.cpp).

The image has no JDK, so the Java half has never run here.  The native half of the binding can: built against the jni.h subset of
tests/jni_stub with the JNIEnv members defined over an in-process object table, the Java_net_preibisch_simulation_gpu_MvsimNative_*
functions are called from ctypes with fake long[]/int[]/FloatBuffer objects.  Checked: the capacity checks in front of every pointer,
the status -> exception-class mapping, that no JNI call is made with an exception pending, and (on the GPU) that what comes out of the
shim is bit-identical to the same call through the C ABI.  Not checked, and not claimed: anything about a real JVM.
"""
import ctypes as C
import os
import re
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "multiview-simulation_amd")
PREFIX = "Java_net_preibisch_simulation_gpu_MvsimNative_"

i32, i64, f32, f64, u8, ptr = C.c_int32, C.c_int64, C.c_float, C.c_double, C.c_ubyte, C.c_void_p

# (result, arguments after JNIEnv* and jclass) of the functions the tests call -- the Java declarations of MvsimNative.java
SIGNATURES = {
    "create": (i64, [i32]),
    "destroy": (None, [i64]),
    "deviceCount": (i32, []),
    "rotateAroundAxis": (None, [i64, ptr, ptr, i32, i32, ptr]),
    "convolve": (None, [i64, ptr, ptr, ptr, ptr, i32, ptr]),
    "convolveSlabs": (None, [i64, ptr, ptr, ptr, ptr, ptr, i32, ptr, ptr]),
    "extractSlices": (None, [i64, ptr, ptr, i32, f32, i64, i32, ptr]),
    "axisRotation": (None, [ptr, i32, i32, ptr]),
    "allocPinned": (ptr, [i64, i64]),
    "freePinned": (None, [ptr]),
    "copyFloats": (None, [i64, ptr, i64, ptr, i32, i32, u8]),
    "drawSpheres": (i64, [i64, ptr, ptr, f64, f64, i32, u8, ptr]),
    "splatSpheres": (None, [i64, ptr, ptr, ptr, ptr]),
    "normalizeWeights": (None, [i64, ptr, i64, f32]),
    "simulateView": (f64, [i64, ptr, ptr, ptr, ptr, i32, i32, f64, f32, f32, i32, f32, i64, i32, ptr, ptr, ptr, ptr]),
    "simulateViewAsync": (i64, [i64, ptr, i64, ptr, ptr, ptr, i32, i32, f64, f32, f32, i32, f32, i64, i32, ptr]),
    "waitView": (f64, [i64, i64]),
    "simulateViewsBatch": (None, [i64, ptr, ptr, ptr, ptr, ptr, f64, f32, f32, i32, f32, ptr, ptr]),
}


class FakeJvm:
    """The shim + the fake JNIEnv in one shared object, and the little of `java.nio` the tests need."""

    def __init__(self, so_path):
        self.lib = C.CDLL(so_path)
        lib = self.lib
        lib.fake_env.restype = ptr
        for name, args in (("fake_long_array", [C.POINTER(i64), C.c_int]), ("fake_int_array", [C.POINTER(i32), C.c_int]),
                           ("fake_float_array", [C.POINTER(f32), C.c_int]), ("fake_double_array", [C.c_int]),
                           ("fake_object_array", [C.POINTER(ptr), C.c_int]), ("fake_buffer", [ptr, i64]), ("fake_buffer_address", [ptr])):
            getattr(lib, name).restype = ptr
            getattr(lib, name).argtypes = args
        lib.fake_buffer_capacity.restype = i64
        lib.fake_buffer_capacity.argtypes = [ptr]
        lib.fake_read_longs.argtypes = [ptr, C.POINTER(i64), C.c_int]
        lib.fake_read_doubles.argtypes = [ptr, C.POINTER(f64), C.c_int]
        lib.fake_read_floats.argtypes = [ptr, C.POINTER(f32), C.c_int]
        lib.fake_take_exception.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int]
        lib.fake_violations.argtypes = [C.c_char_p, C.c_int]
        self.env = lib.fake_env()
        self._keep = []
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, PREFIX + name)
            fn.restype = res
            fn.argtypes = [ptr, ptr] + args

    def reset(self):
        self.lib.fake_reset()
        self._keep.clear()

    def call(self, name, *args):
        return getattr(self.lib, PREFIX + name)(self.env, None, *args)

    def longs(self, values):
        a = (i64 * len(values))(*values)
        return self.lib.fake_long_array(a, len(values))

    def ints(self, values):
        a = (i32 * len(values))(*values)
        return self.lib.fake_int_array(a, len(values))

    def floats(self, values):
        a = (f32 * len(values))(*values)
        return self.lib.fake_float_array(a, len(values))

    def doubles(self, n):
        return self.lib.fake_double_array(n)

    def objects(self, objs):
        a = (ptr * len(objs))(*objs)
        return self.lib.fake_object_array(a, len(objs))

    def float_buffer(self, array, capacity=None):
        """A direct FloatBuffer over a float32 numpy array (kept alive until reset); `capacity` overrides what it reports."""
        assert array.dtype == np.float32 and array.flags.c_contiguous
        self._keep.append(array)
        return self.lib.fake_buffer(array.ctypes.data, array.size if capacity is None else capacity)

    def heap_buffer(self, capacity):
        return self.lib.fake_buffer(None, capacity)

    def read_longs(self, obj, n):
        out = (i64 * n)()
        assert self.lib.fake_read_longs(obj, out, n) == 0
        return list(out)

    def float_array(self, array):
        """A Java float[] holding a copy of `array`."""
        a = np.ascontiguousarray(array, dtype=np.float32)
        return self.lib.fake_float_array(a.ctypes.data_as(C.POINTER(f32)), a.size)

    def read_floats(self, obj, n):
        out = np.empty(n, np.float32)
        assert self.lib.fake_read_floats(obj, out.ctypes.data_as(C.POINTER(f32)), n) == 0
        return out

    def read_doubles(self, obj, n):
        out = (f64 * n)()
        assert self.lib.fake_read_doubles(obj, out, n) == 0
        return list(out)

    def exception(self):
        """(class, message) of the pending exception, cleared; None when nothing was thrown."""
        cls, msg = C.create_string_buffer(256), C.create_string_buffer(1024)
        if not self.lib.fake_take_exception(cls, 256, msg, 1024):
            return None
        return cls.value.decode(), msg.value.decode()

    def assert_clean(self):
        what = C.create_string_buffer(256)
        assert self.lib.fake_violations(what, 256) == 0, what.value.decode()


@pytest.fixture(scope="module")
def jvm(tmp_path_factory):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    if not os.path.exists(os.path.join(PKG, "libmvsim.so")):
        pytest.fail("libmvsim.so is not built (python -c 'import __graft_entry__ as g; g.build()')")
    so = str(tmp_path_factory.mktemp("jni") / "libmvsim_jni_fake.so")
    r = subprocess.run([gxx, "-shared", "-fPIC", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "tests", "jni_stub"),
                        "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "java", "jni", "mvsim_jni.cpp"),
                        os.path.join(ROOT, "tests", "jni_fake", "fake_jni.cpp"), "-L" + PKG, "-lmvsim", "-Wl,-rpath," + PKG, "-o", so],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    j = FakeJvm(so)
    yield j
    j.reset()


@pytest.fixture()
def vm(jvm):
    jvm.reset()
    yield jvm
    jvm.assert_clean()


IAE = "java/lang/IllegalArgumentException"


# ---- without a GPU: everything the shim decides BEFORE the C ABI sees a pointer -------------------------------------------------
def test_every_native_method_is_exported_by_the_built_shim(jvm):
    """The shared object built from java/jni/mvsim_jni.cpp exports one Java_..._<name> per `native` method of MvsimNative.java."""
    java = open(os.path.join(ROOT, "java", "src", "main", "java", "net", "preibisch", "simulation", "gpu", "MvsimNative.java")).read()
    natives = re.findall(r"\bnative\s+[\w\[\]<>.]+\s+(\w+)\s*\(", java)
    assert len(natives) >= 30
    for name in natives:
        assert hasattr(jvm.lib, PREFIX + name), name


def test_axis_rotation_fills_the_double_array_like_the_c_abi(vm, mvs):
    m12 = vm.doubles(12)
    vm.call("axisRotation", vm.longs([40, 30, 20]), 1, 35, m12)
    assert vm.exception() is None
    got = np.array(vm.read_doubles(m12, 12)).reshape(3, 4)
    assert np.array_equal(got, mvs.SimulateMultiViewDataset.axisRotation((40, 30, 20), 1, 35))
    assert np.allclose(got[:, :3] @ got[:, :3].T, np.eye(3), atol=1e-12)


def test_copy_floats_between_heap_arrays_and_blocks(vm):
    """MvsimNative.copyFloats (round 6): the facade's bulk copies between a Java float[] and a staging block on the library's host
    threads.  The array is held with GetPrimitiveArrayCritical -- no JNI call inside the region (the fake counts that as a violation)
    and the region is closed again --, ranges are checked before anything moves, 20 MB travel through the multi-threaded path
    (4 MiB chunks), a few floats through the plain one."""
    rng = np.random.default_rng(9)
    n = 5_000_003
    src = rng.random(n, dtype=np.float32)
    block = src.copy()
    arr = vm.float_array(np.zeros(n + 7, np.float32))
    vm.call("copyFloats", 0, vm.float_buffer(block), 0, arr, 7, n, 1)                   # block -> array[7:]
    assert vm.exception() is None and vm.lib.fake_critical_depth() == 0
    got = vm.read_floats(arr, n + 7)
    assert np.array_equal(got[7:], src) and not got[:7].any()
    back = np.zeros(n, np.float32)
    vm.call("copyFloats", 0, vm.float_buffer(back), 3, arr, 7 + 3, n - 3, 0)            # array[10:] -> block[3:]
    assert vm.exception() is None and vm.lib.fake_critical_depth() == 0
    assert np.array_equal(back[3:], src[3:]) and not back[:3].any()
    small = vm.float_array(np.arange(10, dtype=np.float32))
    blk = np.zeros(10, np.float32)
    vm.call("copyFloats", 0, vm.float_buffer(blk), 2, small, 4, 5, 0)
    assert vm.exception() is None and np.array_equal(blk, [0, 0, 4, 5, 6, 7, 8, 0, 0, 0])
    vm.call("copyFloats", 0, vm.float_buffer(blk), 0, small, 0, 0, 1)                  # nothing to do
    assert vm.exception() is None
    # ranges: outside the array, outside the block, a heap buffer, a null array
    vm.call("copyFloats", 0, vm.float_buffer(blk), 0, small, 6, 5, 1)
    assert vm.exception() == ("java/lang/ArrayIndexOutOfBoundsException", "copyFloats: range outside the array")
    vm.call("copyFloats", 0, vm.float_buffer(blk), 0, None, 0, 5, 1)
    assert vm.exception() == ("java/lang/ArrayIndexOutOfBoundsException", "copyFloats: range outside the array")
    vm.call("copyFloats", 0, vm.float_buffer(blk), 6, small, 0, 5, 1)
    assert vm.exception() == (IAE, "copyFloats: range outside the block")
    vm.call("copyFloats", 0, vm.heap_buffer(10), 0, small, 0, 5, 1)
    assert vm.exception() == (IAE, "a direct FloatBuffer is required")
    assert vm.lib.fake_critical_depth() == 0 and np.array_equal(vm.read_floats(small, 10), np.arange(10, dtype=np.float32))


def test_wrong_arguments_become_illegal_argument_exceptions_before_any_copy(vm):
    a = np.zeros(4 * 4 * 4, dtype=np.float32)
    dim = vm.longs([4, 4, 4])
    # dims: long[2], a zero extent
    vm.call("rotateAroundAxis", 0, vm.float_buffer(a), vm.longs([4, 4]), 0, 10, vm.float_buffer(a.copy()))
    assert vm.exception() == (IAE, "dims: long[3] expected")
    vm.call("rotateAroundAxis", 0, vm.float_buffer(a), vm.longs([4, 0, 4]), 0, 10, vm.float_buffer(a.copy()))
    assert vm.exception() == (IAE, "dims must be >= 1")
    # an output buffer one float short; a heap (non-direct) buffer; a null buffer
    vm.call("rotateAroundAxis", 0, vm.float_buffer(a), dim, 0, 10, vm.float_buffer(a.copy(), capacity=63))
    assert vm.exception() == (IAE, "rotateAroundAxis: output buffer smaller than the dimensions")
    vm.call("rotateAroundAxis", 0, vm.heap_buffer(64), dim, 0, 10, vm.float_buffer(a.copy()))
    assert vm.exception() == (IAE, "a direct FloatBuffer is required")
    vm.call("rotateAroundAxis", 0, None, dim, 0, 10, vm.float_buffer(a.copy()))
    assert vm.exception() == (IAE, "rotateAroundAxis: input buffer smaller than the dimensions")
    # extractSlices: the output must hold (Nz-1)/inc+1 planes -- 2 planes of 16 for Nz = 4, inc = 3
    out = np.zeros(31, dtype=np.float32)
    vm.call("extractSlices", 0, vm.float_buffer(a), dim, 3, 0.0, 1, 0, vm.float_buffer(out))
    assert vm.exception() == (IAE, "extractSlices: output buffer smaller than (Nz-1)/inc+1 planes")
    vm.call("extractSlices", 0, vm.float_buffer(a), dim, 0, 0.0, 1, 0, vm.float_buffer(out))
    assert vm.exception() == (IAE, "extractSlices: inc must be >= 1")
    # axisRotation: double[11]
    vm.call("axisRotation", dim, 0, 10, vm.doubles(11))
    assert vm.exception() == (IAE, "axisRotation: double[12] expected")


def test_list_arguments_are_checked_against_each_other(vm):
    a = np.zeros(64, dtype=np.float32)
    dim, kdim = vm.longs([4, 4, 4]), vm.longs([3, 3, 3])
    psf = np.ones(27, dtype=np.float32)
    # splatSpheres: 4 ints per sphere
    vm.call("splatSpheres", 0, vm.float_buffer(a), dim, vm.ints([1, 1, 1, 1, 2, 2, 2]), vm.floats([1.0, 2.0]))
    assert vm.exception() == (IAE, "splatSpheres: 4 ints per sphere expected")
    # simulateViewsBatch: one PSF, seed and buffer per view
    acqs = vm.objects([vm.float_buffer(np.zeros(64, dtype=np.float32))])
    vm.call("simulateViewsBatch", 0, vm.float_buffer(a), dim, vm.objects([vm.float_buffer(psf), vm.float_buffer(psf.copy())]), kdim,
            vm.ints([0, 90]), 0.0, 0.0, 1.0, 1, 0.0, vm.longs([1, 2]), acqs)
    assert vm.exception() == (IAE, "simulateViewsBatch: one PSF, seed and acquisition buffer per view")
    # ... and every acquisition buffer against nx * ny * ((nz-1)/inc+1)
    acqs = vm.objects([vm.float_buffer(np.zeros(64, dtype=np.float32)), vm.float_buffer(np.zeros(63, dtype=np.float32))])
    vm.call("simulateViewsBatch", 0, vm.float_buffer(a), dim, vm.objects([vm.float_buffer(psf), vm.float_buffer(psf.copy())]), kdim,
            vm.ints([0, 90]), 0.0, 0.0, 1.0, 1, 0.0, vm.longs([1, 2]), acqs)
    assert vm.exception() == (IAE, "simulateViewsBatch: acquisition buffer smaller than nx * ny * ((nz - 1) / inc + 1) floats")
    # z-slab lists: the plane counts must add up to nz, and every buffer must hold its planes
    halves = [np.zeros(32, dtype=np.float32) for _ in range(4)]
    vm.call("convolveSlabs", 0, vm.objects([vm.float_buffer(halves[0]), vm.float_buffer(halves[1])]), vm.longs([2, 1]), dim, vm.float_buffer(psf),
            kdim, 0, vm.objects([vm.float_buffer(halves[2]), vm.float_buffer(halves[3])]), vm.longs([2, 2]))
    assert vm.exception() == (IAE, "convolve: input slabs do not match the dimensions")
    vm.call("convolveSlabs", 0, vm.objects([vm.float_buffer(halves[0]), vm.float_buffer(halves[1])]), vm.longs([2, 2]), dim, vm.float_buffer(psf),
            kdim, 0, vm.objects([vm.float_buffer(halves[2]), vm.float_buffer(halves[3], capacity=31)]), vm.longs([2, 2]))
    assert vm.exception() == (IAE, "convolve: output slabs do not match the dimensions")
    # normalizeWeights: 1..32 views
    vm.call("normalizeWeights", 0, vm.objects([]), 64, 1.0)
    assert vm.exception() == (IAE, "normalizeWeights: 1..32 views")


def test_status_codes_map_to_exception_classes_with_the_library_message(vm):
    """MVSIM_EINVAL from the C ABI itself (a null context handle, an axis out of range) arrives as IllegalArgumentException carrying
    mvsim_last_error(); a failing call returns a neutral value."""
    a = np.zeros(64, dtype=np.float32)
    vm.call("rotateAroundAxis", 0, vm.float_buffer(a), vm.longs([4, 4, 4]), 0, 10, vm.float_buffer(a.copy()))
    cls, msg = vm.exception()
    assert cls == IAE and "ctx is null" in msg
    vm.call("axisRotation", vm.longs([4, 4, 4]), 7, 10, vm.doubles(12))
    cls, msg = vm.exception()
    assert cls == IAE and "axis" in msg
    assert vm.call("waitView", 0, 5) == 0.0
    assert vm.exception()[0] == IAE
    # allocPinned never throws: null on failure (the Java side falls back to a plain direct buffer)
    assert vm.call("allocPinned", 0, -1) is None
    assert vm.exception() is None


# ---- on the GPU: through the shim == through the C ABI ----------------------------------------------------------------------------
@pytest.fixture()
def gvm(vm, ctx):
    return vm, ctx


@pytest.mark.gpu
def test_create_and_destroy_a_context(vm):
    assert vm.call("deviceCount") >= 1
    h = vm.call("create", 0)
    assert vm.exception() is None and h != 0
    vm.call("destroy", h)
    assert vm.call("create", 99) == 0
    cls, msg = vm.exception()
    assert cls in (IAE, "java/lang/RuntimeException") and msg


@pytest.mark.gpu
def test_operators_through_the_shim_equal_the_c_abi(gvm, synth):
    vm, ctx = gvm
    h = ctx._h.value
    img = synth.sphere_phantom(40)
    dim = vm.longs([40, 40, 40])
    out = np.empty_like(img)
    vm.call("rotateAroundAxis", h, vm.float_buffer(img.reshape(-1).copy()), dim, 0, 35, vm.float_buffer(out.reshape(-1)))
    assert vm.exception() is None
    assert np.array_equal(out, ctx.rotate_around_axis(img, 0, 35))

    psf = synth.gaussian_psf(9)
    vm.call("convolve", h, vm.float_buffer(img.reshape(-1).copy()), dim, vm.float_buffer(psf.reshape(-1).copy()), vm.longs([9, 9, 9]), 0,
            vm.float_buffer(out.reshape(-1)))
    assert vm.exception() is None
    want = ctx.convolve(img, psf)
    assert np.array_equal(out, want)

    # the same convolution with the volume in two z slabs of different heights on either side
    flat = img.reshape(-1)
    plane = 1600
    ins = [flat[:plane * 25].copy(), flat[plane * 25:].copy()]
    outs = [np.empty(plane * 10, dtype=np.float32), np.empty(plane * 30, dtype=np.float32)]
    vm.call("convolveSlabs", h, vm.objects([vm.float_buffer(x) for x in ins]), vm.longs([25, 15]), dim, vm.float_buffer(psf.reshape(-1).copy()),
            vm.longs([9, 9, 9]), 0, vm.objects([vm.float_buffer(x) for x in outs]), vm.longs([10, 30]))
    assert vm.exception() is None
    assert np.array_equal(np.concatenate(outs).reshape(40, 40, 40), want)

    acq = np.empty((14, 40, 40), dtype=np.float32)
    vm.call("extractSlices", h, vm.float_buffer(want.reshape(-1).copy()), dim, 3, 10.0, 77, 2, vm.float_buffer(acq.reshape(-1)))
    assert vm.exception() is None
    assert np.array_equal(acq, ctx.extract_slices(want, 3, 10.0, 77, stream=2))


@pytest.mark.gpu
def test_views_through_the_shim_equal_the_c_abi(gvm, synth, mvs):
    vm, ctx = gvm
    h = ctx._h.value
    n, k = 48, 9
    gt = synth.sphere_phantom(n) + np.float32(0.5)
    psf = synth.gaussian_psf(k)
    dim, kdim = vm.longs([n, n, n]), vm.longs([k, k, k])
    nzo = (n - 1) // 3 + 1
    p = ctx.view_params(axis=0, degrees=45, delta=0.01, min_value=0.0, target_average=1.0, inc=3, snr=12.0, seed=5, stream=1)
    want = ctx.simulate_view(gt, psf, p, want=("con", "acq"))

    # synchronous, with one optional intermediate (con) and two nulls
    con = np.empty(n ** 3, dtype=np.float32)
    acq = np.empty(nzo * n * n, dtype=np.float32)
    corr = vm.call("simulateView", h, vm.float_buffer(gt.reshape(-1).copy()), dim, vm.float_buffer(psf.reshape(-1).copy()), kdim, 0, 45, 0.01, 0.0, 1.0,
                   3, 12.0, 5, 1, None, None, vm.float_buffer(con), vm.float_buffer(acq))
    assert vm.exception() is None
    assert corr == want["corr"]
    assert np.array_equal(con.reshape(n, n, n), want["con"]) and np.array_equal(acq.reshape(nzo, n, n), want["acq"])

    # asynchronous: ticket, then waitView
    acq2 = ctx.pinned_empty(nzo * n * n)
    gtp = ctx.pinned_empty(n ** 3)
    gtp[:] = gt.reshape(-1)
    ticket = vm.call("simulateViewAsync", h, vm.float_buffer(gtp), 1, dim, vm.float_buffer(psf.reshape(-1).copy()), kdim, 0, 45, 0.01, 0.0, 1.0, 3, 12.0,
                     5, 1, vm.float_buffer(acq2))
    assert vm.exception() is None and ticket >= 0
    assert vm.call("waitView", h, ticket) == want["corr"]
    assert vm.exception() is None
    assert np.array_equal(np.asarray(acq2).reshape(nzo, n, n), want["acq"])

    # the whole view loop in one call: view v takes stream v, axis 0 -- the Java facade's convention
    degs, seeds = [0, 90, 135], [11, 12, 13]
    psfs = [psf, synth.gaussian_psf(k, sigma=(1.2, 1.5, 2.5)), psf]
    bufs = [np.empty(nzo * n * n, dtype=np.float32) for _ in degs]
    vm.call("simulateViewsBatch", h, vm.float_buffer(gt.reshape(-1).copy()), dim, vm.objects([vm.float_buffer(q.reshape(-1).copy()) for q in psfs]), kdim,
            vm.ints(degs), 0.01, 0.0, 1.0, 3, 12.0, vm.longs(seeds), vm.objects([vm.float_buffer(b) for b in bufs]))
    assert vm.exception() is None
    for v, (d, s) in enumerate(zip(degs, seeds)):
        pv = ctx.view_params(axis=0, degrees=d, delta=0.01, min_value=0.0, target_average=1.0, inc=3, snr=12.0, seed=s, stream=v)
        assert np.array_equal(bufs[v].reshape(nzo, n, n), ctx.simulate_view(gt, psfs[v], pv)["acq"]), v


@pytest.mark.gpu
def test_phantom_and_pinned_blocks_through_the_shim(gvm, mvs):
    vm, ctx = gvm
    h = ctx._h.value
    # allocPinned hands back a direct ByteBuffer of `bytes` bytes over a page-locked block; freePinned takes it back
    block = vm.call("allocPinned", h, 1 << 20)
    assert block and vm.lib.fake_buffer_capacity(block) == 1 << 20 and vm.lib.fake_buffer_address(block)
    vm.call("freePinned", block)
    assert vm.exception() is None

    # drawSpheres: the java.util.Random state goes in and comes back through long[1]
    n = 160
    img = np.zeros(n ** 3, dtype=np.float32)
    rnd = mvs.JavaRandom(464232194)
    state_in = rnd._s
    st = vm.longs([state_in])
    count = vm.call("drawSpheres", h, vm.float_buffer(img), vm.longs([n, n, n]), 0.0, 1.0, 1, 0, st)
    assert vm.exception() is None and count > 0
    want = np.zeros((n, n, n), dtype=np.float32)
    rnd2 = mvs.JavaRandom(464232194)
    assert ctx.draw_spheres(want, 0.0, 1.0, 1, False, rnd2) == count
    assert np.array_equal(img.reshape(n, n, n), want)
    state_out = rnd2._s
    assert vm.read_longs(st, 1)[0] == state_out and state_out != state_in

    # normalizeWeights over a FloatBuffer[]
    rs = np.random.default_rng(3)
    w = [rs.random(4096, dtype=np.float32) for _ in range(3)]
    mine = [x.copy() for x in w]
    vm.call("normalizeWeights", h, vm.objects([vm.float_buffer(x) for x in mine]), 4096, 2.0)
    assert vm.exception() is None
    ref = [x.copy() for x in w]
    ctx.normalize_weights(ref, 2.0)
    for a, b in zip(mine, ref):
        assert np.array_equal(a, b)
