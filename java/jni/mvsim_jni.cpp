// JNI shim: net.preibisch.simulation.gpu.MvsimNative  ->  C ABI of libmvsim.so (include/mvsim.h).
//
// SOURCE ONLY in this repository (the build image has no JDK / jni.h): this file has never been loaded by a JVM.  What runs here:
// a syntax and type check against the hand-written jni.h subset of tests/jni_stub (tests/test_host_logic.py), and the functions
// below EXECUTED against a fake JNIEnv (tests/jni_fake, tests/test_jni_shim.py): the capacity checks, the exception mapping and,
// on the GPU, results bit-identical to the same calls through the C ABI.  Neither says anything about a real JVM.
// Build on a host with a JDK:
//   g++ -shared -fPIC -std=c++17 -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -I../../include
//       mvsim_jni.cpp -L../../multiview-simulation_amd -lmvsim -Wl,-rpath,'$ORIGIN' -o libmvsim_jni.so
//
// Every buffer is a direct java.nio.FloatBuffer: GetDirectBufferAddress gives the host pointer and
// GetDirectBufferCapacity its size in floats, which is checked against the dimensions BEFORE the C ABI sees the
// pointer (wrong dims from Java must become an IllegalArgumentException, not an out-of-bounds hipMemcpy).  Nothing is
// retained after a synchronous call returns; the asynchronous view keeps using its buffers until waitView, which the
// Java side guarantees by holding the staging blocks.  Status codes map to IllegalArgumentException (MVSIM_EINVAL),
// OutOfMemoryError (MVSIM_ENOMEM), RuntimeException (rest).  No JNI call is made with an exception pending.
#include <jni.h>

#include <cstdint>
#include <vector>

#include "mvsim.h"

namespace {

void throw_new(JNIEnv* env, const char* cls, const char* msg)
{
    if (env->ExceptionCheck()) return;                      // keep the first exception
    jclass c = env->FindClass(cls);
    if (c) env->ThrowNew(c, msg);
}

void throw_for(JNIEnv* env, int status)
{
    if (status == MVSIM_OK) return;
    const char* cls = status == MVSIM_EINVAL   ? "java/lang/IllegalArgumentException"
                      : status == MVSIM_ENOMEM ? "java/lang/OutOfMemoryError"
                                               : "java/lang/RuntimeException";
    throw_new(env, cls, mvsim_last_error());
}

// dims {nx, ny, nz}; ok == false after a pending ArrayIndexOutOfBoundsException or non-positive extents
struct Dim {
    int64_t d[3] = {1, 1, 1};
    bool ok = false;
    Dim(JNIEnv* env, jlongArray a)
    {
        if (!a || env->GetArrayLength(a) < 3) { throw_new(env, "java/lang/IllegalArgumentException", "dims: long[3] expected"); return; }
        jlong tmp[3] = {1, 1, 1};
        env->GetLongArrayRegion(a, 0, 3, tmp);
        if (env->ExceptionCheck()) return;
        if (tmp[0] < 1 || tmp[1] < 1 || tmp[2] < 1) { throw_new(env, "java/lang/IllegalArgumentException", "dims must be >= 1"); return; }
        d[0] = tmp[0]; d[1] = tmp[1]; d[2] = tmp[2];
        ok = true;
    }
    int64_t n() const { return d[0] * d[1] * d[2]; }
};

// host pointer of a direct FloatBuffer holding at least `need` floats; nullptr (+ exception) otherwise.
// optional == true: a null buffer is allowed and gives nullptr without an exception.
float* fptr(JNIEnv* env, jobject buf, int64_t need, const char* what, bool optional = false)
{
    if (env->ExceptionCheck()) return nullptr;              // an earlier argument already failed: no JNI call with that pending
    if (!buf) {
        if (!optional) throw_new(env, "java/lang/IllegalArgumentException", what);
        return nullptr;
    }
    void* p = env->GetDirectBufferAddress(buf);
    const jlong cap = env->GetDirectBufferCapacity(buf);    // in elements of the buffer's type
    if (!p || cap < 0) { throw_new(env, "java/lang/IllegalArgumentException", "a direct FloatBuffer is required"); return nullptr; }
    if (cap < need) { throw_new(env, "java/lang/IllegalArgumentException", what); return nullptr; }
    return static_cast<float*>(p);
}

mvsim_ctx* ctx_of(jlong h) { return reinterpret_cast<mvsim_ctx*>(static_cast<intptr_t>(h)); }
mvsim_group* group_of(jlong h) { return reinterpret_cast<mvsim_group*>(static_cast<intptr_t>(h)); }

void fill_params(mvsim_view_params* p, jint axis, jint degrees, jdouble delta, jfloat min_value, jfloat target, jint inc, jfloat snr,
                 jlong seed, jint stream)
{
    mvsim_view_params_default(p);
    p->axis = axis; p->degrees = degrees; p->delta = delta; p->min_value = min_value; p->target_average = target;
    p->inc = inc; p->snr = snr; p->seed = static_cast<uint64_t>(seed); p->stream = static_cast<uint32_t>(stream);
}

}  // namespace

extern "C" {

#define JNI_FN(name) Java_net_preibisch_simulation_gpu_MvsimNative_##name

JNIEXPORT jlong JNICALL JNI_FN(create)(JNIEnv* env, jclass, jint device)
{
    mvsim_ctx* c = nullptr;
    throw_for(env, mvsim_create(device, &c));
    return static_cast<jlong>(reinterpret_cast<intptr_t>(c));
}

JNIEXPORT void JNICALL JNI_FN(destroy)(JNIEnv*, jclass, jlong h) { mvsim_destroy(ctx_of(h)); }

JNIEXPORT jint JNICALL JNI_FN(deviceCount)(JNIEnv*, jclass)
{
    int n = 0;
    mvsim_device_count(&n);
    return n;
}

JNIEXPORT void JNICALL JNI_FN(rotateAroundAxis)(JNIEnv* env, jclass, jlong h, jobject in, jlongArray dim, jint axis,
                                                jint degrees, jobject out)
{
    Dim d(env, dim);
    if (!d.ok) return;
    float* pi = fptr(env, in, d.n(), "rotateAroundAxis: input buffer smaller than the dimensions");
    float* po = fptr(env, out, d.n(), "rotateAroundAxis: output buffer smaller than the dimensions");
    if (!pi || !po) return;
    throw_for(env, mvsim_rotate_around_axis(ctx_of(h), pi, d.d, axis, degrees, po));
}

JNIEXPORT void JNICALL JNI_FN(attenuate3d)(JNIEnv* env, jclass, jlong h, jobject in, jlongArray dim, jdouble delta,
                                           jobject out)
{
    Dim d(env, dim);
    if (!d.ok) return;
    float* pi = fptr(env, in, d.n(), "attenuate3d: input buffer smaller than the dimensions");
    float* po = fptr(env, out, d.n(), "attenuate3d: output buffer smaller than the dimensions");
    if (!pi || !po) return;
    throw_for(env, mvsim_attenuate3d(ctx_of(h), pi, d.d, delta, po));
}

JNIEXPORT void JNICALL JNI_FN(normImage)(JNIEnv* env, jclass, jlong h, jobject img, jlong n)
{
    float* p = fptr(env, img, n, "normImage: buffer smaller than n");
    if (!p) return;
    throw_for(env, mvsim_norm_image(ctx_of(h), p, n));
}

JNIEXPORT void JNICALL JNI_FN(convolve)(JNIEnv* env, jclass, jlong h, jobject img, jlongArray dim, jobject psf,
                                        jlongArray kdim, jint method, jobject out)
{
    Dim d(env, dim);
    if (!d.ok) return;
    Dim k(env, kdim);
    if (!k.ok) return;
    float* pi = fptr(env, img, d.n(), "convolve: image buffer smaller than the dimensions");
    float* pp = fptr(env, psf, k.n(), "convolve: PSF buffer smaller than its dimensions");
    float* po = fptr(env, out, d.n(), "convolve: output buffer smaller than the dimensions");
    if (!pi || !pp || !po) return;
    throw_for(env, mvsim_convolve(ctx_of(h), pi, d.d, pp, k.d, method, po));
}

JNIEXPORT jdouble JNICALL JNI_FN(adjustImage)(JNIEnv* env, jclass, jlong h, jobject img, jlong n, jfloat min_value,
                                              jfloat target)
{
    double corr = 0.0;
    float* p = fptr(env, img, n, "adjustImage: buffer smaller than n");
    if (!p) return corr;
    throw_for(env, mvsim_adjust_image(ctx_of(h), p, n, min_value, target, &corr));
    return corr;
}

JNIEXPORT void JNICALL JNI_FN(extractSlices)(JNIEnv* env, jclass, jlong h, jobject in, jlongArray dim, jint inc,
                                             jfloat snr, jlong seed, jint stream, jobject out)
{
    Dim d(env, dim);
    if (!d.ok) return;
    if (inc < 1) { throw_new(env, "java/lang/IllegalArgumentException", "extractSlices: inc must be >= 1"); return; }
    float* pi = fptr(env, in, d.n(), "extractSlices: input buffer smaller than the dimensions");
    float* po = fptr(env, out, d.d[0] * d.d[1] * mvsim_extract_nz(d.d[2], inc), "extractSlices: output buffer smaller than (Nz-1)/inc+1 planes");
    if (!pi || !po) return;
    throw_for(env, mvsim_extract_slices(ctx_of(h), pi, d.d, inc, snr, static_cast<uint64_t>(seed), static_cast<uint32_t>(stream), po));
}

JNIEXPORT void JNICALL JNI_FN(poissonProcess)(JNIEnv* env, jclass, jlong h, jobject img, jlong n, jdouble snr,
                                              jlong seed, jint stream, jlong index_offset)
{
    float* p = fptr(env, img, n, "poissonProcess: buffer smaller than n");
    if (!p) return;
    throw_for(env, mvsim_poisson_process(ctx_of(h), p, n, snr, static_cast<uint64_t>(seed), static_cast<uint32_t>(stream),
                                         static_cast<uint64_t>(index_offset)));
}

JNIEXPORT void JNICALL JNI_FN(makeIsotropic)(JNIEnv* env, jclass, jlong h, jobject in, jlongArray dim, jint inc,
                                             jobject out)
{
    Dim d(env, dim);
    if (!d.ok) return;
    if (inc < 1) { throw_new(env, "java/lang/IllegalArgumentException", "makeIsotropic: inc must be >= 1"); return; }
    float* pi = fptr(env, in, d.n(), "makeIsotropic: input buffer smaller than the dimensions");
    float* po = fptr(env, out, d.d[0] * d.d[1] * mvsim_isotropic_nz(d.d[2], inc), "makeIsotropic: output buffer smaller than (Nz-1)*inc+1 planes");
    if (!pi || !po) return;
    throw_for(env, mvsim_make_isotropic(ctx_of(h), pi, d.d, inc, po));
}

JNIEXPORT void JNICALL JNI_FN(computeWeightImage)(JNIEnv* env, jclass, jlong h, jlongArray dim, jobject out)
{
    Dim d(env, dim);
    if (!d.ok) return;
    float* po = fptr(env, out, d.n(), "computeWeightImage: output buffer smaller than the dimensions");
    if (!po) return;
    throw_for(env, mvsim_compute_weight_image(ctx_of(h), d.d, po));
}

JNIEXPORT void JNICALL JNI_FN(axisRotation)(JNIEnv* env, jclass, jlongArray dim, jint axis, jint degrees,
                                            jdoubleArray m12)
{
    Dim d(env, dim);
    if (!d.ok) return;
    if (!m12 || env->GetArrayLength(m12) < 12) { throw_new(env, "java/lang/IllegalArgumentException", "axisRotation: double[12] expected"); return; }
    double m[12];
    const int rc = mvsim_axis_rotation(d.d, axis, degrees, m);
    if (rc != MVSIM_OK) { throw_for(env, rc); return; }
    env->SetDoubleArrayRegion(m12, 0, 12, m);
}

// ---- per-stage operators with z-slab lists (mvsim_*_zslabs) ----------------------------------------------------------
namespace {
// the host pointers and plane counts of a FloatBuffer[] / long[] pair; every buffer is checked against plane * nz[i] floats
// and the planes against `planes` before the C ABI sees anything.  ok == false: an exception is pending.
struct SlabArgs {
    std::vector<float*>  ptr;
    std::vector<int64_t> nz;
    bool ok = false;
    SlabArgs(JNIEnv* env, jobjectArray bufs, jlongArray counts, int64_t plane, int64_t planes, const char* what)
    {
        if (!bufs || !counts) { throw_new(env, "java/lang/IllegalArgumentException", what); return; }
        const jsize n = env->GetArrayLength(bufs);
        if (n < 1 || env->GetArrayLength(counts) != n) { throw_new(env, "java/lang/IllegalArgumentException", what); return; }
        std::vector<jlong> c(static_cast<size_t>(n));
        env->GetLongArrayRegion(counts, 0, n, c.data());
        if (env->ExceptionCheck()) return;
        int64_t total = 0;
        for (jsize i = 0; i < n; ++i) {
            if (c[i] < 1) { throw_new(env, "java/lang/IllegalArgumentException", what); return; }
            jobject b = env->GetObjectArrayElement(bufs, i);
            if (env->ExceptionCheck()) return;
            float* p = fptr(env, b, plane * c[i], what);
            if (!p) return;
            ptr.push_back(p);
            nz.push_back(c[i]);
            total += c[i];
        }
        if (total != planes) { throw_new(env, "java/lang/IllegalArgumentException", what); return; }
        ok = true;
    }
};
}  // namespace

JNIEXPORT void JNICALL JNI_FN(rotateAroundAxisSlabs)(JNIEnv* env, jclass, jlong h, jobjectArray in, jlongArray in_nz, jlongArray dim,
                                                     jint axis, jint degrees, jobjectArray out, jlongArray out_nz)
{
    Dim d(env, dim);
    if (!d.ok) return;
    SlabArgs si(env, in, in_nz, d.d[0] * d.d[1], d.d[2], "rotateAroundAxis: input slabs do not match the dimensions");
    if (!si.ok) return;
    SlabArgs so(env, out, out_nz, d.d[0] * d.d[1], d.d[2], "rotateAroundAxis: output slabs do not match the dimensions");
    if (!so.ok) return;
    throw_for(env, mvsim_rotate_around_axis_zslabs(ctx_of(h), si.ptr.data(), si.nz.data(), (int)si.ptr.size(), d.d, axis, degrees,
                                                   so.ptr.data(), so.nz.data(), (int)so.ptr.size()));
}

JNIEXPORT void JNICALL JNI_FN(attenuate3dSlabs)(JNIEnv* env, jclass, jlong h, jobjectArray in, jlongArray in_nz, jlongArray dim,
                                                jdouble delta, jobjectArray out, jlongArray out_nz)
{
    Dim d(env, dim);
    if (!d.ok) return;
    SlabArgs si(env, in, in_nz, d.d[0] * d.d[1], d.d[2], "attenuate3d: input slabs do not match the dimensions");
    if (!si.ok) return;
    SlabArgs so(env, out, out_nz, d.d[0] * d.d[1], d.d[2], "attenuate3d: output slabs do not match the dimensions");
    if (!so.ok) return;
    throw_for(env, mvsim_attenuate3d_zslabs(ctx_of(h), si.ptr.data(), si.nz.data(), (int)si.ptr.size(), d.d, delta, so.ptr.data(),
                                            so.nz.data(), (int)so.ptr.size()));
}

JNIEXPORT void JNICALL JNI_FN(convolveSlabs)(JNIEnv* env, jclass, jlong h, jobjectArray in, jlongArray in_nz, jlongArray dim, jobject psf,
                                             jlongArray kdim, jint method, jobjectArray out, jlongArray out_nz)
{
    Dim d(env, dim);
    if (!d.ok) return;
    Dim k(env, kdim);
    if (!k.ok) return;
    float* pp = fptr(env, psf, k.n(), "convolve: PSF buffer smaller than its dimensions");
    if (!pp) return;
    SlabArgs si(env, in, in_nz, d.d[0] * d.d[1], d.d[2], "convolve: input slabs do not match the dimensions");
    if (!si.ok) return;
    SlabArgs so(env, out, out_nz, d.d[0] * d.d[1], d.d[2], "convolve: output slabs do not match the dimensions");
    if (!so.ok) return;
    throw_for(env, mvsim_convolve_zslabs(ctx_of(h), si.ptr.data(), si.nz.data(), (int)si.ptr.size(), d.d, pp, k.d, method, so.ptr.data(),
                                         so.nz.data(), (int)so.ptr.size()));
}

JNIEXPORT void JNICALL JNI_FN(extractSlicesSlabs)(JNIEnv* env, jclass, jlong h, jobjectArray in, jlongArray in_nz, jlongArray dim, jint inc,
                                                  jfloat snr, jlong seed, jint stream, jobjectArray out, jlongArray out_nz)
{
    Dim d(env, dim);
    if (!d.ok) return;
    if (inc < 1) { throw_new(env, "java/lang/IllegalArgumentException", "extractSlices: inc must be >= 1"); return; }
    SlabArgs si(env, in, in_nz, d.d[0] * d.d[1], d.d[2], "extractSlices: input slabs do not match the dimensions");
    if (!si.ok) return;
    SlabArgs so(env, out, out_nz, d.d[0] * d.d[1], mvsim_extract_nz(d.d[2], inc), "extractSlices: output slabs do not hold (Nz-1)/inc+1 planes");
    if (!so.ok) return;
    throw_for(env, mvsim_extract_slices_zslabs(ctx_of(h), si.ptr.data(), si.nz.data(), (int)si.ptr.size(), d.d, inc, snr,
                                               static_cast<uint64_t>(seed), static_cast<uint32_t>(stream), so.ptr.data(), so.nz.data(),
                                               (int)so.ptr.size()));
}

JNIEXPORT jobject JNICALL JNI_FN(allocPinned)(JNIEnv* env, jclass, jlong h, jlong bytes)
{
    void* p = nullptr;
    if (bytes <= 0 || mvsim_host_alloc(ctx_of(h), static_cast<size_t>(bytes), &p) != MVSIM_OK || !p) return nullptr;
    jobject b = env->NewDirectByteBuffer(p, bytes);
    if (!b) (void)mvsim_host_free(nullptr, p);              // the JVM could not wrap it: do not leak the block
    return b;
}

JNIEXPORT void JNICALL JNI_FN(freePinned)(JNIEnv* env, jclass, jobject block)
{
    if (block) (void)mvsim_host_free(nullptr, env->GetDirectBufferAddress(block));
}

// Bulk copies between a Java float[] and a (page-locked) direct block on the library's host threads (mvsim_host_copy).  The array is
// held with GetPrimitiveArrayCritical for the duration of the copy -- the worker threads touch plain memory, never the JNIEnv, and no
// JNI call is made inside the critical region (JNI specification, "GetPrimitiveArrayCritical").  to_array == JNI_TRUE: block -> array
// (Buffers.toImg's copy into the NEW ArrayImg every operator of the reference returns, SimulateMultiViewDataset.java:109,198: the
// threads also take the array's first-touch page faults in parallel); JNI_FALSE: array -> block (Buffers.toBlock / toSlabs).
JNIEXPORT void JNICALL JNI_FN(copyFloats)(JNIEnv* env, jclass, jlong h, jobject block, jlong block_off, jfloatArray array, jint array_off,
                                          jint count, jboolean to_array)
{
    if (count == 0) return;
    if (!array || count < 0 || array_off < 0 || block_off < 0 || (jlong)array_off + count > env->GetArrayLength(array)) {
        throw_new(env, "java/lang/ArrayIndexOutOfBoundsException", "copyFloats: range outside the array");
        return;
    }
    float* b = fptr(env, block, block_off + count, "copyFloats: range outside the block");
    if (!b) return;
    void* a = env->GetPrimitiveArrayCritical(array, nullptr);
    if (!a) { throw_new(env, "java/lang/OutOfMemoryError", "copyFloats: GetPrimitiveArrayCritical"); return; }
    float* arr = static_cast<float*>(a) + array_off;
    const int rc = to_array ? mvsim_host_copy(ctx_of(h), arr, b + block_off, static_cast<size_t>(count) * sizeof(float))
                            : mvsim_host_copy(ctx_of(h), b + block_off, arr, static_cast<size_t>(count) * sizeof(float));
    env->ReleasePrimitiveArrayCritical(array, a, to_array ? 0 : JNI_ABORT);   // array -> block: nothing to write back
    throw_for(env, rc);
}

JNIEXPORT jlong JNICALL JNI_FN(drawSpheres)(JNIEnv* env, jclass, jlong h, jobject img, jlongArray dim, jdouble min_value,
                                            jdouble max_value, jint scale, jboolean half_pixel_offset, jlongArray rnd_state)
{
    Dim d(env, dim);
    if (!d.ok) return 0;
    float* p = fptr(env, img, d.n(), "drawSpheres: buffer smaller than the dimensions");
    if (!p) return 0;
    if (!rnd_state || env->GetArrayLength(rnd_state) < 1) { throw_new(env, "java/lang/IllegalArgumentException", "drawSpheres: long[1] state expected"); return 0; }
    jlong st = 0;
    env->GetLongArrayRegion(rnd_state, 0, 1, &st);
    if (env->ExceptionCheck()) return 0;
    uint64_t state = static_cast<uint64_t>(st);
    int64_t n = 0;
    const int rc = mvsim_draw_spheres(ctx_of(h), p, d.d, min_value, max_value, scale, half_pixel_offset ? 1 : 0, &state, &n);
    if (rc != MVSIM_OK) { throw_for(env, rc); return 0; }
    st = static_cast<jlong>(state);
    env->SetLongArrayRegion(rnd_state, 0, 1, &st);
    return static_cast<jlong>(n);
}

JNIEXPORT void JNICALL JNI_FN(splatSpheres)(JNIEnv* env, jclass, jlong h, jobject img, jlongArray dim, jintArray geometry,
                                            jfloatArray values)
{
    Dim d(env, dim);
    if (!d.ok) return;
    float* p = fptr(env, img, d.n(), "splatSpheres: buffer smaller than the dimensions");
    if (!p) return;
    if (!geometry || !values) { throw_new(env, "java/lang/IllegalArgumentException", "splatSpheres: null sphere list"); return; }
    const jsize n = env->GetArrayLength(values);
    if (env->GetArrayLength(geometry) != 4 * n) { throw_new(env, "java/lang/IllegalArgumentException", "splatSpheres: 4 ints per sphere expected"); return; }
    std::vector<jint> g(static_cast<size_t>(4 * n));
    std::vector<jfloat> v(static_cast<size_t>(n));
    if (n > 0) {
        env->GetIntArrayRegion(geometry, 0, 4 * n, g.data());
        if (env->ExceptionCheck()) return;
        env->GetFloatArrayRegion(values, 0, n, v.data());
        if (env->ExceptionCheck()) return;
    }
    std::vector<mvsim_sphere> s(static_cast<size_t>(n));
    for (jsize i = 0; i < n; ++i) s[i] = mvsim_sphere{g[4 * i], g[4 * i + 1], g[4 * i + 2], g[4 * i + 3], v[i]};
    throw_for(env, mvsim_splat_spheres(ctx_of(h), p, d.d, s.data(), static_cast<int64_t>(n)));
}

JNIEXPORT void JNICALL JNI_FN(downSample2x)(JNIEnv* env, jclass, jlong h, jobject in, jlongArray dim, jobject out)
{
    Dim d(env, dim);
    if (!d.ok) return;
    float* pi = fptr(env, in, d.n(), "downSample2x: input buffer smaller than the dimensions");
    const int64_t no = (d.d[0] / 2 - 1) * (d.d[1] / 2 - 1) * (d.d[2] / 2 - 1);
    float* po = fptr(env, out, no > 0 ? no : 0, "downSample2x: output buffer smaller than (N/2-1)^3");
    if (!pi || !po) return;
    throw_for(env, mvsim_downsample2x(ctx_of(h), pi, d.d, po));
}

JNIEXPORT void JNICALL JNI_FN(normalizeWeights)(JNIEnv* env, jclass, jlong h, jobjectArray weights, jlong n, jfloat osem)
{
    if (!weights) { throw_new(env, "java/lang/IllegalArgumentException", "normalizeWeights: null list"); return; }
    const jsize nv = env->GetArrayLength(weights);
    if (nv < 1 || nv > MVSIM_MAX_VIEWS) { throw_new(env, "java/lang/IllegalArgumentException", "normalizeWeights: 1..32 views"); return; }
    float* ptr[MVSIM_MAX_VIEWS];
    for (jsize v = 0; v < nv; ++v) {
        jobject b = env->GetObjectArrayElement(weights, v);
        if (env->ExceptionCheck()) return;
        ptr[v] = fptr(env, b, n, "normalizeWeights: view buffer smaller than n");
        if (!ptr[v]) return;
    }
    throw_for(env, mvsim_normalize_weights(ctx_of(h), ptr, (int)nv, n, osem));
}

JNIEXPORT jdouble JNICALL JNI_FN(simulateView)(JNIEnv* env, jclass, jlong h, jobject gt, jlongArray dim, jobject psf,
                                               jlongArray kdim, jint axis, jint degrees, jdouble delta,
                                               jfloat min_value, jfloat target, jint inc, jfloat snr, jlong seed,
                                               jint stream, jobject rot, jobject att, jobject con, jobject acq)
{
    double corr = 0.0;
    Dim d(env, dim);
    if (!d.ok) return corr;
    Dim k(env, kdim);
    if (!k.ok) return corr;
    if (inc < 1) { throw_new(env, "java/lang/IllegalArgumentException", "simulateView: inc must be >= 1"); return corr; }
    float* pg = fptr(env, gt, d.n(), "simulateView: ground-truth buffer smaller than the dimensions");
    float* pp = fptr(env, psf, k.n(), "simulateView: PSF buffer smaller than its dimensions");
    float* pa = fptr(env, acq, d.d[0] * d.d[1] * mvsim_extract_nz(d.d[2], inc), "simulateView: acquisition buffer too small");
    if (!pg || !pp || !pa) return corr;
    mvsim_view_outputs o = {fptr(env, rot, d.n(), "simulateView: rot buffer too small", true), fptr(env, att, d.n(), "simulateView: att buffer too small", true),
                            fptr(env, con, d.n(), "simulateView: con buffer too small", true), pa};
    if (env->ExceptionCheck()) return corr;
    mvsim_view_params p;
    fill_params(&p, axis, degrees, delta, min_value, target, inc, snr, seed, stream);
    throw_for(env, mvsim_simulate_view(ctx_of(h), pg, d.d, pp, k.d, &p, &o, &corr));
    return corr;
}

JNIEXPORT jlong JNICALL JNI_FN(simulateViewAsync)(JNIEnv* env, jclass, jlong h, jobject gt, jlong gt_generation, jlongArray dim,
                                                  jobject psf, jlongArray kdim, jint axis, jint degrees, jdouble delta, jfloat min_value,
                                                  jfloat target, jint inc, jfloat snr, jlong seed, jint stream, jobject acq)
{
    Dim d(env, dim);
    if (!d.ok) return -1;
    Dim k(env, kdim);
    if (!k.ok) return -1;
    if (inc < 1) { throw_new(env, "java/lang/IllegalArgumentException", "simulateViewAsync: inc must be >= 1"); return -1; }
    float* pg = fptr(env, gt, d.n(), "simulateViewAsync: ground-truth buffer smaller than the dimensions");
    float* pp = fptr(env, psf, k.n(), "simulateViewAsync: PSF buffer smaller than its dimensions");
    float* pa = fptr(env, acq, d.d[0] * d.d[1] * mvsim_extract_nz(d.d[2], inc), "simulateViewAsync: acquisition buffer too small");
    if (!pg || !pp || !pa) return -1;
    mvsim_view_params p;
    fill_params(&p, axis, degrees, delta, min_value, target, inc, snr, seed, stream);
    mvsim_view_outputs o = {nullptr, nullptr, nullptr, pa};
    int64_t ticket = -1;
    throw_for(env, mvsim_simulate_view_async(ctx_of(h), pg, static_cast<uint64_t>(gt_generation), d.d, pp, k.d, &p, &o, &ticket));
    return static_cast<jlong>(ticket);
}

JNIEXPORT jdouble JNICALL JNI_FN(waitView)(JNIEnv* env, jclass, jlong h, jlong ticket)
{
    double corr = 0.0;
    throw_for(env, mvsim_wait(ctx_of(h), ticket, &corr));
    return corr;
}

JNIEXPORT void JNICALL JNI_FN(waitViewQuiet)(JNIEnv*, jclass, jlong h, jlong ticket) { (void)mvsim_wait(ctx_of(h), ticket, nullptr); }

// mvsim_simulate_views: V views of one ground truth in one call (host buffers; the library stacks them when they cannot fill the chip
// one at a time and brings the acquisitions back as 16-bit counts)
JNIEXPORT void JNICALL JNI_FN(simulateViewsBatch)(JNIEnv* env, jclass, jlong h, jobject gt, jlongArray dim, jobjectArray psfs, jlongArray kdim,
                                                  jintArray degrees, jdouble delta, jfloat min_value, jfloat target, jint inc, jfloat snr,
                                                  jlongArray seeds, jobjectArray acqs)
{
    Dim d(env, dim);
    if (!d.ok) return;
    Dim k(env, kdim);
    if (!k.ok) return;
    if (inc < 1) { throw_new(env, "java/lang/IllegalArgumentException", "simulateViewsBatch: inc must be >= 1"); return; }
    if (!psfs || !degrees || !seeds || !acqs) { throw_new(env, "java/lang/IllegalArgumentException", "simulateViewsBatch: null argument"); return; }
    const jsize n = env->GetArrayLength(degrees);
    if (env->GetArrayLength(psfs) != n || env->GetArrayLength(seeds) != n || env->GetArrayLength(acqs) != n) {
        throw_new(env, "java/lang/IllegalArgumentException", "simulateViewsBatch: one PSF, seed and acquisition buffer per view");
        return;
    }
    float* pg = fptr(env, gt, d.n(), "simulateViewsBatch: ground-truth buffer smaller than the dimensions");
    if (!pg) return;
    const int64_t acq_floats = d.d[0] * d.d[1] * mvsim_extract_nz(d.d[2], inc);
    std::vector<jint> deg(static_cast<size_t>(n));
    std::vector<jlong> sd(static_cast<size_t>(n));
    if (n > 0) {
        env->GetIntArrayRegion(degrees, 0, n, deg.data());
        if (env->ExceptionCheck()) return;
        env->GetLongArrayRegion(seeds, 0, n, sd.data());
        if (env->ExceptionCheck()) return;
    }
    std::vector<float*> pp(static_cast<size_t>(n)), pa(static_cast<size_t>(n));
    std::vector<mvsim_view_params> par(static_cast<size_t>(n));
    for (jsize v = 0; v < n; ++v) {
        jobject b = env->GetObjectArrayElement(psfs, v);
        if (env->ExceptionCheck()) return;
        pp[v] = fptr(env, b, k.n(), "simulateViewsBatch: PSF buffer smaller than its dimensions");
        jobject a = env->GetObjectArrayElement(acqs, v);
        if (env->ExceptionCheck()) return;
        pa[v] = fptr(env, a, acq_floats, "simulateViewsBatch: acquisition buffer smaller than nx * ny * ((nz - 1) / inc + 1) floats");
        if (!pp[v] || !pa[v]) return;
        fill_params(&par[v], 0, deg[v], delta, min_value, target, inc, snr, sd[v], v);
    }
    throw_for(env, mvsim_simulate_views(ctx_of(h), pg, d.d, pp.data(), k.d, par.data(), pa.data(), (int)n));
}

// ---- mvsim_group_* -------------------------------------------------------------------------------------------------
JNIEXPORT jlong JNICALL JNI_FN(groupCreate)(JNIEnv* env, jclass, jint ndev)
{
    mvsim_group* g = nullptr;
    throw_for(env, mvsim_group_create(ndev, nullptr, &g));
    return static_cast<jlong>(reinterpret_cast<intptr_t>(g));
}

JNIEXPORT void JNICALL JNI_FN(groupDestroy)(JNIEnv*, jclass, jlong g) { mvsim_group_destroy(group_of(g)); }

JNIEXPORT void JNICALL JNI_FN(groupBroadcastVolume)(JNIEnv* env, jclass, jlong g, jobject gt, jlongArray dim)
{
    Dim d(env, dim);
    if (!d.ok) return;
    float* p = fptr(env, gt, d.n(), "groupBroadcastVolume: buffer smaller than the dimensions");
    if (!p) return;
    throw_for(env, mvsim_group_broadcast_volume(group_of(g), p, d.d));
}

JNIEXPORT void JNICALL JNI_FN(groupSimulateViews)(JNIEnv* env, jclass, jlong g, jobjectArray psfs, jlongArray kdim, jlongArray dim,
                                                  jintArray degrees, jdouble delta, jfloat min_value, jfloat target, jint inc, jfloat snr,
                                                  jlongArray seeds, jobjectArray acqs)
{
    Dim k(env, kdim);
    if (!k.ok) return;
    Dim d(env, dim);                                       // the dimensions of the broadcast ground truth (GpuGroup keeps them)
    if (!d.ok) return;
    if (inc < 1) { throw_new(env, "java/lang/IllegalArgumentException", "groupSimulateViews: inc must be >= 1"); return; }
    const int64_t acq_floats = d.d[0] * d.d[1] * mvsim_extract_nz(d.d[2], inc);
    if (!psfs || !degrees || !seeds || !acqs) { throw_new(env, "java/lang/IllegalArgumentException", "groupSimulateViews: null argument"); return; }
    const jsize n = env->GetArrayLength(degrees);
    if (env->GetArrayLength(psfs) != n || env->GetArrayLength(seeds) != n || env->GetArrayLength(acqs) != n) {
        throw_new(env, "java/lang/IllegalArgumentException", "groupSimulateViews: one PSF, seed and acquisition buffer per view");
        return;
    }
    std::vector<jint> deg(static_cast<size_t>(n));
    std::vector<jlong> sd(static_cast<size_t>(n));
    if (n > 0) {
        env->GetIntArrayRegion(degrees, 0, n, deg.data());
        if (env->ExceptionCheck()) return;
        env->GetLongArrayRegion(seeds, 0, n, sd.data());
        if (env->ExceptionCheck()) return;
    }
    std::vector<float*> pp(static_cast<size_t>(n)), pa(static_cast<size_t>(n));
    std::vector<mvsim_view_params> par(static_cast<size_t>(n));
    for (jsize v = 0; v < n; ++v) {
        jobject b = env->GetObjectArrayElement(psfs, v);
        if (env->ExceptionCheck()) return;
        pp[v] = fptr(env, b, k.n(), "groupSimulateViews: PSF buffer smaller than its dimensions");
        jobject a = env->GetObjectArrayElement(acqs, v);
        if (env->ExceptionCheck()) return;
        // the C ABI takes no capacity: an undersized direct buffer would take an out-of-bounds copy, so it is refused here
        pa[v] = fptr(env, a, acq_floats, "groupSimulateViews: acquisition buffer smaller than nx * ny * ((nz - 1) / inc + 1) floats");
        if (!pp[v] || !pa[v]) return;
        fill_params(&par[v], 0, deg[v], delta, min_value, target, inc, snr, sd[v], v);
    }
    throw_for(env, mvsim_group_simulate_views(group_of(g), pp.data(), k.d, par.data(), (int)n, pa.data()));
}

}  // extern "C"
