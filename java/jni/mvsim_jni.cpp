// JNI shim: net.preibisch.simulation.gpu.MvsimNative  ->  C ABI of libmvsim.so (include/mvsim.h).
//
// SOURCE ONLY in this repository (the build image has no JDK / jni.h).  Build on a host with a JDK:
//   g++ -shared -fPIC -std=c++17 -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -I../../include \
//       mvsim_jni.cpp -L../../multiview-simulation_amd -lmvsim -Wl,-rpath,'$ORIGIN' -o libmvsim_jni.so
//
// Every buffer is a direct java.nio.FloatBuffer: GetDirectBufferAddress gives the host pointer, nothing
// is retained after the call returns (the C ABI is synchronous for host buffers).  Status codes map to
// IllegalArgumentException (MVSIM_EINVAL), OutOfMemoryError (MVSIM_ENOMEM), RuntimeException (rest).
#include <jni.h>

#include <cstdint>

#include "mvsim.h"

namespace {

void throw_for(JNIEnv* env, int status)
{
    if (status == MVSIM_OK) return;
    const char* cls = status == MVSIM_EINVAL   ? "java/lang/IllegalArgumentException"
                      : status == MVSIM_ENOMEM ? "java/lang/OutOfMemoryError"
                                               : "java/lang/RuntimeException";
    env->ThrowNew(env->FindClass(cls), mvsim_last_error());
}

float* fptr(JNIEnv* env, jobject buf)
{
    return buf ? static_cast<float*>(env->GetDirectBufferAddress(buf)) : nullptr;
}

struct Dim {
    int64_t d[3];
    Dim(JNIEnv* env, jlongArray a)
    {
        jlong tmp[3] = {1, 1, 1};
        env->GetLongArrayRegion(a, 0, 3, tmp);
        d[0] = tmp[0]; d[1] = tmp[1]; d[2] = tmp[2];
    }
};

mvsim_ctx* ctx_of(jlong h) { return reinterpret_cast<mvsim_ctx*>(static_cast<intptr_t>(h)); }

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
    throw_for(env, mvsim_rotate_around_axis(ctx_of(h), fptr(env, in), d.d, axis, degrees, fptr(env, out)));
}

JNIEXPORT void JNICALL JNI_FN(attenuate3d)(JNIEnv* env, jclass, jlong h, jobject in, jlongArray dim, jdouble delta,
                                           jobject out)
{
    Dim d(env, dim);
    throw_for(env, mvsim_attenuate3d(ctx_of(h), fptr(env, in), d.d, delta, fptr(env, out)));
}

JNIEXPORT void JNICALL JNI_FN(normImage)(JNIEnv* env, jclass, jlong h, jobject img, jlong n)
{
    throw_for(env, mvsim_norm_image(ctx_of(h), fptr(env, img), n));
}

JNIEXPORT void JNICALL JNI_FN(convolve)(JNIEnv* env, jclass, jlong h, jobject img, jlongArray dim, jobject psf,
                                        jlongArray kdim, jint method, jobject out)
{
    Dim d(env, dim), k(env, kdim);
    throw_for(env, mvsim_convolve(ctx_of(h), fptr(env, img), d.d, fptr(env, psf), k.d, method, fptr(env, out)));
}

JNIEXPORT jdouble JNICALL JNI_FN(adjustImage)(JNIEnv* env, jclass, jlong h, jobject img, jlong n, jfloat min_value,
                                              jfloat target)
{
    double corr = 0.0;
    throw_for(env, mvsim_adjust_image(ctx_of(h), fptr(env, img), n, min_value, target, &corr));
    return corr;
}

JNIEXPORT void JNICALL JNI_FN(extractSlices)(JNIEnv* env, jclass, jlong h, jobject in, jlongArray dim, jint inc,
                                             jfloat snr, jlong seed, jint stream, jobject out)
{
    Dim d(env, dim);
    throw_for(env, mvsim_extract_slices(ctx_of(h), fptr(env, in), d.d, inc, snr, static_cast<uint64_t>(seed),
                                        static_cast<uint32_t>(stream), fptr(env, out)));
}

JNIEXPORT void JNICALL JNI_FN(poissonProcess)(JNIEnv* env, jclass, jlong h, jobject img, jlong n, jdouble snr,
                                              jlong seed, jint stream, jlong index_offset)
{
    throw_for(env, mvsim_poisson_process(ctx_of(h), fptr(env, img), n, snr, static_cast<uint64_t>(seed),
                                         static_cast<uint32_t>(stream), static_cast<uint64_t>(index_offset)));
}

JNIEXPORT void JNICALL JNI_FN(makeIsotropic)(JNIEnv* env, jclass, jlong h, jobject in, jlongArray dim, jint inc,
                                             jobject out)
{
    Dim d(env, dim);
    throw_for(env, mvsim_make_isotropic(ctx_of(h), fptr(env, in), d.d, inc, fptr(env, out)));
}

JNIEXPORT void JNICALL JNI_FN(computeWeightImage)(JNIEnv* env, jclass, jlong h, jlongArray dim, jobject out)
{
    Dim d(env, dim);
    throw_for(env, mvsim_compute_weight_image(ctx_of(h), d.d, fptr(env, out)));
}

JNIEXPORT void JNICALL JNI_FN(axisRotation)(JNIEnv* env, jclass, jlongArray dim, jint axis, jint degrees,
                                            jdoubleArray m12)
{
    Dim d(env, dim);
    double m[12];
    throw_for(env, mvsim_axis_rotation(d.d, axis, degrees, m));
    env->SetDoubleArrayRegion(m12, 0, 12, m);
}

JNIEXPORT jobject JNICALL JNI_FN(allocPinned)(JNIEnv* env, jclass, jlong h, jlong bytes)
{
    void* p = nullptr;
    if (mvsim_host_alloc(ctx_of(h), static_cast<size_t>(bytes), &p) != MVSIM_OK || !p) return nullptr;
    return env->NewDirectByteBuffer(p, bytes);
}

JNIEXPORT void JNICALL JNI_FN(freePinned)(JNIEnv* env, jclass, jobject block)
{
    if (block) (void)mvsim_host_free(nullptr, env->GetDirectBufferAddress(block));
}

JNIEXPORT jlong JNICALL JNI_FN(drawSpheres)(JNIEnv* env, jclass, jlong h, jobject img, jlongArray dim, jdouble min_value,
                                            jdouble max_value, jint scale, jboolean half_pixel_offset, jlongArray rnd_state)
{
    Dim d(env, dim);
    jlong st = 0;
    env->GetLongArrayRegion(rnd_state, 0, 1, &st);
    uint64_t state = static_cast<uint64_t>(st);
    int64_t n = 0;
    throw_for(env, mvsim_draw_spheres(ctx_of(h), fptr(env, img), d.d, min_value, max_value, scale, half_pixel_offset ? 1 : 0,
                                      &state, &n));
    st = static_cast<jlong>(state);
    env->SetLongArrayRegion(rnd_state, 0, 1, &st);
    return static_cast<jlong>(n);
}

JNIEXPORT void JNICALL JNI_FN(downSample2x)(JNIEnv* env, jclass, jlong h, jobject in, jlongArray dim, jobject out)
{
    Dim d(env, dim);
    throw_for(env, mvsim_downsample2x(ctx_of(h), fptr(env, in), d.d, fptr(env, out)));
}

JNIEXPORT void JNICALL JNI_FN(normalizeWeights)(JNIEnv* env, jclass, jlong h, jobjectArray weights, jlong n, jfloat osem)
{
    const jsize nv = env->GetArrayLength(weights);
    if (nv > MVSIM_MAX_VIEWS) { throw_for(env, MVSIM_EINVAL); return; }
    float* ptr[MVSIM_MAX_VIEWS];
    for (jsize v = 0; v < nv; ++v) ptr[v] = fptr(env, env->GetObjectArrayElement(weights, v));
    throw_for(env, mvsim_normalize_weights(ctx_of(h), ptr, (int)nv, n, osem));
}

JNIEXPORT jdouble JNICALL JNI_FN(simulateView)(JNIEnv* env, jclass, jlong h, jobject gt, jlongArray dim, jobject psf,
                                               jlongArray kdim, jint axis, jint degrees, jdouble delta,
                                               jfloat min_value, jfloat target, jint inc, jfloat snr, jlong seed,
                                               jint stream, jobject rot, jobject att, jobject con, jobject acq)
{
    Dim d(env, dim), k(env, kdim);
    mvsim_view_params p;
    mvsim_view_params_default(&p);
    p.axis = axis; p.degrees = degrees; p.delta = delta; p.min_value = min_value; p.target_average = target;
    p.inc = inc; p.snr = snr; p.seed = static_cast<uint64_t>(seed); p.stream = static_cast<uint32_t>(stream);
    mvsim_view_outputs o = {fptr(env, rot), fptr(env, att), fptr(env, con), fptr(env, acq)};
    double corr = 0.0;
    throw_for(env, mvsim_simulate_view(ctx_of(h), fptr(env, gt), d.d, fptr(env, psf), k.d, &p, &o, &corr));
    return corr;
}

}  // extern "C"
