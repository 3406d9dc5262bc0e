/*
 * NOT the JDK's jni.h.  A hand-written subset of the JNI C++ surface -- the typedefs and the JNIEnv member functions
 * java/jni/mvsim_jni.cpp uses, with the signatures of the JNI specification (chapter 4, "JNI Functions") -- so that the
 * shim can be syntax- and type-checked (g++ -fsyntax-only) in an image without a JDK.  It declares, it defines nothing
 * (tests/jni_fake/fake_jni.cpp defines the members over a fake object table, for tests/test_jni_shim.py); passing either
 * check pins nothing about the behaviour of the shim on a real JVM.  The build image has no JDK; on a host that has one, the JDK's own headers are used.
 */
#ifndef MVSIM_TEST_JNI_STUB_H
#define MVSIM_TEST_JNI_STUB_H

#include <cstdint>

#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL
#define JNI_TRUE 1
#define JNI_FALSE 0
#define JNI_COMMIT 1
#define JNI_ABORT 2

typedef unsigned char jboolean;
typedef int32_t       jint;
typedef int64_t       jlong;
typedef float         jfloat;
typedef double        jdouble;
typedef jint          jsize;

class _jobject {};
class _jclass : public _jobject {};
class _jthrowable : public _jobject {};
class _jarray : public _jobject {};
class _jintArray : public _jarray {};
class _jlongArray : public _jarray {};
class _jfloatArray : public _jarray {};
class _jdoubleArray : public _jarray {};
class _jobjectArray : public _jarray {};
typedef _jobject*      jobject;
typedef _jclass*       jclass;
typedef _jthrowable*   jthrowable;
typedef _jarray*       jarray;
typedef _jintArray*    jintArray;
typedef _jlongArray*   jlongArray;
typedef _jfloatArray*  jfloatArray;
typedef _jdoubleArray* jdoubleArray;
typedef _jobjectArray* jobjectArray;

struct JNIEnv_ {
    jclass   FindClass(const char* name);
    jint     ThrowNew(jclass clazz, const char* msg);
    jboolean ExceptionCheck();
    jsize    GetArrayLength(jarray array);
    jobject  GetObjectArrayElement(jobjectArray array, jsize index);
    void     GetIntArrayRegion(jintArray array, jsize start, jsize len, jint* buf);
    void     GetLongArrayRegion(jlongArray array, jsize start, jsize len, jlong* buf);
    void     SetLongArrayRegion(jlongArray array, jsize start, jsize len, const jlong* buf);
    void     GetFloatArrayRegion(jfloatArray array, jsize start, jsize len, jfloat* buf);
    void     SetDoubleArrayRegion(jdoubleArray array, jsize start, jsize len, const jdouble* buf);
    void*    GetPrimitiveArrayCritical(jarray array, jboolean* isCopy);
    void     ReleasePrimitiveArrayCritical(jarray array, void* carray, jint mode);
    jobject  NewDirectByteBuffer(void* address, jlong capacity);
    void*    GetDirectBufferAddress(jobject buf);
    jlong    GetDirectBufferCapacity(jobject buf);
};
typedef JNIEnv_ JNIEnv;

#endif
