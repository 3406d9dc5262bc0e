// A FAKE JNIEnv for tests: definitions of the JNIEnv member functions tests/jni_stub/jni.h declares, over an in-process
// object table, plus a small C driver API so that a test (ctypes) can make "Java" arrays and direct buffers, call the
// Java_net_preibisch_simulation_gpu_MvsimNative_* functions of java/jni/mvsim_jni.cpp, and read back what they threw.
//
// What this buys: the NATIVE half of the shim is executed -- its capacity checks, its status -> exception mapping, the order
// and meaning of the arguments it hands to the C ABI -- and compared with direct C-ABI calls.  What it does not buy: anything
// about a real JVM (class loading, the Java half, GC and direct-buffer lifetime, the real jni.h's layout).  The image has
// no JDK; INTEGRATION.md says how to build and run the real thing.
//
// Semantics kept from the JNI specification: *ArrayRegion calls outside the array raise ArrayIndexOutOfBoundsException and
// copy nothing; GetDirectBufferAddress of a non-direct buffer is NULL and its capacity -1; ThrowNew with an exception already
// pending is a test failure here (the shim promises never to do that), recorded in `violations`.
#include <jni.h>

#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace {

enum Kind { K_CLASS, K_LONGS, K_INTS, K_FLOATS, K_DOUBLES, K_OBJECTS, K_BUFFER };

struct Obj : _jobject {
    Kind kind;
    std::string name;               // K_CLASS
    std::vector<jlong> longs;
    std::vector<jint> ints;
    std::vector<jfloat> floats;
    std::vector<jdouble> doubles;
    std::vector<Obj*> objects;
    void* addr = nullptr;           // K_BUFFER (nullptr: a heap buffer, not direct)
    jlong capacity = -1;            // K_BUFFER, in ELEMENTS of the buffer's type, as GetDirectBufferCapacity reports it
    explicit Obj(Kind k) : kind(k) {}
};

struct State {
    std::vector<std::unique_ptr<Obj>> objects;
    std::map<std::string, Obj*> classes;
    std::string pending_class, pending_msg;
    bool pending = false;
    int violations = 0;             // JNI calls made with an exception pending (other than ExceptionCheck), or on wrong kinds
    int critical = 0;               // open GetPrimitiveArrayCritical regions: no other JNI call may be made inside one
    std::string violation;
    Obj* make(Kind k)
    {
        objects.emplace_back(new Obj(k));
        return objects.back().get();
    }
    void violate(const char* what)
    {
        ++violations;
        if (violation.empty()) violation = what;
    }
    void raise(const char* cls, const char* msg)
    {
        pending = true;
        pending_class = cls;
        pending_msg = msg;
    }
};

State g;
JNIEnv_ g_env;

Obj* as(jobject o, Kind k, const char* what)
{
    Obj* p = static_cast<Obj*>(o);
    if (!p || p->kind != k) {
        g.violate(what);
        return nullptr;
    }
    return p;
}

void no_pending(const char* what)
{
    if (g.pending) g.violate(what);
    if (g.critical > 0) g.violate("a JNI call inside a GetPrimitiveArrayCritical region");
}

template <class T>
bool region_ok(const std::vector<T>& v, jsize start, jsize len)
{
    if (start < 0 || len < 0 || static_cast<size_t>(start) + static_cast<size_t>(len) > v.size()) {
        g.raise("java/lang/ArrayIndexOutOfBoundsException", "array region");
        return false;
    }
    return true;
}

}  // namespace

// ---- the JNIEnv members the shim uses ----------------------------------------------------------------------------------
jclass JNIEnv_::FindClass(const char* name)
{
    no_pending("FindClass with an exception pending");
    Obj*& c = g.classes[name];
    if (!c) {
        c = g.make(K_CLASS);
        c->name = name;
    }
    return reinterpret_cast<jclass>(static_cast<_jobject*>(c));
}

jint JNIEnv_::ThrowNew(jclass clazz, const char* msg)
{
    no_pending("ThrowNew with an exception pending");
    Obj* c = as(clazz, K_CLASS, "ThrowNew: not a class");
    if (!c) return -1;
    g.raise(c->name.c_str(), msg ? msg : "");
    return 0;
}

jboolean JNIEnv_::ExceptionCheck() { return g.pending ? JNI_TRUE : JNI_FALSE; }

jsize JNIEnv_::GetArrayLength(jarray array)
{
    no_pending("GetArrayLength with an exception pending");
    Obj* a = static_cast<Obj*>(static_cast<_jobject*>(array));
    if (!a) {
        g.violate("GetArrayLength(null)");
        return 0;
    }
    switch (a->kind) {
        case K_LONGS: return static_cast<jsize>(a->longs.size());
        case K_INTS: return static_cast<jsize>(a->ints.size());
        case K_FLOATS: return static_cast<jsize>(a->floats.size());
        case K_DOUBLES: return static_cast<jsize>(a->doubles.size());
        case K_OBJECTS: return static_cast<jsize>(a->objects.size());
        default: g.violate("GetArrayLength: not an array"); return 0;
    }
}

jobject JNIEnv_::GetObjectArrayElement(jobjectArray array, jsize index)
{
    no_pending("GetObjectArrayElement with an exception pending");
    Obj* a = as(array, K_OBJECTS, "GetObjectArrayElement: not an Object[]");
    if (!a) return nullptr;
    if (index < 0 || static_cast<size_t>(index) >= a->objects.size()) {
        g.raise("java/lang/ArrayIndexOutOfBoundsException", "object array index");
        return nullptr;
    }
    return a->objects[static_cast<size_t>(index)];
}

void JNIEnv_::GetIntArrayRegion(jintArray array, jsize start, jsize len, jint* buf)
{
    no_pending("GetIntArrayRegion with an exception pending");
    Obj* a = as(array, K_INTS, "GetIntArrayRegion: not an int[]");
    if (a && region_ok(a->ints, start, len) && len) std::memcpy(buf, a->ints.data() + start, sizeof(jint) * static_cast<size_t>(len));
}

void JNIEnv_::GetLongArrayRegion(jlongArray array, jsize start, jsize len, jlong* buf)
{
    no_pending("GetLongArrayRegion with an exception pending");
    Obj* a = as(array, K_LONGS, "GetLongArrayRegion: not a long[]");
    if (a && region_ok(a->longs, start, len) && len) std::memcpy(buf, a->longs.data() + start, sizeof(jlong) * static_cast<size_t>(len));
}

void JNIEnv_::SetLongArrayRegion(jlongArray array, jsize start, jsize len, const jlong* buf)
{
    no_pending("SetLongArrayRegion with an exception pending");
    Obj* a = as(array, K_LONGS, "SetLongArrayRegion: not a long[]");
    if (a && region_ok(a->longs, start, len) && len) std::memcpy(a->longs.data() + start, buf, sizeof(jlong) * static_cast<size_t>(len));
}

void JNIEnv_::GetFloatArrayRegion(jfloatArray array, jsize start, jsize len, jfloat* buf)
{
    no_pending("GetFloatArrayRegion with an exception pending");
    Obj* a = as(array, K_FLOATS, "GetFloatArrayRegion: not a float[]");
    if (a && region_ok(a->floats, start, len) && len) std::memcpy(buf, a->floats.data() + start, sizeof(jfloat) * static_cast<size_t>(len));
}

void JNIEnv_::SetDoubleArrayRegion(jdoubleArray array, jsize start, jsize len, const jdouble* buf)
{
    no_pending("SetDoubleArrayRegion with an exception pending");
    Obj* a = as(array, K_DOUBLES, "SetDoubleArrayRegion: not a double[]");
    if (a && region_ok(a->doubles, start, len) && len) std::memcpy(a->doubles.data() + start, buf, sizeof(jdouble) * static_cast<size_t>(len));
}

void* JNIEnv_::GetPrimitiveArrayCritical(jarray array, jboolean* isCopy)
{
    no_pending("GetPrimitiveArrayCritical with an exception pending");
    Obj* a = static_cast<Obj*>(static_cast<_jobject*>(array));
    if (!a || a->kind != K_FLOATS) { g.violate("GetPrimitiveArrayCritical: not a float[] (the only kind the shim holds critically)"); return nullptr; }
    if (isCopy) *isCopy = JNI_FALSE;
    g.critical += 1;
    return a->floats.data();
}

void JNIEnv_::ReleasePrimitiveArrayCritical(jarray array, void* carray, jint mode)
{
    Obj* a = static_cast<Obj*>(static_cast<_jobject*>(array));
    if (!a || a->kind != K_FLOATS || carray != a->floats.data() || g.critical <= 0) { g.violate("ReleasePrimitiveArrayCritical without its Get"); return; }
    (void)mode;                                             // the fake never copies: there is nothing to commit or abort
    g.critical -= 1;
}

jobject JNIEnv_::NewDirectByteBuffer(void* address, jlong capacity)
{
    no_pending("NewDirectByteBuffer with an exception pending");
    Obj* b = g.make(K_BUFFER);
    b->addr = address;
    b->capacity = capacity;         // a ByteBuffer: its elements are bytes
    return b;
}

void* JNIEnv_::GetDirectBufferAddress(jobject buf)
{
    no_pending("GetDirectBufferAddress with an exception pending");
    Obj* b = as(buf, K_BUFFER, "GetDirectBufferAddress: not a buffer");
    return b ? b->addr : nullptr;
}

jlong JNIEnv_::GetDirectBufferCapacity(jobject buf)
{
    no_pending("GetDirectBufferCapacity with an exception pending");
    Obj* b = as(buf, K_BUFFER, "GetDirectBufferCapacity: not a buffer");
    return b && b->addr ? b->capacity : -1;
}

// ---- the driver API of the test ------------------------------------------------------------------------------------------
extern "C" {

JNIEXPORT void* fake_env(void) { return &g_env; }

// forget every object and any pending exception (between test cases)
JNIEXPORT void fake_reset(void)
{
    g.objects.clear();
    g.classes.clear();
    g.pending = false;
    g.pending_class.clear();
    g.pending_msg.clear();
    g.violations = 0;
    g.violation.clear();
    g.critical = 0;
}

JNIEXPORT void* fake_long_array(const int64_t* v, int n)
{
    Obj* a = g.make(K_LONGS);
    a->longs.assign(v, v + n);
    return static_cast<_jobject*>(a);
}

JNIEXPORT void* fake_int_array(const int32_t* v, int n)
{
    Obj* a = g.make(K_INTS);
    a->ints.assign(v, v + n);
    return static_cast<_jobject*>(a);
}

JNIEXPORT void* fake_float_array(const float* v, int n)
{
    Obj* a = g.make(K_FLOATS);
    a->floats.assign(v, v + n);
    return static_cast<_jobject*>(a);
}

JNIEXPORT void* fake_double_array(int n)
{
    Obj* a = g.make(K_DOUBLES);
    a->doubles.assign(static_cast<size_t>(n), 0.0);
    return static_cast<_jobject*>(a);
}

JNIEXPORT void* fake_object_array(void* const* v, int n)
{
    Obj* a = g.make(K_OBJECTS);
    for (int i = 0; i < n; ++i) a->objects.push_back(static_cast<Obj*>(static_cast<_jobject*>(v[i])));
    return static_cast<_jobject*>(a);
}

// a direct buffer of `capacity` ELEMENTS at `addr` (a FloatBuffer view reports floats); addr == NULL: a heap buffer
JNIEXPORT void* fake_buffer(void* addr, int64_t capacity)
{
    Obj* b = g.make(K_BUFFER);
    b->addr = addr;
    b->capacity = capacity;
    return static_cast<_jobject*>(b);
}

JNIEXPORT int fake_read_longs(void* array, int64_t* out, int n)
{
    Obj* a = static_cast<Obj*>(static_cast<_jobject*>(array));
    if (!a || a->kind != K_LONGS || static_cast<size_t>(n) > a->longs.size()) return -1;
    std::memcpy(out, a->longs.data(), sizeof(int64_t) * static_cast<size_t>(n));
    return 0;
}

JNIEXPORT int fake_read_floats(void* array, float* out, int n)
{
    Obj* a = static_cast<Obj*>(static_cast<_jobject*>(array));
    if (!a || a->kind != K_FLOATS || static_cast<size_t>(n) > a->floats.size()) return -1;
    std::memcpy(out, a->floats.data(), sizeof(float) * static_cast<size_t>(n));
    return 0;
}

JNIEXPORT int fake_critical_depth(void) { return g.critical; }

JNIEXPORT int fake_read_doubles(void* array, double* out, int n)
{
    Obj* a = static_cast<Obj*>(static_cast<_jobject*>(array));
    if (!a || a->kind != K_DOUBLES || static_cast<size_t>(n) > a->doubles.size()) return -1;
    std::memcpy(out, a->doubles.data(), sizeof(double) * static_cast<size_t>(n));
    return 0;
}

JNIEXPORT void* fake_buffer_address(void* buf)
{
    Obj* b = static_cast<Obj*>(static_cast<_jobject*>(buf));
    return b && b->kind == K_BUFFER ? b->addr : nullptr;
}

JNIEXPORT int64_t fake_buffer_capacity(void* buf)
{
    Obj* b = static_cast<Obj*>(static_cast<_jobject*>(buf));
    return b && b->kind == K_BUFFER ? b->capacity : -1;
}

// the pending exception: 1 and its class / message copied out (and cleared), 0 when nothing is pending
JNIEXPORT int fake_take_exception(char* cls, int cls_cap, char* msg, int msg_cap)
{
    if (!g.pending) return 0;
    std::strncpy(cls, g.pending_class.c_str(), static_cast<size_t>(cls_cap - 1));
    cls[cls_cap - 1] = 0;
    std::strncpy(msg, g.pending_msg.c_str(), static_cast<size_t>(msg_cap - 1));
    msg[msg_cap - 1] = 0;
    g.pending = false;
    return 1;
}

// JNI rules the shim broke since the last reset (0 expected), with the first one's description
JNIEXPORT int fake_violations(char* what, int cap)
{
    std::strncpy(what, g.violation.c_str(), static_cast<size_t>(cap - 1));
    what[cap - 1] = 0;
    return g.violations;
}

}  // extern "C"
