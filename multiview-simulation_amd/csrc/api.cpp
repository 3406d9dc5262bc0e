// C ABI of libmvsim (include/mvsim.h): context, memory, host-buffer and device-buffer stage
// operators and the fused per-view pipeline.  Host orchestration only -- every voxel of
// arithmetic happens in the HIP kernels (kernels.hip, fftconv.hip, stencil.hip).
#include "common.h"

#include <algorithm>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

#include <emmintrin.h>

namespace mvsim {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- run-time switches ------------------------------------------------------------------------------
int parse_option(Options& o, const char* name, const char* value)
{
    if (!name || !value) return MVSIM_EINVAL;
    const std::string n(name), v(value);
    auto flag = [&](bool* dst) { if (v == "1" || v == "on" || v == "true") *dst = true; else if (v == "0" || v == "off" || v == "false") *dst = false; else return MVSIM_EINVAL; return MVSIM_OK; };
    if (n == "fft_zpass") {
        if (v == "auto") o.zpass = 0; else if (v == "direct") o.zpass = 1; else if (v == "fft") o.zpass = 2; else if (v == "inline") o.zpass = 3;
        else return MVSIM_EINVAL;
        return MVSIM_OK;
    }
    if (n == "fft_backend") {
        if (v == "custom" || v == "auto") o.rocfft = false; else if (v == "rocfft") o.rocfft = true; else return MVSIM_EINVAL;
        return MVSIM_OK;
    }
    if (n == "fused_rotate") {
        if (v == "1" || v == "on" || v == "lds") o.fused_rotate = 1; else if (v == "0" || v == "off") o.fused_rotate = 0;
        else if (v == "2" || v == "lane") o.fused_rotate = 2; else if (v == "3" || v == "auto") o.fused_rotate = 3;
        else return MVSIM_EINVAL;
        return MVSIM_OK;
    }
    if (n == "poisson_queue") {
        if (v == "1" || v == "on") o.poisson_queue = 1; else if (v == "0" || v == "off") o.poisson_queue = 0;
        else return MVSIM_EINVAL;
        return MVSIM_OK;
    }
    if (n == "poisson_queue_share") {                      // sixteenths of a block's voxels its queue segment holds
        if (v == "auto") { o.poisson_queue_share = 0; return MVSIM_OK; }
        if (v.empty() || v.size() > 2 || v.find_first_not_of("0123456789") != std::string::npos) return MVSIM_EINVAL;
        const int k = atoi(v.c_str());
        if (k < 1 || k > 16) return MVSIM_EINVAL;
        o.poisson_queue_share = k;
        return MVSIM_OK;
    }
    if (n == "attenuate") {
        if (v == "serial") o.attenuate_scan = false; else if (v == "scan") o.attenuate_scan = true; else return MVSIM_EINVAL;
        return MVSIM_OK;
    }
    if (n == "early_sum") return flag(&o.early_sum);
    if (n == "zconv_strided") return flag(&o.zconv_strided);
    if (n == "exp") {                                  // A/B bits of tools/ and the tests (common.h): 0 .. 15
        if (v.empty() || v.size() > 2 || v.find_first_not_of("0123456789") != std::string::npos) return MVSIM_EINVAL;
        const int k = atoi(v.c_str());
        if (k > 15) return MVSIM_EINVAL;
        o.exp = k;
        return MVSIM_OK;
    }
    if (n == "fuse_tail") return flag(&o.fuse_tail);
    if (n == "psf_overlap") return flag(&o.psf_overlap);
    if (n == "fused_fftx") {
        if (v == "auto") o.fused_fftx = 2; else if (v == "1" || v == "on") o.fused_fftx = 1; else if (v == "0" || v == "off") o.fused_fftx = 0;
        else return MVSIM_EINVAL;
        return MVSIM_OK;
    }
    if (n == "tail_overlap") {
        if (v == "0" || v == "off") o.tail_overlap = 0; else if (v == "1" || v == "on" || v == "own") o.tail_overlap = 1;
        else if (v == "2" || v == "any") o.tail_overlap = 2; else return MVSIM_EINVAL;
        return MVSIM_OK;
    }
    if (n == "acq_transfer") {
        if (v == "auto" || v == "u16") o.acq_u16 = 1; else if (v == "f32") o.acq_u16 = 0; else return MVSIM_EINVAL;
        return MVSIM_OK;
    }
    if (n == "host_threads") {
        if (v == "auto") { o.host_threads = 0; return MVSIM_OK; }
        if (v.empty() || v.size() > 3 || v.find_first_not_of("0123456789") != std::string::npos) return MVSIM_EINVAL;
        const int k = atoi(v.c_str());
        if (k < 1 || k > 256) return MVSIM_EINVAL;
        o.host_threads = k;
        return MVSIM_OK;
    }
    if (n == "view_batch") {
        if (v == "auto") o.view_batch = 2; else if (v == "1" || v == "on") o.view_batch = 1; else if (v == "0" || v == "off") o.view_batch = 0;
        else return MVSIM_EINVAL;
        return MVSIM_OK;
    }
    if (n == "view_lanes") {
        if (v == "auto") { o.view_lanes = 0; return MVSIM_OK; }
        if (v.empty() || v.size() > 2 || v.find_first_not_of("0123456789") != std::string::npos) return MVSIM_EINVAL;
        const int k = atoi(v.c_str());
        if (k < 1 || k > MVSIM_MAX_VIEWS) return MVSIM_EINVAL;
        o.view_lanes = k;
        return MVSIM_OK;
    }
    if (n == "graph") { bool g = false; const int rc = flag(&g); o.graph = g ? 1 : 0; return rc; }
    if (n == "broadcast") {
        if (v == "scatter_allgather" || v == "auto") { o.bcast_ring = false; o.bcast_peer_copy = false; o.bcast_pipelined = false; }
        else if (v == "ring") { o.bcast_ring = true; o.bcast_peer_copy = false; o.bcast_pipelined = false; }
        else if (v == "peer_copy") { o.bcast_ring = false; o.bcast_peer_copy = true; o.bcast_pipelined = false; }
        else if (v == "pipelined") { o.bcast_ring = false; o.bcast_peer_copy = false; o.bcast_pipelined = true; }
        else return MVSIM_EINVAL;
        return MVSIM_OK;
    }
    if (n == "skip_empty") return flag(&o.skip_empty);
    if (n == "fft_pad") {
        long long a = 0, b = 0, c = 0;
        if (v == "auto" || v.empty()) { o.fft_pad[0] = o.fft_pad[1] = o.fft_pad[2] = 0; return MVSIM_OK; }
        if (sscanf(v.c_str(), "%lld,%lld,%lld", &a, &b, &c) != 3 || a < 0 || b < 0 || c < 0) return MVSIM_EINVAL;
        o.fft_pad[0] = a; o.fft_pad[1] = b; o.fft_pad[2] = c;
        return MVSIM_OK;
    }
    return MVSIM_EINVAL;
}

// Process defaults, read from the environment exactly once (MVSIM_FFT_ZPASS=fft|direct, MVSIM_FFT_BACKEND=rocfft,
// MVSIM_FFT_PAD=px,py,pz, MVSIM_NO_FUSED_ROTATE, MVSIM_POISSON_NOQUEUE, MVSIM_NO_EARLY_SUM, MVSIM_GRAPH).
const Options& env_options()
{
    static Options o;
    static std::once_flag once;
    std::call_once(once, [] {
        if (const char* e = getenv("MVSIM_FFT_ZPASS")) (void)parse_option(o, "fft_zpass", e);
        if (const char* e = getenv("MVSIM_FFT_BACKEND")) (void)parse_option(o, "fft_backend", e);
        if (const char* e = getenv("MVSIM_FFT_PAD")) (void)parse_option(o, "fft_pad", e);
        if (getenv("MVSIM_NO_FUSED_ROTATE")) o.fused_rotate = 0;
        if (getenv("MVSIM_POISSON_NOQUEUE")) o.poisson_queue = 0;
        if (getenv("MVSIM_NO_EARLY_SUM")) o.early_sum = false;
        if (getenv("MVSIM_NO_FUSE_TAIL")) o.fuse_tail = false;
        if (const char* e = getenv("MVSIM_GRAPH")) (void)parse_option(o, "graph", e);
        if (const char* e = getenv("MVSIM_BROADCAST")) (void)parse_option(o, "broadcast", e);
        // MVSIM_OPTIONS="name=value;name=value": any option by its mvsim_set_option name (experiments, A/B runs)
        if (const char* e = getenv("MVSIM_OPTIONS")) {
            std::string all(e);
            size_t pos = 0;
            while (pos < all.size()) {
                size_t end = all.find(';', pos);
                if (end == std::string::npos) end = all.size();
                const std::string kv = all.substr(pos, end - pos);
                const size_t eq = kv.find('=');
                if (eq != std::string::npos) (void)parse_option(o, kv.substr(0, eq).c_str(), kv.substr(eq + 1).c_str());
                pos = end + 1;
            }
        }
    });
    return o;
}

int ensure_lds_attr(mvsim_ctx* ctx, const void* kernel, size_t bytes)
{
    if (bytes <= 64 * 1024) return MVSIM_OK;
    if (bytes > 160 * 1024) { set_error("kernel needs %zu bytes of LDS (> 160 KiB)", bytes); return MVSIM_EINVAL; }
    if (ctx->lds_attr_set.count(kernel)) return MVSIM_OK;
    MVSIM_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    ctx->lds_attr_set.insert(kernel);
    return MVSIM_OK;
}

// Bumped whenever a workspace moves or goes away: captured view graphs hold raw workspace addresses and must not be
// replayed across such a change (view_graph_launch compares the epoch it captured under).
int DevBuf::reserve(size_t need)
{
    if (need <= bytes) return MVSIM_OK;
    if (epoch) *epoch += 1;
    if (p) { (void)hipFree(p); p = nullptr; bytes = 0; }
    hipError_t e = hipMalloc(&p, need);
    if (e != hipSuccess) {
        p = nullptr;
        set_error("hipMalloc(%zu bytes) failed: %s", need, hipGetErrorString(e));
        return MVSIM_ENOMEM;
    }
    bytes = need;
    return MVSIM_OK;
}

void DevBuf::release()
{
    if (p && epoch) *epoch += 1;
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
}

int PinnedRing::acquire(size_t need, int* slot)
{
    const int i = next;
    next = (next + 1) % SLOTS;
    if (busy[i]) { MVSIM_HIP(hipEventSynchronize(ev[i])); busy[i] = false; }
    if (!ev[i]) MVSIM_HIP(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
    if (bytes[i] < need) {
        if (p[i]) { (void)hipHostFree(p[i]); p[i] = nullptr; bytes[i] = 0; }
        MVSIM_HIP(hipHostMalloc(&p[i], need, hipHostMallocDefault));
        bytes[i] = need;
    }
    *slot = i;
    return MVSIM_OK;
}

void PinnedRing::release_all()
{
    for (int i = 0; i < SLOTS; ++i) {
        if (busy[i]) (void)hipEventSynchronize(ev[i]);
        if (ev[i]) (void)hipEventDestroy(ev[i]);
        if (p[i]) (void)hipHostFree(p[i]);
        p[i] = nullptr; bytes[i] = 0; ev[i] = nullptr; busy[i] = false;
    }
}

// ---- affine model (mpicbg AffineModel3D semantics; SimulateMultiViewDataset.java:80-102) -------
static void ident(double m[12])
{
    for (int i = 0; i < 12; ++i) m[i] = 0.0;
    m[0] = m[5] = m[10] = 1.0;
}

// a <- b o a
static void pre_concat(double a[12], const double b[12])
{
    double r[12];
    for (int i = 0; i < 3; ++i) {
        const double* bi = b + 4 * i;
        for (int j = 0; j < 3; ++j) r[4 * i + j] = bi[0] * a[j] + bi[1] * a[4 + j] + bi[2] * a[8 + j];
        r[4 * i + 3] = bi[0] * a[3] + bi[1] * a[7] + bi[2] * a[11] + bi[3];
    }
    std::memcpy(a, r, sizeof(r));
}

void axis_rotation_host(const int64_t dim[3], int axis, int degrees, double m[12])
{
    // centre = (max - min) / 2 in integer arithmetic (SMVD:84-86)
    double c[3];
    for (int d = 0; d < 3; ++d) c[d] = (double)((dim[d] - 1) / 2);
    // (float)Math.toRadians(degrees) (SMVD:90)
    const double theta = (double)(float)((double)degrees * 0.017453292519943295);
    const double co = std::cos(theta), si = std::sin(theta);
    double t1[12], rot[12], t2[12];
    ident(t1); ident(rot); ident(t2);
    t1[3] = -c[0]; t1[7] = -c[1]; t1[11] = -c[2];
    t2[3] = c[0];  t2[7] = c[1];  t2[11] = c[2];
    switch (axis) {
        case 0:  rot[5] = co; rot[6] = -si; rot[9] = si;  rot[10] = co; break;
        case 1:  rot[0] = co; rot[2] = si;  rot[8] = -si; rot[10] = co; break;
        default: rot[0] = co; rot[1] = -si; rot[4] = si;  rot[5] = co;  break;
    }
    pre_concat(t1, rot);   // SMVD:98
    pre_concat(t1, t2);    // SMVD:99
    std::memcpy(m, t1, 12 * sizeof(double));
}

void affine_invert_host(const double m[12], double v[12])
{
    const double a = m[0], b = m[1], c = m[2], d = m[4], e = m[5], f = m[6], g = m[8], h = m[9], i = m[10];
    const double det = a * e * i + d * h * c + g * b * f - c * e * g - f * h * a - i * b * d;
    v[0] = (e * i - f * h) / det;  v[1] = (c * h - b * i) / det;  v[2] = (b * f - c * e) / det;
    v[4] = (f * g - d * i) / det;  v[5] = (a * i - c * g) / det;  v[6] = (c * d - a * f) / det;
    v[8] = (d * h - e * g) / det;  v[9] = (b * g - a * h) / det;  v[10] = (a * e - b * d) / det;
    v[3]  = -v[0] * m[3] - v[1] * m[7] - v[2] * m[11];
    v[7]  = -v[4] * m[3] - v[5] * m[7] - v[6] * m[11];
    v[11] = -v[8] * m[3] - v[9] * m[7] - v[10] * m[11];
}

static int check_dim(const int64_t dim[3])
{
    MVSIM_CHECK_ARG(dim != nullptr, "dim is null");
    MVSIM_CHECK_ARG(dim[0] >= 1 && dim[1] >= 1 && dim[2] >= 1, "dimensions must be >= 1");
    MVSIM_CHECK_ARG(dim[0] <= 65535 * 4 && dim[1] <= 65535 && dim[2] <= 65535, "dimension too large for one launch");
    return MVSIM_OK;
}

static int64_t nvox(const int64_t dim[3]) { return dim[0] * dim[1] * dim[2]; }

int join_tail(mvsim_ctx* ctx)
{
    if (ctx && ctx->tail_pending) {
        MVSIM_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_tail, 0));
        ctx->tail_pending = false;
    }
    return MVSIM_OK;
}

// every entry point starts here: the device, and a pending tail ordered in front of what the call enqueues
static int set_device(mvsim_ctx* ctx, bool keep_tail = false)
{
    MVSIM_CHECK_ARG(ctx != nullptr, "ctx is null");
    ev_rebalance(ctx);
    MVSIM_HIP(hipSetDevice(ctx->device));
    if (!keep_tail) MVSIM_TRY(join_tail(ctx));
    return MVSIM_OK;
}

// The queue share of the context's next sampled view (QueueMode).  Option given: that.  Auto: what this context's views have needed so
// far -- k_poisson_refused leaves the sixteenths the fullest refused segment would have needed in a page-locked word, read here without
// synchronising: a view whose segments refuse voxels still gives the right counts (slower), and the views after it get the larger queue.
static int queue_mode_next(mvsim_ctx* ctx, QueueMode* qm)
{
    qm->share = 0; qm->hint = nullptr;
    if (ctx->opt.poisson_queue != 1) return MVSIM_OK;
    if (ctx->opt.poisson_queue_share > 0) { qm->share = ctx->opt.poisson_queue_share; return MVSIM_OK; }
    const unsigned int seen = *reinterpret_cast<volatile unsigned int*>(ctx->queue_hint);
    if ((int)seen >= ctx->queue_share_learned && seen != 0u) ctx->queue_share_learned = seen >= 15u ? 16 : (int)seen + 1;   // one sixteenth of headroom
    qm->share = QUEUE_SHARE_AUTO + ctx->queue_share_learned;
    qm->hint = ctx->queue_hint;
    return MVSIM_OK;
}

// Tools.normImage on the host (Tools.java:112-132), in place (Q5): double sum, (float)(v / sum).
static void psf_normalise_host(float* psf_host, int64_t n)
{
    // pairwise (cascade) double summation: same order of magnitude of error as mpicbg RealSum.  A binary counter of
    // partial sums (level l holds the sum of 2^l consecutive elements); aligned blocks of 16 enter it at level 4 with
    // their balanced tree written out -- the same additions in the same association as 16 single pushes (IEEE addition is
    // commutative), several times faster: this loop is on the host path of every view (29 791 taps for a 31^3 PSF).
    double lvl[64]; bool used[64] = {};
    int64_t i = 0;
    for (; i + 16 <= n; i += 16) {
        const float* q = psf_host + i;
        double t[8];
        for (int k = 0; k < 8; ++k) t[k] = (double)q[2 * k] + (double)q[2 * k + 1];
        double s = ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
        int l = 4;
        while (used[l]) { used[l] = false; s += lvl[l]; ++l; }
        used[l] = true; lvl[l] = s;
    }
    for (; i < n; ++i) {
        double s = (double)psf_host[i];
        int l = 0;
        while (used[l]) { used[l] = false; s += lvl[l]; ++l; }
        used[l] = true; lvl[l] = s;
    }
    double sum = 0.0;
    for (int l = 0; l < 64; ++l) if (used[l]) sum += lvl[l];
    for (int64_t i = 0; i < n; ++i) psf_host[i] = (float)((double)psf_host[i] / sum);
}

// ---- host side of the 16-bit acquisition transfer ----------------------------------------------------------------------------
// A 512^3 acquisition is 0.54 GB of float32 that hold small integers (Poisson counts, Tools.java:84): it crosses PCIe as 0.27 GB of
// uint16 and is widened here, by a few host threads with streaming stores (the destination -- the caller's buffer -- is written once
// and not read back by us), while the next view's transfer is already running.  Process-wide pool, created on first use.
namespace {
class HostPool {
public:
    static HostPool& get() { static HostPool p; return p; }
    // fn(chunk) for chunk = 0 .. chunks-1 on up to `threads` threads (the caller's thread takes part); returns when all are done.
    // One job at a time: callers on different host threads (one context each) queue up behind each other.
    void run(int chunks, int threads, const std::function<void(int)>& fn)
    {
        if (chunks <= 0) return;
        std::lock_guard<std::mutex> one_job(run_m_);
        threads = std::max(1, std::min(threads, chunks));
        std::unique_lock<std::mutex> lk(m_);
        while ((int)workers_.size() < threads - 1) {
            const int id = (int)workers_.size();
            workers_.emplace_back([this, id] { loop(id); });
        }
        fn_ = &fn; next_ = 0; total_ = chunks; pending_ = chunks; helpers_ = threads - 1; gen_ += 1;
        cv_.notify_all();
        lk.unlock();
        work();
        lk.lock();
        done_.wait(lk, [this] { return pending_ == 0; });
        fn_ = nullptr;
    }
private:
    HostPool() = default;
    ~HostPool()
    {
        { std::lock_guard<std::mutex> lk(m_); stop_ = true; }
        cv_.notify_all();
        for (auto& t : workers_) t.join();
    }
    void work()
    {
        for (;;) {
            int c;
            const std::function<void(int)>* f;
            {
                std::lock_guard<std::mutex> lk(m_);
                if (!fn_ || next_ >= total_) return;
                c = next_++; f = fn_;
            }
            (*f)(c);                                   // (run() does not return before pending_ is 0, so *f outlives every call)
            std::lock_guard<std::mutex> lk(m_);
            if (--pending_ == 0) done_.notify_all();
        }
    }
    void loop(int id)
    {
        unsigned long long seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return stop_ || (gen_ != seen && fn_ && next_ < total_ && id < helpers_); });
                if (stop_) return;
                seen = gen_;
            }
            work();
        }
    }
    std::mutex m_, run_m_;
    std::condition_variable cv_, done_;
    std::vector<std::thread> workers_;
    const std::function<void(int)>* fn_ = nullptr;
    int next_ = 0, total_ = 0, pending_ = 0, helpers_ = 0;
    unsigned long long gen_ = 0;
    bool stop_ = false;
};

// dst[i] = (float) src[i], i in [0, n): 8 values per step, streaming stores where the destination is 16-byte aligned
void widen_u16(const unsigned short* src, float* dst, long long n)
{
    long long i = 0;
    while (i < n && (reinterpret_cast<uintptr_t>(dst + i) & 15) != 0) { dst[i] = (float)src[i]; ++i; }
    const __m128i zero = _mm_setzero_si128();
    for (; i + 8 <= n; i += 8) {
        const __m128i v = _mm_loadu_si128(reinterpret_cast<const __m128i*>(src + i));
        _mm_stream_ps(dst + i, _mm_cvtepi32_ps(_mm_unpacklo_epi16(v, zero)));
        _mm_stream_ps(dst + i + 4, _mm_cvtepi32_ps(_mm_unpackhi_epi16(v, zero)));
    }
    for (; i < n; ++i) dst[i] = (float)src[i];
    _mm_sfence();
}
}  // namespace

static int host_threads_of(const mvsim_ctx* ctx)
{
    if (ctx->opt.host_threads > 0) return ctx->opt.host_threads;
    const unsigned hw = std::thread::hardware_concurrency();
    return (int)std::max(1u, std::min(16u, hw ? hw : 1u));
}

// Normalise the PSF on the host exactly as Tools.normImage does, in place (Q5), then place it in device memory.
static int psf_prepare(mvsim_ctx* ctx, float* psf_host, const int64_t kdim[3], const int64_t dim[3])
{
    MVSIM_CHECK_ARG(psf_host != nullptr && kdim != nullptr, "psf is null");
    MVSIM_CHECK_ARG(kdim[0] >= 1 && kdim[1] >= 1 && kdim[2] >= 1, "psf dimensions must be >= 1");
    (void)dim;
    const int64_t n = kdim[0] * kdim[1] * kdim[2];
    psf_normalise_host(psf_host, n);
    MVSIM_TRY(ctx->psf_dev.reserve((size_t)n * sizeof(float)));
    int slot = 0;
    MVSIM_TRY(ctx->pinned.acquire((size_t)n * sizeof(float), &slot));
    std::memcpy(ctx->pinned.p[slot], psf_host, (size_t)n * sizeof(float));
    MVSIM_HIP(hipMemcpyAsync(ctx->psf_dev.p, ctx->pinned.p[slot], (size_t)n * sizeof(float), hipMemcpyHostToDevice,
                             ctx->stream));
    MVSIM_HIP(hipEventRecord(ctx->pinned.ev[slot], ctx->stream));
    ctx->pinned.busy[slot] = true;
    return MVSIM_OK;
}

static int pick_method(int method, const int64_t kdim[3])
{
    if (method == 1 || method == 2) return method;
    // direct stencil costs 2*K^3 flop/voxel; measured at 512^3 (tools/stencil_bench.py, profiles/r03_stencil_bench.txt): 3^3
    // 0.67x the FFT passes' time, 5^3 1.06x, 7^3 1.76x -- the FFT path takes over between 4 and 5 taps per axis
    return (kdim[0] * kdim[1] * kdim[2] <= 4 * 4 * 4) ? 2 : 1;
}

static int convolve_dev_impl(mvsim_ctx* ctx, const float* img, const int64_t dim[3], const int64_t kdim[3],
                             int method, float* out, ConvTail* tail = nullptr)
{
    MVSIM_CHECK_ARG(img != out, "convolve cannot run in place");
    if (pick_method(method, kdim) == 2) {
        if (tail) { tail->zstride = 1; tail->corr_done = false; }
        ev_begin(ctx, ST_CONVOLVE);
        MVSIM_TRY(launch_stencil(ctx, img, dim, ctx->psf_dev.as<float>(), kdim, out));
        ev_end(ctx, ST_CONVOLVE);
        return MVSIM_OK;
    }
    return fft_convolve(ctx, img, dim, ctx->psf_dev.as<float>(), kdim, out, tail);
}

static int scal_ptr(mvsim_ctx* ctx, double** partial, double** scal)
{
    MVSIM_TRY(ctx->partials.reserve(PARTIALS_BYTES));
    *partial = ctx->partials.as<double>();
    *scal = scal_of(ctx);
    return MVSIM_OK;
}

}  // namespace mvsim

using namespace mvsim;

static void async_release(mvsim_ctx* ctx);
static void view_graphs_release(mvsim_ctx* ctx);

extern "C" {

const char* mvsim_version(void) { return "mvsim 0.1.0 (gfx950)"; }
const char* mvsim_last_error(void) { return g_err; }

int mvsim_device_count(int* count)
{
    MVSIM_CHECK_ARG(count != nullptr, "count is null");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; set_error("hipGetDeviceCount: %s", hipGetErrorString(e)); return MVSIM_ENODEV; }
    *count = n;
    return MVSIM_OK;
}

int mvsim_create(int device, mvsim_ctx** out)
{
    MVSIM_CHECK_ARG(out != nullptr, "ctx out pointer is null");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        set_error("no HIP device available (libmvsim has no CPU fallback)");
        return MVSIM_ENODEV;
    }
    MVSIM_CHECK_ARG(device >= 0 && device < n, "device index out of range");
    MVSIM_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    MVSIM_HIP(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("device %d is %s; libmvsim is built for gfx950 only", device, prop.gcnArchName);
        return MVSIM_ENODEV;
    }
    mvsim_ctx* ctx = new (std::nothrow) mvsim_ctx();
    if (!ctx) { set_error("out of host memory"); return MVSIM_ENOMEM; }
    ctx->device = device;
    ctx->num_cu = prop.multiProcessorCount;
    ctx->opt = env_options();
    if (hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) { set_error("hipStreamCreate failed"); delete ctx; return MVSIM_EHIP; }
    ctx->stream = ctx->own_stream;
    // (here and not at the first sampled view: that one may be inside a stream capture)
    if (hipHostMalloc(reinterpret_cast<void**>(&ctx->queue_hint), 4 * sizeof(unsigned int), hipHostMallocDefault) != hipSuccess) {
        set_error("hipHostMalloc failed");
        (void)hipStreamDestroy(ctx->own_stream);
        delete ctx;
        return MVSIM_EHIP;
    }
    ctx->queue_hint[0] = ctx->queue_hint[1] = ctx->queue_hint[2] = ctx->queue_hint[3] = 0u;
    *out = ctx;
    return MVSIM_OK;
}

int mvsim_destroy(mvsim_ctx* ctx)
{
    if (!ctx) return MVSIM_OK;
    (void)hipSetDevice(ctx->device);
    (void)join_tail(ctx);
    (void)hipStreamSynchronize(ctx->stream);
    for (mvsim_ctx* l : ctx->lanes) (void)mvsim_destroy(l);
    ctx->lanes.clear();
    for (hipEvent_t e : ctx->lane_done) (void)hipEventDestroy(e);
    ctx->lane_done.clear();
    if (ctx->ev_lane_fork) { (void)hipEventDestroy(ctx->ev_lane_fork); ctx->ev_lane_fork = nullptr; }
    mvsim_comm_destroy(ctx);
    async_release(ctx);
    view_graphs_release(ctx);
    fft_release(ctx);
    ctx->vol_a.release(); ctx->vol_b.release(); ctx->vol_c.release(); ctx->out_buf.release();
    ctx->psf_dev.release(); ctx->view_tab.release(); ctx->sync_u16.release(); ctx->stencil_psf.release(); ctx->partials.release(); ctx->partials_e.release(); ctx->pqueue.release(); ctx->sphere_list.release(); ctx->weight_img.release(); ctx->weight_dim[0] = 0; ctx->plane_flags.release();
    ctx->host_gt.release(); ctx->host_rot.release(); ctx->host_att.release(); ctx->host_con.release();
    ctx->pinned.release_all();
    if (ctx->ev_created)
        for (int k = 0; k < mvsim_ctx::TIMING_SLOTS; ++k)
            for (int s = 0; s < ST_COUNT; ++s) { (void)hipEventDestroy(ctx->evr[k][s][0]); (void)hipEventDestroy(ctx->evr[k][s][1]); }
    if (ctx->tail_stream) { (void)hipStreamSynchronize(ctx->tail_stream); (void)hipStreamDestroy(ctx->tail_stream); (void)hipEventDestroy(ctx->ev_tail_fork); (void)hipEventDestroy(ctx->ev_tail); }
    if (ctx->side_stream) { (void)hipStreamDestroy(ctx->side_stream); (void)hipEventDestroy(ctx->ev_fork); (void)hipEventDestroy(ctx->ev_join); }
    if (ctx->empty_hint) (void)hipHostFree(ctx->empty_hint);
    if (ctx->queue_hint) (void)hipHostFree(ctx->queue_hint);
    if (ctx->sync_u16_host) (void)hipHostFree(ctx->sync_u16_host);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return MVSIM_OK;
}

int mvsim_set_stream(mvsim_ctx* ctx, void* hip_stream)
{
    MVSIM_TRY(set_device(ctx));
    ctx->stream = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : ctx->own_stream;
    return MVSIM_OK;
}

int mvsim_join(mvsim_ctx* ctx)
{
    return set_device(ctx);                              // joins a pending tail; nothing else to do
}

int mvsim_set_option(mvsim_ctx* ctx, const char* name, const char* value)
{
    MVSIM_CHECK_ARG(ctx != nullptr, "ctx is null");
    if (parse_option(ctx->opt, name, value) != MVSIM_OK) {
        set_error("invalid argument: option %s = %s", name ? name : "(null)", value ? value : "(null)");
        return MVSIM_EINVAL;
    }
    return MVSIM_OK;
}

int mvsim_synchronize(mvsim_ctx* ctx)
{
    MVSIM_TRY(set_device(ctx));
    MVSIM_HIP(hipStreamSynchronize(ctx->stream));
    return MVSIM_OK;
}

int mvsim_release_caches(mvsim_ctx* ctx)
{
    MVSIM_TRY(set_device(ctx));
    MVSIM_HIP(hipStreamSynchronize(ctx->stream));
    for (mvsim_ctx* l : ctx->lanes) MVSIM_TRY(mvsim_release_caches(l));
    MVSIM_HIP(hipSetDevice(ctx->device));
    async_release(ctx);
    view_graphs_release(ctx);                             // captured launches point into the workspaces released below
    fft_release(ctx);
    ctx->vol_a.release(); ctx->vol_b.release(); ctx->vol_c.release(); ctx->out_buf.release();
    ctx->pqueue.release(); ctx->psf_dev.release(); ctx->view_tab.release(); ctx->sync_u16.release(); ctx->stencil_psf.release(); ctx->sphere_list.release(); ctx->weight_img.release(); ctx->weight_dim[0] = 0; ctx->plane_flags.release();
    ctx->host_gt.release(); ctx->host_rot.release(); ctx->host_att.release(); ctx->host_con.release();
    if (ctx->sync_u16_host) { (void)hipHostFree(ctx->sync_u16_host); ctx->sync_u16_host = nullptr; ctx->sync_u16_host_bytes = 0; }
    return MVSIM_OK;
}

// Host-to-host copy on the library's host threads (the JNI shim's bulk copies between a Java heap array and a page-locked staging
// block: one JVM thread moves 0.54 GB in 16 ms into a live array and in 79 ms into a fresh one -- first-touch page faults --,
// profiles/r05_slab_copy.txt; several threads fault and copy in parallel).  ctx may be null: the process-wide default thread count.
int mvsim_host_copy(mvsim_ctx* ctx, void* dst, const void* src, size_t bytes)
{
    if (bytes == 0) return MVSIM_OK;
    if (!dst || !src) { set_error("invalid argument: host_copy with a null pointer"); return MVSIM_EINVAL; }
    const size_t chunk = (size_t)4 << 20;
    const unsigned hw = std::thread::hardware_concurrency();
    const int threads = ctx ? host_threads_of(ctx) : (int)std::max(1u, std::min(16u, hw ? hw : 1u));
    if (bytes <= chunk || threads <= 1) { std::memmove(dst, src, bytes); return MVSIM_OK; }
    const uintptr_t d = reinterpret_cast<uintptr_t>(dst), sr = reinterpret_cast<uintptr_t>(src);
    if (d < sr + bytes && sr < d + bytes) { std::memmove(dst, src, bytes); return MVSIM_OK; }      // overlapping ranges: one ordered move
    HostPool::get().run((int)((bytes + chunk - 1) / chunk), threads, [&](int c) {
        const size_t a = (size_t)c * chunk, n = std::min(chunk, bytes - a);
        std::memcpy(static_cast<char*>(dst) + a, static_cast<const char*>(src) + a, n);
    });
    return MVSIM_OK;
}

int mvsim_dev_alloc(mvsim_ctx* ctx, size_t bytes, void** dptr)
{
    MVSIM_TRY(set_device(ctx));
    MVSIM_CHECK_ARG(dptr != nullptr, "dptr is null");
    *dptr = nullptr;
    if (bytes == 0) return MVSIM_OK;
    MVSIM_HIP(hipMalloc(dptr, bytes));
    return MVSIM_OK;
}

int mvsim_dev_free(mvsim_ctx* ctx, void* dptr)
{
    MVSIM_TRY(set_device(ctx));
    if (dptr && ctx->peer_copy) {
        // peers' IPC mappings registered for this allocation must not outlive it (mvsim_comm_register_volume)
        void* base = nullptr;
        size_t size = 0;
        if (hipMemGetAddressRange(reinterpret_cast<hipDeviceptr_t*>(&base), &size, dptr) == hipSuccess) comm_forget_range(ctx, base, size);
        else comm_forget_range(ctx, dptr, 0);
    }
    if (dptr) MVSIM_HIP(hipFree(dptr));
    return MVSIM_OK;
}

// Page-locked host memory: copies between it and HBM run at PCIe speed instead of through the runtime's staging of
// pageable memory.  The JNI shim wraps such blocks in direct ByteBuffers (NewDirectByteBuffer), numpy wraps them through
// the buffer protocol.
int mvsim_host_alloc(mvsim_ctx* ctx, size_t bytes, void** hptr)
{
    MVSIM_TRY(set_device(ctx));
    MVSIM_CHECK_ARG(hptr != nullptr, "hptr is null");
    *hptr = nullptr;
    if (bytes == 0) return MVSIM_OK;
    hipError_t e = hipHostMalloc(hptr, bytes, hipHostMallocDefault);
    if (e != hipSuccess) {
        set_error("hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        return e == hipErrorOutOfMemory ? MVSIM_ENOMEM : MVSIM_EHIP;
    }
    return MVSIM_OK;
}

int mvsim_host_free(mvsim_ctx* ctx, void* hptr)
{
    if (ctx) {                                         // ctx may be NULL: blocks can outlive their context
        MVSIM_TRY(set_device(ctx));
        // a block that comes back from the allocator at the same address is a different ground truth
        for (int s = 0; s < mvsim_ctx::ASYNC_SLOTS; ++s)
            if (hptr && ctx->async_gt_src[s] == hptr) ctx->async_gt_src[s] = nullptr;
    }
    if (hptr) MVSIM_HIP(hipHostFree(hptr));
    return MVSIM_OK;
}

int mvsim_upload(mvsim_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes)
{
    MVSIM_TRY(set_device(ctx));
    if (bytes == 0) return MVSIM_OK;
    MVSIM_CHECK_ARG(dst_dev && src_host, "null pointer");
    MVSIM_HIP(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    MVSIM_HIP(hipStreamSynchronize(ctx->stream));
    return MVSIM_OK;
}

int mvsim_dev_memset(mvsim_ctx* ctx, void* dptr, int value, size_t bytes)
{
    MVSIM_TRY(set_device(ctx));
    if (bytes == 0) return MVSIM_OK;
    MVSIM_CHECK_ARG(dptr != nullptr, "null pointer");
    MVSIM_HIP(hipMemsetAsync(dptr, value, bytes, ctx->stream));
    return MVSIM_OK;
}

int mvsim_download(mvsim_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes)
{
    MVSIM_TRY(set_device(ctx));
    if (bytes == 0) return MVSIM_OK;
    MVSIM_CHECK_ARG(dst_host && src_dev, "null pointer");
    MVSIM_HIP(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    MVSIM_HIP(hipStreamSynchronize(ctx->stream));
    return MVSIM_OK;
}

// ---- host helpers ---------------------------------------------------------------------------------
int mvsim_axis_rotation(const int64_t dim[3], int axis, int degrees, double m[12])
{
    MVSIM_CHECK_ARG(dim && m, "null pointer");
    MVSIM_CHECK_ARG(axis >= 0 && axis <= 2, "axis must be 0, 1 or 2");
    axis_rotation_host(dim, axis, degrees, m);
    return MVSIM_OK;
}

int64_t mvsim_extract_nz(int64_t nz, int inc) { return inc < 1 ? -1 : (nz - 1) / inc + 1; }
int64_t mvsim_isotropic_nz(int64_t nz_acq, int inc) { return inc < 1 ? -1 : (nz_acq - 1) * inc + 1; }
double  mvsim_poisson_mul(double snr) { return std::pow(snr / std::sqrt(5.0), 2.0); }

void mvsim_view_params_default(mvsim_view_params* p)
{
    if (!p) return;
    p->axis = 0; p->degrees = 15; p->delta = (double)0.01f /* `final float attenuation = 0.01f` widened, SMVD:533,573 */; p->min_value = 0.0001f; p->target_average = 1.0f;
    p->inc = 3; p->snr = 25.0f; p->seed = 464232194ULL; p->stream = 0; p->conv_method = 0;
}

// ---- device-resident stage operators --------------------------------------------------------------
int mvsim_rotate_around_axis_dev(mvsim_ctx* ctx, const float* in, const int64_t dim[3], int axis, int degrees,
                                 float* out)
{
    MVSIM_TRY(set_device(ctx));
    MVSIM_TRY(check_dim(dim));
    MVSIM_CHECK_ARG(in && out && in != out, "null or aliased buffers");
    MVSIM_CHECK_ARG(axis >= 0 && axis <= 2, "axis must be 0, 1 or 2");
    double m[12];
    Affine inv;
    axis_rotation_host(dim, axis, degrees, m);
    affine_invert_host(m, inv.m);
    ev_begin(ctx, ST_ROTATE);
    MVSIM_TRY(launch_rotate(ctx->stream, in, out, dim, inv));
    ev_end(ctx, ST_ROTATE);
    return MVSIM_OK;
}

int mvsim_attenuate3d_dev(mvsim_ctx* ctx, const float* in, const int64_t dim[3], double delta, float* out)
{
    MVSIM_TRY(set_device(ctx));
    MVSIM_TRY(check_dim(dim));
    MVSIM_CHECK_ARG(in && out && in != out, "null or aliased buffers");
    MVSIM_CHECK_ARG(dim[0] <= dim[1], "attenuate3d: Nx > Ny walks outside the interval in the reference (steps = dimension(0))");
    ev_begin(ctx, ST_ATTENUATE);
    if (ctx->opt.attenuate_scan) MVSIM_TRY(launch_attenuate_scan(ctx->stream, in, out, dim, delta));
    else MVSIM_TRY(launch_attenuate(ctx->stream, in, out, dim, delta));
    ev_end(ctx, ST_ATTENUATE);
    return MVSIM_OK;
}

int mvsim_convolve_dev(mvsim_ctx* ctx, const float* img, const int64_t dim[3], float* psf_host,
                       const int64_t kdim[3], int method, float* out)
{
    MVSIM_TRY(set_device(ctx));
    MVSIM_TRY(check_dim(dim));
    MVSIM_CHECK_ARG(img && out, "null buffer");
    MVSIM_CHECK_ARG(method >= 0 && method <= 2, "method must be 0, 1 or 2");
    MVSIM_TRY(psf_prepare(ctx, psf_host, kdim, dim));
    return convolve_dev_impl(ctx, img, dim, kdim, method, out);
}

int mvsim_adjust_image_dev(mvsim_ctx* ctx, float* img, int64_t n, float min_value, float target_average,
                           double* correction)
{
    MVSIM_TRY(set_device(ctx));
    MVSIM_CHECK_ARG(img && n >= 1, "null buffer or empty image");
    double *partial, *scal;
    MVSIM_TRY(scal_ptr(ctx, &partial, &scal));
    ev_begin(ctx, ST_ADJUST);
    MVSIM_TRY(launch_sum(ctx->stream, img, n, partial, scal));
    MVSIM_TRY(launch_adjust_corr(ctx->stream, scal, n, min_value, target_average));
    MVSIM_TRY(launch_adjust_apply(ctx->stream, img, n, scal, min_value));
    ev_end(ctx, ST_ADJUST);
    if (correction) {
        MVSIM_HIP(hipMemcpyAsync(correction, scal + 1, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        MVSIM_HIP(hipStreamSynchronize(ctx->stream));
    }
    return MVSIM_OK;
}

int mvsim_extract_slices_dev(mvsim_ctx* ctx, const float* in, const int64_t dim[3], int inc, float snr,
                             uint64_t seed, uint32_t stream, float* out)
{
    MVSIM_TRY(set_device(ctx));
    MVSIM_TRY(check_dim(dim));
    MVSIM_CHECK_ARG(in && out, "null buffer");
    MVSIM_CHECK_ARG(inc >= 1, "inc must be >= 1");
    const bool noise = snr >= 0.0f;   // SMVD:211
    void* qws = nullptr;
    QueueMode qm;
    MVSIM_TRY(queue_mode_next(ctx, &qm));
    if (noise) { MVSIM_TRY(ctx->pqueue.reserve(poisson_queue_bytes_planes(dim[0] * dim[1], mvsim_extract_nz(dim[2], inc), qm.share))); qws = ctx->pqueue.p; }
    ev_begin(ctx, ST_EXTRACT);
    MVSIM_TRY(launch_extract(ctx->stream, in, out, dim, inc, false, nullptr, 0.0f, noise,
                             mvsim_poisson_mul((double)snr), seed, stream, 0, qws, qm));
    ev_end(ctx, ST_EXTRACT);
    return MVSIM_OK;
}

int mvsim_make_isotropic_dev(mvsim_ctx* ctx, const float* in, const int64_t dim[3], int inc, float* out)
{
    MVSIM_TRY(set_device(ctx));
    MVSIM_TRY(check_dim(dim));
    MVSIM_CHECK_ARG(in && out, "null buffer");
    MVSIM_CHECK_ARG(inc >= 1, "inc must be >= 1");
    return launch_make_isotropic(ctx->stream, in, out, dim, inc);
}

int mvsim_compute_weight_image_dev(mvsim_ctx* ctx, const int64_t dim[3], float* out)
{
    MVSIM_TRY(set_device(ctx));
    MVSIM_TRY(check_dim(dim));
    MVSIM_CHECK_ARG(out, "null buffer");
    return launch_weight_image(ctx->stream, out, dim);
}

// ---- cross-view weight normalisation -------------------------------------------------------------------
int mvsim_sum_views_dev(mvsim_ctx* ctx, const float* const* vols, int n_views, int64_t n, float* out)
{
    MVSIM_TRY(set_device(ctx));
    MVSIM_CHECK_ARG(vols && out && n >= 1, "null pointer or empty image");
    MVSIM_CHECK_ARG(n_views >= 1 && n_views <= MVSIM_MAX_VIEWS, "n_views must be in [1, MVSIM_MAX_VIEWS]");
    for (int v = 0; v < n_views; ++v) MVSIM_CHECK_ARG(vols[v] != nullptr, "null view pointer");
    return launch_weights(ctx->stream, const_cast<float* const*>(vols), n_views, n, nullptr, out, 0.0f, true);
}

int mvsim_normalize_weights_dev(mvsim_ctx* ctx, float* const* weights, int n_views, int64_t n, const float* sum_or_null,
                                float osem)
{
    MVSIM_TRY(set_device(ctx));
    MVSIM_CHECK_ARG(weights && n >= 1, "null pointer or empty image");
    MVSIM_CHECK_ARG(n_views >= 1 && n_views <= MVSIM_MAX_VIEWS, "n_views must be in [1, MVSIM_MAX_VIEWS]");
    for (int v = 0; v < n_views; ++v) MVSIM_CHECK_ARG(weights[v] != nullptr, "null view pointer");
    return launch_weights(ctx->stream, weights, n_views, n, sum_or_null, nullptr, osem, false);
}

int mvsim_normalize_weights(mvsim_ctx* ctx, float* const* weights, int n_views, int64_t n, float osem)
{
    MVSIM_TRY(set_device(ctx));
    MVSIM_CHECK_ARG(weights && n >= 1, "null pointer or empty image");
    MVSIM_CHECK_ARG(n_views >= 1 && n_views <= MVSIM_MAX_VIEWS, "n_views must be in [1, MVSIM_MAX_VIEWS]");
    const size_t bytes = (size_t)n * sizeof(float);
    std::vector<DevBuf> bufs((size_t)n_views);
    std::vector<float*> dptr((size_t)n_views);
    int rc = MVSIM_OK;
    for (int v = 0; v < n_views && rc == MVSIM_OK; ++v) {
        if (!weights[v]) { set_error("invalid argument: null view pointer"); rc = MVSIM_EINVAL; break; }
        rc = bufs[v].reserve(bytes);
        if (rc == MVSIM_OK && hipMemcpyAsync(bufs[v].p, weights[v], bytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) {
            set_error("upload of view %d failed", v); rc = MVSIM_EHIP;
        }
        dptr[v] = bufs[v].as<float>();
    }
    if (rc == MVSIM_OK) rc = launch_weights(ctx->stream, dptr.data(), n_views, n, nullptr, nullptr, osem, false);
    for (int v = 0; v < n_views && rc == MVSIM_OK; ++v)
        if (hipMemcpyAsync(weights[v], bufs[v].p, bytes, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) {
            set_error("download of view %d failed", v); rc = MVSIM_EHIP;
        }
    (void)hipStreamSynchronize(ctx->stream);
    for (auto& b : bufs) b.release();
    return rc;
}

// ---- fused per-view pipeline ------------------------------------------------------------------------
// Everything a view enqueues on the context stream behind the PSF upload: kernel launches only (no allocation once the
// workspaces have their size, no host synchronisation) -- which is what makes it capturable into a hipGraph.
// The host slabs of the *_zslabs entry points belong to the caller again when the call returns -- on EVERY path: copies from
// page-locked slabs are truly asynchronous, so an error return must not leave one in flight.
struct SyncOnExit {
    mvsim_ctx* c;
    ~SyncOnExit() { if (c && c->stream) (void)hipStreamSynchronize(c->stream); }
};
struct StreamSwap {               // enqueue on another stream for a scope; the context's stream comes back on every exit path
    mvsim_ctx* c;
    hipStream_t saved;
    StreamSwap(mvsim_ctx* ctx, hipStream_t s) : c(ctx), saved(ctx->stream) { ctx->stream = s; }
    ~StreamSwap() { c->stream = saved; }
};

static bool ranges_meet(const void* a, size_t abytes, const char* lo, const char* hi)
{
    const char* p = reinterpret_cast<const char*>(a);
    return a && lo && p < hi && lo < p + abytes;
}

// overlap_ok: the tail (extract + Poisson) may stay pending on the tail stream when the call returns (join_tail)
static int view_enqueue(mvsim_ctx* ctx, const float* gt, const int64_t dim[3], const int64_t kdim[3], const mvsim_view_params* p,
                        const mvsim_view_outputs* o, bool overlap_ok = false)
{
    const int64_t n = nvox(dim);
    const size_t vbytes = (size_t)n * sizeof(float);
    float* rot = o->rot;
    float* att = o->att;
    float* con = o->con;
    // (the attenuated volume gets its scratch only where a kernel writes it: behind the fused rotate + attenuate + x transform
    // nothing does, and a 512^3 / 1024^3 view keeps 0.5 / 4 GiB of HBM it would never touch)
    if (ctx->tail_pending) {
        // the previous view's tail still writes its acquisition and reads its convolved volume: this view's first stage
        // may run beside it only if it touches neither
        bool meet = false;
        for (int r = 0; r < 2; ++r)
            meet = meet || ranges_meet(gt, vbytes, ctx->tail_lo[r], ctx->tail_hi[r]) || ranges_meet(rot, vbytes, ctx->tail_lo[r], ctx->tail_hi[r]) ||
                   ranges_meet(att, vbytes, ctx->tail_lo[r], ctx->tail_hi[r]);
        if (meet) MVSIM_TRY(join_tail(ctx));
    }

    double m[12];
    Affine inv;
    axis_rotation_host(dim, p->axis, p->degrees, m);
    affine_invert_host(m, inv.m);
    // rotation about x: rotate and attenuate run as one kernel and `rot` is written only when requested
    bool fused = false;
    // ... and when the FFT passes follow, their x transform rides in the same kernel: `att` leaves the chip only if asked for
    bool x_done = false;
    const int* plane_nz = nullptr;
    ev_begin(ctx, ST_ROTATE);
    if (pick_method(p->conv_method, kdim) == 1)
        MVSIM_TRY(rotate_attenuate_fftx(ctx, gt, rot, o->att, dim, kdim, inv, p->delta, &x_done, &plane_nz));
    fused = x_done;
    if (!x_done) {
        if (!att) { MVSIM_TRY(ctx->vol_b.reserve(vbytes)); att = ctx->vol_b.as<float>(); }
        MVSIM_TRY(launch_rotate_attenuate(ctx->stream, gt, rot, att, dim, inv, p->delta, ctx->opt.fused_rotate, &fused));
    }
    if (!fused) {
        MVSIM_TRY(join_tail(ctx));                      // the rotation scratch is the buffer a pending tail reads
        if (!rot) { MVSIM_TRY(ctx->vol_a.reserve(vbytes)); rot = ctx->vol_a.as<float>(); }
        MVSIM_TRY(launch_rotate(ctx->stream, gt, rot, dim, inv));
    }
    ev_end(ctx, ST_ROTATE);
    // everything below reuses the workspaces of the previous view
    MVSIM_TRY(join_tail(ctx));
    if (!fused) {
        ev_begin(ctx, ST_ATTENUATE);
        MVSIM_TRY(launch_attenuate(ctx->stream, rot, att, dim, p->delta));
        ev_end(ctx, ST_ATTENUATE);
    }
    if (!con) { MVSIM_TRY(ctx->vol_a.reserve(vbytes)); con = ctx->vol_a.as<float>(); }   // vol_a: rot scratch is dead by now

    double *partial, *scal;
    MVSIM_TRY(scal_ptr(ctx, &partial, &scal));
    const int method = pick_method(p->conv_method, kdim);
    const bool materialise = o->con != nullptr;
    const bool noise = p->snr >= 0.0f;
    // only every inc-th plane is acquired: when the adjusted volume itself is not asked for, the last two passes of the
    // convolution need not produce the other planes (the convolution says whether it could honour that)
    ConvTail tail;
    const long long plane_vox = (long long)dim[0] * dim[1];
    tail.zstride = (!materialise && p->inc > 1 && (!noise || ctx->opt.poisson_queue == 1)) ? p->inc : 1;
    tail.corr_n = n; tail.min_value = p->min_value; tail.target_average = p->target_average;
    tail.x_done = x_done;
    tail.plane_nz = x_done ? plane_nz : nullptr;
    if (method == 1 && ctx->opt.fuse_tail && (!noise || ctx->opt.poisson_queue == 1)) {
        const size_t qb = noise ? fused_tail_queue_bytes(dim, kdim, p->inc, materialise, ctx->opt) : 0;
        if (!noise || qb > 0) {
            if (qb) MVSIM_TRY(ctx->pqueue.reserve(qb));
            tail.want_fuse = true;
            tail.min_value = p->min_value; tail.target_average = p->target_average;
            tail.con_adj = materialise ? con : nullptr;
            tail.acq = o->acq; tail.inc = p->inc; tail.noise = noise;
            tail.mul = mvsim_poisson_mul((double)p->snr); tail.seed = p->seed; tail.stream = p->stream;
        }
    }
    MVSIM_TRY(convolve_dev_impl(ctx, att, dim, kdim, method, con, &tail));
    if (tail.fused) return MVSIM_OK;   // pass E adjusted, extracted and sampled (phase 1); the resolver is enqueued behind it

    ev_begin(ctx, ST_ADJUST);
    if (method == 2) MVSIM_TRY(launch_sum(ctx->stream, con, n, partial, scal));   // FFT path sums in its crop epilogue
    if (!tail.corr_done) MVSIM_TRY(launch_adjust_corr(ctx->stream, scal, n, p->min_value, p->target_average));
    if (materialise) MVSIM_TRY(launch_adjust_apply(ctx->stream, con, n, scal, p->min_value));
    ev_end(ctx, ST_ADJUST);

    void* qws = nullptr;
    const int64_t n_out = dim[0] * dim[1] * mvsim_extract_nz(dim[2], p->inc);
    QueueMode qm;
    MVSIM_TRY(queue_mode_next(ctx, &qm));
    if (noise) { MVSIM_TRY(ctx->pqueue.reserve(poisson_queue_bytes_planes(dim[0] * dim[1], mvsim_extract_nz(dim[2], p->inc), qm.share))); qws = ctx->pqueue.p; }
    // The tail runs on a stream of its own and is joined by whatever the context does next (join_tail): the next view's
    // rotate+attenuate leaves most of the chip idle and runs beside it.
    // (not beside the fused rotate + attenuate + x transform of the next view: that kernel is bound by vector issue like the
    // sampler itself, and the two together measured slower than one after the other -- 17.6 against 17.2 ms per 8 views)
    if (x_done) overlap_ok = false;
    hipStream_t tail_on = ctx->stream;
    if (overlap_ok) {
        if (!ctx->tail_stream) {
            MVSIM_HIP(hipStreamCreateWithFlags(&ctx->tail_stream, hipStreamNonBlocking));
            MVSIM_HIP(hipEventCreateWithFlags(&ctx->ev_tail_fork, hipEventDisableTiming));
            MVSIM_HIP(hipEventCreateWithFlags(&ctx->ev_tail, hipEventDisableTiming));
        }
        MVSIM_HIP(hipEventRecord(ctx->ev_tail_fork, ctx->stream));
        MVSIM_HIP(hipStreamWaitEvent(ctx->tail_stream, ctx->ev_tail_fork, 0));
        tail_on = ctx->tail_stream;
    }
    StreamSwap swap(ctx, tail_on);
    ev_begin(ctx, ST_EXTRACT);
    if (tail.zstride > 1) {
        // `con` holds the acquired planes only: read them in order, count the RNG in source planes
        const int64_t cdim[3] = {dim[0], dim[1], mvsim_extract_nz(dim[2], p->inc)};
        MVSIM_TRY(launch_extract(ctx->stream, con, o->acq, cdim, 1, true, scal, p->min_value, noise,
                                 mvsim_poisson_mul((double)p->snr), p->seed, p->stream, 0, qws, qm, p->inc));
    } else {
        MVSIM_TRY(launch_extract(ctx->stream, con, o->acq, dim, p->inc, !materialise, scal, p->min_value, noise,
                                 mvsim_poisson_mul((double)p->snr), p->seed, p->stream, 0, qws, qm));
    }
    ev_end(ctx, ST_EXTRACT);
    if (overlap_ok) {
        MVSIM_HIP(hipEventRecord(ctx->ev_tail, ctx->tail_stream));
        ctx->tail_pending = true;
        ctx->tail_lo[0] = reinterpret_cast<const char*>(o->acq); ctx->tail_hi[0] = ctx->tail_lo[0] + (size_t)n_out * sizeof(float);
        ctx->tail_lo[1] = reinterpret_cast<const char*>(con);    ctx->tail_hi[1] = ctx->tail_lo[1] + vbytes;
    }
    return MVSIM_OK;
}


// hipGraph replay of a view (option "graph"): host launch cost is what bounds small volumes (130 us to issue the 14
// launches of a 128^3 view against ~0.1 ms of device time).  A captured view bakes in its pointers and by-value
// parameters (affine model of the angle, seed, stream id), so the cache is keyed by ALL of them: the first call with a key
// runs eagerly (workspaces, twiddles), the second is captured and instantiated, later ones replay.  The PSF upload stays
// outside the graph (its pinned staging slot changes from call to call).
static std::string view_graph_key(mvsim_ctx* ctx, const float* gt, const int64_t dim[3], const int64_t kdim[3],
                                  const mvsim_view_params* p, const mvsim_view_outputs* o)
{
    std::string k;
    auto add = [&](const void* ptr, size_t bytes) { k.append(reinterpret_cast<const char*>(ptr), bytes); };
    add(&gt, sizeof(gt)); add(dim, 3 * sizeof(int64_t)); add(kdim, 3 * sizeof(int64_t));
    // field by field: the struct has padding bytes
    add(&p->axis, sizeof(p->axis)); add(&p->degrees, sizeof(p->degrees)); add(&p->delta, sizeof(p->delta));
    add(&p->min_value, sizeof(p->min_value)); add(&p->target_average, sizeof(p->target_average)); add(&p->inc, sizeof(p->inc));
    add(&p->snr, sizeof(p->snr)); add(&p->seed, sizeof(p->seed)); add(&p->stream, sizeof(p->stream));
    add(&p->conv_method, sizeof(p->conv_method));
    add(&o->rot, sizeof(o->rot)); add(&o->att, sizeof(o->att)); add(&o->con, sizeof(o->con)); add(&o->acq, sizeof(o->acq));
    add(&ctx->stream, sizeof(ctx->stream));
    const Options& q = ctx->opt;
    const int oo[14] = {q.zpass, q.rocfft ? 1 : 0, q.fused_rotate, q.poisson_queue, q.early_sum ? 1 : 0, q.fuse_tail ? 1 : 0, q.psf_overlap ? 1 : 0,
                        q.attenuate_scan ? 1 : 0, q.fused_fftx, q.zconv_strided ? 1 : 0, q.skip_empty ? 1 : 0, q.exp, q.poisson_queue_share, ctx->queue_share_learned};
    add(oo, sizeof(oo));
    add(q.fft_pad, sizeof(q.fft_pad));
    return k;
}

static void view_graphs_release(mvsim_ctx* ctx)
{
    for (auto& g : ctx->graphs) {
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
        if (g.graph) (void)hipGraphDestroy(g.graph);
    }
    ctx->graphs.clear();
    ctx->graph_seen.clear();
}

static int view_graph_launch(mvsim_ctx* ctx, const float* gt, const int64_t dim[3], const int64_t kdim[3], const mvsim_view_params* p,
                             const mvsim_view_outputs* o)
{
    const std::string key = view_graph_key(ctx, gt, dim, kdim, p, o);
    ctx->graph_tick += 1;
    const unsigned long long epoch = ctx->alloc_epoch;
    if (ctx->graph_epoch != epoch) {
        // some workspace was (re)allocated since the graphs were captured: their baked-in addresses may be stale
        for (auto& g : ctx->graphs) { (void)hipGraphExecDestroy(g.exec); (void)hipGraphDestroy(g.graph); }
        ctx->graphs.clear();
        ctx->graph_epoch = epoch;
    }
    for (auto& g : ctx->graphs)
        if (g.key == key) {
            g.last_use = ctx->graph_tick;
            MVSIM_HIP(hipGraphLaunch(g.exec, ctx->stream));
            return MVSIM_OK;
        }
    if (!ctx->graph_seen.count(key)) {
        // first sight: run eagerly (this is also what sizes every workspace and table the capture must not allocate)
        if (ctx->graph_seen.size() > 4096) ctx->graph_seen.clear();
        ctx->graph_seen.insert(key);
        return view_enqueue(ctx, gt, dim, kdim, p, o);
    }
    MVSIM_HIP(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeRelaxed));
    const int rc = view_enqueue(ctx, gt, dim, kdim, p, o);
    hipGraph_t graph = nullptr;
    const hipError_t e = hipStreamEndCapture(ctx->stream, &graph);
    if (rc != MVSIM_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
    if (ctx->alloc_epoch != epoch) {
        // an allocation slipped into the capture (a workspace grew): do not keep this graph; run the view eagerly
        if (graph) (void)hipGraphDestroy(graph);
        return view_enqueue(ctx, gt, dim, kdim, p, o);
    }
    if (e != hipSuccess || !graph) { set_error("hipStreamEndCapture failed: %s", hipGetErrorString(e)); return MVSIM_EHIP; }
    hipGraphExec_t exec = nullptr;
    const hipError_t ei = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    if (ei != hipSuccess) { (void)hipGraphDestroy(graph); set_error("hipGraphInstantiate failed: %s", hipGetErrorString(ei)); return MVSIM_EHIP; }
    if (ctx->graphs.size() >= 32) {                       // evict the least recently used
        size_t victim = 0;
        for (size_t i = 1; i < ctx->graphs.size(); ++i) if (ctx->graphs[i].last_use < ctx->graphs[victim].last_use) victim = i;
        (void)hipGraphExecDestroy(ctx->graphs[victim].exec);
        (void)hipGraphDestroy(ctx->graphs[victim].graph);
        ctx->graphs.erase(ctx->graphs.begin() + (long)victim);
    }
    ctx->graphs.push_back(mvsim_ctx::ViewGraph{key, graph, exec, ctx->graph_tick});
    MVSIM_HIP(hipGraphLaunch(exec, ctx->stream));
    return MVSIM_OK;
}

int mvsim_simulate_view_dev(mvsim_ctx* ctx, const float* gt, const int64_t dim[3], float* psf_host,
                            const int64_t kdim[3], const mvsim_view_params* p, const mvsim_view_outputs* o,
                            double* correction)
{
    MVSIM_TRY(set_device(ctx, /*keep_tail=*/true));
    // The previous view's tail may stay in flight beside this view's first stage, and this view's tail behind the call,
    // where nothing but this library can observe the difference: the context's own stream, no host-visible result, no graph.
    const bool overlap = ctx->opt.tail_overlap != 0 && (ctx->opt.tail_overlap == 2 || ctx->stream == ctx->own_stream) &&
                         !ctx->opt.graph && !correction && dim &&
                         dim[0] > 0 && dim[1] > 0 && dim[2] > 0 && dim[0] * dim[1] * dim[2] >= ((int64_t)1 << 24);
    if (!overlap) MVSIM_TRY(join_tail(ctx));
    MVSIM_TRY(check_dim(dim));
    MVSIM_CHECK_ARG(gt && p && o, "null pointer");
    MVSIM_CHECK_ARG(o->acq != nullptr, "outputs.acq is required");
    MVSIM_CHECK_ARG(p->axis >= 0 && p->axis <= 2, "axis must be 0, 1 or 2");
    MVSIM_CHECK_ARG(p->inc >= 1, "inc must be >= 1");
    MVSIM_CHECK_ARG(p->conv_method >= 0 && p->conv_method <= 2, "conv_method must be 0, 1 or 2");
    MVSIM_CHECK_ARG(dim[0] <= dim[1], "attenuate3d: Nx > Ny walks outside the interval in the reference");
    ev_next(ctx);
    MVSIM_TRY(psf_prepare(ctx, psf_host, kdim, dim));
    // a view replays from a graph when asked to, unless stage events are being recorded (they would be captured too) or
    // the convolution would go through rocFFT (library calls inside a capture are not ours to vouch for)
    int64_t P[3];
    const bool capturable = pick_method(p->conv_method, kdim) == 2 || custom_fft_sizes(dim, kdim, P, ctx->opt);
    if (ctx->opt.graph && !ctx->timing && capturable) MVSIM_TRY(view_graph_launch(ctx, gt, dim, kdim, p, o));
    else MVSIM_TRY(view_enqueue(ctx, gt, dim, kdim, p, o, overlap));
    if (correction) {
        double *partial, *scal;
        MVSIM_TRY(scal_ptr(ctx, &partial, &scal));
        MVSIM_HIP(hipMemcpyAsync(correction, scal + 1, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        MVSIM_HIP(hipStreamSynchronize(ctx->stream));
    }
    return MVSIM_OK;
}

// How many views of this size run side by side on LANES -- the form for views that cannot be stacked (an adjusted volume requested,
// another spacing per view, ...).  A view is a chain of ~12 dependent launches that each leave CUs idle below ~2^24 voxels, so two to
// four chains side by side hide part of each other's latency; beyond that the host's launch rate is the bound (~75 us per view), and
// from 289^3 up lanes LOSE (kernels that each fill the CUs run one after the other anyway, and their working sets evict each other).
// Measured (profiles/r05_small_views.txt): 64^3 3.4 -> 4.5 Gvoxel/s and 128^3 16.7 -> 25.0 with four lanes, 256^3 41 -> 49 with two
// (37 with four), 289^3 48 -> 24 with four: DESIGN 4.8.
static int pick_view_lanes(const mvsim_ctx* ctx, const int64_t dim[3], int n_views)
{
    int lanes = ctx->opt.view_lanes;
    if (lanes <= 0) {
        const int64_t n = dim[0] * dim[1] * dim[2];
        lanes = n <= ((int64_t)1 << 22) ? 4 : n <= ((int64_t)1 << 24) ? 2 : 1;
    }
    return std::max(1, std::min(lanes, n_views));
}

// ---- stacked views: ONE launch per stage for all V views ------------------------------------------------------------------
// A view below ~2^26 voxels is a chain of a dozen dependent launches that each leave most of the chip idle (128^3: a few hundred
// blocks per launch and ~10 us per link of the chain; 289^3: the rotate + attenuate kernel is 1 300 waves of serial latency), and the
// host needs ~5 us per launch on top.  The views of `main`'s loop are independent and alike (SimulateMultiViewDataset.java:567-585:
// same volume, same PSF size, same spacing), so their kernels take the view as one more grid dimension: the attenuated volumes, the
// spectra, the PSFs' taps and the convolved planes of the V views lie back to back in the workspaces, the passes that work on planes
// or rows (A, B, D, E, the PSF's own) simply see V times as many, and the kernels with per-view operands -- rotate + attenuate (the
// view's inverse model), the z pass (its taps, its sum), extract + Poisson (its acquisition, RNG key, adjustImage factor) -- read them
// from small device tables.  Every voxel goes through the same arithmetic in the same order as in a single view: bit-identical.
static bool views_batchable(const mvsim_ctx* ctx, const int64_t dim[3], const int64_t kdim[3], const mvsim_view_params* params,
                            const mvsim_view_outputs* outs, int n_views)
{
    if (n_views < 2 || ctx->opt.view_batch == 0 || ctx->opt.graph || ctx->opt.rocfft) return false;
    const int64_t n = nvox(dim);
    if (ctx->opt.view_batch == 2 && n > ((int64_t)1 << 26)) return false;            // auto: views that fill the chip by themselves stay single
    if ((size_t)n * n_views * sizeof(float) > ((size_t)24 << 30)) return false;      // the stacked workspaces are ~6 x this
    const mvsim_view_params& p0 = params[0];
    if (pick_method(p0.conv_method, kdim) != 1 || !custom_fft_batchable(ctx, dim, kdim)) return false;
    const bool noise = p0.snr >= 0.0f;
    if (noise && ctx->opt.poisson_queue != 1) return false;
    for (int v = 0; v < n_views; ++v) {
        const mvsim_view_params& p = params[v];
        if (p.axis != 0 || p.delta != p0.delta || p.inc != p0.inc || p.snr != p0.snr || p.min_value != p0.min_value ||
            p.target_average != p0.target_average || pick_method(p.conv_method, kdim) != 1)
            return false;
        if (outs[v].rot || outs[v].att || outs[v].con) return false;                   // intermediates on request: the single-view path
        double m[12];
        Affine inv;
        axis_rotation_host(dim, 0, p.degrees, m);
        affine_invert_host(m, inv.m);
        if (!(inv.m[0] == 1.0 && inv.m[1] == 0.0 && inv.m[2] == 0.0 && inv.m[3] == 0.0 && inv.m[4] == 0.0 && inv.m[8] == 0.0)) return false;
    }
    return true;
}

static int views_enqueue_batched(mvsim_ctx* ctx, const float* gt, const int64_t dim[3], float* const* psf_host, const int64_t kdim[3],
                                 const mvsim_view_params* params, const mvsim_view_outputs* outs, int V)
{
    const int64_t n = nvox(dim), k3 = kdim[0] * kdim[1] * kdim[2];
    const mvsim_view_params& p0 = params[0];
    const bool noise = p0.snr >= 0.0f;
    const int64_t nzo = mvsim_extract_nz(dim[2], p0.inc), plane_vox = dim[0] * dim[1];
    const int64_t n_out = plane_vox * nzo;
    ev_next(ctx);
    // one upload for everything the views bring: [V inverse models][V extract tables][V normalised PSFs]
    const size_t tab_a = (size_t)V * sizeof(Affine), tab_e = (size_t)V * sizeof(ExtractView);
    const size_t off_e = (tab_a + 255) & ~(size_t)255, off_p = (off_e + tab_e + 255) & ~(size_t)255;
    const size_t up_bytes = off_p + (size_t)V * k3 * sizeof(float);
    MVSIM_TRY(ctx->view_tab.reserve(up_bytes));
    MVSIM_TRY(ctx->vol_b.reserve((size_t)V * n * sizeof(float)));                       // att[v]
    const int zstride = p0.inc > 1 ? p0.inc : 1;                                        // (as view_enqueue: only the planes extractSlices reads)
    const int64_t con_planes = zstride > 1 ? nzo : dim[2];
    MVSIM_TRY(ctx->vol_a.reserve((size_t)V * plane_vox * con_planes * sizeof(float)));  // con[v]
    QueueMode qm;
    MVSIM_TRY(queue_mode_next(ctx, &qm));
    const size_t qbytes = noise ? ((poisson_queue_bytes_planes(plane_vox, nzo, qm.share) + 255) & ~(size_t)255) : 0;
    if (noise) MVSIM_TRY(ctx->pqueue.reserve(qbytes * V));
    double *partial, *scal0;
    MVSIM_TRY(scal_ptr(ctx, &partial, &scal0));
    int slot = 0;
    MVSIM_TRY(ctx->pinned.acquire(up_bytes, &slot));
    char* hp = reinterpret_cast<char*>(ctx->pinned.p[slot]);
    char* dp = ctx->view_tab.as<char>();
    Affine* atab = reinterpret_cast<Affine*>(hp);
    ExtractView* etab = reinterpret_cast<ExtractView*>(hp + off_e);
    float* con = ctx->vol_a.as<float>();
    bool vec_all = true;
    for (int v = 0; v < V; ++v) {
        double m[12];
        axis_rotation_host(dim, 0, params[v].degrees, m);
        affine_invert_host(m, atab[v].m);
        ExtractView& e = etab[v];
        e.in = con + (size_t)v * plane_vox * con_planes;
        e.out = outs[v].acq;
        e.scal = scal_of(ctx, v);
        e.queue = nullptr; e.qcount = nullptr;
        if (noise) poisson_queue_split(ctx->pqueue.as<char>() + (size_t)v * qbytes, &e.queue, &e.qcount);
        e.k0 = (uint32_t)params[v].seed; e.k1 = (uint32_t)(params[v].seed >> 32); e.stream = params[v].stream; e.pad = 0;
        vec_all = vec_all && ((reinterpret_cast<uintptr_t>(e.in) | reinterpret_cast<uintptr_t>(e.out)) % 16 == 0);
    }
    // Tools.normImage of the V PSFs (in place, Q5) and their copy into the upload block: one host thread per view -- a 51^3 stack is 0.13 ms of
    // summation and division, and nothing reaches the GPU before the last of them is done
    //  -- unless two views name the same (or overlapping) PSF memory, e.g. `[psf] * 8`: n sequential calls normalise that buffer n
    // times one after the other, and so does this: in view order on the calling thread, each view taking the taps as they are at its turn
    auto psf_stage = [&](int v) {
        psf_normalise_host(psf_host[v], k3);
        std::memcpy(hp + off_p + (size_t)v * k3 * sizeof(float), psf_host[v], (size_t)k3 * sizeof(float));
    };
    bool psf_aliased = false;
    for (int v = 0; v < V && !psf_aliased; ++v)
        for (int w = v + 1; w < V; ++w) {
            const uintptr_t a = reinterpret_cast<uintptr_t>(psf_host[v]), b = reinterpret_cast<uintptr_t>(psf_host[w]);
            const uintptr_t len = (uintptr_t)k3 * sizeof(float);
            if (a < b + len && b < a + len) { psf_aliased = true; break; }
        }
    if (psf_aliased) for (int v = 0; v < V; ++v) psf_stage(v);
    else HostPool::get().run(V, host_threads_of(ctx), psf_stage);
    MVSIM_HIP(hipMemcpyAsync(dp, hp, up_bytes, hipMemcpyHostToDevice, ctx->stream));
    MVSIM_HIP(hipEventRecord(ctx->pinned.ev[slot], ctx->stream));
    ctx->pinned.busy[slot] = true;

    ev_begin(ctx, ST_ROTATE);
    MVSIM_TRY(launch_rotate_attenuate_views(ctx->stream, gt, ctx->vol_b.as<float>(), dim, reinterpret_cast<const Affine*>(dp), V, p0.delta));
    ev_end(ctx, ST_ROTATE);

    ConvTail tail;
    tail.views = V;
    tail.zstride = zstride;
    tail.corr_n = n; tail.min_value = p0.min_value; tail.target_average = p0.target_average;
    int64_t P[3];
    if (!custom_fft_sizes(dim, kdim, P, ctx->opt)) { set_error("stacked views: no hand-written FFT size"); return MVSIM_EINVAL; }
    MVSIM_TRY(custom_fft_convolve(ctx, ctx->vol_b.as<float>(), dim, reinterpret_cast<const float*>(dp + off_p), kdim, P, con, &tail));
    if (!tail.corr_done || tail.zstride != zstride) { set_error("stacked views: the convolution did not deliver the factors / planes asked for"); return MVSIM_EHIP; }

    ev_begin(ctx, ST_EXTRACT);
    const ExtractView* evt = reinterpret_cast<const ExtractView*>(dp + off_e);
    if (zstride > 1) {
        const int64_t cdim[3] = {dim[0], dim[1], nzo};                                  // `con` holds the acquired planes only
        MVSIM_TRY(launch_extract_views(ctx->stream, cdim, 1, true, p0.min_value, noise, mvsim_poisson_mul((double)p0.snr), qm,
                                       p0.inc, V, evt, vec_all));
    } else {
        MVSIM_TRY(launch_extract_views(ctx->stream, dim, p0.inc, true, p0.min_value, noise, mvsim_poisson_mul((double)p0.snr), qm,
                                       0, V, evt, vec_all));
    }
    ev_end(ctx, ST_EXTRACT);
    return MVSIM_OK;
}

int mvsim_simulate_views_dev(mvsim_ctx* ctx, const float* gt, const int64_t dim[3], float* const* psf_host, const int64_t kdim[3],
                             const mvsim_view_params* params, const mvsim_view_outputs* outs, int n_views)
{
    MVSIM_TRY(set_device(ctx));
    MVSIM_TRY(check_dim(dim));
    MVSIM_CHECK_ARG(gt && psf_host && kdim && params && outs, "null pointer");
    MVSIM_CHECK_ARG(n_views >= 0 && n_views <= MVSIM_MAX_VIEWS, "n_views must be in [0, MVSIM_MAX_VIEWS]");
    MVSIM_CHECK_ARG(!ctx->is_lane, "a lane context cannot fan out itself");
    const int64_t n = nvox(dim);
    const size_t vbytes = (size_t)n * sizeof(float);
    for (int v = 0; v < n_views; ++v) {
        const mvsim_view_params* p = &params[v];
        MVSIM_CHECK_ARG(psf_host[v] != nullptr, "null PSF");
        MVSIM_CHECK_ARG(outs[v].acq != nullptr, "outputs.acq is required");
        MVSIM_CHECK_ARG(p->axis >= 0 && p->axis <= 2, "axis must be 0, 1 or 2");
        MVSIM_CHECK_ARG(p->inc >= 1, "inc must be >= 1");
        MVSIM_CHECK_ARG(p->conv_method >= 0 && p->conv_method <= 2, "conv_method must be 0, 1 or 2");
    }
    MVSIM_CHECK_ARG(dim[0] <= dim[1], "attenuate3d: Nx > Ny walks outside the interval in the reference");
    // views that run side by side must not write what another one reads or writes (sequential calls would order them)
    {
        struct Range { const char* lo; const char* hi; };
        std::vector<Range> w;
        for (int v = 0; v < n_views; ++v) {
            const size_t abytes = (size_t)(dim[0] * dim[1] * mvsim_extract_nz(dim[2], params[v].inc)) * sizeof(float);
            const float* ptr[4] = {outs[v].rot, outs[v].att, outs[v].con, outs[v].acq};
            for (int a = 0; a < 4; ++a)
                if (ptr[a]) w.push_back(Range{reinterpret_cast<const char*>(ptr[a]), reinterpret_cast<const char*>(ptr[a]) + (a == 3 ? abytes : vbytes)});
        }
        bool meet = false;
        for (size_t i = 0; i < w.size() && !meet; ++i) {
            meet = ranges_meet(gt, vbytes, w[i].lo, w[i].hi);
            for (size_t j = i + 1; j < w.size() && !meet; ++j) meet = w[i].lo < w[j].hi && w[j].lo < w[i].hi;
        }
        MVSIM_CHECK_ARG(!meet, "simulate_views: output buffers overlap each other or the ground truth");
    }
    if (views_batchable(ctx, dim, kdim, params, outs, n_views)) {
        const int rc = views_enqueue_batched(ctx, gt, dim, psf_host, kdim, params, outs, n_views);
        ev_rebalance(ctx);
        return rc;
    }
    const int nl = pick_view_lanes(ctx, dim, n_views);
    if (nl <= 1) {
        for (int v = 0; v < n_views; ++v) MVSIM_TRY(mvsim_simulate_view_dev(ctx, gt, dim, psf_host[v], kdim, &params[v], &outs[v], nullptr));
        return MVSIM_OK;
    }
    while ((int)ctx->lanes.size() < nl) {
        // the event first: lanes and lane_done grow together or not at all
        hipEvent_t e = nullptr;
        MVSIM_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        mvsim_ctx* lane = nullptr;
        const int lrc = mvsim_create(ctx->device, &lane);
        if (lrc != MVSIM_OK) { (void)hipEventDestroy(e); return lrc; }
        lane->is_lane = true;
        ctx->lanes.push_back(lane);
        ctx->lane_done.push_back(e);
    }
    if (!ctx->ev_lane_fork) MVSIM_HIP(hipEventCreateWithFlags(&ctx->ev_lane_fork, hipEventDisableTiming));
    // fork: nothing a lane enqueues may overtake what this context's stream holds (the ground truth's producer, say)
    MVSIM_HIP(hipEventRecord(ctx->ev_lane_fork, ctx->stream));
    for (int l = 0; l < nl; ++l) {
        mvsim_ctx* lane = ctx->lanes[(size_t)l];
        lane->opt = ctx->opt;
        lane->opt.tail_overlap = 0; lane->opt.graph = 0; lane->opt.view_lanes = 1;
        MVSIM_HIP(hipStreamWaitEvent(lane->stream, ctx->ev_lane_fork, 0));
    }
    int rc = MVSIM_OK;
    for (int v = 0; v < n_views && rc == MVSIM_OK; ++v) {
        mvsim_ctx* lane = ctx->lanes[(size_t)(v % nl)];
        rc = psf_prepare(lane, psf_host[v], kdim, dim);
        if (rc == MVSIM_OK) rc = view_enqueue(lane, gt, dim, kdim, &params[v], &outs[v], false);
        ev_rebalance(lane);
    }
    // join -- also after a failure: whatever was enqueued is ordered in front of this context's next work
    for (int l = 0; l < nl; ++l) {
        mvsim_ctx* lane = ctx->lanes[(size_t)l];
        if (hipEventRecord(ctx->lane_done[(size_t)l], lane->stream) != hipSuccess ||
            hipStreamWaitEvent(ctx->stream, ctx->lane_done[(size_t)l], 0) != hipSuccess) {
            if (rc == MVSIM_OK) { set_error("simulate_views: joining lane %d failed", l); rc = MVSIM_EHIP; }
        }
    }
    return rc;
}

int mvsim_simulate_iteration_dev(mvsim_ctx* ctx, const float* gt, const int64_t dim[3], float* psf_host, const int64_t kdim[3],
                                 const mvsim_view_params* p, int back_degrees, const mvsim_view_outputs* o,
                                 const mvsim_iteration_outputs* more)
{
    MVSIM_CHECK_ARG(ctx && more, "null pointer");
    MVSIM_TRY(mvsim_simulate_view_dev(ctx, gt, dim, psf_host, kdim, p, o, nullptr));
    MVSIM_TRY(set_device(ctx));                            // the rest reads the acquisition: a pending tail runs first
    Affine inv;
    double m[12];
    if (more->iso || more->view) {
        const int64_t adim[3] = {dim[0], dim[1], mvsim_extract_nz(dim[2], p->inc)};
        const int64_t idim[3] = {dim[0], dim[1], mvsim_isotropic_nz(adim[2], p->inc)};
        float* iso = more->iso;
        if (!iso) { MVSIM_TRY(ctx->vol_c.reserve((size_t)(idim[0] * idim[1] * idim[2]) * sizeof(float))); iso = ctx->vol_c.as<float>(); }
        MVSIM_CHECK_ARG(iso != more->view, "iso and view must be different buffers");
        MVSIM_TRY(launch_make_isotropic(ctx->stream, o->acq, iso, adim, p->inc));
        if (more->view) {
            axis_rotation_host(idim, p->axis, back_degrees, m);
            affine_invert_host(m, inv.m);
            MVSIM_TRY(launch_rotate(ctx->stream, iso, more->view, idim, inv));
        }
    }
    if (more->view_weights) {
        if (ctx->weight_dim[0] != dim[0] || ctx->weight_dim[1] != dim[1] || ctx->weight_dim[2] != dim[2] || !ctx->weight_img.p) {
            MVSIM_TRY(ctx->weight_img.reserve((size_t)nvox(dim) * sizeof(float)));
            MVSIM_TRY(launch_weight_image(ctx->stream, ctx->weight_img.as<float>(), dim));
            for (int d = 0; d < 3; ++d) ctx->weight_dim[d] = dim[d];
        }
        axis_rotation_host(dim, p->axis, back_degrees, m);
        affine_invert_host(m, inv.m);
        MVSIM_TRY(launch_rotate(ctx->stream, ctx->weight_img.as<float>(), more->view_weights, dim, inv));
    }
    if (more->view_psf) {
        // psf_dev holds the normalised PSF this view was convolved with (psf_prepare)
        axis_rotation_host(kdim, p->axis, back_degrees, m);
        affine_invert_host(m, inv.m);
        MVSIM_TRY(launch_rotate(ctx->stream, ctx->psf_dev.as<float>(), more->view_psf, kdim, inv));
    }
    return MVSIM_OK;
}

// ---- z-slab tiling of one view (BASELINE configs[3]/[4]) ------------------------------------------------
int mvsim_slab_range(int64_t nz, int nranks, int rank, int64_t* z0, int64_t* z1)
{
    if (nz < 1 || nranks < 1 || rank < 0 || rank >= nranks || !z0 || !z1) {
        set_error("invalid argument: slab range");
        return MVSIM_EINVAL;
    }
    *z0 = nz * rank / nranks;
    *z1 = nz * (rank + 1) / nranks;
    return MVSIM_OK;
}

// rotate + attenuate the slab's planes and the halo the PSF reaches, convolve the slab; the slab's share of adjustImage's sum is left in
// the context's scalar slot (device).  Nothing here waits for the device.
static int slab_convolve_enqueue(mvsim_ctx* ctx, const float* gt, const int64_t dim[3], float* psf_host,
                                 const int64_t kdim[3], const mvsim_view_params* p, int64_t z0, int64_t z1)
{
    MVSIM_TRY(set_device(ctx));
    MVSIM_TRY(check_dim(dim));
    MVSIM_CHECK_ARG(gt && p, "null pointer");
    MVSIM_CHECK_ARG(p->axis == 0, "slab tiling supports rotation about x (axis 0)");
    MVSIM_CHECK_ARG(dim[0] <= dim[1], "attenuate3d: Nx > Ny walks outside the interval in the reference");
    MVSIM_CHECK_ARG(0 <= z0 && z0 < z1 && z1 <= dim[2], "slab must satisfy 0 <= z0 < z1 <= Nz");
    MVSIM_CHECK_ARG(kdim && kdim[2] >= 1 && kdim[2] <= 64, "slab tiling needs a PSF depth <= 64 (direct z pass)");
    const int64_t nz = dim[2], kz = kdim[2], c = kz / 2, hl = kz - 1 - c;
    // planes the taps of the slab's outputs reach: [z0 - hl, z1 - 1 + c], folded back at the global faces
    int64_t za = z0 - hl, zb = z1 + c;                     // [za, zb)
    if (za < 0) { zb = std::max<int64_t>(zb, std::min<int64_t>(nz, -za + 1)); za = 0; }
    if (zb > nz) { za = std::min<int64_t>(za, std::max<int64_t>(0, 2 * nz - 1 - zb)); zb = nz; }
    if (kz >= nz) { za = 0; zb = nz; }
    int64_t P[3];
    if (!custom_fft_sizes(dim, kdim, P, ctx->opt)) {
        set_error("slab tiling: no hand-written FFT size for this volume / PSF");
        return MVSIM_EINVAL;
    }
    const size_t pbytes = (size_t)dim[0] * dim[1] * sizeof(float);
    MVSIM_TRY(ctx->vol_a.reserve(pbytes * (size_t)(z1 - z0)));
    MVSIM_TRY(psf_prepare(ctx, psf_host, kdim, dim));
    double m[12];
    Affine inv;
    axis_rotation_host(dim, p->axis, p->degrees, m);
    affine_invert_host(m, inv.m);
    // (round 6) the slab's planes through the fused rotate + attenuate + x-transform kernel, as an untiled view's: the attenuated planes
    // never cross HBM (z_first = za, zb - za planes; F holds them from plane 0, where pass B of the slab's convolution expects them)
    bool x_done = false;
    MVSIM_TRY(rotate_attenuate_fftx(ctx, gt, nullptr, nullptr, dim, kdim, inv, p->delta, &x_done, nullptr, (int)za, (int)(zb - za)));
    if (!x_done) {
        MVSIM_TRY(ctx->vol_b.reserve(pbytes * (size_t)(zb - za)));
        bool fused = false;
        MVSIM_TRY(launch_rotate_attenuate_planes(ctx->stream, gt, nullptr, ctx->vol_b.as<float>(), dim, inv, p->delta,
                                                 (int)za, (int)(zb - za), ctx->opt.fused_rotate == 2 ? 2 : 1, &fused));   // planes [za, zb) exist only fused
        if (!fused) {
            set_error("slab tiling needs the fused rotate+attenuate kernel");
            return MVSIM_EINVAL;
        }
    }
    const SlabRange slab{(int)za, (int)(zb - za), (int)z0, (int)(z1 - z0)};
    // a slab that starts at a multiple of the view's spacing convolves along z -- and sends through passes D and E -- only the planes
    // extractSlices reads, like an untiled compact view (the sum over ALL of the slab's planes comes from the z pass's input rows)
    ConvTail tail;
    tail.zstride = (p->inc > 1 && z0 % p->inc == 0) ? p->inc : 1;
    tail.x_done = x_done;
    ctx->slab_z0 = ctx->slab_z1 = -1;
    MVSIM_TRY(custom_fft_convolve_slab(ctx, x_done ? nullptr : ctx->vol_b.as<float>(), dim, ctx->psf_dev.as<float>(), kdim, P, slab,
                                       ctx->vol_a.as<float>(), &tail));
    ctx->slab_z0 = z0; ctx->slab_z1 = z1; ctx->slab_zstride = tail.zstride;
    return MVSIM_OK;
}

int mvsim_view_slab_convolve_dev(mvsim_ctx* ctx, const float* gt, const int64_t dim[3], float* psf_host,
                                 const int64_t kdim[3], const mvsim_view_params* p, int64_t z0, int64_t z1,
                                 double* slab_sum)
{
    MVSIM_CHECK_ARG(ctx != nullptr && slab_sum != nullptr, "null pointer");
    MVSIM_TRY(slab_convolve_enqueue(ctx, gt, dim, psf_host, kdim, p, z0, z1));
    double *partial, *scal;
    MVSIM_TRY(scal_ptr(ctx, &partial, &scal));
    MVSIM_HIP(hipMemcpyAsync(slab_sum, scal, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    MVSIM_HIP(hipStreamSynchronize(ctx->stream));
    return MVSIM_OK;
}

// adjust with the sum the context's scalar slot holds (the view's, once reduced), extract, Poisson: the slab's acquired planes
static int slab_finish_enqueue(mvsim_ctx* ctx, const int64_t dim[3], const mvsim_view_params* p, int64_t z0, int64_t z1, float* acq,
                               int64_t* n_planes);

int mvsim_view_slab_finish_dev(mvsim_ctx* ctx, const int64_t dim[3], const mvsim_view_params* p, int64_t z0,
                               int64_t z1, double total_sum, float* acq, int64_t* n_planes)
{
    MVSIM_TRY(set_device(ctx));
    MVSIM_TRY(check_dim(dim));
    MVSIM_CHECK_ARG(p && acq, "null pointer");
    MVSIM_CHECK_ARG(0 <= z0 && z0 < z1 && z1 <= dim[2], "slab must satisfy 0 <= z0 < z1 <= Nz");
    MVSIM_CHECK_ARG(p->inc >= 1, "inc must be >= 1");
    const int64_t plane = dim[0] * dim[1];
    MVSIM_CHECK_ARG(ctx->slab_z0 == z0 && ctx->slab_z1 == z1 && ctx->vol_a.bytes >= (size_t)(plane * (z1 - z0)) * sizeof(float),
                    "no convolved slab [z0, z1) in this context (mvsim_view_slab_convolve_dev first)");
    MVSIM_CHECK_ARG(ctx->slab_zstride == 1 || ctx->slab_zstride == p->inc, "the slab was convolved for another spacing");
    double *partial, *scal;
    MVSIM_TRY(scal_ptr(ctx, &partial, &scal));
    MVSIM_HIP(hipMemcpyAsync(scal, &total_sum, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    MVSIM_HIP(hipStreamSynchronize(ctx->stream));          // total_sum lives on the caller's stack
    return slab_finish_enqueue(ctx, dim, p, z0, z1, acq, n_planes);
}

static int slab_finish_enqueue(mvsim_ctx* ctx, const int64_t dim[3], const mvsim_view_params* p, int64_t z0, int64_t z1, float* acq,
                               int64_t* n_planes)
{
    const int64_t plane = dim[0] * dim[1];
    double *partial, *scal;
    MVSIM_TRY(scal_ptr(ctx, &partial, &scal));
    MVSIM_TRY(launch_adjust_corr(ctx->stream, scal, nvox(dim), p->min_value, p->target_average));
    const int64_t k0 = (z0 + p->inc - 1) / p->inc, k1 = (z1 + p->inc - 1) / p->inc;     // acquired planes k: z0 <= k*inc < z1
    if (n_planes) *n_planes = k1 - k0;
    if (k1 <= k0) return MVSIM_OK;
    QueueMode qm;
    MVSIM_TRY(queue_mode_next(ctx, &qm));
    if (ctx->slab_zstride > 1) {
        // compact slab: vol_a holds the planes z0 + k * inc alone, in order (z0 is a multiple of inc: plane k0 * inc = z0)
        const int64_t cdim[3] = {dim[0], dim[1], k1 - k0};
        const bool noise_c = p->snr >= 0.0f;
        void* q = nullptr;
        if (noise_c) { MVSIM_TRY(ctx->pqueue.reserve(poisson_queue_bytes_planes(plane, k1 - k0, qm.share))); q = ctx->pqueue.p; }
        return launch_extract(ctx->stream, ctx->vol_a.as<float>(), acq, cdim, 1, true, scal, p->min_value, noise_c, mvsim_poisson_mul((double)p->snr),
                              p->seed, p->stream, (uint64_t)(z0 * plane), q, qm, p->inc);
    }
    const int64_t first = k0 * p->inc;                      // global index of the first acquired source plane
    const int64_t ldim[3] = {dim[0], dim[1], z1 - first};
    const bool noise = p->snr >= 0.0f;
    void* qws = nullptr;
    if (noise) {
        MVSIM_TRY(ctx->pqueue.reserve(poisson_queue_bytes_planes(plane, k1 - k0, qm.share)));
        qws = ctx->pqueue.p;
    }
    return launch_extract(ctx->stream, ctx->vol_a.as<float>() + plane * (first - z0), acq, ldim, p->inc, true, scal,
                          p->min_value, noise, mvsim_poisson_mul((double)p->snr), p->seed, p->stream,
                          (uint64_t)(first * plane), qws, qm);
}

// One tiled view's slab in ONE call, nothing through the host (round 6): the slab's share of adjustImage's sum stays on the device, is
// reduced over the ranks in place -- on THIS context's stream, by the communicator of `comm_ctx` (null: this context's own; a context
// without one, or a job of one rank, reduces nothing) -- and the adjusted, extracted, sampled planes follow behind it.  Asynchronous.
int mvsim_view_slab_dev(mvsim_ctx* ctx, mvsim_ctx* comm_ctx, const float* gt, const int64_t dim[3], float* psf_host,
                        const int64_t kdim[3], const mvsim_view_params* p, int64_t z0, int64_t z1, float* acq, int64_t* n_planes)
{
    MVSIM_CHECK_ARG(ctx != nullptr && p != nullptr && acq != nullptr, "null pointer");
    MVSIM_CHECK_ARG(p->inc >= 1, "inc must be >= 1");
    MVSIM_TRY(slab_convolve_enqueue(ctx, gt, dim, psf_host, kdim, p, z0, z1));
    double *partial, *scal;
    MVSIM_TRY(scal_ptr(ctx, &partial, &scal));
    mvsim_ctx* cc = comm_ctx ? comm_ctx : ctx;
    MVSIM_CHECK_ARG(cc->device == ctx->device, "the communicator's context lives on another device");
    MVSIM_TRY(comm_allreduce_f64_on_stream(cc, scal, ctx->stream));
    return slab_finish_enqueue(ctx, dim, p, z0, z1, acq, n_planes);
}

// ---- host-buffer entry points (JNI boundary) ---------------------------------------------------------
static int up(mvsim_ctx* ctx, DevBuf& b, const float* h, size_t bytes)
{
    MVSIM_TRY(b.reserve(bytes));
    MVSIM_HIP(hipMemcpyAsync(b.p, h, bytes, hipMemcpyHostToDevice, ctx->stream));
    return MVSIM_OK;
}
static int down(mvsim_ctx* ctx, float* h, const void* d, size_t bytes)
{
    MVSIM_HIP(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, ctx->stream));
    MVSIM_HIP(hipStreamSynchronize(ctx->stream));
    return MVSIM_OK;
}

// Download of an acquisition that holds Poisson COUNTS (sampled: snr >= 0; Tools.java:84 stores them as floats): packed to uint16 on
// the device, half the bytes over PCIe, widened into the caller's buffer by the host threads -- or float32 after all when a value does
// not survive the round trip (the device says so).  Synchronous, like down().  Small outputs are not worth the extra launch.
static int down_counts(mvsim_ctx* ctx, float* h, const float* d, int64_t n, bool sampled)
{
    if (!sampled || ctx->opt.acq_u16 == 0 || n < ((int64_t)1 << 20) || (reinterpret_cast<uintptr_t>(d) & 15) != 0)
        return down(ctx, h, d, (size_t)n * sizeof(float));
    const size_t body = ((size_t)n * sizeof(unsigned short) + 255) & ~(size_t)255;
    MVSIM_TRY(ctx->sync_u16.reserve(body + 256));
    if (ctx->sync_u16_host_bytes < body + 256) {
        if (ctx->sync_u16_host) { (void)hipHostFree(ctx->sync_u16_host); ctx->sync_u16_host = nullptr; ctx->sync_u16_host_bytes = 0; }
        MVSIM_HIP(hipHostMalloc(&ctx->sync_u16_host, body + 256, hipHostMallocDefault));
        ctx->sync_u16_host_bytes = body + 256;
    }
    unsigned int* flag = reinterpret_cast<unsigned int*>(ctx->sync_u16.as<char>() + body);
    MVSIM_HIP(hipMemsetAsync(flag, 0, sizeof(unsigned int), ctx->stream));
    MVSIM_TRY(launch_pack_u16(ctx->stream, d, ctx->sync_u16.as<unsigned short>(), n, flag));
    MVSIM_HIP(hipMemcpyAsync(ctx->sync_u16_host, ctx->sync_u16.p, body + sizeof(unsigned int), hipMemcpyDeviceToHost, ctx->stream));
    MVSIM_HIP(hipStreamSynchronize(ctx->stream));
    ctx->u16_views += 1;
    if (*reinterpret_cast<const unsigned int*>(reinterpret_cast<const char*>(ctx->sync_u16_host) + body) != 0u) {
        ctx->u16_fallbacks += 1;
        return down(ctx, h, d, (size_t)n * sizeof(float));
    }
    const unsigned short* src = reinterpret_cast<const unsigned short*>(ctx->sync_u16_host);
    const long long chunk = (long long)1 << 20;
    HostPool::get().run((int)((n + chunk - 1) / chunk), host_threads_of(ctx), [&](int c) {
        const long long a = (long long)c * chunk, b = std::min<long long>(n, a + chunk);
        widen_u16(src + a, h + a, b - a);
    });
    return MVSIM_OK;
}

int mvsim_rotate_around_axis(mvsim_ctx* ctx, const float* in, const int64_t dim[3], int axis, int degrees, float* out)
{
    MVSIM_TRY(set_device(ctx));
    MVSIM_TRY(check_dim(dim));
    MVSIM_CHECK_ARG(in && out, "null buffer");
    const size_t bytes = (size_t)nvox(dim) * sizeof(float);
    MVSIM_TRY(up(ctx, ctx->vol_a, in, bytes));
    MVSIM_TRY(ctx->vol_b.reserve(bytes));
    MVSIM_TRY(mvsim_rotate_around_axis_dev(ctx, ctx->vol_a.as<float>(), dim, axis, degrees, ctx->vol_b.as<float>()));
    return down(ctx, out, ctx->vol_b.p, bytes);
}

int mvsim_attenuate3d(mvsim_ctx* ctx, const float* in, const int64_t dim[3], double delta, float* out)
{
    MVSIM_TRY(set_device(ctx));
    MVSIM_TRY(check_dim(dim));
    MVSIM_CHECK_ARG(in && out, "null buffer");
    const size_t bytes = (size_t)nvox(dim) * sizeof(float);
    MVSIM_TRY(up(ctx, ctx->vol_a, in, bytes));
    MVSIM_TRY(ctx->vol_b.reserve(bytes));
    MVSIM_TRY(mvsim_attenuate3d_dev(ctx, ctx->vol_a.as<float>(), dim, delta, ctx->vol_b.as<float>()));
    return down(ctx, out, ctx->vol_b.p, bytes);
}

int mvsim_norm_image(mvsim_ctx* ctx, float* img, int64_t n)
{
    MVSIM_TRY(set_device(ctx));
    MVSIM_CHECK_ARG(img && n >= 1, "null buffer or empty image");
    const size_t bytes = (size_t)n * sizeof(float);
    double *partial, *scal;
    MVSIM_TRY(scal_ptr(ctx, &partial, &scal));
    MVSIM_TRY(up(ctx, ctx->vol_a, img, bytes));
    MVSIM_TRY(launch_sum(ctx->stream, ctx->vol_a.as<float>(), n, partial, scal));
    MVSIM_TRY(launch_norm_apply(ctx->stream, ctx->vol_a.as<float>(), n, scal));
    return down(ctx, img, ctx->vol_a.p, bytes);
}

int mvsim_convolve(mvsim_ctx* ctx, const float* img, const int64_t dim[3], float* psf, const int64_t kdim[3],
                   int method, float* out)
{
    MVSIM_TRY(set_device(ctx));
    MVSIM_TRY(check_dim(dim));
    MVSIM_CHECK_ARG(img && out, "null buffer");
    const size_t bytes = (size_t)nvox(dim) * sizeof(float);
    MVSIM_TRY(up(ctx, ctx->vol_a, img, bytes));
    MVSIM_TRY(ctx->vol_b.reserve(bytes));
    MVSIM_TRY(mvsim_convolve_dev(ctx, ctx->vol_a.as<float>(), dim, psf, kdim, method, ctx->vol_b.as<float>()));
    return down(ctx, out, ctx->vol_b.p, bytes);
}

int mvsim_adjust_image(mvsim_ctx* ctx, float* img, int64_t n, float min_value, float target_average, double* correction)
{
    MVSIM_TRY(set_device(ctx));
    MVSIM_CHECK_ARG(img && n >= 1, "null buffer or empty image");
    const size_t bytes = (size_t)n * sizeof(float);
    MVSIM_TRY(up(ctx, ctx->vol_a, img, bytes));
    MVSIM_TRY(mvsim_adjust_image_dev(ctx, ctx->vol_a.as<float>(), n, min_value, target_average, correction));
    return down(ctx, img, ctx->vol_a.p, bytes);
}

int mvsim_extract_slices(mvsim_ctx* ctx, const float* in, const int64_t dim[3], int inc, float snr, uint64_t seed,
                         uint32_t stream, float* out)
{
    MVSIM_TRY(set_device(ctx));
    MVSIM_TRY(check_dim(dim));
    MVSIM_CHECK_ARG(in && out, "null buffer");
    MVSIM_CHECK_ARG(inc >= 1, "inc must be >= 1");
    const size_t bytes = (size_t)nvox(dim) * sizeof(float);
    const size_t obytes = (size_t)(dim[0] * dim[1] * mvsim_extract_nz(dim[2], inc)) * sizeof(float);
    MVSIM_TRY(up(ctx, ctx->vol_a, in, bytes));
    MVSIM_TRY(ctx->out_buf.reserve(obytes));
    MVSIM_TRY(mvsim_extract_slices_dev(ctx, ctx->vol_a.as<float>(), dim, inc, snr, seed, stream, ctx->out_buf.as<float>()));
    return down_counts(ctx, out, ctx->out_buf.as<float>(), (int64_t)(obytes / sizeof(float)), snr >= 0.0f);
}

int mvsim_poisson_process(mvsim_ctx* ctx, float* img, int64_t n, double snr, uint64_t seed, uint32_t stream,
                          uint64_t index_offset)
{
    MVSIM_TRY(set_device(ctx));
    MVSIM_CHECK_ARG(img && n >= 1, "null buffer or empty image");
    const size_t bytes = (size_t)n * sizeof(float);
    MVSIM_TRY(up(ctx, ctx->vol_a, img, bytes));
    MVSIM_TRY(ctx->out_buf.reserve(bytes));
    const int64_t dim[3] = {n, 1, 1};
    QueueMode qm;
    MVSIM_TRY(queue_mode_next(ctx, &qm));
    MVSIM_TRY(ctx->pqueue.reserve(poisson_queue_bytes_planes(n, 1, qm.share)));
    ev_begin(ctx, ST_EXTRACT);
    MVSIM_TRY(launch_extract(ctx->stream, ctx->vol_a.as<float>(), ctx->out_buf.as<float>(), dim, 1, false, nullptr,
                             0.0f, true, mvsim_poisson_mul(snr), seed, stream, index_offset, ctx->pqueue.p, qm));
    ev_end(ctx, ST_EXTRACT);
    return down_counts(ctx, img, ctx->out_buf.as<float>(), n, true);
}

int mvsim_draw_spheres_dev(mvsim_ctx* ctx, float* img, const int64_t dim[3], double min_value, double max_value,
                           int scale, int half_pixel_offset, uint64_t* rnd_state, int64_t* n_spheres)
{
    MVSIM_TRY(set_device(ctx));
    MVSIM_TRY(check_dim(dim));
    MVSIM_CHECK_ARG(img && rnd_state, "null pointer");
    MVSIM_CHECK_ARG(scale >= 1 && scale <= 64, "scale must be in 1..64");
    return draw_spheres_dev(ctx, img, dim, min_value, max_value, scale, half_pixel_offset, rnd_state, n_spheres);
}

int mvsim_downsample2x_dev(mvsim_ctx* ctx, const float* in, const int64_t dim[3], float* out)
{
    MVSIM_TRY(set_device(ctx));
    MVSIM_TRY(check_dim(dim));
    MVSIM_CHECK_ARG(in && out && in != out, "null or aliased buffer");
    MVSIM_CHECK_ARG(dim[0] >= 4 && dim[1] >= 4 && dim[2] >= 4, "downSample2x needs at least 4 samples per dimension");
    return launch_downsample2x(ctx->stream, in, dim, out);
}

int mvsim_draw_spheres(mvsim_ctx* ctx, float* img, const int64_t dim[3], double min_value, double max_value,
                       int scale, int half_pixel_offset, uint64_t* rnd_state, int64_t* n_spheres)
{
    MVSIM_TRY(set_device(ctx));
    MVSIM_TRY(check_dim(dim));
    MVSIM_CHECK_ARG(img && rnd_state, "null pointer");
    const size_t bytes = (size_t)nvox(dim) * sizeof(float);
    MVSIM_TRY(up(ctx, ctx->vol_a, img, bytes));
    MVSIM_TRY(mvsim_draw_spheres_dev(ctx, ctx->vol_a.as<float>(), dim, min_value, max_value, scale, half_pixel_offset,
                                     rnd_state, n_spheres));
    return down(ctx, img, ctx->vol_a.p, bytes);
}

int mvsim_downsample2x(mvsim_ctx* ctx, const float* in, const int64_t dim[3], float* out)
{
    MVSIM_TRY(set_device(ctx));
    MVSIM_TRY(check_dim(dim));
    MVSIM_CHECK_ARG(in && out, "null buffer");
    MVSIM_CHECK_ARG(dim[0] >= 4 && dim[1] >= 4 && dim[2] >= 4, "downSample2x needs at least 4 samples per dimension");
    const size_t bytes = (size_t)nvox(dim) * sizeof(float);
    const size_t obytes = (size_t)(dim[0] / 2 - 1) * (size_t)(dim[1] / 2 - 1) * (size_t)(dim[2] / 2 - 1) * sizeof(float);
    MVSIM_TRY(up(ctx, ctx->vol_a, in, bytes));
    MVSIM_TRY(ctx->out_buf.reserve(obytes));
    MVSIM_TRY(mvsim_downsample2x_dev(ctx, ctx->vol_a.as<float>(), dim, ctx->out_buf.as<float>()));
    return down(ctx, out, ctx->out_buf.p, obytes);
}

int mvsim_make_isotropic(mvsim_ctx* ctx, const float* in, const int64_t dim[3], int inc, float* out)
{
    MVSIM_TRY(set_device(ctx));
    MVSIM_TRY(check_dim(dim));
    MVSIM_CHECK_ARG(in && out, "null buffer");
    MVSIM_CHECK_ARG(inc >= 1, "inc must be >= 1");
    const size_t bytes = (size_t)nvox(dim) * sizeof(float);
    const size_t obytes = (size_t)(dim[0] * dim[1] * mvsim_isotropic_nz(dim[2], inc)) * sizeof(float);
    MVSIM_TRY(up(ctx, ctx->vol_a, in, bytes));
    MVSIM_TRY(ctx->out_buf.reserve(obytes));
    MVSIM_TRY(mvsim_make_isotropic_dev(ctx, ctx->vol_a.as<float>(), dim, inc, ctx->out_buf.as<float>()));
    return down(ctx, out, ctx->out_buf.p, obytes);
}

int mvsim_compute_weight_image(mvsim_ctx* ctx, const int64_t dim[3], float* out)
{
    MVSIM_TRY(set_device(ctx));
    MVSIM_TRY(check_dim(dim));
    MVSIM_CHECK_ARG(out, "null buffer");
    const size_t bytes = (size_t)nvox(dim) * sizeof(float);
    MVSIM_TRY(ctx->vol_a.reserve(bytes));
    MVSIM_TRY(mvsim_compute_weight_image_dev(ctx, dim, ctx->vol_a.as<float>()));
    return down(ctx, out, ctx->vol_a.p, bytes);
}

int mvsim_simulate_view(mvsim_ctx* ctx, const float* gt, const int64_t dim[3], float* psf_host, const int64_t kdim[3],
                        const mvsim_view_params* p, const mvsim_view_outputs* o, double* correction)
{
    MVSIM_TRY(set_device(ctx));
    MVSIM_TRY(check_dim(dim));
    MVSIM_CHECK_ARG(gt && p && o && o->acq, "null pointer (outputs.acq is required)");
    MVSIM_CHECK_ARG(p->inc >= 1, "inc must be >= 1");
    const int64_t n = nvox(dim);
    const size_t vbytes = (size_t)n * sizeof(float);
    const size_t obytes = (size_t)(dim[0] * dim[1] * mvsim_extract_nz(dim[2], p->inc)) * sizeof(float);
    // ground truth goes to its own buffer; requested intermediates get device twins (kept by the context: a driver
    // that calls this once per view must not pay a 0.5 GB hipMalloc/hipFree pair per buffer and call)
    DevBuf &gt_d = ctx->host_gt, &rot_d = ctx->host_rot, &att_d = ctx->host_att, &con_d = ctx->host_con;
    int rc = up(ctx, gt_d, gt, vbytes);
    mvsim_view_outputs dev = {nullptr, nullptr, nullptr, nullptr};
    if (rc == MVSIM_OK && o->rot) { rc = rot_d.reserve(vbytes); dev.rot = rot_d.as<float>(); }
    if (rc == MVSIM_OK && o->att) { rc = att_d.reserve(vbytes); dev.att = att_d.as<float>(); }
    if (rc == MVSIM_OK && o->con) { rc = con_d.reserve(vbytes); dev.con = con_d.as<float>(); }
    if (rc == MVSIM_OK) rc = ctx->out_buf.reserve(obytes);
    dev.acq = ctx->out_buf.as<float>();
    if (rc == MVSIM_OK) rc = mvsim_simulate_view_dev(ctx, gt_d.as<float>(), dim, psf_host, kdim, p, &dev, correction);
    if (rc == MVSIM_OK) rc = join_tail(ctx);                // the copies below read what the tail writes
    if (rc == MVSIM_OK && o->rot) rc = down(ctx, o->rot, dev.rot, vbytes);
    if (rc == MVSIM_OK && o->att) rc = down(ctx, o->att, dev.att, vbytes);
    if (rc == MVSIM_OK && o->con) rc = down(ctx, o->con, dev.con, vbytes);
    if (rc == MVSIM_OK) rc = down_counts(ctx, o->acq, dev.acq, (int64_t)(obytes / sizeof(float)), p->snr >= 0.0f);
    (void)hipStreamSynchronize(ctx->stream);
    return rc;
}

int mvsim_splat_spheres_dev(mvsim_ctx* ctx, float* img, const int64_t dim[3], const mvsim_sphere* spheres, int64_t n)
{
    MVSIM_TRY(set_device(ctx));
    MVSIM_TRY(check_dim(dim));
    MVSIM_CHECK_ARG(img && (spheres || n == 0) && n >= 0, "null pointer or negative count");
    return splat_spheres_dev(ctx, img, dim, spheres, n);
}

int mvsim_splat_spheres(mvsim_ctx* ctx, float* img, const int64_t dim[3], const mvsim_sphere* spheres, int64_t n)
{
    MVSIM_TRY(set_device(ctx));
    MVSIM_TRY(check_dim(dim));
    MVSIM_CHECK_ARG(img && (spheres || n == 0) && n >= 0, "null pointer or negative count");
    const size_t bytes = (size_t)nvox(dim) * sizeof(float);
    MVSIM_TRY(up(ctx, ctx->vol_a, img, bytes));
    MVSIM_TRY(splat_spheres_dev(ctx, ctx->vol_a.as<float>(), dim, spheres, n));
    return down(ctx, img, ctx->vol_a.p, bytes);
}

// the slot's view has landed (ev_d2h synchronised): 16-bit counts become the caller's float32 acquisition -- or, when the device
// flagged a value that does not fit (or is no integer), the float32 buffer is fetched after all
static int async_land(mvsim_ctx* ctx, int s)
{
    if (!ctx->async_as_u16[s]) return MVSIM_OK;
    ctx->async_as_u16[s] = false;
    const long long n = ctx->async_out_n[s];
    const size_t body = ((size_t)n * sizeof(unsigned short) + 255) & ~(size_t)255;
    const unsigned int flag = *reinterpret_cast<const unsigned int*>(reinterpret_cast<const char*>(ctx->async_u16_host[s]) + body);
    ctx->u16_views += 1;
    if (flag != 0u) {
        ctx->u16_fallbacks += 1;
        MVSIM_HIP(hipMemcpy(ctx->async_out_acq[s], ctx->async_acq[s].p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
        return MVSIM_OK;
    }
    const unsigned short* src = reinterpret_cast<const unsigned short*>(ctx->async_u16_host[s]);
    float* dst = ctx->async_out_acq[s];
    const long long chunk = (long long)1 << 20;                   // 1 Mi values: 2 MB in, 4 MB out
    const int chunks = (int)((n + chunk - 1) / chunk);
    HostPool::get().run(chunks, host_threads_of(ctx), [&](int c) {
        const long long a = (long long)c * chunk, b = std::min(n, a + chunk);
        widen_u16(src + a, dst + a, b - a);
    });
    return MVSIM_OK;
}

// ---- pipelined host-buffer views ----------------------------------------------------------------------
static int async_setup(mvsim_ctx* ctx)
{
    if (ctx->async_ready) return MVSIM_OK;
    MVSIM_HIP(hipStreamCreateWithFlags(&ctx->h2d_stream, hipStreamNonBlocking));
    MVSIM_HIP(hipStreamCreateWithFlags(&ctx->d2h_stream, hipStreamNonBlocking));
    for (int s = 0; s < mvsim_ctx::ASYNC_SLOTS; ++s) {
        MVSIM_HIP(hipEventCreateWithFlags(&ctx->ev_h2d[s], hipEventDisableTiming));
        MVSIM_HIP(hipEventCreateWithFlags(&ctx->ev_compute[s], hipEventDisableTiming));
        MVSIM_HIP(hipEventCreateWithFlags(&ctx->ev_d2h[s], hipEventDisableTiming));
    }
    MVSIM_HIP(hipHostMalloc(reinterpret_cast<void**>(&ctx->async_corr), mvsim_ctx::ASYNC_SLOTS * sizeof(double), hipHostMallocDefault));
    ctx->async_ready = true;
    return MVSIM_OK;
}

static void async_release(mvsim_ctx* ctx)
{
    if (!ctx->async_ready) return;
    (void)hipStreamSynchronize(ctx->h2d_stream);
    (void)hipStreamSynchronize(ctx->d2h_stream);
    for (int s = 0; s < mvsim_ctx::ASYNC_SLOTS; ++s)          // views nobody waited for still owe their caller the widened counts
        if (ctx->async_inflight[s]) (void)async_land(ctx, s);
    for (int s = 0; s < mvsim_ctx::ASYNC_SLOTS; ++s) {
        (void)hipEventDestroy(ctx->ev_h2d[s]); (void)hipEventDestroy(ctx->ev_compute[s]); (void)hipEventDestroy(ctx->ev_d2h[s]);
        ctx->async_gt[s].release(); ctx->async_acq[s].release(); ctx->async_u16[s].release();
        if (ctx->async_u16_host[s]) { (void)hipHostFree(ctx->async_u16_host[s]); ctx->async_u16_host[s] = nullptr; ctx->async_u16_host_bytes[s] = 0; }
        ctx->async_inflight[s] = false; ctx->async_gt_src[s] = nullptr; ctx->async_as_u16[s] = false;
    }
    (void)hipStreamDestroy(ctx->h2d_stream);
    (void)hipStreamDestroy(ctx->d2h_stream);
    (void)hipHostFree(ctx->async_corr);
    ctx->async_corr = nullptr;
    ctx->async_ready = false;
}

int mvsim_simulate_view_async(mvsim_ctx* ctx, const float* gt, uint64_t gt_generation, const int64_t dim[3], float* psf_host,
                              const int64_t kdim[3], const mvsim_view_params* p, const mvsim_view_outputs* o, int64_t* ticket)
{
    MVSIM_TRY(set_device(ctx));
    MVSIM_TRY(check_dim(dim));
    MVSIM_CHECK_ARG(gt && p && o && o->acq && ticket, "null pointer (outputs.acq and ticket are required)");
    MVSIM_CHECK_ARG(p->inc >= 1, "inc must be >= 1");
    MVSIM_TRY(async_setup(ctx));
    const int64_t n = nvox(dim);
    const size_t vbytes = (size_t)n * sizeof(float);
    const size_t obytes = (size_t)(dim[0] * dim[1] * mvsim_extract_nz(dim[2], p->inc)) * sizeof(float);
    const long long k = ctx->async_next;
    const int s = (int)(k % mvsim_ctx::ASYNC_SLOTS);
    // the staging set is free once its previous view has landed on the host
    if (ctx->async_inflight[s]) {
        MVSIM_HIP(hipEventSynchronize(ctx->ev_d2h[s]));
        ctx->async_inflight[s] = false;
        ctx->async_corr_done[ctx->async_ticket[s] % mvsim_ctx::ASYNC_HISTORY] = ctx->async_corr[s];
        MVSIM_TRY(async_land(ctx, s));
    }
    const bool wants_twins = o->rot || o->att || o->con;
    // counts as uint16 over PCIe: sampled views only (a view without noise holds reals), 16-byte rows for the packer
    const long long n_out = (long long)(obytes / sizeof(float));
    const bool as_u16 = ctx->opt.acq_u16 != 0 && p->snr >= 0.0f;
    const size_t u16_body = ((size_t)n_out * sizeof(unsigned short) + 255) & ~(size_t)255;
    if (as_u16) {
        MVSIM_TRY(ctx->async_u16[s].reserve(u16_body + 256));
        if (ctx->async_u16_host_bytes[s] < u16_body + 256) {
            if (ctx->async_u16_host[s]) { (void)hipHostFree(ctx->async_u16_host[s]); ctx->async_u16_host[s] = nullptr; ctx->async_u16_host_bytes[s] = 0; }
            MVSIM_HIP(hipHostMalloc(&ctx->async_u16_host[s], u16_body + 256, hipHostMallocDefault));
            ctx->async_u16_host_bytes[s] = u16_body + 256;
        }
    }
    mvsim_view_outputs dev = {nullptr, nullptr, nullptr, nullptr};
    {
        // the staging buffer may move when a later view is larger: what it held is gone then, whatever the caller's pointer
        // and generation say; a view of another size never matches the cached upload either
        const void* before = ctx->async_gt[s].p;
        MVSIM_TRY(ctx->async_gt[s].reserve(vbytes));
        if (ctx->async_gt[s].p != before || ctx->async_gt_bytes[s] != vbytes) ctx->async_gt_src[s] = nullptr;
    }
    MVSIM_TRY(ctx->async_acq[s].reserve(obytes));
    if (o->rot) { MVSIM_TRY(ctx->host_rot.reserve(vbytes)); dev.rot = ctx->host_rot.as<float>(); }
    if (o->att) { MVSIM_TRY(ctx->host_att.reserve(vbytes)); dev.att = ctx->host_att.as<float>(); }
    if (o->con) { MVSIM_TRY(ctx->host_con.reserve(vbytes)); dev.con = ctx->host_con.as<float>(); }
    dev.acq = ctx->async_acq[s].as<float>();
    // upload: only when this staging set does not hold this ground truth already; the copy must not overtake the
    // view that last read the set (ev_compute)
    if (ctx->async_gt_src[s] != gt || ctx->async_gt_gen[s] != gt_generation) {
        if (k >= mvsim_ctx::ASYNC_SLOTS) MVSIM_HIP(hipStreamWaitEvent(ctx->h2d_stream, ctx->ev_compute[s], 0));
        MVSIM_HIP(hipMemcpyAsync(ctx->async_gt[s].p, gt, vbytes, hipMemcpyHostToDevice, ctx->h2d_stream));
        MVSIM_HIP(hipEventRecord(ctx->ev_h2d[s], ctx->h2d_stream));
        MVSIM_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_h2d[s], 0));
        ctx->async_gt_src[s] = gt;
        ctx->async_gt_gen[s] = gt_generation;
        ctx->async_gt_bytes[s] = vbytes;
    }
    // compute: the acquisition buffer of the set is free (waited for above); the single-buffered intermediates are free
    // once the neighbour's downloads are through
    if (wants_twins || ctx->async_twins_busy) {
        for (int q = 0; q < mvsim_ctx::ASYNC_SLOTS; ++q)
            if (ctx->async_inflight[q]) MVSIM_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_d2h[q], 0));
    }
    int rc = mvsim_simulate_view_dev(ctx, ctx->async_gt[s].as<float>(), dim, psf_host, kdim, p, &dev, nullptr);
    if (rc == MVSIM_OK) rc = join_tail(ctx);                // the copies below read what the tail writes
    if (rc != MVSIM_OK) { ctx->async_gt_src[s] = nullptr; return rc; }
    {
        double *partial, *scal;
        MVSIM_TRY(scal_ptr(ctx, &partial, &scal));
        MVSIM_HIP(hipMemcpyAsync(&ctx->async_corr[s], scal + 1, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    }
    if (as_u16) {
        unsigned int* flag = reinterpret_cast<unsigned int*>(ctx->async_u16[s].as<char>() + u16_body);
        MVSIM_HIP(hipMemsetAsync(flag, 0, sizeof(unsigned int), ctx->stream));
        MVSIM_TRY(launch_pack_u16(ctx->stream, dev.acq, ctx->async_u16[s].as<unsigned short>(), n_out, flag));
    }
    MVSIM_HIP(hipEventRecord(ctx->ev_compute[s], ctx->stream));
    // download
    MVSIM_HIP(hipStreamWaitEvent(ctx->d2h_stream, ctx->ev_compute[s], 0));
    if (o->rot) MVSIM_HIP(hipMemcpyAsync(o->rot, dev.rot, vbytes, hipMemcpyDeviceToHost, ctx->d2h_stream));
    if (o->att) MVSIM_HIP(hipMemcpyAsync(o->att, dev.att, vbytes, hipMemcpyDeviceToHost, ctx->d2h_stream));
    if (o->con) MVSIM_HIP(hipMemcpyAsync(o->con, dev.con, vbytes, hipMemcpyDeviceToHost, ctx->d2h_stream));
    if (as_u16) MVSIM_HIP(hipMemcpyAsync(ctx->async_u16_host[s], ctx->async_u16[s].p, u16_body + sizeof(unsigned int), hipMemcpyDeviceToHost, ctx->d2h_stream));
    else MVSIM_HIP(hipMemcpyAsync(o->acq, dev.acq, obytes, hipMemcpyDeviceToHost, ctx->d2h_stream));
    MVSIM_HIP(hipEventRecord(ctx->ev_d2h[s], ctx->d2h_stream));
    ctx->async_as_u16[s] = as_u16;
    ctx->async_out_acq[s] = o->acq;
    ctx->async_out_n[s] = n_out;
    ctx->async_inflight[s] = true;
    ctx->async_ticket[s] = k;
    ctx->async_twins_busy = wants_twins;
    ctx->async_next = k + 1;
    *ticket = (int64_t)k;
    return MVSIM_OK;
}

int mvsim_wait(mvsim_ctx* ctx, int64_t ticket, double* correction)
{
    MVSIM_TRY(set_device(ctx));
    MVSIM_CHECK_ARG(ctx->async_ready && ticket >= 0 && ticket < ctx->async_next, "no such ticket");
    MVSIM_CHECK_ARG(ticket + mvsim_ctx::ASYNC_HISTORY > ctx->async_next, "ticket too old");
    const int s = (int)(ticket % mvsim_ctx::ASYNC_SLOTS);
    if (ctx->async_ticket[s] == ticket && ctx->async_inflight[s]) {
        MVSIM_HIP(hipEventSynchronize(ctx->ev_d2h[s]));
        ctx->async_inflight[s] = false;
        ctx->async_corr_done[ticket % mvsim_ctx::ASYNC_HISTORY] = ctx->async_corr[s];
        MVSIM_TRY(async_land(ctx, s));
    }
    // otherwise the view has landed already: a later call on the same staging set, or an earlier wait, saw to that
    if (correction) *correction = ctx->async_corr_done[ticket % mvsim_ctx::ASYNC_HISTORY];
    return MVSIM_OK;
}

// Host buffers in and out for the stacked / side-by-side views of mvsim_simulate_views_dev: the ground truth goes up once, the V views
// run in one call, and the acquisitions come back together -- as uint16 counts where the views are sampled (one transfer of half the
// bytes, widened by the host threads), as float32 for a view that holds a value beyond 65 535 or was simulated without noise.
int mvsim_simulate_views(mvsim_ctx* ctx, const float* gt_host, const int64_t dim[3], float* const* psf_host, const int64_t kdim[3],
                         const mvsim_view_params* params, float* const* acq_host, int n_views)
{
    MVSIM_TRY(set_device(ctx));
    MVSIM_TRY(check_dim(dim));
    MVSIM_CHECK_ARG(gt_host && psf_host && kdim && params && acq_host, "null pointer");
    MVSIM_CHECK_ARG(n_views >= 0 && n_views <= MVSIM_MAX_VIEWS, "n_views must be in [0, MVSIM_MAX_VIEWS]");
    if (n_views == 0) return MVSIM_OK;
    SyncOnExit sync{ctx};
    // views of the pipelined entry point that are still in flight own the 16-bit staging this call is about to use: land them first
    if (ctx->async_ready)
        for (int q = 0; q < mvsim_ctx::ASYNC_SLOTS; ++q)
            if (ctx->async_inflight[q]) {
                MVSIM_HIP(hipEventSynchronize(ctx->ev_d2h[q]));
                ctx->async_inflight[q] = false;
                ctx->async_corr_done[ctx->async_ticket[q] % mvsim_ctx::ASYNC_HISTORY] = ctx->async_corr[q];
                MVSIM_TRY(async_land(ctx, q));
            }
    const int64_t n = nvox(dim), plane = dim[0] * dim[1];
    // per-view acquisition sizes (the spacing may differ from view to view), 256-byte aligned slots
    std::vector<size_t> off((size_t)n_views + 1, 0), off16((size_t)n_views + 1, 0);
    std::vector<long long> cnt((size_t)n_views);
    for (int v = 0; v < n_views; ++v) {
        MVSIM_CHECK_ARG(acq_host[v] != nullptr && params[v].inc >= 1, "null acquisition buffer or inc < 1");
        cnt[(size_t)v] = plane * mvsim_extract_nz(dim[2], params[v].inc);
        off[(size_t)v + 1] = off[(size_t)v] + (((size_t)cnt[(size_t)v] * sizeof(float) + 255) & ~(size_t)255);
        off16[(size_t)v + 1] = off16[(size_t)v] + (((size_t)cnt[(size_t)v] * sizeof(unsigned short) + 255) & ~(size_t)255);
    }
    MVSIM_TRY(up(ctx, ctx->host_gt, gt_host, (size_t)n * sizeof(float)));
    MVSIM_TRY(ctx->out_buf.reserve(off[(size_t)n_views]));
    std::vector<mvsim_view_outputs> outs((size_t)n_views, mvsim_view_outputs{nullptr, nullptr, nullptr, nullptr});
    for (int v = 0; v < n_views; ++v) outs[(size_t)v].acq = reinterpret_cast<float*>(ctx->out_buf.as<char>() + off[(size_t)v]);
    MVSIM_TRY(mvsim_simulate_views_dev(ctx, ctx->host_gt.as<float>(), dim, psf_host, kdim, params, outs.data(), n_views));
    MVSIM_TRY(set_device(ctx));
    // pack the sampled views, fetch everything that was packed in ONE transfer (+ the flags), widen; float32 for the rest
    const size_t flags_at = off16[(size_t)n_views], u16_bytes = flags_at + (size_t)n_views * sizeof(unsigned int);
    bool any16 = false;
    std::vector<char> as16((size_t)n_views, 0);
    for (int v = 0; v < n_views; ++v) { as16[(size_t)v] = (ctx->opt.acq_u16 != 0 && params[v].snr >= 0.0f) ? 1 : 0; any16 = any16 || as16[(size_t)v]; }
    if (any16) {
        // the synchronous staging pair (down_counts' own): mvsim_destroy / mvsim_release_caches free it whether or not the pipelined
        // entry points ever set their slots up (the async slots leaked from a context that only came through here)
        MVSIM_TRY(ctx->sync_u16.reserve(u16_bytes));
        if (ctx->sync_u16_host_bytes < u16_bytes) {
            if (ctx->sync_u16_host) { (void)hipHostFree(ctx->sync_u16_host); ctx->sync_u16_host = nullptr; ctx->sync_u16_host_bytes = 0; }
            MVSIM_HIP(hipHostMalloc(&ctx->sync_u16_host, u16_bytes, hipHostMallocDefault));
            ctx->sync_u16_host_bytes = u16_bytes;
        }
        char* d16 = ctx->sync_u16.as<char>();
        MVSIM_HIP(hipMemsetAsync(d16 + flags_at, 0, (size_t)n_views * sizeof(unsigned int), ctx->stream));
        for (int v = 0; v < n_views; ++v)
            if (as16[(size_t)v])
                MVSIM_TRY(launch_pack_u16(ctx->stream, outs[(size_t)v].acq, reinterpret_cast<unsigned short*>(d16 + off16[(size_t)v]), cnt[(size_t)v],
                                          reinterpret_cast<unsigned int*>(d16 + flags_at) + v));
        MVSIM_HIP(hipMemcpyAsync(ctx->sync_u16_host, d16, u16_bytes, hipMemcpyDeviceToHost, ctx->stream));
    }
    for (int v = 0; v < n_views; ++v)
        if (!as16[(size_t)v])
            MVSIM_HIP(hipMemcpyAsync(acq_host[v], outs[(size_t)v].acq, (size_t)cnt[(size_t)v] * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    MVSIM_HIP(hipStreamSynchronize(ctx->stream));
    if (any16) {
        const char* h16 = reinterpret_cast<const char*>(ctx->sync_u16_host);
        const unsigned int* flags = reinterpret_cast<const unsigned int*>(h16 + flags_at);
        struct Job { const unsigned short* src; float* dst; long long n; };
        std::vector<Job> jobs;
        const long long chunk = (long long)1 << 20;
        for (int v = 0; v < n_views; ++v) {
            if (!as16[(size_t)v]) continue;
            ctx->u16_views += 1;
            if (flags[v] != 0u) {
                ctx->u16_fallbacks += 1;
                MVSIM_HIP(hipMemcpy(acq_host[v], outs[(size_t)v].acq, (size_t)cnt[(size_t)v] * sizeof(float), hipMemcpyDeviceToHost));
                continue;
            }
            const unsigned short* src = reinterpret_cast<const unsigned short*>(h16 + off16[(size_t)v]);
            for (long long a = 0; a < cnt[(size_t)v]; a += chunk) jobs.push_back(Job{src + a, acq_host[v] + a, std::min(chunk, cnt[(size_t)v] - a)});
        }
        HostPool::get().run((int)jobs.size(), host_threads_of(ctx), [&](int j) { widen_u16(jobs[(size_t)j].src, jobs[(size_t)j].dst, jobs[(size_t)j].n); });
    }
    return MVSIM_OK;
}

int mvsim_get_plane_stats(mvsim_ctx* ctx, int64_t stats[3])
{
    MVSIM_CHECK_ARG(ctx != nullptr && stats != nullptr, "null pointer");
    stats[0] = stats[1] = stats[2] = 0;
    if (ctx->empty_hint && ctx->empty_hint[0] >= 0) {
        const volatile int* h = ctx->empty_hint;
        stats[0] = h[2]; stats[1] = h[0]; stats[2] = h[1] >= 0 ? h[1] : 0;
    }
    return MVSIM_OK;
}

int mvsim_get_queue_stats(mvsim_ctx* ctx, int64_t stats[6])
{
    MVSIM_CHECK_ARG(ctx != nullptr && stats != nullptr, "null pointer");
    for (int i = 0; i < 6; ++i) stats[i] = 0;
    stats[0] = (int64_t)ctx->pqueue.bytes;
    if (!ctx->pqueue.p) return MVSIM_OK;
    MVSIM_TRY(mvsim_synchronize(ctx));
    long long q[5] = {0, 0, 0, 0, 0};
    MVSIM_TRY(poisson_queue_read_stats(ctx->pqueue.p, ctx->pqueue.bytes, q));
    for (int i = 0; i < 5; ++i) stats[1 + i] = q[i];
    return MVSIM_OK;
}

int mvsim_get_transfer_stats(mvsim_ctx* ctx, int64_t* views_as_u16, int64_t* fallbacks)
{
    MVSIM_CHECK_ARG(ctx != nullptr, "ctx is null");
    if (views_as_u16) *views_as_u16 = ctx->u16_views;
    if (fallbacks) *fallbacks = ctx->u16_fallbacks;
    return MVSIM_OK;
}

int mvsim_simulate_view_zslabs(mvsim_ctx* ctx, const float* const* gt_slabs, const int64_t* gt_slab_nz, int n_gt_slabs,
                               const int64_t dim[3], float* psf_host, const int64_t kdim[3], const mvsim_view_params* p,
                               float* const* acq_slabs, const int64_t* acq_slab_nz, int n_acq_slabs, double* correction)
{
    MVSIM_TRY(set_device(ctx));
    SyncOnExit sync{ctx};
    MVSIM_TRY(check_dim(dim));
    MVSIM_CHECK_ARG(gt_slabs && gt_slab_nz && acq_slabs && acq_slab_nz && p, "null pointer");
    MVSIM_CHECK_ARG(n_gt_slabs >= 1 && n_acq_slabs >= 1 && p->inc >= 1, "slab counts and inc must be >= 1");
    const int64_t plane = dim[0] * dim[1];
    const int64_t nzo = mvsim_extract_nz(dim[2], p->inc);
    int64_t zs = 0, za = 0;
    for (int i = 0; i < n_gt_slabs; ++i) { MVSIM_CHECK_ARG(gt_slabs[i] && gt_slab_nz[i] >= 1, "empty ground-truth slab"); zs += gt_slab_nz[i]; }
    for (int j = 0; j < n_acq_slabs; ++j) { MVSIM_CHECK_ARG(acq_slabs[j] && acq_slab_nz[j] >= 1, "empty acquisition slab"); za += acq_slab_nz[j]; }
    MVSIM_CHECK_ARG(zs == dim[2], "ground-truth slabs must add up to dim[2] planes");
    MVSIM_CHECK_ARG(za == nzo, "acquisition slabs must add up to mvsim_extract_nz(dim[2], inc) planes");
    MVSIM_TRY(ctx->host_gt.reserve((size_t)(plane * dim[2]) * sizeof(float)));
    MVSIM_TRY(ctx->out_buf.reserve((size_t)(plane * nzo) * sizeof(float)));
    int64_t z = 0;
    for (int i = 0; i < n_gt_slabs; ++i) {
        MVSIM_HIP(hipMemcpyAsync(ctx->host_gt.as<float>() + plane * z, gt_slabs[i], (size_t)(plane * gt_slab_nz[i]) * sizeof(float),
                                 hipMemcpyHostToDevice, ctx->stream));
        z += gt_slab_nz[i];
    }
    mvsim_view_outputs dev = {nullptr, nullptr, nullptr, ctx->out_buf.as<float>()};
    int rc = mvsim_simulate_view_dev(ctx, ctx->host_gt.as<float>(), dim, psf_host, kdim, p, &dev, correction);
    if (rc == MVSIM_OK) rc = join_tail(ctx);                // the copies below read what the tail writes
    z = 0;
    for (int j = 0; j < n_acq_slabs && rc == MVSIM_OK; ++j) {
        if (hipMemcpyAsync(acq_slabs[j], dev.acq + plane * z, (size_t)(plane * acq_slab_nz[j]) * sizeof(float), hipMemcpyDeviceToHost,
                           ctx->stream) != hipSuccess) { set_error("download of acquisition slab %d failed", j); rc = MVSIM_EHIP; }
        z += acq_slab_nz[j];
    }
    if (hipStreamSynchronize(ctx->stream) != hipSuccess && rc == MVSIM_OK) { set_error("stream synchronise failed"); rc = MVSIM_EHIP; }
    return rc;
}

// ---- per-stage operators on host buffers given as z slabs (volumes beyond one 2 GiB direct buffer) --------------------
namespace {
struct SlabList {
    const void* const* ptr;
    const int64_t*     nz;
    int                count;
};
int check_slabs(const SlabList& l, int64_t planes, const char* what)
{
    if (!l.ptr || !l.nz || l.count < 1) { set_error("invalid argument: %s slab list is empty", what); return MVSIM_EINVAL; }
    int64_t z = 0;
    for (int i = 0; i < l.count; ++i) {
        if (!l.ptr[i] || l.nz[i] < 1) { set_error("invalid argument: %s slab %d is empty", what, i); return MVSIM_EINVAL; }
        z += l.nz[i];
    }
    if (z != planes) { set_error("invalid argument: %s slabs hold %lld planes, expected %lld", what, (long long)z, (long long)planes); return MVSIM_EINVAL; }
    return MVSIM_OK;
}
int upload_slabs(mvsim_ctx* ctx, const SlabList& l, int64_t plane, float* dev)
{
    int64_t z = 0;
    for (int i = 0; i < l.count; ++i) {
        MVSIM_HIP(hipMemcpyAsync(dev + plane * z, l.ptr[i], (size_t)(plane * l.nz[i]) * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
        z += l.nz[i];
    }
    return MVSIM_OK;
}
int download_slabs(mvsim_ctx* ctx, const SlabList& l, int64_t plane, const float* dev)
{
    int rc = MVSIM_OK;
    int64_t z = 0;
    for (int i = 0; i < l.count && rc == MVSIM_OK; ++i) {
        if (hipMemcpyAsync(const_cast<void*>(l.ptr[i]), dev + plane * z, (size_t)(plane * l.nz[i]) * sizeof(float), hipMemcpyDeviceToHost,
                           ctx->stream) != hipSuccess) { set_error("download of slab %d failed", i); rc = MVSIM_EHIP; }
        z += l.nz[i];
    }
    if (hipStreamSynchronize(ctx->stream) != hipSuccess && rc == MVSIM_OK) { set_error("stream synchronise failed"); rc = MVSIM_EHIP; }
    return rc;
}
// stage the input slabs in vol_a, make room for `out_planes` planes in vol_b
int slabs_in(mvsim_ctx* ctx, const float* const* in_slabs, const int64_t* in_nz, int n_in, const int64_t dim[3], float* const* out_slabs,
             const int64_t* out_nz, int n_out, int64_t out_planes, SlabList* in, SlabList* out)
{
    MVSIM_TRY(set_device(ctx));
    MVSIM_TRY(check_dim(dim));
    *in = SlabList{reinterpret_cast<const void* const*>(in_slabs), in_nz, n_in};
    *out = SlabList{reinterpret_cast<const void* const*>(out_slabs), out_nz, n_out};
    MVSIM_TRY(check_slabs(*in, dim[2], "input"));
    MVSIM_TRY(check_slabs(*out, out_planes, "output"));
    const int64_t plane = dim[0] * dim[1];
    MVSIM_TRY(ctx->vol_a.reserve((size_t)(plane * dim[2]) * sizeof(float)));
    MVSIM_TRY(ctx->vol_b.reserve((size_t)(plane * out_planes) * sizeof(float)));
    return upload_slabs(ctx, *in, plane, ctx->vol_a.as<float>());
}
}  // namespace

int mvsim_rotate_around_axis_zslabs(mvsim_ctx* ctx, const float* const* in_slabs, const int64_t* in_slab_nz, int n_in, const int64_t dim[3],
                                    int axis, int degrees, float* const* out_slabs, const int64_t* out_slab_nz, int n_out)
{
    SlabList in, out;
    SyncOnExit sync{ctx};
    MVSIM_TRY(slabs_in(ctx, in_slabs, in_slab_nz, n_in, dim, out_slabs, out_slab_nz, n_out, dim ? dim[2] : 0, &in, &out));
    MVSIM_TRY(mvsim_rotate_around_axis_dev(ctx, ctx->vol_a.as<float>(), dim, axis, degrees, ctx->vol_b.as<float>()));
    return download_slabs(ctx, out, dim[0] * dim[1], ctx->vol_b.as<float>());
}

int mvsim_attenuate3d_zslabs(mvsim_ctx* ctx, const float* const* in_slabs, const int64_t* in_slab_nz, int n_in, const int64_t dim[3],
                             double delta, float* const* out_slabs, const int64_t* out_slab_nz, int n_out)
{
    SlabList in, out;
    SyncOnExit sync{ctx};
    MVSIM_TRY(slabs_in(ctx, in_slabs, in_slab_nz, n_in, dim, out_slabs, out_slab_nz, n_out, dim ? dim[2] : 0, &in, &out));
    MVSIM_TRY(mvsim_attenuate3d_dev(ctx, ctx->vol_a.as<float>(), dim, delta, ctx->vol_b.as<float>()));
    return download_slabs(ctx, out, dim[0] * dim[1], ctx->vol_b.as<float>());
}

int mvsim_convolve_zslabs(mvsim_ctx* ctx, const float* const* in_slabs, const int64_t* in_slab_nz, int n_in, const int64_t dim[3],
                          float* psf, const int64_t kdim[3], int method, float* const* out_slabs, const int64_t* out_slab_nz, int n_out)
{
    SlabList in, out;
    SyncOnExit sync{ctx};
    MVSIM_TRY(slabs_in(ctx, in_slabs, in_slab_nz, n_in, dim, out_slabs, out_slab_nz, n_out, dim ? dim[2] : 0, &in, &out));
    MVSIM_TRY(mvsim_convolve_dev(ctx, ctx->vol_a.as<float>(), dim, psf, kdim, method, ctx->vol_b.as<float>()));
    return download_slabs(ctx, out, dim[0] * dim[1], ctx->vol_b.as<float>());
}

int mvsim_extract_slices_zslabs(mvsim_ctx* ctx, const float* const* in_slabs, const int64_t* in_slab_nz, int n_in, const int64_t dim[3],
                                int inc, float snr, uint64_t seed, uint32_t stream, float* const* out_slabs, const int64_t* out_slab_nz,
                                int n_out)
{
    MVSIM_CHECK_ARG(inc >= 1, "inc must be >= 1");
    SlabList in, out;
    SyncOnExit sync{ctx};
    MVSIM_TRY(slabs_in(ctx, in_slabs, in_slab_nz, n_in, dim, out_slabs, out_slab_nz, n_out, dim ? mvsim_extract_nz(dim[2], inc) : 0, &in, &out));
    MVSIM_TRY(mvsim_extract_slices_dev(ctx, ctx->vol_a.as<float>(), dim, inc, snr, seed, stream, ctx->vol_b.as<float>()));
    MVSIM_TRY(join_tail(ctx));
    return download_slabs(ctx, out, dim[0] * dim[1], ctx->vol_b.as<float>());
}

int mvsim_stencil_geometry(const int64_t kdim[3], int64_t geometry[5])
{
    if (!kdim || !geometry) { set_error("invalid argument: null pointer"); return MVSIM_EINVAL; }
    if (!stencil_chunk_geometry(kdim, geometry)) {
        set_error("direct stencil: PSF outside 1..64 taps per axis");
        return MVSIM_EINVAL;
    }
    return MVSIM_OK;
}

int mvsim_fft_geometry(const int64_t dim[3], const int64_t kdim[3], int64_t geometry[5])
{
    if (!dim || !kdim || !geometry) { set_error("invalid argument: null pointer"); return MVSIM_EINVAL; }
    MVSIM_TRY(check_dim(dim));
    if (!custom_fft_geometry(dim, kdim, geometry, env_options())) {
        set_error("no hand-written FFT size for this volume / PSF");
        return MVSIM_EINVAL;
    }
    return MVSIM_OK;
}

// ---- timings ---------------------------------------------------------------------------------------
int mvsim_enable_timing(mvsim_ctx* ctx, int enable)
{
    MVSIM_TRY(set_device(ctx));
    if (enable && !ctx->ev_created) {
        for (int k = 0; k < mvsim_ctx::TIMING_SLOTS; ++k)
            for (int s = 0; s < ST_COUNT; ++s) {
                MVSIM_HIP(hipEventCreate(&ctx->evr[k][s][0]));
                MVSIM_HIP(hipEventCreate(&ctx->evr[k][s][1]));
            }
        ctx->ev_created = true;
    }
    ctx->timing = enable != 0;
    ctx->ev_cur = 0;
    ctx->ev_calls = 0;
    for (int s = 0; s < ST_COUNT; ++s) ctx->ev_used[0][s] = false;
    return MVSIM_OK;
}

// Averages over the calls recorded since timing was enabled or last read (at most the last TIMING_SLOTS calls).
int mvsim_get_timings(mvsim_ctx* ctx, mvsim_timings* t)
{
    MVSIM_TRY(set_device(ctx));
    MVSIM_CHECK_ARG(t != nullptr, "timings pointer is null");
    MVSIM_CHECK_ARG(ctx->ev_created, "timing was never enabled");
    MVSIM_HIP(hipStreamSynchronize(ctx->stream));
    double sum[ST_COUNT] = {};
    int cnt[ST_COUNT] = {};
    const long long calls = ctx->ev_calls == 0 ? 1 : ctx->ev_calls;    // stage operators outside simulate_view use slot 0
    const int nslots = (int)(calls < mvsim_ctx::TIMING_SLOTS ? calls : mvsim_ctx::TIMING_SLOTS);
    for (int j = 0; j < nslots; ++j) {
        const int k = ((ctx->ev_cur - j) % mvsim_ctx::TIMING_SLOTS + mvsim_ctx::TIMING_SLOTS) % mvsim_ctx::TIMING_SLOTS;
        for (int s = 0; s < ST_COUNT; ++s) {
            if (!ctx->ev_used[k][s]) continue;
            float ms = 0.f;
            MVSIM_HIP(hipEventElapsedTime(&ms, ctx->evr[k][s][0], ctx->evr[k][s][1]));
            sum[s] += ms;
            cnt[s] += 1;
            ctx->ev_used[k][s] = false;
        }
    }
    float ms[ST_COUNT];
    float total = 0.f;
    for (int s = 0; s < ST_COUNT; ++s) {
        ms[s] = cnt[s] ? (float)(sum[s] / cnt[s]) : 0.f;
        // a PSF spectrum that ran on the side stream overlaps passes A and B: it is part of convolve_ms, not a stage of its own
        if (s == ST_PSF && ctx->psf_on_side) ms[s] = 0.f;
        if (s < ST_PASS_A) total += ms[s];              // the passes are nested inside ST_CONVOLVE
    }
    t->rotate_ms = ms[ST_ROTATE]; t->attenuate_ms = ms[ST_ATTENUATE]; t->psf_ms = ms[ST_PSF];
    t->convolve_ms = ms[ST_CONVOLVE]; t->adjust_ms = ms[ST_ADJUST]; t->extract_ms = ms[ST_EXTRACT];
    t->total_ms = total;
    t->pass_a_ms = ms[ST_PASS_A]; t->pass_b_ms = ms[ST_PASS_B]; t->pass_c_ms = ms[ST_PASS_C];
    t->pass_d_ms = ms[ST_PASS_D]; t->pass_e_ms = ms[ST_PASS_E];
    ctx->last = *t;
    ctx->ev_cur = 0;
    ctx->ev_calls = 0;
    return MVSIM_OK;
}

}  // extern "C"
