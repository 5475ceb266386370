// Library state of libssdr_al.so: device selection, the library stream, error text, grow-only buffers.
#include "ssdr_internal.hpp"
#include <mutex>
#include <unistd.h>

namespace ssdr {

static thread_local std::string g_err;

void set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
    g_err = buf;
}

Context& ctx() { static Context c; return c; }

static std::mutex g_init_mu;
static pid_t g_init_pid = 0;          // the process that created the HIP context

// The reference calls knn_search from forked DataLoader workers (SURVEY 8b "Threading").  A HIP context does not survive fork():
// nothing here touches HIP before the first op (importing / loading the library is fork-safe), and a child that inherits an
// initialised parent is refused with a clear error instead of hanging inside the runtime.
static int forked_child() {
    set_error("this process was forked after its parent initialised the HIP context (pid %d -> %d): a HIP context does not survive fork(). "
              "Fork before the first libssdr_al call (the library initialises HIP lazily, in the first op), or use the 'spawn' start method / "
              "num_workers=0", (int)g_init_pid, (int)getpid());
    return SSDR_ERR_INTERNAL;
}

__global__ void bind_kernel(int* p) { if (threadIdx.x == 0) *p = 1; }

static int do_init(int device) {
    std::lock_guard<std::mutex> lk(g_init_mu);
    Context& c = ctx();
    if (c.ready) return g_init_pid == getpid() ? SSDR_OK : forked_child();
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_error("no HIP device available (%s); libssdr_al has no CPU fallback",
                  e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
        return SSDR_ERR_NO_DEVICE;
    }
    if (device < 0) device = 0;
    if (device >= n) { set_error("device %d out of range (%d devices)", device, n); return SSDR_ERR_INVALID; }
    SSDR_HIP(hipSetDevice(device));
    SSDR_HIP(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));     // no implicit ordering against the legacy NULL stream (a host framework's default stream would serialise every stream of the pipeline)
    SSDR_HIP(hipEventCreate(&c.ev0));
    SSDR_HIP(hipEventCreate(&c.ev1));
    int cu = 0;
    if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cu > 0) c.num_cu = cu;
    // The runtime spreads the streams over its few hardware queues as they come into use, the legacy NULL stream included — and a
    // plain hipMemcpy (weight upload, status read-back) goes through the NULL stream.  A pipeline whose stage streams were created
    // BEFORE the NULL stream's first use ran 6.4 ms per step instead of 4.8 (same kernels, worse overlap: the late-comer ends up sharing a
    // queue with a busy stage).  Bring the NULL stream and the library stream into use now, before a caller creates anything.
    void* probe = nullptr; int zero = 0;
    SSDR_HIP(hipMalloc(&probe, 256));
    SSDR_HIP(hipMemcpy(probe, &zero, sizeof(zero), hipMemcpyHostToDevice));           // NULL stream
    hipLaunchKernelGGL(bind_kernel, dim3(1), dim3(64), 0, c.stream, reinterpret_cast<int*>(probe));
    SSDR_HIP(hipStreamSynchronize(c.stream));
    SSDR_HIP(hipFree(probe));
    c.device = device; c.ready = true; g_init_pid = getpid();
    return SSDR_OK;
}

static std::mutex g_forget_mu;
static std::vector<void (*)(hipStream_t)>& forgetters() { static auto* v = new std::vector<void (*)(hipStream_t)>; return *v; }
void register_stream_forgetter(void (*f)(hipStream_t)) { std::lock_guard<std::mutex> lk(g_forget_mu); forgetters().push_back(f); }
void forget_stream(hipStream_t s) {
    std::vector<void (*)(hipStream_t)> fs;
    { std::lock_guard<std::mutex> lk(g_forget_mu); fs = forgetters(); }
    for (auto f : fs) f(s);
}

int ensure_init() { return ctx().ready ? (g_init_pid == getpid() ? SSDR_OK : forked_child()) : do_init(0); }

int DevBuf::reserve(size_t bytes) {
    if (bytes <= cap) return SSDR_OK;
    if (p) { SSDR_HIP(hipDeviceSynchronize()); SSDR_HIP(hipFree(p)); p = nullptr; cap = 0; }   // any stream may still use it
    size_t want = bytes + bytes / 4 + 256;
    SSDR_HIP(hipMalloc(&p, want));
    cap = want;
    return SSDR_OK;
}
void DevBuf::release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }

// ---- kernel timing ------------------------------------------------------------------------------------
struct ProfRec { std::string name; hipEvent_t e0, e1; double work, work2; };
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof_recs;
static std::vector<hipEvent_t> g_prof_pool;
static std::string g_prof_text;

static hipEvent_t prof_event() {
    if (!g_prof_pool.empty()) { hipEvent_t e = g_prof_pool.back(); g_prof_pool.pop_back(); return e; }
    hipEvent_t e = nullptr; (void)hipEventCreate(&e); return e;
}

ProfScope::ProfScope(const char* name, hipStream_t s, double work, double work2) : stream(s) {
    if (!g_prof_on) return;
    ProfRec r; r.name = name; r.e0 = prof_event(); r.e1 = prof_event(); r.work = work; r.work2 = work2;
    (void)hipEventRecord(r.e0, s);
    g_prof_recs.push_back(r); slot = (int)g_prof_recs.size() - 1;
}
ProfScope::~ProfScope() { if (slot >= 0) (void)hipEventRecord(g_prof_recs[slot].e1, stream); }

}  // namespace ssdr

extern "C" {
int ssdr_prof_enable(int on) { ssdr::g_prof_on = on != 0; return SSDR_OK; }
/* Synchronises, folds the recorded launches into one line per kernel name: "name calls total_ms total_work total_work2\n". */
const char* ssdr_prof_report(void) {
    using namespace ssdr;
    struct Acc { long calls = 0; double ms = 0, work = 0, work2 = 0; };
    std::vector<std::pair<std::string, Acc>> acc;
    for (auto& r : g_prof_recs) {
        (void)hipEventSynchronize(r.e1);
        float ms = 0.f; (void)hipEventElapsedTime(&ms, r.e0, r.e1);
        size_t i = 0; for (; i < acc.size(); ++i) if (acc[i].first == r.name) break;
        if (i == acc.size()) acc.push_back({r.name, Acc()});
        acc[i].second.calls++; acc[i].second.ms += ms; acc[i].second.work += r.work; acc[i].second.work2 += r.work2;
        g_prof_pool.push_back(r.e0); g_prof_pool.push_back(r.e1);
    }
    g_prof_recs.clear();
    g_prof_text.clear();
    char buf[256];
    for (auto& a : acc) { snprintf(buf, sizeof(buf), "%s %ld %.6f %.6e %.6e\n", a.first.c_str(), a.second.calls, a.second.ms, a.second.work, a.second.work2); g_prof_text += buf; }
    return g_prof_text.c_str();
}
const char* ssdr_version(void) { return "ssdr_al-gfx950 0.1"; }
const char* ssdr_last_error(void) { return ssdr::g_err.c_str(); }
int ssdr_init(int device) { return ssdr::ctx().ready ? ssdr::ensure_init() : ssdr::do_init(device); }
void ssdr_shutdown(void) {}
int ssdr_stream_sync(void* stream) {
    SSDR_TRY(ssdr::ensure_init());
    SSDR_HIP(hipStreamSynchronize(ssdr::pick_stream(stream)));
    return SSDR_OK;
}
float ssdr_last_gpu_ms(void) { return ssdr::ctx().last_ms; }
int ssdr_stream_create(void** out_stream) {
    if (!out_stream) { ssdr::set_error("stream_create: NULL"); return SSDR_ERR_INVALID; }
    SSDR_TRY(ssdr::ensure_init());
    hipStream_t s = nullptr;
    SSDR_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *out_stream = s;
    return SSDR_OK;
}
/* priority: 0 = the default, > 0 = the highest priority the device offers, < 0 = the lowest.  Streams of another priority live on hardware queues of
 * their own: short dependent launches on a high-priority stream are not held behind the chip-filling kernels of the other streams. */
int ssdr_stream_create_priority(void** out_stream, int priority) {
    if (!out_stream) { ssdr::set_error("stream_create_priority: NULL"); return SSDR_ERR_INVALID; }
    SSDR_TRY(ssdr::ensure_init());
    int least = 0, greatest = 0;
    SSDR_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));          // numerically lower = higher priority
    hipStream_t s = nullptr;
    SSDR_HIP(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, priority > 0 ? greatest : (priority < 0 ? least : 0)));
    *out_stream = s;
    return SSDR_OK;
}
/* the library's own stream (what stream == NULL means everywhere), e.g. to wrap it as a framework's external stream */
int ssdr_main_stream(void** out_stream) {
    if (!out_stream) { ssdr::set_error("main_stream: NULL"); return SSDR_ERR_INVALID; }
    SSDR_TRY(ssdr::ensure_init());
    *out_stream = ssdr::ctx().stream;
    return SSDR_OK;
}
int ssdr_stream_destroy(void* stream) {
    if (!stream) return SSDR_OK;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    SSDR_HIP(hipStreamSynchronize(s));            // the per-stream scratch below may still be in use by what the stream holds
    ssdr::forget_stream(s);
    SSDR_HIP(hipStreamDestroy(s));
    return SSDR_OK;
}
/* work enqueued on `waiter` after this call starts only after everything enqueued on `waited` so far has finished */
int ssdr_stream_wait(void* waiter, void* waited) {
    SSDR_TRY(ssdr::ensure_init());
    hipEvent_t e = nullptr;
    SSDR_HIP(hipEventCreate(&e));
    SSDR_HIP(hipEventRecord(e, ssdr::pick_stream(waited)));
    SSDR_HIP(hipStreamWaitEvent(ssdr::pick_stream(waiter), e, 0));
    SSDR_HIP(hipEventDestroy(e));
    return SSDR_OK;
}
/* Events: a point in a stream's work that other streams can wait for later (ssdr_stream_wait names "everything so far" at the moment of the call;
 * an event names it at the moment of its record, and the wait may be issued any time afterwards — what a caller with several batches in flight needs to
 * express "this buffer set's last reader", pipeline.ALRound). */
int ssdr_event_create(void** ev) {
    if (!ev) { ssdr::set_error("event_create: NULL"); return SSDR_ERR_INVALID; }
    SSDR_TRY(ssdr::ensure_init());
    hipEvent_t e = nullptr;
    SSDR_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    *ev = e;
    return SSDR_OK;
}
int ssdr_event_record(void* ev, void* stream) {
    if (!ev) { ssdr::set_error("event_record: NULL"); return SSDR_ERR_INVALID; }
    SSDR_TRY(ssdr::ensure_init());
    SSDR_HIP(hipEventRecord(static_cast<hipEvent_t>(ev), ssdr::pick_stream(stream)));
    return SSDR_OK;
}
int ssdr_stream_wait_event(void* stream, void* ev) {
    if (!ev) { ssdr::set_error("stream_wait_event: NULL"); return SSDR_ERR_INVALID; }
    SSDR_TRY(ssdr::ensure_init());
    SSDR_HIP(hipStreamWaitEvent(ssdr::pick_stream(stream), static_cast<hipEvent_t>(ev), 0));
    return SSDR_OK;
}
int ssdr_event_destroy(void* ev) {
    if (ev) (void)hipEventDestroy(static_cast<hipEvent_t>(ev));
    return SSDR_OK;
}
int ssdr_dev_alloc(size_t bytes, void** d_ptr) {
    if (!d_ptr) { ssdr::set_error("dev_alloc: NULL"); return SSDR_ERR_INVALID; }
    SSDR_TRY(ssdr::ensure_init());
    SSDR_HIP(hipMalloc(d_ptr, bytes ? bytes : 1));
    return SSDR_OK;
}
int ssdr_dev_free(void* d_ptr) { if (d_ptr) SSDR_HIP(hipFree(d_ptr)); return SSDR_OK; }
int ssdr_memcpy_h2d(void* d_dst, const void* src, size_t bytes) {
    SSDR_TRY(ssdr::ensure_init());
    SSDR_HIP(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, ssdr::ctx().stream));
    SSDR_HIP(hipStreamSynchronize(ssdr::ctx().stream));
    return SSDR_OK;
}
int ssdr_memcpy_h2d_on(void* d_dst, const void* src, size_t bytes, void* stream) {
    SSDR_TRY(ssdr::ensure_init());
    hipStream_t s = ssdr::pick_stream(stream);
    SSDR_HIP(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, s));
    SSDR_HIP(hipStreamSynchronize(s));
    return SSDR_OK;
}
int ssdr_memcpy_d2h_on(void* dst, const void* d_src, size_t bytes, void* stream) {
    SSDR_TRY(ssdr::ensure_init());
    hipStream_t s = ssdr::pick_stream(stream);
    SSDR_HIP(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, s));
    SSDR_HIP(hipStreamSynchronize(s));
    return SSDR_OK;
}
int ssdr_memcpy_d2h(void* dst, const void* d_src, size_t bytes) {
    SSDR_TRY(ssdr::ensure_init());
    SSDR_HIP(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, ssdr::ctx().stream));
    SSDR_HIP(hipStreamSynchronize(ssdr::ctx().stream));
    return SSDR_OK;
}
}
