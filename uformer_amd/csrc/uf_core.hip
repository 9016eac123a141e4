// Error plumbing and version of libuformer_hip.
#include <stdarg.h>

#include <mutex>
#include <string>
#include <vector>

#include <stdlib.h>
#include <string.h>

#include "uf_internal.h"

namespace uf {

static thread_local char g_err[512] = "";

// UF_VARIANT="key=value,key=value,...": the ONE environment variable that selects between launch variants of a kernel for A/B runs and for the
// bit-identity tests (round 6: it replaces eight separate UF_* switches).  Every variant of a key computes identical bits; without the key the shape picks.
//   attn=0|1|2|3    attn_block: first form / low-register form with tight / relaxed register bounds / single-operand-tile form (C = 256)
//   leff2=1|2       leff2: 8 producer waves wherever they are built / never
//   persist=0|1     leff2: one tile per workgroup / the persistent tile walk everywhere
//   gemm_dma=0|1    gemm_kernel: register-staged / LDS-DMA operand staging on every shape that supports it
//   wgrad4=0|1      linear_wgrad4 (256 x 256 tiles) never / on every shape it supports
//   stem=1, head=1  the first (global-load) forms of input_proj / output_proj instead of the LDS-staged ones
//   down=1|2        Downsample: the im2col-loader GEMM everywhere / the LDS-patch form wherever it is built (round 6)
//   attpair=0       window_attn_bwd2: (head, chunk) in launch order instead of head pairs per XCD (round 6)
//   downdx=1        uf_downsample_bwd: the input gradient through the patch matrix (GEMM + col2im) instead of the LDS-patch kernel (round 6)
// Read per call (a getenv and a scan of a short string beside a kernel launch): the tests flip keys inside one process.  Returns `dflt` without the key.
int variant(const char* key, int dflt) {
    const char* e = getenv("UF_VARIANT");
    if (!e) return dflt;
    const size_t n = strlen(key);
    for (const char* q = e; *q;) {
        if (!strncmp(q, key, n) && q[n] == '=') return atoi(q + n + 1);
        while (*q && *q != ',') ++q;
        if (*q == ',') ++q;
    }
    return dflt;
}

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: kernel launch failed: %s", what, hipGetErrorString(e));
        return UF_ERR_LAUNCH;
    }
    return UF_OK;
}

// ---- opt-in per-kernel-class timing with HIP events on the launch stream -------------------
namespace {
struct Rec { hipEvent_t a, b; int cls; };
struct Cls { std::string name; double flops = 0, bytes = 0; long launches = 0; };
bool g_timing = false;
std::mutex g_mu;
std::vector<Rec> g_recs;
std::vector<Cls> g_cls;
}  // namespace

bool timing_enabled() { return g_timing; }

ScopedTimer::ScopedTimer(const char* name, double flops, double bytes, hipStream_t st) : st_(st), idx_(-1) {
    if (!g_timing) return;
    std::lock_guard<std::mutex> lk(g_mu);
    int c = -1;
    for (size_t i = 0; i < g_cls.size(); ++i)
        if (g_cls[i].name == name) { c = (int)i; break; }
    if (c < 0) { g_cls.push_back(Cls{name}); c = (int)g_cls.size() - 1; }
    g_cls[c].flops += flops; g_cls[c].bytes += bytes; g_cls[c].launches += 1;
    Rec r; r.cls = c;
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
    hipEventRecord(r.a, st_);
    g_recs.push_back(r);
    idx_ = (int)g_recs.size() - 1;
}

ScopedTimer::~ScopedTimer() {
    if (idx_ < 0) return;
    std::lock_guard<std::mutex> lk(g_mu);
    if (idx_ < (int)g_recs.size()) hipEventRecord(g_recs[idx_].b, st_);
}


// Extra in-order queues + fork/join events of the multi-stream mode.  A LANE = {side streams, fork event, join events}; a
// call to uf_uformer_fwd owns one lane exclusively from its first hipEventRecord to its last hipStreamWaitEvent, so two host
// threads (autograd worker + main thread, or two replicas of a process) driving the same GPU never record or wait on each
// other's events.  Lanes live in a per-device pool guarded by a mutex and are created on demand (normally: one per device);
// a lane handed back is reused by the next call -- work already enqueued on its side streams keeps its order (in-order
// queues), and hipStreamWaitEvent binds to the record that preceded it, so re-recording an event for the next call is safe.
namespace {
struct LanePool { std::mutex mu; std::vector<Lane*> idle; };
LanePool& lane_pool(int dev) {
    static LanePool pools[64];   // constant-initialised members; each pool is guarded by its own mutex
    return pools[dev];
}
}  // namespace
// takes an idle lane of the current device (or makes one) and grows it to `want` side streams; nullptr on failure
Lane* acquire_lane(int want, int* dev_out) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    *dev_out = dev;
    LanePool& pool = lane_pool(dev);
    Lane* ln = nullptr;
    {
        std::lock_guard<std::mutex> lk(pool.mu);
        if (!pool.idle.empty()) { ln = pool.idle.back(); pool.idle.pop_back(); }
    }
    if (!ln) ln = new Lane();
    bool ok = ln->fork || hipEventCreateWithFlags(&ln->fork, hipEventDisableTiming) == hipSuccess;
    for (; ok && ln->n < want && ln->n < MAX_SIDE; ++ln->n) {
        ok = hipStreamCreateWithFlags(&ln->s[ln->n], hipStreamNonBlocking) == hipSuccess &&
             hipEventCreateWithFlags(&ln->join[ln->n], hipEventDisableTiming) == hipSuccess;
        if (!ok) break;
    }
    if (!ok || ln->n < want) {   // hand it back as it is; the caller reports the failure
        std::lock_guard<std::mutex> lk(pool.mu);
        pool.idle.push_back(ln);
        return nullptr;
    }
    return ln;
}
void release_lane(Lane* ln, int dev) {
    LanePool& pool = lane_pool(dev);
    std::lock_guard<std::mutex> lk(pool.mu);
    pool.idle.push_back(ln);
}
// the lane's general-purpose events (fork points of the block backward's side stream), created on first use
bool lane_events(Lane* ln, int want) {
    for (; ln->n_ev < want && ln->n_ev < MAX_LANE_EVENTS; ++ln->n_ev)
        if (hipEventCreateWithFlags(&ln->ev[ln->n_ev], hipEventDisableTiming) != hipSuccess) return false;
    return ln->n_ev >= want;
}

}  // namespace uf

extern "C" int uf_timing_enable(int on) {
    std::lock_guard<std::mutex> lk(uf::g_mu);
    uf::g_timing = on != 0;
    return UF_OK;
}

extern "C" int uf_timing_report(char* json, size_t n) {
    using namespace uf;
    std::lock_guard<std::mutex> lk(g_mu);
    std::vector<double> ms(g_cls.size(), 0.0);
    for (auto& r : g_recs) {
        float t = 0.f;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) ms[r.cls] += t;
        hipEventDestroy(r.a);
        hipEventDestroy(r.b);
    }
    std::string out = "[";
    for (size_t i = 0; i < g_cls.size(); ++i) {
        char buf[320];
        snprintf(buf, sizeof(buf), "%s{\"kernel\":\"%s\",\"launches\":%ld,\"ms\":%.6f,\"flops\":%.6e,\"bytes\":%.6e}",
                 i ? "," : "", g_cls[i].name.c_str(), g_cls[i].launches, ms[i], g_cls[i].flops, g_cls[i].bytes);
        out += buf;
    }
    out += "]";
    g_recs.clear();
    g_cls.clear();
    if (json && n) {
        const size_t c = out.size() < n - 1 ? out.size() : n - 1;
        memcpy(json, out.data(), c);
        json[c] = 0;
    }
    return (int)out.size();
}

namespace uf { void debug_set_tbuf(void* p); }
// development aid: device buffer that instrumented kernels fill with s_memtime stamps (NULL = off)
extern "C" int uf_debug_set_tbuf(void* p) { uf::debug_set_tbuf(p); return UF_OK; }

extern "C" int uf_version(void) { return UF_ABI_VERSION; }

extern "C" int uf_last_error(char* buf, size_t n) {
    const size_t len = strlen(uf::g_err);
    if (buf && n) {
        const size_t c = len < n - 1 ? len : n - 1;
        memcpy(buf, uf::g_err, c);
        buf[c] = 0;
    }
    return (int)len;
}
