// Error plumbing of libvq2 (thread-local, see include/vq2.h).
#include "vq2_common.h"
#include <atomic>
#include <mutex>
#include <string.h>
#include <utility>
#include <vector>

namespace vq2 {

static thread_local char g_err[512] = "";

int set_error(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int check_launch(const char *what) {
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return VQ2_OK;
    return set_error(VQ2_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
}

// ------------------------------------------------------------------ per-launch HIP-event profiler
// Off by default (zero cost: one relaxed load per launch).  bench.py switches it on to obtain the
// roofline numbers live: every instrumented launch is bracketed by two events on ITS stream.
struct ProfRec { const char *name; hipEvent_t a, b; double flops, bytes; };
static std::atomic<int> g_prof_on{0};
static std::mutex g_prof_mu;
static std::vector<ProfRec> g_prof;
static std::vector<std::pair<hipEvent_t, hipEvent_t>> g_prof_free;

bool prof_enabled() { return g_prof_on.load(std::memory_order_relaxed) != 0; }
int prof_level() { return g_prof_on.load(std::memory_order_relaxed); }

// intern a formatted kernel+shape label (pointers stay valid for the life of the process)
const char *prof_label(const char *fmt, ...) {
    char tmp[160];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(tmp, sizeof(tmp), fmt, ap);
    va_end(ap);
    static std::vector<char *> pool;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (char *p : pool) if (strcmp(p, tmp) == 0) return p;
    pool.push_back(strdup(tmp));
    return pool.back();
}

int prof_begin(const char *name, double flops, double bytes, hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    ProfRec r{name, nullptr, nullptr, flops, bytes};
    if (!g_prof_free.empty()) { r.a = g_prof_free.back().first; r.b = g_prof_free.back().second; g_prof_free.pop_back(); }
    else { if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return -1; }
    (void)hipEventRecord(r.a, s);
    g_prof.push_back(r);
    return (int)g_prof.size() - 1;
}

void prof_end(int id, hipStream_t s) {
    if (id < 0) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (id < (int)g_prof.size()) (void)hipEventRecord(g_prof[id].b, s);
}

}  // namespace vq2

extern "C" int vq2_prof_enable(int level) {
    vq2::g_prof_on.store(level < 0 ? 0 : level);
    return VQ2_OK;
}

// Waits for the recorded events (this is the one entry point that synchronises), aggregates per
// kernel name and writes `name launches total_ms flops bytes` lines into buf; clears the records.
extern "C" int vq2_prof_report(char *buf, size_t cap) {
    using namespace vq2;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    struct Agg { const char *name; long n; double ms, flops, bytes; };
    std::vector<Agg> agg;
    for (auto &r : g_prof) {
        float ms = 0.f;
        if (hipEventSynchronize(r.b) != hipSuccess || hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) ms = 0.f;
        Agg *a = nullptr;
        for (auto &x : agg) if (x.name == r.name || strcmp(x.name, r.name) == 0) { a = &x; break; }
        if (!a) { agg.push_back(Agg{r.name, 0, 0, 0, 0}); a = &agg.back(); }
        a->n += 1; a->ms += ms; a->flops += r.flops; a->bytes += r.bytes;
        g_prof_free.emplace_back(r.a, r.b);
    }
    g_prof.clear();
    size_t off = 0;
    if (buf && cap) buf[0] = 0;
    for (auto &a : agg) {
        int w = snprintf(buf + off, off < cap ? cap - off : 0, "%s %ld %.6f %.6e %.6e\n", a.name, a.n, a.ms, a.flops,
                         a.bytes);
        if (w < 0 || off + (size_t)w >= cap) return set_error(VQ2_ERR_INVALID, "prof_report: buffer too small");
        off += (size_t)w;
    }
    return VQ2_OK;
}

extern "C" int vq2_version(void) { return VQ2_API_VERSION; }
extern "C" const char *vq2_last_error(void) { return vq2::g_err; }
