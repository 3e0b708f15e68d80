// Optional per-kernel timing with HIP events recorded on the launch stream.
// Used by bench.py to report the dominant kernel's average duration (roofline leg);
// disabled by default, so the product path pays one predictable branch per launch.
#include "urn_common.h"
#include "urn_prof.h"
#include <vector>

namespace {
struct Rec { hipEvent_t a, b; int kind; };
bool g_on = false;
std::vector<Rec> g_recs;
}  // namespace

bool urn_prof_on() { return g_on; }

void urn_prof_begin(int kind, hipStream_t st)
{
    Rec r;
    r.kind = kind;
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
    (void)hipEventRecord(r.a, st);
    g_recs.push_back(r);
}

void urn_prof_end(hipStream_t st)
{
    if (!g_recs.empty()) (void)hipEventRecord(g_recs.back().b, st);
}

extern "C" int urn_prof_enable(int on)
{
    for (auto &r : g_recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    g_recs.clear();
    g_on = on != 0;
    return URN_OK;
}

extern "C" int urn_prof_read(int kind, double *total_ms, int64_t *launches)
{
    URN_CHECK_ARG(total_ms && launches, "null pointer");
    double t = 0.0;
    int64_t n = 0;
    for (auto &r : g_recs) {
        if (r.kind != kind) continue;
        if (hipEventSynchronize(r.b) != hipSuccess) { urn_set_error("urn_prof_read: event sync failed"); return URN_EHIP; }
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) { urn_set_error("urn_prof_read: elapsed failed"); return URN_EHIP; }
        t += ms;
        ++n;
    }
    *total_ms = t;
    *launches = n;
    return URN_OK;
}
