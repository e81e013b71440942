// se_core.hip -- library-level entry points of include/sehip.h (version, error text).
#include "se_common.h"
#include <atomic>
#include <mutex>

namespace se {

// ---- phase timing: a measuring aid of the product library (bench.py's per-leg rooflines), off unless se_phase_timing(1) ----
constexpr int PT_CAP = 96;
static std::atomic<int> pt_on{0};
static std::mutex pt_mu;                       // guards everything below
static int pt_n = 0;
static hipEvent_t pt_ev[PT_CAP];
static bool pt_ev_made[PT_CAP];
static const char *pt_name[PT_CAP];
static unsigned *pt_host = nullptr;            // pinned, library-owned: the 4 statistics words of the last se_retrieve_topk, copied on ITS stream
static hipEvent_t pt_cnt_ev;                   // ... recorded behind that copy
static bool pt_cnt_made = false, pt_cnt_valid = false;
static long long pt_rows = 0;

bool phase_timing_on() { return pt_on.load(std::memory_order_relaxed) != 0; }

void phase_mark(const char *name, hipStream_t s)
{
    if (!phase_timing_on()) return;
    std::lock_guard<std::mutex> lk(pt_mu);
    if (pt_n >= PT_CAP) return;
    if (!pt_ev_made[pt_n]) {
        if (hipEventCreate(&pt_ev[pt_n]) != hipSuccess) return;
        pt_ev_made[pt_n] = true;
    }
    if (hipEventRecord(pt_ev[pt_n], s) != hipSuccess) return;
    pt_name[pt_n++] = name;
}

// The statistics words live in the CALLER's workspace, which may be freed or reused before se_phase_timing_read runs (round-5
// advisor finding: the read used to copy from that pointer): they are copied here, on the call's own stream, into a pinned buffer
// the library owns.
void phase_note_counters(const unsigned *dev_counters, long long rows, hipStream_t s)
{
    if (!phase_timing_on()) return;
    std::lock_guard<std::mutex> lk(pt_mu);
    pt_cnt_valid = false;
    if (!pt_host && hipHostMalloc((void **)&pt_host, 4 * sizeof(unsigned), hipHostMallocDefault) != hipSuccess) { pt_host = nullptr; return; }
    if (!pt_cnt_made) {
        if (hipEventCreate(&pt_cnt_ev) != hipSuccess) return;
        pt_cnt_made = true;
    }
    if (hipMemcpyAsync(pt_host, dev_counters, 4 * sizeof(unsigned), hipMemcpyDeviceToHost, s) != hipSuccess) return;
    if (hipEventRecord(pt_cnt_ev, s) != hipSuccess) return;
    pt_rows = rows;
    pt_cnt_valid = true;
}

char *err_buf()
{
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace se

extern "C" int se_version(void) { return 310; /* 0.3.1: se_phase_timing / se_phase_timing_read */ }
extern "C" const char *se_last_error(void) { return se::err_buf(); }
extern "C" const char *se_build_arch(void) { return "gfx950"; }

extern "C" int se_phase_timing(int on)
{
    std::lock_guard<std::mutex> lk(se::pt_mu);
    se::pt_n = 0;
    se::pt_cnt_valid = false;
    se::pt_on.store(on ? 1 : 0, std::memory_order_relaxed);
    return SE_OK;
}

extern "C" int se_phase_timing_read(const char **names_host, float *ms_host, int cap, int64_t *counters_host)
{
    std::lock_guard<std::mutex> lk(se::pt_mu);
    struct Reset { ~Reset() { se::pt_n = 0; se::pt_cnt_valid = false; } } reset_on_every_exit;   // (also when a HIP call below fails)
    int out = 0;
    if (se::pt_n > 0) SE_HIP_CHECK(hipEventSynchronize(se::pt_ev[se::pt_n - 1]));
    for (int i = 1; i < se::pt_n && out < cap; i++) {
        float ms = 0.f;
        SE_HIP_CHECK(hipEventElapsedTime(&ms, se::pt_ev[i - 1], se::pt_ev[i]));
        if (names_host) names_host[out] = se::pt_name[i];
        if (ms_host) ms_host[out] = ms;
        out++;
    }
    if (counters_host) {
        counters_host[0] = counters_host[1] = counters_host[2] = counters_host[3] = counters_host[4] = -1;
        if (se::pt_cnt_valid) {
            SE_HIP_CHECK(hipEventSynchronize(se::pt_cnt_ev));
            for (int i = 0; i < 4; i++) counters_host[i] = (int64_t)se::pt_host[i];
            counters_host[4] = (int64_t)se::pt_rows;
        }
    }
    return out;
}
