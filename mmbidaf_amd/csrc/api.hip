// Error plumbing and version entry points of libmmbidaf_hip.so.
#include <stdarg.h>

#include <mutex>
#include <vector>

#include "common.h"

namespace mmb {

char* err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

// ---------------------------------------------------------------- opt-in kernel timing
struct EvPair {
    hipEvent_t a, b;
};
static std::mutex g_prof_mu;
static uint32_t g_prof_mask = 0;
static std::vector<EvPair> g_prof_done[MMB_K_COUNT];
static std::vector<EvPair> g_prof_pool;

ProfScope::ProfScope(int id_, hipStream_t s) : id(id_), stream(s), slot(nullptr) {
    if (!(g_prof_mask & (1u << id))) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    EvPair* p = new EvPair;
    if (!g_prof_pool.empty()) {
        *p = g_prof_pool.back();
        g_prof_pool.pop_back();
    } else if (hipEventCreate(&p->a) != hipSuccess || hipEventCreate(&p->b) != hipSuccess) {
        delete p;
        return;
    }
    (void)hipEventRecord(p->a, stream);
    slot = p;
}

ProfScope::~ProfScope() {
    if (!slot) return;
    EvPair* p = static_cast<EvPair*>(slot);
    (void)hipEventRecord(p->b, stream);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_done[id].push_back(*p);
    delete p;
}

}  // namespace mmb

extern "C" int mmb_profile_enable(uint32_t kernel_mask) {
    std::lock_guard<std::mutex> lk(mmb::g_prof_mu);
    mmb::g_prof_mask = kernel_mask;
    return MMB_OK;
}

extern "C" int mmb_profile_read(int kernel_id, double* total_ms, int* launches) {
    MMB_REQUIRE(kernel_id >= 0 && kernel_id < MMB_K_COUNT && total_ms && launches, "mmb_profile_read: bad argument");
    std::lock_guard<std::mutex> lk(mmb::g_prof_mu);
    double tot = 0.0;
    int n = 0;
    for (auto& p : mmb::g_prof_done[kernel_id]) {
        float ms = 0.f;
        if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            tot += ms;
            ++n;
        }
        mmb::g_prof_pool.push_back(p);
    }
    mmb::g_prof_done[kernel_id].clear();
    *total_ms = tot;
    *launches = n;
    return MMB_OK;
}

extern "C" const char* mmb_kernel_name(int kernel_id) {
    static const char* names[MMB_K_COUNT] = {
        "att_rank1_kernel", "att_col_kernel", "att_combine_kernel", "att_row_kernel",
        "att_bwd_pre_kernel", "att_bwd_j1_kernel", "att_bwd_j2_kernel", "att_bwd_jfin_kernel", "att_bwd_i_kernel",
        "gemm_kernel", "lstm_rec_fwd_kernel", "lstm_rec_bwd_kernel", "split_"};
    return (kernel_id >= 0 && kernel_id < MMB_K_COUNT) ? names[kernel_id] : "";
}

extern "C" int mmb_version(void) { return MMB_VERSION; }
extern "C" const char* mmb_last_error(void) { return mmb::err_buf(); }
