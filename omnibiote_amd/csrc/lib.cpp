// Error plumbing and version of libomnibiote_hip.so.
#include "common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void obte_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* obte_last_error(void) { return g_err; }
extern "C" int obte_abi_version(void) { return 1; }
// sizeof of every public argument struct, in header order, so that a binding can check its own layout
extern "C" int obte_struct_sizes(int64_t* out, int cap) {
    const int64_t v[] = {(int64_t)sizeof(obte_gemm_args), (int64_t)sizeof(obte_attn_fwd_args), (int64_t)sizeof(obte_attn_bwd_args),
                         (int64_t)sizeof(obte_mt_args), (int64_t)sizeof(obte_block_desc)};
    const int n = (int)(sizeof(v) / sizeof(v[0]));
    for (int i = 0; i < n && i < cap; ++i) out[i] = v[i];
    return n;
}

// ---- device status word ---------------------------------------------------------------------------------------
// Kernels that can detect a failure they cannot recover from (today: a bounded spin of the one-kernel attention backward's
// hand-off chain that gave up) OR a bit into ONE word of pinned, device-visible host memory, system scope.  The host reads it
// with a plain load at any point where the work in question has been synchronised with (obte_device_status): no copy, no
// dependence on which workspace a call used, and a failure is never silent.  A second word is the fault-injection request of
// the tests (obte_fault_inject): read on the host only and handed to the launches that honour it.
#include <atomic>
#include <mutex>
namespace {
std::once_flag g_status_once;
int32_t* g_status = nullptr;          // [0] status bits, kernels OR into it; lives for the life of the process
std::atomic<int> g_fault_inject{0};
}
int32_t* obte_status_word() {
    std::call_once(g_status_once, [] {
        void* h = nullptr;
        if (hipHostMalloc(&h, 256, hipHostMallocMapped | hipHostMallocPortable) == hipSuccess && h) {
            memset(h, 0, 256);
            g_status = (int32_t*)h;
        }
    });
    return g_status;
}
int obte_fault_injection() { return g_fault_inject.load(std::memory_order_relaxed); }
extern "C" int obte_device_status(int clear) {
    int32_t* w = obte_status_word();
    if (!w) { obte_set_error("obte_device_status: the status word could not be allocated"); return OBTE_ELAUNCH; }
    const int v = __atomic_load_n(w, __ATOMIC_ACQUIRE);
    if (clear && v) __atomic_and_fetch(w, ~v, __ATOMIC_ACQ_REL);
    if (v & OBTE_STATUS_ATTN_BWD_HANDOFF)
        obte_set_error("device status 0x%x: the attention backward's dQ hand-off chain timed out (a workgroup of a (batch, head) never signalled): "
                       "the gradients of that launch are invalid", v);
    else if (v) obte_set_error("device status 0x%x", v);
    return v;
}
extern "C" int obte_fault_inject(int what) { return g_fault_inject.exchange(what); }

// ---- opt-in launch profiler ---------------------------------------------------------------------------------
#include <mutex>
#include <vector>
namespace {
struct ProfRec { hipEvent_t a, b; int64_t d0, d1, d2; int kind; bool closed; };
std::mutex g_prof_mu;
std::vector<ProfRec> g_prof;
bool g_prof_on = false;
}

int obte_prof_begin(hipStream_t st, int kind, int64_t d0, int64_t d1, int64_t d2) {
    if (!g_prof_on) return -1;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    ProfRec r{};
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return -1;
    r.d0 = d0; r.d1 = d1; r.d2 = d2; r.kind = kind; r.closed = false;
    (void)hipEventRecord(r.a, st);
    g_prof.push_back(r);
    return (int)g_prof.size() - 1;
}

void obte_prof_end(int idx, hipStream_t st) {
    if (idx < 0) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (idx >= (int)g_prof.size()) return;
    (void)hipEventRecord(g_prof[idx].b, st);
    g_prof[idx].closed = true;
}

extern "C" int obte_profile_enable(int on) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& r : g_prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    g_prof.clear();
    g_prof_on = on != 0;
    return OBTE_OK;
}

extern "C" int obte_profile_collect(double* ms, int64_t* dims, int32_t* kind, int cap) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    int n = 0;
    for (auto& r : g_prof) {
        if (r.closed && n < cap) {
            float t = 0.f;
            (void)hipEventSynchronize(r.b);
            if (hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) {
                ms[n] = t; dims[3 * n] = r.d0; dims[3 * n + 1] = r.d1; dims[3 * n + 2] = r.d2; kind[n] = r.kind;
                ++n;
            }
        }
        (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b);
    }
    g_prof.clear();
    return n;
}
