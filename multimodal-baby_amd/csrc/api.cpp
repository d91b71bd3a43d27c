// Error plumbing, version query and the optional per-kernel-class HIP-event timer of libcvcl_hip.so
// (no exceptions cross the C ABI).
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../include/cvcl_hip.h"

static thread_local char g_err[512] = "";

void cvcl_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

bool cvcl_env_on(const char* name) {
    const char* e = getenv(name);
    return !(e && e[0] == '0');
}
int cvcl_lab_int(const char* name, int dflt) {
#ifdef CVCL_LAB
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
#else
    (void)name;
    return dflt;
#endif
}

// co-scheduling hint for the one-workgroup-per-CU GEMM kernels (cvcl_hip.h): how many CUs a launch may fill; 0 = all
static int g_gemm_cu_share = 0;
int cvcl_gemm_cu_share() { return g_gemm_cu_share; }
extern "C" int cvcl_set_gemm_cu_share(int cus) {
    const int prev = g_gemm_cu_share;
    g_gemm_cu_share = cus > 0 ? (cus & ~7) : 0;
    return prev;
}

extern "C" int cvcl_abi_version(void) { return CVCL_ABI_VERSION; }
extern "C" const char* cvcl_last_error(void) { return g_err; }

// ---- per-kernel-class timing with HIP events recorded on the stream each kernel is launched on ----
namespace {
struct Rec { hipEvent_t a, b; int cls; };
bool g_prof = false;
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
hipEvent_t get_event() {
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
}  // namespace

bool cvcl_prof_on() { return g_prof; }
void* cvcl_prof_begin(void* stream, int cls) {
    if (!g_prof) return nullptr;
    Rec r{get_event(), get_event(), cls};
    if (!r.a || !r.b) return nullptr;
    (void)hipEventRecord(r.a, (hipStream_t)stream);
    g_recs.push_back(r);
    return (void*)(uintptr_t)g_recs.size();          // 1-based handle
}
void cvcl_prof_end(void* handle, void* stream) {
    if (!handle) return;
    (void)hipEventRecord(g_recs[(uintptr_t)handle - 1].b, (hipStream_t)stream);
}

extern "C" int cvcl_prof_enable(int on) {
    for (auto& r : g_recs) { g_pool.push_back(r.a); g_pool.push_back(r.b); }
    g_recs.clear();
    g_prof = on != 0;
    return CVCL_OK;
}

// the part of an event bracket that is not the kernel: the same two event records around a kernel that does nothing
__global__ void cvcl_null_kernel() {}
extern "C" int cvcl_prof_null_bracket_us(void* stream, int n, double* avg_us) {
    if (!avg_us || n <= 0) { cvcl_set_error("cvcl_prof_null_bracket_us: bad args"); return CVCL_EINVAL; }
    std::vector<Rec> recs;
    for (int i = 0; i < n; ++i) {
        Rec r{get_event(), get_event(), 0};
        if (!r.a || !r.b) { cvcl_set_error("cvcl_prof_null_bracket_us: no events"); return CVCL_ELAUNCH; }
        (void)hipEventRecord(r.a, (hipStream_t)stream);
        hipLaunchKernelGGL(cvcl_null_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream);
        (void)hipEventRecord(r.b, (hipStream_t)stream);
        recs.push_back(r);
    }
    double tot = 0.0;
    int cnt = 0;
    for (auto& r : recs) {
        float ms = 0.f;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) { tot += ms; ++cnt; }
        g_pool.push_back(r.a); g_pool.push_back(r.b);
    }
    *avg_us = cnt ? tot / cnt * 1e3 : 0.0;
    return CVCL_OK;
}

extern "C" int cvcl_prof_collect(double* ms_per_class, long* launches_per_class, int n_classes) {
    if (!ms_per_class || !launches_per_class || n_classes <= 0) { cvcl_set_error("cvcl_prof_collect: bad args"); return CVCL_EINVAL; }
    for (int i = 0; i < n_classes; ++i) { ms_per_class[i] = 0.0; launches_per_class[i] = 0; }
    for (auto& r : g_recs) {
        if (hipEventSynchronize(r.b) != hipSuccess) { cvcl_set_error("cvcl_prof_collect: event sync failed"); return CVCL_ELAUNCH; }
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) continue;
        if (r.cls >= 0 && r.cls < n_classes) { ms_per_class[r.cls] += ms; launches_per_class[r.cls] += 1; }
    }
    return CVCL_OK;
}
