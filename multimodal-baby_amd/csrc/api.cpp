// Error plumbing + version query of libcvcl_hip.so (no exceptions cross the C ABI).
#include <cstdarg>
#include <cstdio>

#include "../../include/cvcl_hip.h"

static thread_local char g_err[512] = "";

void cvcl_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int cvcl_abi_version(void) { return CVCL_ABI_VERSION; }
extern "C" const char* cvcl_last_error(void) { return g_err; }
