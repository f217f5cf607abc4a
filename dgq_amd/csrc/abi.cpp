// Version + error reporting of libdgq_hip.so (no exceptions cross the C ABI).
#include <cstdarg>
#include <cstdio>
#include "../../include/dgq_hip.h"

static thread_local char g_err[512] = "";

void dgq_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int dgq_version(void) { return DGQ_ABI_VERSION; }
extern "C" const char* dgq_last_error(void) { return g_err; }
