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
