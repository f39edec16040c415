// Error plumbing and version of libmore4d_hip.so (the kernels live in the .hip files).
#include <stdarg.h>
#include <stdio.h>

#include "more4d_hip.h"

static thread_local char g_err[512] = "";

extern "C" __attribute__((visibility("hidden"))) void m4d_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* m4d_last_error(void) { return g_err; }

extern "C" int m4d_version(void) { return 100; }
