// Error plumbing and version of libtimediff_hip.so.
#include <stdarg.h>
#include <stdio.h>

#include "../../include/timediff_hip.h"

static thread_local char g_err[512] = "";

void td_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

extern "C" const char *td_last_error(void) { return g_err; }
extern "C" int td_version(void) { return 100; }
