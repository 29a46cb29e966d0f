// v3d_api.cpp -- error reporting and version string of libv3d_hip.
#include "v3d_common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void v3d_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* v3d_last_error(void) { return g_err; }
extern "C" const char* v3d_version(void) { return "libv3d_hip 0.1 (gfx950)"; }
