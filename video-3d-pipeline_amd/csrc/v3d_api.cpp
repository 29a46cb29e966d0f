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
extern "C" const char* v3d_version(void) { return "libv3d_hip 0.2 (gfx950)"; }

// gf_band1, gf_band2, gf_tiled, gf_fused, gf_band, gf_cols, gf_int1, corr_gather, corr_fused.
// gf_band: 432 rows = 5 bands of a 4K frame: 34 frames (one lock-step launch upstream) are 5.98 rounds of the 512 resident
// workgroups, where round 2's 270 rows were 9.56 (2.56 -> 2.43 ms).  A FIXED height, not the per-launch optimum (value 0):
// the second stage's sliding sums round differently for a different band origin, and a frame's bits must not depend on how
// many frames share its launch.
v3d_lib_options g_v3d_opt = { 90, 270, 0, 1, 432, 256, 1, 0, 1 };

extern "C" int v3d_set_option(const char* key, int value)
{
    if (!key) { v3d_set_error("null key"); return V3D_ERR_ARG; }
    if (!strcmp(key, "gf_band1") || !strcmp(key, "gf_band2") || !strcmp(key, "gf_band")) {
        if ((value < 8 && !(value == 0 && key[7] == '\0')) || value > 65536) { v3d_set_error("option %s: value %d out of range", key, value); return V3D_ERR_ARG; }
        (key[7] == '1' ? g_v3d_opt.gf_band1 : key[7] == '2' ? g_v3d_opt.gf_band2 : g_v3d_opt.gf_band) = value;
    } else if (!strcmp(key, "gf_tiled")) g_v3d_opt.gf_tiled = value != 0;
    else if (!strcmp(key, "gf_fused")) g_v3d_opt.gf_fused = value != 0;
    else if (!strcmp(key, "gf_cols")) {
        if (value != 256 && value != 512) { v3d_set_error("option gf_cols: 256 or 512"); return V3D_ERR_ARG; }
        g_v3d_opt.gf_cols = value;
    }
    else if (!strcmp(key, "gf_int1")) g_v3d_opt.gf_int1 = value != 0;
    else if (!strcmp(key, "corr_gather")) g_v3d_opt.corr_gather = value != 0;
    else if (!strcmp(key, "corr_fused")) g_v3d_opt.corr_fused = value != 0;
    else { v3d_set_error("unknown option %s", key); return V3D_ERR_ARG; }
    return V3D_OK;
}

extern "C" int v3d_get_option(const char* key, int* value)
{
    if (!key || !value) { v3d_set_error("null argument"); return V3D_ERR_ARG; }
    if (!strcmp(key, "gf_band1")) *value = g_v3d_opt.gf_band1;
    else if (!strcmp(key, "gf_band2")) *value = g_v3d_opt.gf_band2;
    else if (!strcmp(key, "gf_band")) *value = g_v3d_opt.gf_band;
    else if (!strcmp(key, "gf_tiled")) *value = g_v3d_opt.gf_tiled;
    else if (!strcmp(key, "gf_fused")) *value = g_v3d_opt.gf_fused;
    else if (!strcmp(key, "gf_cols")) *value = g_v3d_opt.gf_cols;
    else if (!strcmp(key, "gf_int1")) *value = g_v3d_opt.gf_int1;
    else if (!strcmp(key, "corr_gather")) *value = g_v3d_opt.corr_gather;
    else if (!strcmp(key, "corr_fused")) *value = g_v3d_opt.corr_fused;
    else { v3d_set_error("unknown option %s", key); return V3D_ERR_ARG; }
    return V3D_OK;
}
