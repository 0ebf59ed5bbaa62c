// abi.hip — error reporting for libhulc2_amd.so (see include/hulc2_amd.h for the conventions).
#include "hulc_abi_internal.h"
#include <stdio.h>
#include <string.h>

static thread_local char g_err[512] = "";

int hulc_fail(int code, const char* msg) {
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}

int hulc_check_launch(const char* where) {
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) return 0;
    snprintf(g_err, sizeof(g_err), "%s: launch failed: %s", where, hipGetErrorString(e));
    return -100;
}

extern "C" const char* hulc_last_error(void) { return g_err; }
extern "C" int hulc_abi_version(void) { return 7; }

// (ABI 6) How many cooperative launches share the device from now on (see include/hulc2_amd.h): a host-side setting read when a launch is
// issued — a captured graph keeps the grids it was captured with.
static int g_coop_share = 1;
int hulc_coop_share(void) { return g_coop_share; }
extern "C" int hulc_set_coop_share(int n) {
    if (n != 1 && n != 2 && n != 4) return hulc_fail(-2, "hulc_set_coop_share: 1, 2 or 4");
    const int old = g_coop_share;
    g_coop_share = n;
    return old;
}

// (ABI 6) A brand-new non-blocking stream of the current device (never one a graph capture has used before: see include/hulc2_amd.h).
extern "C" int hulc_stream_create(void** out) {
    if (!out) return hulc_fail(-1, "hulc_stream_create: null pointer");
    hipStream_t s = nullptr;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) { *out = nullptr; return hulc_fail(-100, "hulc_stream_create: hipStreamCreateWithFlags failed"); }
    *out = (void*)s;
    return 0;
}
