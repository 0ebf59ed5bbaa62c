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
extern "C" int hulc_abi_version(void) { return 5; }
