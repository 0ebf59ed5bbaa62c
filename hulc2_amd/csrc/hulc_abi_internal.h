// hulc_abi_internal.h — error plumbing shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/hulc2_amd.h"

int hulc_fail(int code, const char* msg);          // records msg for hulc_last_error(), returns code
int hulc_check_launch(const char* where);          // hipGetLastError() -> 0 / -100
