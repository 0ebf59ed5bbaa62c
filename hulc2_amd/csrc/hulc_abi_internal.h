// hulc_abi_internal.h — error plumbing shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/hulc2_amd.h"

int hulc_fail(int code, const char* msg);          // records msg for hulc_last_error(), returns code
int hulc_check_launch(const char* where);          // hipGetLastError() -> 0 / -100
int hulc_coop_share(void);                         // hulc_set_coop_share's current value (1: a cooperative launch may take every CU)

// wgrad_taps.hip: the conv_taps_wp items of hulc_wgrad_group that run as nine-tap tiles (one pass over the operands)
int hulc_wgrad_taps_takes(const hulc_wgrad_item* d);
long hulc_wgrad_taps_workspace(const hulc_wgrad_item* const* items, int n);       // slab bytes
int hulc_wgrad_taps_launch(const hulc_wgrad_item* const* items, int n, void* slabs, hipStream_t s);
