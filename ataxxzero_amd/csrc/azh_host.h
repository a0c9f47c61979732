// Host-side helpers shared by the translation units of libataxxzero_hip.so.
#pragma once

#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#define AZH_NO_REFERENCE_ABI
#include "../../include/ataxxzero_hip.h"

// Record the message for azh_last_error() and return `code`.
int azh_fail(int code, const char *fmt, ...);
// 0 when a HIP device is usable; otherwise records why and returns non-zero.
// There is no CPU fallback: every compute entry point fails loudly without a GPU.
int azh_require_device(void);

#define AZH_HIP(expr)                                                                       \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess)                                                               \
            return azh_fail(-100 - (int)_e, "%s failed at %s:%d: %s", #expr, __FILE__,     \
                            __LINE__, hipGetErrorString(_e));                               \
    } while (0)

// net_kernels.hip
int azh_net_launch(azh_net *net, int dtype, const unsigned long long *d_boards, const int *d_list,
                   const int *d_count, int max_n, unsigned long long blockers, float *d_logits,
                   float *d_values, hipStream_t stream, unsigned long long *d_stamps = nullptr, int thin = 0,
                   const void *advance_hook = nullptr);
// advance_hook (all three launches): an azh::AdvanceHook (engine_device.h) or null — the device-resident search loop's queued
// moves are played by the first `workers` workgroups of the launch
// symmetry-averaged evaluation (nn_evals.py:48-62); scratch: [8 max_n][833] and [8 max_n] floats
int azh_net_launch_sym(azh_net *net, int dtype, const unsigned long long *d_boards, const int *d_list,
                       const int *d_count, int max_n, unsigned long long blockers, float *d_tmp_logits,
                       float *d_tmp_values, float *d_logits, float *d_values, hipStream_t stream, int thin = 0,
                       const void *advance_hook = nullptr);
// two nets, two dense leaf lists, one launch (arena); 1 = not applicable here, nothing launched (launch them one by one)
int azh_net_launch_pair(azh_net *net_a, azh_net *net_b, int dtype, const unsigned long long *d_boards, const int *d_list_a,
                        const int *d_count_a, const int *d_list_b, const int *d_count_b, int max_n,
                        unsigned long long blockers, float *d_logits, float *d_values, hipStream_t stream, int thin = 0,
                        const void *advance_hook = nullptr);
