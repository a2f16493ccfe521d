// common.h -- shared host-side helpers of the C-ABI library (not part of the ABI)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/x265_amd.h"

namespace x265amd {

inline int hip_status(hipError_t e) { return e == hipSuccess ? X265AMD_OK : X265AMD_ERR_HIP - (int)e; }

#define X265AMD_HIP(expr)                                   \
    do {                                                    \
        hipError_t e_ = (expr);                             \
        if (e_ != hipSuccess) return ::x265amd::hip_status(e_); \
    } while (0)

// set by x265amd_init(); entry points refuse to launch before it
bool initialised();
int lds_limit_bytes();
int cu_count();

inline int ilog2(int n)
{
    int l = 0;
    while ((1 << l) < n) l++;
    return l;
}

}  // namespace x265amd
