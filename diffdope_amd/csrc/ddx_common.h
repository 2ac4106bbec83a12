// ddx_common.h -- shared host/device helpers of libddx (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/ddx.h"

#define DDX_WAVE 64

void ddx_set_error(const char* fmt, ...);

#define DDX_REQUIRE(cond, code, ...)          \
    do {                                      \
        if (!(cond)) {                        \
            ddx_set_error(__VA_ARGS__);       \
            return (code);                    \
        }                                     \
    } while (0)

#define DDX_HIP(call)                                                                  \
    do {                                                                               \
        hipError_t _e = (call);                                                        \
        if (_e != hipSuccess) {                                                        \
            ddx_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__); \
            return (int)_e;                                                            \
        }                                                                              \
    } while (0)

#define DDX_LAUNCH_CHECK()                                                             \
    do {                                                                               \
        hipError_t _e = hipGetLastError();                                             \
        if (_e != hipSuccess) {                                                        \
            ddx_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); \
            return (int)_e;                                                            \
        }                                                                              \
    } while (0)

int ddx_xfm_fwd_strided(const float* points, const float* matrix0, int mstride, int B, int N, float* out, hipStream_t s);

static inline int ddx_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// ---------------------------------------------------------------------------------------------
// wave64 reductions (DPP/permute via __shfl_xor; all 64 lanes participate)
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
