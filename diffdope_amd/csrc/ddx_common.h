// ddx_common.h -- shared host/device helpers of libddx (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/ddx.h"

#define DDX_WAVE 64

void ddx_set_error(const char* fmt, ...);
int ddx_compat_flags(void);  // process-wide DDX_COMPAT_* bits (ddx_set_compat)

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
// wave64 sum, result in every lane.  DPP only (VALU): quad swaps, half-row and row mirrors give each lane its
// 16-lane row sum, four v_readlane + three adds fold the rows.  The generic __shfl_xor tree compiles to
// ds_bpermute_b32, i.e. every step goes through the LDS crossbar: 19 sums x 6 steps per wave made the shading
// kernel's epilogue LDS-throughput bound (9 us of a 25 us kernel, measured by ablation).  Fixed association
// order => bit-reproducible.
// One DPP step of the wave reduction applied to N independent values back to back: written value-by-value the compiler
// emits each 4-step chain on its own with an s_nop between dependent DPP ops; step-by-step the next op of a chain is
// N instructions away and needs none.
#define DDX_DPP_STEP(v, ctrl) (v) += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), 0xF, 0xF, true))
template <int N>
__device__ __forceinline__ void wave_sum_n(float (&v)[N])
{
#pragma unroll
    for (int i = 0; i < N; ++i) DDX_DPP_STEP(v[i], 0xB1);   // quad_perm [1,0,3,2]
#pragma unroll
    for (int i = 0; i < N; ++i) DDX_DPP_STEP(v[i], 0x4E);   // quad_perm [2,3,0,1]
#pragma unroll
    for (int i = 0; i < N; ++i) DDX_DPP_STEP(v[i], 0x141);  // row_half_mirror
#pragma unroll
    for (int i = 0; i < N; ++i) DDX_DPP_STEP(v[i], 0x140);  // row_mirror
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v[i]), 0));
        const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v[i]), 16));
        const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v[i]), 32));
        const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v[i]), 48));
        v[i] = (r0 + r1) + (r2 + r3);
    }
}

// The same sums, left in the lanes of the LAST row only (lanes 48..63; other lanes hold partial sums): the two cross-row
// levels as DPP row broadcasts (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3) -- 6 VALU ops per value
// where the readlane version above spends 4 + 4 readlanes + 3 adds + the SGPR->VGPR moves.  Association:
// (r3 + r2) + (r1 + r0) with the rows summed by the same four in-row steps.
template <int N>
__device__ __forceinline__ void wave_sum_n_lastrow(float (&v)[N])
{
#pragma unroll
    for (int i = 0; i < N; ++i) DDX_DPP_STEP(v[i], 0xB1);   // quad_perm [1,0,3,2]
#pragma unroll
    for (int i = 0; i < N; ++i) DDX_DPP_STEP(v[i], 0x4E);   // quad_perm [2,3,0,1]
#pragma unroll
    for (int i = 0; i < N; ++i) DDX_DPP_STEP(v[i], 0x141);  // row_half_mirror
#pragma unroll
    for (int i = 0; i < N; ++i) DDX_DPP_STEP(v[i], 0x140);  // row_mirror
    // (through the builtin a masked row broadcast becomes v_mov 0 / v_mov_dpp / v_add: the combiner only folds full row masks.
    // In-place v_add_f32_dpp leaves the masked-off rows untouched, which is the wanted result.  s_nop 1: a DPP read needs
    // two wait states after the VALU write of its source, and the hazard recogniser does not look into asm.)
#pragma unroll
    for (int i = 0; i < N; ++i)
        v[i] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v[i]), 0x142, 0xA, 0xF, false));
#pragma unroll
    for (int i = 0; i < N; ++i)
        v[i] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v[i]), 0x143, 0xC, 0xF, false));
}

__device__ __forceinline__ float wave_sum(float v)
{
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
    return (r0 + r1) + (r2 + r3);
}
