// Shared device/host helpers for libngp_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/ngp_hip.h"

#define NGP_WAVE 64
#define NGP_HALF 32

static inline int ngp_check_launch()
{
    return hipGetLastError() == hipSuccess ? NGP_OK : NGP_ELAUNCH;
}

static inline unsigned ngp_blocks(int64_t work, int per_block)
{
    int64_t b = (work + per_block - 1) / per_block;
    return (unsigned)(b < 1 ? 1 : b);
}

// ---- cross-lane helpers over a 32-lane half-wave (one ray segment per half) ----------
// __shfl_* with width=32 never crosses the half-wave boundary.
__device__ __forceinline__ float half_sum(float v)
{
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 32);
    return v;
}

__device__ __forceinline__ float half_incl_scan_add(float v, int lane32)
{
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) {
        float u = __shfl_up(v, o, 32);
        if (lane32 >= o) v += u;
    }
    return v;
}

__device__ __forceinline__ float half_incl_scan_mul(float v, int lane32)
{
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) {
        float u = __shfl_up(v, o, 32);
        if (lane32 >= o) v *= u;
    }
    return v;
}

// first set bit (0-based) of this half-wave's 32-bit slice of a 64-bit ballot, or -1
__device__ __forceinline__ int half_first(unsigned long long ballot, int lane64)
{
    unsigned m = (unsigned)(ballot >> (lane64 & 32));
    return m ? (__ffs((int)m) - 1) : -1;
}
