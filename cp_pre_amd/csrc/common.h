// Shared host/device helpers for libcp_pre_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/cp_pre_hip.h"

#define PRE_LAUNCH_CHECK()                         \
    do {                                           \
        hipError_t e__ = hipGetLastError();        \
        if (e__ != hipSuccess) return (int)e__;    \
    } while (0)

// The dispatcher deals workgroups round-robin over the 8 XCDs (blocks b and b+8 share
// an L2).  Remap so that each XCD walks a CONTIGUOUS range of logical tiles: neighbouring
// tiles (which share halo rows) then meet in one L2.  Speed only, never correctness.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblocks)
{
    const unsigned nx = 8u;
    const unsigned per = nblocks / nx, rem = nblocks % nx;
    const unsigned xcd = bid % nx, idx = bid / nx;
    // XCDs [0, rem) own per+1 tiles, the rest own per tiles
    const unsigned start = xcd * per + (xcd < rem ? xcd : rem);
    return start + idx;
}

// Order-preserving map fp32 -> uint32 (ascending floats <-> ascending uints), and back.
__device__ __forceinline__ uint32_t f2key(float f)
{
    const uint32_t u = __float_as_uint(f);
    return u ^ ((uint32_t)((int32_t)u >> 31) | 0x80000000u);      // negative: flip all bits; else: set the sign bit
}
__device__ __forceinline__ float key2f(uint32_t k)
{
    uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}

static inline hipStream_t as_stream(void *s) { return (hipStream_t)s; }
