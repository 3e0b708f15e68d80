// Shared helpers for liburesnet_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/uresnet_hip.h"

void urn_set_error(const char *fmt, ...);

#define URN_CHECK_ARG(cond, msg)                                   \
    do {                                                           \
        if (!(cond)) {                                             \
            urn_set_error("%s: %s", __func__, msg);                \
            return URN_EINVAL;                                     \
        }                                                          \
    } while (0)

#define URN_LAUNCH_CHECK()                                                         \
    do {                                                                           \
        hipError_t e_ = hipGetLastError();                                         \
        if (e_ != hipSuccess) {                                                    \
            urn_set_error("%s: launch failed: %s", __func__, hipGetErrorString(e_)); \
            return URN_EHIP;                                                       \
        }                                                                          \
    } while (0)

static inline int urn_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// 64-bit coordinate key: batch | x | y | z, 16 bits each (coords < 32768).
// The all-0x7F pattern is reserved for "empty" (batch id 0x7F7F is rejected).
#define URN_EMPTY_KEY 0x7F7F7F7F7F7F7F7Full
#define URN_EMPTY_VAL 0x7F7F7F7F

__device__ __forceinline__ uint64_t urn_key(int x, int y, int z, int b)
{
    return ((uint64_t)(uint16_t)b << 48) | ((uint64_t)(uint16_t)x << 32) |
           ((uint64_t)(uint16_t)y << 16) | (uint64_t)(uint16_t)z;
}

__device__ __forceinline__ uint64_t urn_mix(uint64_t k)
{
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
    return k;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
