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

#ifdef __HIPCC__
// v0 = sum_k p[(2k) * ld], v1 = sum_k p[(2k + 1) * ld] over the `slots` rows of an accumulated-statistics slab, added in slot
// order (the result does not depend on how the loads are scheduled).  The loads of eight slots are issued together: a plain
// loop waits for every pair of loads before it asks for the next (ISA: load, load, s_waitcnt vmcnt(1), add, vmcnt(0), add,
// branch) -- eight dependent round trips, 4-5 us at the head of every kernel that derives a BatchNorm's coefficients.
__device__ __forceinline__ void urn_slab_sum2(const double *p, long ld, int slots, double &v0, double &v1)
{
    v0 = 0.0; v1 = 0.0;
    int k = 0;
    for (; k + 8 <= slots; k += 8) {
        double a[8], b[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { a[j] = p[(long)(2 * (k + j)) * ld]; b[j] = p[(long)(2 * (k + j) + 1) * ld]; }
#pragma unroll
        for (int j = 0; j < 8; ++j) { v0 += a[j]; v1 += b[j]; }
    }
    for (; k < slots; ++k) { v0 += p[(long)(2 * k) * ld]; v1 += p[(long)(2 * k + 1) * ld]; }
}

// 16-bit weight fragments (urn_gconv_args.wt_frag_prec != 0): where the 8 bytes of lane `lane` of block (.., kb) go, in
// 8-byte units.  e = linear index with the 16-channel group kb fastest, then the lane (the fp32 order scaled down).  With an
// EVEN number of groups two consecutive groups share a kilobyte, a lane's 16 bytes = [group kb | group kb + 1]: one
// 16-byte load fetches both -- they are the eight contraction slots of ONE v_mfma_f32_16x16x32_* (the pair-list loop is
// bound by the vector-memory INSTRUCTIONS a CU can issue, and the weight blocks were 32 of its 43 per block at 128 -> 64x4).
__host__ __device__ __forceinline__ long urn_frag16_slot(long e, int kbn)
{
    if (kbn & 1) return e;
    const long blk = e >> 6;               // (.., kb)
    const int lane = (int)(e & 63);
    return ((blk & ~1L) << 6) + 2 * lane + (blk & 1);   // kbn even: blk even <=> kb even
}

// four floats rounded (RNE) to bf16 (PREC 1) or fp16 (PREC 2), packed into 8 bytes: one lane's operand of v_mfma_f32_16x16x16_*
template <int PREC>
__device__ __forceinline__ uint2 urn_round16x4(f32x4 v)
{
    if constexpr (PREC == 1) {
        typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
        return __builtin_bit_cast(uint2, __builtin_convertvector(v, bf16x4));
    } else {
        typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
        return __builtin_bit_cast(uint2, __builtin_convertvector(v, h16x4));
    }
}
#endif
