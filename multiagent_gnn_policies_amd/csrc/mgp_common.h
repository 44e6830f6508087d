// Shared device/host helpers for libmgp (gfx950 only: wave = 64 lanes, 256 CUs, 160 KiB LDS/CU).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mgp.h"

#define MGP_WAVE 64

#define MGP_CHECK_PTR(p)                                   \
    do {                                                   \
        if ((p) == nullptr) return MGP_EINVAL;             \
        if ((reinterpret_cast<uintptr_t>(p) & 3u) != 0)    \
            return MGP_EALIGN;                             \
    } while (0)

#define MGP_CHECK_PTR8(p)                                  \
    do {                                                   \
        if ((p) == nullptr) return MGP_EINVAL;             \
        if ((reinterpret_cast<uintptr_t>(p) & 7u) != 0)    \
            return MGP_EALIGN;                             \
    } while (0)

// hipGetLastError() is sticky per thread: the host framework may leave e.g. hipErrorNotReady behind
// (event queries).  Every entry point clears it before launching so mgp_launch_status() reports only
// this call's launches.
static inline void mgp_clear_error() { (void)hipGetLastError(); }

extern thread_local int mgp_tls_hip_error;      // defined in capi.hip; read by mgp_last_hip_error()

static inline int mgp_launch_status() {
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return MGP_OK;
    mgp_tls_hip_error = (int)e;
    return MGP_ELAUNCH;
}

static inline bool mgp_aligned16(const void* p) {
    return (reinterpret_cast<uintptr_t>(p) & 15u) == 0;
}

static inline int mgp_ceil_div(int a, int b) { return (a + b - 1) / b; }

// Sum across the 64 lanes of a wave; every lane gets the total.
__device__ __forceinline__ float mgp_wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, MGP_WAVE);
    return v;
}
__device__ __forceinline__ double mgp_wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, MGP_WAVE);
    return v;
}
