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
extern thread_local void* mgp_tls_launch_events[2];   // defined in capi.hip; set by mgp_set_launch_events()

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

// One element of torch.optim.Adam (defaults; reference gnn_dagger.py:49,93): exp_avg.lerp_(g, 1-b1);
// exp_avg_sq.mul_(b2).addcmul_(g, g, 1-b2); p.addcdiv_(exp_avg, sqrt(exp_avg_sq)/sqrt(bc2) + eps, -lr/bc1).
// The roundings are spelled out (explicit fused multiply-adds, nothing left to -ffp-contract) so that every kernel that
// applies the step -- mgp_adam_step*, the fused reduce of mgp_train_step*, its data-parallel forms -- produces the same
// bits from the same gradient.
__device__ __forceinline__ void mgp_adam_elem(float& p, float& m, float& v, float g, float one_m_b1, float b2,
                                              float one_m_b2, float step_size, float bc2_sqrt, float eps)
{
    const float mi = __fmaf_rn(g - m, one_m_b1, m);
    const float vi = __fmaf_rn(__fmul_rn(one_m_b2, g), g, __fmul_rn(v, b2));
    const float denom = __fadd_rn(__fdiv_rn(__fsqrt_rn(vi), bc2_sqrt), eps);
    p = __fmaf_rn(-step_size, __fdiv_rn(mi, denom), p);
    m = mi;
    v = vi;
}

// Raise a kernel's dynamic-LDS limit ONCE per (device, kernel, size) and thread: hipFuncSetAttribute is a driver call -- measured
// at up to 0.35 ms per call in some processes, during which the runtime also held back the submission of launches already
// enqueued (a 100-step loop of launches with one call each sat on the host for 35 ms before the first kernel started).
// The attribute belongs to the CURRENT device's function object, so the device is part of the key (a process that drives a
// second GPU from the same thread must set it there too); a full table falls back to the plain driver call, never to a skip.
inline hipError_t mgp_allow_dyn_lds(const void* fn, size_t lds)
{
    struct Entry { const void* fn; size_t lds; int dev; };
    constexpr int CAP = 192;
    static thread_local Entry tab[CAP];
    static thread_local int used = 0;
    if (lds <= 48 * 1024) return hipSuccess;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = -1; }
    if (dev >= 0)
        for (int i = 0; i < used; ++i)
            if (tab[i].fn == fn && tab[i].dev == dev) {
                if (tab[i].lds >= lds) return hipSuccess;
                const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                if (e == hipSuccess) tab[i].lds = lds;
                return e;
            }
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess && dev >= 0 && used < CAP) { tab[used].fn = fn; tab[used].lds = lds; tab[used].dev = dev; ++used; }
    return e;
}
