// Graph-shift aggregation of ONE (tap, column block) on the matrix pipe, by one wave -- shared by the fused Actor forward
// (actor_fused.hip: actor_fwd_mfma_kernel) and the stand-alone aggregation (agg.hip: agg_fwd_mfma_kernel).
//   Y[c, n] = sum_m X[c, m] * G[m, n]   for the block's <= 16 four-column groups, ALL N contraction rows (N % 4 == 0,
//   N <= 4 S), c < F <= 4 FH.
// Lane (li, lq) loads the float4 G[row][4 g + 0..3] (g = its column group) with row = 16 (s >> 2) + 4 lq + (s & 3) for every
// row step s up front -- the whole operator share is requested before anything else, issue order = consumption order
// (vmcnt retires in order; the scheduler is kept from reversing the batch) -- and its X operands of four consecutive steps as
// one aligned float4.  v_mfma_f32_4x4x1 (16 independent 4x4 outer products per instruction; lanes 4 q .. 4 q + 3 are block q,
// D[i] of lane l = A(lane 4 (l >> 2) + i) * B(lane l)): block = (lq, li >> 2) holds one row and four adjacent column groups;
// instruction (t, h) multiplies features 4 h + 0..3 (A: the lane with li & 3 == i supplies X[4 h + i][row]) into float t of
// every lane's G quad, i.e. column 4 g + t.  6 of the 8 A rows carry features at F = 6 (the 16x16x4 shape would use 6 of
// 16: twice the pipe time).  A lane accumulates the sum over ITS row class lq; the four classes are added through a
// per-wave LDS area (`red`: 4 float4 per lane, value-major so every b128 access of the wave is contiguous) in fixed
// order, one feature half at a time (a wave's LDS operations execute in order).  emit(h, tot) then receives, in lane
// (li, lq), channels 4 h + 0..3 of column 4 g + lq.
#pragma once
#include "mgp_device.h"

#ifndef AGG_STAMP
#define AGG_STAMP(i) do { } while (0)
#endif

namespace {

template <int S, int FH, class Emit>
__device__ __forceinline__ void agg_mfma_unit(const float* __restrict__ Gk, const float* __restrict__ Xk, long sxc, int F,
                                              int N, int lane, f32x4* red, Emit emit)
{
    const int li = lane & 15, lq = lane >> 4;
    f32x4 xa[FH][S / 4];
    f32x4 gv[S];
#pragma unroll
    for (int h = 0; h < FH; ++h) {
        const float* xr = Xk + (size_t)min(4 * h + (li & 3), F - 1) * sxc;
#pragma unroll
        for (int t4 = 0; t4 < S / 4; ++t4)
            xa[h][t4] = *reinterpret_cast<const f32x4*>(xr + min(16 * t4 + 4 * lq, N - 4));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < S; ++s) {
        gv[s] = *reinterpret_cast<const f32x4*>(Gk + (size_t)min(16 * (s >> 2) + 4 * lq + (s & 3), N - 1) * N);
        __builtin_amdgcn_sched_barrier(0);
    }
    AGG_STAMP(1);
    f32x4 acc[4][FH];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int h = 0; h < FH; ++h) acc[t][h] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < S; ++s) {                             // rows past N: clamped addresses times a = 0
        const bool rok = 16 * (s >> 2) + 4 * lq + (s & 3) < N;
#pragma unroll
        for (int h = 0; h < FH; ++h) {
            const float a = (rok && 4 * h + (li & 3) < F) ? xa[h][s >> 2][s & 3] : 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t)
                acc[t][h] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, gv[s][t], acc[t][h], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    AGG_STAMP(2);
#pragma unroll
    for (int h = 0; h < FH; ++h) {
#pragma unroll
        for (int t = 0; t < 4; ++t) red[t * 64 + lane] = acc[t][h];
        const f32x4* p = red + lq * 64 + li;
        f32x4 tot = p[0];
#pragma unroll
        for (int q = 1; q < 4; ++q) tot += p[16 * q];
        emit(h, tot);
    }
}

// The same unit for a PART of the contraction rows: row steps S0 .. S0 + S - 1 of the enumeration above (a wave of the
// four-wave kernel of agg.hip owns half -- or, with one column block, a quarter -- of the rows of its columns: a quarter of
// the registers, so that several workgroups share a CU and one workgroup's row sums and stores overlap the next one's
// stream).  `part(h, tot)` receives the wave's partial sums in the layout of emit(); the caller adds the parts in fixed order.
#ifndef AGG_NT
#define AGG_NT 1                          // operator loads with the non-temporal hint (read once, never again)
#endif
// D: requests in flight per lane.  D = S: the whole row part is requested up front.  D < S ([r6], A/B form 44): D requests up front, the
// request of step s + D behind the products of step s -- a wave's requests are then spread over the launch instead of landing
// together (the memory system serves them roughly in issue order: the waves that issued last otherwise hold ALL their products at
// the end of the stream; tools/harness/stream_floor.hip prices that tail at ~1.3 us of 8.8).
template <int S, int S0, int FH, bool NT = (AGG_NT != 0), int D = S, class Part>
__device__ __forceinline__ void agg_mfma_rows(const float* __restrict__ Gk, const float* __restrict__ Xk, long sxc, int F,
                                              int N, int lane, f32x4* red, Part part)
{
    const int li = lane & 15, lq = lane >> 4;
    constexpr int T0 = S0 >> 2, TQ = ((S0 + S + 3) >> 2) - T0;       // X quads (16 rows each) the steps touch
    f32x4 xa[FH][TQ];
    f32x4 gv[S];
#pragma unroll
    for (int h = 0; h < FH; ++h) {
        const float* xr = Xk + (size_t)min(4 * h + (li & 3), F - 1) * sxc;
#pragma unroll
        for (int t4 = 0; t4 < TQ; ++t4)
            xa[h][t4] = *reinterpret_cast<const f32x4*>(xr + min(16 * (T0 + t4) + 4 * lq, N - 4));
    }
    __builtin_amdgcn_sched_barrier(0);
    auto request = [&](int s) {
        const f32x4* gp = reinterpret_cast<const f32x4*>(Gk + (size_t)min(16 * ((S0 + s) >> 2) + 4 * lq + ((S0 + s) & 3), N - 1) * N);
        gv[s] = NT ? __builtin_nontemporal_load(gp) : *gp;
        __builtin_amdgcn_sched_barrier(0);
    };
#pragma unroll
    for (int s = 0; s < (D < S ? D : S); ++s) request(s);
    f32x4 acc[4][FH];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int h = 0; h < FH; ++h) acc[t][h] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < S; ++s) {                             // rows past N: clamped addresses times a = 0
        const bool rok = 16 * ((S0 + s) >> 2) + 4 * lq + ((S0 + s) & 3) < N;
#pragma unroll
        for (int h = 0; h < FH; ++h) {
            const float a = (rok && 4 * h + (li & 3) < F) ? xa[h][((S0 + s) >> 2) - T0][(S0 + s) & 3] : 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t)
                acc[t][h] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, gv[s][t], acc[t][h], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (D < S && s + D < S) request(s + D);
    }
#pragma unroll
    for (int h = 0; h < FH; ++h) {
#pragma unroll
        for (int t = 0; t < 4; ++t) red[t * 64 + lane] = acc[t][h];
        const f32x4* p = red + lq * 64 + li;
        f32x4 tot = p[0];
#pragma unroll
        for (int q = 1; q < 4; ++q) tot += p[16 * q];
        part(h, tot);
    }
}

// agg_mfma_rows in two halves -- the requests of a unit's row part, and its products / row-class sums -- for the kernel that
// puts SEVERAL units' requests in flight before the first product (agg.hip: agg_fwd_mfma3_kernel: a wave streams its row part of
// tap 0, 1, 2 of one episode; a tap's sums, part combine and stores run under the later taps' flight).
template <int S, int S0, int FH>
struct AggRowsRegs {
    static constexpr int T0 = S0 >> 2, TQ = ((S0 + S + 3) >> 2) - T0;
    f32x4 xa[FH][TQ];
    f32x4 gv[S];
};

template <int S, int S0, int FH, bool NT>
__device__ __forceinline__ void agg_mfma_rows_request(AggRowsRegs<S, S0, FH>& r, const float* __restrict__ Gk,
                                                      const float* __restrict__ Xk, long sxc, int F, int N, int lane)
{
    const int li = lane & 15, lq = lane >> 4;
    constexpr int T0 = AggRowsRegs<S, S0, FH>::T0, TQ = AggRowsRegs<S, S0, FH>::TQ;
#pragma unroll
    for (int h = 0; h < FH; ++h) {
        const float* xr = Xk + (size_t)min(4 * h + (li & 3), F - 1) * sxc;
#pragma unroll
        for (int t4 = 0; t4 < TQ; ++t4)
            r.xa[h][t4] = *reinterpret_cast<const f32x4*>(xr + min(16 * (T0 + t4) + 4 * lq, N - 4));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const f32x4* gp = reinterpret_cast<const f32x4*>(Gk + (size_t)min(16 * ((S0 + s) >> 2) + 4 * lq + ((S0 + s) & 3), N - 1) * N);
        r.gv[s] = NT ? __builtin_nontemporal_load(gp) : *gp;
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int S, int S0, int FH, class Part>
__device__ __forceinline__ void agg_mfma_rows_products(const AggRowsRegs<S, S0, FH>& r, int F, int N, int lane, f32x4* red, Part part)
{
    const int li = lane & 15, lq = lane >> 4;
    constexpr int T0 = AggRowsRegs<S, S0, FH>::T0;
    f32x4 acc[4][FH];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int h = 0; h < FH; ++h) acc[t][h] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const bool rok = 16 * ((S0 + s) >> 2) + 4 * lq + ((S0 + s) & 3) < N;
#pragma unroll
        for (int h = 0; h < FH; ++h) {
            const float a = (rok && 4 * h + (li & 3) < F) ? r.xa[h][((S0 + s) >> 2) - T0][(S0 + s) & 3] : 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t)
                acc[t][h] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, r.gv[s][t], acc[t][h], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int h = 0; h < FH; ++h) {
#pragma unroll
        for (int t = 0; t < 4; ++t) red[t * 64 + lane] = acc[t][h];
        const f32x4* p = red + lq * 64 + li;
        f32x4 tot = p[0];
#pragma unroll
        for (int q = 1; q < 4; ++q) tot += p[16 * q];
        part(h, tot);
    }
}

}  // namespace
