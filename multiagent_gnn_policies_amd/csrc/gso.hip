// Delayed-GSO recursion and delay line (reference state_with_delay.py:38-53):
//   G_next[b,0] = I ; G_next[b,1] = A[b] ; G_next[b,j] = A[b] @ G_prev[b,j-1]  (j >= 2)
//   Xd_next[b,0] = X_t[b] ; Xd_next[b,j] = Xd_prev[b,j-1]
//
// A[b] is a radius-graph operator: row i holds deg(i) non-zeros (typically ~5-20 of N).  The product is
// evaluated row-wise as a weighted sum of the deg(i) source rows -- exact for ANY dense A (zeros are
// skipped, which cannot change an fp32 sum of finite values) and O(N^2 deg) instead of O(N^3).  fp32
// MFMA would run at the fp32 VALU rate anyway (MI355X: both 157 TF), so nothing is lost when A is dense.
// One wave per output row: the row of A is compacted to an (index, weight) list in LDS with ballots,
// then lanes own float4 column groups and walk the list; source rows come from L2 (G_prev[b] is 40 KB
// at N = 100 and is re-read by every row of the episode).  Summation order = ascending source row
// index: deterministic.
#include "mgp_common.h"

namespace {

constexpr int GSO_THREADS = 256;
constexpr int GSO_WAVES = GSO_THREADS / 64;
constexpr int GSO_ROWS = 4;                 // rows of A per workgroup (1 per wave: shortest dependent chain)

template <int V> struct F4 { };

template <int V>
__device__ __forceinline__ void row_axpy(float (&acc)[V], float w, const float* __restrict__ p)
{
    if constexpr (V == 4) {
        const float4 g = *reinterpret_cast<const float4*>(p);
        acc[0] = fmaf(w, g.x, acc[0]); acc[1] = fmaf(w, g.y, acc[1]);
        acc[2] = fmaf(w, g.z, acc[2]); acc[3] = fmaf(w, g.w, acc[3]);
    } else {
        acc[0] = fmaf(w, p[0], acc[0]);
    }
}

// grid: 1-D.  Workgroup id -> (episode b, slot) with b % 8 == id % 8: the dispatcher places workgroup id on XCD
// id % 8 (observed, used for speed only), so every row tile of an episode shares ONE XCD's L2 and the source
// rows of G_prev[b] (re-read deg times) are fetched from HBM once instead of once per XCD.
// slot < nrt: row tile; slot >= nrt: delay-line copy duty.
// Products are written for dst slices j in [j_lo, j_hi): dst[b,j] = A[b] @ src[b,j-1].
// If write_base: dst[b,0] = I, dst[b,1] = (has_prev ? A[b] : 0), and slices >= 2 are zeroed when !has_prev.
template <int V>
__global__ __launch_bounds__(GSO_THREADS)
void gso_rows_kernel(const float* __restrict__ A, const float* __restrict__ src, float* __restrict__ dst,
                     long sAb, int mode, int B, int K, int N, int j_lo, int j_hi, int write_base, int has_prev, int nrt,
                     int nslots,
                     const float* __restrict__ X_t, const float* __restrict__ Xd_prev, float* __restrict__ Xd_next,
                     int F)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // id = (b / 8) * (8 * slots) + slot * 8 + (b % 8)
    const int slots = nslots;
    const int grp = blockIdx.x / (8 * slots), rem8 = blockIdx.x - grp * (8 * slots);
    const int slot = rem8 / 8, b = grp * 8 + (rem8 & 7);
    if (b >= B) return;

    if (slot >= nrt) {
        // delay line duty: copy (K,F,N) floats for episode b, spread over the extra blocks
        const int nb = slots - nrt, bi = slot - nrt;
        const long per = (long)K * F * N;
        float* out = Xd_next + (long)b * per;
        for (long i = (long)bi * GSO_THREADS + tid; i < per; i += (long)nb * GSO_THREADS) {
            const long j = i / ((long)F * N);
            float v;
            if (j == 0) { if (mode == 1) continue; v = X_t[(long)b * F * N + i]; }
            else v = has_prev ? Xd_prev[(long)b * per + i - (long)F * N] : 0.f;
            out[i] = v;
        }
        return;
    }

    int* idx = reinterpret_cast<int*>(smem) + (size_t)wave * 2 * N;      // per-wave [N] indices
    float* wgt = reinterpret_cast<float*>(idx + N);                       // per-wave [N] weights
    const size_t NN = (size_t)N * N;
    const float* Ab = A + (size_t)b * sAb;
    const float* srcb = src ? src + (size_t)b * K * NN : nullptr;
    float* dstb = dst + (size_t)b * K * NN;

    for (int rr = wave; rr < GSO_ROWS; rr += GSO_WAVES) {
        const int i = slot * GSO_ROWS + rr;
        if (i >= N) break;                                            // wave-uniform
        const float* arow = Ab + (size_t)i * N;
        // ---- compact the non-zeros of A[b,i,:] (ascending m)
        int cnt = 0;
        for (int m0 = 0; m0 < N; m0 += 64) {
            const int m = m0 + lane;
            const float a = (m < N) ? arow[m] : 0.f;
            const bool nz = (a != 0.f);
            const unsigned long long mask = __ballot(nz);
            if (nz) {
                const int pos = cnt + __popcll(mask & ((1ull << lane) - 1ull));
                idx[pos] = m;
                wgt[pos] = a;
            }
            cnt += __popcll(mask);
            if (write_base) {
                if (m < N) {
                    dstb[(size_t)i * N + m] = (m == i) ? 1.f : 0.f;                 // slice 0 = I
                    if (K > 1) dstb[NN + (size_t)i * N + m] = has_prev ? a : 0.f;   // slice 1 = A (A @ I)
                }
            }
        }
        // make this wave's LDS list visible to all its lanes (single wave: a wave barrier suffices)
        if (mode == 1 && !has_prev && K > 1) {               // episode start: slice 1 (the sim's A_t) is zeroed too
            float* r1 = dstb + NN + (size_t)i * N;
            for (int n = lane; n < N; n += 64) r1[n] = 0.f;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        for (int j = max(j_lo, 2); j < j_hi; ++j) {
            float* orow = dstb + (size_t)j * NN + (size_t)i * N;
            if (!has_prev) {
                if (write_base || mode == 1) for (int n = lane; n < N; n += 64) orow[n] = 0.f;
                continue;
            }
            const float* sj = srcb + (size_t)(j - 1) * NN;
            for (int n0 = lane * V; n0 < N; n0 += 64 * V) {
                float acc[V];
#pragma unroll
                for (int v = 0; v < V; ++v) acc[v] = 0.f;
                int e = 0;
                for (; e + 4 <= cnt; e += 4) {
                    const int m0_ = idx[e], m1_ = idx[e + 1], m2_ = idx[e + 2], m3_ = idx[e + 3];
                    const float w0 = wgt[e], w1 = wgt[e + 1], w2 = wgt[e + 2], w3 = wgt[e + 3];
                    row_axpy<V>(acc, w0, sj + (size_t)m0_ * N + n0);
                    row_axpy<V>(acc, w1, sj + (size_t)m1_ * N + n0);
                    row_axpy<V>(acc, w2, sj + (size_t)m2_ * N + n0);
                    row_axpy<V>(acc, w3, sj + (size_t)m3_ * N + n0);
                }
                for (; e < cnt; ++e) row_axpy<V>(acc, wgt[e], sj + (size_t)idx[e] * N + n0);
                if constexpr (V == 4) {
                    *reinterpret_cast<float4*>(orow + n0) = make_float4(acc[0], acc[1], acc[2], acc[3]);
                } else {
                    orow[n0] = acc[0];
                }
            }
        }
        __builtin_amdgcn_wave_barrier();      // list is rewritten by the next row of this wave
    }
}

// ---- N <= 128, N % 4 == 0: a row needs only N/4 <= 32 float4 lanes, so each HALF-wave owns one output row (two
// rows per wave in flight, half as many waves as gso_rows_kernel for the same work; same arithmetic and order).
constexpr int GH_ROWS = 2 * GSO_WAVES;       // rows per workgroup

__global__ __launch_bounds__(GSO_THREADS)
void gso_rows_half_kernel(const float* __restrict__ A, const float* __restrict__ src, float* __restrict__ dst,
                          long sAb, int mode, int B, int K, int N, int j_lo, int j_hi, int write_base, int has_prev,
                          int nrt, int nslots, const float* __restrict__ X_t, const float* __restrict__ Xd_prev,
                          float* __restrict__ Xd_next, int F)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = blockIdx.x / (8 * nslots), rem8 = blockIdx.x - grp * (8 * nslots);
    const int slot = rem8 / 8, b = grp * 8 + (rem8 & 7);
    if (b >= B) return;
    if (slot >= nrt) {                                       // delay-line duty (same contract as gso_rows_kernel)
        const int nb = nslots - nrt, bi = slot - nrt;
        const long per = (long)K * F * N;
        float* out = Xd_next + (long)b * per;
        for (long i = (long)bi * GSO_THREADS + tid; i < per; i += (long)nb * GSO_THREADS) {
            const long j = i / ((long)F * N);
            float v;
            if (j == 0) { if (mode == 1) continue; v = X_t[(long)b * F * N + i]; }
            else v = has_prev ? Xd_prev[(long)b * per + i - (long)F * N] : 0.f;
            out[i] = v;
        }
        return;
    }
    const int half = lane >> 5, hl = lane & 31;
    int* idx = reinterpret_cast<int*>(smem) + (size_t)(wave * 2 + half) * 2 * N;   // per half-wave [N] indices
    float* wgt = reinterpret_cast<float*>(idx + N);                                 // per half-wave [N] weights
    const size_t NN = (size_t)N * N;
    const float* Ab = A + (size_t)b * sAb;
    const float* srcb = src ? src + (size_t)b * K * NN : nullptr;
    float* dstb = dst + (size_t)b * K * NN;
    const int i = slot * GH_ROWS + wave * 2 + half;
    const bool valid = i < N;
    const float* arow = Ab + (size_t)(valid ? i : 0) * N;
    int cnt = 0;
    for (int m0 = 0; m0 < N; m0 += 32) {
        const int m = m0 + hl;
        const float a = (valid && m < N) ? arow[m] : 0.f;
        const bool nz = (a != 0.f);
        const unsigned long long mask = __ballot(nz);
        const unsigned hm = (unsigned)(mask >> (32 * half));
        if (nz) {
            const int pos = cnt + __popc(hm & ((1u << hl) - 1u));
            idx[pos] = m;
            wgt[pos] = a;
        }
        cnt += __popc(hm);
        if (write_base && valid && m < N) {
            dstb[(size_t)i * N + m] = (m == i) ? 1.f : 0.f;                     // slice 0 = I
            if (K > 1) dstb[NN + (size_t)i * N + m] = has_prev ? a : 0.f;       // slice 1 = A (A @ I)
        }
    }
    if (mode == 1 && !has_prev && K > 1 && valid) {          // episode start: slice 1 (the sim's A_t) is zeroed too
        float* r1 = dstb + NN + (size_t)i * N;
        for (int n = hl; n < N; n += 32) r1[n] = 0.f;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int n0 = hl * 4;
    for (int j = max(j_lo, 2); j < j_hi; ++j) {
        if (!valid) continue;
        float* orow = dstb + (size_t)j * NN + (size_t)i * N;
        if (!has_prev) {
            if (write_base || mode == 1) for (int n = hl; n < N; n += 32) orow[n] = 0.f;
            continue;
        }
        if (n0 >= N) continue;
        const float* sj = srcb + (size_t)(j - 1) * NN + n0;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        int e = 0;
        for (; e + 4 <= cnt; e += 4) {
            const int m0_ = idx[e], m1_ = idx[e + 1], m2_ = idx[e + 2], m3_ = idx[e + 3];
            const float w0 = wgt[e], w1 = wgt[e + 1], w2 = wgt[e + 2], w3 = wgt[e + 3];
            row_axpy<4>(acc, w0, sj + (size_t)m0_ * N);
            row_axpy<4>(acc, w1, sj + (size_t)m1_ * N);
            row_axpy<4>(acc, w2, sj + (size_t)m2_ * N);
            row_axpy<4>(acc, w3, sj + (size_t)m3_ * N);
        }
        for (; e < cnt; ++e) row_axpy<4>(acc, wgt[e], sj + (size_t)idx[e] * N);
        *reinterpret_cast<float4*>(orow + n0) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
}

int launch_gso(const float* A, long sAb, int mode, const float* src, float* dst, int B, int K, int N, int j_lo, int j_hi,
               int write_base, int has_prev, const float* X_t, const float* Xd_prev, float* Xd_next, int F,
               hipStream_t st)
{
    mgp_clear_error();
    const int nrt = mgp_ceil_div(N, GSO_ROWS);
    int extra = 0;
    if (Xd_next != nullptr) {
        const long per = (long)K * F * N;
        extra = (int)((per + GSO_THREADS * 4 - 1) / (GSO_THREADS * 4));
        if (extra < 1) extra = 1;
        if (extra > 64) extra = 64;
    }
    const size_t lds = (size_t)GSO_WAVES * 2 * N * sizeof(float);
    if (lds > 150 * 1024) return MGP_EUNSUPPORTED;
    const bool vec = (N % 4 == 0) && mgp_aligned16(dst) && (src == nullptr || mgp_aligned16(src));
    if (vec && N <= 128) {
        const int nrt2 = mgp_ceil_div(N, GH_ROWS), ns2 = nrt2 + extra;
        const size_t lds2 = (size_t)GH_ROWS * 2 * N * sizeof(float);
        hipLaunchKernelGGL(gso_rows_half_kernel, dim3((unsigned)(((B + 7) / 8) * 8 * ns2)), dim3(GSO_THREADS), lds2, st,
                           A, src, dst, sAb, mode, B, K, N, j_lo, j_hi, write_base, has_prev, nrt2, ns2, X_t, Xd_prev,
                           Xd_next, F);
        return mgp_launch_status();
    }
    const int nslots = nrt + extra;
    dim3 grid((unsigned)(((B + 7) / 8) * 8 * nslots));
    if (vec) {
        if (mgp_allow_dyn_lds(reinterpret_cast<const void*>(gso_rows_kernel<4>), lds) != hipSuccess) return MGP_ELAUNCH;
        hipLaunchKernelGGL((gso_rows_kernel<4>), grid, dim3(GSO_THREADS), lds, st, A, src, dst, sAb, mode, B, K, N,
                           j_lo, j_hi, write_base, has_prev, nrt, nslots, X_t, Xd_prev, Xd_next, F);
    } else {
        if (mgp_allow_dyn_lds(reinterpret_cast<const void*>(gso_rows_kernel<1>), lds) != hipSuccess) return MGP_ELAUNCH;
        hipLaunchKernelGGL((gso_rows_kernel<1>), grid, dim3(GSO_THREADS), lds, st, A, src, dst, sAb, mode, B, K, N,
                           j_lo, j_hi, write_base, has_prev, nrt, nslots, X_t, Xd_prev, Xd_next, F);
    }
    return mgp_launch_status();
}

}  // namespace

extern "C" int mgp_gso_update(const float* A, const float* G_prev, float* G_next,
                              const float* X_t, const float* Xd_prev, float* Xd_next,
                              int B, int K, int F, int N, int has_prev, void* stream)
{
    if (B < 0 || K <= 0 || F <= 0 || N <= 0) return MGP_EINVAL;
    if (B == 0) return MGP_OK;
    if (B > 65535) return MGP_EINVAL;
    MGP_CHECK_PTR(A); MGP_CHECK_PTR(G_next); MGP_CHECK_PTR(X_t); MGP_CHECK_PTR(Xd_next);
    if (has_prev) {
        if (K > 2) MGP_CHECK_PTR(G_prev);
        if (K > 1) MGP_CHECK_PTR(Xd_prev);
        if (G_prev == G_next || Xd_prev == Xd_next) return MGP_EINVAL;
    }
    return launch_gso(A, (long)N * N, 0, has_prev ? G_prev : nullptr, G_next, B, K, N, 2, K, 1, has_prev ? 1 : 0,
                      X_t, has_prev ? Xd_prev : nullptr, Xd_next, F, static_cast<hipStream_t>(stream));
}

extern "C" int mgp_gso_advance(const float* G_prev, float* G_next, const float* Xd_prev, float* Xd_next,
                               int B, int K, int F, int N, int has_prev, void* stream)
{
    if (B < 0 || K <= 0 || F <= 0 || N <= 0) return MGP_EINVAL;
    if (B == 0 || K == 1) return MGP_OK;
    if (B > 65535) return MGP_EINVAL;
    MGP_CHECK_PTR(G_next); MGP_CHECK_PTR(Xd_next);
    if (has_prev) {
        MGP_CHECK_PTR(G_prev); MGP_CHECK_PTR(Xd_prev);
        if (G_prev == G_next || Xd_prev == Xd_next) return MGP_EINVAL;
    }
    // A_t lives in slice 1 of G_next (batch stride K*N*N)
    return launch_gso(G_next + (size_t)N * N, (long)K * N * N, 1, has_prev ? G_prev : nullptr, G_next, B, K, N, 2, K, 0,
                      has_prev ? 1 : 0, Xd_next, has_prev ? Xd_prev : nullptr, Xd_next, F, static_cast<hipStream_t>(stream));
}

extern "C" int mgp_gso_powers(const float* A, float* P, int B, int K, int N, void* stream)
{
    if (B < 0 || K <= 0 || N <= 0) return MGP_EINVAL;
    if (B == 0) return MGP_OK;
    if (B > 65535) return MGP_EINVAL;
    MGP_CHECK_PTR(A); MGP_CHECK_PTR(P);
    hipStream_t st = static_cast<hipStream_t>(stream);
    // slices 0 (I) and 1 (A); then one dependent launch per further power
    int rc = launch_gso(A, (long)N * N, 0, nullptr, P, B, K, N, 2, 2, 1, 1, nullptr, nullptr, nullptr, 0, st);
    for (int j = 2; j < K && rc == MGP_OK; ++j)
        rc = launch_gso(A, (long)N * N, 0, P, P, B, K, N, j, j + 1, 0, 1, nullptr, nullptr, nullptr, 0, st);
    return rc;
}
