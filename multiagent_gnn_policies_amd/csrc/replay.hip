// Compact DAGGER replay: minibatch states rebuilt from the frame ring the collecting rollout kernel fills
// (mgp_rollout_collect, rollout.hip).  Replaces the reference's torch.cat over 20 Python state objects of 290 KB each
// (gnn_dagger.py:83-86, replay_buffer.py:6-49) and the dense device replay of round 1 (128 KB per transition at N = 100):
// a frame is the features x_t (6,N), the membership bits of the network A_t (N x 2 u64), the expert label (2,N) and the
// age of the state -- 4.8 KB -- and the K-tap state of a transition is its frame plus its K-1 predecessors in the ring:
//     delay_state[k] = x_{t-k}                       (state_with_delay.py:50-53; zero before the episode's reset)
//     delay_gso[0] = I, delay_gso[j] = A_t A_{t-1} .. A_{t-j+1}      (:44-47; zero for j > age: the reference starts every
//                                                                      episode from zero-filled slices)
// The products are evaluated row by row along the bit rows, e_i A_t A_{t-1} .., with the summation order of the rollout
// kernels' own dense rebuild (ascending neighbour index); row weights are (float)(1 / max(deg, 1)) (mean pooling) or 1.
#include "mgp_common.h"

namespace {

constexpr int RG_THREADS = 256;
constexpr int RG_WAVES = RG_THREADS / 64;
constexpr int RG_SPLIT = 16;              // workgroups per operator slice (row sixteenths: one or two rows per wave)

// grid (Bt, 1 + (K - 1) * RG_SPLIT): y = 0 copies the delay line, the label and the identity slice of sample x; y >= 1
// builds rows [part * ceil(N / RG_SPLIT), ...) of slice j = 1 + (y - 1) / RG_SPLIT.  Many small workgroups instead of one per sample: the
// row products are dependent LDS chains (a 20-sample minibatch took 24 us on 20 workgroups of 1024 threads, 16 us on 180 of 256, 9 us on 660).
template <int NW>
__global__ __launch_bounds__(RG_THREADS)
void replay_gather_kernel(const float* __restrict__ feat, const unsigned long long* __restrict__ bits,
                          const float* __restrict__ label, const int* __restrict__ age, const long* __restrict__ idx,
                          const int* __restrict__ cursor, int Bt, int lanes, int ring_steps, int K, int N, int mean_pooling,
                          float* __restrict__ X, float* __restrict__ G, float* __restrict__ Y)
{   // Bt = minibatch size = stride of `idx` per cursor step; gridDim.x = Bt * (minibatches gathered by this launch)
    extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
    const int H = K > 1 ? K - 1 : 1, Np = (N + 3) & ~3;
    unsigned long long* sb = reinterpret_cast<unsigned long long*>(smraw);                  // [H][N][NW]
    float* sw = reinterpret_cast<float*>(sb + (size_t)H * N * NW);                           // [H][N]
    float* rball = sw + ((H * N + 3) & ~3);                                                 // [waves][2][Np]
    const int b = blockIdx.x, role = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long cur = cursor != nullptr ? (long)cursor[0] : 0L;
    const long r = idx[cur * Bt + b];
    const long ring = (long)ring_steps * lanes;
    const int a = age[r];
    float* Gb = G + (size_t)b * K * N * N;
    if (role == 0) {
        // delay line, label, identity slice
        for (int e = tid; e < K * 6 * N; e += RG_THREADS) {
            const int k = e / (6 * N), rem = e - k * 6 * N;
            long rk = r - (long)k * lanes; rk = rk < 0 ? rk + ring : rk;
            X[(size_t)b * K * 6 * N + e] = (a >= k) ? feat[(size_t)rk * 6 * N + rem] : 0.f;
        }
        for (int e = tid; e < 2 * N; e += RG_THREADS) Y[(size_t)b * 2 * N + e] = label[(size_t)r * 2 * N + e];
        for (int e = tid; e < N * N; e += RG_THREADS) { const int i = e / N, n = e - i * N; Gb[e] = (i == n) ? 1.f : 0.f; }
        return;
    }
    const int j = 1 + (role - 1) / RG_SPLIT, part = (role - 1) % RG_SPLIT;
    const int rows_per = (N + RG_SPLIT - 1) / RG_SPLIT, i_lo = part * rows_per, i_hi = min(N, i_lo + rows_per);
    float* Gj = Gb + (size_t)j * N * N;
    if (a < j) {                                              // no j-step history yet: zero slice (reference: zero-filled)
        for (int e = i_lo * N + tid; e < i_hi * N; e += RG_THREADS) Gj[e] = 0.f;
        return;
    }
    // history networks of this slice: slot q = A_{t-q}, q < j
    for (int e = tid; e < j * N; e += RG_THREADS) {
        const int q = e / N, row = e - q * N;
        long rq = r - (long)q * lanes; rq = rq < 0 ? rq + ring : rq;
        int cnt = 0;
#pragma unroll
        for (int wd = 0; wd < NW; ++wd) {
            const unsigned long long w = bits[((size_t)rq * N + row) * NW + wd];
            sb[(size_t)e * NW + wd] = w;
            cnt += __popcll(w);
        }
        const double deg = (double)cnt;
        sw[e] = (float)(mean_pooling ? 1.0 / (deg == 0.0 ? 1.0 : deg) : 1.0);
    }
    __syncthreads();
    float* rbuf = rball + wave * 2 * Np;
    for (int i = i_lo + wave; i < i_hi; i += RG_WAVES) {
        float* r0 = rbuf;
        float* r1 = rbuf + Np;
        const float wi = sw[i];
        const unsigned long long* rowT = sb + (size_t)i * NW;
        for (int n = lane; n < N; n += 64) r0[n] = ((rowT[n >> 6] >> (n & 63)) & 1ull) ? wi : 0.f;
        for (int q = 1; q < j; ++q) {
            const float* wq = sw + q * N;
            for (int n = lane; n < N; n += 64) {
                const unsigned long long* rw = sb + ((size_t)q * N + n) * NW;
                float sacc = 0.f;
#pragma unroll
                for (int wd = 0; wd < NW; ++wd) {
                    unsigned long long w = rw[wd];
                    while (w) { const int m = 64 * wd + __builtin_ctzll(w); w &= w - 1ull; sacc = fmaf(r0[m], wq[m], sacc); }
                }
                r1[n] = sacc;
            }
            float* tsw = r0; r0 = r1; r1 = tsw;
        }
        for (int n = lane; n < N; n += 64) Gj[(size_t)i * N + n] = r0[n];
    }
}

// The same for frames of large flocks (mgp_sparse_policy_collect: N > 256, NW words per bit row, the row weights stored with
// the frame): the history does not fit the LDS, so a wave builds ONE row i of slice j from HBM/L2 -- e_i A_t, then times
// A_{t-1} .. A_{t-j+1} by scattering along the bit rows of the non-zero entries (symmetric membership), row vectors
// ping-pong in LDS -- the loop of sp_to_dense_kernel (sparse_policy.hip) on ring frames.
// grid (Bt * nb, 1 + (K - 1) * ceil(N / 4)): y = 0 copies the delay line, the label and the identity slice; else 4 rows of a slice.
__global__ __launch_bounds__(RG_THREADS)
void replay_gather_rows_kernel(const float* __restrict__ feat, const unsigned long long* __restrict__ bits,
                               const float* __restrict__ wrow, const float* __restrict__ label, const int* __restrict__ age,
                               const long* __restrict__ idx, const int* __restrict__ cursor, int Bt, int lanes,
                               int ring_steps, int K, int N, int NW, float* __restrict__ X, float* __restrict__ G,
                               float* __restrict__ Y)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
    const int Np = (N + 3) & ~3;
    const int b = blockIdx.x, role = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long cur = cursor != nullptr ? (long)cursor[0] : 0L;
    const long r = idx[cur * Bt + b];
    const long ring = (long)ring_steps * lanes;
    const int a = age[r];
    float* Gb = G + (size_t)b * K * N * N;
    if (role == 0) {
        for (int e = tid; e < K * 6 * N; e += RG_THREADS) {
            const int k = e / (6 * N), rem = e - k * 6 * N;
            long rk = r - (long)k * lanes; rk = rk < 0 ? rk + ring : rk;
            X[(size_t)b * K * 6 * N + e] = (a >= k) ? feat[(size_t)rk * 6 * N + rem] : 0.f;
        }
        for (int e = tid; e < 2 * N; e += RG_THREADS) Y[(size_t)b * 2 * N + e] = label[(size_t)r * 2 * N + e];
        for (int e = tid; e < N * N; e += RG_THREADS) { const int i = e / N, n = e - i * N; Gb[e] = (i == n) ? 1.f : 0.f; }
        return;
    }
    const int groups = (N + 3) / 4;
    const int j = 1 + (role - 1) / groups, i = ((role - 1) % groups) * 4 + wave;
    if (i >= N) return;                                       // whole wave; no workgroup barrier below
    float* Gr = Gb + ((size_t)j * N + i) * N;
    if (a < j) {                                              // no j-step history yet: zero row (reference: zero-filled slices)
        for (int n = lane; n < N; n += 64) Gr[n] = 0.f;
        return;
    }
    float* r0 = reinterpret_cast<float*>(smraw) + (size_t)wave * 2 * Np;
    float* r1 = r0 + Np;
    {   // e_i . A_t
        const unsigned long long* row = bits + ((size_t)r * N + i) * NW;
        const float wi = wrow[(size_t)r * N + i];
        for (int n = lane; n < N; n += 64) r0[n] = ((row[n >> 6] >> (n & 63)) & 1ull) ? wi : 0.f;
    }
    for (int q = 1; q < j; ++q) {                             // r1 = r0 . A_{t-q}
        long rq = r - (long)q * lanes; rq = rq < 0 ? rq + ring : rq;
        const unsigned long long* net = bits + (size_t)rq * N * NW;
        const float* wn = wrow + (size_t)rq * N;
        for (int n = lane; n < N; n += 64) r1[n] = 0.f;
        for (int m0 = 0; m0 < N; m0 += 64) {
            const float rv = (m0 + lane < N) ? r0[m0 + lane] : 0.f;
            unsigned long long nz = __ballot(rv != 0.f);
            while (nz) {                                      // wave-uniform loop over the non-zero entries, ascending m
                const int m = m0 + __builtin_ctzll(nz);
                nz &= nz - 1ull;
                const float val = r0[m] * wn[m];
                for (int wd = lane; wd < NW; wd += 64) {      // lane l walks words l, l + 64, ..: distinct columns
                    unsigned long long w = net[(size_t)m * NW + wd];
                    while (w) { const int n = 64 * wd + __builtin_ctzll(w); w &= w - 1ull; r1[n] += val; }
                }
            }
        }
        float* t = r0; r0 = r1; r1 = t;
    }
    for (int n = lane; n < N; n += 64) Gr[n] = r0[n];
}

// The aggregated first-layer input of a minibatch straight from the frame ring -- the operator slices are never formed:
//     Z[s, f K + k, n] = (x_{t-k} . A_t A_{t-1} .. A_{t-k+1})[f, n]      (reference actor.py:64-75 on the state of
//                                                                         state_with_delay.py:44-53; zero for k > age)
// evaluated left to right as k sparse products along the bit rows (symmetric membership: bit row n is column n),
// (v . A)[f, n] = sum over the set bits m of row n, ascending, of v[f, m] w[m] -- the power iteration of the rollout kernels
// (rollout.hip phase A, sparse_policy.hip spl_gather_kernel) on ring frames.  The dense gather above writes K N^2 floats per
// sample for the update kernel to read back (12 MB at N = 1000, 240 MB per minibatch of 20); this one reads K - 1 bit matrices
// (128 KB each at N = 1000) and writes 6 K N floats (72 KB).  Row order f K + k = the column order of the first layer's weight.
// grid (samples, K): workgroup (s, k) owns tap k -- tap 0 copies x_t and the label.
// LDS: v0, v1 [Np][8] floats (feature vectors per agent, ping-pong) | sw [Np] row weights of the product's network
constexpr int RA_THREADS = 512;

template <int NWC>                                            // words per bit row: 2 / 4 (N <= 256, weights from the row populations) or 0: NW words, weights stored
__global__ __launch_bounds__(RA_THREADS)
void replay_aggregate_kernel(const float* __restrict__ feat, const unsigned long long* __restrict__ bits,
                             const float* __restrict__ wrow, const float* __restrict__ label, const int* __restrict__ age,
                             const long* __restrict__ idx, const int* __restrict__ cursor, int Bt, int lanes, int ring_steps,
                             int K, int N, int NWr, int mean_pooling, float* __restrict__ Z, float* __restrict__ Y)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
    const int Np = (N + 3) & ~3;
    const int NW = NWC ? NWC : NWr;
    float* v0 = reinterpret_cast<float*>(smraw);
    float* v1 = v0 + (size_t)Np * 8;
    float* sw = v1 + (size_t)Np * 8;
    const int s = blockIdx.x, k = blockIdx.y, tid = threadIdx.x;
    const long cur = cursor != nullptr ? (long)cursor[0] : 0L;
    const long r = idx[cur * Bt + s];
    const long ring = (long)ring_steps * lanes;
    const int a = age[r];
    float* Zs = Z + (size_t)s * 6 * K * N;
    if (k == 0) {                                             // tap 0: G_0 = I; and the label
        for (int e = tid; e < 6 * N; e += RA_THREADS) { const int f = e / N, n = e - f * N; Zs[(size_t)f * K * N + n] = feat[(size_t)r * 6 * N + e]; }
        for (int e = tid; e < 2 * N; e += RA_THREADS) Y[(size_t)s * 2 * N + e] = label[(size_t)r * 2 * N + e];
        return;
    }
    if (a < k) {                                              // no k-step history yet (reference: zero-filled slices)
        for (int e = tid; e < 6 * N; e += RA_THREADS) { const int f = e / N, n = e - f * N; Zs[((size_t)f * K + k) * N + n] = 0.f; }
        return;
    }
    {
        long rk = r - (long)k * lanes; rk = rk < 0 ? rk + ring : rk;
        for (int e = tid; e < 6 * N; e += RA_THREADS) { const int f = e / N, n = e - f * N; v0[n * 8 + f] = feat[(size_t)rk * 6 * N + e]; }
    }
    for (int q = 0; q < k; ++q) {                             // v1 = v0 . A_{t-q}
        long rq = r - (long)q * lanes; rq = rq < 0 ? rq + ring : rq;
        const unsigned long long* net = bits + (size_t)rq * N * NW;
        unsigned long long mine[NWC ? NWC : 1];               // N <= 256 <= threads: the row this thread owns stays in registers
        if (NWC) {
            int cnt = 0;
#pragma unroll
            for (int wd = 0; wd < NWC; ++wd) { mine[wd] = net[(size_t)min(tid, N - 1) * NWC + wd]; cnt += __popcll(mine[wd]); }
            const double deg = (double)cnt;
            if (tid < N) sw[tid] = (float)(mean_pooling ? 1.0 / (deg == 0.0 ? 1.0 : deg) : 1.0);
        } else {
            for (int n = tid; n < N; n += RA_THREADS) sw[n] = wrow[(size_t)rq * N + n];
        }
        __syncthreads();                                      // v0 and sw complete
        const bool last = q == k - 1;
        for (int n = tid; n < N; n += RA_THREADS) {
            float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            auto walk = [&](unsigned long long w, int wd) {
                while (w) {
                    const int m = 64 * wd + __builtin_ctzll(w);
                    w &= w - 1ull;
                    const float g = sw[m];
                    const float4 x0 = *reinterpret_cast<const float4*>(v0 + m * 8);
                    const float2 x1 = *reinterpret_cast<const float2*>(v0 + m * 8 + 4);
                    acc[0] = fmaf(x0.x, g, acc[0]); acc[1] = fmaf(x0.y, g, acc[1]); acc[2] = fmaf(x0.z, g, acc[2]);
                    acc[3] = fmaf(x0.w, g, acc[3]); acc[4] = fmaf(x1.x, g, acc[4]); acc[5] = fmaf(x1.y, g, acc[5]);
                }
            };
            if (NWC) {
#pragma unroll
                for (int wd = 0; wd < NWC; ++wd) walk(mine[wd], wd);
            } else {
                const unsigned long long* row = net + (size_t)n * NW;
                for (int wd0 = 0; wd0 < NW; wd0 += 8) {       // eight words requested together
                    unsigned long long w[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) w[u] = row[min(wd0 + u, NW - 1)];
#pragma unroll
                    for (int u = 0; u < 8; ++u) if (wd0 + u < NW) walk(w[u], wd0 + u);
                }
            }
            if (last) {
#pragma unroll
                for (int f = 0; f < 6; ++f) Zs[((size_t)f * K + k) * N + n] = acc[f];
            } else {
                *reinterpret_cast<float4*>(v1 + n * 8) = make_float4(acc[0], acc[1], acc[2], acc[3]);
                *reinterpret_cast<float2*>(v1 + n * 8 + 4) = make_float2(acc[4], acc[5]);
            }
        }
        if (!last) __syncthreads();                           // v1 complete, sw and v0 free
        float* t = v0; v0 = v1; v1 = t;
    }
}

}  // namespace

extern "C" int mgp_replay_aggregate(const float* feat, const unsigned long long* bits, const float* wrow, const float* label,
                                    const int* age, const long* idx, const int* cursor, int Bt, int nb, int lanes,
                                    int ring_steps, int K, int N, int mean_pooling, float* Z, float* Y, void* stream)
{
    if (Bt < 0 || nb < 1 || lanes < 1 || ring_steps < 1 || K < 1 || K > 5 || N < 4) return MGP_EINVAL;
    if (N > 2048) return MGP_EUNSUPPORTED;
    if (Bt == 0) return MGP_OK;
    if ((long)Bt * nb > 2147483647L / 64) return MGP_EINVAL;
    MGP_CHECK_PTR(feat); MGP_CHECK_PTR8(bits); MGP_CHECK_PTR(label); MGP_CHECK_PTR(age); MGP_CHECK_PTR8(idx);
    MGP_CHECK_PTR(Z); MGP_CHECK_PTR(Y);
    if (N > 256) MGP_CHECK_PTR(wrow);                         // frames of the factored path carry their row weights
    if (cursor != nullptr && (reinterpret_cast<uintptr_t>(cursor) & 3u)) return MGP_EALIGN;
    const int Np = (N + 3) & ~3;
    const size_t lds = (size_t)Np * (8 + 8 + 1) * sizeof(float);
    const dim3 grid(Bt * nb, K);
    hipStream_t st = static_cast<hipStream_t>(stream);
    mgp_clear_error();
#define MGP_RA_LAUNCH(NWC_, NW_)                                                                                          \
    do {                                                                                                                  \
        if (mgp_allow_dyn_lds(reinterpret_cast<const void*>(replay_aggregate_kernel<NWC_>), lds) != hipSuccess) return MGP_ELAUNCH; \
        hipLaunchKernelGGL(replay_aggregate_kernel<NWC_>, grid, dim3(RA_THREADS), lds, st, feat, bits, wrow, label, age, idx, \
                           cursor, Bt, lanes, ring_steps, K, N, NW_, mean_pooling, Z, Y);                                 \
    } while (0)
    if (N <= 128) MGP_RA_LAUNCH(2, 2);
    else if (N <= 256) MGP_RA_LAUNCH(4, 4);
    else MGP_RA_LAUNCH(0, mgp_sparse_words(N));
#undef MGP_RA_LAUNCH
    return mgp_launch_status();
}

extern "C" int mgp_replay_gather_rows(const float* feat, const unsigned long long* bits, const float* wrow,
                                      const float* label, const int* age, const long* idx, const int* cursor, int Bt, int nb,
                                      int lanes, int ring_steps, int K, int N, float* X, float* G, float* Y, void* stream)
{
    if (Bt < 0 || nb < 1 || lanes < 1 || ring_steps < 1 || K < 1 || K > 5 || N < 4) return MGP_EINVAL;
    if (N > 4096) return MGP_EUNSUPPORTED;
    if (Bt == 0) return MGP_OK;
    if ((long)Bt * nb > 65535L * 32) return MGP_EINVAL;
    MGP_CHECK_PTR(feat); MGP_CHECK_PTR8(bits); MGP_CHECK_PTR(wrow); MGP_CHECK_PTR(label); MGP_CHECK_PTR(age);
    MGP_CHECK_PTR8(idx); MGP_CHECK_PTR(X); MGP_CHECK_PTR(G); MGP_CHECK_PTR(Y);
    if (cursor != nullptr && (reinterpret_cast<uintptr_t>(cursor) & 3u)) return MGP_EALIGN;
    const int NW = mgp_sparse_words(N), Np = (N + 3) & ~3;
    const int lds = RG_WAVES * 2 * Np * 4;
    const long gy = 1 + (long)(K - 1) * ((N + 3) / 4);
    if (gy > 65535) return MGP_EUNSUPPORTED;
    mgp_clear_error();
    if (mgp_allow_dyn_lds(reinterpret_cast<const void*>(replay_gather_rows_kernel), (size_t)lds) != hipSuccess) return MGP_ELAUNCH;
    hipLaunchKernelGGL(replay_gather_rows_kernel, dim3(Bt * nb, (unsigned)gy), dim3(RG_THREADS), lds,
                       static_cast<hipStream_t>(stream), feat, bits, wrow, label, age, idx, cursor, Bt, lanes, ring_steps, K, N,
                       NW, X, G, Y);
    return mgp_launch_status();
}

extern "C" int mgp_replay_gather_many(const float* feat, const unsigned long long* bits, const float* label, const int* age,
                                      const long* idx, const int* cursor, int Bt, int nb, int lanes, int ring_steps, int K,
                                      int N, int mean_pooling, float* X, float* G, float* Y, void* stream)
{
    if (Bt < 0 || nb < 1 || lanes < 1 || ring_steps < 1 || K < 1 || K > 5 || N < 4) return MGP_EINVAL;
    if (N > 256) return MGP_EUNSUPPORTED;
    if (Bt == 0) return MGP_OK;
    if ((long)Bt * nb > 2147483647L / 64) return MGP_EINVAL;
    MGP_CHECK_PTR(feat); MGP_CHECK_PTR8(bits); MGP_CHECK_PTR(label); MGP_CHECK_PTR(age); MGP_CHECK_PTR8(idx);
    MGP_CHECK_PTR(X); MGP_CHECK_PTR(G); MGP_CHECK_PTR(Y);
    if (cursor != nullptr && (reinterpret_cast<uintptr_t>(cursor) & 3u)) return MGP_EALIGN;
    const int H = K > 1 ? K - 1 : 1, Np = (N + 3) & ~3, NW = N > 128 ? 4 : 2;      // words per bit row: the collecting kernels' layout
    const int lds = H * N * NW * 8 + ((H * N + 3) & ~3) * 4 + RG_WAVES * 2 * Np * 4;
    mgp_clear_error();
    const dim3 grid(Bt * nb, 1 + (K - 1) * RG_SPLIT);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (NW == 2)
        hipLaunchKernelGGL(replay_gather_kernel<2>, grid, dim3(RG_THREADS), lds, st, feat, bits, label,
                           age, idx, cursor, Bt, lanes, ring_steps, K, N, mean_pooling, X, G, Y);
    else
        hipLaunchKernelGGL(replay_gather_kernel<4>, grid, dim3(RG_THREADS), lds, st, feat, bits, label,
                           age, idx, cursor, Bt, lanes, ring_steps, K, N, mean_pooling, X, G, Y);
    return mgp_launch_status();
}

extern "C" int mgp_replay_gather(const float* feat, const unsigned long long* bits, const float* label, const int* age,
                                 const long* idx, const int* cursor, int Bt, int lanes, int ring_steps, int K, int N,
                                 int mean_pooling, float* X, float* G, float* Y, void* stream)
{
    return mgp_replay_gather_many(feat, bits, label, age, idx, cursor, Bt, 1, lanes, ring_steps, K, N, mean_pooling, X, G, Y,
                                  stream);
}
