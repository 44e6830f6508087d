// Graph-shift aggregation  Y[b,k,c,n] = sum_m X[b,k,c,m] * G[b,k,m,n]   (reference actor.py:69-71)
// and its backward w.r.t. X  dX[b,k,c,m] = sum_n dY[b,k,c,n] * G[b,k,m,n].
//
// HBM-bound: 2*C flops per 4 bytes of G (C = 6 -> 3 flop/B, ridge is ~20 flop/B).  The design goal is
// therefore to read every byte of the dense (B,K,N,N) operator exactly once, fully coalesced:
//   * the contraction index m is the ROW index of G, so a workgroup that owns whole rows streams
//     G[b,k] as one flat array of float4 (N <= 256), or as 1 KiB row segments (N > 256);
//   * a thread owns V=4 adjacent output columns and CT channels (CT*V accumulators) and walks down
//     the rows with stride R = 256/colgroups; the X[b,k] tile is staged TRANSPOSED in LDS ([m][c]) so the
//     CT multipliers of one row come from one or two wide, mostly-broadcast ds_reads;
//   * the R row-phases are combined through LDS in a fixed order (deterministic, no atomics).
#include <cstdlib>
#include "mgp_common.h"
#include "agg_mfma.h"

namespace {

constexpr int AGG_THREADS = 256;
constexpr int AGG_UNROLL = 8;       // rows in flight per thread
constexpr int AGG_RED_CH = 8;       // channels reduced per LDS pass

template <int V> struct VecLoad;
template <> struct VecLoad<4> {
    static __device__ __forceinline__ void load(const float* p, float (&g)[4]) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        g[0] = t.x; g[1] = t.y; g[2] = t.z; g[3] = t.w;
    }
};
template <> struct VecLoad<1> {
    static __device__ __forceinline__ void load(const float* p, float (&g)[1]) { g[0] = *p; }
};

// grid: x = column tile + ntiles * channel chunk, y = k, z = b
template <int CT, int V>
__global__ __launch_bounds__(AGG_THREADS)
void agg_fwd_kernel(const float* __restrict__ X, const float* __restrict__ G, float* __restrict__ Y,
                    int K, int C, int N, int tw, int ntiles, int MC,
                    long sxb, long sxk, long sxc, long syb, long syk, long syc)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tile = blockIdx.x % ntiles;
    const int c0 = (blockIdx.x / ntiles) * CT;
    const int k = blockIdx.y, b = blockIdx.z;
    const int n0 = tile * tw;
    const int cols = min(tw, N - n0);
    const int cgt = (cols + V - 1) / V;          // column groups in this tile (<= 256)
    const int R = AGG_THREADS / cgt;             // row phases
    const int twp = cgt * V;
    const int tid = threadIdx.x;
    const int cg = tid % cgt, r = tid / cgt;
    const bool active = r < R;

    float* xs = smem;                            // [MC][CT]
    float* red = smem + (size_t)MC * CT;         // [R][AGG_RED_CH][twp]

    float acc[CT][V];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int v = 0; v < V; ++v) acc[c][v] = 0.f;

    const float* Gbk = G + ((size_t)b * K + k) * (size_t)N * N + n0 + cg * V;
    const float* Xbk = X + b * sxb + k * sxk;
    // a V=1 tail lane past the tile edge only happens when V==1 (cols % 4 == 0 is required for V=4)
    const bool col_ok = (cg * V) < cols;

    for (int m0 = 0; m0 < N; m0 += MC) {
        const int mc = min(MC, N - m0);
        __syncthreads();
        for (int i = tid; i < mc * CT; i += AGG_THREADS) {
            const int c = i / mc, mm = i - c * mc;
            xs[mm * CT + c] = (c0 + c < C) ? Xbk[(c0 + c) * sxc + m0 + mm] : 0.f;
        }
        __syncthreads();
        if (active && col_ok) {
            for (int mm = r; mm < mc; mm += R * AGG_UNROLL) {
                float g[AGG_UNROLL][V];
#pragma unroll
                for (int u = 0; u < AGG_UNROLL; ++u) {
                    const int row = mm + u * R;
                    if (row < mc) {
                        VecLoad<V>::load(Gbk + (size_t)(m0 + row) * N, g[u]);
                    } else {
#pragma unroll
                        for (int v = 0; v < V; ++v) g[u][v] = 0.f;
                    }
                }
#pragma unroll
                for (int u = 0; u < AGG_UNROLL; ++u) {
                    const int row = min(mm + u * R, mc - 1);     // g is zero past the end
                    const float* xr = xs + row * CT;
#pragma unroll
                    for (int c = 0; c < CT; ++c) {
                        const float x = xr[c];
#pragma unroll
                        for (int v = 0; v < V; ++v) acc[c][v] = fmaf(x, g[u][v], acc[c][v]);
                    }
                }
            }
        }
    }

    float* Ybk = Y + b * syb + k * syk;
    if (R == 1) {
        if (active && col_ok) {          // threads past the last column group hold zero accumulators: must not store
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                if (c0 + c < C) {
#pragma unroll
                    for (int v = 0; v < V; ++v) Ybk[(c0 + c) * syc + n0 + cg * V + v] = acc[c][v];
                }
            }
        }
        return;
    }
    // combine the R row phases, AGG_RED_CH channels per pass, fixed order r = 0..R-1
#pragma unroll
    for (int cb = 0; cb < CT; cb += AGG_RED_CH) {
        constexpr int CH = (CT < AGG_RED_CH) ? CT : AGG_RED_CH;
        __syncthreads();
        if (active && col_ok) {
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                if (cb + c < CT) {
#pragma unroll
                    for (int v = 0; v < V; ++v) red[((size_t)r * CH + c) * twp + cg * V + v] = acc[cb + c < CT ? cb + c : 0][v];
                }
            }
        }
        __syncthreads();
        for (int i = tid; i < CH * cols; i += AGG_THREADS) {
            const int c = i / cols, col = i - c * cols;
            if (cb + c < CT && c0 + cb + c < C) {
                float s = 0.f;
                for (int rr = 0; rr < R; ++rr) s += red[((size_t)rr * CH + c) * twp + col];
                Ybk[(c0 + cb + c) * syc + n0 + col] = s;
            }
        }
    }
}

// dX[b,k,c,m] = sum_n dY[b,k,c,n] * G[b,k,m,n]: one wave per row m of G, lanes stride along n.
// grid: x = row tile + nrt * channel chunk, y = k, z = b
constexpr int BWX_ROWS = 32;         // rows of G per workgroup
template <int CT>
__global__ __launch_bounds__(AGG_THREADS)
void agg_bwd_x_kernel(const float* __restrict__ dY, const float* __restrict__ G, float* __restrict__ dX,
                      int K, int C, int N, int nrt, int NC,
                      long sgb, long sgk, long sgc, long sdb, long sdk, long sdc)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];   // dys[CT][NC]
    const int rt = blockIdx.x % nrt;
    const int c0 = (blockIdx.x / nrt) * CT;
    const int k = blockIdx.y, b = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m_begin = rt * BWX_ROWS;
    const int m_end = min(N, m_begin + BWX_ROWS);
    const float* Gbk = G + ((size_t)b * K + k) * (size_t)N * N;
    const float* dYbk = dY + b * sgb + k * sgk;
    float* dXbk = dX + b * sdb + k * sdk;

    constexpr int RPW = BWX_ROWS / 4;      // rows per wave
    float acc[RPW][CT];
#pragma unroll
    for (int i = 0; i < RPW; ++i)
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[i][c] = 0.f;

    for (int n0 = 0; n0 < N; n0 += NC) {
        const int nc = min(NC, N - n0);
        __syncthreads();
        for (int i = tid; i < CT * nc; i += AGG_THREADS) {
            const int c = i / nc, nn = i - c * nc;
            smem[c * NC + nn] = (c0 + c < C) ? dYbk[(c0 + c) * sgc + n0 + nn] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const int m = m_begin + wave + 4 * i;
            if (m < m_end) {
                const float* grow = Gbk + (size_t)m * N + n0;
                for (int nn = lane; nn < nc; nn += 64) {
                    const float g = grow[nn];
#pragma unroll
                    for (int c = 0; c < CT; ++c) acc[i][c] = fmaf(smem[c * NC + nn], g, acc[i][c]);
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const int m = m_begin + wave + 4 * i;
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const float s = mgp_wave_sum(acc[i][c]);
            if (lane == 0 && m < m_end && c0 + c < C) dXbk[(c0 + c) * sdc + m] = s;
        }
    }
}

template <int CT, int V>
int launch_agg_fwd(const float* X, const float* G, float* Y, int B, int K, int C, int N,
                   long sxb, long sxk, long sxc, long syb, long syk, long syc, hipStream_t st)
{
    const int tw = (N <= 256) ? N : 64 * V;
    const int ntiles = mgp_ceil_div(N, tw);
    const int nchunks = mgp_ceil_div(C, CT);
    int MC = 8192 / CT;
    if (MC > N) MC = N;
    constexpr int CH = (CT < AGG_RED_CH) ? CT : AGG_RED_CH;
    const size_t lds = ((size_t)MC * CT + (size_t)AGG_THREADS * V * CH) * sizeof(float);
    dim3 grid(ntiles * nchunks, K, B);
    hipLaunchKernelGGL((agg_fwd_kernel<CT, V>), grid, dim3(AGG_THREADS), lds, st,
                       X, G, Y, K, C, N, tw, ntiles, MC, sxb, sxk, sxc, syb, syk, syc);
    return mgp_launch_status();
}

// ---- forward on the matrix pipe (agg_mfma.h): one workgroup (8 waves) per episode, wave = (tap, column block of <= 16
// four-column groups) streaming ALL N rows of its columns with every request issued up front; row sums live in the
// 4x4x1 MFMA accumulators, no row-phase combine through LDS, no barrier at all.  16 <= N <= 128 (N % 4 == 0), C <= 8.
// 9.3 us at B = 256, N = 100, K = 3 (3.7 TB/s) against 12.3 us for the VALU kernel above.
constexpr int AGM_THREADS = 512;

template <int S, int FH>
__global__ __launch_bounds__(AGM_THREADS)
void agg_fwd_mfma_kernel(const float* __restrict__ X, const float* __restrict__ G, float* __restrict__ Y,
                         int K, int C, int N, int nblk, long sxb, long sxk, long sxc, long syb, long syk, long syc)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, lq = lane >> 4;
    const int b = blockIdx.x;
    const int gtot = N >> 2, g0 = (nblk == 2) ? ((gtot + 1) >> 1) : gtot;
    f32x4* red = reinterpret_cast<f32x4*>(smem) + wave * (4 * 64);
    if (wave < K * nblk) {                                     // one unit per wave (the dispatcher checks K * nblk <= 8)
        const int k = wave / nblk, blk = wave - k * nblk;
        const int ng = blk ? gtot - g0 : g0;
        const int g = blk * g0 + min(li, ng - 1);
        float* Yk = Y + (size_t)b * syb + (size_t)k * syk + 4 * g + lq;
        agg_mfma_unit<S, FH>(G + ((size_t)b * K + k) * (size_t)N * N + 4 * g, X + (size_t)b * sxb + (size_t)k * sxk, sxc, C, N,
                             lane, red, [&](int h, const f32x4& tot) {
                                 if (li < ng) {
#pragma unroll
                                     for (int i = 0; i < 4; ++i)
                                         if (4 * h + i < C) Yk[(size_t)(4 * h + i) * syc] = tot[i];
                                 }
                             });
    }
}

template <int S, int FH>
int launch_agg_fwd_mfma(const float* X, const float* G, float* Y, int B, int K, int C, int N, int nblk,
                        long sxb, long sxk, long sxc, long syb, long syk, long syc, hipStream_t st)
{
    hipLaunchKernelGGL((agg_fwd_mfma_kernel<S, FH>), dim3((unsigned)B), dim3(AGM_THREADS), (AGM_THREADS / 64) * 4 * 64 * 16, st,
                       X, G, Y, K, C, N, nblk, sxb, sxk, sxc, syb, syk, syc);
    return mgp_launch_status();
}

// ---- the same aggregation with FOUR waves per (episode, tap): wave = (column block, row part); a wave holds a quarter of the
// operator share of the eight-wave kernel above (14 float4 instead of 28 at N = 100: <= 128 VGPRs), so three to four
// workgroups share a CU and, beyond one workgroup per CU (B K > 256), one workgroup's row sums, part combine and stores
// overlap the next one's stream -- the eight-wave kernel holds 240 VGPRs per wave: ONE workgroup per CU, every launch phase
// exposed once per episode (4.15 TB/s at B = 2048 against 3.4 at B = 256).  Parts are added in fixed order through LDS.
template <int S, int FH, int NH, int NBLK, bool NT, bool NTS = false, int D = S>   // S row steps per wave, NH row parts, NBLK column blocks: 64 NH NBLK threads; NTS: non-temporal stores; D: requests in flight per lane (agg_mfma.h)
__global__ __launch_bounds__(64 * NH * NBLK)         // (forcing <= 128 VGPRs for a fourth wave per SIMD spills and measured 7 % slower)
void agg_fwd_mfma4_kernel(const float* __restrict__ X, const float* __restrict__ G, float* __restrict__ Y,
                          int K, int C, int N, long sxb, long sxk, long sxc, long syb, long syk, long syc)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NW = NH * NBLK;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, lq = lane >> 4;
    const int b = blockIdx.x / K, k = blockIdx.x - b * K;
    const int blk = wave / NH, part = wave - blk * NH;
    const int gtot = N >> 2, g0 = (NBLK == 2) ? ((gtot + 1) >> 1) : gtot;
    const int ng = blk ? gtot - g0 : g0;
    const int g = blk * g0 + min(li, ng - 1);
    f32x4* red = reinterpret_cast<f32x4*>(smem) + wave * (4 * 64);
    f32x4* comb = reinterpret_cast<f32x4*>(smem) + NW * (4 * 64);           // [wave][FH][64]
    const float* Gk = G + ((size_t)b * K + k) * (size_t)N * N + 4 * g;
    const float* Xk = X + (size_t)b * sxb + (size_t)k * sxk;
    f32x4 mine[FH];
    auto keep = [&](int h, const f32x4& tot) { mine[h] = tot; };
    if (part == 0) agg_mfma_rows<S, 0, FH, NT, D>(Gk, Xk, sxc, C, N, lane, red, keep);
    else if (part == 1) agg_mfma_rows<S, S, FH, NT, D>(Gk, Xk, sxc, C, N, lane, red, keep);
    else if (NH == 4 && part == 2) agg_mfma_rows<S, (NH == 4 ? 2 * S : 0), FH, NT, D>(Gk, Xk, sxc, C, N, lane, red, keep);
    else agg_mfma_rows<S, (NH == 4 ? 3 * S : 0), FH, NT, D>(Gk, Xk, sxc, C, N, lane, red, keep);
    if (part != 0) {
#pragma unroll
        for (int h = 0; h < FH; ++h) comb[(wave * FH + h) * 64 + lane] = mine[h];
    }
    __syncthreads();
    if (part == 0 && li < ng) {
        float* Yk = Y + (size_t)b * syb + (size_t)k * syk + 4 * g + lq;
#pragma unroll
        for (int h = 0; h < FH; ++h) {
            f32x4 tot = mine[h];
#pragma unroll
            for (int q = 1; q < NH; ++q) tot += comb[((wave + q) * FH + h) * 64 + lane];     // fixed order: part 1, 2, 3
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (4 * h + i < C) {
                    if (NTS) __builtin_nontemporal_store(tot[i], Yk + (size_t)(4 * h + i) * syc);
                    else Yk[(size_t)(4 * h + i) * syc] = tot[i];
                }
        }
    }
}

template <int S, int FH, int NH, int NBLK, bool NT = true, int D = S>
int launch_agg_fwd_mfma4(const float* X, const float* G, float* Y, int B, int K, int C, int N,
                         long sxb, long sxk, long sxc, long syb, long syk, long syc, hipStream_t st)
{
    constexpr int NW = NH * NBLK;
    const size_t lds = (size_t)(NW * 4 * 64 + NW * FH * 64) * 16;
    // results with the non-temporal hint once the launch is several workgroups per CU deep (B K >= 3072: 15 MB of results in a
    // 246 MB read stream at B = 2048 -- 50.8 -> 48.4 us in the harness; at B = 256 the plain stores are the faster ones: 8.5 vs 8.9)
    static const int nts_from = getenv("MGP_AGG_NTS") ? atoi(getenv("MGP_AGG_NTS")) : 3072;
    // [r6] ... and with half of a wave's requests (7 of 14) issued behind its products instead of up front: the memory system serves
    // requests roughly in issue order, so with everything up front the waves that issued last hold ALL their products at the end of
    // the stream; spread over the launch, B = 2048: 53.5 -> 51.2 us (0.643 -> 0.672 of 8 TB/s), 1024: 30.3 -> 29.5, 4096: 108.5 ->
    // 105; no change at B <= 512, where a launch is one wave of workgroups (profiles/r06_agg_forms.txt, forms 44-46).
    constexpr int DD = (D == S && S == 14) ? 7 : D;
    if ((long)B * K >= nts_from)
        hipLaunchKernelGGL((agg_fwd_mfma4_kernel<S, FH, NH, NBLK, NT, true, DD>), dim3((unsigned)(B * K)), dim3(64 * NW), lds, st,
                           X, G, Y, K, C, N, sxb, sxk, sxc, syb, syk, syc);
    else
        hipLaunchKernelGGL((agg_fwd_mfma4_kernel<S, FH, NH, NBLK, NT, false, D>), dim3((unsigned)(B * K)), dim3(64 * NW), lds, st,
                           X, G, Y, K, C, N, sxb, sxk, sxc, syb, syk, syc);
    return mgp_launch_status();
}

// ---- [r6] ONE workgroup per episode, eight waves = (column block, row part of S steps), the K = 3 taps one after the other in
// every wave: the requests of all three taps are issued up front (3 S float4 per lane), so a tap's products, row-class sums,
// part combine (one barrier) and stores run while the later taps' rows are still in flight -- round 5's review: "inside a wave,
// issue the row stream as two half-batches; run the row sums and stores of the first under the second's flight".  One
// workgroup fills a CU (3 S + 18 + 32 float4 registers per lane, 80 KB of LDS): the same bytes in flight per CU as three
// four-wave workgroups.  Selected with MGP_AGG_FORM=43 (A/B; profiles/r06_agg_forms.txt).
template <int S, int FH, int PART, bool NT>
__device__ __forceinline__ void agg3_wave(const float* __restrict__ X, const float* __restrict__ G, float* __restrict__ Y, int C, int N,
                                          long sxk, long sxc, long syk, long syc, int b_unused, int lane, int wave, int blk, int g, int ng,
                                          f32x4* red, f32x4* comb)
{
    constexpr int S0 = PART * S;
    const int li = lane & 15, lq = lane >> 4;
    AggRowsRegs<S, S0, FH> r0, r1, r2;
    const size_t gstride = (size_t)N * N;
    agg_mfma_rows_request<S, S0, FH, NT>(r0, G + 4 * g, X, sxc, C, N, lane);
    agg_mfma_rows_request<S, S0, FH, NT>(r1, G + gstride + 4 * g, X + sxk, sxc, C, N, lane);
    agg_mfma_rows_request<S, S0, FH, NT>(r2, G + 2 * gstride + 4 * g, X + 2 * sxk, sxc, C, N, lane);
    auto tail = [&](int k, const f32x4 (&mine)[FH]) {
        f32x4* cb = comb + (size_t)k * 8 * FH * 64;               // [tap][wave][FH][64]
        if (PART != 0) {
#pragma unroll
            for (int h = 0; h < FH; ++h) cb[(wave * FH + h) * 64 + lane] = mine[h];
        }
        __syncthreads();
        if (PART == 0 && li < ng) {
            float* Yk = Y + (size_t)k * syk + 4 * g + lq;
#pragma unroll
            for (int h = 0; h < FH; ++h) {
                f32x4 tot = mine[h];
#pragma unroll
                for (int q = 1; q < 4; ++q) tot += cb[((wave + q) * FH + h) * 64 + lane];     // fixed order: part 1, 2, 3
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (4 * h + i < C) Yk[(size_t)(4 * h + i) * syc] = tot[i];
            }
        }
    };
    f32x4 mine[FH];
    auto keep = [&](int h, const f32x4& tot) { mine[h] = tot; };
    agg_mfma_rows_products<S, S0, FH>(r0, C, N, lane, red, keep); tail(0, mine);
    agg_mfma_rows_products<S, S0, FH>(r1, C, N, lane, red, keep); tail(1, mine);
    agg_mfma_rows_products<S, S0, FH>(r2, C, N, lane, red, keep); tail(2, mine);
}

template <int S, int FH, bool NT>
__global__ __launch_bounds__(512)
void agg_fwd_mfma3_kernel(const float* __restrict__ X, const float* __restrict__ G, float* __restrict__ Y,
                          int C, int N, long sxb, long sxk, long sxc, long syb, long syk, long syc)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15;
    const int b = blockIdx.x;
    const int blk = wave >> 2, part = wave & 3;
    const int gtot = N >> 2, g0 = (gtot + 1) >> 1;
    const int ng = blk ? gtot - g0 : g0;
    const int g = blk * g0 + min(li, ng - 1);
    f32x4* red = reinterpret_cast<f32x4*>(smem) + wave * (4 * 64);
    f32x4* comb = reinterpret_cast<f32x4*>(smem) + 8 * (4 * 64);
    const float* Gb = G + (size_t)b * 3 * (size_t)N * N;
    const float* Xb = X + (size_t)b * sxb;
    float* Yb = Y + (size_t)b * syb;
    if (part == 0) agg3_wave<S, FH, 0, NT>(Xb, Gb, Yb, C, N, sxk, sxc, syk, syc, b, lane, wave, blk, g, ng, red, comb);
    else if (part == 1) agg3_wave<S, FH, 1, NT>(Xb, Gb, Yb, C, N, sxk, sxc, syk, syc, b, lane, wave, blk, g, ng, red, comb);
    else if (part == 2) agg3_wave<S, FH, 2, NT>(Xb, Gb, Yb, C, N, sxk, sxc, syk, syc, b, lane, wave, blk, g, ng, red, comb);
    else agg3_wave<S, FH, 3, NT>(Xb, Gb, Yb, C, N, sxk, sxc, syk, syc, b, lane, wave, blk, g, ng, red, comb);
}

template <int S, int FH>
int launch_agg_fwd_mfma3(const float* X, const float* G, float* Y, int B, int C, int N,
                         long sxb, long sxk, long sxc, long syb, long syk, long syc, hipStream_t st)
{
    const size_t lds = (size_t)(8 * 4 * 64 + 3 * 8 * FH * 64) * 16;
    if (mgp_allow_dyn_lds(reinterpret_cast<const void*>(agg_fwd_mfma3_kernel<S, FH, true>), lds) != hipSuccess) return MGP_ELAUNCH;
    hipLaunchKernelGGL((agg_fwd_mfma3_kernel<S, FH, true>), dim3((unsigned)B), dim3(512), lds, st, X, G, Y, C, N, sxb, sxk, sxc, syb, syk, syc);
    return mgp_launch_status();
}

template <int V>
int dispatch_agg_fwd(const float* X, const float* G, float* Y, int B, int K, int C, int N,
                     long sxb, long sxk, long sxc, long syb, long syk, long syc, hipStream_t st)
{
#define MGP_AGG_CASE(CT) return launch_agg_fwd<CT, V>(X, G, Y, B, K, C, N, sxb, sxk, sxc, syb, syk, syc, st)
    if (C <= 4) MGP_AGG_CASE(4);
    if (C <= 6) MGP_AGG_CASE(6);
    if (C <= 8) MGP_AGG_CASE(8);
    if (C <= 16) MGP_AGG_CASE(16);
    MGP_AGG_CASE(32);
#undef MGP_AGG_CASE
}

template <int CT>
int launch_agg_bwd_x(const float* dY, const float* G, float* dX, int B, int K, int C, int N,
                     long sgb, long sgk, long sgc, long sdb, long sdk, long sdc, hipStream_t st)
{
    const int nrt = mgp_ceil_div(N, BWX_ROWS);
    const int nchunks = mgp_ceil_div(C, CT);
    int NC = 8192 / CT;
    if (NC > N) NC = N;
    const size_t lds = (size_t)CT * NC * sizeof(float);
    dim3 grid(nrt * nchunks, K, B);
    hipLaunchKernelGGL((agg_bwd_x_kernel<CT>), grid, dim3(AGG_THREADS), lds, st,
                       dY, G, dX, K, C, N, nrt, NC, sgb, sgk, sgc, sdb, sdk, sdc);
    return mgp_launch_status();
}

}  // namespace

extern "C" int mgp_agg_fwd(const float* X, const float* G, float* Y, int B, int K, int C, int N,
                           long sxb, long sxk, long sxc, long syb, long syk, long syc, void* stream)
{
    if (B < 0 || K <= 0 || C <= 0 || N <= 0) return MGP_EINVAL;
    if (B == 0) return MGP_OK;
    MGP_CHECK_PTR(X); MGP_CHECK_PTR(G); MGP_CHECK_PTR(Y);
    if (B > 65535 || K > 65535) return MGP_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    mgp_clear_error();
    const bool vec = (N % 4 == 0) && mgp_aligned16(G);
    const int nblk = N > 64 ? 2 : 1;
    const bool quads = mgp_aligned16(X) && sxb % 4 == 0 && sxk % 4 == 0 && sxc % 4 == 0;     // X quads are aligned float4 loads
    static const int agg_form = getenv("MGP_AGG_FORM") ? atoi(getenv("MGP_AGG_FORM")) : 4;   // 8: the eight-wave kernel (A/B switch)
    if (vec && N >= 16 && N <= 128 && C <= 8 && quads && agg_form != 8 && (size_t)B * K <= 0x7FFFFFFFull) {
#define MGP_AG4_CASE(S_, NH_, NB_, NT_) return C <= 4 ? launch_agg_fwd_mfma4<S_, 1, NH_, NB_, NT_>(X, G, Y, B, K, C, N, sxb, sxk, sxc, syb, syk, syc, st) \
                                                      : launch_agg_fwd_mfma4<S_, 2, NH_, NB_, NT_>(X, G, Y, B, K, C, N, sxb, sxk, sxc, syb, syk, syc, st)
        if (N <= 64) MGP_AG4_CASE(4, 4, 1, true);              // one column block, four row parts of 4 steps (16 rows each)
        if (N <= 112) {
            if (agg_form == 43 && K == 3 && N > 64)            // (A/B: one workgroup per episode, the taps in sequence; the barriers need whole waves: N > 64 has two column blocks)
                return C <= 4 ? launch_agg_fwd_mfma3<7, 1>(X, G, Y, B, C, N, sxb, sxk, sxc, syb, syk, syc, st)
                              : launch_agg_fwd_mfma3<7, 2>(X, G, Y, B, C, N, sxb, sxk, sxc, syb, syk, syc, st);
            if (agg_form == 44 || agg_form == 45 || agg_form == 46) { // (A/B: 7 / 10 / 4 of the 14 requests up front at EVERY batch size)
                const int d = agg_form == 44 ? 7 : (agg_form == 45 ? 10 : 4);
#define MGP_AG4_D(D_) return C <= 4 ? launch_agg_fwd_mfma4<14, 1, 2, 2, true, D_>(X, G, Y, B, K, C, N, sxb, sxk, sxc, syb, syk, syc, st) \
                                    : launch_agg_fwd_mfma4<14, 2, 2, 2, true, D_>(X, G, Y, B, K, C, N, sxb, sxk, sxc, syb, syk, syc, st)
                if (d == 7) MGP_AG4_D(7); if (d == 10) MGP_AG4_D(10); MGP_AG4_D(4);
#undef MGP_AG4_D
            }
            if (agg_form == 41) MGP_AG4_CASE(14, 2, 2, false); // (A/B: default cache policy)
            if (agg_form == 42) MGP_AG4_CASE(7, 4, 2, true);   // (A/B: eight waves of 7 steps)
            MGP_AG4_CASE(14, 2, 2, true);                      // two column blocks, two row parts of 14 steps
        }
        MGP_AG4_CASE(16, 2, 2, true);
#undef MGP_AG4_CASE
    }
    if (vec && N >= 16 && N <= 128 && C <= 8 && K * nblk <= AGM_THREADS / 64 && quads) {
#define MGP_AGM_CASE(S_) return C <= 4 ? launch_agg_fwd_mfma<S_, 1>(X, G, Y, B, K, C, N, nblk, sxb, sxk, sxc, syb, syk, syc, st) \
                                       : launch_agg_fwd_mfma<S_, 2>(X, G, Y, B, K, C, N, nblk, sxb, sxk, sxc, syb, syk, syc, st)
        if (N <= 64) MGP_AGM_CASE(16);
        if (N <= 112) MGP_AGM_CASE(28);
        MGP_AGM_CASE(32);
#undef MGP_AGM_CASE
    }
    if (vec) return dispatch_agg_fwd<4>(X, G, Y, B, K, C, N, sxb, sxk, sxc, syb, syk, syc, st);
    return dispatch_agg_fwd<1>(X, G, Y, B, K, C, N, sxb, sxk, sxc, syb, syk, syc, st);
}

extern "C" int mgp_agg_bwd_x(const float* dY, const float* G, float* dX, int B, int K, int C, int N,
                             long sgb, long sgk, long sgc, long sdb, long sdk, long sdc, void* stream)
{
    if (B < 0 || K <= 0 || C <= 0 || N <= 0) return MGP_EINVAL;
    if (B == 0) return MGP_OK;
    MGP_CHECK_PTR(dY); MGP_CHECK_PTR(G); MGP_CHECK_PTR(dX);
    if (B > 65535 || K > 65535) return MGP_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    mgp_clear_error();
    if (C <= 2) return launch_agg_bwd_x<2>(dY, G, dX, B, K, C, N, sgb, sgk, sgc, sdb, sdk, sdc, st);
    if (C <= 4) return launch_agg_bwd_x<4>(dY, G, dX, B, K, C, N, sgb, sgk, sgc, sdb, sdk, sdc, st);
    return launch_agg_bwd_x<8>(dY, G, dX, B, K, C, N, sgb, sgk, sgc, sdb, sdk, sdc, st);
}
