// One DAGGER update in two launches (reference learner/gnn_dagger.py:85-93):
//
//     pred = actor(X, G); loss = F.mse_loss(pred, target); loss.backward(); Adam.step()
//
// Kernel 1 (train_tile_kernel): one workgroup per (batch item, 16 agent columns).  The aggregation, the filter + tanh MLP,
//   the MSE gradient and the whole parameter backward of those 16 columns happen inside the workgroup, activations never
//   leave LDS (the five-launch path writes them to HBM as `saved` and reads them back), and the workgroup emits one partial
//   of every dW / db plus its share of the squared error.  The three GEMM shapes of the MLP (W . in, delta . in^T,
//   W^T . delta) run as 16 x 16 tiles of v_mfma_f32_16x16x4_f32 dealt to the four waves -- measured neutral against the VALU
//   loops they replaced (29.9k -> 29.1k cycles per workgroup): the kernel is ten barrier-separated phases of 2-3k cycles
//   each, and inside a phase the operand fetch, the dependent accumulate chain, tanh and the LDS store are a latency chain
//   (one wave per SIMD, 140 workgroups on 256 CUs: nothing to overlap with) whichever pipe does the multiplies.
// Kernel 2 (train_reduce_kernel): adds the partials in a fixed order (bit-reproducible run to run) into the flat gradient
//   W_0 | b_0 | W_1 | b_1 | ... -- the order torch enumerates Actor.parameters() -- writes the loss, and, for the
//   single-GPU update, applies Adam in the same pass (each workgroup owns 64 parameters; the last workgroup to finish
//   advances the device-side step counter).  A data-parallel run stops after the gradient (all-reduce, then mgp_adam_step*).
//
// B = 20, N = 100, K = 3, 6-32-32-2: 140 workgroups; 15.9 (forward) + 4.5 (mse) + 10.0 (backward) + 4.7 (scatter) + 4.0 (adam)
// microseconds of kernel time in the five-launch graph.
#include <math.h>
#include <string.h>
#include <stdlib.h>
#include "mgp_common.h"
#include "p2p_device.h"
#include "mgp_device.h"

namespace {

constexpr int TS_THREADS = 256;
constexpr int TS_COLS = 16;               // agent columns per workgroup
constexpr int TS_CSH = 4;                 // log2(TS_COLS)
constexpr int TS_GROUPS = TS_THREADS / TS_COLS;
constexpr int TS_CS = TS_COLS + 1;        // odd LDS row stride: conflict-free walks along a column
constexpr int TS_MAXW = 128;              // layer widths covered (F*K <= 64)
constexpr int TS_LDS_LIMIT = 150 * 1024;  // [18 -> 128 -> 128 -> 2] at N = 100: 141 KB

struct TrainParams {
    const float* W[MGP_MAX_LAYERS];
    const float* b[MGP_MAX_LAYERS];
    int dims[MGP_MAX_LAYERS + 1];
    int poff[MGP_MAX_LAYERS];             // offset of layer l's (dW, db) block inside one partial / the flat gradient
    int ioff[MGP_MAX_LAYERS];             // offset (floats) of layer l's input rows inside the LDS activation area
    int n_layers;
    const float* flat;                    // W_0 | b_0 | W_1 | ... when the caller's buffers are contiguous in that order
    const long* idx;                      // batch item b reads row idx[cursor[0] * B + b] of X / G / target (replay gather
    const int* cursor;                    //  fused into the update); nullptr: row b
};

// LDS (floats): xs [K*F][N] | gs [K][MC][16] (a chunk of G rows, this tile's columns) | red [MP][FK][16] |
//               acts (inputs of every layer, [rows][17]) | d0, d1 [maxw][17] | wall [Ptot] (W_l then b_l, flat order)
// Every first-round global read (G tile, X, parameters, targets) is issued before the first LDS write, so a workgroup
// pays ONE exposed memory latency -- the first version staged weights layer by layer (23 us in fourteen serial round
// trips).  The workgroup runs one wave per SIMD, so nothing hides an LDS round trip either: the inner loops load a
// batch of operands into registers, then multiply (sched_barrier keeps the compiler from re-serialising them).
#ifdef MGP_TS_PROFILE
__device__ unsigned long long mgp_ts_stamps[32];
#define TS_STAMP(i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) mgp_ts_stamps[i] = __builtin_readcyclecounter(); } while (0)
#else
#define TS_STAMP(i) do { } while (0)
#endif
constexpr int TS_GB = 20;                 // G values in flight per thread (covers N = 100 with K = 3: 20 rows per piece)

// n floats global -> LDS with TS_THREADS threads, eight loads in flight per thread before the first LDS write
__device__ __forceinline__ void stage_lds(float* dst, const float* __restrict__ src, int n, int tid, int first = 0)
{
    for (int base = first; base < n; base += 8 * TS_THREADS) {
        float r[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { const int i = base + q * TS_THREADS + tid; r[q] = src[min(i, n - 1)]; }
#pragma unroll
        for (int q = 0; q < 8; ++q) { const int i = base + q * TS_THREADS + tid; if (i < n) dst[i] = r[q]; }
    }
}

// element e of the G tile [K][mcn rows][16 columns] of rows mc0.. of episode b, columns n0..n0+15 (0 past column N)
__device__ __forceinline__ float g_tile_elem(const float* __restrict__ Gb, int e, int ne, int mcn, int mc0, int n0, int N)
{
    const float inv_mcn = 1.0f / (float)mcn;
    e = min(e, ne - 1);
    const int c = e & (TS_COLS - 1), r = e >> TS_CSH;
    const int k = (int)(((float)r + 0.5f) * inv_mcn), m = r - k * mcn;      // r / mcn exactly (r < 2^11, mcn <= 128)
    const float v = Gb[((size_t)k * N + mc0 + m) * N + min(n0 + c, N - 1)];
    return (n0 + c < N) ? v : 0.f;
}

__device__ __forceinline__ void stage_g_tile(float* gs, const float* __restrict__ Gb, int ne, int mcn, int mc0, int n0,
                                             int N, int tid, int first)
{
    for (int base = first; base < ne; base += 8 * TS_THREADS) {
        float r[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) r[q] = g_tile_elem(Gb, base + q * TS_THREADS + tid, ne, mcn, mc0, n0, N);
#pragma unroll
        for (int q = 0; q < 8; ++q) { const int e = base + q * TS_THREADS + tid; if (e < ne) gs[e] = r[q]; }
    }
}

// One 16 x 16 tile D += A (16 x kmax) . B (kmax x 16) on v_mfma_f32_16x16x4_f32 (exact fp32, k-ordered fma chain), operands
// in LDS: A[i][k] = pa[i * a_si + k * a_sk] for i < imax, B[k][j] = pb[k * b_sk + j * b_sj] for j < jmax, zero elsewhere
// and for k >= kmax.  Lane (li, lq) supplies A[li][4 s + lq] and B[4 s + lq][li] of k-step s and ends up with
// D[4 lq + rr][li] in acc[rr].  Every load is unconditional on a clamped index and masked afterwards (hipcc turns a
// conditional LDS load into a branch and waits for it at the join: eight serial round trips per group instead of one);
// the operands of four k-steps are loaded before their MFMAs are issued (one wave per SIMD: nothing else hides the trip).
struct TsOperand { const float* p; int s_outer, s_k, omax; };      // outer = i (A) or j (B)

__device__ __forceinline__ f32x4 ts_mfma_tile(int kmax, const TsOperand& A, const TsOperand& Bm, f32x4 acc, int li, int lq)
{
    const int ksteps = (kmax + 3) >> 2;
    const bool ia = li < A.omax, jb = li < Bm.omax;
    const float* pa = A.p + min(li, A.omax - 1) * A.s_outer;
    const float* pb = Bm.p + min(li, Bm.omax - 1) * Bm.s_outer;
    for (int s0 = 0; s0 < ksteps; s0 += 4) {
        float a[4], bv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = 4 * (s0 + q) + lq, kc = min(k, kmax - 1);
            a[q] = pa[kc * A.s_k];
            bv[q] = pb[kc * Bm.s_k];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool kon = 4 * (s0 + q) + lq < kmax;
            a[q] = (kon && ia) ? a[q] : 0.f;
            bv[q] = (kon && jb) ? bv[q] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], bv[q], acc, 0, 0, 0);
    }
    return acc;
}

// PRE: X is the AGGREGATED first-layer input Z (B, F K, N), rows in W_0's column order f K + k (mgp_replay_aggregate builds it
// from the frame ring along the bit rows; G unused): the tile's F K x 16 block goes straight into `acts`, no xs / gs / red areas
// (LDS: acts | d0 | d1 | wall -- 17 KB for 18-32-32-2, so several workgroups share a CU), no aggregation phase.
// CW > 0: the reference's policy shapes compiled in (cfg/dagger.cfg, cfg/k.cfg: F K = CFK inputs, two hidden layers of CW, two outputs) --
// layer count, widths, LDS offsets and every MFMA tile's k-range are constants, the layer loops are unrolled (the generic form
// computes clamped operand addresses with run-time strides per load: most of a phase's instructions).
template <bool PRE, int CW = 0, int CFK = 18>
__global__ __launch_bounds__(TS_THREADS)
void train_tile_kernel(const float* __restrict__ X, const float* __restrict__ G, const float* __restrict__ target,
                       float* __restrict__ part, TrainParams P, int Pstride_, int K, int F, int N, int MP, int MC,
                       int acts_floats_, int maxw_, float grad_scale)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr bool CS = CW > 0;
    const int tid = threadIdx.x;
    const int n0 = blockIdx.x * TS_COLS, b = blockIdx.y;
    const int FK = CS ? CFK : F * K;
    const int L = CS ? 3 : P.n_layers;
    auto dim_ = [&](int l) -> int { return CS ? (l == 0 ? 6 : (l == 3 ? 2 : CW)) : P.dims[l]; };
    auto poff_ = [&](int l) -> int { return CS ? (l == 0 ? 0 : (l == 1 ? (CFK + 1) * CW : (CFK + 1) * CW + CW * CW + CW)) : P.poff[l]; };
    auto ioff_ = [&](int l) -> int { return CS ? (l == 0 ? 0 : (l == 1 ? CFK * TS_CS : (CFK + CW) * TS_CS)) : P.ioff[l]; };
    const int Pstride = CS ? (CFK + 1) * CW + CW * CW + CW + 2 * CW + 2 + 1 : Pstride_;
    const int acts_floats = CS ? (CFK + 2 * CW) * TS_CS : acts_floats_;
    const int maxw = CS ? CW : maxw_;
    const int nA = dim_(L);
    float* xs = smem;                                         // [K*F][N]
    float* gs = xs + (PRE ? (size_t)0 : (size_t)FK * N);      // [K][MC][16]   one chunk of G rows, this tile's columns
    float* red = gs + (PRE ? (size_t)0 : (size_t)K * MC * TS_COLS);   // [MP][FK][16]
    float* acts = red + (PRE ? (size_t)0 : (size_t)MP * FK * TS_COLS);
    float* d0 = acts + acts_floats;
    float* d1 = d0 + (size_t)maxw * TS_CS;
    float* wall = d1 + (size_t)maxw * TS_CS;
    float* my = part + ((size_t)b * gridDim.x + blockIdx.x) * Pstride;
    const int col = tid & (TS_COLS - 1), g = tid >> TS_CSH;
    const size_t src = (P.idx != nullptr) ? (size_t)P.idx[(size_t)P.cursor[0] * gridDim.y + b] : (size_t)b;
    const float* Gb = G + src * K * N * N;
    const float* xb = X + src * FK * N;
    TS_STAMP(0);

    // ---- every first-round global read goes out before the first LDS write: G tile (20 per thread), X, parameters
    //      (8 each), the targets -- one exposed memory latency for the whole workgroup
    if (PRE) {
        const int nz = FK * TS_COLS, nw = Pstride - 1;          // F K <= 64: at most four tile elements per thread
        float rz[4], rw[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = min(q * TS_THREADS + tid, nz - 1), r = e >> TS_CSH, c = e & (TS_COLS - 1);
            rz[q] = xb[(size_t)r * N + min(n0 + c, N - 1)];
        }
        if (P.flat != nullptr) {
#pragma unroll
            for (int q = 0; q < 8; ++q) rw[q] = P.flat[min(q * TS_THREADS + tid, nw - 1)];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = q * TS_THREADS + tid, r = e >> TS_CSH, c = e & (TS_COLS - 1);
            if (e < nz) acts[r * TS_CS + c] = (n0 + c < N) ? rz[q] : 0.f;
        }
        if (P.flat != nullptr) {
#pragma unroll
            for (int q = 0; q < 8; ++q) { const int i = q * TS_THREADS + tid; if (i < nw) wall[i] = rw[q]; }
            stage_lds(wall, P.flat, nw, tid, 8 * TS_THREADS);
        } else {
            for (int l = 0; l < L; ++l) {
                const int cin = (l == 0) ? FK : dim_(l), cout = dim_(l + 1);
                stage_lds(wall + poff_(l), P.W[l], cout * cin, tid);
                stage_lds(wall + poff_(l) + cout * cin, P.b[l], cout, tid);
            }
        }
    } else {
        const int mcn = min(MC, N), ne = K * mcn * TS_COLS, nx = FK * N, nw = Pstride - 1;
        float rg[TS_GB], rx[8], rw[8];
#pragma unroll
        for (int q = 0; q < TS_GB; ++q) rg[q] = g_tile_elem(Gb, q * TS_THREADS + tid, ne, mcn, 0, n0, N);
#pragma unroll
        for (int q = 0; q < 8; ++q) rx[q] = xb[min(q * TS_THREADS + tid, nx - 1)];
        if (P.flat != nullptr) {
#pragma unroll
            for (int q = 0; q < 8; ++q) rw[q] = P.flat[min(q * TS_THREADS + tid, nw - 1)];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < TS_GB; ++q) { const int e = q * TS_THREADS + tid; if (e < ne) gs[e] = rg[q]; }
#pragma unroll
        for (int q = 0; q < 8; ++q) { const int i = q * TS_THREADS + tid; if (i < nx) xs[i] = rx[q]; }
        if (P.flat != nullptr) {
#pragma unroll
            for (int q = 0; q < 8; ++q) { const int i = q * TS_THREADS + tid; if (i < nw) wall[i] = rw[q]; }
            stage_lds(wall, P.flat, nw, tid, 8 * TS_THREADS);
        } else {
            for (int l = 0; l < L; ++l) {
                const int cin = (l == 0) ? FK : dim_(l), cout = dim_(l + 1);
                stage_lds(wall + poff_(l), P.W[l], cout * cin, tid);
                stage_lds(wall + poff_(l) + cout * cin, P.b[l], cout, tid);
            }
        }
        stage_lds(xs, xb, nx, tid, 8 * TS_THREADS);
        stage_g_tile(gs, Gb, ne, mcn, 0, n0, N, tid, TS_GB * TS_THREADS);
    }
    // targets of this workgroup's outputs, in the accumulator layout of the last layer's MFMA tile
    float tgt[4];                                                  // lane (li, lq) of wave w: outputs 16 w + 4 lq + rr of column li
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int o = 16 * (tid >> 6) + 4 * ((tid & 63) >> 4) + rr;
        tgt[rr] = (o < nA && n0 + (tid & 15) < N) ? target[(src * nA + o) * N + n0 + (tid & 15)] : 0.f;
    }

    if (!PRE) {
    // ---- aggregation y[k,f,col] = sum_m X[b,k,f,m] G[b,k,m,n0+col]: thread = (col, tap k, piece mp of the chunk's rows).
    //      Eight accumulators whatever F is (feature index clamped, surplus results dropped): no branch in the row loop.
    const bool agg_on = g < K * MP;
    const int ak = agg_on ? g / MP : 0, amp = agg_on ? g - ak * MP : 0;
    float acc[8];
#pragma unroll
    for (int f = 0; f < 8; ++f) acc[f] = 0.f;
    for (int mc0 = 0; mc0 < N; mc0 += MC) {
        const int mcn = min(MC, N - mc0);
        if (mc0 > 0) {                                          // N > MC only: next chunk of G rows
            __syncthreads();
            stage_g_tile(gs, Gb, K * mcn * TS_COLS, mcn, mc0, n0, N, tid, 0);
        }
        __syncthreads();
        if (mc0 == 0) TS_STAMP(1);
        if (agg_on) {
            const int piece = (mcn + MP - 1) / MP;
            const int p0 = amp * piece, p1 = min(mcn, p0 + piece);
            const float* gk = gs + (size_t)ak * mcn * TS_COLS + col;
            const float* xk = xs + (size_t)ak * F * N + mc0;
            for (int m = p0; m < p1; m += 4) {
                float gq[4], xq[4][8];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int mm = min(m + q, p1 - 1);
                    gq[q] = gk[mm * TS_COLS];
#pragma unroll
                    for (int f = 0; f < 8; ++f) xq[q][f] = xk[min(f, F - 1) * N + mm];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float gvq = (m + q < p1) ? gq[q] : 0.f;
#pragma unroll
                    for (int f = 0; f < 8; ++f) acc[f] = fmaf(xq[q][f], gvq, acc[f]);
                }
            }
        }
    }
    if (agg_on) {
#pragma unroll
        for (int f = 0; f < 8; ++f)
            if (f < F) red[((size_t)amp * FK + (f * K + ak)) * TS_COLS + col] = acc[f];      // row (f,k) = W_0's column order
    }
    __syncthreads();
    TS_STAMP(2);
    for (int i = tid; i < FK * TS_COLS; i += TS_THREADS) {
        const int r = i >> TS_CSH, c = i & (TS_COLS - 1);
        float s = 0.f;
        for (int mp = 0; mp < MP; ++mp) s += red[((size_t)mp * FK + r) * TS_COLS + c];        // fixed order
        acts[r * TS_CS + c] = s;
    }

    }
    // ---- filter + tanh MLP on the 16 columns, on the matrix pipe: out (cout x 16) = W (cout x cin) . in (cin x 16), one
    //      16-row m-tile per wave and trip; the input of every layer stays in LDS for the backward pass
    const int lane = tid & 63, wave = tid >> 6, li = lane & 15, lq = lane >> 4;
    float* dcur = d0;
    float* dnext = d1;
    float sq = 0.f;
    auto forward_layer = [&](const int l) __attribute__((always_inline)) {
        const int cin = (l == 0) ? FK : dim_(l);
        const int cout = dim_(l + 1);
        const float* in = acts + ioff_(l);
        const float* wl = wall + poff_(l);
        __syncthreads();                                            // inputs complete
        TS_STAMP(3 + l);
        for (int mt = wave; 16 * mt < cout; mt += TS_THREADS / 64) {
            f32x4 acc;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) { const int o = 16 * mt + 4 * lq + rr; acc[rr] = (o < cout) ? wl[cout * cin + o] : 0.f; }
            acc = ts_mfma_tile(cin, TsOperand{wl + 16 * mt * cin, cin, 1, cout - 16 * mt}, TsOperand{in, 1, TS_CS, TS_COLS}, acc, li, lq);
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int o = 16 * mt + 4 * lq + rr;
                if (o < cout) {
                    if (l < L - 1) {
                        acts[ioff_(l + 1) + o * TS_CS + li] = tanh_fast(acc[rr]);
                    } else {
                        // d loss / d pred = 2 (pred - target) / n  (reference F.mse_loss, mean over every element)
                        float d = 0.f;
                        if (n0 + li < N) d = acc[rr] - (mt == wave ? tgt[rr] : target[(src * nA + o) * N + n0 + li]);
                        dcur[o * TS_CS + li] = grad_scale * d;
                        sq = fmaf(d, d, sq);
                    }
                }
            }
        }
    };
    if (CS) { forward_layer(0); forward_layer(1); forward_layer(2); }
    else for (int l = 0; l < L; ++l) forward_layer(l);
    // squared-error share of this workgroup (fixed order: wave sums, then the four added pairwise)
    {
        __shared__ float shq[TS_THREADS / 64];
        sq = mgp_wave_sum(sq);
        if ((tid & 63) == 0) shq[tid >> 6] = sq;
        __syncthreads();
        if (tid == 0) my[Pstride - 1] = (shq[0] + shq[1]) + (shq[2] + shq[3]);
    }
    TS_STAMP(8);

    // ---- backward, parameters only (ind_agg = 0: nothing flows into X or G -- reference actor.py:64-71 inputs are leaves)
    //      dW (cout x cin) = delta (cout x 16) . in^T (16 x cin) and, for the layer below, W^T (cin x cout) . delta (cout x 16):
    //      16 x 16 tiles dealt to the four waves
    auto backward_layer = [&](const int l) __attribute__((always_inline)) {
        const int cin = (l == 0) ? FK : dim_(l);
        const int cout = dim_(l + 1);
        const float* in = acts + ioff_(l);
        const float* wl = wall + poff_(l);
        if (l < L - 1) __syncthreads();                             // dcur complete (the loss block synchronised l = L-1)
        TS_STAMP(9 + (L - 1 - l));
        float* myl = my + poff_(l);
        for (int o = tid; o < cout; o += TS_THREADS) {
            float s = 0.f;
#pragma unroll
            for (int cl = 0; cl < TS_COLS; ++cl) s += dcur[o * TS_CS + cl];
            myl[(size_t)cout * cin + o] = s;
        }
        {
            const int ntc = (cin + 15) >> 4, ntiles = ((cout + 15) >> 4) * ntc;
            for (int t = wave; t < ntiles; t += TS_THREADS / 64) {
                const int mt = t / ntc, nt = t - mt * ntc;
                f32x4 acc = ts_mfma_tile(TS_COLS, TsOperand{dcur + 16 * mt * TS_CS, TS_CS, 1, cout - 16 * mt},
                                         TsOperand{in + 16 * nt * TS_CS, TS_CS, 1, cin - 16 * nt}, f32x4{0.f, 0.f, 0.f, 0.f}, li, lq);
                const int c = 16 * nt + li;
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int o = 16 * mt + 4 * lq + rr;
                    if (o < cout && c < cin) myl[(size_t)o * cin + c] = acc[rr];
                }
            }
        }
        if (l > 0) {
            for (int mt = wave; 16 * mt < cin; mt += TS_THREADS / 64) {
                f32x4 acc = ts_mfma_tile(cout, TsOperand{wl + 16 * mt, 1, cin, cin - 16 * mt}, TsOperand{dcur, 1, TS_CS, TS_COLS},
                                         f32x4{0.f, 0.f, 0.f, 0.f}, li, lq);
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int c = 16 * mt + 4 * lq + rr;
                    if (c < cin) {
                        const float z = in[c * TS_CS + li];
                        dnext[c * TS_CS + li] = acc[rr] * (1.f - z * z);
                    }
                }
            }
            float* t = dcur; dcur = dnext; dnext = t;
        }
    };
    if (CS) { backward_layer(2); backward_layer(1); backward_layer(0); }
    else for (int l = L - 1; l >= 0; --l) backward_layer(l);
    TS_STAMP(14);
}

struct AdamArgs {
    float* p;            // nullptr: gradient only
    float* m;
    float* v;
    int* step_dev;
    int* ticket;
    float lr, b1, b2, eps;
    int* cursor;         // replay-indexed updates: advanced together with the step counter (may be nullptr)
    float* loss_hist;    // loss of update u of the round -> loss_hist[u % hist_cap] (may be nullptr)
    int hist_cap;
};

// workgroup = 64 entries of the partial x 16 interleaved groups of tiles; entry Ptot is the squared error
constexpr int TR_GROUPS = 16;
constexpr int TR_BATCH = 10;               // tiles per group fetched in one batch (B = 20, N = 100: 140 tiles = 9 per group)
// P2P: data-parallel update -- the entry this thread owns (gradient element or squared error) is exchanged with the other
// ranks' (p2p_device.h: pushed into every peer's mailbox, the W values added in rank order, / W) between the local
// reduction and Adam, so the whole data-parallel update stays two launches and every rank applies bit-identical steps.
// EPW entries per workgroup (x 1024 / EPW interleaved groups of tiles): 64 for the reference's batch (140 tiles: 28 workgroups);
// 16 when there are many tiles (B = 20, N = 1000: 1,260 tiles, 8.7 MB of partials -- 109 workgroups instead of 28 pull them in)
template <bool P2P, int EPW>
__global__ __launch_bounds__(64 * TR_GROUPS)
void train_reduce_kernel(const float* __restrict__ part, int ntiles, int Pstride, float* __restrict__ flat_grad,
                         float* __restrict__ loss, float inv_n, AdamArgs A, P2PDev X)
{
    unsigned xseq = 0;
    bool bad = false;                                    // P2P: an exchange timed out, now or earlier (sticky status word):
    if (P2P) {                                           // a late peer's share counted as 0 would be a wrong, rank-divergent
        xseq = (unsigned)X.ctl[0] + 1u;                  // step, so THIS entry is not stepped -- but `bad` is per thread: in the
        bad = X.ctl[2] != 0;                             // update where a peer is late, entries that arrived in time ARE stepped.
    }                                                    // The weights are not to be used after that: the host raises at its next
                                                         // status check and restores the round's starting point (DAGGER.end_updates)
    constexpr int NG = 64 * TR_GROUPS / EPW;          // groups of tiles
    __shared__ float sh[NG][EPW];
    __shared__ float shc[2];
    const int pl = threadIdx.x % EPW, g = threadIdx.x / EPW;
    const int i = blockIdx.x * EPW + pl;
    const int Ptot = Pstride - 1;
    // bias corrections once per workgroup (two fp64 pow: ~0.7 us), by one thread off wave 0 -- computed between the issue of
    // that wave's partial loads and their first use (in front of them it delayed the wave, and so the barrier, by its length)
    const bool corr = A.p != nullptr && threadIdx.x == 64;
    int step_now = 0;
    if (corr) step_now = *A.step_dev;
    // the parameter and its moments are requested together with the first partials (they do not depend on the gradient: one
    // memory round trip less between the reduction and the step)
    float pp = 0.f, pm = 0.f, pv = 0.f;
    const bool owner = A.p != nullptr && g == 0 && i < Ptot;
    if (owner) { pp = A.p[i]; pm = A.m[i]; pv = A.v[i]; }
    auto bias_corrections = [&]() {
        const double step = (double)(step_now + 1);
        shc[0] = (float)((double)A.lr / (1.0 - pow((double)A.b1, step)));
        shc[1] = (float)sqrt(1.0 - pow((double)A.b2, step));
    };
    if (corr && g >= ntiles) bias_corrections();        // (fewer tiles than groups: this thread's loop below is empty)
    float s = 0.f;
    if (i < Pstride) {
        // TR_BATCH tiles of this group are requested together (the partials come from the other XCDs' tile workgroups, i.e.
        // from memory: one dependent round trip per tile otherwise -- 79 of them at B = 20, N = 1000, 23.6 us for this kernel)
        // and added in ascending tile order, batch after batch
        for (int t0 = g; t0 < ntiles; t0 += NG * TR_BATCH) {
            float v[TR_BATCH];
#pragma unroll
            for (int q = 0; q < TR_BATCH; ++q) {
                const int t = t0 + NG * q;
                v[q] = part[(size_t)min(t, ntiles - 1) * Pstride + i];
            }
            if (corr && t0 == g) {
                __builtin_amdgcn_sched_barrier(0);
                bias_corrections();
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int q = 0; q < TR_BATCH; ++q) s += (t0 + NG * q < ntiles) ? v[q] : 0.f;
        }
    }
    sh[g][pl] = s;
    __syncthreads();
    if (g == 0 && i < Pstride) {
        s = 0.f;
#pragma unroll
        for (int q = 0; q < NG; ++q) s += sh[q][pl];
        if (i == Ptot) s *= inv_n;
        if (P2P) s = p2p_exchange_mean(X, i, s, xseq, &bad);
        if (i == Ptot) {
            if (loss != nullptr) loss[0] = s;
            if (A.loss_hist != nullptr) A.loss_hist[*A.cursor % A.hist_cap] = s;   // (the cursor moves after every
        } else {                                                                         //  workgroup is through: ticket below)
            flat_grad[i] = s;
            if (A.p != nullptr && !bad) {                // torch.optim.Adam defaults, same expressions as mgp_adam_step_dev
                const float one_m_b1 = (float)(1.0 - (double)A.b1), one_m_b2 = (float)(1.0 - (double)A.b2);
                mgp_adam_elem(pp, pm, pv, s, one_m_b1, A.b2, one_m_b2, shc[0], shc[1], A.eps);
                A.p[i] = pp; A.m[i] = pm; A.v[i] = pv;
            }
        }
    }
    if (A.p != nullptr) {
        // every workgroup has read the step counter by now; the last one to get here advances it
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            if (atomicAdd(A.ticket, 1) == (int)gridDim.x - 1) {
                *A.ticket = 0;
                *A.step_dev += 1;
                if (A.cursor != nullptr) *A.cursor += 1;
                if (P2P) X.ctl[0] = (int)xseq;
            }
        }
    }
}

struct TrainPlan {
    int MP, MC, acts_floats, maxw, maxin, Ptot, ntx;
    size_t lds;
    int poff[MGP_MAX_LAYERS], ioff[MGP_MAX_LAYERS];
};

bool make_train_plan(const int* dims, int n_layers, int B, int K, int N, TrainPlan* pl, bool pre = false)
{
    if (dims == nullptr || n_layers <= 0 || n_layers > MGP_MAX_LAYERS || K <= 0 || K > TS_GROUPS || N <= 0 || B <= 0) return false;
    const int F = dims[0];
    if (F <= 0 || F > 8 || F * K > 64) return false;
    for (int i = 1; i <= n_layers; ++i) if (dims[i] <= 0 || dims[i] > TS_MAXW) return false;
    const int FK = F * K;
    pl->MP = TS_GROUPS / K;
    pl->MC = N < 128 ? N : 128;                                          // G rows staged per chunk
    int maxw = dims[n_layers], maxin = FK, poff = 0, ioff = 0;
    for (int l = 0; l < n_layers; ++l) {
        const int cin = (l == 0) ? FK : dims[l], cout = dims[l + 1];
        pl->poff[l] = poff; poff += cout * cin + cout;
        pl->ioff[l] = ioff; ioff += cin * TS_CS;
        if (cout > maxw) maxw = cout;
        if (cin > maxw && l > 0) maxw = cin;
        if (cin > maxin) maxin = cin;
    }
    pl->acts_floats = ioff; pl->maxw = maxw; pl->maxin = maxin; pl->Ptot = poff;
    pl->ntx = (N + TS_COLS - 1) / TS_COLS;
    if ((long)B * pl->ntx > 8192 || B > 65535) return false;             // partials: tiles x (Ptot + 1) floats
    pl->lds = ((pre ? (size_t)0 : (size_t)FK * N + (size_t)K * pl->MC * TS_COLS + (size_t)pl->MP * FK * TS_COLS) + ioff
               + (size_t)2 * maxw * TS_CS + (size_t)poff) * sizeof(float);
    return pl->lds <= TS_LDS_LIMIT;
}

int launch_train(const float* X, const float* G, const float* target, const float* const* W, const float* const* b,
                 const int* dims, int n_layers, float* flat_grad, float* loss, float* workspace, const AdamArgs& A,
                 int B, int K, int N, hipStream_t st, const long* idx = nullptr, const MgpP2P* comm = nullptr, bool pre = false)
{
    TrainPlan pl;
    if (!make_train_plan(dims, n_layers, B, K, N, &pl, pre)) return MGP_EUNSUPPORTED;
    TrainParams P;
    P.n_layers = n_layers;
    for (int i = 0; i <= n_layers; ++i) P.dims[i] = dims[i];
    for (int l = 0; l < n_layers; ++l) {
        MGP_CHECK_PTR(W[l]); MGP_CHECK_PTR(b[l]);
        P.W[l] = W[l]; P.b[l] = b[l]; P.poff[l] = pl.poff[l]; P.ioff[l] = pl.ioff[l];
    }
    bool contiguous = true;
    for (int l = 0; l < n_layers; ++l) {
        const int cin = (l == 0) ? dims[0] * K : dims[l];
        contiguous = contiguous && W[l] == W[0] + pl.poff[l] && b[l] == W[l] + (size_t)dims[l + 1] * cin;
    }
    P.flat = contiguous ? W[0] : nullptr;
    P.idx = idx; P.cursor = A.cursor;
    const int Pstride = pl.Ptot + 1;
    const long n_out = (long)B * dims[n_layers] * N;
    mgp_clear_error();
    static const bool no_cs = getenv("MGP_TRAIN_CS") != nullptr && atoi(getenv("MGP_TRAIN_CS")) == 0;      // (A/B switch)
    const bool cs_shape = pre && !no_cs && n_layers == 3 && dims[0] == 6 && dims[3] == 2 && dims[1] == dims[2] && K >= 1 && K <= 4
                          && (dims[1] == 32 || (dims[1] == 64 && K == 3));
#define MGP_TS_CS(CW_, CFK_)                                                                                                      \
    do {                                                                                                                          \
        if (mgp_allow_dyn_lds(reinterpret_cast<const void*>(train_tile_kernel<true, CW_, CFK_>), pl.lds) != hipSuccess) return MGP_ELAUNCH; \
        hipLaunchKernelGGL((train_tile_kernel<true, CW_, CFK_>), dim3(pl.ntx, B), dim3(TS_THREADS), pl.lds, st, X, G, target, workspace, \
                           P, Pstride, K, dims[0], N, pl.MP, pl.MC, pl.acts_floats, pl.maxw, 2.0f / (float)n_out);               \
    } while (0)
    if (cs_shape) {                                           // cfg/dagger.cfg, cfg/k.cfg (K = 1..4), hidden_size 64
        if (dims[1] == 64) MGP_TS_CS(64, 18);
        else if (K == 1) MGP_TS_CS(32, 6);
        else if (K == 2) MGP_TS_CS(32, 12);
        else if (K == 3) MGP_TS_CS(32, 18);
        else MGP_TS_CS(32, 24);
#undef MGP_TS_CS
    } else if (pre) {
        if (mgp_allow_dyn_lds(reinterpret_cast<const void*>(train_tile_kernel<true>), pl.lds) != hipSuccess) return MGP_ELAUNCH;
        hipLaunchKernelGGL(train_tile_kernel<true>, dim3(pl.ntx, B), dim3(TS_THREADS), pl.lds, st, X, G, target, workspace, P,
                           Pstride, K, dims[0], N, pl.MP, pl.MC, pl.acts_floats, pl.maxw, 2.0f / (float)n_out);
    } else if (!no_cs && n_layers == 3 && dims[0] == 6 && K == 3 && dims[1] == 32 && dims[2] == 32 && dims[3] == 2) {
        // the dense form (callers that hand over delay_state / delay_gso: the reference's gradient_step signature) at cfg/dagger.cfg
        if (mgp_allow_dyn_lds(reinterpret_cast<const void*>(train_tile_kernel<false, 32, 18>), pl.lds) != hipSuccess) return MGP_ELAUNCH;
        hipLaunchKernelGGL((train_tile_kernel<false, 32, 18>), dim3(pl.ntx, B), dim3(TS_THREADS), pl.lds, st, X, G, target, workspace, P,
                           Pstride, K, dims[0], N, pl.MP, pl.MC, pl.acts_floats, pl.maxw, 2.0f / (float)n_out);
    } else {
        if (mgp_allow_dyn_lds(reinterpret_cast<const void*>(train_tile_kernel<false>), pl.lds) != hipSuccess) return MGP_ELAUNCH;
        hipLaunchKernelGGL(train_tile_kernel<false>, dim3(pl.ntx, B), dim3(TS_THREADS), pl.lds, st, X, G, target, workspace, P,
                           Pstride, K, dims[0], N, pl.MP, pl.MC, pl.acts_floats, pl.maxw, 2.0f / (float)n_out);
    }
    int rc = mgp_launch_status();
    if (rc != MGP_OK) return rc;
    const bool wide = (long)B * pl.ntx > 512;                 // many tiles: narrower workgroups, more of them (see the kernel)
    if (comm != nullptr) {
        if (!comm->connected || comm->dev.n < Pstride || A.p == nullptr) return MGP_EINVAL;
        if (wide)
            hipLaunchKernelGGL((train_reduce_kernel<true, 16>), dim3((unsigned)((Pstride + 15) / 16)), dim3(64 * TR_GROUPS), 0, st,
                               workspace, B * pl.ntx, Pstride, flat_grad, loss, (float)(1.0 / (double)n_out), A, comm->dev);
        else
            hipLaunchKernelGGL((train_reduce_kernel<true, 64>), dim3((unsigned)((Pstride + 63) / 64)), dim3(64 * TR_GROUPS), 0, st,
                               workspace, B * pl.ntx, Pstride, flat_grad, loss, (float)(1.0 / (double)n_out), A, comm->dev);
        return mgp_launch_status();
    }
    P2PDev none;
    memset(&none, 0, sizeof(none));
    if (wide)
        hipLaunchKernelGGL((train_reduce_kernel<false, 16>), dim3((unsigned)((Pstride + 15) / 16)), dim3(64 * TR_GROUPS), 0, st,
                           workspace, B * pl.ntx, Pstride, flat_grad, loss, (float)(1.0 / (double)n_out), A, none);
    else
        hipLaunchKernelGGL((train_reduce_kernel<false, 64>), dim3((unsigned)((Pstride + 63) / 64)), dim3(64 * TR_GROUPS), 0, st,
                           workspace, B * pl.ntx, Pstride, flat_grad, loss, (float)(1.0 / (double)n_out), A, none);
    return mgp_launch_status();
}

}  // namespace

extern "C" int mgp_train_supported(const int* dims, int n_layers, int B, int K, int N)
{
    TrainPlan pl;
    return make_train_plan(dims, n_layers, B, K, N, &pl) ? 1 : 0;
}

extern "C" long mgp_train_workspace(const int* dims, int n_layers, int B, int K, int N)
{
    TrainPlan pl;
    if (!make_train_plan(dims, n_layers, B, K, N, &pl) && !make_train_plan(dims, n_layers, B, K, N, &pl, true)) return 0;
    return (long)B * pl.ntx * (pl.Ptot + 1) + 1;                         // + the ticket word of mgp_train_step
}

extern "C" int mgp_train_grads(const float* X, const float* G, const float* target, const float* const* W,
                               const float* const* b, const int* dims, int n_layers, float* flat_grad, float* loss,
                               float* workspace, int B, int K, int N, void* stream)
{
    if (dims == nullptr || W == nullptr || b == nullptr) return MGP_EINVAL;
    if (B <= 0 || K <= 0 || N <= 0 || n_layers <= 0 || n_layers > MGP_MAX_LAYERS) return MGP_EINVAL;
    MGP_CHECK_PTR(X); MGP_CHECK_PTR(G); MGP_CHECK_PTR(target); MGP_CHECK_PTR(flat_grad); MGP_CHECK_PTR(workspace);
    if (loss != nullptr && (reinterpret_cast<uintptr_t>(loss) & 3u)) return MGP_EALIGN;
    AdamArgs A = {nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, 0.f, 0.f, 0.f, nullptr, nullptr, 0};
    return launch_train(X, G, target, W, b, dims, n_layers, flat_grad, loss, workspace, A, B, K, N,
                        static_cast<hipStream_t>(stream));
}

static int train_step_impl(const float* X, const float* G, const float* target, const long* idx, int* cursor, float* loss_hist,
                           int hist_cap, float* flat_param, float* flat_grad, float* m, float* v, const int* dims,
                           int n_layers, float lr, float beta1, float beta2, float eps, int* step_dev, float* loss,
                           float* workspace, int B, int K, int N, void* stream, const MgpP2P* comm = nullptr, bool pre = false)
{
    if (dims == nullptr) return MGP_EINVAL;
    if (B <= 0 || K <= 0 || N <= 0 || n_layers <= 0 || n_layers > MGP_MAX_LAYERS) return MGP_EINVAL;
    MGP_CHECK_PTR(X); MGP_CHECK_PTR(target); MGP_CHECK_PTR(flat_param); MGP_CHECK_PTR(flat_grad);
    if (!pre) MGP_CHECK_PTR(G);
    MGP_CHECK_PTR(m); MGP_CHECK_PTR(v); MGP_CHECK_PTR(step_dev); MGP_CHECK_PTR(workspace);
    if (loss != nullptr && (reinterpret_cast<uintptr_t>(loss) & 3u)) return MGP_EALIGN;
    TrainPlan pl;
    if (!make_train_plan(dims, n_layers, B, K, N, &pl, pre)) return MGP_EUNSUPPORTED;
    const float* W[MGP_MAX_LAYERS];
    const float* b[MGP_MAX_LAYERS];
    for (int l = 0; l < n_layers; ++l) {
        const int cin = (l == 0) ? dims[0] * K : dims[l];
        W[l] = flat_param + pl.poff[l];
        b[l] = W[l] + (size_t)dims[l + 1] * cin;
    }
    int* ticket = reinterpret_cast<int*>(workspace + (size_t)B * pl.ntx * (pl.Ptot + 1));
    AdamArgs A = {flat_param, m, v, step_dev, ticket, lr, beta1, beta2, eps, cursor, loss_hist, hist_cap};
    return launch_train(X, G, target, W, b, dims, n_layers, flat_grad, loss, workspace, A, B, K, N,
                        static_cast<hipStream_t>(stream), idx, comm, pre);
}

extern "C" int mgp_train_step(const float* X, const float* G, const float* target, float* flat_param, float* flat_grad,
                              float* m, float* v, const int* dims, int n_layers, float lr, float beta1, float beta2,
                              float eps, int* step_dev, float* loss, float* workspace, int B, int K, int N, void* stream)
{
    return train_step_impl(X, G, target, nullptr, nullptr, nullptr, 0, flat_param, flat_grad, m, v, dims, n_layers, lr, beta1,
                           beta2, eps, step_dev, loss, workspace, B, K, N, stream);
}

extern "C" int mgp_train_step_indexed(const float* Xr, const float* Gr, const float* Yr, const long* idx, int* cursor,
                                      float* loss_hist, int hist_cap, float* flat_param, float* flat_grad, float* m, float* v,
                                      const int* dims, int n_layers, float lr, float beta1, float beta2, float eps,
                                      int* step_dev, float* workspace, int B, int K, int N, void* stream)
{
    if (idx == nullptr || cursor == nullptr || loss_hist == nullptr || hist_cap <= 0) return MGP_EINVAL;
    if ((reinterpret_cast<uintptr_t>(idx) & 7u) || (reinterpret_cast<uintptr_t>(cursor) & 3u)) return MGP_EALIGN;
    return train_step_impl(Xr, Gr, Yr, idx, cursor, loss_hist, hist_cap, flat_param, flat_grad, m, v, dims, n_layers, lr, beta1,
                           beta2, eps, step_dev, nullptr, workspace, B, K, N, stream);
}


// Data-parallel forms: the same two launches with the one-shot exchange of p2p_device.h between the local reduction and
// Adam (every rank of `comm` must make the same calls in the same order).  idx / cursor / loss_hist may all be NULL
// (mgp_train_step semantics, loss -> loss[0]) or all given (mgp_train_step_indexed semantics).
extern "C" int mgp_train_step_p2p(const float* X, const float* G, const float* target, const long* idx, int* cursor,
                                  float* loss_hist, int hist_cap, float* flat_param, float* flat_grad, float* m, float* v,
                                  const int* dims, int n_layers, float lr, float beta1, float beta2, float eps,
                                  int* step_dev, float* loss, float* workspace, int B, int K, int N, MgpP2P* comm, void* stream)
{
    if (comm == nullptr) return MGP_EINVAL;
    const bool indexed = idx != nullptr || cursor != nullptr || loss_hist != nullptr;
    if (indexed) {
        if (idx == nullptr || cursor == nullptr || loss_hist == nullptr || hist_cap <= 0) return MGP_EINVAL;
        if ((reinterpret_cast<uintptr_t>(idx) & 7u) || (reinterpret_cast<uintptr_t>(cursor) & 3u)) return MGP_EALIGN;
    }
    return train_step_impl(X, G, target, idx, cursor, loss_hist, hist_cap, flat_param, flat_grad, m, v, dims, n_layers, lr, beta1,
                           beta2, eps, step_dev, loss, workspace, B, K, N, stream, comm);
}


// The same update on the AGGREGATED first-layer input Z (B, F K, N), rows f K + k in W_0's column order -- what
// mgp_replay_aggregate builds from the frame ring without forming the operator slices (reference gnn_dagger.py:85-93 with
// actor.py:64-75's aggregation already applied; parameters are the only leaves, so nothing flows back through it).
extern "C" int mgp_train_agg_supported(const int* dims, int n_layers, int B, int K, int N)
{
    TrainPlan pl;
    return make_train_plan(dims, n_layers, B, K, N, &pl, true) ? 1 : 0;
}

extern "C" int mgp_train_grads_agg(const float* Z, const float* target, const float* const* W, const float* const* b,
                                   const int* dims, int n_layers, float* flat_grad, float* loss, float* workspace, int B, int K,
                                   int N, void* stream)
{
    if (dims == nullptr || W == nullptr || b == nullptr) return MGP_EINVAL;
    if (B <= 0 || K <= 0 || N <= 0 || n_layers <= 0 || n_layers > MGP_MAX_LAYERS) return MGP_EINVAL;
    MGP_CHECK_PTR(Z); MGP_CHECK_PTR(target); MGP_CHECK_PTR(flat_grad); MGP_CHECK_PTR(workspace);
    if (loss != nullptr && (reinterpret_cast<uintptr_t>(loss) & 3u)) return MGP_EALIGN;
    AdamArgs A = {nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, 0.f, 0.f, 0.f, nullptr, nullptr, 0};
    return launch_train(Z, nullptr, target, W, b, dims, n_layers, flat_grad, loss, workspace, A, B, K, N,
                        static_cast<hipStream_t>(stream), nullptr, nullptr, true);
}

// idx / cursor / loss_hist: all NULL (mgp_train_step semantics, loss -> loss[0]) or all given (mgp_train_step_indexed
// semantics: batch item i reads row idx[cursor[0] * B + i] of Z / target); comm: NULL or the data-parallel exchange
// (mgp_train_step_p2p semantics).
extern "C" int mgp_train_step_agg(const float* Z, const float* target, const long* idx, int* cursor, float* loss_hist,
                                  int hist_cap, float* flat_param, float* flat_grad, float* m, float* v, const int* dims,
                                  int n_layers, float lr, float beta1, float beta2, float eps, int* step_dev, float* loss,
                                  float* workspace, int B, int K, int N, MgpP2P* comm, void* stream)
{
    const bool indexed = idx != nullptr || cursor != nullptr || loss_hist != nullptr;
    if (indexed) {
        if (idx == nullptr || cursor == nullptr || loss_hist == nullptr || hist_cap <= 0) return MGP_EINVAL;
        if ((reinterpret_cast<uintptr_t>(idx) & 7u) || (reinterpret_cast<uintptr_t>(cursor) & 3u)) return MGP_EALIGN;
    }
    return train_step_impl(Z, nullptr, target, idx, cursor, loss_hist, hist_cap, flat_param, flat_grad, m, v, dims, n_layers, lr,
                           beta1, beta2, eps, step_dev, loss, workspace, B, K, N, stream, comm, true);
}
