// Policy step on the FACTORED state, for episodes too large for the LDS-resident rollout (N > 256; BASELINE configs[2]:
// N = 1000).  Same mathematics as rollout.hip:
//
//     y_j(t) = x_{t-j} G_j(t) = x_{t-j} A_t A_{t-1} ... A_{t-j+1}       (state_with_delay.py:44-47, actor.py:64-71)
//
// evaluated left to right along the networks' membership BIT ROWS (mgp_flock_step_sparse writes them: 128 B per row at
// N = 1000 where the dense row is 4 KB), so a step never touches a dense N x N operator: the two-launch dense path moves
// 12 MB per episode and step at N = 1000, this one ~0.3 MB.  The state lives in HBM/L2 between launches:
//     bits  (B, H, N, NW) u64   membership rows of the last H = K-1 networks (ring over time)
//     wrow  (B, H, N)     f32   row weights (1/deg or 1)
//     feat  (B, K, N, 8)  f32   features x_t .. x_{t-K+1} as (N, 8) rows (6 used), ring over time
// One step = K launches: K-2 gather stages (sp_gather_kernel: stage q multiplies the running products of taps j >= q by
// A_{t-q+1}), the policy tail (sp_policy_kernel: last gather stage + filter GEMM + tanh MLP on MFMA + output layer, the
// arithmetic of rollout.hip's phases A-C) and the simulator (mgp_flock_step_sparse).  sp_to_dense_kernel rebuilds the
// dense delayed operator slices of the reference contract from the bit rows when a caller asks for them.
#include "mgp_common.h"
#include "mgp_device.h"
#include "rollout_common.h"
#include "sparse_common.h"

namespace {

#ifdef MGP_SP_PROFILE
__device__ unsigned long long mgp_sp_stamps[2 * 16 * 16];  // [kernel][wave][stamp]
#define SP_STAMP(k, i) do { if (blockIdx.x == 1 && blockIdx.y == 3 && (threadIdx.x & 63) == 0) mgp_sp_stamps[((k) * 16 + (threadIdx.x >> 6)) * 16 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define SP_STAMP(k, i) do { } while (0)
#endif

constexpr int SP_THREADS = 256;
constexpr int SP_COLS = 64;               // agent columns per workgroup: 4 lanes per column / one 16-column MFMA tile per wave

// sum over the set bits m of `w` (base index `base`) of wq[m] * src[m][0..5]   (src rows are 8 floats)
__device__ __forceinline__ void sp_gather_word(unsigned long long w, int base, const float* __restrict__ wq,
                                               const float* __restrict__ src, float (&sa)[6])
{
    while (w) {
        const int m = base + __builtin_ctzll(w);
        w &= w - 1ull;
        const float gv = wq[m];
        const float4 x0 = *reinterpret_cast<const float4*>(src + (size_t)m * 8);
        const float2 x1 = *reinterpret_cast<const float2*>(src + (size_t)m * 8 + 4);
        sa[0] = fmaf(x0.x, gv, sa[0]); sa[1] = fmaf(x0.y, gv, sa[1]); sa[2] = fmaf(x0.z, gv, sa[2]);
        sa[3] = fmaf(x0.w, gv, sa[3]); sa[4] = fmaf(x1.x, gv, sa[4]); sa[5] = fmaf(x1.y, gv, sa[5]);
    }
}

// (v . A)[n, 0..5] for column n = blockIdx.x * 64 + (tid >> 2): the quad's four lanes take a quarter of the row's words
// each and are added by DPP; valid in every lane of the quad
__device__ __forceinline__ void sp_gather_column(const unsigned long long* __restrict__ brow, int NW,
                                                 const float* __restrict__ wq, const float* __restrict__ src, int part,
                                                 bool live, float (&sa)[6])
{
#pragma unroll
    for (int f = 0; f < 6; ++f) sa[f] = 0.f;
    if (live) {
        const int wpl = NW >> 2;                              // words per lane (NW is a multiple of 8)
        for (int wd = part * wpl; wd < (part + 1) * wpl; ++wd) sp_gather_word(brow[wd], 64 * wd, wq, src, sa);
    }
#pragma unroll
    for (int f = 0; f < 6; ++f) { sa[f] += dpp_f<0xB1>(sa[f]); sa[f] += dpp_f<0x4E>(sa[f]); }
}

struct SpTaps {                            // one gather stage: taps processed together (same network)
    const float* src[SP_MAXTAPS];
    float* dst[SP_MAXTAPS];
    long ss[SP_MAXTAPS], ds[SP_MAXTAPS];  // batch strides (floats)
};

// grid: x = column tile, y = tap of the stage, z = b
__global__ __launch_bounds__(SP_THREADS)
void sp_gather_kernel(const unsigned long long* __restrict__ bits, long sBb, const float* __restrict__ wq, long sWb,
                      SpTaps T, int N, int NW)
{
    const int tid = threadIdx.x, b = blockIdx.z, tap = blockIdx.y;
    const int n = blockIdx.x * SP_COLS + (tid >> 2), part = tid & 3;
    const bool live = n < N;
    float sa[6];
    sp_gather_column(bits + (size_t)b * sBb + (size_t)min(n, N - 1) * NW, NW, wq + (size_t)b * sWb,
                     T.src[tap] + (size_t)b * T.ss[tap], part, live, sa);
    if (live && part == 0) {
        float* d = T.dst[tap] + (size_t)b * T.ds[tap] + (size_t)n * 8;
        *reinterpret_cast<float4*>(d) = make_float4(sa[0], sa[1], sa[2], sa[3]);
        *reinterpret_cast<float4*>(d + 4) = make_float4(sa[4], sa[5], 0.f, 0.f);
    }
}

struct SpPolicy {
    const float* tap[SP_MAXTAPS + 1];     // tap 0: x_t rows; taps 1..K-2: finished products; tap K-1: INPUT of the last stage
    long ts[SP_MAXTAPS + 1];              // batch strides (floats)
    const unsigned long long* bits;       // network of the last stage, A_{t-K+2} (unused when K == 1)
    long sBb;
    const float* wq; long sWb;
    const float* image;                   // weight image (sp_weight_image_kernel), wtot floats
    int wtot;
    const unsigned short* nbr;            // compact neighbour rows of the last stage's network (staged form with lists), or NULL
    long sNbr;
};

// DAGGER data collection on the factored state (gnn_dagger.py:154-178; the semantics of rollout.hip's collecting build):
// the policy tail also files the FRAME of the state the step starts from -- features x_t, bit rows and row weights of A_t,
// the expert's action for x_t (the label), the age -- and hands the simulator the expert's action instead of the policy's
// when the lane's counter-based coin says so.  Frame planes point at this step's ring slot: [B][...].
struct SpCollect {
    float* feat;                          // [B][6][N]
    unsigned long long* bits;             // [B][N][NW]
    float* wrow;                          // [B][N]
    float* label;                         // [B][2][N]
    int* age;                             // [B]
    const unsigned long long* net;        // rows of A_t (ring slot hs), batch stride sNb words
    const float* wnet;                    // its row weights, batch stride sWn
    long sNb, sWn;
    const float* expert;                  // (B,N,2): the expert's action for x_t (by-product of the simulator kernel)
    const float* beta;                    // (B)
    const unsigned int* episode;          // (B)
    unsigned int seed;
    int age_now;
};

// grid: x = tile of 64 columns, y = b.  LDS: act [64][RO_CS] | weight image
template <bool CL>
__global__ __launch_bounds__(SP_THREADS)
void sp_policy_kernel(SpPolicy P, float* __restrict__ action, int K, int N, int NW, unsigned long long dimsA,
                      unsigned int dims8, unsigned long long woffA, unsigned long long woffB, int n_layers, SpCollect C)
{
    extern __shared__ __attribute__((aligned(16))) float spm[];
    float* act = spm;
    float* wl = spm + SP_COLS * RO_CS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.y;
    const int c0 = blockIdx.x * SP_COLS;
    const int FK = 6 * K;
    {
        const float4* src = reinterpret_cast<const float4*>(P.image);
        float4* dst = reinterpret_cast<float4*>(wl);
        for (int i = tid; i < (P.wtot + 3) / 4; i += SP_THREADS) dst[i] = src[i];
        float4* za = reinterpret_cast<float4*>(act);
        for (int i = tid; i < SP_COLS * RO_CS / 4; i += SP_THREADS) za[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    // taps whose product is finished (tap 0 = x_t itself): channel (f, j) of column c -> MFMA B-fragment slot rpos(f K + j)
    for (int j = 0; j < K - 1 || (j == 0 && K == 1); ++j) {
        const float* src = P.tap[j] + (size_t)b * P.ts[j];
        for (int i = tid; i < SP_COLS * 8; i += SP_THREADS) {
            const int c = i >> 3, f = i & 7;
            if (f < 6 && c0 + c < N) act[c * RO_CS + rpos(f * K + j)] = src[(size_t)(c0 + c) * 8 + f];
        }
    }
    bool expert_drives = false;
    if (CL) {
        const double bq = floor((double)C.beta[b] * 4294967296.0);            // P(expert drives) in units of 2^-32
        const unsigned long long thr = bq <= 0.0 ? 0ull : (bq >= 4294967296.0 ? 4294967296ull : (unsigned long long)bq);
        expert_drives = (unsigned long long)dagger_coin(C.seed, C.episode[b], (unsigned int)C.age_now) < thr;
        const int cols = min(SP_COLS, N - c0);
        // frame of x_t, this tile's columns / rows: features (6,N) from the (N,8) rows, label (2,N) from the expert's (N,2)
        const float* xt = P.tap[0] + (size_t)b * P.ts[0];
        float* ff = C.feat + (size_t)b * 6 * N;
        for (int i = tid; i < 6 * SP_COLS; i += SP_THREADS) {
            const int f = i >> 6, c = i & 63;
            if (c < cols) ff[(size_t)f * N + c0 + c] = xt[(size_t)(c0 + c) * 8 + f];
        }
        const float* ex = C.expert + (size_t)b * N * 2;
        float* lb = C.label + (size_t)b * 2 * N;
        for (int i = tid; i < 2 * SP_COLS; i += SP_THREADS) {
            const int a = i >> 6, c = i & 63;
            if (c < cols) lb[(size_t)a * N + c0 + c] = ex[(size_t)(c0 + c) * 2 + a];
        }
        const unsigned long long* nr = C.net + (size_t)b * C.sNb + (size_t)c0 * NW;
        unsigned long long* fb = C.bits + ((size_t)b * N + c0) * NW;
        for (int i = tid; i < cols * NW; i += SP_THREADS) fb[i] = nr[i];
        if (tid < cols) C.wrow[(size_t)b * N + c0 + tid] = C.wnet[(size_t)b * C.sWn + c0 + tid];
        if (blockIdx.x == 0 && tid == 0) C.age[b] = C.age_now;
    }
    if (K >= 2) {                                              // last tap: its last factor is applied here
        const int c = tid >> 2, part = tid & 3, n = c0 + c;
        float sa[6];
        sp_gather_column(P.bits + (size_t)b * P.sBb + (size_t)min(n, N - 1) * NW, NW, P.wq + (size_t)b * P.sWb,
                         P.tap[K - 1] + (size_t)b * P.ts[K - 1], part, n < N, sa);
        if (part == 0 && n < N) {
#pragma unroll
            for (int f = 0; f < 6; ++f) act[c * RO_CS + rpos(f * K + K - 1)] = sa[f];
        }
    }
    __syncthreads();
    // filter GEMM + tanh hidden layers: wave w owns columns 16 w .. 16 w + 15 (rollout.hip phase B)
    const int li = lane & 15, lq = lane >> 4;
    float* pcol = act + (wave * 16 + li) * RO_CS;
    for (int l = 0; l < n_layers - 1; ++l) {
        const int cin = (l == 0) ? FK : ((l < 8) ? (int)((dimsA >> (8 * l)) & 255ull) : (int)dims8);
        const int cout = (l + 1 < 8) ? (int)((dimsA >> (8 * (l + 1))) & 255ull) : (int)dims8;
        const int MT = mtiles(cout);
        const float* wfrag = wl + (int)((((l < 4) ? woffA : woffB) >> (16 * (l & 3))) & 0xFFFFull);
        if (MT == 2) ro_mlp_cols<2>(pcol, wfrag + lane * RO_WFS, wfrag + 2 * 64 * RO_WFS + lq * 4, pad4(cin) / 4, lq);
        else ro_mlp_cols<1>(pcol, wfrag + lane * RO_WFS, wfrag + 64 * RO_WFS + lq * 4, pad4(cin) / 4, lq);
    }
    // 2-wide output layer: lane L takes column L >> 2 of the wave's tile and 8 channels, quad sum by DPP (phase C)
    const int lo_ = n_layers - 1;
    const float* w2 = wl + (int)((((lo_ < 4) ? woffA : woffB) >> (16 * (lo_ & 3))) & 0xFFFFull);
    const int ccol = wave * 16 + (lane >> 2), cg = lane & 3;
    const float* zsrc = act + ccol * RO_CS + cg * RO_KS;
    const float4 z0 = *reinterpret_cast<const float4*>(zsrc);
    const float4 z1 = *reinterpret_cast<const float4*>(zsrc + 4);
    const float zc[8] = {z0.x, z0.y, z0.z, z0.w, z1.x, z1.y, z1.z, z1.w};
    f32x2 u2 = {0.f, 0.f}, u2b = {0.f, 0.f};
#pragma unroll
    for (int s_ = 0; s_ < RO_KS; s_ += 2) {
        const float2 wa = *reinterpret_cast<const float2*>(w2 + 2 * (4 * s_ + cg));
        const float2 wb = *reinterpret_cast<const float2*>(w2 + 2 * (4 * (s_ + 1) + cg));
        u2 = __builtin_elementwise_fma((f32x2){zc[s_], zc[s_]}, (f32x2){wa.x, wa.y}, u2);
        u2b = __builtin_elementwise_fma((f32x2){zc[s_ + 1], zc[s_ + 1]}, (f32x2){wb.x, wb.y}, u2b);
    }
    u2 = u2 + u2b;
    float ux = u2.x, uy = u2.y;
    ux += dpp_f<0xB1>(ux); uy += dpp_f<0xB1>(uy);
    ux += dpp_f<0x4E>(ux); uy += dpp_f<0x4E>(uy);
    if (cg == 0 && c0 + ccol < N) {
        const float2 bb = *reinterpret_cast<const float2*>(w2 + 2 * 4 * RO_KS);
        float ax = ux + bb.x, ay = uy + bb.y;
        if (CL && expert_drives) {                             // gnn_dagger.py:157-161: the stored label drives the step
            const float2 e2 = *reinterpret_cast<const float2*>(C.expert + ((size_t)b * N + c0 + ccol) * 2);
            ax = e2.x; ay = e2.y;
        }
        action[((size_t)b * 2 + 0) * N + c0 + ccol] = ax;
        action[((size_t)b * 2 + 1) * N + c0 + ccol] = ay;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// LDS-staged forms of the two kernels above: the default wherever an episode's source rows fit the LDS (N <= ~2400).
// Measured on the direct forms (64 x 1000, K = 3): gather 10.5 us, policy tail 13.8 us for ~1 MB of useful reads -- every
// neighbour entry is three lane-divergent global loads (weight, 16 + 8 bytes of the row), ~2.7 M cache-line lookups per
// stage, and the L1's one-line-per-cycle tag rate is what the kernel waits for (requesting three entries of a word together
// made it SLOWER: 2.4 x the lookups, 1.45 x the time).  Here a workgroup of 1024 threads owns 256 columns, copies the
// episode's source rows and row weights into the LDS with coalesced 16-byte loads (68 KB for two taps at N = 1000) and gathers
// from there: four lanes per column as before, no divergent global load left except the column's own bit words.
constexpr int SPL_THREADS = 1024;
constexpr int SPL_COLS = 256;             // columns per workgroup: 4 lanes per column, one 16-column MFMA tile per wave
constexpr size_t SPL_LDS_MAX = 156 * 1024;

// sum over the set bits m of `w` of lw[m] * ls[tap][m][0..5], all taps of the stage along one scan of the word
template <int NT>
__device__ __forceinline__ void spl_gather_word(unsigned long long w, int base, const float* lw, const float* ls, int N,
                                                float (&sa)[NT][6])
{
    while (w) {
        const int m = base + __builtin_ctzll(w);
        w &= w - 1ull;
        const float gv = lw[m];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float* r = ls + ((size_t)t * N + m) * 8;
            const float4 x0 = *reinterpret_cast<const float4*>(r);
            const float2 x1 = *reinterpret_cast<const float2*>(r + 4);
            sa[t][0] = fmaf(x0.x, gv, sa[t][0]); sa[t][1] = fmaf(x0.y, gv, sa[t][1]); sa[t][2] = fmaf(x0.z, gv, sa[t][2]);
            sa[t][3] = fmaf(x0.w, gv, sa[t][3]); sa[t][4] = fmaf(x1.x, gv, sa[t][4]); sa[t][5] = fmaf(x1.y, gv, sa[t][5]);
        }
    }
}

// one neighbour entry m: lw[m] * ls[tap][m][0..5] for all taps of the stage
template <int NT>
__device__ __forceinline__ void spl_gather_entry(int m, const float* lw, const float* ls, int N, float (&sa)[NT][6])
{
    const float gv = lw[m];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const float* r = ls + ((size_t)t * N + m) * 8;
        const float4 x0 = *reinterpret_cast<const float4*>(r);
        const float2 x1 = *reinterpret_cast<const float2*>(r + 4);
        sa[t][0] = fmaf(x0.x, gv, sa[t][0]); sa[t][1] = fmaf(x0.y, gv, sa[t][1]); sa[t][2] = fmaf(x0.z, gv, sa[t][2]);
        sa[t][3] = fmaf(x0.w, gv, sa[t][3]); sa[t][4] = fmaf(x1.x, gv, sa[t][4]); sa[t][5] = fmaf(x1.y, gv, sa[t][5]);
    }
}

// The column from its compact LIST (mgp_flock_step_cells_nbr: 32 bytes per row instead of the 128-byte bit row at N = 1000
// to request, and the entries dealt evenly): `lst` = this lane's 8 bytes of the row = entries part, part + 4, part + 8,
// part + 12; the count sits in the last u16 of lane 3.  Count 0xFFFF (more than 15 neighbours): the bit row, from global.
template <int NT>
__device__ __forceinline__ void spl_gather_list(uint2 lst, int part, const unsigned long long* __restrict__ brow, int wpl,
                                                const float* lw, const float* ls, int N, bool live, float (&sa)[NT][6])
{
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int f = 0; f < 6; ++f) sa[t][f] = 0.f;
    const unsigned int cntw = dpp_u<0xFF>(lst.y) >> 16;
    if (live) {
        if (cntw != 0xFFFFu) {
            const int mine = ((int)cntw - part + 3) >> 2;
            for (int k = 0; k < mine; ++k) {
                const unsigned int pair = (k < 2) ? lst.x : lst.y;
                spl_gather_entry<NT>((int)((k & 1) ? (pair >> 16) : (pair & 0xFFFFu)), lw, ls, N, sa);
            }
        } else {
            for (int q = 0; q < wpl; ++q) spl_gather_word<NT>(brow[q], 64 * (part * wpl + q), lw, ls, N, sa);
        }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int f = 0; f < 6; ++f) { sa[t][f] += dpp_f<0xB1>(sa[t][f]); sa[t][f] += dpp_f<0x4E>(sa[t][f]); }
}

// the column's words of this lane (a quarter of the row): the first four from registers (requested before the staging
// barrier; N <= 1024 has no more), the rest from global memory
template <int NT>
__device__ __forceinline__ void spl_gather_column(const unsigned long long (&wreg)[4], const unsigned long long* __restrict__ brow,
                                                  int wpl, int word0, const float* lw, const float* ls, int N, bool live,
                                                  float (&sa)[NT][6])
{
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int f = 0; f < 6; ++f) sa[t][f] = 0.f;
    if (live) {
#pragma unroll
        for (int q = 0; q < 4; ++q) spl_gather_word<NT>(wreg[q], 64 * (word0 + q), lw, ls, N, sa);
        for (int q = 4; q < wpl; ++q) spl_gather_word<NT>(brow[q], 64 * (word0 + q), lw, ls, N, sa);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int f = 0; f < 6; ++f) { sa[t][f] += dpp_f<0xB1>(sa[t][f]); sa[t][f] += dpp_f<0x4E>(sa[t][f]); }
}

// grid: x = tile of 256 columns, y = b.  LDS: lw [Np] | ls [NT][N][8]
template <int NT, bool LS>
__global__ __launch_bounds__(SPL_THREADS)
void spl_gather_kernel(const unsigned long long* __restrict__ bits, long sBb, const float* __restrict__ wq, long sWb,
                       SpTaps T, int N, int NW, const unsigned short* __restrict__ nbr, long sNb)
{
    extern __shared__ __attribute__((aligned(16))) float spm[];
    const int Np = (N + 3) & ~3;
    float* lw = spm;
    float* ls = spm + Np;
    const int tid = threadIdx.x, b = blockIdx.y;
    const int n = blockIdx.x * SPL_COLS + (tid >> 2), part = tid & 3;
    const bool live = n < N;
    const int wpl = NW >> 2;                                   // words per lane (NW is a multiple of 8)
    const unsigned long long* brow = bits + (size_t)b * sBb + (size_t)min(n, N - 1) * NW + part * wpl;
    unsigned long long wreg[4];
    SP_STAMP(0, 0);
#ifdef MGP_SP_PROFILE
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    SP_STAMP(0, 5);
#endif
    uint2 lst = make_uint2(0u, 0u);
    if (LS) lst = *reinterpret_cast<const uint2*>(nbr + (size_t)b * sNb + (size_t)min(n, N - 1) * 16 + 4 * part);
    else {
#pragma unroll
        for (int q = 0; q < 4; ++q) wreg[q] = (q < wpl) ? brow[q] : 0ull;
    }
    // every request of the copy is issued before the first LDS store (a copy loop is one memory round trip PER ITERATION:
    // 8k cycles for five of them); N > 1024 finishes with plain loops
    float4 rr[NT][2];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const float4* s4 = reinterpret_cast<const float4*>(T.src[t] + (size_t)b * T.ss[t]);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int i = tid + q * SPL_THREADS;
            rr[t][q] = (i < 2 * N) ? s4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    const float rw = (tid < N) ? wq[(size_t)b * sWb + tid] : 0.f;
#ifdef MGP_SP_PROFILE
    SP_STAMP(0, 6);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SP_STAMP(0, 7);
#endif
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        float4* d4 = reinterpret_cast<float4*>(ls + (size_t)t * N * 8);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int i = tid + q * SPL_THREADS;
            if (i < 2 * N) d4[i] = rr[t][q];
        }
        if (2 * N > 2 * SPL_THREADS) {
            const float4* s4 = reinterpret_cast<const float4*>(T.src[t] + (size_t)b * T.ss[t]);
            for (int i = tid + 2 * SPL_THREADS; i < 2 * N; i += SPL_THREADS) d4[i] = s4[i];
        }
    }
    if (tid < N) lw[tid] = rw;
    for (int i = tid + SPL_THREADS; i < N; i += SPL_THREADS) lw[i] = wq[(size_t)b * sWb + i];
    SP_STAMP(0, 1);
    __syncthreads();
    SP_STAMP(0, 2);
    float sa[NT][6];
    if (LS) spl_gather_list<NT>(lst, part, brow, wpl, lw, ls, N, live, sa);
    else spl_gather_column<NT>(wreg, brow, wpl, part * wpl, lw, ls, N, live, sa);
    SP_STAMP(0, 3);
#pragma unroll
    for (int t = 0; t < NT; ++t)
        if (live && part == (t & 3)) {
            float* d = T.dst[t] + (size_t)b * T.ds[t] + (size_t)n * 8;
            *reinterpret_cast<float4*>(d) = make_float4(sa[t][0], sa[t][1], sa[t][2], sa[t][3]);
            *reinterpret_cast<float4*>(d + 4) = make_float4(sa[t][4], sa[t][5], 0.f, 0.f);
        }
    SP_STAMP(0, 4);
}

// grid: x = tile of 256 columns, y = b.  LDS: act [256][RO_CS] | weight image | lw [Np] | ls [N][8]
template <bool CL, bool LS>
__global__ __launch_bounds__(SPL_THREADS)
void spl_policy_kernel(SpPolicy P, float* __restrict__ action, int K, int N, int NW, unsigned long long dimsA,
                       unsigned int dims8, unsigned long long woffA, unsigned long long woffB, int n_layers, SpCollect C)
{
    extern __shared__ __attribute__((aligned(16))) float spm[];
    const int Np = (N + 3) & ~3, wt4 = (P.wtot + 3) & ~3;
    float* act = spm;
    float* wl = spm + SPL_COLS * RO_CS;
    float* lw = wl + wt4;
    float* ls = lw + Np;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.y;
    const int c0 = blockIdx.x * SPL_COLS;
    const int FK = 6 * K;
    const int gc = tid >> 2, part = tid & 3, gn = c0 + gc;
    const int wpl = NW >> 2;
    const unsigned long long* brow = P.bits + (size_t)b * P.sBb + (size_t)min(gn, N - 1) * NW + part * wpl;
    unsigned long long wreg[4] = {0ull, 0ull, 0ull, 0ull};
    SP_STAMP(1, 0);
    uint2 lst = make_uint2(0u, 0u);
    if (K >= 2) {
        if (LS) lst = *reinterpret_cast<const uint2*>(P.nbr + (size_t)b * P.sNbr + (size_t)min(gn, N - 1) * 16 + 4 * part);
        else {
#pragma unroll
            for (int q = 0; q < 4; ++q) wreg[q] = (q < wpl) ? brow[q] : 0ull;
        }
    }
    // every request is issued before the first LDS store of the copies (see spl_gather_kernel): the weight image, the input
    // rows and weights of the last stage (every row of the episode), the finished taps of the own columns (tap 0 = x_t itself)
    const int ntap = (K == 1) ? 1 : K - 1;
    float4 rimg = make_float4(0.f, 0.f, 0.f, 0.f), rs[2] = {rimg, rimg};
    float rw = 0.f, tv[2 * SP_MAXTAPS];
    if (tid < wt4 / 4) rimg = reinterpret_cast<const float4*>(P.image)[tid];
    if (K >= 2) {
        const float4* s4 = reinterpret_cast<const float4*>(P.tap[K - 1] + (size_t)b * P.ts[K - 1]);
#pragma unroll
        for (int q = 0; q < 2; ++q) if (tid + q * SPL_THREADS < 2 * N) rs[q] = s4[tid + q * SPL_THREADS];
        if (tid < N) rw = P.wq[(size_t)b * P.sWb + tid];
    }
#pragma unroll
    for (int q = 0; q < 2 * SP_MAXTAPS; ++q) {                    // element e = j * 2048 + c * 8 + f of the finished taps
        const int j = q >> 1, i = (tid + q * SPL_THREADS) & 2047, c = i >> 3, f = i & 7;
        tv[q] = (j < ntap && f < 6 && c0 + c < N) ? P.tap[j][(size_t)b * P.ts[j] + (size_t)(c0 + c) * 8 + f] : 0.f;
    }
    {
        float4* za = reinterpret_cast<float4*>(act);
        for (int i = tid; i < SPL_COLS * RO_CS / 4; i += SPL_THREADS) za[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        float4* dst = reinterpret_cast<float4*>(wl);
        if (tid < wt4 / 4) dst[tid] = rimg;
        for (int i = tid + SPL_THREADS; i < wt4 / 4; i += SPL_THREADS) dst[i] = reinterpret_cast<const float4*>(P.image)[i];
        if (K >= 2) {
            const float4* s4 = reinterpret_cast<const float4*>(P.tap[K - 1] + (size_t)b * P.ts[K - 1]);
            float4* d4 = reinterpret_cast<float4*>(ls);
#pragma unroll
            for (int q = 0; q < 2; ++q) if (tid + q * SPL_THREADS < 2 * N) d4[tid + q * SPL_THREADS] = rs[q];
            for (int i = tid + 2 * SPL_THREADS; i < 2 * N; i += SPL_THREADS) d4[i] = s4[i];
            if (tid < N) lw[tid] = rw;
            for (int i = tid + SPL_THREADS; i < N; i += SPL_THREADS) lw[i] = P.wq[(size_t)b * P.sWb + i];
        }
    }
    SP_STAMP(1, 1);
    __syncthreads();                                            // act is zero, the copies are complete
    SP_STAMP(1, 2);
    // channel (f, j) of column c -> MFMA B-fragment slot rpos(f K + j)
#pragma unroll
    for (int q = 0; q < 2 * SP_MAXTAPS; ++q) {
        const int j = q >> 1, i = (tid + q * SPL_THREADS) & 2047, c = i >> 3, f = i & 7;
        if (j < ntap && f < 6 && c0 + c < N) act[c * RO_CS + rpos(f * K + j)] = tv[q];
    }
    bool expert_drives = false;
    if (CL) {
        const double bq = floor((double)C.beta[b] * 4294967296.0);            // P(expert drives) in units of 2^-32
        const unsigned long long thr = bq <= 0.0 ? 0ull : (bq >= 4294967296.0 ? 4294967296ull : (unsigned long long)bq);
        expert_drives = (unsigned long long)dagger_coin(C.seed, C.episode[b], (unsigned int)C.age_now) < thr;
        const int cols = min(SPL_COLS, N - c0);
        const float* xt = P.tap[0] + (size_t)b * P.ts[0];
        float* ff = C.feat + (size_t)b * 6 * N;
        for (int i = tid; i < 6 * SPL_COLS; i += SPL_THREADS) {
            const int f = i >> 8, c = i & 255;
            if (c < cols) ff[(size_t)f * N + c0 + c] = xt[(size_t)(c0 + c) * 8 + f];
        }
        const float* ex = C.expert + (size_t)b * N * 2;
        float* lb = C.label + (size_t)b * 2 * N;
        for (int i = tid; i < 2 * SPL_COLS; i += SPL_THREADS) {
            const int a = i >> 8, c = i & 255;
            if (c < cols) lb[(size_t)a * N + c0 + c] = ex[(size_t)(c0 + c) * 2 + a];
        }
        const unsigned long long* nr = C.net + (size_t)b * C.sNb + (size_t)c0 * NW;
        unsigned long long* fb = C.bits + ((size_t)b * N + c0) * NW;
        for (int i = tid; i < cols * NW; i += SPL_THREADS) fb[i] = nr[i];
        if (tid < cols) C.wrow[(size_t)b * N + c0 + tid] = C.wnet[(size_t)b * C.sWn + c0 + tid];
        if (blockIdx.x == 0 && tid == 0) C.age[b] = C.age_now;
    }
    SP_STAMP(1, 3);
    if (K >= 2) {                                              // last tap: its last factor is applied here
        float sa[1][6];
        if (LS) spl_gather_list<1>(lst, part, brow, wpl, lw, ls, N, gn < N, sa);
        else spl_gather_column<1>(wreg, brow, wpl, part * wpl, lw, ls, N, gn < N, sa);
        if (part == 0 && gn < N) {
#pragma unroll
            for (int f = 0; f < 6; ++f) act[gc * RO_CS + rpos(f * K + K - 1)] = sa[0][f];
        }
    }
    SP_STAMP(1, 4);
    __syncthreads();
    SP_STAMP(1, 5);
    if (c0 + wave * 16 >= N) return;                           // whole waves; no workgroup barrier below
    // filter GEMM + tanh hidden layers: wave w owns columns 16 w .. 16 w + 15 (rollout.hip phase B)
    const int li = lane & 15, lq = lane >> 4;
    float* pcol = act + (wave * 16 + li) * RO_CS;
    for (int l = 0; l < n_layers - 1; ++l) {
        const int cin = (l == 0) ? FK : ((l < 8) ? (int)((dimsA >> (8 * l)) & 255ull) : (int)dims8);
        const int cout = (l + 1 < 8) ? (int)((dimsA >> (8 * (l + 1))) & 255ull) : (int)dims8;
        const int MT = mtiles(cout);
        const float* wfrag = wl + (int)((((l < 4) ? woffA : woffB) >> (16 * (l & 3))) & 0xFFFFull);
        if (MT == 2) ro_mlp_cols<2>(pcol, wfrag + lane * RO_WFS, wfrag + 2 * 64 * RO_WFS + lq * 4, pad4(cin) / 4, lq);
        else ro_mlp_cols<1>(pcol, wfrag + lane * RO_WFS, wfrag + 64 * RO_WFS + lq * 4, pad4(cin) / 4, lq);
    }
    SP_STAMP(1, 6);
    // 2-wide output layer: lane L takes column L >> 2 of the wave's tile and 8 channels, quad sum by DPP (phase C)
    const int lo_ = n_layers - 1;
    const float* w2 = wl + (int)((((lo_ < 4) ? woffA : woffB) >> (16 * (lo_ & 3))) & 0xFFFFull);
    const int ccol = wave * 16 + (lane >> 2), cg = lane & 3;
    const float* zsrc = act + ccol * RO_CS + cg * RO_KS;
    const float4 z0 = *reinterpret_cast<const float4*>(zsrc);
    const float4 z1 = *reinterpret_cast<const float4*>(zsrc + 4);
    const float zc[8] = {z0.x, z0.y, z0.z, z0.w, z1.x, z1.y, z1.z, z1.w};
    f32x2 u2 = {0.f, 0.f}, u2b = {0.f, 0.f};
#pragma unroll
    for (int s_ = 0; s_ < RO_KS; s_ += 2) {
        const float2 wa = *reinterpret_cast<const float2*>(w2 + 2 * (4 * s_ + cg));
        const float2 wb = *reinterpret_cast<const float2*>(w2 + 2 * (4 * (s_ + 1) + cg));
        u2 = __builtin_elementwise_fma((f32x2){zc[s_], zc[s_]}, (f32x2){wa.x, wa.y}, u2);
        u2b = __builtin_elementwise_fma((f32x2){zc[s_ + 1], zc[s_ + 1]}, (f32x2){wb.x, wb.y}, u2b);
    }
    u2 = u2 + u2b;
    float ux = u2.x, uy = u2.y;
    ux += dpp_f<0xB1>(ux); uy += dpp_f<0xB1>(uy);
    ux += dpp_f<0x4E>(ux); uy += dpp_f<0x4E>(uy);
    if (cg == 0 && c0 + ccol < N) {
        const float2 bb = *reinterpret_cast<const float2*>(w2 + 2 * 4 * RO_KS);
        float ax = ux + bb.x, ay = uy + bb.y;
        if (CL && expert_drives) {                             // gnn_dagger.py:157-161: the stored label drives the step
            const float2 e2 = *reinterpret_cast<const float2*>(C.expert + ((size_t)b * N + c0 + ccol) * 2);
            ax = e2.x; ay = e2.y;
        }
        action[((size_t)b * 2 + 0) * N + c0 + ccol] = ax;
        action[((size_t)b * 2 + 1) * N + c0 + ccol] = ay;
    }
    SP_STAMP(1, 7);
}

template <typename F>
int spl_allow_lds(F* fn, size_t lds)
{
    return mgp_allow_dyn_lds(reinterpret_cast<const void*>(fn), lds) == hipSuccess ? MGP_OK : MGP_ELAUNCH;
}

struct SpWeights {
    const float* W[MGP_MAX_LAYERS];
    const float* b[MGP_MAX_LAYERS];
    int dims[MGP_MAX_LAYERS + 1];
    int woff[MGP_MAX_LAYERS];
    int n_layers;
};

__global__ __launch_bounds__(SP_THREADS)
void sp_weight_image_kernel(SpWeights P, int K, float* __restrict__ image)
{
    for (int l = 0; l < P.n_layers; ++l) {
        const int cin = (l == 0) ? 6 * K : P.dims[l], cout = P.dims[l + 1];
        const bool last = l == P.n_layers - 1;
        const int tot = ro_weight_image_size(cout, last);
        for (int e = threadIdx.x; e < tot; e += SP_THREADS)
            image[P.woff[l] + e] = ro_weight_image_elem(P.W[l], P.b[l], cin, cout, last, e);
    }
}

// Dense slices of the delayed operator from the bit rows: row i of G_j = e_i A_T A_{T-1} ... A_{T-j+1}, one wave per row,
// row vectors ping-pong in LDS.  hs = ring slot of the newest network; networks that do not exist yet (fewer than K-1
// steps since the reset) are all-zero rows with zero weights, which makes the products vanish as the reference's do.
// grid: x = group of 4 rows, y = b.  LDS: [4 waves][2][Np]
__global__ __launch_bounds__(SP_THREADS)
void sp_to_dense_kernel(const unsigned long long* __restrict__ bits, const float* __restrict__ wq, float* __restrict__ G,
                        int K, int H, int N, int NW, int hs)
{
    extern __shared__ __attribute__((aligned(16))) float spm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.y;
    const int i = blockIdx.x * 4 + wave;
    const int Np = (N + 3) & ~3;
    if (i >= N) return;                                       // whole wave (no workgroup barrier below)
    float* r0 = spm + (size_t)wave * 2 * Np;
    float* r1 = r0 + Np;
    const unsigned long long* bb = bits + (size_t)b * H * N * NW;
    const float* wb = wq + (size_t)b * H * N;
    float* Gb = G + (size_t)b * K * N * N;
    {   // e_i . A_T
        const unsigned long long* row = bb + ((size_t)hs * N + i) * NW;
        const float wi = wb[(size_t)hs * N + i];
        for (int n = lane; n < N; n += 64) {
            const float v = ((row[n >> 6] >> (n & 63)) & 1ull) ? wi : 0.f;
            r0[n] = v;
            Gb[((size_t)1 * N + i) * N + n] = v;
        }
    }
    for (int j = 2; j < K; ++j) {                             // r1 = r0 . A_{T-j+1}: scatter along the rows m with r0[m] != 0
        int hq = hs - (j - 1); hq = hq < 0 ? hq + H : hq;
        const unsigned long long* net = bb + (size_t)hq * N * NW;
        const float* wn = wb + (size_t)hq * N;
        for (int n = lane; n < N; n += 64) r1[n] = 0.f;
        for (int m0 = 0; m0 < N; m0 += 64) {
            const float rv = (m0 + lane < N) ? r0[m0 + lane] : 0.f;
            unsigned long long nz = __ballot(rv != 0.f);
            while (nz) {                                      // wave-uniform loop over the non-zero entries of this chunk
                const int m = m0 + __builtin_ctzll(nz);
                nz &= nz - 1ull;
                const float val = r0[m] * wn[m];
                // row m of the (symmetric) pattern: lane l walks words l, l + 64, ...: distinct columns, no write conflicts
                for (int wd = lane; wd < NW; wd += 64) {
                    unsigned long long w = net[(size_t)m * NW + wd];
                    while (w) { const int n = 64 * wd + __builtin_ctzll(w); w &= w - 1ull; r1[n] += val; }
                }
            }
        }
        for (int n = lane; n < N; n += 64) Gb[((size_t)j * N + i) * N + n] = r1[n];
        float* t = r0; r0 = r1; r1 = t;
    }
}

int sp_mode = 0;                          // mgp_sparse_force_direct: 0 default, 1 direct kernels, 2 staged kernels on bit rows only

}  // namespace

extern "C" int mgp_sparse_policy_supported(const int* dims, int n_layers, int K, int N)
{
    int woff[MGP_MAX_LAYERS], wtot = 0;
    if (N < 1 || N > 4096) return 0;
    return sp_plan(dims, n_layers, K, woff, &wtot) == MGP_OK ? 1 : 0;
}

extern "C" long mgp_sparse_policy_image_floats(const int* dims, int n_layers, int K)
{
    int woff[MGP_MAX_LAYERS], wtot = 0;
    return sp_plan(dims, n_layers, K, woff, &wtot) == MGP_OK ? wtot : 0;
}

extern "C" int mgp_sparse_policy_image(const float* const* W, const float* const* b, const int* dims, int n_layers, int K,
                                       float* image, void* stream)
{
    if (W == nullptr || b == nullptr) return MGP_EINVAL;
    SpWeights P;
    int wtot = 0;
    int rc = sp_plan(dims, n_layers, K, P.woff, &wtot);
    if (rc != MGP_OK) return rc;
    MGP_CHECK_PTR(image);
    if (!mgp_aligned16(image)) return MGP_EALIGN;
    P.n_layers = n_layers;
    for (int l = 0; l <= n_layers; ++l) P.dims[l] = dims[l];
    for (int l = 0; l < n_layers; ++l) { MGP_CHECK_PTR(W[l]); MGP_CHECK_PTR(b[l]); P.W[l] = W[l]; P.b[l] = b[l]; }
    mgp_clear_error();
    hipLaunchKernelGGL(sp_weight_image_kernel, dim3(1), dim3(SP_THREADS), 0, static_cast<hipStream_t>(stream), P, K, image);
    return mgp_launch_status();
}

/* One policy evaluation on the factored state: action (B,1,2,N) <- Actor(x_t .. x_{t-K+1}; A_t .. A_{t-K+2}).
 * cur = ring slot of x_t in feat (B,K,N,8); hs = ring slot of A_t in bits (B,H,N,NW) / wrow (B,H,N), H = max(K-1, 1).
 * scratch: 2 * (K-1) * B * N * 8 floats for K >= 3 (running products between stages), else unused. */
static int sp_policy_step(const unsigned long long* bits, const float* wrow, const float* feat,
                          const float* image, const int* dims, int n_layers, float* scratch, float* action,
                          int B, int K, int N, int cur, int hs, const MgpSparseCollect* col, const unsigned short* nbr,
                          void* stream)
{
    if (sp_mode != 0) nbr = nullptr;
    const long sN = (long)(K > 2 ? K - 1 : 1) * N * 16;
    int woff[MGP_MAX_LAYERS], wtot = 0;
    int rc = sp_plan(dims, n_layers, K, woff, &wtot);
    if (rc != MGP_OK) return rc;
    if (B <= 0 || N <= 0 || N > 4096 || B > 65535) return MGP_EINVAL;
    MGP_CHECK_PTR8(bits); MGP_CHECK_PTR(wrow); MGP_CHECK_PTR(feat); MGP_CHECK_PTR(image); MGP_CHECK_PTR(action);
    if (!mgp_aligned16(feat) || !mgp_aligned16(image)) return MGP_EALIGN;
    if (K >= 3) { MGP_CHECK_PTR(scratch); if (!mgp_aligned16(scratch)) return MGP_EALIGN; }
    const int H = K > 2 ? K - 1 : 1;
    const int NW = mgp_sparse_words(N);
    if (cur < 0 || cur >= K || hs < 0 || hs >= H) return MGP_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long sB = (long)H * N * NW, sW = (long)H * N, sF = (long)K * N * 8, sV = (long)N * 8;
    auto fslot = [&](int j) { int s = cur - j; if (s < 0) s += K; return feat + (size_t)s * N * 8; };
    auto hslot = [&](int q) { int s = hs - (q - 1); if (s < 0) s += H; return s; };
    // running products of taps j >= 1 between stages: V[pp][j-1], each (B, N, 8)
    auto vbuf = [&](int pp, int j) { return scratch + ((size_t)pp * (K - 1) + (j - 1)) * (size_t)B * N * 8; };
    const int ntiles = mgp_ceil_div(N, SP_COLS);
    const int Np = (N + 3) & ~3;
    mgp_clear_error();
    SpPolicy P = {};
    P.tap[0] = fslot(0); P.ts[0] = sF;
    for (int q = 1; q <= K - 2; ++q) {                         // stage q: taps q .. K-1 times A_{t-q+1}
        SpTaps T = {};
        int nt = 0;
        for (int j = q; j <= K - 1; ++j, ++nt) {
            if (q == 1) { T.src[nt] = fslot(j); T.ss[nt] = sF; }
            else { T.src[nt] = vbuf((q - 1) & 1, j); T.ss[nt] = sV; }
            T.dst[nt] = vbuf(q & 1, j); T.ds[nt] = sV;         // tap q's product is finished here: the tail reads it from there
        }
        const int s = hslot(q);
        const size_t glds = ((size_t)Np + (size_t)nt * N * 8) * sizeof(float);
        if (sp_mode != 1 && glds <= SPL_LDS_MAX) {                  // staged form: source rows of all taps of the stage in LDS
            const dim3 gg(mgp_ceil_div(N, SPL_COLS), B), gb(SPL_THREADS);
            const unsigned long long* bq = bits + (size_t)s * N * NW;
            const float* wq_ = wrow + (size_t)s * N;
            const unsigned short* nq = nbr != nullptr ? nbr + (size_t)s * N * 16 : nullptr;
#define SPL_GO(NT_)                                                                                                              \
            if (nq != nullptr) { rc = spl_allow_lds(spl_gather_kernel<NT_, true>, glds); if (rc) return rc;                       \
                hipLaunchKernelGGL((spl_gather_kernel<NT_, true>), gg, gb, glds, st, bq, sB, wq_, sW, T, N, NW, nq, sN); }        \
            else { rc = spl_allow_lds(spl_gather_kernel<NT_, false>, glds); if (rc) return rc;                                    \
                hipLaunchKernelGGL((spl_gather_kernel<NT_, false>), gg, gb, glds, st, bq, sB, wq_, sW, T, N, NW, nq, sN); }
            switch (nt) {
            case 1: SPL_GO(1) break;
            case 2: SPL_GO(2) break;
            case 3: SPL_GO(3) break;
            default: SPL_GO(4) break;
            }
#undef SPL_GO
        } else {
            hipLaunchKernelGGL(sp_gather_kernel, dim3(ntiles, nt, B), dim3(SP_THREADS), 0, st, bits + (size_t)s * N * NW, sB,
                               wrow + (size_t)s * N, sW, T, N, NW);
        }
        rc = mgp_launch_status();
        if (rc != MGP_OK) return rc;
        P.tap[q] = vbuf(q & 1, q); P.ts[q] = sV;
    }
    // NOTE on buffer reuse: stage q writes V[q & 1][j] for every j >= q and reads V[(q-1) & 1][j]; a finished tap q sits in
    // V[q & 1][q], which later stages q' > q never write (they only touch taps j >= q').
    if (K >= 2) {
        if (K == 2) { P.tap[1] = fslot(1); P.ts[1] = sF; }
        else { P.tap[K - 1] = vbuf((K - 2) & 1, K - 1); P.ts[K - 1] = sV; }
        const int s = hslot(K - 1);
        P.bits = bits + (size_t)s * N * NW; P.sBb = sB;
        P.wq = wrow + (size_t)s * N; P.sWb = sW;
        P.nbr = nbr != nullptr ? nbr + (size_t)s * N * 16 : nullptr; P.sNbr = sN;
    }
    P.image = image; P.wtot = wtot;
    unsigned long long dimsA = 0ull, woffA = 0ull, woffB = 0ull;
    unsigned int dims8 = 0u;
    for (int l = 0; l <= n_layers; ++l) {
        if (l < 8) dimsA |= (unsigned long long)(dims[l] & 255) << (8 * l);
        else dims8 = (unsigned int)dims[l];
    }
    for (int l = 0; l < n_layers; ++l) {
        if (woff[l] > 0xFFFF) return MGP_EUNSUPPORTED;
        if (l < 4) woffA |= (unsigned long long)woff[l] << (16 * l);
        else woffB |= (unsigned long long)woff[l] << (16 * (l - 4));
    }
    const size_t lds = ((size_t)SP_COLS * RO_CS + wtot) * sizeof(float);
    const size_t plds = ((size_t)SPL_COLS * RO_CS + ((wtot + 3) & ~3) + (K >= 2 ? (size_t)Np + (size_t)N * 8 : 0)) * sizeof(float);
    const bool staged = sp_mode != 1 && plds <= SPL_LDS_MAX;
    const bool lists = staged && K >= 2 && P.nbr != nullptr;
    const dim3 pg(mgp_ceil_div(N, SPL_COLS), B);
    SpCollect C = {};
    if (col != nullptr) {
        const size_t fr = (size_t)col->ring_step * B;          // first frame of this ring step
        C.feat = col->feat + fr * 6 * N; C.bits = col->bits + fr * N * NW; C.wrow = col->wrow + fr * N;
        C.label = col->label + fr * 2 * N; C.age = col->age + fr;
        C.net = bits + (size_t)hs * N * NW; C.sNb = sB; C.wnet = wrow + (size_t)hs * N; C.sWn = sW;
        C.expert = col->expert; C.beta = col->beta; C.episode = col->episode; C.seed = col->seed; C.age_now = col->age_now;
        if (staged && lists) {
            rc = spl_allow_lds(spl_policy_kernel<true, true>, plds); if (rc) return rc;
            hipLaunchKernelGGL((spl_policy_kernel<true, true>), pg, dim3(SPL_THREADS), plds, st, P, action, K, N, NW, dimsA, dims8,
                               woffA, woffB, n_layers, C);
        } else if (staged) {
            rc = spl_allow_lds(spl_policy_kernel<true, false>, plds); if (rc) return rc;
            hipLaunchKernelGGL((spl_policy_kernel<true, false>), pg, dim3(SPL_THREADS), plds, st, P, action, K, N, NW, dimsA, dims8,
                               woffA, woffB, n_layers, C);
        } else {
            hipLaunchKernelGGL(sp_policy_kernel<true>, dim3(ntiles, B), dim3(SP_THREADS), lds, st, P, action, K, N, NW, dimsA,
                               dims8, woffA, woffB, n_layers, C);
        }
    } else if (staged && lists) {
        rc = spl_allow_lds(spl_policy_kernel<false, true>, plds); if (rc) return rc;
        hipLaunchKernelGGL((spl_policy_kernel<false, true>), pg, dim3(SPL_THREADS), plds, st, P, action, K, N, NW, dimsA, dims8,
                           woffA, woffB, n_layers, C);
    } else if (staged) {
        rc = spl_allow_lds(spl_policy_kernel<false, false>, plds); if (rc) return rc;
        hipLaunchKernelGGL((spl_policy_kernel<false, false>), pg, dim3(SPL_THREADS), plds, st, P, action, K, N, NW, dimsA, dims8,
                           woffA, woffB, n_layers, C);
    } else {
        hipLaunchKernelGGL(sp_policy_kernel<false>, dim3(ntiles, B), dim3(SP_THREADS), lds, st, P, action, K, N, NW, dimsA,
                           dims8, woffA, woffB, n_layers, C);
    }
    return mgp_launch_status();
}

/* Test hook: 1 = keep the direct (global-memory) gather / policy kernels even where the staged forms fit; returns the old value. */
extern "C" int mgp_sparse_force_direct(int mode)
{
    const int old = sp_mode;
    sp_mode = (mode == 1 || mode == 2) ? mode : 0;
    return old;
}

extern "C" int mgp_sparse_policy_step(const unsigned long long* bits, const float* wrow, const float* feat,
                                      const float* image, const int* dims, int n_layers, float* scratch, float* action,
                                      int B, int K, int N, int cur, int hs, void* stream)
{
    return sp_policy_step(bits, wrow, feat, image, dims, n_layers, scratch, action, B, K, N, cur, hs, nullptr, nullptr, stream);
}

/* The same evaluation as one collected DAGGER step (see SpCollect): files the frame of the current state at ring step
 * col->ring_step and writes into `action` what drives the step -- the expert's action where the lane's coin says so. */
extern "C" int mgp_sparse_policy_collect(const unsigned long long* bits, const float* wrow, const float* feat,
                                         const float* image, const int* dims, int n_layers, float* scratch, float* action,
                                         int B, int K, int N, int cur, int hs, const MgpSparseCollect* col, void* stream)
{
    if (col == nullptr) return MGP_EINVAL;
    MGP_CHECK_PTR(col->feat); MGP_CHECK_PTR8(col->bits); MGP_CHECK_PTR(col->wrow); MGP_CHECK_PTR(col->label);
    MGP_CHECK_PTR(col->age); MGP_CHECK_PTR(col->expert); MGP_CHECK_PTR(col->beta); MGP_CHECK_PTR(col->episode);
    if (reinterpret_cast<uintptr_t>(col->expert) & 7u) return MGP_EALIGN;
    if (col->ring_steps < 1 || col->ring_step < 0 || col->ring_step >= col->ring_steps || col->age_now < 0) return MGP_EINVAL;
    return sp_policy_step(bits, wrow, feat, image, dims, n_layers, scratch, action, B, K, N, cur, hs, col, nullptr, stream);
}

/* T closed-loop steps on the factored state, enqueued from one call (the loop of learner/sparse_rollout.py without a host
 * round trip per launch): per step mgp_sparse_policy_step / _collect, then the simulator (cell list up to N = 2048, all
 * pairs beyond) into the next ring slots; x_a holds the state on entry, x_a / x_b ping-pong.  rewards (T,B) or NULL;
 * expert (B,N,2) or NULL (required with collect).  collect: ring_step / age_now of the FIRST step, advanced per step (ring
 * step modulo ring_steps).  On return *cur / *hs are the ring slots of the final state and the state is in (T odd ? x_b : x_a). */
extern "C" int mgp_sparse_rollout(unsigned long long* bits, float* wrow, float* feat, const float* image, const int* dims,
                                  int n_layers, float* scratch, float* action, double* x_a, double* x_b, double* rewards,
                                  float* expert, const MgpFlockParams* p, int B, int K, int N, int T, int* cur, int* hs,
                                  const MgpSparseCollect* collect, unsigned short* nbr, void* stream)
{
    if (cur == nullptr || hs == nullptr || T < 0 || p == nullptr) return MGP_EINVAL;
    if (N > 2048) nbr = nullptr;                               // (the all-pairs simulator does not write lists)
    if (collect != nullptr && expert == nullptr) return MGP_EINVAL;
    const int H = K > 2 ? K - 1 : 1;
    const int NW = mgp_sparse_words(N);
    int c = *cur, h = *hs;
    if (T > 0) {                                                // one launch of persistent workgroups where the shape is covered
        if (collect != nullptr) {
            MGP_CHECK_PTR(collect->feat); MGP_CHECK_PTR8(collect->bits); MGP_CHECK_PTR(collect->wrow); MGP_CHECK_PTR(collect->label);
            MGP_CHECK_PTR(collect->age); MGP_CHECK_PTR(collect->expert); MGP_CHECK_PTR(collect->beta); MGP_CHECK_PTR(collect->episode);
            if (collect->ring_steps < 1 || collect->ring_step < 0 || collect->ring_step >= collect->ring_steps || collect->age_now < 0)
                return MGP_EINVAL;
        }
        const int rc = spp_rollout(bits, wrow, feat, image, dims, n_layers, scratch, action, x_a, x_b, rewards, expert, p, B, K, N,
                                   T, c, h, sp_mode != 0 ? nullptr : nbr, collect, static_cast<hipStream_t>(stream));
        if (rc == MGP_OK) { *cur = (c + T) % K; *hs = (h + T) % H; return MGP_OK; }
        if (rc != MGP_EUNSUPPORTED) return rc;
    }
    MgpSparseCollect col = {};
    if (collect != nullptr) col = *collect;
    double* xs[2] = {x_a, x_b};
    for (int t = 0; t < T; ++t) {
        if (collect != nullptr) {
            MGP_CHECK_PTR(col.feat); MGP_CHECK_PTR8(col.bits); MGP_CHECK_PTR(col.wrow); MGP_CHECK_PTR(col.label);
            MGP_CHECK_PTR(col.age); MGP_CHECK_PTR(col.expert); MGP_CHECK_PTR(col.beta); MGP_CHECK_PTR(col.episode);
            if (col.ring_steps < 1 || col.ring_step < 0 || col.ring_step >= col.ring_steps || col.age_now < 0) return MGP_EINVAL;
        }
        int rc = sp_policy_step(bits, wrow, feat, image, dims, n_layers, scratch, action, B, K, N, c, h,
                                collect != nullptr ? &col : nullptr, nbr, stream);
        if (rc != MGP_OK) return rc;
        const int nh = (h + 1) % H, nc = (c + 1) % K;
        double* rw = rewards != nullptr ? rewards + (size_t)t * B : nullptr;
        unsigned long long* bq = bits + (size_t)nh * N * NW;
        float* wq = wrow + (size_t)nh * N;
        float* fq = feat + (size_t)nc * N * 8;
        // the Actor's output layout (B,1,2,N): agent stride 1, axis stride N
        rc = (N <= 2048)
            ? mgp_flock_step_cells_nbr(xs[t & 1], xs[(t & 1) ^ 1], action, 1, N, bq, (long)H * N * NW, wq, (long)H * N, fq,
                                       (long)K * N * 8, nbr != nullptr ? nbr + (size_t)nh * N * 16 : nullptr, (long)H * N * 16,
                                       rw, expert, p, B, N, stream)
            : mgp_flock_step_sparse(xs[t & 1], xs[(t & 1) ^ 1], action, 1, N, bq, (long)H * N * NW, wq, (long)H * N, fq,
                                    (long)K * N * 8, rw, expert, p, B, N, stream);
        if (rc != MGP_OK) return rc;
        h = nh; c = nc;
        if (collect != nullptr) { col.ring_step = (col.ring_step + 1) % col.ring_steps; col.age_now += 1; }
    }
    *cur = c; *hs = h;
    return MGP_OK;
}

/* Dense delayed operator of the reference contract from the factored state: G (B,K,N,N) slices 1..K-1 (slice 0, the
 * identity, is left alone).  hs = ring slot of the newest network. */
extern "C" int mgp_sparse_to_dense(const unsigned long long* bits, const float* wrow, float* G, int B, int K, int N, int hs,
                                   void* stream)
{
    if (B <= 0 || N <= 0 || K < 1 || K > SP_MAXTAPS + 1 || N > 4096 || B > 65535) return MGP_EINVAL;
    if (K == 1) return MGP_OK;
    MGP_CHECK_PTR8(bits); MGP_CHECK_PTR(wrow); MGP_CHECK_PTR(G);
    const int H = K > 2 ? K - 1 : 1;
    if (hs < 0 || hs >= H) return MGP_EINVAL;
    const int NW = mgp_sparse_words(N);
    const size_t lds = (size_t)4 * 2 * ((N + 3) & ~3) * sizeof(float);
    mgp_clear_error();
    if (mgp_allow_dyn_lds(reinterpret_cast<const void*>(sp_to_dense_kernel), lds) != hipSuccess) return MGP_ELAUNCH;
    hipLaunchKernelGGL(sp_to_dense_kernel, dim3(mgp_ceil_div(N, 4), B), dim3(SP_THREADS), lds, static_cast<hipStream_t>(stream),
                       bits, wrow, G, K, H, N, NW, hs);
    return mgp_launch_status();
}
