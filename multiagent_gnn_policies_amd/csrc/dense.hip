// Per-agent dense layers = the reference's Conv2d((step,1), stride (step,1)) + tanh
// (reference actor.py:37-38,73-77):   out[b,o,t,n] = act(bias[o] + sum_c W[o,c] * in[b,c,t,n]).
// Generic (any Cin/Cout/T/N) VALU kernels used by the composed Actor path and by every shape the
// fused kernel (actor_fused.hip) does not cover.  A workgroup owns a tile of 64 agent columns; one wave
// per output-channel group so the weight reads are wave-uniform LDS broadcasts.
#include "mgp_common.h"

namespace {

constexpr int DN_THREADS = 256;
constexpr int DN_COLS = 64;       // agent columns per workgroup
constexpr int DN_CC = 32;         // input channels staged per pass

__device__ __forceinline__ float act_fwd(float z, int act) { return act == MGP_ACT_TANH ? tanhf(z) : z; }

// grid: x = n tile, y = t, z = b ; outputs o = o_base + og + 4*i owned by wave og (og = tid/64)
template <int OT>
__global__ __launch_bounds__(DN_THREADS)
void dense_fwd_kernel(const float* __restrict__ in, const float* __restrict__ W, const float* __restrict__ bias,
                      float* __restrict__ out, int Cin, int Cout, int T, int N, int o_base,
                      long sib, long sic, long sit, int act)
{
    __shared__ float ins[DN_CC][DN_COLS];
    __shared__ __attribute__((aligned(16))) float ws[4 * OT][DN_CC];
    const int tid = threadIdx.x, col = tid & 63, og = tid >> 6;
    const int n0 = blockIdx.x * DN_COLS, t = blockIdx.y, b = blockIdx.z;
    const int n = n0 + col;
    const float* inb = in + b * sib + t * sit;

    float acc[OT];
#pragma unroll
    for (int i = 0; i < OT; ++i) acc[i] = 0.f;

    for (int c0 = 0; c0 < Cin; c0 += DN_CC) {
        __syncthreads();
        for (int i = tid; i < DN_CC * DN_COLS; i += DN_THREADS) {
            const int cc = i >> 6, cl = i & 63;
            ins[cc][cl] = (c0 + cc < Cin && n0 + cl < N) ? inb[(c0 + cc) * sic + n0 + cl] : 0.f;
        }
        for (int i = tid; i < 4 * OT * DN_CC; i += DN_THREADS) {
            const int oo = i / DN_CC, cc = i - oo * DN_CC;
            const int o = o_base + oo;
            ws[oo][cc] = (o < Cout && c0 + cc < Cin) ? W[(size_t)o * Cin + c0 + cc] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int cc = 0; cc < DN_CC; cc += 4) {
            const float x0 = ins[cc][col], x1 = ins[cc + 1][col], x2 = ins[cc + 2][col], x3 = ins[cc + 3][col];
#pragma unroll
            for (int i = 0; i < OT; ++i) {
                const float4 w = *reinterpret_cast<const float4*>(&ws[og + 4 * i][cc]);
                acc[i] = fmaf(w.x, x0, acc[i]);
                acc[i] = fmaf(w.y, x1, acc[i]);
                acc[i] = fmaf(w.z, x2, acc[i]);
                acc[i] = fmaf(w.w, x3, acc[i]);
            }
        }
    }
    if (n < N) {
#pragma unroll
        for (int i = 0; i < OT; ++i) {
            const int o = o_base + og + 4 * i;
            if (o < Cout) {
                const float z = acc[i] + bias[o];
                out[(((size_t)b * Cout + o) * T + t) * N + n] = act_fwd(z, act);
            }
        }
    }
}

// ---- backward, stage 1: per column tile: delta, partial dW / db, dIn --------------------------------
// LDS: delta[Cout][65], ins[DN_CC][65].  Partials go to part[tile][Cout*Cin + Cout].
__global__ __launch_bounds__(DN_THREADS)
void dense_bwd_kernel(const float* __restrict__ dOut, const float* __restrict__ outp, const float* __restrict__ in,
                      const float* __restrict__ W, float* __restrict__ part, float* __restrict__ dIn,
                      int Cin, int Cout, int T, int N, long sib, long sic, long sit, int act)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* delta = smem;                               // [Cout][65]
    float* ins = smem + (size_t)Cout * 65;             // [DN_CC][65]
    const int tid = threadIdx.x;
    const int n0 = blockIdx.x * DN_COLS, t = blockIdx.y, b = blockIdx.z;
    const int ntx = gridDim.x;
    const size_t tile = ((size_t)b * T + t) * ntx + blockIdx.x;
    float* my = part + tile * ((size_t)Cout * Cin + Cout);

    for (int i = tid; i < Cout * DN_COLS; i += DN_THREADS) {
        const int o = i >> 6, cl = i & 63;
        float d = 0.f;
        if (n0 + cl < N) {
            const size_t idx = (((size_t)b * Cout + o) * T + t) * N + n0 + cl;
            d = dOut[idx];
            if (act == MGP_ACT_TANH) { const float z = outp[idx]; d *= (1.f - z * z); }
        }
        delta[o * 65 + cl] = d;
    }
    __syncthreads();
    // db partial
    for (int o = tid; o < Cout; o += DN_THREADS) {
        float s = 0.f;
        for (int cl = 0; cl < DN_COLS; ++cl) s += delta[o * 65 + cl];
        my[(size_t)Cout * Cin + o] = s;
    }
    const float* inb = in + b * sib + t * sit;
    for (int c0 = 0; c0 < Cin; c0 += DN_CC) {
        const int ccn = min(DN_CC, Cin - c0);
        __syncthreads();
        for (int i = tid; i < DN_CC * DN_COLS; i += DN_THREADS) {
            const int cc = i >> 6, cl = i & 63;
            ins[cc * 65 + cl] = (cc < ccn && n0 + cl < N) ? inb[(c0 + cc) * sic + n0 + cl] : 0.f;
        }
        __syncthreads();
        // dW partial: pairs (o, cc)
        for (int p = tid; p < Cout * ccn; p += DN_THREADS) {
            const int o = p / ccn, cc = p - o * ccn;
            float s = 0.f;
#pragma unroll 8
            for (int cl = 0; cl < DN_COLS; ++cl) s = fmaf(delta[o * 65 + cl], ins[cc * 65 + cl], s);
            my[(size_t)o * Cin + c0 + cc] = s;
        }
        // dIn: thread (col, channel group): dIn[c][col] = sum_o W[o][c] * delta[o][col]
        if (dIn != nullptr) {
            const int col = tid & 63, cgp = tid >> 6;
            if (n0 + col < N) {
                for (int cc = cgp; cc < ccn; cc += 4) {
                    float s = 0.f;
                    for (int o = 0; o < Cout; ++o) s = fmaf(W[(size_t)o * Cin + c0 + cc], delta[o * 65 + col], s);
                    dIn[(((size_t)b * Cin + c0 + cc) * T + t) * N + n0 + col] = s;
                }
            }
        }
    }
}

// stage 2: sum partials over tiles in tile order (deterministic)
__global__ __launch_bounds__(DN_THREADS)
void dense_bwd_reduce_kernel(const float* __restrict__ part, float* __restrict__ dW, float* __restrict__ db,
                             int Cin, int Cout, long ntiles)
{
    const long P = (long)Cout * Cin + Cout;
    const long i = (long)blockIdx.x * DN_THREADS + threadIdx.x;
    if (i >= P) return;
    float s = 0.f;
    for (long tl = 0; tl < ntiles; ++tl) s += part[tl * P + i];
    if (i < (long)Cout * Cin) dW[i] = s; else db[i - (long)Cout * Cin] = s;
}

}  // namespace

extern "C" int mgp_dense_fwd(const float* in, const float* W, const float* bias, float* out,
                             int B, int Cin, int Cout, int T, int N, long sib, long sic, long sit,
                             int act, void* stream)
{
    if (B < 0 || Cin <= 0 || Cout <= 0 || T <= 0 || N <= 0) return MGP_EINVAL;
    if (act != MGP_ACT_NONE && act != MGP_ACT_TANH) return MGP_EINVAL;
    if (B == 0) return MGP_OK;
    MGP_CHECK_PTR(in); MGP_CHECK_PTR(W); MGP_CHECK_PTR(bias); MGP_CHECK_PTR(out);
    if (B > 65535 || T > 65535) return MGP_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    mgp_clear_error();
    dim3 grid(mgp_ceil_div(N, DN_COLS), T, B);
    // output channels are processed in chunks of 4*OT
    for (int o_base = 0; o_base < Cout;) {
        const int rem = Cout - o_base;
#define MGP_DN_CASE(OT)                                                                                  \
        hipLaunchKernelGGL((dense_fwd_kernel<OT>), grid, dim3(DN_THREADS), 0, st, in, W, bias, out, Cin,  \
                           Cout, T, N, o_base, sib, sic, sit, act);                                      \
        o_base += 4 * OT
        if (rem <= 4) { MGP_DN_CASE(1); }
        else if (rem <= 8) { MGP_DN_CASE(2); }
        else if (rem <= 16) { MGP_DN_CASE(4); }
        else if (rem <= 32) { MGP_DN_CASE(8); }
        else { MGP_DN_CASE(16); }
#undef MGP_DN_CASE
        const int rc = mgp_launch_status();
        if (rc != MGP_OK) return rc;
    }
    return MGP_OK;
}

extern "C" long mgp_dense_bwd_workspace(int B, int Cin, int Cout, int T, int N)
{
    if (B <= 0 || Cin <= 0 || Cout <= 0 || T <= 0 || N <= 0) return 0;
    const long ntiles = (long)B * T * mgp_ceil_div(N, DN_COLS);
    return ntiles * ((long)Cout * Cin + Cout);
}

extern "C" int mgp_dense_bwd(const float* dOut, const float* out, const float* in, const float* W,
                             float* dW, float* db, float* dIn, int B, int Cin, int Cout, int T, int N,
                             long sib, long sic, long sit, int act, float* workspace, void* stream)
{
    if (B <= 0 || Cin <= 0 || Cout <= 0 || T <= 0 || N <= 0) return MGP_EINVAL;
    if (act != MGP_ACT_NONE && act != MGP_ACT_TANH) return MGP_EINVAL;
    MGP_CHECK_PTR(dOut); MGP_CHECK_PTR(in); MGP_CHECK_PTR(W); MGP_CHECK_PTR(dW); MGP_CHECK_PTR(db);
    MGP_CHECK_PTR(workspace);
    if (act == MGP_ACT_TANH) MGP_CHECK_PTR(out);
    if (dIn != nullptr && (reinterpret_cast<uintptr_t>(dIn) & 3u)) return MGP_EALIGN;
    if (B > 65535 || T > 65535) return MGP_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    mgp_clear_error();
    const int ntx = mgp_ceil_div(N, DN_COLS);
    dim3 grid(ntx, T, B);
    const size_t lds = ((size_t)Cout * 65 + (size_t)DN_CC * 65) * sizeof(float);
    if (lds > 150 * 1024) return MGP_EUNSUPPORTED;
    if (mgp_allow_dyn_lds(reinterpret_cast<const void*>(dense_bwd_kernel), lds) != hipSuccess) return MGP_ELAUNCH;
    hipLaunchKernelGGL(dense_bwd_kernel, grid, dim3(DN_THREADS), lds, st, dOut, out, in, W, workspace, dIn,
                       Cin, Cout, T, N, sib, sic, sit, act);
    int rc = mgp_launch_status();
    if (rc != MGP_OK) return rc;
    const long P = (long)Cout * Cin + Cout;
    const long ntiles = (long)B * T * ntx;
    hipLaunchKernelGGL(dense_bwd_reduce_kernel, dim3((unsigned)((P + DN_THREADS - 1) / DN_THREADS)),
                       dim3(DN_THREADS), 0, st, workspace, dW, db, Cin, Cout, ntiles);
    return mgp_launch_status();
}
