// What a single launch can stream on this device at the sizes of the dense-contract kernels: a read-only kernel with nothing
// but the loads (every thread requests U float4 up front -- nontemporal --, adds them, one float per workgroup is written),
// swept over workgroup shapes, for the operator bytes of cfg-2 (256 x 3 x 100 x 100 fp32 = 30.7 MB), its B = 2048 form and
// cfg-3 (768 MB).  The best row is the floor `roofline.frac` of agg_fwd / actor_fwd can be read against: launch ramp + first-byte
// latency + drain are part of any launch of that size.  Times: HIP events around 200 back-to-back launches over 11 buffers
// (no buffer is re-read while it could still sit in a cache: 11 x 30.7 MB > the 256 MB of the last-level cache).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -o scratch/stream_floor tools/harness/stream_floor.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int U>
__global__ void read_kernel(const f32x4* __restrict__ src, size_t n4, float* __restrict__ out)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    float acc = 0.f;
    while (i < n4) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(src + (i + u * stride < n4 ? i + u * stride : i));
#pragma unroll
        for (int u = 0; u < U; ++u) acc += (v[u].x + v[u].y) + (v[u].z + v[u].w);
        i += U * stride;
    }
    if (acc == 123.456f) out[blockIdx.x] = acc;              // (never true for the fill used: keeps the loads alive, no store traffic)
}

template <int U>
float run(const std::vector<f32x4*>& bufs, size_t n4, int wgs, int threads, float* out, int iters)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 20; ++it) hipLaunchKernelGGL(read_kernel<U>, dim3(wgs), dim3(threads), 0, nullptr, bufs[it % bufs.size()], n4, out);
    hipDeviceSynchronize();
    hipEventRecord(e0, nullptr);
    for (int it = 0; it < iters; ++it) hipLaunchKernelGGL(read_kernel<U>, dim3(wgs), dim3(threads), 0, nullptr, bufs[it % bufs.size()], n4, out);
    hipEventRecord(e1, nullptr); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return 1e3f * ms / iters;
}

// The request pattern of agg_fwd_mfma4_kernel<14, 2, 2, 2> (N = 100: workgroup = (episode, tap), wave = (column block, row half),
// lane (li, lq) requests the float4 G[row][4 g ..] of rows 16 (s >> 2) + 4 lq + (s & 3), all 14 up front) with the arithmetic
// removed: what the pattern alone costs against the flat stream above.  FLAT = true: the same workgroups, the same bytes, but thread t
// of the workgroup takes float4 t, t + 256, .. of the slice (2,500 float4: ten per thread, the last partly masked).
template <bool FLAT>
__global__ __launch_bounds__(256) void agg_pattern_kernel(const float* __restrict__ G, float* __restrict__ out, int N)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lq = lane >> 4;
    const float* Gk = G + (size_t)blockIdx.x * N * N;
    f32x4 gv[14];
    if (FLAT) {
        const f32x4* g4 = reinterpret_cast<const f32x4*>(Gk);
#pragma unroll
        for (int s = 0; s < 10; ++s) { const int q = threadIdx.x + 256 * s; gv[s] = __builtin_nontemporal_load(g4 + (q < 2500 ? q : 2499)); }
#pragma unroll
        for (int s = 10; s < 14; ++s) gv[s] = f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
        const int blk = wave >> 1, part = wave & 1;
        const int gtot = N >> 2, g0 = (gtot + 1) >> 1, ng = blk ? gtot - g0 : g0, g = blk * g0 + (li < ng ? li : ng - 1);
        const int S0 = part * 14;
#pragma unroll
        for (int s = 0; s < 14; ++s) {
            int row = 16 * ((S0 + s) >> 2) + 4 * lq + ((S0 + s) & 3);
            row = row < N ? row : N - 1;
            gv[s] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(Gk + (size_t)row * N + 4 * g));
        }
    }
    float acc = 0.f;
#pragma unroll
    for (int s = 0; s < 14; ++s) acc += (gv[s].x + gv[s].y) + (gv[s].z + gv[s].w);
    if (acc == 123.456f) out[blockIdx.x] = acc;
}

template <bool FLAT>
float run_pattern(const std::vector<f32x4*>& bufs, int slices, float* out, int iters)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 20; ++it) hipLaunchKernelGGL(agg_pattern_kernel<FLAT>, dim3(slices), dim3(256), 0, nullptr, (const float*)bufs[it % bufs.size()], out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0, nullptr);
    for (int it = 0; it < iters; ++it) hipLaunchKernelGGL(agg_pattern_kernel<FLAT>, dim3(slices), dim3(256), 0, nullptr, (const float*)bufs[it % bufs.size()], out, 100);
    hipEventRecord(e1, nullptr); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return 1e3f * ms / iters;
}

// agg_fwd_mfma4_kernel<14, 2, 2, 2> taken apart (N = 100, F = 6, K = 3; X (B, K, 6, N), Y (B, K, 6, N)): what each part adds to the
// request stream.  LV 2: + the X requests, the 4x4x1 MFMAs and the row-class sums through the wave's LDS area (agg_mfma_rows); LV 3: + the part combine behind the barrier; LV 4: + the stores =
// the kernel itself (csrc/agg.hip).
#include "../../multiagent_gnn_policies_amd/csrc/agg_mfma.h"
template <int LV>
__global__ __launch_bounds__(256) void agg_parts_kernel(const float* __restrict__ X, const float* __restrict__ G, float* __restrict__ Y, float* __restrict__ out)
{
    constexpr int S = 14, FH = 2, N = 100, C = 6, K = 3;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lq = lane >> 4;
    const int blk = wave >> 1, part = wave & 1;
    const int b = blockIdx.x / K, k = blockIdx.x - b * K;
    const int gtot = N >> 2, g0 = (gtot + 1) >> 1, ng = blk ? gtot - g0 : g0, g = blk * g0 + min(li, ng - 1);
    const float* Gk = G + (size_t)blockIdx.x * N * N + 4 * g;
    const float* Xk = X + ((size_t)b * K + k) * C * N;
    f32x4* red = reinterpret_cast<f32x4*>(smem) + wave * (4 * 64);
    f32x4* comb = reinterpret_cast<f32x4*>(smem) + 4 * (4 * 64);
    f32x4 mine[FH];
    if (LV >= 2) {
        auto keep = [&](int h, const f32x4& tot) { mine[h] = tot; };
        if (part == 0) agg_mfma_rows<S, 0, FH, true>(Gk, Xk, N, C, N, lane, red, keep);
        else agg_mfma_rows<S, S, FH, true>(Gk, Xk, N, C, N, lane, red, keep);
    } else {
        const int S0 = part * S, T0 = S0 >> 2;
        constexpr int TQ = 4;
        f32x4 xa[FH][TQ], gv[S];
#pragma unroll
        for (int h = 0; h < FH; ++h) {
            const float* xr = Xk + (size_t)min(4 * h + (li & 3), C - 1) * N;
#pragma unroll
            for (int t4 = 0; t4 < TQ; ++t4) xa[h][t4] = *reinterpret_cast<const f32x4*>(xr + min(16 * (T0 + t4) + 4 * lq, N - 4));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < S; ++s) {
            gv[s] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(Gk + (size_t)min(16 * ((S0 + s) >> 2) + 4 * lq + ((S0 + s) & 3), N - 1) * N));
            __builtin_amdgcn_sched_barrier(0);
        }
        f32x4 acc[4][FH];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int h = 0; h < FH; ++h) acc[t][h] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < S; ++s) {
#pragma unroll
            for (int h = 0; h < FH; ++h) {
                const float a = xa[h][((S0 + s) >> 2) - T0][(S0 + s) & 3];
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t][h] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, gv[s][t], acc[t][h], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int h = 0; h < FH; ++h) mine[h] = (acc[0][h] + acc[1][h]) + (acc[2][h] + acc[3][h]);
    }
    if (LV >= 3) {
        if (part != 0) {
#pragma unroll
            for (int h = 0; h < FH; ++h) comb[(wave * FH + h) * 64 + lane] = mine[h];
        }
        __syncthreads();
        if (part == 0) {
#pragma unroll
            for (int h = 0; h < FH; ++h) mine[h] += comb[((wave + 1) * FH + h) * 64 + lane];
        }
    }
    if (LV >= 4) {
        // LV 5: the same stores with the non-temporal hint; LV 6: the same stores into a 600 KB window (256 slices: stays in the L2s)
        if (part == 0 && li < ng) {
            float* Yk = Y + (LV == 6 ? (size_t)(blockIdx.x & 255) : ((size_t)b * K + k)) * C * N + 4 * g + lq;
#pragma unroll
            for (int h = 0; h < FH; ++h)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (4 * h + i < C) {
                        if (LV == 5) __builtin_nontemporal_store(mine[h][i], Yk + (size_t)(4 * h + i) * N);
                        else Yk[(size_t)(4 * h + i) * N] = mine[h][i];
                    }
        }
    } else {
        const float a = (mine[0].x + mine[0].y) + (mine[1].z + mine[1].w);
        if (a == 123.456f) out[blockIdx.x] = a;
    }
}

template <int LV>
float run_parts(const std::vector<f32x4*>& bufs, const float* X, float* Y, int slices, float* out, int iters)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t lds = (size_t)(4 * 4 * 64 + 4 * 2 * 64) * 16;
    for (int it = 0; it < 20; ++it) hipLaunchKernelGGL(agg_parts_kernel<LV>, dim3(slices), dim3(256), lds, nullptr, X, (const float*)bufs[it % bufs.size()], Y, out);
    hipDeviceSynchronize();
    hipEventRecord(e0, nullptr);
    for (int it = 0; it < iters; ++it) hipLaunchKernelGGL(agg_parts_kernel<LV>, dim3(slices), dim3(256), lds, nullptr, X, (const float*)bufs[it % bufs.size()], Y, out);
    hipEventRecord(e1, nullptr); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return 1e3f * ms / iters;
}

// LV 7 (experiment): the kernel with its X operands read from an LDS copy of the slice's (8, N) feature block instead of 32 registers
// of up-front float4 loads (fewer live registers: one more wave per SIMD)
__global__ __launch_bounds__(256) void agg_xlds_kernel(const float* __restrict__ X, const float* __restrict__ G, float* __restrict__ Y)
{
    constexpr int S = 14, FH = 2, N = 100, C = 6, K = 3, NP = 112;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lq = lane >> 4;
    const int blk = wave >> 1, part = wave & 1;
    const int b = blockIdx.x / K, k = blockIdx.x - b * K;
    const int gtot = N >> 2, g0 = (gtot + 1) >> 1, ng = blk ? gtot - g0 : g0, g = blk * g0 + min(li, ng - 1);
    const float* Gk = G + (size_t)blockIdx.x * N * N + 4 * g;
    const float* Xk = X + ((size_t)b * K + k) * C * N;
    f32x4* red = reinterpret_cast<f32x4*>(smem) + wave * (4 * 64);
    f32x4* comb = reinterpret_cast<f32x4*>(smem) + 4 * (4 * 64);
    float* xs = reinterpret_cast<float*>(comb + 4 * 2 * 64);            // [8][NP]
    const int S0 = part * S;
    f32x4 gv[S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
        gv[s] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(Gk + (size_t)min(16 * ((S0 + s) >> 2) + 4 * lq + ((S0 + s) & 3), N - 1) * N));
        __builtin_amdgcn_sched_barrier(0);
    }
    for (int e = threadIdx.x; e < 8 * NP; e += 256) { const int c = e / NP, n = e - c * NP; xs[e] = (c < C && n < N) ? Xk[c * N + n] : 0.f; }
    __syncthreads();
    f32x4 acc[4][FH];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int h = 0; h < FH; ++h) acc[t][h] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* xl = xs + (li & 3) * NP + 4 * lq;
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int ro = 16 * ((S0 + s) >> 2) + ((S0 + s) & 3);        // + 4 lq: rows past N read the zero padding (NP = 112)
#pragma unroll
        for (int h = 0; h < FH; ++h) {
            const float a = xl[4 * h * NP + ro];
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t][h] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, gv[s][t], acc[t][h], 0, 0, 0);
        }
    }
    f32x4 mine[FH];
#pragma unroll
    for (int h = 0; h < FH; ++h) {
#pragma unroll
        for (int t = 0; t < 4; ++t) red[t * 64 + lane] = acc[t][h];
        const f32x4* p = red + lq * 64 + li;
        f32x4 tot = p[0];
#pragma unroll
        for (int q = 1; q < 4; ++q) tot += p[16 * q];
        mine[h] = tot;
    }
    if (part != 0) {
#pragma unroll
        for (int h = 0; h < FH; ++h) comb[(wave * FH + h) * 64 + lane] = mine[h];
    }
    __syncthreads();
    if (part == 0 && li < ng) {
        float* Yk = Y + ((size_t)b * K + k) * C * N + 4 * g + lq;
#pragma unroll
        for (int h = 0; h < FH; ++h) {
            const f32x4 tot = mine[h] + comb[((wave + 1) * FH + h) * 64 + lane];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (4 * h + i < C) Yk[(size_t)(4 * h + i) * N] = tot[i];
        }
    }
}

float run_xlds(const std::vector<f32x4*>& bufs, const float* X, float* Y, int slices, int iters)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t lds = (size_t)(4 * 4 * 64 + 4 * 2 * 64) * 16 + 8 * 112 * 4;
    for (int it = 0; it < 20; ++it) hipLaunchKernelGGL(agg_xlds_kernel, dim3(slices), dim3(256), lds, nullptr, X, (const float*)bufs[it % bufs.size()], Y);
    hipDeviceSynchronize();
    hipEventRecord(e0, nullptr);
    for (int it = 0; it < iters; ++it) hipLaunchKernelGGL(agg_xlds_kernel, dim3(slices), dim3(256), lds, nullptr, X, (const float*)bufs[it % bufs.size()], Y);
    hipEventRecord(e1, nullptr); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return 1e3f * ms / iters;
}

int main()
{
    struct Case { const char* tag; size_t bytes; int nbuf; } cases[] = {
        {"cfg-2 operator, B = 256 (30.7 MB)", (size_t)256 * 3 * 100 * 100 * 4, 11},
        {"B = 2048 (245.8 MB)", (size_t)2048 * 3 * 100 * 100 * 4, 3},
        {"cfg-3 operator (768 MB)", (size_t)64 * 3 * 1000 * 1000 * 4, 2}};
    float* out; hipMalloc(&out, 1 << 20);
    for (auto& c : cases) {
        std::vector<f32x4*> bufs(c.nbuf);
        for (auto& b : bufs) { hipMalloc(&b, c.bytes); hipMemset(b, 0x3c, c.bytes); }
        const size_t n4 = c.bytes / 16;
        printf("%s\n", c.tag);
        float best = 1e30f; char bests[128] = "";
        const int wgss[] = {256, 512, 768, 1024, 2048, 4096, 8192};
        const int thrs[] = {256, 512, 1024};
        for (int threads : thrs) for (int wgs : wgss) {
            float t[3] = {run<4>(bufs, n4, wgs, threads, out, 200), run<8>(bufs, n4, wgs, threads, out, 200), run<16>(bufs, n4, wgs, threads, out, 200)};
            for (int k = 0; k < 3; ++k) if (t[k] < best) { best = t[k]; snprintf(bests, sizeof bests, "%d workgroups x %d threads, %d float4 in flight per thread", wgs, threads, 4 << k); }
            printf("  %5d x %4d threads: U=4 %7.2f us  U=8 %7.2f us  U=16 %7.2f us\n", wgs, threads, t[0], t[1], t[2]);
        }
        printf("  best: %.2f us = %.2f TB/s (%.3f of 8 TB/s): %s\n", best, c.bytes / best / 1e6, c.bytes / best / 1e6 / 8.0, bests);
        if (c.bytes % (100 * 100 * 4) == 0 && c.bytes < (size_t)500e6) {
            const int slices = (int)(c.bytes / (100 * 100 * 4));
            const float tp = run_pattern<false>(bufs, slices, out, 200), tf = run_pattern<true>(bufs, slices, out, 200);
            printf("  one workgroup of 256 threads per 100 x 100 slice (%d workgroups): the aggregation kernel's requests %.2f us = %.2f TB/s | flat float4 per slice %.2f us = %.2f TB/s\n",
                   slices, tp, c.bytes / tp / 1e6, tf, c.bytes / tf / 1e6);
            float *X, *Y; hipMalloc(&X, (size_t)slices * 6 * 100 * 4); hipMalloc(&Y, (size_t)slices * 6 * 100 * 4); hipMemset(X, 0x3c, (size_t)slices * 6 * 100 * 4);
            const float t2 = run_parts<2>(bufs, X, Y, slices, out, 200);
            const float t3 = run_parts<3>(bufs, X, Y, slices, out, 200), t4 = run_parts<4>(bufs, X, Y, slices, out, 200);
            const float t5 = run_parts<5>(bufs, X, Y, slices, out, 200), t6 = run_parts<6>(bufs, X, Y, slices, out, 200);
            printf("  the aggregation kernel in parts: + X requests, MFMAs, row-class sums through LDS %.2f us | + part combine behind the barrier %.2f us | + stores (the kernel) %.2f us | non-temporal stores %.2f us | stores into a 600 KB window %.2f us\n", t2, t3, t4, t5, t6);
            printf("  experiment: X operands from an LDS copy instead of registers: %.2f us\n", run_xlds(bufs, X, Y, slices, 200));
            hipFree(X); hipFree(Y);
        }
        for (auto& b : bufs) hipFree(b);
    }
    return 0;
}
