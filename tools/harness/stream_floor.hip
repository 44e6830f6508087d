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
        for (auto& b : bufs) hipFree(b);
    }
    return 0;
}
