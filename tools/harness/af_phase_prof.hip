// Standalone phase profiler for the fused Actor forward (in-kernel cycle stamps at phase boundaries).
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DMGP_AF_MLP_STAMPS] -o af_prof tools/harness/af_phase_prof.hip && ./af_prof 256 100
#define MGP_AF_PROFILE 1
#include "../../multiagent_gnn_policies_amd/csrc/actor_fused.hip"
#include <cstdio>
#include <vector>
thread_local int mgp_tls_hip_error = 0;
int main(int argc, char** argv) {
    int B = 256, K = 3, N = 100, F = 6;
    if (argc > 1) B = atoi(argv[1]);
    if (argc > 2) N = atoi(argv[2]);
    int H = 32;
    if (argc > 3) H = atoi(argv[3]);                   // hidden width of both layers (32: the compiled policy shape)
    int dims[4] = {F, H, H, 2};
    size_t nG = (size_t)B * K * N * N, nX = (size_t)B * K * F * N;
    const int NSETS = 10;
    std::vector<float*> Gs(NSETS), Xs(NSETS);
    std::vector<float> h(nG, 0.01f);
    for (int i = 0; i < NSETS; ++i) {
        hipMalloc(&Gs[i], nG * 4); hipMalloc(&Xs[i], nX * 4);
        hipMemcpy(Gs[i], h.data(), nG * 4, hipMemcpyHostToDevice);
        hipMemcpy(Xs[i], h.data(), nX * 4, hipMemcpyHostToDevice);
    }
    float *W0, *W1, *W2, *b0, *b1, *b2, *out;
    hipMalloc(&W0, H * 18 * 4); hipMalloc(&W1, H * H * 4); hipMalloc(&W2, 2 * H * 4);
    hipMalloc(&b0, H * 4); hipMalloc(&b1, H * 4); hipMalloc(&b2, 8); hipMalloc(&out, (size_t)B * 2 * N * 4);
    hipMemcpy(W0, h.data(), H * 18 * 4, hipMemcpyHostToDevice); hipMemcpy(W1, h.data(), H * H * 4, hipMemcpyHostToDevice);
    hipMemcpy(W2, h.data(), 2 * H * 4, hipMemcpyHostToDevice); hipMemcpy(b0, h.data(), H * 4, hipMemcpyHostToDevice);
    hipMemcpy(b1, h.data(), H * 4, hipMemcpyHostToDevice); hipMemcpy(b2, h.data(), 8, hipMemcpyHostToDevice);
    const float* W[3] = {W0, W1, W2}; const float* bb[3] = {b0, b1, b2};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) mgp_actor_fwd(Xs[i % NSETS], Gs[i % NSETS], W, bb, dims, 3, out, nullptr, B, K, N, nullptr);
    hipDeviceSynchronize();
    const int IT = 100;
    hipEventRecord(e0, nullptr);
    for (int i = 0; i < IT; ++i) {
        int rc = mgp_actor_fwd(Xs[i % NSETS], Gs[i % NSETS], W, bb, dims, 3, out, nullptr, B, K, N, nullptr);
        if (rc) { printf("rc %d\n", rc); return 1; }
    }
    hipEventRecord(e1, nullptr); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("B=%d N=%d avg %.2f us per launch -> %.1f GB/s (G bytes)\n", B, N, 1e3 * ms / IT, nG * 4.0 / (ms / IT) / 1e6);
    unsigned long long st[64];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(mgp_af_stamps), sizeof(st));
    const char* names[64] = {0};
    printf("block 0 thread 0 cycle stamps (delta from start, cycles @100MHz-ish counter or shader clock):\n");
    for (int i = 1; i < 64; ++i) if (st[i]) printf("  stamp %2d : %10llu\n", i, st[i] - st[0]);
    return 0;
}
