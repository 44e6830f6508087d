// Standalone phase profiler for the two-launch DAGGER update (in-kernel cycle stamps of workgroup (0,0), thread 0).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -o scratch/ts_prof tools/harness/train_phase_prof.hip && ./scratch/ts_prof 20 100 3 [1]
// fourth argument 1: the update on the aggregated input (mgp_train_step_agg: X holds Z (B, 6 K, N), no G)
#define MGP_TS_PROFILE 1
#include "../../multiagent_gnn_policies_amd/csrc/train_step.hip"
#include <cstdio>
#include <vector>
thread_local int mgp_tls_hip_error = 0;
int main(int argc, char** argv) {
    int B = argc > 1 ? atoi(argv[1]) : 20, N = argc > 2 ? atoi(argv[2]) : 100, K = argc > 3 ? atoi(argv[3]) : 3;
    const bool agg = argc > 4 && atoi(argv[4]) == 1;
    const int dims[4] = {6, 32, 32, 2};
    const int L = 3;
    long P = 0; for (int l = 0; l < L; ++l) P += (long)dims[l + 1] * (l ? dims[l] : dims[0] * K) + dims[l + 1];
    std::vector<float> hp(P); for (long i = 0; i < P; ++i) hp[i] = 0.05f * (float)((i * 37 % 41) - 20) / 20.f;
    std::vector<float> hx((size_t)B * K * 6 * N), hg(agg ? (size_t)4 : (size_t)B * K * N * N, 0.f), ht((size_t)B * 2 * N);
    for (size_t i = 0; i < hx.size(); ++i) hx[i] = 0.1f * (float)((int)((i * 13) % 29) - 14);
    for (size_t i = 0; i < ht.size(); ++i) ht[i] = 0.1f * (float)((int)((i * 7) % 23) - 11);
    if (!agg) for (int b = 0; b < B; ++b) for (int k = 0; k < K; ++k) for (int i = 0; i < N; ++i) for (int d = -4; d <= 4; ++d)
        hg[(((size_t)b * K + k) * N + i) * N + (i + d + N) % N] = 0.125f;
    float *X, *G, *T, *p, *g, *m, *v, *loss, *ws; int* step;
    hipMalloc(&X, hx.size() * 4); hipMalloc(&G, hg.size() * 4); hipMalloc(&T, ht.size() * 4);
    hipMalloc(&p, P * 4); hipMalloc(&g, P * 4); hipMalloc(&m, P * 4); hipMalloc(&v, P * 4); hipMalloc(&loss, 4); hipMalloc(&step, 4);
    const long nws = mgp_train_workspace(dims, L, B, K, N);
    hipMalloc(&ws, nws * 4); hipMemset(ws, 0, nws * 4);
    hipMemcpy(X, hx.data(), hx.size() * 4, hipMemcpyHostToDevice); hipMemcpy(G, hg.data(), hg.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(T, ht.data(), ht.size() * 4, hipMemcpyHostToDevice); hipMemcpy(p, hp.data(), P * 4, hipMemcpyHostToDevice);
    hipMemset(m, 0, P * 4); hipMemset(v, 0, P * 4); hipMemset(step, 0, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto update = [&]() { return agg ? mgp_train_step_agg(X, T, nullptr, nullptr, nullptr, 0, p, g, m, v, dims, L, 5e-5f, 0.9f, 0.999f, 1e-8f, step, loss, ws, B, K, N, nullptr, nullptr)
                                     : mgp_train_step(X, G, T, p, g, m, v, dims, L, 5e-5f, 0.9f, 0.999f, 1e-8f, step, loss, ws, B, K, N, nullptr); };
    for (int it = 0; it < 5; ++it) update();
    hipDeviceSynchronize();
    const int IT = 200;
    hipEventRecord(e0, nullptr);
    for (int it = 0; it < IT; ++it) { int rc = update(); if (rc) { printf("rc %d\n", rc); return 1; } }
    hipEventRecord(e1, nullptr); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    float hl; int hs; hipMemcpy(&hl, loss, 4, hipMemcpyDeviceToHost); hipMemcpy(&hs, step, 4, hipMemcpyDeviceToHost);
    printf("B=%d N=%d K=%d two-launch update%s: %.2f us per update (loss %.5f, step counter %d)\n", B, N, K, agg ? " on the aggregated input" : "", 1e3 * ms / IT, hl, hs);
    unsigned long long st[32];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(mgp_ts_stamps), sizeof(st));
    const char* names[] = {"start", "all global reads landed (barrier)", "aggregation pieces done (barrier)", "layer 0 inputs ready", "layer 1 inputs ready",
                           "layer 2 inputs ready", "-", "-", "forward + loss done", "backward layer L-1", "backward layer L-2", "backward layer L-3", "-", "-", "end"};
    for (int i = 0; i < 15; ++i) if (names[i][0] != '-' && !(agg && (i == 1 || i == 2))) printf("  stamp %2d : %8llu  %s\n", i, st[i] - st[0], names[i]);
    for (int i = 20; i < 28; ++i) if (st[i]) printf("  stamp %2d : %8llu\n", i, st[i] - st[0]);
    return 0;
}
