// Standalone phase profiler for the flocking step kernel (in-kernel cycle stamps of workgroup (0,0), thread 0).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o fl_prof tools/harness/flock_phase_prof.hip && ./fl_prof 256 100
#define MGP_FL_PROFILE 1
#include "../../multiagent_gnn_policies_amd/csrc/flock.hip"
#include <cstdio>
#include <vector>
#include <cmath>
thread_local int mgp_tls_hip_error = 0;
int main(int argc, char** argv) {
    int B = argc > 1 ? atoi(argv[1]) : 256, N = argc > 2 ? atoi(argv[2]) : 100;
    std::vector<double> hx((size_t)B * N * 4);
    for (int b = 0; b < B; ++b) for (int i = 0; i < N; ++i) {
        int gx = i % 10, gy = i / 10;
        hx[((size_t)b * N + i) * 4 + 0] = 0.6 * gx + 0.01 * ((i * 7 + b) % 13);
        hx[((size_t)b * N + i) * 4 + 1] = 0.6 * gy + 0.01 * ((i * 5 + b) % 11);
        hx[((size_t)b * N + i) * 4 + 2] = 0.1 * ((i * 3) % 17) - 0.8;
        hx[((size_t)b * N + i) * 4 + 3] = 0.1 * ((i * 11) % 19) - 0.9;
    }
    double *x, *xo, *rew; float *u, *A, *feat, *ex;
    hipMalloc(&x, hx.size() * 8); hipMalloc(&xo, hx.size() * 8); hipMalloc(&rew, B * 8);
    hipMalloc(&u, (size_t)B * N * 2 * 4); hipMalloc(&A, (size_t)B * N * N * 4); hipMalloc(&feat, (size_t)B * 6 * N * 4);
    hipMalloc(&ex, (size_t)B * N * 2 * 4);
    hipMemcpy(x, hx.data(), hx.size() * 8, hipMemcpyHostToDevice); hipMemset(u, 0, (size_t)B * N * 2 * 4);
    MgpFlockParams p = {1.0, 0.01, 10.0, 1.0, 0.1, 10.0, 1.0, 1, 0};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 4; ++it) { mgp_flock_step(x, xo, u, 2, 1, A, nullptr, feat, nullptr, rew, ex, 0, 0, &p, B, N, nullptr); std::swap(x, xo); }
    hipDeviceSynchronize();
    const int IT = 100;
    hipEventRecord(e0, nullptr);
    for (int it = 0; it < IT; ++it) { int rc = mgp_flock_step(x, xo, u, 2, 1, A, nullptr, feat, nullptr, rew, ex, 0, 0, &p, B, N, nullptr); if (rc) { printf("rc %d\n", rc); return 1; } std::swap(x, xo); }
    hipEventRecord(e1, nullptr); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("B=%d N=%d ping-pong sim step: %.2f us per launch\n", B, N, 1e3 * ms / IT);
    {   // fused sim + delayed-GSO transition (K = 3)
        const int K = 3; float *G0, *G1, *X0, *X1;
        hipMalloc(&G0, (size_t)B * K * N * N * 4); hipMalloc(&G1, (size_t)B * K * N * N * 4);
        hipMalloc(&X0, (size_t)B * K * 6 * N * 4); hipMalloc(&X1, (size_t)B * K * 6 * N * 4);
        hipMemset(G0, 0, (size_t)B * K * N * N * 4); hipMemset(G1, 0, (size_t)B * K * N * N * 4);
        hipMemset(X0, 0, (size_t)B * K * 6 * N * 4); hipMemset(X1, 0, (size_t)B * K * 6 * N * 4);
        for (int it = 0; it < 4; ++it) { mgp_flock_step_advance(x, xo, u, 2, 1, G0, G1, X0, X1, rew, ex, &p, B, K, N, 1, nullptr); std::swap(x, xo); std::swap(G0, G1); std::swap(X0, X1); }
        hipDeviceSynchronize();
        hipEventRecord(e0, nullptr);
        for (int it = 0; it < IT; ++it) { int rc = mgp_flock_step_advance(x, xo, u, 2, 1, G0, G1, X0, X1, rew, ex, &p, B, K, N, 1, nullptr); if (rc) { printf("rc %d\n", rc); return 1; } std::swap(x, xo); std::swap(G0, G1); std::swap(X0, X1); }
        hipEventRecord(e1, nullptr); hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1);
        printf("B=%d N=%d fused sim + state step (K=3): %.2f us per launch\n", B, N, 1e3 * ms / IT);
    }
    unsigned long long st[32];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(mgp_fl_stamps), sizeof(st));
    const char* names[] = {"start", "x/u loaded + integrated (barrier)", "reward sums done (wg 0)", "pairwise phases done", "partials combined, features/expert written", "barrier before the row phases", "delayed-GSO rows written", "network rows written"};
    for (int i = 0; i < 8; ++i) printf("  stamp %d : %8llu  %s\n", i, st[i] - st[0], names[i]);
    const char* names2[] = {"start", "agents integrated, source slice in LDS (barrier)", "episode sums, relative coordinates (barrier)", "membership, features, lists (barrier)", "network rows stored", "product rows stored"};
    printf("flock_advance_kernel (one workgroup per episode), workgroup 0:\n");
    for (int i = 8; i < 14; ++i) printf("  stamp %d : %8lld  %s\n", i, (long long)(st[i] - st[8]), names2[i - 8]);
    return 0;
}
