// Standalone timing harness of the factored-state step (N > 256): K launches per env step -- gather stages, policy tail,
// cell-list simulator -- on a jittered lattice, cfg-3 shape by default.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o scratch/sp_prof tools/harness/sp_step_prof.hip && ./scratch/sp_prof 64 1000 3 100
#include "../../multiagent_gnn_policies_amd/csrc/sparse_sim.hip"
#include "../../multiagent_gnn_policies_amd/csrc/sparse_policy.hip"
#include "../../multiagent_gnn_policies_amd/csrc/sparse_persist.hip"   // (mgp_sparse_rollout links against it; this harness keeps the K-launch form)
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
thread_local int mgp_tls_hip_error = 0;
thread_local void* mgp_tls_launch_events[2] = {nullptr, nullptr};
extern "C" int mgp_flock_step_sparse(const double*, double*, const float*, long, long, unsigned long long*, long, float*, long, float*, long,
                                     double*, float*, const MgpFlockParams*, int, int, void*) { return MGP_EUNSUPPORTED; }   // (flock.hip is not linked here)
extern "C" int mgp_sparse_words(int N) { return N <= 0 ? 0 : 8 * (((((N + 7) / 8) + 63) & ~63) / 64); }
int main(int argc, char** argv) {
    setenv("MGP_SP_PERSIST", "0", 1);                        // one call per step here: the K launches are what is timed
    int B = argc > 1 ? atoi(argv[1]) : 64, N = argc > 2 ? atoi(argv[2]) : 1000, K = argc > 3 ? atoi(argv[3]) : 3;
    int T = argc > 4 ? atoi(argv[4]) : 100;
    const int side = (int)ceil(sqrt((double)N));
    std::vector<double> hx((size_t)B * N * 4);
    for (int b = 0; b < B; ++b) for (int i = 0; i < N; ++i) {
        // hash-permuted lattice sites: neighbours are NOT adjacent in index (as after a disc / sorted-by-radius reset)
        const int s = (int)(((unsigned long long)i * 7919ull + 13ull * b) % (unsigned long long)(side * side));
        int gx = s % side, gy = s / side;
        double jx = 0.01 * ((i * 7 + b) % 13), jy = 0.01 * ((i * 5 + b) % 11);
        hx[((size_t)b * N + i) * 4 + 0] = 0.6 * gx + jx;
        hx[((size_t)b * N + i) * 4 + 1] = 0.6 * gy + jy;
        hx[((size_t)b * N + i) * 4 + 2] = 0.02 * ((i * 3) % 17) - 0.16;
        hx[((size_t)b * N + i) * 4 + 3] = 0.02 * ((i * 11) % 19) - 0.18;
    }
    // distinct sites: side*side >= N and 7919 coprime with side*side for the shapes used (checked below)
    {
        std::vector<int> seen(side * side, 0); int dup = 0;
        for (int i = 0; i < N; ++i) { int s = (int)(((unsigned long long)i * 7919ull) % (unsigned long long)(side * side)); dup += seen[s]++; }
        if (dup) printf("warning: %d duplicate sites\n", dup);
    }
    const int dims[4] = {6, 32, 32, 2};
    std::vector<float> hw[3], hb[3];
    float *W[3], *bb[3];
    for (int l = 0; l < 3; ++l) {
        const int cin = l == 0 ? 6 * K : dims[l], cout = dims[l + 1];
        hw[l].resize((size_t)cin * cout); hb[l].resize(cout);
        for (size_t i = 0; i < hw[l].size(); ++i) hw[l][i] = 0.05f * (float)((int)((i * 37) % 23) - 11) / 11.f;
        for (int i = 0; i < cout; ++i) hb[l][i] = 0.01f * i;
        hipMalloc(&W[l], hw[l].size() * 4); hipMalloc(&bb[l], hb[l].size() * 4);
        hipMemcpy(W[l], hw[l].data(), hw[l].size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(bb[l], hb[l].data(), hb[l].size() * 4, hipMemcpyHostToDevice);
    }
    const int H = K > 2 ? K - 1 : 1, NW = mgp_sparse_words(N);
    double *x[2], *rew; float *wrow, *feat, *scratch, *act, *expert, *image; unsigned long long* bits;
    hipMalloc(&x[0], hx.size() * 8); hipMalloc(&x[1], hx.size() * 8); hipMalloc(&rew, (size_t)B * 8);
    hipMemcpy(x[0], hx.data(), hx.size() * 8, hipMemcpyHostToDevice);
    hipMalloc(&bits, (size_t)B * H * N * NW * 8); hipMemset(bits, 0, (size_t)B * H * N * NW * 8);
    hipMalloc(&wrow, (size_t)B * H * N * 4); hipMemset(wrow, 0, (size_t)B * H * N * 4);
    hipMalloc(&feat, (size_t)B * K * N * 8 * 4); hipMemset(feat, 0, (size_t)B * K * N * 8 * 4);
    hipMalloc(&scratch, (size_t)2 * K * B * N * 8 * 4); hipMemset(scratch, 0, (size_t)2 * K * B * N * 8 * 4);
    hipMalloc(&act, (size_t)B * 2 * N * 4); hipMalloc(&expert, (size_t)B * N * 2 * 4);
    hipMalloc(&image, mgp_sparse_policy_image_floats(dims, 3, K) * 4);
    if (mgp_sparse_policy_image(W, bb, dims, 3, K, image, nullptr)) { printf("image failed\n"); return 1; }
    MgpFlockParams p = {1.0, 0.01, 10.0, 1.0, 0.1, 10.0, 1.0, 1, 0, 1, 0};
    int cur = 0, hs = 0, xi = 0, rc;
    unsigned short* nbr = nullptr;
    if (!getenv("SP_NOLISTS")) { hipMalloc(&nbr, (size_t)B * H * N * 16 * 2); hipMemset(nbr, 0, (size_t)B * H * N * 16 * 2); }
    rc = mgp_flock_step_cells_nbr(x[0], x[1], nullptr, 2, 1, bits, (long)H * N * NW, wrow, (long)H * N, feat, (long)K * N * 8, nbr, (long)H * N * 16, nullptr, expert, &p, B, N, nullptr);
    if (rc) { printf("observe rc %d\n", rc); return 1; }
    const char* only = getenv("SP_ONLY");                       // "policy": gather + policy launches only (no producer before the gather)
    auto step = [&]() {
        if (only && only[0] == 'p') return mgp_sparse_policy_step(bits, wrow, feat, image, dims, 3, scratch, act, B, K, N, cur, hs, nullptr);
        const int r = mgp_sparse_rollout(bits, wrow, feat, image, dims, 3, scratch, act, x[xi], x[xi ^ 1], rew, expert, &p, B, K, N, 1, &cur, &hs, nullptr, nbr, nullptr);
        xi ^= 1;
        return r;
    };
    for (int t = 0; t < 10; ++t) if ((rc = step())) { printf("step rc %d\n", rc); return 1; }
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, nullptr);
    for (int t = 0; t < T; ++t) step();
    hipEventRecord(e1, nullptr); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // state summary: mean degree from the row weights, checksum of x and the action
    std::vector<float> hwq((size_t)B * H * N); hipMemcpy(hwq.data(), wrow, hwq.size() * 4, hipMemcpyDeviceToHost);
    double dsum = 0; for (int b = 0; b < B; ++b) for (int i = 0; i < N; ++i) { float w = hwq[((size_t)b * H + hs) * N + i]; dsum += w > 0 ? 1.0 / w : 0; }
    hipMemcpy(hx.data(), x[xi], hx.size() * 8, hipMemcpyDeviceToHost);
    double cs = 0; for (size_t i = 0; i < hx.size(); ++i) cs += hx[i] * (double)((i % 97) + 1);
    std::vector<float> ha((size_t)B * 2 * N); hipMemcpy(ha.data(), act, ha.size() * 4, hipMemcpyDeviceToHost);
    double ca = 0; for (size_t i = 0; i < ha.size(); ++i) ca += ha[i] * (double)((i % 89) + 1);
#ifdef MGP_SP_PROFILE
    {
        unsigned long long st[256];
        hipMemcpyFromSymbol(st, HIP_SYMBOL(mgp_ss_stamps), sizeof(st));
        const char* names[] = {"start", "loaded + integrated", "block reductions done", "reward done", "histogram done (barrier)", "scan done (barrier)",
                               "scatter + sort done", "row search done", "outputs written (barrier)", "bit rows written", "row search: pass 1 (tests) done", "row search: row data loaded"};
        {
            unsigned long long sq[512];
            hipMemcpyFromSymbol(sq, HIP_SYMBOL(mgp_sp_stamps), sizeof(sq));
            const char* gn[] = {"start", "requests issued, staged stores done", "barrier", "gather done", "written", "kernel arguments arrived", "requests issued", "requests returned"};
            const char* pn[] = {"start", "requests issued, staged stores done", "barrier", "taps in act (+ frame)", "gather done", "barrier", "hidden layers done", "action written"};
            const int wv[] = {0, 5, 10, 15};
            printf("spl_gather_kernel, workgroup (1,3): cycles since start, lane 0 of waves 0 5 10 15\n");
            for (int i = 0; i < 8; ++i) { printf("  stamp %d :", i); for (int w = 0; w < 4; ++w) printf(" %7lld", (long long)(sq[(0 * 16 + wv[w]) * 16 + i] - sq[0])); printf("  %s\n", gn[i]); }
            printf("spl_policy_kernel, workgroup (1,3): cycles since start, lane 0 of waves 0 5 10 15\n");
            for (int i = 0; i < 8; ++i) { printf("  stamp %d :", i); for (int w = 0; w < 4; ++w) printf(" %7lld", (long long)(sq[(1 * 16 + wv[w]) * 16 + i] - sq[16 * 16])); printf("  %s\n", pn[i]); }
        }
        printf("sp_sim_kernel, workgroup (1,3): cycles since start, lane 0 of waves 0 5 10 15\n");
        for (int i = 0; i < 12; ++i) {
            printf("  stamp %d :", i);
            for (int w = 0; w < 4; ++w) printf(" %7lld", (long long)(st[(5 * w) * 16 + i] - st[0]));
            printf("  %s\n", names[i]);
        }
    }
#endif
    printf("B=%d N=%d K=%d T=%d factored step: %.2f us per step -> %.3e agent-steps/s | mean degree %.2f | checksum x %.12e action %.9e\n",
           B, N, K, T, 1e3 * ms / T, (double)B * N * T / (1e-3 * ms), dsum / ((double)B * N), cs, ca);
    return 0;
}
