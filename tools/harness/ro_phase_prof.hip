// Standalone phase profiler for the episode-resident rollout kernel (cycle stamps of workgroup 0, thread 0, step 1).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o ro_prof tools/harness/ro_phase_prof.hip && ./ro_prof 256 100 3 200
#define MGP_RO_PROFILE 1
#include "../../multiagent_gnn_policies_amd/csrc/rollout.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <cmath>
thread_local int mgp_tls_hip_error = 0;
thread_local void* mgp_tls_launch_events[2] = {nullptr, nullptr};
// the narrow build forwards widths > 32 to the second compilation (rollout_wide.hip); this harness links only the narrow one
extern "C" int mgp_rollout_wide_supported_(const int*, int, int, int) { return 0; }
extern "C" int mgp_rollout_wide_steps_ex_(double*, float*, float*, const float* const*, const float* const*, const int*, int, float*,
                                          double*, const MgpFlockParams*, int, int, int, int, const float*, void*, int, void*) { return MGP_EUNSUPPORTED; }
extern "C" int mgp_rollout_wide_collect_(double*, float*, float*, const float* const*, const float* const*, const int*, int, double*,
                                         const MgpFlockParams*, int, int, int, int, const float*, void*, int, const MgpCollect*, void*) { return MGP_EUNSUPPORTED; }
extern "C" long mgp_rollout_wide_image_floats_(const int*, int, int, int) { return 0; }
extern "C" int mgp_rollout_wide_image_(const float* const*, const float* const*, const int*, int, int, int, float*, void*) { return MGP_EUNSUPPORTED; }
int main(int argc, char** argv) {
    int B = argc > 1 ? atoi(argv[1]) : 256, N = argc > 2 ? atoi(argv[2]) : 100, K = argc > 3 ? atoi(argv[3]) : 3;
    int T = argc > 4 ? atoi(argv[4]) : 200;
    std::vector<double> hx((size_t)B * N * 4);
    const bool rnd = argc > 6;                                   // 7th argument: irregular (hash-scattered) positions
    for (int b = 0; b < B; ++b) for (int i = 0; i < N; ++i) {
        int gx = i % 10, gy = i / 10;
        double jx = 0.01 * ((i * 7 + b) % 13), jy = 0.01 * ((i * 5 + b) % 11);
        if (rnd) {                                               // uniform-ish scatter over a 7 x 7 box: Poisson-like degrees, mean ~6
            unsigned h1 = (unsigned)(i * 2654435761u + b * 40503u), h2 = (unsigned)(i * 2246822519u + b * 3266489917u + 12345u);
            jx = 7.0 * ((h1 >> 8) & 0xFFFF) / 65536.0 - 0.6 * gx; jy = 7.0 * ((h2 >> 8) & 0xFFFF) / 65536.0 - 0.6 * gy;
        }
        hx[((size_t)b * N + i) * 4 + 0] = 0.6 * gx + jx;
        hx[((size_t)b * N + i) * 4 + 1] = 0.6 * gy + jy;
        hx[((size_t)b * N + i) * 4 + 2] = 0.1 * ((i * 3) % 17) - 0.8;
        hx[((size_t)b * N + i) * 4 + 3] = 0.1 * ((i * 11) % 19) - 0.9;
    }
    const int HW = getenv("RO_HIDDEN") ? atoi(getenv("RO_HIDDEN")) : 32;      // hidden width (two layers)
    const int dims[4] = {6, HW, HW, 2};
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
    double *x, *rew; float *G, *Xd, *act;
    hipMalloc(&x, hx.size() * 8); hipMalloc(&rew, (size_t)B * T * 8);
    hipMalloc(&G, (size_t)B * K * N * N * 4); hipMalloc(&Xd, (size_t)B * K * 6 * N * 4); hipMalloc(&act, (size_t)B * 2 * N * 4);
    hipMemcpy(x, hx.data(), hx.size() * 8, hipMemcpyHostToDevice);
    hipMemset(G, 0, (size_t)B * K * N * N * 4); hipMemset(Xd, 0, (size_t)B * K * 6 * N * 4);
    if (const char* dump = getenv("RO_STATE")) {                 // optional: start from a state dumped by tools/dump_rollout_state.py
        FILE* f = fopen(dump, "rb");
        if (!f) { printf("cannot open %s\n", dump); return 1; }
        std::vector<float> hg((size_t)B * K * N * N), hxd((size_t)B * K * 6 * N);
        size_t ok = fread(hx.data(), 8, hx.size(), f) + fread(hg.data(), 4, hg.size(), f) + fread(hxd.data(), 4, hxd.size(), f);
        for (int l = 0; l < 3; ++l) { ok += fread(hw[l].data(), 4, hw[l].size(), f); ok += fread(hb[l].data(), 4, hb[l].size(), f); }
        fclose(f);
        hipMemcpy(x, hx.data(), hx.size() * 8, hipMemcpyHostToDevice);
        hipMemcpy(G, hg.data(), hg.size() * 4, hipMemcpyHostToDevice); hipMemcpy(Xd, hxd.data(), hxd.size() * 4, hipMemcpyHostToDevice);
        for (int l = 0; l < 3; ++l) { hipMemcpy(W[l], hw[l].data(), hw[l].size() * 4, hipMemcpyHostToDevice); hipMemcpy(bb[l], hb[l].data(), hb[l].size() * 4, hipMemcpyHostToDevice); }
        printf("state from %s (%zu values)\n", dump, ok);
    }
    MgpFlockParams p = {1.0, 0.01, 10.0, 1.0, 0.1, 10.0, 1.0, 1, 0, 1, 0};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    // RO_CARRY=1: the repeated-launch form (prebuilt weight image, factored hand-over of the history, dense slices not rebuilt)
    const bool use_carry = getenv("RO_CARRY") != nullptr;
    float* image = nullptr; void* carry = nullptr;
    if (use_carry) {
        hipMalloc(&image, mgp_rollout_image_floats(dims, 3, K, N) * 4);
        hipMalloc(&carry, (size_t)B * mgp_rollout_carry_bytes(K, N)); hipMemset(carry, 0, (size_t)B * mgp_rollout_carry_bytes(K, N));
        if (mgp_rollout_image(W, bb, dims, 3, K, N, image, nullptr)) { printf("image failed\n"); return 1; }
    }
    const int fl = use_carry ? (MGP_RO_ENTER_CARRY | MGP_RO_EXIT_CARRY | MGP_RO_SKIP_DENSE) : 0;
    int rc = mgp_rollout_steps_ex(x, G, Xd, W, bb, dims, 3, act, rew, &p, B, K, N, T, image, carry, fl, nullptr);
    if (rc) { printf("rc %d\n", rc); return 1; }
    hipDeviceSynchronize();
    const int IT = argc > 5 ? atoi(argv[5]) : 5;
    hipEventRecord(e0, nullptr);
    for (int it = 0; it < IT; ++it) mgp_rollout_steps_ex(x, G, Xd, W, bb, dims, 3, act, rew, &p, B, K, N, T, image, carry, fl, nullptr);
    hipEventRecord(e1, nullptr); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%sB=%d N=%d K=%d T=%d resident rollout: %.1f us per launch, %.2f us per step -> %.3e agent-steps/s\n", use_carry ? "[carry] " : "", B, N, K, T,
           1e3 * ms / IT, 1e3 * ms / IT / T, (double)B * N * T / (1e-3 * ms / IT));
    {   // bit-level fingerprint of the final state and the last launch's rewards (A/B builds of the same arithmetic must agree)
        std::vector<unsigned long long> fx((size_t)B * N * 4), fr((size_t)B * T);
        hipMemcpy(fx.data(), x, fx.size() * 8, hipMemcpyDeviceToHost); hipMemcpy(fr.data(), rew, fr.size() * 8, hipMemcpyDeviceToHost);
        unsigned long long h1 = 1469598103934665603ull, h2 = h1;
        for (unsigned long long v : fx) { h1 ^= v; h1 *= 1099511628211ull; }
        for (unsigned long long v : fr) { h2 ^= v; h2 *= 1099511628211ull; }
        printf("fingerprint: state %016llx rewards %016llx\n", h1, h2);
    }
    std::vector<unsigned int> vs(4096 * 4);
    hipMemcpyFromSymbol(vs.data(), HIP_SYMBOL(mgp_ro_vstat), vs.size() * 4);
    {
        unsigned long long tot[3] = {0, 0, 0}; unsigned int mxr = 0, mxf = 0;
        for (int b = 0; b < B && b < 4096; ++b) {
            for (int m = 0; m < 3; ++m) tot[m] += vs[b * 4 + m];
            mxr = std::max(mxr, vs[b * 4 + 1]); mxf = std::max(mxf, vs[b * 4 + 2]);
        }
        printf("S1 modes over all launches (cheap / rebuild / full): workgroup 0 %u / %u / %u; all workgroups %llu / %llu / %llu; most rebuilds in one workgroup %u, most full steps %u\n",
               vs[0], vs[1], vs[2], tot[0], tot[1], tot[2], mxr, mxf);
    }
    unsigned long long st[512];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(mgp_ro_stamps), sizeof(st));
    const char* names[] = {"step start", "A done (barrier)", "B hidden layers (waves 0-6) + G_1 expansion (waves 7-15) done (barrier)", "B+C: hidden layers, output layer, integration (waves 0-6) / bit clearing (waves 7-15) done (barrier)",
                           "D2/D3 done (barrier)", "step done", "A: first gather stage of this wave done", "D1: membership bits done (barrier)",
                           "D2/D3: lists + neighbour feature terms done (waves 0-6)", "C: max published (atomic issued)", "D: lists written", "E: rows of slices >= 2 done (barrier)", "B: layer 0 tile done (before barrier)", "B: layer 1 tile done", "C: output layer + quad sums done", "C: integrated, coordinates stored", "S1: pair tests done", "S1: fallback / fading done", "S1: row word combined", "S1: list written", "S2: gather group loop done", "S2: reward wave done", "S2: features group done", "S2: gather group done", "S2: Verlet helper wave done"};
    const int order[] = {0, 6, 1, 12, 13, 14, 15, 3, 16, 17, 18, 19, 7, 8, 22, 20, 23, 21, 24, 4, 5};
    printf("cycles since step start, lane 0 of waves 0 .. 15\n");
    const int wv[] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15};
    for (int oi = 0; oi < 21; ++oi) {
        const int i = order[oi];
        printf("  stamp %2d :", i);
        for (int w = 0; w < 16; ++w) {
            const long long d = (long long)(st[wv[w] * 32 + i] - st[0]);
            if (d > -100000 && d < 10000000) printf(" %6lld", d); else printf("      -");
        }
        printf("  %s\n", names[i]);
    }
    printf("exit, wave 0 (cycles since exit start): after slice K-1 %lld | after slice K-2 %lld | kernel end %lld\n", (long long)(st[11] - st[10]), (long long)(st[12] - st[10]), (long long)(st[2] - st[10]));
    return 0;
}
