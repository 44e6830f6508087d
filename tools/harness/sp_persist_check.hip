// The persistent form of the factored-state rollout (csrc/sparse_persist.hip) against the K-launch form on the same state:
// every output buffer compared bit for bit, then both timed.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o scratch/sp_persist tools/harness/sp_persist_check.hip && ./scratch/sp_persist 64 1000 100
// (the library builds sparse_policy.hip / sparse_persist.hip with contraction on and sparse_sim.hip with it off; the harness is ONE
// translation unit built with it off: its K-launch form can differ from the library's in the last bit of an action, the comparison
// between the two forms inside the harness is like for like.)
#include "../../multiagent_gnn_policies_amd/csrc/sparse_sim.hip"
#include "../../multiagent_gnn_policies_amd/csrc/sparse_policy.hip"
#include "../../multiagent_gnn_policies_amd/csrc/sparse_persist.hip"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>
thread_local int mgp_tls_hip_error = 0;
thread_local void* mgp_tls_launch_events[2] = {nullptr, nullptr};
extern "C" int mgp_flock_step_sparse(const double*, double*, const float*, long, long, unsigned long long*, long, float*, long, float*, long,
                                     double*, float*, const MgpFlockParams*, int, int, void*) { return MGP_EUNSUPPORTED; }
extern "C" int mgp_sparse_words(int N) { return N <= 0 ? 0 : 8 * (((((N + 7) / 8) + 63) & ~63) / 64); }

struct Buf { void* d; size_t bytes; const char* name; std::vector<unsigned char> snap, a, b; };

int main(int argc, char** argv)
{
    int B = argc > 1 ? atoi(argv[1]) : 64, N = argc > 2 ? atoi(argv[2]) : 1000, T = argc > 3 ? atoi(argv[3]) : 100;
    const double pitch = argc > 4 ? atof(argv[4]) : 0.6;        // lattice pitch in units of the radius: 0.6 -> degree ~7, 0.3 -> ~30 (list overflow)
    const int K = 3, H = 2, NW = mgp_sparse_words(N);
    const int side = (int)ceil(sqrt((double)N));
    std::vector<double> hx((size_t)B * N * 4);
    for (int b = 0; b < B; ++b) for (int i = 0; i < N; ++i) {
        const int s = (int)(((unsigned long long)i * 7919ull + 13ull * b) % (unsigned long long)(side * side));
        int gx = s % side, gy = s / side;
        double jx = 0.01 * ((i * 7 + b) % 13), jy = 0.01 * ((i * 5 + b) % 11);
        hx[((size_t)b * N + i) * 4 + 0] = pitch * gx + jx;
        hx[((size_t)b * N + i) * 4 + 1] = pitch * gy + jy;
        hx[((size_t)b * N + i) * 4 + 2] = 0.02 * ((i * 3) % 17) - 0.16;
        hx[((size_t)b * N + i) * 4 + 3] = 0.02 * ((i * 11) % 19) - 0.18;
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
    double *x[2], *rew; float *wrow, *feat, *scratch, *act, *expert, *image; unsigned long long* bits; unsigned short* nbr;
    const size_t nx = hx.size() * 8, nbits = (size_t)B * H * N * NW * 8, nw = (size_t)B * H * N * 4, nf = (size_t)B * K * N * 8 * 4;
    const size_t nscr = (size_t)4 * B * N * 8 * 4, nact = (size_t)B * 2 * N * 4, nrew = (size_t)T * B * 8, nnbr = (size_t)B * H * N * 16 * 2;
    hipMalloc(&x[0], nx); hipMalloc(&x[1], nx); hipMalloc(&rew, nrew);
    hipMemcpy(x[0], hx.data(), nx, hipMemcpyHostToDevice); hipMemset(x[1], 0, nx); hipMemset(rew, 0, nrew);
    hipMalloc(&bits, nbits); hipMemset(bits, 0, nbits);
    hipMalloc(&wrow, nw); hipMemset(wrow, 0, nw);
    hipMalloc(&feat, nf); hipMemset(feat, 0, nf);
    hipMalloc(&scratch, nscr); hipMemset(scratch, 0, nscr);
    hipMalloc(&act, nact); hipMemset(act, 0, nact);
    hipMalloc(&expert, nact); hipMemset(expert, 0, nact);
    hipMalloc(&nbr, nnbr); hipMemset(nbr, 0, nnbr);
    hipMalloc(&image, mgp_sparse_policy_image_floats(dims, 3, K) * 4);
    if (mgp_sparse_policy_image(W, bb, dims, 3, K, image, nullptr)) { printf("image failed\n"); return 1; }
    MgpFlockParams p = {1.0, 0.01, 10.0, 1.0, 0.1, 10.0, 1.0, 1, 0, 1, 0};
    int rc = mgp_flock_step_cells_nbr(x[0], x[1], nullptr, 2, 1, bits, (long)H * N * NW, wrow, (long)H * N, feat, (long)K * N * 8, nbr, (long)H * N * 16,
                                      nullptr, expert, &p, B, N, nullptr);
    if (rc) { printf("observe rc %d\n", rc); return 1; }
    printf("persistent form covers this shape: %d\n", mgp_sparse_rollout_persistent(dims, 3, K, N, &p));
    int cur = 0, hs = 0;
    auto run = [&](int persist, int steps, double* xa, double* xb, double* rw) {
        setenv("MGP_SP_PERSIST", persist ? "1" : "0", 1);
        return mgp_sparse_rollout(bits, wrow, feat, image, dims, 3, scratch, act, xa, xb, rw, expert, &p, B, K, N, steps, &cur, &hs, nullptr, nbr, nullptr);
    };
    // warm-up on the K-launch form: 10 steps (even: the state is back in x[0])
    if ((rc = run(0, 10, x[0], x[1], nullptr))) { printf("warm-up rc %d\n", rc); return 1; }
    hipDeviceSynchronize();
    const int cur0 = cur, hs0 = hs;
    Buf bufs[] = {{x[0], nx, "x_a"}, {x[1], nx, "x_b"}, {bits, nbits, "bits"}, {wrow, nw, "wrow"}, {feat, nf, "feat"}, {act, nact, "action"},
                  {expert, nact, "expert"}, {rew, nrew, "rewards"}, {nbr, nnbr, "nbr"}};
    const int NB = sizeof(bufs) / sizeof(bufs[0]);
    for (int i = 0; i < NB; ++i) { bufs[i].snap.resize(bufs[i].bytes); hipMemcpy(bufs[i].snap.data(), bufs[i].d, bufs[i].bytes, hipMemcpyDeviceToHost); }
    auto restore = [&]() { for (int i = 0; i < NB; ++i) hipMemcpy(bufs[i].d, bufs[i].snap.data(), bufs[i].bytes, hipMemcpyHostToDevice); cur = cur0; hs = hs0; };
    auto grab = [&](bool first) {
        hipDeviceSynchronize();
        for (int i = 0; i < NB; ++i) { auto& v = first ? bufs[i].a : bufs[i].b; v.resize(bufs[i].bytes); hipMemcpy(v.data(), bufs[i].d, bufs[i].bytes, hipMemcpyDeviceToHost); }
    };
    auto compare = [&](const char* what) {
        int bad = 0;
        for (int i = 0; i < NB; ++i) {
            // x: only the buffer that holds the final state is specified
            if ((i == 0 && (T & 1)) || (i == 1 && !(T & 1))) continue;
            size_t diff = 0, first = 0;
            for (size_t k = 0; k < bufs[i].bytes; ++k) if (bufs[i].a[k] != bufs[i].b[k]) { if (!diff) first = k; ++diff; }
            if (diff) { printf("  %s: %s differs in %zu bytes (first at %zu)\n", what, bufs[i].name, diff, first); ++bad; }
        }
        printf("%s: %s\n", what, bad ? "MISMATCH" : "bit-identical (x, bits, wrow, feat, action, expert, rewards, nbr)");
        return bad;
    };
    int fails = 0;
    restore(); if ((rc = run(0, T, x[0], x[1], rew))) { printf("K-launch rc %d\n", rc); return 1; } grab(true);
    const int curA = cur, hsA = hs;
    restore(); if ((rc = run(1, T, x[0], x[1], rew))) { printf("persistent rc %d\n", rc); return 1; } grab(false);
    printf("status %d, ring slots %d %d vs %d %d\n", mgp_sparse_rollout_status(scratch, B, K, N, nullptr), cur, hs, curA, hsA);
    fails += compare("one call of T steps");
    if (T >= 4) {                                               // chunked: T1 + T2 (T1 odd: the second call starts from x_b)
        const int T1 = (T / 2) | 1, T2 = T - T1;
        restore();
        if ((rc = run(1, T1, x[0], x[1], rew))) { printf("persistent rc %d\n", rc); return 1; }
        if ((rc = run(1, T2, x[1], x[0], rew + (size_t)T1 * B))) { printf("persistent rc %d\n", rc); return 1; }
        grab(false);
        // the final state is in x[1] if T2 is odd ... i.e. where a single call leaves it
        fails += compare("two calls (T1 odd)");
    }
    {   // mean degree and overflow rows of the final network
        std::vector<unsigned short> hn(nnbr / 2); hipMemcpy(hn.data(), nbr, nnbr, hipMemcpyDeviceToHost);
        std::vector<float> hwq(nw / 4); hipMemcpy(hwq.data(), wrow, nw, hipMemcpyDeviceToHost);
        double dsum = 0; long over = 0;
        for (int b = 0; b < B; ++b) for (int i = 0; i < N; ++i) {
            const float w = hwq[((size_t)b * H + hs) * N + i]; dsum += w > 0 ? 1.0 / w : 0;
            over += hn[(((size_t)b * H + hs) * N + i) * 16 + 15] == 0xFFFF;
        }
        printf("mean degree %.2f, rows on the bit-row fallback %ld of %d\n", dsum / ((double)B * N), over, B * N);
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int persist = 0; persist < 2; ++persist) {
        restore();
        const int reps = 5;
        run(persist, T, x[0], x[1], rew); if (T & 1) run(persist, T, x[1], x[0], rew);
        hipDeviceSynchronize();
        hipEventRecord(e0, nullptr);
        for (int r = 0; r < reps; ++r) { run(persist, T, x[0], x[1], rew); if (T & 1) run(persist, T, x[1], x[0], rew); }
        hipEventRecord(e1, nullptr); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const int steps = reps * T * ((T & 1) ? 2 : 1);
        printf("%s form: B=%d N=%d T=%d  %.2f us per step -> %.3e agent-steps/s\n", persist ? "persistent" : "K-launch  ", B, N, T,
               1e3 * ms / steps, (double)B * N * steps / (1e-3 * ms));
    }
    printf("status %d\n", mgp_sparse_rollout_status(scratch, B, K, N, nullptr));
#ifdef MGP_SP_PROFILE
    {
        unsigned long long st[16 * 32];
        hipMemcpyFromSymbol(st, HIP_SYMBOL(mgp_pp_stamps), sizeof(st));
        const char* names[] = {"step start", "gather stage 1 done, stores issued", "arrived (a), ring refill done", "exchange (a) complete", "siblings' rows in LDS",
                               "tail gather done (barrier)", "MLP done, action stored", "arrived (b)", "exchange (b) complete", "cell list built", "row search done, outputs stored",
                               "arrived (c)", "exchange (c) complete", "siblings' weights in LDS", "sim: action requested", "sim: integrated, wave reductions", "sim: block reductions (barrier)",
                               "sim: grid + histogram issued", "sim: scan done", "sim: scattered (barrier)"};
        const int wv[] = {0, 5, 10, 15};
        printf("spp_rollout_kernel, workgroup (tile 1, episode 3), a steady-state step (T - 4): cycles since the step's start, lane 0 of waves 0 5 10 15\n");
        for (int i = 0; i < 20; ++i) { printf("  stamp %2d :", i); for (int w = 0; w < 4; ++w) printf(" %7lld", (long long)(st[wv[w] * 32 + i] - st[0])); printf("  %s\n", names[i]); }
    }
#endif
    return fails ? 2 : 0;
}
