// Launch anatomy of the episode-resident rollout kernel: what a launch costs BEYOND its steps, and where.
// Every launch starts from the SAME saved state (x, delay line, carry restored before each launch), so launches of different
// length are comparable step for step (the r03 sweep let the flock evolve across launches: the "fixed cost" it derived grew
// with T because early, dense states are slower -- not because the launch was).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o scratch/ro_launch tools/harness/ro_launch_prof.hip
//   RO_STATE=/tmp/ro_state5.bin scratch/ro_launch 256 100 3 "1 2 3 5 10 20 40" 30
// Per launch length T: kernel time from the launch's own begin / end events (hipExtLaunchKernel), and from the 100 MHz wall clock
// every workgroup stamps (RO_WALL): dispatch ramp (first to last workgroup start), entry, the first three steps, exit, end skew.
#define MGP_RO_PROFILE 1
#include "../../multiagent_gnn_policies_amd/csrc/rollout.hip"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <sstream>
#include <string>
#include <vector>
thread_local int mgp_tls_hip_error = 0;
thread_local void* mgp_tls_launch_events[2] = {nullptr, nullptr};
extern "C" int mgp_rollout_wide_supported_(const int*, int, int, int) { return 0; }
extern "C" int mgp_rollout_wide_steps_ex_(double*, float*, float*, const float* const*, const float* const*, const int*, int, float*,
                                          double*, const MgpFlockParams*, int, int, int, int, const float*, void*, int, void*) { return MGP_EUNSUPPORTED; }
extern "C" int mgp_rollout_wide_collect_(double*, float*, float*, const float* const*, const float* const*, const int*, int, double*,
                                         const MgpFlockParams*, int, int, int, int, const float*, void*, int, const MgpCollect*, void*) { return MGP_EUNSUPPORTED; }
extern "C" long mgp_rollout_wide_image_floats_(const int*, int, int, int) { return 0; }
extern "C" int mgp_rollout_wide_image_(const float* const*, const float* const*, const int*, int, int, int, float*, void*) { return MGP_EUNSUPPORTED; }

__global__ __launch_bounds__(1024) void empty_kernel(long long* out)
{
    extern __shared__ unsigned char sm[];
    if (threadIdx.x == 0) { sm[0] = 1; out[blockIdx.x] = wall_clock64(); }
}

static double med(std::vector<double> v) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[v.size() / 2]; }

int main(int argc, char** argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 256, N = argc > 2 ? atoi(argv[2]) : 100, K = argc > 3 ? atoi(argv[3]) : 3;
    std::vector<int> Ts;
    { std::istringstream is(argc > 4 ? argv[4] : "1 2 3 5 10 20 40"); int t; while (is >> t) Ts.push_back(t); }
    const int R = argc > 5 ? atoi(argv[5]) : 30;
    const int Tmax = *std::max_element(Ts.begin(), Ts.end());
    std::vector<double> hx((size_t)B * N * 4);
    for (int b = 0; b < B; ++b) for (int i = 0; i < N; ++i) {
        const int gx = i % 10, gy = i / 10;
        hx[((size_t)b * N + i) * 4 + 0] = 0.6 * gx + 0.01 * ((i * 7 + b) % 13);
        hx[((size_t)b * N + i) * 4 + 1] = 0.6 * gy + 0.01 * ((i * 5 + b) % 11);
        hx[((size_t)b * N + i) * 4 + 2] = 0.1 * ((i * 3) % 17) - 0.8;
        hx[((size_t)b * N + i) * 4 + 3] = 0.1 * ((i * 11) % 19) - 0.9;
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
    }
    double *x, *x0, *rew; float *G, *Xd, *Xd0, *act;
    const size_t xb = hx.size() * 8, gb = (size_t)B * K * N * N * 4, xdb = (size_t)B * K * 6 * N * 4;
    hipMalloc(&x, xb); hipMalloc(&x0, xb); hipMalloc(&rew, (size_t)B * Tmax * 8);
    hipMalloc(&G, gb); hipMalloc(&Xd, xdb); hipMalloc(&Xd0, xdb); hipMalloc(&act, (size_t)B * 2 * N * 4);
    hipMemset(G, 0, gb); hipMemset(Xd, 0, xdb);
    std::vector<float> hg((size_t)B * K * N * N), hxd((size_t)B * K * 6 * N);
    if (const char* dump = getenv("RO_STATE")) {
        FILE* f = fopen(dump, "rb");
        if (!f) { printf("cannot open %s\n", dump); return 1; }
        size_t ok = fread(hx.data(), 8, hx.size(), f) + fread(hg.data(), 4, hg.size(), f) + fread(hxd.data(), 4, hxd.size(), f);
        for (int l = 0; l < 3; ++l) { ok += fread(hw[l].data(), 4, hw[l].size(), f); ok += fread(hb[l].data(), 4, hb[l].size(), f); }
        fclose(f);
        hipMemcpy(G, hg.data(), hg.size() * 4, hipMemcpyHostToDevice); hipMemcpy(Xd, hxd.data(), hxd.size() * 4, hipMemcpyHostToDevice);
        printf("state from %s (%zu values)\n", dump, ok);
    }
    hipMemcpy(x, hx.data(), xb, hipMemcpyHostToDevice);
    for (int l = 0; l < 3; ++l) { hipMemcpy(W[l], hw[l].data(), hw[l].size() * 4, hipMemcpyHostToDevice); hipMemcpy(bb[l], hb[l].data(), hb[l].size() * 4, hipMemcpyHostToDevice); }
    MgpFlockParams p = {1.0, 0.01, 10.0, 1.0, 0.1, 10.0, 1.0, 1, 0, 1, 0};
    float* image = nullptr; void *carry = nullptr, *carry0 = nullptr;
    const size_t cb = (size_t)B * mgp_rollout_carry_bytes(K, N);
    hipMalloc(&image, mgp_rollout_image_floats(dims, 3, K, N) * 4);
    hipMalloc(&carry, cb); hipMalloc(&carry0, cb); hipMemset(carry, 0, cb);
    if (mgp_rollout_image(W, bb, dims, 3, K, N, image, nullptr)) { printf("image failed\n"); return 1; }
    const int fl = MGP_RO_ENTER_CARRY | MGP_RO_EXIT_CARRY | MGP_RO_SKIP_DENSE;
    // warm-up: K steps from the dense state (the carry then holds the full history), the saved state is what every launch starts from
    int rc = mgp_rollout_steps_ex(x, G, Xd, W, bb, dims, 3, act, rew, &p, B, K, N, K + 2, image, carry, MGP_RO_EXIT_CARRY | MGP_RO_SKIP_DENSE, nullptr);
    if (rc) { printf("rc %d\n", rc); return 1; }
    hipDeviceSynchronize();
    hipMemcpy(x0, x, xb, hipMemcpyDeviceToDevice); hipMemcpy(Xd0, Xd, xdb, hipMemcpyDeviceToDevice); hipMemcpy(carry0, carry, cb, hipMemcpyDeviceToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    // an empty kernel of the same shape (256 x 1024 threads, the same dynamic LDS): what the dispatch alone costs
    {
        long long* ob; hipMalloc(&ob, B * 8);
        const int lds = 76 * 1024;
        hipFuncSetAttribute(reinterpret_cast<const void*>(empty_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        std::vector<double> us;
        for (int r = 0; r < R + 3; ++r) {
            hipExtLaunchKernelGGL(empty_kernel, dim3(B), dim3(1024), lds, nullptr, e0, e1, 0, ob);
            hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (r >= 3) us.push_back(1e3 * ms);
        }
        std::vector<long long> ho(B);
        hipMemcpy(ho.data(), ob, B * 8, hipMemcpyDeviceToHost);
        printf("empty kernel, %d x 1024 threads, %d KB LDS: %.2f us (event begin->end, median of %d), workgroup start spread %.2f us\n",
               B, lds / 1024, med(us), R, 0.01 * (double)(*std::max_element(ho.begin(), ho.end()) - *std::min_element(ho.begin(), ho.end())));
    }
    printf("T | kernel us (events) | wall first-start..last-end | start spread | entry | step0 step1 step2 | steps 3.. avg | exit | end spread   (us, medians over %d launches / over workgroups)\n", R);
    std::vector<long long> hwall((size_t)4096 * 8);
    for (int T : Ts) {
        std::vector<double> ker, wall, ramp, entry, s0, s1, s2, rest, ex, endsp;
        for (int r = 0; r < R + 3; ++r) {
            hipMemcpyAsync(x, x0, xb, hipMemcpyDeviceToDevice, nullptr); hipMemcpyAsync(Xd, Xd0, xdb, hipMemcpyDeviceToDevice, nullptr);
            hipMemcpyAsync(carry, carry0, cb, hipMemcpyDeviceToDevice, nullptr);
            hipDeviceSynchronize();
            mgp_tls_launch_events[0] = e0; mgp_tls_launch_events[1] = e1;
            { static std::vector<unsigned int> z(4096 * 4, 0u); hipMemcpyToSymbol(HIP_SYMBOL(mgp_ro_vstat), z.data(), z.size() * 4); }
            rc = mgp_rollout_steps_ex(x, G, Xd, W, bb, dims, 3, act, rew, &p, B, K, N, T, image, carry, fl, nullptr);
            if (rc) { printf("rc %d\n", rc); return 1; }
            hipDeviceSynchronize();
            if (r < 3) continue;
            float ms; hipEventElapsedTime(&ms, e0, e1);
            ker.push_back(1e3 * ms);
            hipMemcpyFromSymbol(hwall.data(), HIP_SYMBOL(mgp_ro_wall), hwall.size() * 8);
            long long t0 = hwall[0], t1 = hwall[6], ts = hwall[0], te = hwall[6];
            std::vector<double> en, a0, a1, a2, ar, xx;
            for (int b = 0; b < B; ++b) {
                const long long* w = &hwall[(size_t)b * 8];
                t0 = std::min(t0, w[0]); ts = std::max(ts, w[0]); t1 = std::max(t1, w[6]); te = std::min(te, w[6]);
                en.push_back(0.01 * (w[1] - w[0]));
                a0.push_back(0.01 * (w[2] - w[1]));
                if (T > 1) a1.push_back(0.01 * (w[3] - w[2]));
                if (T > 2) a2.push_back(0.01 * (w[4] - w[3]));
                if (T > 3) ar.push_back(0.01 * (w[5] - w[4]) / (T - 3));
                xx.push_back(0.01 * (w[6] - w[5]));
            }
            wall.push_back(0.01 * (t1 - t0)); ramp.push_back(0.01 * (ts - t0)); endsp.push_back(0.01 * (t1 - te));
            entry.push_back(med(en)); s0.push_back(med(a0)); s1.push_back(med(a1)); s2.push_back(med(a2)); rest.push_back(med(ar)); ex.push_back(med(xx));
        }
        printf("%3d | %8.2f | %8.2f | %6.2f | %6.2f | %6.2f %6.2f %6.2f | %6.2f | %6.2f | %6.2f\n", T, med(ker), med(wall), med(ramp), med(entry),
               med(s0), med(s1), med(s2), med(rest), med(ex), med(endsp));
        if (const char* dump = getenv("RO_WG_DUMP")) {
            // per workgroup (= episode) of the LAST launch of this length: begin-to-end duration, next to the episode's mean and
            // largest degree at the state the harness was handed -- which episodes set a launch's duration
            if (T == (getenv("RO_WG_DUMP_T") ? atoi(getenv("RO_WG_DUMP_T")) : 20)) {
                FILE* f = fopen(dump, "w");
                if (!f) { printf("cannot write %s\n", dump); return 1; }
                std::vector<unsigned int> vs(4096 * 4);
                hipMemcpyFromSymbol(vs.data(), HIP_SYMBOL(mgp_ro_vstat), vs.size() * 4);
                fprintf(f, "# episode  duration_us (T = %d)  mean_degree  max_degree  first_entry_us  S1 modes decided inside the launch: cheap rebuild full\n", T);
                for (int b = 0; b < B; ++b) {
                    int dsum = 0, dmax = 0;
                    for (int i = 0; i < N; ++i) {
                        int d = 0;
                        for (int j = 0; j < N; ++j) {
                            if (j == i) continue;
                            const double dx = hx[((size_t)b * N + i) * 4] - hx[((size_t)b * N + j) * 4], dy = hx[((size_t)b * N + i) * 4 + 1] - hx[((size_t)b * N + j) * 4 + 1];
                            d += dx * dx + dy * dy < 1.0;
                        }
                        dsum += d; dmax = d > dmax ? d : dmax;
                    }
                    const long long* w = &hwall[(size_t)b * 8];
                    fprintf(f, "%d %.2f %.2f %d %.2f  %u %u %u\n", b, 0.01 * (w[6] - w[0]), (double)dsum / N, dmax, 0.01 * (w[7] - w[0]), vs[b * 4], vs[b * 4 + 1], vs[b * 4 + 2]);
                }
                fclose(f);
            }
        }
    }
    return 0;
}
