// Flocking simulation step + expert controller (FLOCK-SPEC v1).  gym_flock is not part of the
// reference tree (parity unpinned); the call sites served are env.step (reference gnn_dagger.py:163),
// env.env.controller (gnn_dagger.py:156, gnn_baseline.py:16) and the observation tuple consumed by
// state_with_delay.py:22-35.
//
// All state and pairwise arithmetic is fp64 with the operation order of the spec and NO fused
// multiply-add (this file is compiled with -ffp-contract=off), so the radius test r2 < R^2 -- the only
// discontinuity -- agrees bit-for-bit with the fp64 numpy restatement.  Outputs are emitted in the
// layouts the consumers want: the network matrix as dense fp32 (B,N,N) rows (what Actor / gso_update
// read) and the features already transposed to (B,6,N).
//
// Kernels: flock_integrate (one workgroup per episode: double integrator + velocity-variance reward)
// and flock_pairwise (one wave per agent row: lanes stride over the other agents, wave reductions for the
// degree and the six feature sums, then a second coalesced sweep writes the normalised row).
#include "mgp_common.h"

namespace {

constexpr int FL_THREADS = 256;
constexpr int FL_RPW = 4;                       // agent rows per wave
constexpr int FL_ROWS = FL_RPW * (FL_THREADS / 64);

__device__ __forceinline__ double clipd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

__device__ __forceinline__ double block_sum(double v, double* sh /* [4] */)
{
    v = mgp_wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

// grid: x = b
__global__ __launch_bounds__(FL_THREADS)
void flock_integrate_kernel(double* __restrict__ x, const float* __restrict__ u, long su_agent, long su_axis,
                            double* __restrict__ reward, MgpFlockParams p, int N)
{
    __shared__ double sh[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    double* xb = x + (size_t)b * N * 4;
    double svx = 0.0, svy = 0.0;
    for (int i = tid; i < N; i += FL_THREADS) {
        double px = xb[i * 4 + 0], py = xb[i * 4 + 1], vx = xb[i * 4 + 2], vy = xb[i * 4 + 3];
        if (u != nullptr) {
            double ux = 0.0, uy = 0.0;
            if (i >= p.n_leaders) {
                const float* ub = u + (size_t)b * N * 2 + (size_t)i * su_agent;
                ux = clipd((double)ub[0], -p.max_accel, p.max_accel) * p.action_gain;
                uy = clipd((double)ub[su_axis], -p.max_accel, p.max_accel) * p.action_gain;
            }
            px = (px + vx * p.dt) + ((ux * p.dt) * p.dt) * 0.5;
            py = (py + vy * p.dt) + ((uy * p.dt) * p.dt) * 0.5;
            vx = vx + ux * p.dt;
            vy = vy + uy * p.dt;
            xb[i * 4 + 0] = px; xb[i * 4 + 1] = py; xb[i * 4 + 2] = vx; xb[i * 4 + 3] = vy;
        }
        svx += vx; svy += vy;
    }
    if (reward == nullptr) return;
    const double mx = block_sum(svx, sh) / (double)N;
    const double my = block_sum(svy, sh) / (double)N;
    double dv = 0.0;
    for (int i = tid; i < N; i += FL_THREADS) {
        // each thread re-reads what it wrote itself above
        const double ex = xb[i * 4 + 2] - mx, ey = xb[i * 4 + 3] - my;
        dv += ex * ex + ey * ey;
    }
    const double var = block_sum(dv, sh) / (double)N;
    if (tid == 0) reward[b] = -1.0 * var * p.reward_scale;
}

// grid: x = row tile, y = b.  LDS: px,py,vx,vy [N] doubles.
__global__ __launch_bounds__(FL_THREADS)
void flock_pairwise_kernel(const double* __restrict__ x, float* __restrict__ A, double* __restrict__ A64,
                           float* __restrict__ feat, double* __restrict__ feat64, MgpFlockParams p, int N)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double* spx = sm; double* spy = sm + N; double* svx = sm + 2 * (size_t)N; double* svy = sm + 3 * (size_t)N;
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double* xb = x + (size_t)b * N * 4;
    for (int i = tid; i < N; i += FL_THREADS) {
        spx[i] = xb[i * 4 + 0]; spy[i] = xb[i * 4 + 1]; svx[i] = xb[i * 4 + 2]; svy[i] = xb[i * 4 + 3];
    }
    __syncthreads();
    const double R2 = p.comm_radius2;
    for (int rr = 0; rr < FL_RPW; ++rr) {
        const int i = blockIdx.x * FL_ROWS + wave * FL_RPW + rr;
        if (i >= N) break;
        const double xi = spx[i], yi = spy[i], vxi = svx[i], vyi = svy[i];
        int deg = 0;
        double f0 = 0, f1 = 0, f2 = 0, f3 = 0, f4 = 0, f5 = 0;
        for (int j = lane; j < N; j += 64) {
            const double dx = xi - spx[j], dy = yi - spy[j];
            const double r2 = dx * dx + dy * dy;
            if (j != i && r2 < R2) {
                const double r4 = r2 * r2;
                deg += 1;
                f0 += vxi - svx[j];
                f1 += dx / r4;
                f2 += dx / r2;
                f3 += vyi - svy[j];
                f4 += dy / r4;
                f5 += dy / r2;
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) deg += __shfl_xor(deg, off, 64);
        f0 = mgp_wave_sum(f0); f1 = mgp_wave_sum(f1); f2 = mgp_wave_sum(f2);
        f3 = mgp_wave_sum(f3); f4 = mgp_wave_sum(f4); f5 = mgp_wave_sum(f5);
        if (lane == 0) {
            if (feat != nullptr) {
                float* fb = feat + (size_t)b * 6 * N + i;
                fb[0 * (size_t)N] = (float)f0; fb[1 * (size_t)N] = (float)f1; fb[2 * (size_t)N] = (float)f2;
                fb[3 * (size_t)N] = (float)f3; fb[4 * (size_t)N] = (float)f4; fb[5 * (size_t)N] = (float)f5;
            }
            if (feat64 != nullptr) {
                double* fd = feat64 + ((size_t)b * N + i) * 6;
                fd[0] = f0; fd[1] = f1; fd[2] = f2; fd[3] = f3; fd[4] = f4; fd[5] = f5;
            }
        }
        const double wd = p.mean_pooling ? 1.0 / (double)(deg == 0 ? 1 : deg) : 1.0;
        const float wf = (float)wd;
        for (int j = lane; j < N; j += 64) {
            const double dx = xi - spx[j], dy = yi - spy[j];
            const double r2 = dx * dx + dy * dy;
            const bool nb = (j != i && r2 < R2);
            if (A != nullptr) A[((size_t)b * N + i) * N + j] = nb ? wf : 0.f;
            if (A64 != nullptr) A64[((size_t)b * N + i) * N + j] = nb ? wd : 0.0;
        }
    }
}

// Expert controller.  grid: x = row tile, y = b.
__global__ __launch_bounds__(FL_THREADS)
void flock_controller_kernel(const double* __restrict__ x, float* __restrict__ u, double* __restrict__ u64,
                             MgpFlockParams p, int centralized, int N)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double* spx = sm; double* spy = sm + N; double* svx = sm + 2 * (size_t)N; double* svy = sm + 3 * (size_t)N;
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double* xb = x + (size_t)b * N * 4;
    for (int i = tid; i < N; i += FL_THREADS) {
        spx[i] = xb[i * 4 + 0]; spy[i] = xb[i * 4 + 1]; svx[i] = xb[i * 4 + 2]; svy[i] = xb[i * 4 + 3];
    }
    __syncthreads();
    const double R2 = p.comm_radius2;
    for (int rr = 0; rr < FL_RPW; ++rr) {
        const int i = blockIdx.x * FL_ROWS + wave * FL_RPW + rr;
        if (i >= N) break;
        const double xi = spx[i], yi = spy[i], vxi = svx[i], vyi = svy[i];
        double s0 = 0, s1 = 0, s2 = 0, s3 = 0;      // sum dvx, dvy, gx, gy
        for (int j = lane; j < N; j += 64) {
            if (j == i) continue;
            const double dx = xi - spx[j], dy = yi - spy[j];
            const double r2 = dx * dx + dy * dy;
            const bool nb = r2 < R2;
            if (nb || centralized) {
                s0 += vxi - svx[j];
                s1 += vyi - svy[j];
                if (!(r2 > R2)) {
                    const double r4 = r2 * r2;
                    s2 += -2.0 * (dx / r4) + 2.0 * (dx / r2);
                    s3 += -2.0 * (dy / r4) + 2.0 * (dy / r2);
                }
            }
        }
        s0 = mgp_wave_sum(s0); s1 = mgp_wave_sum(s1); s2 = mgp_wave_sum(s2); s3 = mgp_wave_sum(s3);
        if (lane == 0) {
            const double ux = clipd(-s2 - s0, -p.ctrl_clip, p.ctrl_clip) * p.ctrl_gain;
            const double uy = clipd(-s1 - s3, -p.ctrl_clip, p.ctrl_clip) * p.ctrl_gain;
            if (u != nullptr) { u[((size_t)b * N + i) * 2 + 0] = (float)ux; u[((size_t)b * N + i) * 2 + 1] = (float)uy; }
            if (u64 != nullptr) { u64[((size_t)b * N + i) * 2 + 0] = ux; u64[((size_t)b * N + i) * 2 + 1] = uy; }
        }
    }
}

int check_params(const MgpFlockParams* p)
{
    if (p == nullptr) return MGP_EINVAL;
    if (!(p->comm_radius2 > 0.0) || !(p->dt > 0.0)) return MGP_EINVAL;
    if (p->n_leaders < 0) return MGP_EINVAL;
    return MGP_OK;
}

}  // namespace

extern "C" int mgp_flock_step(double* x, const float* u, long su_agent, long su_axis,
                              float* A, double* A64, float* feat, double* feat64,
                              double* reward, const MgpFlockParams* p, int B, int N, void* stream)
{
    if (B < 0 || N <= 0) return MGP_EINVAL;
    int rc = check_params(p);
    if (rc != MGP_OK) return rc;
    if (B == 0) return MGP_OK;
    if (B > 65535 || N > 4096) return MGP_EUNSUPPORTED;
    MGP_CHECK_PTR8(x);
    hipStream_t st = static_cast<hipStream_t>(stream);
    mgp_clear_error();
    if (u != nullptr || reward != nullptr) {
        hipLaunchKernelGGL(flock_integrate_kernel, dim3(B), dim3(FL_THREADS), 0, st, x, u, su_agent, su_axis,
                           reward, *p, N);
        rc = mgp_launch_status();
        if (rc != MGP_OK) return rc;
    }
    if (A != nullptr || A64 != nullptr || feat != nullptr || feat64 != nullptr) {
        const size_t lds = (size_t)4 * N * sizeof(double);
        if (lds > 48 * 1024 &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(flock_pairwise_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return MGP_ELAUNCH;
        dim3 grid(mgp_ceil_div(N, FL_ROWS), B);
        hipLaunchKernelGGL(flock_pairwise_kernel, grid, dim3(FL_THREADS), lds, st, x, A, A64, feat, feat64, *p, N);
        rc = mgp_launch_status();
    }
    return rc;
}

extern "C" int mgp_flock_controller(const double* x, float* u, double* u64, const MgpFlockParams* p,
                                    int centralized, int B, int N, void* stream)
{
    if (B < 0 || N <= 0) return MGP_EINVAL;
    int rc = check_params(p);
    if (rc != MGP_OK) return rc;
    if (B == 0) return MGP_OK;
    if (B > 65535 || N > 4096) return MGP_EUNSUPPORTED;
    MGP_CHECK_PTR8(x);
    if (u == nullptr && u64 == nullptr) return MGP_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    mgp_clear_error();
    const size_t lds = (size_t)4 * N * sizeof(double);
    if (lds > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(flock_controller_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return MGP_ELAUNCH;
    dim3 grid(mgp_ceil_div(N, FL_ROWS), B);
    hipLaunchKernelGGL(flock_controller_kernel, grid, dim3(FL_THREADS), lds, st, x, u, u64, *p, centralized ? 1 : 0, N);
    return mgp_launch_status();
}
