// Flocking simulation step + expert controller (FLOCK-SPEC v1).  gym_flock is not part of the
// reference tree (parity unpinned); the call sites served are env.step (reference gnn_dagger.py:163),
// env.env.controller (gnn_dagger.py:156, gnn_baseline.py:16) and the observation tuple consumed by
// state_with_delay.py:22-35.
//
// All state and pairwise arithmetic is fp64 with the operation order of the spec and NO fused
// multiply-add (this file is compiled with -ffp-contract=off), so the radius test r2 < R^2 -- the only
// discontinuity -- agrees bit-for-bit with the fp64 numpy restatement.  Outputs are emitted in the
// layouts the consumers want: the network matrix as dense fp32 (B,N,N) rows (what Actor / gso_update
// read), the features already transposed to (B,6,N), the expert action as (B,N,2).
//
// One kernel per step.  A workgroup (1024 threads) owns up to 128 agent rows of one episode: a THREAD owns one
// row i and one eighth of the j range (8 threads per row), walking j with the other agents' state broadcast from LDS -- no
// cross-lane reductions, one fp64 division per pair (q = 1/r2).  The eight pieces meet in LDS (fixed order), then the
// workgroup writes its rows of the network matrix in one flat, coalesced sweep.  For N <= 128 the double
// integrator and the velocity-variance reward are fused in front (one workgroup == one episode);
// larger N runs flock_integrate first.
#include "mgp_common.h"

namespace {

constexpr int FL_THREADS = 1024;
constexpr int FL_WAVES = FL_THREADS / 64;
constexpr int FL_ROWS = 128;                    // agent rows per workgroup
constexpr int FL_SPLIT = FL_THREADS / FL_ROWS;  // threads per row (the j range is cut into FL_SPLIT pieces)

__device__ __forceinline__ double clipd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

__device__ __forceinline__ double block_sum(double v, double* sh /* [FL_WAVES] */)
{
    v = mgp_wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < FL_WAVES; ++w) t += sh[w];
    return t;
}

struct FlockOut {
    float* A; double* A64; float* feat; double* feat64; double* reward;
    float* expert; double* expert64; int centralized;
    long sAb, sFb;          // batch strides (elements) of A and feat: lets the sim write straight into delay_gso[:,1] / delay_state[:,0]
};

// integrate one agent in registers (spec section 1)
__device__ __forceinline__ void integrate_one(double& px, double& py, double& vx, double& vy, const float* ub,
                                              long su_axis, bool leader, const MgpFlockParams& p)
{
    double ux = 0.0, uy = 0.0;
    if (!leader) {
        ux = clipd((double)ub[0], -p.max_accel, p.max_accel) * p.action_gain;
        uy = clipd((double)ub[su_axis], -p.max_accel, p.max_accel) * p.action_gain;
    }
    px = (px + vx * p.dt) + ((ux * p.dt) * p.dt) * 0.5;
    py = (py + vy * p.dt) + ((uy * p.dt) * p.dt) * 0.5;
    vx = vx + ux * p.dt;
    vy = vy + uy * p.dt;
}

// grid: x = b (N > 128 only)
__global__ __launch_bounds__(FL_THREADS)
void flock_integrate_kernel(double* __restrict__ x, const float* __restrict__ u, long su_agent, long su_axis,
                            MgpFlockParams p, int N)
{
    const int b = blockIdx.x;
    double* xb = x + (size_t)b * N * 4;
    for (int i = threadIdx.x; i < N; i += FL_THREADS) {
        double px = xb[i * 4 + 0], py = xb[i * 4 + 1], vx = xb[i * 4 + 2], vy = xb[i * 4 + 3];
        integrate_one(px, py, vx, vy, u + (size_t)b * N * 2 + (size_t)i * su_agent, su_axis, i < p.n_leaders, p);
        xb[i * 4 + 0] = px; xb[i * 4 + 1] = py; xb[i * 4 + 2] = vx; xb[i * 4 + 3] = vy;
    }
}

// grid: x = row chunk, y = b.  LDS (doubles): px,py,vx,vy [N] | part [FL_SPLIT][FL_ROWS][8] | wrow [FL_ROWS]
template <bool FUSE_INTEGRATE>
__global__ __launch_bounds__(FL_THREADS)
void flock_step_kernel(double* __restrict__ x, const float* __restrict__ u, long su_agent, long su_axis,
                       FlockOut o, MgpFlockParams p, int N)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    __shared__ double sh[FL_WAVES];
    double* spx = sm; double* spy = sm + N; double* svx = sm + 2 * (size_t)N; double* svy = sm + 3 * (size_t)N;
    double* part = sm + 4 * (size_t)N;                     // [FL_SPLIT][FL_ROWS][8]: deg,f0..f5 of each j piece
    double* wrow = part + FL_SPLIT * FL_ROWS * 8;                     // [FL_ROWS] network weight of the row (fp64)
    const int b = blockIdx.y, tid = threadIdx.x;
    const int i0 = blockIdx.x * FL_ROWS;
    const int rows = min(FL_ROWS, N - i0);
    double* xb = x + (size_t)b * N * 4;

    // ---- load (and, when fused, integrate) every agent of the episode into LDS
    double sum_vx = 0.0, sum_vy = 0.0;
    for (int i = tid; i < N; i += FL_THREADS) {
        double px = xb[i * 4 + 0], py = xb[i * 4 + 1], vx = xb[i * 4 + 2], vy = xb[i * 4 + 3];
        if (FUSE_INTEGRATE && u != nullptr) {
            integrate_one(px, py, vx, vy, u + (size_t)b * N * 2 + (size_t)i * su_agent, su_axis, i < p.n_leaders, p);
            xb[i * 4 + 0] = px; xb[i * 4 + 1] = py; xb[i * 4 + 2] = vx; xb[i * 4 + 3] = vy;
        }
        spx[i] = px; spy[i] = py; svx[i] = vx; svy[i] = vy;
        sum_vx += vx; sum_vy += vy;
    }
    __syncthreads();
    // ---- episode-level sums (reward, centralised controller): every workgroup of the episode recomputes them
    double tot_vx = 0.0, tot_vy = 0.0;
    if (o.reward != nullptr || (o.centralized && (o.expert != nullptr || o.expert64 != nullptr))) {
        tot_vx = block_sum(sum_vx, sh);
        tot_vy = block_sum(sum_vy, sh);
        if (o.reward != nullptr && blockIdx.x == 0) {
            const double mx = tot_vx / (double)N, my = tot_vy / (double)N;
            double dv = 0.0;
            for (int i = tid; i < N; i += FL_THREADS) {
                const double ex = svx[i] - mx, ey = svy[i] - my;
                dv += ex * ex + ey * ey;
            }
            const double var = block_sum(dv, sh) / (double)N;
            if (tid == 0) o.reward[b] = -1.0 * var * p.reward_scale;
        }
    }
    // ---- pairwise pass: thread = (row, j-half)
    const int rl = tid % FL_ROWS, half = tid / FL_ROWS;
    const int i = i0 + rl;
    const double R2 = p.comm_radius2;
    double deg = 0.0, f0 = 0, f1 = 0, f2 = 0, f3 = 0, f4 = 0, f5 = 0;
    if (rl < rows) {
        const double xi = spx[i], yi = spy[i], vxi = svx[i], vyi = svy[i];
        const int jh = (N + FL_SPLIT - 1) / FL_SPLIT;
        const int j0 = half * jh, j1 = min(N, j0 + jh);
        for (int j = j0; j < j1; ++j) {
            const double dx = xi - spx[j], dy = yi - spy[j];
            const double r2 = dx * dx + dy * dy;
            if (j != i && r2 < R2) {
                const double q = 1.0 / r2;
                const double qq = q * q;
                deg += 1.0;
                f0 += vxi - svx[j];
                f1 += dx * qq;
                f2 += dx * q;
                f3 += vyi - svy[j];
                f4 += dy * qq;
                f5 += dy * q;
            }
        }
        if (half > 0) {
            double* pr = part + ((size_t)half * FL_ROWS + rl) * 8;
            pr[0] = deg; pr[1] = f0; pr[2] = f1; pr[3] = f2; pr[4] = f3; pr[5] = f4; pr[6] = f5;
        }
    }
    __syncthreads();
    if (half == 0 && rl < rows) {
#pragma unroll
        for (int h = 1; h < FL_SPLIT; ++h) {                 // ascending j pieces: deterministic
            const double* pr = part + ((size_t)h * FL_ROWS + rl) * 8;
            deg += pr[0]; f0 += pr[1]; f1 += pr[2]; f2 += pr[3]; f3 += pr[4]; f4 += pr[5]; f5 += pr[6];
        }
        wrow[rl] = p.mean_pooling ? 1.0 / (deg == 0.0 ? 1.0 : deg) : 1.0;
        if (o.feat != nullptr) {
            float* fb = o.feat + (size_t)b * o.sFb + i;
            fb[0 * (size_t)N] = (float)f0; fb[1 * (size_t)N] = (float)f1; fb[2 * (size_t)N] = (float)f2;
            fb[3 * (size_t)N] = (float)f3; fb[4 * (size_t)N] = (float)f4; fb[5 * (size_t)N] = (float)f5;
        }
        if (o.feat64 != nullptr) {
            double* fd = o.feat64 + ((size_t)b * N + i) * 6;
            fd[0] = f0; fd[1] = f1; fd[2] = f2; fd[3] = f3; fd[4] = f4; fd[5] = f5;
        }
        if (o.expert != nullptr || o.expert64 != nullptr) {
            double tvx = f0, tvy = f3;
            if (o.centralized) {
                tvx = (double)N * svx[i] - tot_vx;
                tvy = (double)N * svy[i] - tot_vy;
            }
            const double ux = clipd(-tvx - (2.0 * f2 - 2.0 * f1), -p.ctrl_clip, p.ctrl_clip) * p.ctrl_gain;
            const double uy = clipd(-tvy - (2.0 * f5 - 2.0 * f4), -p.ctrl_clip, p.ctrl_clip) * p.ctrl_gain;
            if (o.expert != nullptr) {
                o.expert[((size_t)b * N + i) * 2 + 0] = (float)ux; o.expert[((size_t)b * N + i) * 2 + 1] = (float)uy;
            }
            if (o.expert64 != nullptr) {
                o.expert64[((size_t)b * N + i) * 2 + 0] = ux; o.expert64[((size_t)b * N + i) * 2 + 1] = uy;
            }
        }
    }
    if (o.A == nullptr && o.A64 == nullptr) return;
    __syncthreads();
    // ---- network rows i0..i0+rows-1: one flat coalesced sweep, membership recomputed from LDS (same fp64 ops)
    const size_t base = ((size_t)b * N + i0) * N;
    const size_t baseA = (size_t)b * o.sAb + (size_t)i0 * N;
    int ri = tid / N, j = tid - ri * N;                     // (row, col) of flat index tid
    const int dri = FL_THREADS / N, dj = FL_THREADS - dri * N;
    for (int idx = tid; idx < rows * N; idx += FL_THREADS) {
        const int gi = i0 + ri;
        const double dx = spx[gi] - spx[j], dy = spy[gi] - spy[j];
        const double r2 = dx * dx + dy * dy;
        const bool nb = (j != gi) && (r2 < R2);
        const double w = nb ? wrow[ri] : 0.0;
        if (o.A != nullptr) o.A[baseA + idx] = (float)w;
        if (o.A64 != nullptr) o.A64[base + idx] = w;
        ri += dri; j += dj;
        if (j >= N) { j -= N; ri += 1; }
    }
}

int check_params(const MgpFlockParams* p)
{
    if (p == nullptr) return MGP_EINVAL;
    if (!(p->comm_radius2 > 0.0) || !(p->dt > 0.0)) return MGP_EINVAL;
    if (p->n_leaders < 0) return MGP_EINVAL;
    return MGP_OK;
}

int launch_flock(double* x, const float* u, long su_agent, long su_axis, const FlockOut& o,
                 const MgpFlockParams* p, int B, int N, hipStream_t st)
{
    mgp_clear_error();
    const bool fuse = N <= FL_ROWS;
    int rc = MGP_OK;
    if (!fuse && u != nullptr) {
        hipLaunchKernelGGL(flock_integrate_kernel, dim3(B), dim3(FL_THREADS), 0, st, x, u, su_agent, su_axis, *p, N);
        rc = mgp_launch_status();
        if (rc != MGP_OK) return rc;
    }
    const size_t lds = ((size_t)4 * N + FL_SPLIT * FL_ROWS * 8 + FL_ROWS) * sizeof(double);
    dim3 grid(mgp_ceil_div(N, FL_ROWS), B);
    if (fuse) {
        if (lds > 48 * 1024 &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(flock_step_kernel<true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return MGP_ELAUNCH;
        hipLaunchKernelGGL((flock_step_kernel<true>), grid, dim3(FL_THREADS), lds, st, x, u, su_agent, su_axis, o, *p, N);
    } else {
        if (lds > 48 * 1024 &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(flock_step_kernel<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return MGP_ELAUNCH;
        hipLaunchKernelGGL((flock_step_kernel<false>), grid, dim3(FL_THREADS), lds, st, x, u, su_agent, su_axis, o, *p, N);
    }
    return mgp_launch_status();
}

}  // namespace

extern "C" int mgp_flock_step(double* x, const float* u, long su_agent, long su_axis,
                              float* A, double* A64, float* feat, double* feat64,
                              double* reward, float* expert, long sAb, long sFb,
                              const MgpFlockParams* p, int B, int N, void* stream)
{
    if (B < 0 || N <= 0) return MGP_EINVAL;
    int rc = check_params(p);
    if (rc != MGP_OK) return rc;
    if (B == 0) return MGP_OK;
    if (B > 65535 || N > 4096) return MGP_EUNSUPPORTED;
    MGP_CHECK_PTR8(x);
    if (sAb < 0 || sFb < 0) return MGP_EINVAL;
    FlockOut o = {A, A64, feat, feat64, reward, expert, nullptr, 0, sAb ? sAb : (long)N * N, sFb ? sFb : 6L * N};
    return launch_flock(x, u, su_agent, su_axis, o, p, B, N, static_cast<hipStream_t>(stream));
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
    FlockOut o = {nullptr, nullptr, nullptr, nullptr, nullptr, u, u64, centralized ? 1 : 0, 0, 0};
    // no action => the state is only read
    return launch_flock(const_cast<double*>(x), nullptr, 2, 1, o, p, B, N, static_cast<hipStream_t>(stream));
}
