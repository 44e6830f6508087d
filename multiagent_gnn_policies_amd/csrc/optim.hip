// DAGGER update helpers (reference gnn_dagger.py:91-93): F.mse_loss + its gradient, and torch.optim.Adam
// (defaults: no amsgrad, no weight decay) on one flat fp32 parameter buffer.
#include <math.h>
#include "mgp_common.h"

namespace {

constexpr int OP_THREADS = 1024;

// single workgroup: deterministic fixed-order reduction (n is B*nA*N, a few thousand to ~1e5)
__global__ __launch_bounds__(OP_THREADS)
void mse_grad_kernel(const float* __restrict__ pred, const float* __restrict__ target, float* __restrict__ dPred,
                     float* __restrict__ loss, long n)
{
    __shared__ double sh[OP_THREADS / 64];
    const float scale = 2.0f / (float)n;
    double s = 0.0;
    for (long i = threadIdx.x; i < n; i += OP_THREADS) {
        const float d = pred[i] - target[i];
        if (dPred != nullptr) dPred[i] = scale * d;
        s += (double)d * (double)d;
    }
    s = mgp_wave_sum(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0 && loss != nullptr) {
        double t = 0.0;
        for (int w = 0; w < OP_THREADS / 64; ++w) t += sh[w];
        loss[0] = (float)(t / (double)n);
    }
}

__global__ __launch_bounds__(256)
void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                 long n, float one_m_b1, float b2, float one_m_b2, float step_size, float bc2_sqrt, float eps)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    mgp_adam_elem(p[i], m[i], v[i], g[i], one_m_b1, b2, one_m_b2, step_size, bc2_sqrt, eps);
}

// Adam with the step count resident on the device (HIP-graph replayable: no per-step host scalars).  Every thread
// derives the bias corrections from *step_dev + 1 in double precision; step_inc_kernel advances the counter after.
__global__ __launch_bounds__(256)
void adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                     long n, float lr, float b1, float b2, float eps, const int* __restrict__ step_dev)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double step = (double)(*step_dev + 1);
    const double bc1 = 1.0 - pow((double)b1, step);
    const double bc2 = 1.0 - pow((double)b2, step);
    const float step_size = (float)((double)lr / bc1);
    const float bc2_sqrt = (float)sqrt(bc2);
    const float one_m_b1 = (float)(1.0 - (double)b1), one_m_b2 = (float)(1.0 - (double)b2);
    mgp_adam_elem(p[i], m[i], v[i], g[i], one_m_b1, b2, one_m_b2, step_size, bc2_sqrt, eps);
}

__global__ void step_inc_kernel(int* step_dev) { *step_dev += 1; }

// bookkeeping of one update of an indexed round (mgp_train_step_indexed does this inside its reduce kernel): file the loss
// at the cursor, advance the cursor
__global__ void file_loss_kernel(const float* loss, float* loss_hist, int hist_cap, int* cursor)
{
    const int c = *cursor;
    if (loss != nullptr && loss_hist != nullptr) loss_hist[c % hist_cap] = *loss;
    *cursor = c + 1;
}

// Small parameter sets (the reference Actor has 1,730): ONE workgroup applies the step and advances the device-side
// step counter itself -- one graph node instead of two, and the bias corrections (two fp64 pow) are computed once, not
// once per thread.  Same arithmetic per element as adam_dev_kernel.
__global__ __launch_bounds__(1024)
void adam_dev_onewg_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                           float* __restrict__ v, long n, float lr, float b1, float b2, float eps, int* __restrict__ step_dev)
{
    __shared__ float sh[2];
    __shared__ int shs;
    if (threadIdx.x == 0) {
        const int s0 = *step_dev;
        const double step = (double)(s0 + 1);
        const double bc1 = 1.0 - pow((double)b1, step);
        const double bc2 = 1.0 - pow((double)b2, step);
        sh[0] = (float)((double)lr / bc1);
        sh[1] = (float)sqrt(bc2);
        shs = s0;
    }
    __syncthreads();
    const float step_size = sh[0], bc2_sqrt = sh[1];
    const float one_m_b1 = (float)(1.0 - (double)b1), one_m_b2 = (float)(1.0 - (double)b2);
    for (long i = threadIdx.x; i < n; i += 1024)
        mgp_adam_elem(p[i], m[i], v[i], g[i], one_m_b1, b2, one_m_b2, step_size, bc2_sqrt, eps);
    if (threadIdx.x == 0) *step_dev = shs + 1;
}

}  // namespace

extern "C" int mgp_adam_step_dev(float* param, const float* grad, float* m, float* v, long n,
                                 float lr, float beta1, float beta2, float eps, int* step_dev, void* stream)
{
    if (n <= 0) return MGP_EINVAL;
    MGP_CHECK_PTR(param); MGP_CHECK_PTR(grad); MGP_CHECK_PTR(m); MGP_CHECK_PTR(v); MGP_CHECK_PTR(step_dev);
    mgp_clear_error();
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (n <= 32768) {
        hipLaunchKernelGGL(adam_dev_onewg_kernel, dim3(1), dim3(1024), 0, st, param, grad, m, v, n, lr, beta1, beta2, eps,
                           step_dev);
        return mgp_launch_status();
    }
    hipLaunchKernelGGL(adam_dev_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, param, grad, m, v, n,
                       lr, beta1, beta2, eps, step_dev);
    hipLaunchKernelGGL(step_inc_kernel, dim3(1), dim3(1), 0, st, step_dev);
    return mgp_launch_status();
}

extern "C" int mgp_mse_grad(const float* pred, const float* target, float* dPred, float* loss, long n, void* stream)
{
    if (n <= 0) return MGP_EINVAL;
    MGP_CHECK_PTR(pred); MGP_CHECK_PTR(target);
    if (dPred == nullptr && loss == nullptr) return MGP_EINVAL;
    mgp_clear_error();
    hipLaunchKernelGGL(mse_grad_kernel, dim3(1), dim3(OP_THREADS), 0, static_cast<hipStream_t>(stream),
                       pred, target, dPred, loss, n);
    return mgp_launch_status();
}

extern "C" int mgp_adam_step(float* param, const float* grad, float* m, float* v, long n,
                             float lr, float beta1, float beta2, float eps, int step, void* stream)
{
    if (n <= 0 || step <= 0) return MGP_EINVAL;
    MGP_CHECK_PTR(param); MGP_CHECK_PTR(grad); MGP_CHECK_PTR(m); MGP_CHECK_PTR(v);
    mgp_clear_error();
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1);
    const float bc2_sqrt = (float)sqrt(bc2);
    const unsigned blocks = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       param, grad, m, v, n, (float)(1.0 - (double)beta1), beta2, (float)(1.0 - (double)beta2),
                       step_size, bc2_sqrt, eps);
    return mgp_launch_status();
}

// mgp_adam_step_dev followed by the bookkeeping of an indexed round of updates: loss_hist[*cursor % hist_cap] = *loss and
// *cursor += 1 (both on the device).  For data-parallel rounds that exchange the gradient with a library collective between
// mgp_train_grads and the step (the one-shot exchange has all of this inside mgp_train_step_p2p).
extern "C" int mgp_adam_step_filed(float* param, const float* grad, float* m, float* v, long n,
                                   float lr, float beta1, float beta2, float eps, int* step_dev,
                                   const float* loss, float* loss_hist, int hist_cap, int* cursor, void* stream)
{
    if (cursor == nullptr || (loss_hist != nullptr && hist_cap <= 0)) return MGP_EINVAL;
    if (reinterpret_cast<uintptr_t>(cursor) & 3u) return MGP_EALIGN;
    const int rc = mgp_adam_step_dev(param, grad, m, v, n, lr, beta1, beta2, eps, step_dev, stream);
    if (rc != MGP_OK) return rc;
    hipLaunchKernelGGL(file_loss_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), loss, loss_hist, hist_cap, cursor);
    return mgp_launch_status();
}
