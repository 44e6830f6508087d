// One-shot gradient exchange over IPC-mapped mailboxes (host side + the stand-alone all-reduce kernel).
// Replaces, for the 6.9 KB flat gradient of a data-parallel DAGGER update, the ring all-reduce torch.distributed would run
// (the reference itself is single-device: train.py:31).  Protocol and reasoning: p2p_device.h.
#include <new>
#include <string.h>
#include "mgp_common.h"
#include "p2p_device.h"

namespace {

constexpr int AR_THREADS = 256;

// buf[i] <- mean over ranks of buf[i], i < n; the last workgroup through publishes the sequence number
__global__ __launch_bounds__(AR_THREADS)
void p2p_allreduce_kernel(float* __restrict__ buf, int n, P2PDev X)
{
    const unsigned seq = (unsigned)X.ctl[0] + 1u;
    const int i = blockIdx.x * AR_THREADS + threadIdx.x;
    if (i < n) buf[i] = p2p_exchange_mean(X, i, buf[i], seq);
    __syncthreads();                                           // every thread of the workgroup has read ctl[0]
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(X.ctl + 1, 1) == (int)gridDim.x - 1) {
            X.ctl[1] = 0;
            X.ctl[0] = (int)seq;
        }
    }
}

}  // namespace

extern "C" int mgp_p2p_create(int world, int rank, int n_floats, MgpP2P** out)
{
    if (out == nullptr || world < 1 || world > MGP_P2P_MAX_WORLD || rank < 0 || rank >= world || n_floats <= 0) return MGP_EINVAL;
    *out = nullptr;
    MgpP2P* c = new (std::nothrow) MgpP2P;
    if (c == nullptr) return MGP_EINVAL;
    memset(c, 0, sizeof(*c));
    mgp_clear_error();
    c->bytes = (size_t)world * 2 * (size_t)n_floats * sizeof(unsigned long long);
    c->mem_kind = 2;
    if (hipExtMallocWithFlags(&c->local, c->bytes, hipDeviceMallocUncached) != hipSuccess) {
        (void)hipGetLastError();
        c->mem_kind = 1;
        if (hipExtMallocWithFlags(&c->local, c->bytes, hipDeviceMallocFinegrained) != hipSuccess) {
            (void)hipGetLastError();
            c->mem_kind = 0;
            if (hipMalloc(&c->local, c->bytes) != hipSuccess) { mgp_launch_status(); delete c; return MGP_ELAUNCH; }
        }
    }
    if (hipMalloc(reinterpret_cast<void**>(&c->ctl), 4 * sizeof(int)) != hipSuccess ||
        hipMemset(c->local, 0, c->bytes) != hipSuccess || hipMemset(c->ctl, 0, 4 * sizeof(int)) != hipSuccess ||
        hipDeviceSynchronize() != hipSuccess) {
        mgp_launch_status();
        if (c->ctl) (void)hipFree(c->ctl);
        (void)hipFree(c->local);
        delete c;
        return MGP_ELAUNCH;
    }
    if (world > 1 && hipIpcGetMemHandle(&c->handle, c->local) != hipSuccess) {
        mgp_launch_status();
        (void)hipFree(c->ctl); (void)hipFree(c->local);
        delete c;
        return MGP_ELAUNCH;
    }
    c->dev.world = world; c->dev.rank = rank; c->dev.n = n_floats; c->dev.ctl = c->ctl;
    c->dev.timeout = 500000000ll;                              // 5 s of the 100 MHz wall clock
    for (int q = 0; q < MGP_P2P_MAX_WORLD; ++q) c->dev.box[q] = static_cast<unsigned long long*>(c->local);
    c->connected = (world == 1);
    *out = c;
    return MGP_OK;
}

extern "C" int mgp_p2p_handle_bytes(void) { return (int)sizeof(hipIpcMemHandle_t); }

extern "C" int mgp_p2p_handle(const MgpP2P* c, void* handle_out)
{
    if (c == nullptr || handle_out == nullptr) return MGP_EINVAL;
    memcpy(handle_out, &c->handle, sizeof(hipIpcMemHandle_t));
    return MGP_OK;
}

extern "C" int mgp_p2p_connect(MgpP2P* c, const void* handles)
{
    if (c == nullptr || handles == nullptr) return MGP_EINVAL;
    if (c->connected) return MGP_OK;
    mgp_clear_error();
    const char* h = static_cast<const char*>(handles);
    for (int q = 0; q < c->dev.world; ++q) {
        if (q == c->dev.rank) continue;
        hipIpcMemHandle_t hq;
        memcpy(&hq, h + (size_t)q * sizeof(hipIpcMemHandle_t), sizeof(hq));
        void* p = nullptr;
        if (hipIpcOpenMemHandle(&p, hq, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
            mgp_launch_status();
            for (int r = 0; r < q; ++r)
                if (c->opened[r]) { (void)hipIpcCloseMemHandle(c->opened[r]); c->opened[r] = nullptr; }
            return MGP_ELAUNCH;
        }
        c->opened[q] = p;
        c->dev.box[q] = static_cast<unsigned long long*>(p);
    }
    c->connected = 1;
    return MGP_OK;
}

extern "C" int mgp_p2p_set_timeout_ms(MgpP2P* c, int ms)
{
    if (c == nullptr || ms <= 0) return MGP_EINVAL;
    c->dev.timeout = (long long)ms * 100000ll;
    return MGP_OK;
}

extern "C" int mgp_p2p_info(const MgpP2P* c, int* world, int* rank, int* n_floats, int* mem_kind)
{
    if (c == nullptr) return MGP_EINVAL;
    if (world) *world = c->dev.world;
    if (rank) *rank = c->dev.rank;
    if (n_floats) *n_floats = c->dev.n;
    if (mem_kind) *mem_kind = c->mem_kind;
    return MGP_OK;
}

// Synchronises `stream`, then returns the status word (0 = every poll so far met its packets; bit 0 = a poll timed out)
// and the sequence number of the last completed exchange.  Not for the hot path.
extern "C" int mgp_p2p_status(MgpP2P* c, int* status, int* seq, void* stream)
{
    if (c == nullptr) return MGP_EINVAL;
    int h[4] = {0, 0, 0, 0};
    mgp_clear_error();
    if (hipStreamSynchronize(static_cast<hipStream_t>(stream)) != hipSuccess ||
        hipMemcpy(h, c->ctl, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) { mgp_launch_status(); return MGP_ELAUNCH; }
    if (status) *status = h[2];
    if (seq) *seq = h[0];
    return MGP_OK;
}

extern "C" int mgp_p2p_destroy(MgpP2P* c)
{
    if (c == nullptr) return MGP_EINVAL;
    (void)hipDeviceSynchronize();
    for (int q = 0; q < MGP_P2P_MAX_WORLD; ++q)
        if (c->opened[q]) (void)hipIpcCloseMemHandle(c->opened[q]);
    if (c->ctl) (void)hipFree(c->ctl);
    if (c->local) (void)hipFree(c->local);
    (void)hipGetLastError();
    delete c;
    return MGP_OK;
}

extern "C" int mgp_p2p_allreduce_mean(MgpP2P* c, float* buf, int n, void* stream)
{
    if (c == nullptr || n <= 0) return MGP_EINVAL;
    MGP_CHECK_PTR(buf);
    if (!c->connected || n > c->dev.n) return MGP_EINVAL;
    mgp_clear_error();
    hipLaunchKernelGGL(p2p_allreduce_kernel, dim3((unsigned)((n + AR_THREADS - 1) / AR_THREADS)), dim3(AR_THREADS), 0,
                       static_cast<hipStream_t>(stream), buf, n, c->dev);
    return mgp_launch_status();
}
