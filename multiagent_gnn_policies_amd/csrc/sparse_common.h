// Helpers shared by the factored-state kernels (sparse_sim.hip, sparse_policy.hip, sparse_persist.hip).
#pragma once
#include "mgp_common.h"
#include "mgp_device.h"
#include "rollout_common.h"

namespace {

constexpr int SS_THREADS = 1024;
constexpr int SS_WAVES = SS_THREADS / 64;
constexpr int SS_ROWS = SS_THREADS / 4;    // rows per workgroup: four lanes per row
constexpr int SS_G = 32;                   // cells per axis, at most
constexpr int SS_MAXN = 2048;              // two agents per thread in the load phase; LDS plan 154 KB at N = 2048
constexpr int SP_MAXTAPS = 4;              // K <= 5
constexpr int SS_SUBCAP = 8;               // hits a lane of the row search can note (sp_sim_kernel and the persistent form alike)

__device__ __forceinline__ double ss_first_lane(double v)
{
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned int lo = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)b);
    const unsigned int hi = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)(b >> 32));
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// inclusive prefix sum over the wave on the DPP path (row_shr 1, 2, 4, 8 inside rows of 16, then row_bcast15 / row_bcast31)
__device__ __forceinline__ int ss_wave_scan(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);     // row_shr:1 (zeros shift in)
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);     // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);     // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);     // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);     // row_bcast15 -> rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);     // row_bcast31 -> rows 2, 3
    return v;
}

// offsets of the layers' blocks inside the weight image of the factored policy kernels (sp_weight_image_kernel)
inline int sp_plan(const int* dims, int n_layers, int K, int* woff, int* wtot)
{
    if (dims == nullptr || n_layers < 1 || n_layers > MGP_MAX_LAYERS) return MGP_EINVAL;
    if (K < 1 || K > SP_MAXTAPS + 1) return MGP_EUNSUPPORTED;
    if (dims[0] != 6 || dims[n_layers] != 2) return MGP_EUNSUPPORTED;
    int tot = 0;
    for (int l = 0; l < n_layers; ++l) {
        const int cin = (l == 0) ? 6 * K : dims[l], cout = dims[l + 1];
        if (cin < 1 || cout < 1 || cin > 4 * RO_KS || cout > 4 * RO_KS) return MGP_EUNSUPPORTED;
        woff[l] = tot;
        tot += ro_weight_image_size(cout, l == n_layers - 1);
        tot = (tot + 3) & ~3;
    }
    *wtot = tot;
    return MGP_OK;
}

}  // namespace

// sparse_persist.hip: the T steps of mgp_sparse_rollout as ONE launch of persistent workgroups where the shape is covered
// (MGP_OK: enqueued; MGP_EUNSUPPORTED: not covered, the caller enqueues the K launches per step; other codes: errors).
int spp_rollout(unsigned long long* bits, float* wrow, float* feat, const float* image, const int* dims, int n_layers,
                float* scratch, float* action, double* x_a, double* x_b, double* rewards, float* expert,
                const MgpFlockParams* p, int B, int K, int N, int T, int cur, int hs, unsigned short* nbr,
                const MgpSparseCollect* collect, hipStream_t st);
