// Fused Actor forward/backward for ind_agg == 0 (placeholder until the fused kernels land: reports
// MGP_EUNSUPPORTED so the host composes mgp_agg_fwd + mgp_dense_fwd, which are always available).
#include "mgp_common.h"

extern "C" long mgp_actor_saved_floats(const int* dims, int n_layers, int B, int K, int N)
{
    if (dims == nullptr || n_layers <= 0 || n_layers > MGP_MAX_LAYERS || B <= 0 || K <= 0 || N <= 0) return 0;
    long tot = (long)B * dims[0] * K * N;
    for (int i = 1; i < n_layers; ++i) tot += (long)B * dims[i] * N;
    return tot;
}

extern "C" int mgp_actor_fwd(const float*, const float*, const float* const*, const float* const*,
                             const int*, int, float*, float*, int, int, int, void*)
{
    return MGP_EUNSUPPORTED;
}

extern "C" long mgp_actor_bwd_workspace(const int*, int, int, int, int) { return 0; }

extern "C" int mgp_actor_bwd(const float*, const float*, const float* const*, const int*, int,
                             float* const*, float* const*, int, int, int, float*, void*)
{
    return MGP_EUNSUPPORTED;
}
