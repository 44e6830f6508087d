// Device helpers shared by the fused kernels (actor_fused.hip, flock.hip, rollout.hip).
#pragma once
#include "mgp_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

__host__ __device__ inline int pad4(int x) { return (x + 3) & ~3; }
__host__ __device__ inline int pad16(int x) { return (x + 15) & ~15; }
// m-tiles (16 output rows each) a layer of `cout` rows is run with: 1, 2 or 4 (3 is padded to 4 to limit the
// number of MLP code instances: this kernel is latency bound and instruction-cache misses show)
__host__ __device__ inline int mtiles(int cout) { const int m = pad16(cout) / 16; return m == 3 ? 4 : m; }

constexpr int AF_MAXW = 64;               // max layer width covered by the fused kernels
constexpr int AF_CS = 68;                 // floats per agent column in the activation buffers (64 channels + pad:
                                          // 68 = 4 mod 64 keeps a 16-lane ds_read_b128 group on disjoint banks)
constexpr int AF_WFS = 20;                // floats per lane in a weight fragment block (16 k-steps + pad, same reason)

// position of channel c inside an agent column of an activation buffer: MFMA B-fragment order, so that lane
// (li, lq) of the wave finds its 16 k-step operands B[k = lq][j = li] contiguous (c = 4 s + lq  ->  lq*16 + s)
__host__ __device__ inline int bpos(int c) { return (c & 3) * 16 + (c >> 2); }

// tanh(x) = 1 - 2 / (1 + exp(2x)): five instructions (v_mul, v_exp_f32, v_add, v_rcp_f32, v_fma), no branches, so
// the evaluations of a tile epilogue pipeline back to back -- the epilogue is instruction-latency bound at 2 waves
// per SIMD (libm's branchy tanhf measured 3x longer).  exp overflow -> rcp(inf) = 0 -> 1; underflow -> -1.
// ABSOLUTE error <= ~2e-7 everywhere (1-ulp v_exp/v_rcp on values in [0,2]); the relative error near 0 is larger,
// which is irrelevant against the 1e-5 absolute parity budget (measured on the goldens: worst 6e-7).
__device__ __forceinline__ float tanh_fast(float x)
{
    const float e = __builtin_amdgcn_exp2f(x * 2.885390081777927f);     // exp(2x) = 2^(2x log2 e)
    return fmaf(-2.f, __builtin_amdgcn_rcpf(1.f + e), 1.f);
}

__device__ __forceinline__ double clipd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

// integrate one agent in registers (FLOCK-SPEC section 1; translation units using this are built -ffp-contract=off)
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

// ---- link fading (FLOCK-SPEC item 8, FlockingStochastic-v0): a radius neighbour pair {i,j} is connected at a step
// iff a 32-bit hash of (seed, pair index, both agents' exact fp64 position words) is >= the drop threshold.  Stateless:
// the network stays a pure function of the state x, the two directions of a pair agree by construction (the position
// words are combined with a commutative add), and all of it is integer arithmetic -- bit-exact against the oracle.
__device__ __forceinline__ unsigned int fmix32(unsigned int h)
{
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}

__device__ __forceinline__ unsigned int fade_word(double px, double py)
{
    const unsigned long long bx = (unsigned long long)__double_as_longlong(px);
    const unsigned long long by = (unsigned long long)__double_as_longlong(py);
    return fmix32((unsigned int)bx + 0x9E3779B1u * (unsigned int)(bx >> 32)
                  + 0x85EBCA77u * (unsigned int)by + 0xC2B2AE3Du * (unsigned int)(by >> 32));
}

__device__ __forceinline__ bool link_up(const MgpFlockParams& p, int i, int j, int N, unsigned int wi, unsigned int wj)
{
    const unsigned int lo = (unsigned int)(i < j ? i : j), hi = (unsigned int)(i < j ? j : i);
    const unsigned int pair = lo * (unsigned int)N + hi;
    return fmix32((wi + wj) ^ (p.link_seed + 0x27D4EB2Fu * pair)) >= p.link_drop;
}

// ---- DAGGER coin (reference gnn_dagger.py:157: np.random.binomial(1, beta) once per environment step).  Counter-based so
// that a step's draw depends only on (seed, global episode index, steps since that episode's reset): the same for any
// chunking of an episode into launches, any lane assignment, any rank.  The expert drives the step iff
// dagger_coin(seed, episode, step) < floor(beta * 2^32)  (beta >= 1: always).  Spec + oracle: oracle/dagger_vec.py.
__device__ __forceinline__ unsigned int dagger_coin(unsigned int seed, unsigned int episode, unsigned int step)
{
    return fmix32(fmix32(seed + 0x9E3779B1u * episode) ^ (0x85EBCA77u * step + 0xC2B2AE3Du));
}
