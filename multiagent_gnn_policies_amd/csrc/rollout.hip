// Episode-resident closed-loop rollout: T x { Actor forward -> simulator step -> delayed-GSO / delay-line transition }
// in ONE launch, one 1024-thread workgroup per episode, the episode's state in LDS.
//
// Reference loop being replaced (one episode, test_model.py:38-44 / gnn_dagger.py:194-203):
//     action = learner.select_action(state)                      -> actor.py:45-86
//     next_state, reward, done, _ = env.step(action)             -> gym_flock (FLOCK-SPEC v1)
//     state = MultiAgentStateWithDelay(..., prev_state=state)    -> state_with_delay.py:44-53
// The two-launch form of one step (mgp_actor_fwd + mgp_flock_step_advance) re-reads the dense delayed operator
// G (B,K,N,N) from HBM twice per step and rewrites it once; that traffic, not arithmetic, is what a step costs at
// N = 100.  Here the dense operator does not exist inside the launch at all.  The state builder's recurrence
// G_j(t) = A_t G_{j-1}(t-1), G_0 = I (state_with_delay.py:44-47) makes tap j of the aggregation
//     y_j(t) = x_{t-j} G_j(t) = x_{t-j} A_t A_{t-1} ... A_{t-j+1},
// and every A is (row weight) x (symmetric 0/1 radius pattern), so the product is evaluated left to right -- the
// K-hop graph shift iterated on a 6 x N block -- along the neighbour lists the simulator phase builds anyway:
// 6 N deg multiply-adds per tap and factor, against 6 N^2 per tap for the dense product plus N^2 deg per slice and
// step to maintain the slices.  LDS holds the fp64 agent states, the delay line (ring), the lists and row weights of
// the last K-1 networks, the MFMA weight fragments and the activation tile (~70 KB at N = 100, K = 3); HBM sees the
// state once on entry and once on exit (plus one reward per step).  The caller's dense slices are read only by the
// first K-1 steps (products that reach back before the launch end with one dense multiplication) and rebuilt on exit.
// Slice 0 of G is the identity by construction (state_with_delay.py:44) and is neither read nor written.
//
// Per step (K + 3 workgroup barriers):
//   A  aggregation: K-1 gather stages; stage q multiplies the running product of every tap j >= q by A_{t-q+1}
//      (thread = (column, tap, parity of the list entry); partial products of taps > q ping-pong in LDS).  Tap 0 is
//      X_0 itself.  Results land in MFMA B-fragment order.
//   B  filter GEMM + tanh hidden layers on fp32 MFMA 16x16x4 (k-ordered fmaf chain, 1e-5 budget): a wave owns 16 agent
//      columns through every hidden layer, activations in place in LDS, no barrier between layers.
//   C  the same wave, no barrier: the 2-wide output layer as a packed-FMA chain (a 16-row MFMA tile would be 7/8
//      padding; a lane takes 8 channels of its column, four lanes are added), then the fp64 integration of the agent (same
//      expression tree as flock.hip / the oracle: bit-exact given the action) and its fp32 coordinates for D1.
//      Meanwhile the nine waves without columns clear the membership bits.
//   D  D1 membership: every unordered pair once (row i tests offsets 1..N/2, 8 threads per row); an fp32 test on
//      coordinates relative to a reference point decides pairs that are clear of the radius by a proven error band, the
//      exact fp64 expression of the spec decides the rest -- the bits are always the oracle's.  Both rows of a pair get
//      their bit by LDS atomic OR.  D2/D3: 4 lanes per row turn the row's bits into an ascending neighbour list (into the
//      history slot of the oldest network) and sum the fp64 feature terms of actual neighbours.  One otherwise idle
//      wave computes the step's reward.  The delay line is a ring: the new features overwrite the oldest tap.
// (The first version of this kernel kept slices 1..K-1 densely in LDS -- 80 KB at N = 100, K = 3 -- aggregated taps >= 2
//  on the matrix cores and advanced the slices with row gathers every step: 8.6 us per step of 256 episodes against 7.2.)
// This translation unit is built with -ffp-contract=off (fp64 spec arithmetic); fp32 fused multiply-adds are explicit.
#include <hip/hip_ext.h>
#include "rollout_common.h"
#if !defined(MGP_RO_WIDE) && !defined(MGP_RO_X128) && !defined(MGP_RO_T512) && !defined(MGP_RO_X2) && !defined(MGP_RO_XD)
#define MGP_RO_BASE 1                      // the build that owns the public entry points (layer widths <= 32)
#endif

namespace {

// [r6] MGP_RO_T512 (rollout_t512.hip): the headline instantiation once more as a 512-thread workgroup of <= 80 KB, so that a CU
// holds TWO episodes when a launch has more episodes than the device has CUs.  A step is a chain of dependent phases of one
// episode (vector pipes about half busy, waves parked 57 % of their cycles); a second, independent chain on the
// CU fills those gaps.  Eight waves: the policy phase as before (7 tile waves), S1 in two row passes of 64 rows, S2's two
// groups one after the other on the same seven waves, reward + Verlet helper on wave 7.  Every sum keeps its order: the bits
// are the 1024-thread build's.
#ifdef MGP_RO_T512
constexpr int RO_THREADS = 512;
#else
constexpr int RO_THREADS = 1024;
#endif
constexpr bool RO_T512 = RO_THREADS == 512;
// [r6] MGP_RO_X2 (rollout_w128x2.hip): two hidden layers of up to 128 channels (cfg/hidden_size.cfg:81-82) at the headline (N, K):
// the second layer's K blocks 2 and 3 streamed through one LDS buffer by LDS-DMA (rollout_common.h, RO_X2_*)
#ifdef MGP_RO_X2
constexpr bool RO_X2 = true;
#else
constexpr bool RO_X2 = false;
#endif
// MGP_RO_XD (rollout_w128xd.hip): three and more hidden layers of up to 128 channels (cfg/hidden_size.cfg:104-106, 128-130) at the
// headline (N, K): EVERY K block of the layers behind the first is streamed, through a ring of three 24 KB buffers, two blocks in
// flight while one is multiplied (one workgroup barrier per block)
#ifdef MGP_RO_XD
constexpr bool RO_XD = true;
#else
constexpr bool RO_XD = false;
#endif
constexpr bool RO_XS = RO_X2 || RO_XD;                        // a build that streams weight blocks
constexpr int RO_WAVES = RO_THREADS / 64;
#ifndef RO_S1L
#define RO_S1L 8                          // (4 lanes per row -- 7 waves instead of 13, 25 candidates each -- measured 1 % slower)
#endif
// (The pair test in dot-product form -- |s_i - s_j|^2 = n_i + (n_j - 2 s_i . s_j), six instructions per candidate instead of
//  eight -- was measured in round 4: a cancellation error ~ M^2 instead of M R widens the band 40x, one to three lanes per step
//  take the fp64 fallback, 0.8k cycles SLOWER per step.  Removed in round 5; LAB_NOTES.md "Pair test (S1)".)
constexpr int RO_PIECES = RO_S1L;         // lanes per agent row in the pairwise pass S1 (adjacent lanes): 8 or 4
constexpr int RO_MAXN = 128;              // RO_THREADS / RO_PIECES rows; membership bits of a row fit 2 x u64
constexpr int RO_LDS_LIMIT = 160 * 1024;
__host__ __device__ constexpr int ro_take(int& off, int bytes) { const int o = off; off += (bytes + 15) & ~15; return o; }
// list entries per lane and pass in the S2 gather group: 4 where (N, K) are compile-time (2 measured 4 % slower there); 2 in the
// run-time sized builds, whose S1T = 4 accumulator sets leave no registers for four entries in flight (they spilled: -20 %)
template <int CK> struct RoGatherUnroll { static constexpr int value = CK ? 4 : 2; };

// ---- [r5] Verlet candidate lists for the membership phase S1 (temporal coherence: agents move <= a few % of the radius per step).
// Every RO_VSKIN-dependent quantity is a pure accelerator: the bits S1 produces are the exact test's (fp32 band + fp64
// fallback) on every step; the candidate list only says which pairs CANNOT be within R and need no test.
//   rebuild (step t_r): row i lists every j with |x_i - x_j| < RV = R (1 + RO_VSKIN)  (fp32 test, widened by the proven error
//       band: conservative), ascending, up to RO_VCAP entries; the helper wave stores the positions p(t_r) and a frame velocity V.
//   later steps: D_i(t) = p_i(t) - p_i(t_r) - V dt (t - t_r) for ANY common V; |x_i - x_j|(t_r) <= |x_i - x_j|(t) + |D_i| + |D_j|,
//       so while 2 max_i |D_i| <= skin - margin every pair within R at step t is on the list of step t_r.  The helper wave
//       evaluates the bound ONE STEP AHEAD (from p, v of the current state and the acceleration clip: |ue| <= max_accel *
//       action_gain per axis), off the critical path, and leaves next step's mode in LDS:
//         CHEAP   test the listed candidates only (eight lanes per row take entries piece, piece + 8, ...: ballots give the
//                 row's hit mask, hits are compacted into the ascending neighbour list)
//         REBUILD full-row pass that writes the candidate lists, then the cheap pass on them
//         FULL    the all-pairs exact pass (round 4's S1): one-step launches, flocks whose relative motion would outrun the
//                 skin within RO_VMINLIFE steps, rows with more than RO_VCAP candidates (per row)
#ifndef RO_VERLET
#define RO_VERLET 1
#endif
#ifndef RO_BC_PRIO
#define RO_BC_PRIO 2                      // policy phase: raised priority for the SIMDs' second tiles until the end of layer RO_BC_PRIO - 1 (0: off);
                                          // measured 1 / 2 / 3 against 0: -1.0 % per step each (B/C 4.98k -> 4.74k cycles at 2), same bits
#endif
#ifndef RO_S2_PRIO
#define RO_S2_PRIO 0                      // S2: wave priority of the gather group (0: as dispatched)
#endif
#ifndef RO_FEAT_PAIR
#define RO_FEAT_PAIR 0                    // feature pass of S2: two list entries per trip with their LDS reads in flight together
                                          // (measured: the feature waves end 0.2k cycles earlier, the phase -- bound by the gather waves -- does not)
#endif
#ifndef RO_FEAT_FMA
#define RO_FEAT_FMA 1
#endif
#ifndef MGP_RO_VL_BUILD
#define MGP_RO_VL_BUILD 1
#endif
// The 64- and 128-wide builds also keep the sized instantiation WITHOUT lists: their weight images leave less LDS (four 64-wide
// layers at N = 100 fit only without the lists), and a shape must not fall to the run-time sized build for that.
#if defined(MGP_RO_WIDE) || defined(MGP_RO_X128)
#define MGP_RO_VL_BOTH 1
#else
#define MGP_RO_VL_BOTH 0
#endif
#ifndef RO_VSKIN
#define RO_VSKIN 0.3f                     // skin in units of the communication radius (<= 1: the error band is proven for pairs within 2R)
#endif
#ifndef RO_VPU
#define RO_VPU 2                          // candidate passes (of eight lanes) per trip of the cheap pass
#endif
#ifndef RO_VTRIPS
#define RO_VTRIPS 4                       // trips a row's list may take: capacity RO_VPU * 8 * RO_VTRIPS candidates
#endif
#ifndef RO_VMINLIFE
#define RO_VMINLIFE 2                     // a rebuild (the exact pass + ~0.4k cycles) is worth it if the list is expected to serve this many steps
#endif
constexpr int RO_VCAP = 8 * RO_VPU * RO_VTRIPS;               // candidates a row's list holds
constexpr int RO_VSTR = 4 * ((RO_VCAP / 4) | 1);              // row stride in bytes: an odd word count (neighbouring rows on other banks)
enum { RO_VM_CHEAP = 0, RO_VM_REBUILD = 1, RO_VM_FULL = 2 };
struct RoVOff { int list, cnt, pref, gap, flag, total; };          // byte offsets relative to the end of the weight image
__host__ __device__ constexpr RoVOff ro_voffsets(int N)
{
    RoVOff v = {};
    int off = 0;
    v.list = ro_take(off, N * RO_VSTR);
    v.cnt = ro_take(off, N * 4);
    v.pref = ro_take(off, 2 * N * 8);
    v.gap = ro_take(off, N * 4);                              // float [N]: distance to the nearest agent at the rebuild - R, agents without candidates
    v.flag = ro_take(off, 64);                                // int [0] next step's mode, [1] steps the lists have served, [2] steps of backoff, [3] next backoff;
                                                              // double [4] at +16: frame offset (x, y), frame velocity (x, y); bytes 48..55: target
                                                              // of masked list writes
    v.total = off;
    return v;
}

// Pair tests of one S1 lane: candidates j0 .. j0 + nt - 1 of row `si` (fp32 coordinates relative to the reference point);
// sign bits instead of compare / select pairs (a v_cmp -> v_cndmask pair costs wait states on gfx9): the sign of r2 - t_in says
// "clearly inside", the sign of t_out - r2 says "clearly outside"; v_alignbit shifts each into a mask (test k ends in bit
// nt - 1 - k).  A NaN distance (diverged episode) classifies arbitrarily -- the state is garbage by then, and every index
// stays valid.  WANT_V: a third threshold in the same pass (the candidate-list build of a rebuild step).
template <int NTC, bool WANT_V>
__device__ __forceinline__ void ro_s1_masks(const float4* sxy, const float six, const float siy, const int j0, const int N,
                                            const int nt_rt, const float t_in, const float t_out, unsigned int& im, unsigned int& om,
                                            const float t_v = 0.f, unsigned int* vm_ = nullptr)
{
    const int nt = NTC ? NTC : nt_rt;
    im = 0u; om = 0u;
    unsigned int vm = 0u;                                     // WANT_V: a third mask -- "clearly farther than the candidate radius"
#pragma unroll
    for (int c0 = 0; c0 < 128 / RO_PIECES; c0 += 8) {
        if (c0 < nt) {
            float2 sj[8];
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (c0 + q < nt) sj[q] = *reinterpret_cast<const float2*>(&sxy[min(j0 + c0 + q, N - 1)]);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (c0 + q < nt) {
                    const float dx = six - sj[q].x, dy = siy - sj[q].y;
                    const float r2 = fmaf(dy, dy, dx * dx);
                    im = __builtin_amdgcn_alignbit(im, __float_as_uint(r2 - t_in), 31);
                    om = __builtin_amdgcn_alignbit(om, __float_as_uint(t_out - r2), 31);
                    if (WANT_V) vm = __builtin_amdgcn_alignbit(vm, __float_as_uint(t_v - r2), 31);
                }
            }
        }
    }
    if (WANT_V) *vm_ = vm;
}
#ifdef MGP_RO_PROFILE
__device__ unsigned long long mgp_ro_stamps[16 * 32];     // [wave][stamp]
#ifndef RO_STAMP_T
#define RO_STAMP_T 5
#endif
#define RO_STAMP(i) do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && t == RO_STAMP_T) mgp_ro_stamps[(threadIdx.x >> 6) * 32 + (i)] = __builtin_readcyclecounter(); } while (0)
#define RO_STAMPX(i) do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) mgp_ro_stamps[(threadIdx.x >> 6) * 32 + (i)] = __builtin_readcyclecounter(); } while (0)
// launch anatomy: the 100 MHz wall clock of EVERY workgroup at kernel begin / entry done / first steps / loop end / kernel end
// (tools/harness/ro_launch_prof.hip: dispatch ramp, entry, cold first step, exit -- what a launch costs beyond its steps)
__device__ long long mgp_ro_wall[4096 * 8];
__device__ unsigned int mgp_ro_vstat[4096 * 4];             // S1 modes chosen per workgroup: cheap / rebuild / full
#define RO_WALL(i) do { if (threadIdx.x == 0 && blockIdx.x < 4096) mgp_ro_wall[blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
#else
#define RO_WALL(i) do { } while (0)
#define RO_STAMP(i) do { } while (0)
#define RO_STAMPX(i) do { } while (0)
#endif

struct RoParams {
    const float* W[MGP_MAX_LAYERS];
    const float* b[MGP_MAX_LAYERS];
    int dims[MGP_MAX_LAYERS + 1];         // 6, h1, ..., 2
    int woff[MGP_MAX_LAYERS];             // offset (floats) of layer l's fragment block inside the weight image
    int n_layers;
    int wtot;                             // floats in the weight image
    int bf;                               // hidden-layer blocks of the image hold split-bf16 pieces (else fp32 fragments)
};

// LDS layout (byte offsets).  Every region except the weight image depends on (N, K) only, and the weight image comes
// last, so a kernel instantiated for a fixed (N, K) has compile-time LDS addresses (ds_read/ds_write immediates).
// H = max(K - 1, 1) history slots hold the neighbour lists / row weights of the last K - 1 networks (ring over time).
struct RoOff {
    int pos;                              // double px, py, vx, vy [4][N] + reference point [2]
    int mask;                             // u64 [N][2] membership bits of the network being built (cleared in phase B)
    int wrow;                             // float [H][N] network weight of row i (1/deg or 1)
    int uact;                             // float [2][N] action (the Actor output layout (nA, N))
    int xt;                               // float [K][Np][8] delay line, ring over taps, transposed (6 features + 2 pad); Np = N
                                          // rounded up to a multiple of 4, rows >= N are zero
    int vb;                               // float [2][K-2][Np][8] partial products of taps >= 2 between gather stages (ping-pong)
    int act;                              // float [ncols16][RO_CS] activations (in place through the layers); row buffers on exit
    int rlist;                            // u8 [H][N][RS] ascending neighbour lists (RS = ro_list_stride(N))
    int rcnt;                             // int [H][N] list lengths
    int sxy;                              // float4 [N] fp32 coordinates relative to the reference point: {sx, sy, sx^2 + sy^2, -}
    int mmax;                             // float [8]: max |relative coordinate| of the step, one slot per MLP wave
    int wtab;                             // float [N + 1]: row weight of a network row by its degree (1/max(deg,1) or 1)
    int uexp;                             // float [2][N] expert action of the current state (data collection) + double [2] velocity sums
    int wl;                               // float weight image: per layer fragments [MT][64][RO_WFS] + bias [MT*16]
};

__host__ __device__ constexpr int ro_hist(int K) { return K > 2 ? K - 1 : 1; }
// list row stride in bytes: room for N - 1 entries, a multiple of 4 with an ODD word count -- neighbouring lanes walk
// neighbouring rows, and an even word stride put them 8 to a bank
__host__ __device__ constexpr int ro_list_stride(int N) { const int w = (N - 1 + 3) >> 2; return 4 * (w | 1); }

// Factored hand-over of the operator history between launches ("carry"): per episode the membership BITS of the last
// H = ro_hist(K) networks, newest first (slot 0 = A_t of the state, slot q = A_{t-q}; NW 64-bit words per row: 2 for
// N <= 128, 4 beyond), then their row weights [H][N] fp32.  An all-zero carry is the history of a reset observation (no
// earlier network exists: every product vanishes, as the reference's zero-filled slices do, state_with_delay.py:44-47).
__host__ __device__ constexpr int ro_carry_nw(int N) { return N > 128 ? 4 : 2; }
__host__ __device__ constexpr size_t ro_carry_words(int K, int N)     // 64-bit words per episode
{
    return (size_t)ro_hist(K) * N * ro_carry_nw(N) + ((size_t)ro_hist(K) * N * 4 + 7) / 8;
}

__host__ __device__ constexpr RoOff ro_offsets(int N, int K)
{
    RoOff c = {};
    int off = 0;
    const int Np = (N + 3) & ~3, H = ro_hist(K);
    c.pos = ro_take(off, (4 * N + 2) * 8);
    c.mask = ro_take(off, 2 * N * 8);
    c.wrow = ro_take(off, H * N * 4);
    c.uact = ro_take(off, 2 * N * 4);
    c.xt = ro_take(off, K * Np * 8 * 4);
    // (K = 3 uses parity 1 of the ping-pong only -- stage 1 writes it, the last stage reads it --: the T512 build, which must stay
    //  under 80 KB, allocates that half alone and biases the pointer)
    c.vb = ro_take(off, (((RO_T512 || RO_XS) && K == 3) ? 1 : 2) * (K > 2 ? K - 2 : 0) * Np * 8 * 4);
    c.act = ro_take(off, ((N + 15) & ~15) * RO_CS * 4);
    c.rlist = ro_take(off, H * N * ro_list_stride(N));
    c.rcnt = ro_take(off, H * N * 4);
    c.sxy = ro_take(off, N * 16);
    c.mmax = ro_take(off, 32);
    c.wtab = ro_take(off, (N + 1) * 4);
    c.uexp = ro_take(off, 2 * N * 4 + 16);
    c.wl = off;
    return c;
}

// CN / CK: compile-time (N, K) of a specialised instantiation (0 = take the run-time arguments): constant LDS addresses,
// loop bounds and divisors shorten every phase's address arithmetic and relieve the SGPR file (the generic build spills).
// FD: link fading (FlockingStochastic-v0) compiled in.
// CL: DAGGER data collection (reference gnn_dagger.py:154-178) compiled in: every step files the state it starts from --
// features, membership bits of its network, expert label, age -- as a compact frame, and the step is driven by the expert
// with probability beta (a counter-based coin: mgp_device.h dagger_coin), else by the policy.
// CM: the reference's policy shape compiled in -- two hidden layers of 32 (cfg/dagger.cfg: hidden_size 32, n_layers 2): the
// layer loop unrolls, no layer metadata is decoded, the MLP's scalar control flow disappears.
// WBF: the hidden layers run on split-bf16 MFMA from piece records (rollout_common.h) -- every build but the checker; the
// 64-wide build falls back to fp32 fragments (WBF = false) for policies whose 40 % larger piece image does not fit the LDS
// (four 64-wide layers at N = 100: cfg/hidden_size.cfg [4, 64]).
// VL: Verlet candidate lists in S1 (above; the launcher selects it when the lists fit the LDS next to the weight image).
template <int CN, int CK, bool FD, bool CL, bool CM = false, bool WBF = RO_BF16_CHAIN, bool VL = false>
__global__ __launch_bounds__(RO_THREADS, RO_T512 ? 4 : 1)
void rollout_kernel(double* __restrict__ x, float* __restrict__ G, float* __restrict__ Xd, float* __restrict__ action,
                    double* __restrict__ rewards, RoParams P, MgpFlockParams p, int K_arg, int N_arg, int T,
                    unsigned long long dimsA, unsigned int dims8, unsigned long long woffA, unsigned long long woffB,
                    int n_layers_arg, const float* __restrict__ image, int image_floats, unsigned long long* __restrict__ carry,
                    int flags, MgpCollect cl)
{
    RO_WALL(0);
    const int N = CN ? CN : N_arg, K = CK ? CK : K_arg;
    constexpr int WFS = ro_wfs(WBF);                          // floats per lane and m-tile of a hidden layer's block
    const int n_layers = CM ? 3 : n_layers_arg;
    const RoOff cv = ro_offsets(N, K);
    const int H = ro_hist(K);
    extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
    double* spx = reinterpret_cast<double*>(smraw + cv.pos);
    double* spy = spx + N; double* svx = spx + 2 * N; double* svy = spx + 3 * N;
    double* cref = spx + 4 * N;                               // reference point of the fp32 membership test (2 doubles)
    unsigned long long* rowmask = reinterpret_cast<unsigned long long*>(smraw + cv.mask);
    float* wrow = reinterpret_cast<float*>(smraw + cv.wrow);
    float* uact = reinterpret_cast<float*>(smraw + cv.uact);
    float* XT = reinterpret_cast<float*>(smraw + cv.xt);
    float* VB = reinterpret_cast<float*>(smraw + cv.vb) - (((RO_T512 || RO_XS) && K == 3) ? (K - 2) * ((N + 3) & ~3) * 8 : 0);
    float* wl = reinterpret_cast<float*>(smraw + cv.wl);
    float* act = reinterpret_cast<float*>(smraw + cv.act);
    unsigned char* rlist = smraw + cv.rlist;
    int* rcnt = reinterpret_cast<int*>(smraw + cv.rcnt);
    float4* sxy = reinterpret_cast<float4*>(smraw + cv.sxy);
    float* wtab = reinterpret_cast<float*>(smraw + cv.wtab);
    float* uexp = reinterpret_cast<float*>(smraw + cv.uexp);                 // [2][N]
    double* vtot = reinterpret_cast<double*>(smraw + cv.uexp + ((2 * N * 4 + 7) & ~7));
    const int RS = ro_list_stride(N);                         // list row stride (bytes)
    // Verlet regions behind the weight image (VL builds; the launcher has checked that they fit)
    const RoVOff vo = ro_voffsets(N);
    unsigned char* vbase = smraw + cv.wl + (CM ? (2 * (2 * 64 * WFS + 32) + ((2 * RO_OUTC + 2 + 15) & ~15)) * 4 : ((image_floats * 4 + 15) & ~15));
    unsigned char* vlist = vbase + vo.list;
    int* vcnt = reinterpret_cast<int*>(vbase + vo.cnt);
    double* vpref = reinterpret_cast<double*>(vbase + vo.pref);
    float* vgap = reinterpret_cast<float*>(vbase + vo.gap);
    int* vflag = reinterpret_cast<int*>(vbase + vo.flag);

    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Np = (N + 3) & ~3;                              // delay-line rows: N rounded up to a multiple of 4
    const int NN = N * N, FK = 6 * K;
    const int ncols16 = pad16(N), NT = ncols16 / 16;
    double* xb = x + (size_t)b * N * 4;
    float* Gb = G + (size_t)b * K * NN;
    float* Xb = Xd + (size_t)b * K * 6 * N;

    // ------------------------------------------------------------------ entry: the episode's state -> LDS
    // (the caller's dense operator slices stay in HBM: they are read by the first K - 1 steps only, see phase A)
    // Every global read of the entry is REQUESTED before the first one is consumed, and each request is coalesced: the delay
    // line is read in its memory order (the transposing index math is on the LDS side), the agent states as one double per
    // thread, the carried history (bits + row weights) before the barrier it is needed behind.  (Round 3 read the delay line
    // element by element in LDS order -- 38 wave-loads that touched 64 cache lines each -- and fetched the carry after the
    // first barrier: 3.3 us from kernel begin to the first step, tools/harness/ro_launch_prof.hip.)
    constexpr int XTP = 4;                                    // delay-line elements per thread (K 6 N <= 5 * 6 * 128 = 3840)
    constexpr int CYP = 2;                                    // (network, row, quarter) items per thread (H N 4 <= 2048)
    static_assert(!RO_T512 || (CN == 100 && CK == 3), "the 512-thread build is instantiated for the headline (N, K) only");
    static_assert(!RO_T512 || (CK * 6 * CN <= XTP * RO_THREADS && 2 * CN * 4 <= CYP * RO_THREADS && 4 * ((CN + 15) & ~15) <= RO_THREADS),
                  "thread maps of the 512-thread build");
    const int nXT = K * 6 * N;
    float xtv[XTP];
#pragma unroll
    for (int r = 0; r < XTP; ++r) { const int e = tid + r * RO_THREADS; xtv[r] = (e < nXT) ? Xb[e] : 0.f; }
    const double posv = (tid < 4 * N) ? xb[tid] : 0.0;        // thread 4 i + c: component c of agent i
    const size_t cwords = ro_carry_words(K, N);
    const bool enter_carry = (flags & MGP_RO_ENTER_CARRY) != 0;
    unsigned long long cy_lo[CYP] = {0ull, 0ull}, cy_hi[CYP] = {0ull, 0ull};
    float cy_w[CYP] = {0.f, 0.f};
    if (enter_carry) {
        const unsigned long long* cb = carry + (size_t)b * cwords;
        const float* cw = reinterpret_cast<const float*>(cb + (size_t)H * N * 2);
#pragma unroll
        for (int r = 0; r < CYP; ++r) {
            const int it = tid + r * RO_THREADS;
            if (it < H * N * 4) {
                const int rq = it >> 2;
                cy_lo[r] = cb[(size_t)rq * 2]; cy_hi[r] = cb[(size_t)rq * 2 + 1];
                cy_w[r] = cw[rq];
            }
        }
    }
    unsigned long long coin_thr = 0ull;
    unsigned int coin_ep = 0u;
    float uexv = 0.f;
    float betav = 0.f;
    if (CL) {
        if (tid < 2 * N) uexv = cl.expert_io[(size_t)b * 2 * N + tid];
        betav = cl.beta[b];
        coin_ep = cl.episode[b];
    }
    // weights in MFMA A-fragment order (see actor_fused.hip): wfrag[mt][lane][RO_WFS], lane = (c & 3) * 16 + (o & 15),
    // slot s = c >> 2, zero padded; then the bias of the layer's MT*16 rows.  A caller that launches repeatedly with the
    // same weights passes the image prebuilt (mgp_rollout_image: the same elements, computed once): a flat 16-byte copy.
    // X2: LDS = [layer 0][layer-1 blocks 0, 1][stream buffer <- block 2][layer-1 bias][output layer]; the image in HBM holds
    // [layer 0][blocks 0 .. 3][bias][output layer]: one flat copy up to and including block 2, one of what lies behind block 3
    float* x2_b0 = wl + RO_X2_L0;                             // (X2 builds only)
    float* x2_buf = x2_b0 + 2 * RO_X2_BLK;
    float* x2_bias1 = x2_buf + RO_X2_BLK;
    float* x2_out = x2_bias1 + 128;
    const float* x2_img_b2 = image + RO_X2_L0 + 2 * RO_X2_BLK;
    // XD: LDS = [layer 0][biases of layers 1 .. L - 1][output layer][ring of three K-block buffers]; image = [layer 0][per layer: 4 blocks +
    // bias][output layer].  Blocks 0 and 1 of layer 1 are requested into ring slots 0, 1 here (waves 14 / 15) and waited for behind
    // the entry's last barrier... in front of it (below).
    const int xd_nb = 4 * (n_layers - 2);                     // streamed K blocks per step (XD builds)
    float* xd_bias = wl + RO_X2_L0;                           // [n_layers - 2][128]
    float* xd_out = xd_bias + 128 * (n_layers - 2);
    float* xd_ring = xd_out + RO_X2_OUT;                      // [3][RO_X2_BLK]
    auto xd_block_src = [&](const int g) -> const unsigned char* {      // global block g = 4 (layer - 1) + kb
        return reinterpret_cast<const unsigned char*>(image + RO_X2_L0 + (g >> 2) * RO_X2_L1 + (g & 3) * RO_X2_BLK);
    };
    auto xd_dma = [&](const int g, const int wu_, const int ln_) {   // by waves 14, 15: 12 requests of 1 KB each
        const unsigned char* src = xd_block_src(g);
        unsigned char* dst = reinterpret_cast<unsigned char*>(xd_ring + (g % 3) * RO_X2_BLK);
        for (int c = (wu_ - (RO_WAVES - 2)) * 1024; c < RO_X2_BLK * 4; c += 2048) ro_lds_dma16(src + c + ln_ * 16, dst + c);
    };
    if (RO_XD) {
        const float4* src4 = reinterpret_cast<const float4*>(image);
        float4* dst4 = reinterpret_cast<float4*>(wl);
        for (int e = tid; e < RO_X2_L0 / 4; e += RO_THREADS) dst4[e] = src4[e];
        for (int e = tid; e < 32 * (n_layers - 2); e += RO_THREADS) {           // biases: 32 float4 per layer, behind its four blocks
            const int l1 = e >> 5, q = e & 31;
            reinterpret_cast<float4*>(xd_bias)[e] = src4[(RO_X2_L0 + l1 * RO_X2_L1 + 4 * RO_X2_BLK) / 4 + q];
        }
        for (int e = tid; e < RO_X2_OUT / 4; e += RO_THREADS)
            reinterpret_cast<float4*>(xd_out)[e] = src4[(RO_X2_L0 + (n_layers - 2) * RO_X2_L1) / 4 + e];
        const int wu_ = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        if (wu_ >= RO_WAVES - 2) { xd_dma(0, wu_, (int)(threadIdx.x & 63)); xd_dma(1, wu_, (int)(threadIdx.x & 63)); }
    } else if (RO_X2) {
        const float4* src4 = reinterpret_cast<const float4*>(image);
        float4* dst4 = reinterpret_cast<float4*>(wl);
        constexpr int HEAD4 = (RO_X2_L0 + 3 * RO_X2_BLK) / 4, TAIL4 = (128 + RO_X2_OUT) / 4;
        for (int e = tid; e < HEAD4; e += RO_THREADS) dst4[e] = src4[e];
        for (int e = tid; e < TAIL4; e += RO_THREADS) dst4[HEAD4 + e] = src4[HEAD4 + RO_X2_BLK / 4 + e];
    } else if (image != nullptr) {
        const float4* src4 = reinterpret_cast<const float4*>(image);
        float4* dst4 = reinterpret_cast<float4*>(wl);
        for (int e = tid; e < image_floats / 4; e += RO_THREADS) dst4[e] = src4[e];
    } else {
        for (int l = 0; l < P.n_layers; ++l) {
            const int cin = (l == 0) ? FK : P.dims[l];
            const int cout = P.dims[l + 1];
            const bool last = l == P.n_layers - 1;
            const int tot = last ? ro_weight_image_size(cout, true) : ro_chain_image_size(cout, false, WBF);
            float* dst = wl + P.woff[l];
            for (int e = tid; e < tot; e += RO_THREADS)
                dst[e] = last ? ro_weight_image_elem(P.W[l], P.b[l], cin, cout, true, e) : ro_chain_image_elem(P.W[l], P.b[l], cin, cout, l, false, e, WBF);
        }
    }
    // ---- LDS regions no load fills (while the requests are in flight)
    for (int i = tid; i < 2 * N; i += RO_THREADS) rowmask[i] = 0ull;
    for (int e = tid; e < K * Np; e += RO_THREADS)                               // the two pad floats of every delay-line row
        *reinterpret_cast<float2*>(XT + (size_t)e * 8 + 6) = make_float2(0.f, 0.f);
    if (Np > N)
        for (int e = tid; e < K * (Np - N) * 6; e += RO_THREADS) {               // rows N .. Np - 1 (N % 4 != 0)
            const int f = e % 6, mk = e / 6, k = mk / (Np - N), m = N + mk - k * (Np - N);
            XT[((size_t)k * Np + m) * 8 + f] = 0.f;
        }
    for (int c = tid; c <= N; c += RO_THREADS) {              // the expression of phase D3, tabulated by degree
        const double deg = (double)c;
        wtab[c] = (float)(p.mean_pooling ? 1.0 / (deg == 0.0 ? 1.0 : deg) : 1.0);
    }
    {
        float4* za = reinterpret_cast<float4*>(act);
        for (int i = tid; i < ncols16 * RO_CS / 4; i += RO_THREADS) za[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        unsigned int* zl = reinterpret_cast<unsigned int*>(rlist);            // every list byte is always a valid row index
        for (int i = tid; i < H * N * RS / 4; i += RO_THREADS) zl[i] = 0u;
        for (int i = tid; i < H * N; i += RO_THREADS) { rcnt[i] = 0; wrow[i] = 0.f; }
    }
    // ---- the requested values -> LDS
#pragma unroll
    for (int r = 0; r < XTP; ++r) {                           // tap k -> ring slot (K - k) % K, cur = 0; row m, feature f
        const int e = tid + r * RO_THREADS;
        if (e < nXT) {
            const int kf = e / N, m = e - kf * N, k = kf / 6, f = kf - k * 6;
            const int slot = (k == 0) ? 0 : K - k;
            XT[((size_t)slot * Np + m) * 8 + f] = xtv[r];
        }
    }
    if (tid < 4 * N) spx[(tid & 3) * N + (tid >> 2)] = posv;
    if (tid < 2) cref[tid] = posv;                            // agent 0's position: the reference point of the fp32 membership test
    if (VL && tid == RO_THREADS - 1) {
        // the first step builds the candidate lists unless the launch is too short to use them; sxy[N] (the unused 32 bytes
        // behind the coordinates) is the SENTINEL the padding entries of a candidate list point at: far outside every radius
        vflag[0] = (T >= RO_VMINLIFE) ? RO_VM_REBUILD : RO_VM_FULL;
        vflag[1] = 0; vflag[2] = 0; vflag[3] = 8;
        sxy[N] = make_float4(3.0e18f, 3.0e18f, 0.f, 0.f);
    }
    if (CL) {
        if (tid < 2 * N) uexp[tid] = uexv;
        const double bq = floor((double)betav * 4294967296.0);            // P(expert drives) in units of 2^-32
        coin_thr = bq <= 0.0 ? 0ull : (bq >= 4294967296.0 ? 4294967296ull : (unsigned long long)bq);
    }
    if (RO_XD) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (ring slots 0, 1 hold blocks 0, 1)
    __syncthreads();
    RO_WALL(7);
    // history networks handed over in factored form (MGP_RO_ENTER_CARRY): bits -> ascending neighbour lists, four lanes per
    // (network, row) as in phase D2; carry slot q = A_{t0 - q} goes to ring slot (H - q) % H, i.e. hs = 0 is the current
    // network and every tap's product is available as lists from the first step on (t_off): the dense slices are not read
    int t_off = 0;
    if (enter_carry) {
        t_off = K - 1;
#pragma unroll
        for (int r = 0; r < CYP; ++r) {
            const int it = tid + r * RO_THREADS;
            if (it < H * N * 4) {
                const int cq = it & 3, rq = it >> 2, q = rq / N, row = rq - q * N;
                const int slot = (q == 0) ? 0 : H - q;
                const unsigned long long lo = cy_lo[r], hi = cy_hi[r];
                const int cnt = __popcll(lo) + __popcll(hi);
                unsigned int chunk; int pos;
                if (cq == 0) { chunk = (unsigned int)lo; pos = 0; }
                else if (cq == 1) { chunk = (unsigned int)(lo >> 32); pos = __popc((unsigned int)lo); }
                else if (cq == 2) { chunk = (unsigned int)hi; pos = __popcll(lo); }
                else { chunk = (unsigned int)(hi >> 32); pos = __popcll(lo) + __popc((unsigned int)hi); }
                unsigned char* lp = rlist + ((size_t)slot * N + row) * RS;
                while (chunk) { lp[pos++] = (unsigned char)(32 * cq + __builtin_ctz(chunk)); chunk &= chunk - 1u; }
                if (cq == 0) {
                    rcnt[slot * N + row] = cnt;
                    wrow[slot * N + row] = cy_w[r];
                    if (q == 0) { rowmask[2 * row] = lo; rowmask[2 * row + 1] = hi; }
                }
            }
        }
    }
    for (int e = tid; e < N * 8; e += RO_THREADS) {             // tap 0 of the first step (later steps: written in D3)
        const int f = e & 7, n = e >> 3;
        if (f < 6) act[n * RO_CS + rpos(f * K)] = XT[e];          // cur = 0: slot 0 holds tap 0
    }
    __syncthreads();

    // Thread roles are RE-DERIVED from the thread index at the top of every phase (ro_fresh_tid: opaque to the optimiser), not
    // kept across the step loop: hoisted out of the loop they live through every phase, and in the run-time sized builds (which
    // spill ~90 registers) they came back from scratch memory in every phase -- 21 scratch loads in S1 alone, 12.6k cycles
    // where the compile-time build takes 3.4k.
    // The compile-time sized builds of widths <= 32 re-derive in phase B/C only (few spills to begin with; re-deriving everywhere
    // cost them 3 %; the 128-wide build gains 13 % from it even there).
#ifndef RO_FRESH_MASK
#define RO_FRESH_MASK 2                    // phases that re-derive their roles in the compile-time sized builds too (bit 0: A, 1: B/C,
                                           // 2: S1, 3: S2).  Measured on the headline build: B/C alone -1.5 %, S1 +3 %, S2 +1 %, all +4 %
#endif
#define ro_fresh_tid_ph(ph) ro_fresh_tid_<(CN == 0 || RO_MAXMT > 4 || (((CM ? 10 : RO_FRESH_MASK) >> (ph)) & 1))>()   // (CM: B/C and S2: -2 %)
    const int dh8 = (N + RO_PIECES - 1) / RO_PIECES;          // candidates per lane of a row in S1: the full row in RO_PIECES pieces (<= 128 / RO_PIECES)
    const double R2 = p.comm_radius2;
    const float R2f = (float)R2, Rf = sqrtf(R2f);
    int cur = 0;                                              // ring slot of tap 0 in XT
    int hs = 0;                                               // history slot of the CURRENT network (valid from step 1 on)
    // Steady state (every factor of every tap known as a list) runs a FUSED schedule with three workgroup barriers per
    // step instead of K + 3: gather stage 1 of step t + 1 (every tap >= 1 times the network that phase D of step t has just
    // built) rides in D2/D3's neighbour walk -- same rows, same list -- and the LAST gather stage runs inside the MLP waves
    // on their own 16 columns, so the hidden layers start without a barrier.  `s1_ready` says the previous step's phase D
    // produced this step's stage 1; steps before that (and the first step of a launch) run the stage-by-stage phase A.
    bool s1_ready = false;
    constexpr int S1T = CK ? (CK > 1 ? CK - 1 : 1) : 4;       // taps >= 1 (K <= 5)
    constexpr int GU = RoGatherUnroll<CK>::value;
    constexpr bool GTAIL = GU == 4;                           // gather passes: one of GU entries per lane, then single entries
    // Gather stage 1 of a step: x_{t-j} . A_t for every tap j >= 1, A_t given as ascending lists (rc_, rl_) with row weights wv_;
    // `curn_` = ring slot tap 0 of that step sits in.  One summation order wherever it runs (S2 of the previous step, or the
    // entry of a launch that is handed the history in factored form): lane `part` of the column's four takes list entries
    // part, part + 4, ... in order, then the quad sum (l ^ 1, l ^ 2).
    auto gather_stage1 = [&](const int gtid, const int curn, const int* rc_new, const unsigned char* rl_new, const float* w_new) {
            // one summation order for a gather stage wherever it runs: lane `part` of the column's four takes list entries
            // part, part + 4, ... in order, then the quad sum (l ^ 1, l ^ 2)
            const int c4 = gtid >> 2, part = gtid & 3;
            float s1[S1T][6];
#pragma unroll
            for (int jj = 0; jj < S1T; ++jj)
#pragma unroll
                for (int f = 0; f < 6; ++f) s1[jj][f] = 0.f;
            if (c4 < N) {
                const int cnt = rc_new[c4];
                const unsigned char* lp = rl_new + c4 * RS;
                // four entries per lane per pass (lists of up to 16 neighbours in ONE pass): the list bytes without waiting for
                // the length, then every operand read of the pass in flight together, then the multiply-adds in list order
                // (entries beyond the list are masked; every list byte is a valid row index)
                // (sized builds: ONE pass of four entries per lane -- sixteen neighbours --, then the rare longer lists one entry per
                //  lane and trip: a second full pass cost a wave ~90 instructions for its one row of 17+ neighbours.  Same order.)
                for (int e = part; e == part || (!GTAIL && e < cnt); e += 4 * GU) {
                    int jn[GU]; float gv[GU];
#pragma unroll
                    for (int u = 0; u < GU; ++u) jn[u] = lp[min(e + 4 * u, RS - 1)];
#pragma unroll
                    for (int u = 0; u < GU; ++u) gv[u] = w_new[jn[u]];
#pragma unroll
                    for (int jj = 0; jj < S1T; ++jj) {
                        if (jj < K - 1) {
                            // tap jj + 1 of step t + 1 = tap jj of step t: ring slot ro_slot(curn, jj + 1) = ro_slot(cur, jj)
                            const float* src = XT + (size_t)ro_slot(curn, jj + 1, K) * Np * 8;
                            float4 x0[GU]; float2 x1[GU];
#pragma unroll
                            for (int u = 0; u < GU; ++u) {
                                x0[u] = *reinterpret_cast<const float4*>(src + jn[u] * 8);
                                x1[u] = *reinterpret_cast<const float2*>(src + jn[u] * 8 + 4);
                            }
#pragma unroll
                            for (int u = 0; u < GU; ++u) {
                                const float gz = (e + 4 * u < cnt) ? gv[u] : 0.f;
                                s1[jj][0] = fmaf(x0[u].x, gz, s1[jj][0]); s1[jj][1] = fmaf(x0[u].y, gz, s1[jj][1]);
                                s1[jj][2] = fmaf(x0[u].z, gz, s1[jj][2]); s1[jj][3] = fmaf(x0[u].w, gz, s1[jj][3]);
                                s1[jj][4] = fmaf(x1[u].x, gz, s1[jj][4]); s1[jj][5] = fmaf(x1[u].y, gz, s1[jj][5]);
                            }
                        }
                    }
                }
                if (GTAIL) {
                    for (int e = part + 4 * GU; e < cnt; e += 4) {
                        const int jn = lp[e];
                        const float gz = w_new[jn];
#pragma unroll
                        for (int jj = 0; jj < S1T; ++jj) {
                            if (jj < K - 1) {
                                const float* src = XT + (size_t)ro_slot(curn, jj + 1, K) * Np * 8 + jn * 8;
                                const float4 x0 = *reinterpret_cast<const float4*>(src);
                                const float2 x1 = *reinterpret_cast<const float2*>(src + 4);
                                s1[jj][0] = fmaf(x0.x, gz, s1[jj][0]); s1[jj][1] = fmaf(x0.y, gz, s1[jj][1]);
                                s1[jj][2] = fmaf(x0.z, gz, s1[jj][2]); s1[jj][3] = fmaf(x0.w, gz, s1[jj][3]);
                                s1[jj][4] = fmaf(x1.x, gz, s1[jj][4]); s1[jj][5] = fmaf(x1.y, gz, s1[jj][5]);
                            }
                        }
                    }
                }
            }
            
#pragma unroll
            for (int jj = 0; jj < S1T; ++jj) {
                if (jj < K - 1) {
#pragma unroll
                    for (int f = 0; f < 6; ++f) { s1[jj][f] += dpp_f<0xB1>(s1[jj][f]); s1[jj][f] += dpp_f<0x4E>(s1[jj][f]); }
                    if (part == 0 && c4 < N) {
                        if (jj == 0) {                        // tap 1: stage 1 is its only factor -> B-fragment slot
#pragma unroll
                            for (int f = 0; f < 6; ++f) act[c4 * RO_CS + rpos(f * K + 1)] = s1[0][f];
                        } else {                              // taps >= 2: running product for stage 2 (buffer parity of q = 1)
                            float* dst = VB + ((size_t)(K - 2) + (jj - 1)) * Np * 8 + c4 * 8;
                            *reinterpret_cast<float4*>(dst) = make_float4(s1[jj][0], s1[jj][1], s1[jj][2], s1[jj][3]);
                            *reinterpret_cast<float2*>(dst + 4) = make_float2(s1[jj][4], s1[jj][5]);
                        }
                    }
                }
            }
    };
    if (enter_carry && K >= 2) {
        // a launch that is handed the history in factored form runs gather stage 1 of its FIRST step here, from the lists the
        // entry has just built (slot 0 = the current network), and so starts in the fused schedule: no stage-by-stage phase A,
        // two barriers less on the first step (round 3: 7.8 us for the first step of a launch, 5.0 us for the others)
        if (tid < 4 * ((N + 15) & ~15)) gather_stage1(tid, 0, rcnt, rlist, wrow);
        s1_ready = true;
        __syncthreads();
    }
    RO_WALL(1);

    for (int t = 0; t < T; ++t) {
        RO_STAMP(0);
        unsigned long long* rm_new = rowmask;                 // bits of the network of the step being simulated
        const int hsn = (hs + 1 == H) ? 0 : hs + 1;           // its history slot (the oldest network's, overwritten in D2)
        unsigned char* rl_new = rlist + (size_t)hsn * N * RS;
        int* rc_new = rcnt + hsn * N;
        float* w_new = wrow + hsn * N;
        // -------------------------------------------------------------- A: aggregation, power-iterated along the lists
        // y_j(t) = x_{t-j} . A_t . A_{t-1} ... A_{t-j+1}   (state_with_delay.py:44-47: G_j(t) = A_t G_{j-1}(t-1), G_0 = I)
        // evaluated left to right: K - 1 gather stages, stage q multiplies the running (6 x N) product of every tap j >= q
        // by A_{t-q+1}.  A is w_m x (symmetric 0/1 pattern), so (v . A)[f, n] = sum over n's neighbour list of w_m v[f, m]:
        // 6 N deg multiply-adds per tap and stage, where the dense operator costs 6 N^2 per tap plus N^2 deg per slice to
        // maintain.  No dense slice exists inside the launch.  The networks of the launch's own steps are known as lists;
        // products that reach back before the launch (the first K - 1 steps) end with one dense multiplication by the
        // caller's slice G_{j-hv}(t0), read from HBM.
        const int hv = min(t + t_off, K - 1);                 // networks available as lists: A_t .. A_{t-hv+1}
        const bool fused = s1_ready;                          // (implies hv == K - 1)
        {   // ---- phase A (stage by stage; empty in the fused steady state of K <= 3)
        const int tid = ro_fresh_tid_ph(0);
        const int fr = tid >> 2, fq = tid & 3;                //   gather stages: column fr, lane fq of 4
        // One summation order for a gather stage wherever it runs (here, inside the MLP waves, inside phase D): four lanes
        // per column, lane `part` takes list entries part, part + 4, ... in order, quad sum (l ^ 1, then l ^ 2) -- so a step
        // computes the same bits whether it is the first of a launch or deep inside one.
        for (int q = fused ? 2 : 1; q <= (fused ? K - 2 : hv); ++q) {
            float sa[S1T][6];
#pragma unroll
            for (int jj = 0; jj < S1T; ++jj)
#pragma unroll
                for (int f = 0; f < 6; ++f) sa[jj][f] = 0.f;
            int hq = hs - (q - 1); hq = hq < 0 ? hq + H : hq;   // slot of A_{t-q+1}
            if (fr < N) {
                const int cnt = rcnt[hq * N + fr];
                const unsigned char* lp = rlist + ((size_t)hq * N + fr) * RS;
                const float* wq = wrow + hq * N;
                for (int e = fq; e < cnt; e += 4) {
                    const int m = lp[e];
                    const float gv = wq[m];
#pragma unroll
                    for (int jj = 0; jj < S1T; ++jj) {
                        if (q + jj <= K - 1) {                // tap j = q + jj
                            const float* src = ((q == 1) ? XT + (size_t)ro_slot(cur, q + jj, K) * Np * 8
                                                         : VB + ((size_t)((q - 1) & 1) * (K - 2) + (q + jj - 2)) * Np * 8) + m * 8;
                            const float4 x0 = *reinterpret_cast<const float4*>(src);
                            const float2 x1 = *reinterpret_cast<const float2*>(src + 4);
                            sa[jj][0] = fmaf(x0.x, gv, sa[jj][0]); sa[jj][1] = fmaf(x0.y, gv, sa[jj][1]);
                            sa[jj][2] = fmaf(x0.z, gv, sa[jj][2]); sa[jj][3] = fmaf(x0.w, gv, sa[jj][3]);
                            sa[jj][4] = fmaf(x1.x, gv, sa[jj][4]); sa[jj][5] = fmaf(x1.y, gv, sa[jj][5]);
                        }
                    }
                }
            }
#pragma unroll
            for (int jj = 0; jj < S1T; ++jj) {
                if (q + jj <= K - 1) {
#pragma unroll
                    for (int f = 0; f < 6; ++f) { sa[jj][f] += dpp_f<0xB1>(sa[jj][f]); sa[jj][f] += dpp_f<0x4E>(sa[jj][f]); }
                    if (fq == 0 && fr < N) {
                        if (jj == 0) {                        // the tap's last factor: result in MFMA B-fragment order
#pragma unroll
                            for (int f = 0; f < 6; ++f) act[fr * RO_CS + rpos(f * K + q)] = sa[0][f];
                        } else {
                            float* dst = VB + ((size_t)(q & 1) * (K - 2) + (q + jj - 2)) * Np * 8 + fr * 8;
                            *reinterpret_cast<float4*>(dst) = make_float4(sa[jj][0], sa[jj][1], sa[jj][2], sa[jj][3]);
                            *reinterpret_cast<float2*>(dst + 4) = make_float2(sa[jj][4], sa[jj][5]);
                        }
                    }
                }
            }
            if (q == 1) RO_STAMP(6);
            if (fused || q < hv || hv < K - 1) __syncthreads();   // (the last stage of an unfused step shares A's closing barrier)
        }
        if (hv < K - 1) {
            // taps j > hv: the product so far (x_{t-j} itself on the launch's first step) times the caller's dense slice j - hv,
            // column gn; the two parity lanes take alternate rows, ten HBM reads in flight each
            //   dense tail: thread (column gn, tap offset gt, parity gq of the row)
            const int gq = tid & 1, gn = (tid >> 1) % N, gt = (tid >> 1) / N;
            const bool on = gt < K - 1 && gt + 1 > hv;
            const int j = gt + 1;
            float sa[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (on) {
                const float* src = (hv == 0) ? XT + (size_t)ro_slot(cur, j, K) * Np * 8
                                             : VB + ((size_t)(hv & 1) * (K - 2) + (j - 2)) * Np * 8;
                const float* gcol = Gb + (size_t)(j - hv) * NN + gn;
                for (int m = gq; m < N; m += 20) {
                    float gv[10];
#pragma unroll
                    for (int u = 0; u < 10; ++u) gv[u] = (m + 2 * u < N) ? gcol[(size_t)(m + 2 * u) * N] : 0.f;
#pragma unroll
                    for (int u = 0; u < 10; ++u) {
                        const int mm = min(m + 2 * u, N - 1);
                        const float4 x0 = *reinterpret_cast<const float4*>(src + mm * 8);
                        const float2 x1 = *reinterpret_cast<const float2*>(src + mm * 8 + 4);
                        sa[0] = fmaf(x0.x, gv[u], sa[0]); sa[1] = fmaf(x0.y, gv[u], sa[1]); sa[2] = fmaf(x0.z, gv[u], sa[2]);
                        sa[3] = fmaf(x0.w, gv[u], sa[3]); sa[4] = fmaf(x1.x, gv[u], sa[4]); sa[5] = fmaf(x1.y, gv[u], sa[5]);
                    }
                }
            }
#pragma unroll
            for (int f = 0; f < 6; ++f) sa[f] += dpp_f<0xB1>(sa[f]);
            if (on && gq == 0) {
#pragma unroll
                for (int f = 0; f < 6; ++f) act[gn * RO_CS + rpos(f * K + j)] = sa[f];
            }
        }
        if (!fused) __syncthreads();                          // fused: the stages above ended with their own barrier
        }
        RO_STAMP(1);
        {   // ---- phase B/C
        const int tid = ro_fresh_tid_ph(1);
        const int lane = tid & 63, wave = tid >> 6, li = lane & 15, lq = lane >> 4;
        // -------------------------------------------------------------- B: filter GEMM + MLP on MFMA
        // Layer metadata comes from bit-packed scalar kernel arguments: P.dims[l] indexed dynamically is re-fetched from
        // the kernel-argument segment every layer of every step (~700 cycles each).
        if (wave < NT) {                                      // wave w owns columns 16 w .. 16 w + 15 through every layer
#if RO_BC_PRIO
            // the second tile of a SIMD (waves 4 .. NT - 1 share SIMDs with waves 0 .. 2) loses every arbitration to the older wave
            // and ends ~0.9k cycles behind it: raised priority for the first part of the phase hands part of that delay to the
            // first tile, and the phase ends with the later of the two
            if (RO_BC_PRIO && wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
            if (fused && K >= 3) {
                // last gather stage (tap K - 1 times A_{t-K+2}) for the wave's own columns: four lanes per column walk its
                // list two entries at a time, DPP quad sum, straight into the B-fragment slot the first layer reads below
                // (LDS operations of one wave are ordered: no barrier)
                const int q = K - 1;
                int hq = hs - (q - 1); hq = hq < 0 ? hq + H : hq;
                const int c4 = wave * 16 + (lane >> 2), part = lane & 3;
                float sa[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                if (c4 < N) {
                    const int cnt = rcnt[hq * N + c4];
                    const unsigned char* lp = rlist + ((size_t)hq * N + c4) * RS;
                    const float* wq = wrow + hq * N;
                    const float* src = VB + ((size_t)((q - 1) & 1) * (K - 2) + (q - 2)) * Np * 8;
                    // four entries per lane per pass (lists of up to 16 neighbours in ONE pass): all list bytes first, then
                    // every operand read of the pass in flight together, then the multiply-adds in list order -- two LDS
                    // round trips per pass instead of two per entry (entries beyond the list: weight 0 on a valid row)
                    // (the first pass reads its list bytes without waiting for the length: every list byte is a valid row index,
                    //  entries beyond the list are masked at the multiply-add -- one LDS round trip less on the critical path)
                    for (int e = part; e == part || (!GTAIL && e < cnt); e += 4 * GU) {
                        int m[4];
#pragma unroll
                        for (int u = 0; u < GU; ++u) m[u] = lp[min(e + 4 * u, RS - 1)];
                        float g[4]; float4 xa[4]; float2 xb[4];
#pragma unroll
                        for (int u = 0; u < GU; ++u) {
                            g[u] = wq[m[u]];
                            xa[u] = *reinterpret_cast<const float4*>(src + m[u] * 8);
                            xb[u] = *reinterpret_cast<const float2*>(src + m[u] * 8 + 4);
                        }
#pragma unroll
                        for (int u = 0; u < GU; ++u) {
                            // entries beyond the list carry weight 0 on a valid (finite) row -- fma(x, 0, s) == s exactly -- instead
                            // of an exec-mask change per entry (-3 % per step)
                            const float gz = (e + 4 * u < cnt) ? g[u] : 0.f;
                            sa[0] = fmaf(xa[u].x, gz, sa[0]); sa[1] = fmaf(xa[u].y, gz, sa[1]); sa[2] = fmaf(xa[u].z, gz, sa[2]);
                            sa[3] = fmaf(xa[u].w, gz, sa[3]); sa[4] = fmaf(xb[u].x, gz, sa[4]); sa[5] = fmaf(xb[u].y, gz, sa[5]);
                        }
                    }
                    if (GTAIL) {
                        for (int e = part + 4 * GU; e < cnt; e += 4) {       // lists beyond sixteen entries: one entry per lane and trip
                            const int m1 = lp[e];
                            const float gz = wq[m1];
                            const float4 xa = *reinterpret_cast<const float4*>(src + m1 * 8);
                            const float2 xb = *reinterpret_cast<const float2*>(src + m1 * 8 + 4);
                            sa[0] = fmaf(xa.x, gz, sa[0]); sa[1] = fmaf(xa.y, gz, sa[1]); sa[2] = fmaf(xa.z, gz, sa[2]);
                            sa[3] = fmaf(xa.w, gz, sa[3]); sa[4] = fmaf(xb.x, gz, sa[4]); sa[5] = fmaf(xb.y, gz, sa[5]);
                        }
                    }
                }
#pragma unroll
                for (int f = 0; f < 6; ++f) { sa[f] += dpp_f<0xB1>(sa[f]); sa[f] += dpp_f<0x4E>(sa[f]); }
                if (part == 0 && c4 < N) {
#pragma unroll
                    for (int f = 0; f < 6; ++f) act[c4 * RO_CS + rpos(f * K + q)] = sa[f];
                }
                RO_STAMP(6);
            }
            const int col = wave * 16 + li;
            // hidden layers on MFMA, activations chained through registers (rollout_common.h): only the first layer reads its B
            // operand (the aggregation result) from LDS, only the last one stores its activations there (for the output layer)
            float zc[RO_MAXMT][4];
#pragma unroll
            for (int a_ = 0; a_ < RO_MAXMT; ++a_)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) zc[a_][rr] = 0.f;
            int mtp = 0;
            if constexpr (RO_X2) {
                // layer 0 (18 -> 128: eight m-tiles on the aggregation tile), then layer 1 (128 -> 128) K block by K block on the
                // accumulator registers: block kb multiplies the first layer's m-tiles 2 kb, 2 kb + 1 (this lane's eight values of
                // its k-group), split into three bf16 pieces on the fly.  Order 2, 0, 1, 3: block 2 sits in the stream buffer when
                // the phase starts; behind barrier X1 (every tile wave is through with it) waves 14 / 15 stream block 3 into the
                // buffer while blocks 0 and 1 are multiplied from their resident copies; barrier X2 stands before its first use.
                float fb[RO_KS];
                const float4* pb = reinterpret_cast<const float4*>(act + col * RO_CS + lq * RO_KS);
#pragma unroll
                for (int i = 0; i < RO_KS / 4; ++i) { const float4 tq = pb[i]; fb[4 * i] = tq.x; fb[4 * i + 1] = tq.y; fb[4 * i + 2] = tq.z; fb[4 * i + 3] = tq.w; }
                ro_layer_bf16<8, true>(fb, wl + lane * RO_WFS, wl + 8 * 64 * RO_WFS + lq * 4, zc, 1);
                RO_STAMP(12);
                f32x4 acc[8];
#pragma unroll
                for (int mt = 0; mt < 8; ++mt) {
                    const float4 bv = *reinterpret_cast<const float4*>(x2_bias1 + mt * 16 + lq * 4);
                    acc[mt] = f32x4{bv.x, bv.y, bv.z, bv.w};
                }
#define RO_X2_BLOCK(KB_, BASE_) do { \
                    const float xk[8] = {zc[2 * (KB_)][0], zc[2 * (KB_)][1], zc[2 * (KB_)][2], zc[2 * (KB_)][3], \
                                         zc[2 * (KB_) + 1][0], zc[2 * (KB_) + 1][1], zc[2 * (KB_) + 1][2], zc[2 * (KB_) + 1][3]}; \
                    ro_bf16x8 b1_, b2_, b3_; \
                    ro_split3(xk, b1_, b2_, b3_); \
                    ro_x2_block((BASE_) + lane * 12, b1_, b2_, b3_, acc); } while (0)
                RO_X2_BLOCK(2, x2_buf);
                __syncthreads();                              // X1
                RO_X2_BLOCK(0, x2_b0);
                RO_X2_BLOCK(1, x2_b0 + RO_X2_BLK);
                __syncthreads();                              // X2
                RO_X2_BLOCK(3, x2_buf);
#undef RO_X2_BLOCK
#pragma unroll
                for (int mt = 0; mt < 8; ++mt)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) zc[mt][rr] = tanh_fast(acc[mt][rr]);
                mtp = 8;
                RO_STAMP(13);
            }
            if constexpr (RO_XD) {
                // layer 0 on the aggregation tile, then every further hidden layer K block by K block from the ring: block g = 4 (l - 1)
                // + kb sits in slot g % 3; barrier B_g (g >= 1) says that it has landed AND that every tile wave is through with block
                // g - 1, whose slot the DMA waves then refill with block g + 2
                float fb[RO_KS];
                const float4* pb = reinterpret_cast<const float4*>(act + col * RO_CS + lq * RO_KS);
#pragma unroll
                for (int i = 0; i < RO_KS / 4; ++i) { const float4 tq = pb[i]; fb[4 * i] = tq.x; fb[4 * i + 1] = tq.y; fb[4 * i + 2] = tq.z; fb[4 * i + 3] = tq.w; }
                ro_layer_bf16<8, true>(fb, wl + lane * RO_WFS, wl + 8 * 64 * RO_WFS + lq * 4, zc, 1);
                RO_STAMP(12);
#pragma unroll 1
                for (int l = 1; l < n_layers - 1; ++l) {
                    f32x4 acc[8];
#pragma unroll
                    for (int mt = 0; mt < 8; ++mt) {
                        const float4 bv = *reinterpret_cast<const float4*>(xd_bias + (l - 1) * 128 + mt * 16 + lq * 4);
                        acc[mt] = f32x4{bv.x, bv.y, bv.z, bv.w};
                    }
#define RO_XD_BLOCK(KB_) do { \
                        const int g_ = 4 * (l - 1) + (KB_); \
                        if (g_ >= 1) __syncthreads(); \
                        const float xk[8] = {zc[2 * (KB_)][0], zc[2 * (KB_)][1], zc[2 * (KB_)][2], zc[2 * (KB_)][3], \
                                             zc[2 * (KB_) + 1][0], zc[2 * (KB_) + 1][1], zc[2 * (KB_) + 1][2], zc[2 * (KB_) + 1][3]}; \
                        ro_bf16x8 b1_, b2_, b3_; \
                        ro_split3(xk, b1_, b2_, b3_); \
                        ro_x2_block<1>(xd_ring + (g_ % 3) * RO_X2_BLK + lane * 12, b1_, b2_, b3_, acc); } while (0)
                    RO_XD_BLOCK(0); RO_XD_BLOCK(1); RO_XD_BLOCK(2); RO_XD_BLOCK(3);
#undef RO_XD_BLOCK
#pragma unroll
                    for (int mt = 0; mt < 8; ++mt)
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) zc[mt][rr] = tanh_fast(acc[mt][rr]);
                }
                mtp = 8;
                RO_STAMP(13);
            }
#pragma unroll
            for (int l = 0; l < (RO_XS ? 0 : (CM ? 2 : n_layers - 1)); ++l) {
                const int cout = CM ? 32 : ro_dim(dimsA, dims8, l + 1);
                const int MT = CM ? 2 : ro_mt(cout);
                // (CM: a 32-wide hidden layer's block is 2 m-tiles of fragments + 32 bias values)
                const float* wfrag = wl + (CM ? l * (2 * 64 * WFS + 32) : (int)((((l < 4) ? woffA : woffB) >> (16 * (l & 3))) & 0xFFFFull));
                float fb[RO_KS];
                int ksteps;
                if (l == 0) {
                    const float4* pb = reinterpret_cast<const float4*>(act + col * RO_CS + lq * RO_KS);
#pragma unroll
                    for (int i = 0; i < RO_KS / 4; ++i) { const float4 tq = pb[i]; fb[4 * i] = tq.x; fb[4 * i + 1] = tq.y; fb[4 * i + 2] = tq.z; fb[4 * i + 3] = tq.w; }
                    ksteps = pad4(FK) / 4;
                } else {
#pragma unroll
                    for (int s_ = 0; s_ < RO_KS; ++s_) fb[s_] = zc[s_ >> 2][s_ & 3];
                    ksteps = 4 * mtp;
                }
                const float* pw = wfrag + lane * WFS;
                const float* pbias = wfrag + MT * 64 * WFS + lq * 4;
                if constexpr (WBF) {
                    // widths <= 32: split-bf16 MFMA (rollout_common.h), the whole K = 32 in one instruction per product (no k-step count)
                    const int nkb = (RO_KB == 2 && l > 0 && ro_dim(dimsA, dims8, l) > 32) ? 2 : 1;   // K blocks of this layer's input
                    if (RO_MAXMT >= 8 && MT == 8) ro_layer_bf16<(RO_MAXMT >= 8 ? 8 : 1), true>(fb, pw, pbias, zc, nkb);
                    else if (RO_MAXMT >= 4 && MT == 4) ro_layer_bf16<(RO_MAXMT >= 4 ? 4 : 1), true>(fb, pw, pbias, zc, nkb);
                    else if (MT == 2) ro_layer_bf16<(RO_MAXMT >= 2 ? 2 : 1), true>(fb, pw, pbias, zc, nkb);
                    else ro_layer_bf16<1, true>(fb, pw, pbias, zc, nkb);
                } else {
                if (RO_MAXMT >= 8 && MT == 8) ro_layer_regs<(RO_MAXMT >= 8 ? 8 : 1), true>(fb, pw, pbias, ksteps, zc);
                else if (MT == 4) ro_layer_regs<(RO_MAXMT >= 4 ? 4 : 1), true>(fb, pw, pbias, ksteps, zc);
                else if (MT == 2) ro_layer_regs<2, true>(fb, pw, pbias, ksteps, zc);
                else ro_layer_regs<1, true>(fb, pw, pbias, ksteps, zc);
                }
                mtp = MT;
                RO_STAMP(12 + l);
#if RO_BC_PRIO
                if (l + 1 == RO_BC_PRIO && wave >= 4) __builtin_amdgcn_s_setprio(0);
#endif
            }
#if RO_BC_PRIO
            if (RO_BC_PRIO > 2 && wave >= 4) __builtin_amdgcn_s_setprio(0);
#endif
            // ---------------------------------------------------------- C: output layer + integration, same wave, no barrier
            const int lo_ = n_layers - 1;
            const float* w2 = RO_XD ? xd_out : RO_X2 ? x2_out
                                    : wl + (CM ? 2 * (2 * 64 * WFS + 32) : (int)((((lo_ < 4) ? woffA : woffB) >> (16 * (lo_ & 3))) & 0xFFFFull));
            if (CM || n_layers > 1) {
                // The 2-wide output layer on the accumulator registers of the last hidden layer: lane (li, lq) holds channels
                // c = 16 a + 4 lq + rr of column li, whose weight pairs (W[0][c], W[1][c]) are two 16-byte reads per m-tile;
                // packed-FMA partials, then the four row groups (lq) of the wave are added with two lane swaps
                // (v_permlane16_swap / v_permlane32_swap: rows 0+1 | 2+3, then halves) -- every lane of a column ends with the
                // same (ux, uy), nothing goes through LDS.  Then the agent is integrated by TWO lanes, the x axis by lane
                // (li, 0) and the y axis by lane (li, 1): the spec's per-axis expression tree (fp64, bit-exact given the
                // action), half the dependent chain.  (History: activations stored to LDS and re-read by four lanes per column,
                // 8 channels each, measured 1.35k cycles for this layer; one zero-padded MFMA m-tile 1.5k.)
                const int axis = lq;                            // 0: x, 1: y, 2 / 3: idle
                f32x2 u2 = {0.f, 0.f}, u2b = {0.f, 0.f};
#pragma unroll
                for (int a_ = 0; a_ < RO_MAXMT; ++a_) {
                    if (a_ < mtp) {
                        const float4 wa = *reinterpret_cast<const float4*>(w2 + 2 * (16 * a_ + 4 * lq));
                        const float4 wb = *reinterpret_cast<const float4*>(w2 + 2 * (16 * a_ + 4 * lq) + 4);
                        u2 = __builtin_elementwise_fma((f32x2){zc[a_][0], zc[a_][0]}, (f32x2){wa.x, wa.y}, u2);
                        u2b = __builtin_elementwise_fma((f32x2){zc[a_][1], zc[a_][1]}, (f32x2){wa.z, wa.w}, u2b);
                        u2 = __builtin_elementwise_fma((f32x2){zc[a_][2], zc[a_][2]}, (f32x2){wb.x, wb.y}, u2);
                        u2b = __builtin_elementwise_fma((f32x2){zc[a_][3], zc[a_][3]}, (f32x2){wb.z, wb.w}, u2b);
                    }
                }
                u2 = u2 + u2b;
                const float2 bb = *reinterpret_cast<const float2*>(w2 + 2 * RO_OUTC);
                const float ux = rows_sum4(u2.x) + bb.x, uy = rows_sum4(u2.y) + bb.y;
                RO_STAMP(14);
                const bool agent = axis < 2 && col < N;
                float sc = 0.f;                                 // fp32 coordinate relative to cref (this lane's axis)
                if (agent) {
                    float ua = axis ? uy : ux;
                    if (CL && (unsigned long long)dagger_coin(cl.seed, coin_ep, (unsigned int)(cl.age0 + t)) < coin_thr)
                        ua = uexp[axis * N + col];              // the expert drives this step (gnn_dagger.py:157-158)
                    uact[axis * N + col] = ua;
                    double pp = spx[axis * N + col], vv = spx[(2 + axis) * N + col];   // (px | py), (vx | vy): [4][N] doubles
                    const double cc = cref[axis];
                    double ue = 0.0;
                    if (col >= p.n_leaders) ue = clipd((double)ua, -p.max_accel, p.max_accel) * p.action_gain;
                    pp = (pp + vv * p.dt) + ((ue * p.dt) * p.dt) * 0.5;      // integrate_one, one axis
                    vv = vv + ue * p.dt;
                    spx[axis * N + col] = pp; spx[(2 + axis) * N + col] = vv;
                    sc = (float)(pp - cc);
                    reinterpret_cast<float*>(sxy)[4 * col + axis] = sc;
                }
                RO_STAMP(15);
            } else {
                // no hidden layer: the output layer reads the aggregation tile from LDS; lane L takes agent column L >> 2 of the
                // wave's tile and the 8 channels c = 4 s + (L & 3), the four partial sums of a column are added by DPP
                const int ccol = wave * 16 + (lane >> 2), cg = lane & 3;
                const float* zsrc = act + ccol * RO_CS + cg * RO_KS;
                float zo[RO_KS];
#pragma unroll
                for (int i = 0; i < RO_KS / 4; ++i) {
                    const float4 zq = *reinterpret_cast<const float4*>(zsrc + 4 * i);
                    zo[4 * i] = zq.x; zo[4 * i + 1] = zq.y; zo[4 * i + 2] = zq.z; zo[4 * i + 3] = zq.w;
                }
                double px = 0.0, py = 0.0, vx = 0.0, vy = 0.0, cx = 0.0, cy = 0.0;
                const bool agent = (cg == 0) && ccol < N;
                if (agent) { px = spx[ccol]; py = spy[ccol]; vx = svx[ccol]; vy = svy[ccol]; cx = cref[0]; cy = cref[1]; }
                f32x2 u2 = {0.f, 0.f}, u2b = {0.f, 0.f};
#pragma unroll
                for (int s_ = 0; s_ < RO_KS; s_ += 2) {           // channel c = 4 s + cg: weights (W[0][c], W[1][c]) at w2[2 c]
                    const float2 wa = *reinterpret_cast<const float2*>(w2 + 2 * (4 * s_ + cg));
                    const float2 wb = *reinterpret_cast<const float2*>(w2 + 2 * (4 * (s_ + 1) + cg));
                    u2 = __builtin_elementwise_fma((f32x2){zo[s_], zo[s_]}, (f32x2){wa.x, wa.y}, u2);
                    u2b = __builtin_elementwise_fma((f32x2){zo[s_ + 1], zo[s_ + 1]}, (f32x2){wb.x, wb.y}, u2b);
                }
                u2 = u2 + u2b;
                float ux = u2.x, uy = u2.y;
                ux += dpp_f<0xB1>(ux); uy += dpp_f<0xB1>(uy);
                ux += dpp_f<0x4E>(ux); uy += dpp_f<0x4E>(uy);
                if (agent) {
                    const float2 bb = *reinterpret_cast<const float2*>(w2 + 2 * RO_OUTC);
                    ux += bb.x; uy += bb.y;
                    if (CL && (unsigned long long)dagger_coin(cl.seed, coin_ep, (unsigned int)(cl.age0 + t)) < coin_thr) {
                        ux = uexp[ccol]; uy = uexp[N + ccol];
                    }
                    uact[ccol] = ux; uact[N + ccol] = uy;
                    const float ub[2] = {ux, uy};
                    integrate_one(px, py, vx, vy, ub, 1, ccol < p.n_leaders, p);
                    spx[ccol] = px; spy[ccol] = py; svx[ccol] = vx; svy[ccol] = vy;
                    const float sx = (float)(px - cx), sy = (float)(py - cy);
                    sxy[ccol] = make_float4(sx, sy, sx * sx + sy * sy, 0.f);
                }
            }
        } else {
            if (RO_XD) {
                // the other waves keep the ring turning: block 2 first (slot 2 is free), then at every barrier B_g the DMA waves first wait
                // for block g (block g + 1 may still be in flight: 12 requests per wave) and behind it refill block g - 1's slot
                const int wu = __builtin_amdgcn_readfirstlane(wave);
                const bool dmaw = wu >= RO_WAVES - 2;
                if (dmaw && xd_nb > 2) xd_dma(2, wu, lane);
                for (int g = 1; g < xd_nb; ++g) {
                    if (dmaw && g >= 2) {
                        if (g + 1 < xd_nb) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
                    __syncthreads();
                    if (dmaw && g + 2 < xd_nb) xd_dma(g + 2, wu, lane);
                }
            }
            if (RO_X2) {
                __syncthreads();                              // X1: every tile wave is through with block 2
                const int wu = __builtin_amdgcn_readfirstlane(wave);
                if (wu >= RO_WAVES - 2) {                     // waves 14, 15: 12 requests of 1 KB each
                    const unsigned char* src = reinterpret_cast<const unsigned char*>(x2_img_b2 + RO_X2_BLK);
                    unsigned char* dst = reinterpret_cast<unsigned char*>(x2_buf);
                    for (int c = (wu - (RO_WAVES - 2)) * 1024; c < RO_X2_BLK * 4; c += 2048) ro_lds_dma16(src + c + lane * 16, dst + c);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __syncthreads();                              // X2: block 3 is in the buffer
            }
            if (VL && __builtin_amdgcn_readfirstlane(vflag[0]) == RO_VM_REBUILD) {
                // S1 of this step rebuilds the candidate lists: the waves without columns pad every row with the sentinel index
                const int it0 = tid - NT * 64, nth = RO_THREADS - NT * 64;
                const unsigned int sw = 0x01010101u * (unsigned int)N;
                uint4* vz = reinterpret_cast<uint4*>(vlist);
                for (int i = it0; i < (N * RO_VSTR + 15) / 16; i += nth) vz[i] = make_uint4(sw, sw, sw, sw);
            }
            if (CL) {
                // meanwhile the other waves file the state this step starts from (reference gnn_dagger.py:178: the transition
                // stores the state BEFORE the step and the expert's action for it): features = tap 0 of the delay line, the
                // bits of its network (still in rowmask), the label phase D3 of the previous step left in uexp, its age
                const int it0 = tid - NT * 64, nth = RO_THREADS - NT * 64;
                const size_t fs = (size_t)((cl.ring_step0 + t) % cl.ring_steps) * gridDim.x + b;
                float* ff = cl.feat + fs * 6 * N;
                for (int e = it0; e < 6 * N; e += nth) { const int f = e / N, n = e - f * N; ff[e] = XT[((size_t)cur * Np + n) * 8 + f]; }
                unsigned long long* fb = cl.bits + fs * 2 * N;
                if (VL) {
                    // (the cheap pass of S1 writes lists only: the bits of the current network are folded back from its list)
                    for (int row = it0; row < N; row += nth) {
                        const unsigned char* lq_ = rlist + ((size_t)hs * N + row) * RS;
                        const int cq_ = rcnt[hs * N + row];
                        unsigned long long lo = 0ull, hi = 0ull;
                        for (int e = 0; e < cq_; ++e) { const int m = lq_[e]; if (m < 64) lo |= 1ull << m; else hi |= 1ull << (m - 64); }
                        fb[2 * row] = lo; fb[2 * row + 1] = hi;
                    }
                } else
                for (int i = it0; i < 2 * N; i += nth) fb[i] = rowmask[i];
                float* fl = cl.label + fs * 2 * N;
                for (int e = it0; e < 2 * N; e += nth) fl[e] = uexp[e];
                if (it0 == 0) cl.age[fs] = cl.age0 + t;
            }
        }
        }
        // how S1 of this step finds the new network (VL builds; written in S2 of the previous step): requested in front of the
        // barrier, so that the round trip is not the first thing S1 waits for
        const int vmode_raw = VL ? vflag[0] : (int)RO_VM_FULL;
        __syncthreads();
        RO_STAMP(3);
        if (RO_XD && t + 1 < T) {                             // the next step's blocks 0, 1 -> ring slots 0, 1 (all tile waves are through)
            const int wu = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
            if (wu >= RO_WAVES - 2) { xd_dma(0, wu, (int)(threadIdx.x & 63)); xd_dma(1, wu, (int)(threadIdx.x & 63)); }
        }
        if (RO_X2 && t + 1 < T) {
            // the next step's block 2 -> stream buffer (every tile wave is through with block 3), by the two waves that have least to
            // do in the simulator phases; waited for in front of the step's last barrier
            const int wu = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
            if (wu >= RO_WAVES - 2) {
                const unsigned char* src = reinterpret_cast<const unsigned char*>(x2_img_b2);
                unsigned char* dst = reinterpret_cast<unsigned char*>(x2_buf);
                const int ln = (int)(threadIdx.x & 63);
                for (int c = (wu - (RO_WAVES - 2)) * 1024; c < RO_X2_BLK * 4; c += 2048) ro_lds_dma16(src + c + ln * 16, dst + c);
            }
        }
        // -------------------------------------------------------------- S1: membership bits + neighbour lists of the new state
        // reward (spec section 4: two-pass population variance of the velocities), one wave, split around the S1 barrier so
        // that its serial fp64 chain is not what the barrier waits for: the sums here, the variance pass in S2
        double rw_mx = 0.0, rw_my = 0.0;
        const int vmode = VL ? __builtin_amdgcn_readfirstlane(vmode_raw) : (int)RO_VM_FULL;
        {   // ---- phase S1
        const int tid = ro_fresh_tid_ph(2);
        const int lane = tid & 63, wave = tid >> 6;
        const int piece = tid % RO_PIECES;                    // membership: piece of the row's candidates
        // (T512: 64 rows per pass, two passes at N = 100; the reward wave sums the velocities in the pass it has no rows in)
        constexpr int S1_ROWS = RO_THREADS / RO_PIECES;
        for (int rb = 0; rb < (RO_T512 ? N : 1); rb += S1_ROWS) {
        const int pi = rb + tid / RO_PIECES;                  // agent row
        if (wave == RO_WAVES - 1 && (rewards != nullptr || CL) && (!RO_T512 || rb + S1_ROWS >= N)) {
            double sx = 0.0, sy = 0.0;
            for (int i = lane; i < N; i += 64) { sx += svx[i]; sy += svy[i]; }
            sx = wave_sum_d(sx); sy = wave_sum_d(sy);
            if (CL && lane == 0) { vtot[0] = sx; vtot[1] = sy; }   // the centralised expert's velocity term (phase S2)
            rw_mx = sx / (double)N; rw_my = sy / (double)N;
        }
        // Eight lanes per row, each tests its piece (dh8 = ceil(N / 8) <= 16 candidates) of the FULL row: every unordered pair
        // is tested by both of its rows -- the fp32 expression is symmetric under i <-> j (dx, dy change sign, their squares do
        // not), so are the fp64 fallback and the fade hash, hence so are the bits -- which is what removes the cross-row
        // atomics, the clearing pass and one barrier round trip of dependent LDS traffic: the row's 128-bit word is OR-combined
        // across its eight lanes on the DPP path and every lane ends up holding it.  The same lanes then write the row's
        // ASCENDING neighbour list (each its own piece, at the offset the population count of the lower pieces gives), its
        // length and its row weight.  (History: D1 tested every unordered pair once and set both bits with LDS atomic ORs:
        // 3.1k cycles for this phase; ballots: 5.6k; the matrix pipe: 3.0k.)
        if (pi < N) {
            // |r2_fp32 - r2_exact| < band for every pair within 2R of each other (coordinates are
            // rounded once relative to cref, M = max |coordinate|); pairs farther than 2R are outside by a wide margin.
            // An infinite band (M = inf) sends every pair to the exact test.
            // The band of a pair needs a bound M on the coordinates of ITS two agents only, and only for pairs within 2R of
            // each other (farther ones miss R^2 by a wide margin): |s_j| <= |s_i| + 2R there, so M = |s_i|_inf + 2R is sound for
            // every candidate of this row -- no maximum over the flock, and a sound band gives the same final bits whatever
            // its width (pairs it cannot certify go to the exact test).
            const float4 si = sxy[pi];
            const int j0 = piece * dh8, nd = max(0, min(dh8, N - j0));          // this lane's candidates j0 .. j0 + nd - 1
            const float M = fmaxf(fabsf(si.x), fabsf(si.y)) + 2.0f * Rf;
            const float band = Rf * (16.f * M + 16.f * Rf) * 5.9604645e-8f + R2f * 1.1920929e-7f;
            const float t_in = R2f - band, t_out = R2f + band;
            // tests executed by every lane of a full-row pass: EXACTLY dh8 in the sized builds (13 at N = 100: round 3 ran two groups
            // of eight), one or two groups of eight otherwise (candidates beyond the piece re-test row N - 1 and are masked below)
            const int nt = CN ? dh8 : ((dh8 + 7) & ~7);
            constexpr int NTC = CN ? (CN + RO_PIECES - 1) / RO_PIECES : 0;
            const unsigned int tmask = (nt >= 32) ? 0xFFFFFFFFu : ((1u << nt) - 1u);
            unsigned int valid = (nd >= 32) ? 0xFFFFFFFFu : ((1u << nd) - 1u);   // candidates beyond the piece re-tested row N - 1
            const int self = pi - j0;
            if (self >= 0 && self < nd) valid &= ~(1u << self);                // (r2 = 0 is "inside": the diagonal is not a link)
            // a lane's bits -> the row's 128-bit word, OR-combined over the row's eight lanes on the DPP path (every lane gets it)
            auto row_word = [&](const unsigned int bits_, unsigned long long& flo_, unsigned long long& fhi_) {
                unsigned long long lo = 0ull, hi = 0ull;
                if (j0 < 64) {
                    lo = (unsigned long long)bits_ << j0;
                    if (j0 > 32) hi = (unsigned long long)bits_ >> (64 - j0);
                } else {
                    hi = (unsigned long long)bits_ << (j0 - 64);
                }
                unsigned int w0 = (unsigned int)lo, w1 = (unsigned int)(lo >> 32), w2_ = (unsigned int)hi, w3 = (unsigned int)(hi >> 32);
                w0 |= dpp_u<0xB1>(w0); w1 |= dpp_u<0xB1>(w1); w2_ |= dpp_u<0xB1>(w2_); w3 |= dpp_u<0xB1>(w3);       // lane ^ 1
                w0 |= dpp_u<0x4E>(w0); w1 |= dpp_u<0x4E>(w1); w2_ |= dpp_u<0x4E>(w2_); w3 |= dpp_u<0x4E>(w3);       // lane ^ 2
                if (RO_PIECES == 8) { w0 |= dpp_u<0x141>(w0); w1 |= dpp_u<0x141>(w1); w2_ |= dpp_u<0x141>(w2_); w3 |= dpp_u<0x141>(w3); }   // i -> 7 - i
                flo_ = ((unsigned long long)w1 << 32) | w0; fhi_ = ((unsigned long long)w3 << 32) | w2_;
            };
            // entries of the row's ascending list in front of this lane's piece
            auto piece_pos = [&](const unsigned long long flo_, const unsigned long long fhi_) -> int {
                if (j0 < 64) return __popcll(flo_ & ((1ull << j0) - 1ull));
                return __popcll(flo_) + __popcll(fhi_ & ((1ull << (j0 - 64)) - 1ull));
            };
            // VL builds keep no bit words in S1 (the bits of a network are folded back from its list where somebody needs them):
            // list positions come from an inclusive prefix sum of the lanes' hit counts over the row's eight lanes -- three DPP
            // steps on one register (pairs, quads, halves), two counts packed into it on rebuild steps
            auto seg8_scan = [&](const unsigned int x) -> unsigned int {
                unsigned int sc = x;
                sc += dpp_u<0xA0>(sc) & (0u - (unsigned int)(piece & 1));                  // quad_perm [0,0,2,2]: odd lanes += left neighbour
                sc += dpp_u<0x55>(sc) & (0u - (unsigned int)((piece >> 1) & 1));           // quad_perm [1,1,1,1]: lanes 2, 3 of a quad += its first pair
                const unsigned int q3 = dpp_u<0xFF>(sc);                                   // quad_perm [3,3,3,3]: the quad's total
                sc += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)q3, 0x114, 0xF, 0xA, false);   // row_shr:4 into banks 1, 3: upper quad += lower quad
                return sc;
            };
            bool exact_row = true;                            // this row takes the all-pairs exact pass
            int vc = 0;                                       // else: candidates on its Verlet list
            unsigned int im = 0u, om = 0u, cand = 0u;
            const bool vbuild = VL && vmode == RO_VM_REBUILD;
            if (vbuild) {
                static_assert(!VL || (RO_PIECES == 8 && !FD), "the candidate lists ride on the eight-lane difference-form pass");
                // the all-pairs exact pass with a third threshold: the candidate list of the row = every j that is not CLEARLY
                // farther than RV = R (1 + skin).  For pairs within 2R of each other |r2_fp32 - r2| <= band (above), RV <= 2R, and
                // RVf^2 is RV^2 to three roundings; a pair that may be within RV at this step stays on the list: r2_fp32 <= r2 + band
                // < RV^2 + band <= tv.  (A NaN / infinite band keeps every pair: the row overflows and takes the exact pass.)
                const float RVf = Rf * (1.0f + RO_VSKIN);
                const float tv = fmaf(RVf, RVf, 4.f * band + RVf * RVf * 4.e-7f);
                unsigned int omv;
                ro_s1_masks<NTC, true>(sxy, si.x, si.y, j0, N, nt, t_in, t_out, im, om, tv, &omv);
                cand = ~(__builtin_bitreverse32(omv) >> (32 - nt)) & tmask & valid;
            } else {
                if (VL && vmode == RO_VM_CHEAP) { vc = vcnt[pi]; exact_row = vc > RO_VCAP; }
                if (exact_row) ro_s1_masks<NTC, false>(sxy, si.x, si.y, j0, N, nt, t_in, t_out, im, om);
            }
            if (exact_row) {
            unsigned int in_m = __builtin_bitreverse32(im) >> (32 - nt);          // test k -> bit k
            unsigned int unc_m = ~(__builtin_bitreverse32(om) >> (32 - nt)) & ~in_m & tmask;
            RO_STAMP(16);
            in_m &= valid;
            unc_m &= valid;
            while (unc_m) {                                   // rare: the spec's own fp64 expression decides
                const int q = __builtin_ctz(unc_m);
                unc_m &= unc_m - 1u;
                const int j = j0 + q;
                const double dx = spx[pi] - spx[j], dy = spy[pi] - spy[j];
                const double r2 = dx * dx + dy * dy;
                if (r2 < R2) in_m |= 1u << q;
            }
            if (FD && p.link_drop != 0u) {                    // FlockingStochastic-v0: links that are faded this step
                unsigned int mq = in_m;
                const unsigned int wi = fade_word(spx[pi], spy[pi]);
                while (mq) {
                    const int q = __builtin_ctz(mq);
                    mq &= mq - 1u;
                    const int j = j0 + q;
                    if (!link_up(p, pi, j, N, wi, fade_word(spx[j], spy[j]))) in_m &= ~(1u << q);
                }
            }
            RO_STAMP(17);
            if (VL) {
                const unsigned int own = (unsigned int)__popc(in_m) | (vbuild ? (unsigned int)__popc(cand) << 16 : 0u);
                const unsigned int incl = seg8_scan(own);
                RO_STAMP(18);
                int pos = (int)((incl - own) & 0xFFFFu);
                unsigned char* lp = rl_new + pi * RS;
                unsigned int mq = in_m;
                while (mq) { lp[pos++] = (unsigned char)(j0 + __builtin_ctz(mq)); mq &= mq - 1u; }
                if (vbuild) {
                    // the candidate list, ascending; a row beyond the capacity keeps a truncated list nobody reads (its count says so:
                    // exact pass every step)
                    int vpos = (int)((incl - own) >> 16);
                    unsigned char* vp = vlist + pi * RO_VSTR;
                    unsigned int cq = cand;
                    while (cq) { if (vpos < RO_VCAP) vp[vpos] = (unsigned char)(j0 + __builtin_ctz(cq)); ++vpos; cq &= cq - 1u; }
                }
                RO_STAMP(19);
                if (piece == RO_PIECES - 1) {                 // the last lane of the row holds the totals
                    const int cnt = (int)(incl & 0xFFFFu);
                    rc_new[pi] = cnt;
                    w_new[pi] = wtab[cnt];
                    if (vbuild) vcnt[pi] = (int)(incl >> 16);
                }
            } else {
            // this lane's bits at their place in the row's 128-bit word
            unsigned long long flo, fhi;
            row_word(in_m, flo, fhi);
            RO_STAMP(18);
            int pos = piece_pos(flo, fhi);
            unsigned char* lp = rl_new + pi * RS;
            unsigned int mq = in_m;
            while (mq) { lp[pos++] = (unsigned char)(j0 + __builtin_ctz(mq)); mq &= mq - 1u; }
            RO_STAMP(19);
            if (piece == 0) {
                const int cnt = __popcll(flo) + __popcll(fhi);
                rm_new[2 * pi] = flo; rm_new[2 * pi + 1] = fhi;
                rc_new[pi] = cnt;
                w_new[pi] = wtab[cnt];                          // (float)(1 / max(deg, 1)) or 1: the value the spec's row weight rounds to
            }
            }
            } else {
                // ---- cheap pass: the row's listed candidates only.  Lane `piece` of the row takes entries piece, piece + 8, ... (the
                // padding entries point at the sentinel: clearly outside); a ballot per pass holds the hits of the wave's eight rows,
                // one byte per row, so the row's hit mask over list positions needs no cross-lane moves; a hit goes to position
                // (hits in front of it) of the ascending neighbour list -- the candidate list is ascending, so is every subset.
                // The bits are the exact pass's: same fp32 expression, same band, same fp64 fallback.
                const unsigned char* vl = vlist + pi * RO_VSTR + piece;
                unsigned char* lp = rl_new + pi * RS;
                unsigned char* dummy = reinterpret_cast<unsigned char*>(vflag) + 48 + piece;  // masked writes land here
                int base = 0;
                const bool rowhi = (tid & 32) != 0;           // this row's byte: in the high half of a ballot?
                const int shb = tid & 24;                     // ... and its bit offset inside that half
                for (int e0 = 0; e0 < vc; e0 += 8 * RO_VPU) {
                    int jv[RO_VPU]; float2 sj[RO_VPU];
                    unsigned long long bin[RO_VPU], bunc = 0ull;
#pragma unroll
                    for (int u = 0; u < RO_VPU; ++u) jv[u] = vl[e0 + 8 * u];
#pragma unroll
                    for (int u = 0; u < RO_VPU; ++u) sj[u] = *reinterpret_cast<const float2*>(&sxy[jv[u]]);
#pragma unroll
                    for (int u = 0; u < RO_VPU; ++u) {
                        const float dx = si.x - sj[u].x, dy = si.y - sj[u].y;
                        const float r2 = fmaf(dy, dy, dx * dx);
                        bin[u] = __builtin_amdgcn_ballot_w64(r2 < t_in);
                        bunc |= __builtin_amdgcn_ballot_w64(!(r2 > t_out)) & ~bin[u];     // (a NaN goes to the fp64 expression: "no link")
                    }
                    if (bunc != 0ull) {                       // rare, wave-uniform: the spec's own fp64 expression decides
#pragma unroll
                        for (int u = 0; u < RO_VPU; ++u) {
                            const int j = min(jv[u], N - 1);   // (sentinel entries are clearly outside: never uncertain)
                            const float dxf = si.x - sj[u].x, dyf = si.y - sj[u].y;
                            const float r2f = fmaf(dyf, dyf, dxf * dxf);
                            const double dx = spx[pi] - spx[j], dy = spy[pi] - spy[j];
                            const double r2 = dx * dx + dy * dy;
                            const bool unc = !(r2f < t_in) && !(r2f > t_out);
                            bin[u] |= __builtin_amdgcn_ballot_w64(unc && r2 < R2);
                        }
                    }
                    unsigned int mw = 0u;                     // hits of this trip over list positions 8 u + piece
#pragma unroll
                    for (int u = 0; u < RO_VPU; ++u) {
                        const unsigned int half = rowhi ? (unsigned int)(bin[u] >> 32) : (unsigned int)bin[u];
                        mw |= ((half >> shb) & 0xFFu) << (8 * u);
                    }
#pragma unroll
                    for (int u = 0; u < RO_VPU; ++u) {
                        const int pos = base + __popc(mw & ((1u << (8 * u + piece)) - 1u));
                        unsigned char* dst = ((mw >> (8 * u + piece)) & 1u) ? lp + pos : dummy;
                        *dst = (unsigned char)jv[u];
                    }
                    base += __popc(mw);
                }
                if (piece == 0) {
                    rc_new[pi] = base;
                    w_new[pi] = wtab[base];
                }
            }
        }
        }
        }
        __syncthreads();
        RO_STAMP(7);
        // -------------------------------------------------------------- S2: fp64 feature terms  ||  gather stage 1 of step t + 1
        // Two groups of 4 x pad16(N) threads (four lanes per row, whole waves) work side by side on the lists S1 has written:
        // the first sums the spec's fp64 feature terms of actual neighbours (and the expert label in collecting builds), the
        // second runs gather stage 1 of step t + 1 -- x_{t+1-j} . A_{t+1} for taps j >= 1 -- when that step will find every
        // factor as a list.  A_{t+1}[m, n] = w_new[m] on the (symmetric) pattern.  (History: one group did both inside one
        // neighbour walk, after building the list itself: 3.7 - 4.7k cycles for this phase.)
        const bool do_s1 = K >= 2 && t + 1 < T && t + 1 + t_off >= K - 1;
        const int curn = (cur + 1 == K) ? 0 : cur + 1;        // ring slot of tap 0 of step t + 1
        {   // ---- phase S2
        const int tid = ro_fresh_tid_ph(3);
        const int lane = tid & 63, wave = tid >> 6;
        const int fr = tid >> 2, fq = tid & 3;                //   features: agent row fr, lane fq of 4
        if (tid == RO_THREADS - 2) { cref[0] = spx[0]; cref[1] = spy[0]; }   // next step's reference point (any point is valid)
        const int grp = 4 * ((N + 15) & ~15);                 // threads per group
        const int vl_wave = (N <= 112 && !RO_T512) ? RO_WAVES - 2 : RO_WAVES - 1;   // the wave that keeps the Verlet books (VL builds)
        if (wave == RO_WAVES - 1 && rewards != nullptr) {     // second half of the reward (velocities change in phase C only)
            double dv = 0.0;
            for (int i = lane; i < N; i += 64) {
                const double ex = svx[i] - rw_mx, ey = svy[i] - rw_my;
                dv += ex * ex + ey * ey;
            }
            const double var = wave_sum_d(dv) / (double)N;
            if (lane == 0) rewards[(size_t)b * T + t] = -1.0 * var * p.reward_scale;
            RO_STAMP(21);
        }
        if (VL && wave == vl_wave && t + 1 < T) {
            // ---- Verlet bookkeeping (header of this file), one step AHEAD and off the critical path, by a wave that has no rows
            // in this phase (N <= 112; the reward wave otherwise): is every pair that can be within R after the NEXT integration
            // still on the candidate lists?  D_i = p_i - pref_i - c with c = V dt (steps since the rebuild) for the frame
            // velocity V chosen at the rebuild (any common vector is sound); the next state adds v_i dt - V dt and at most
            // hacc = |max_accel action_gain| dt^2 / 2 per axis.
            // A pair that was farther apart than RV at the rebuild is safe while D_i + D_j <= skin.  An ESCAPER -- an agent that has
            // left the flock and keeps its course -- outruns that bound within a few steps although it is nowhere near anybody; for
            // an agent that had NO candidate at the rebuild the wave therefore keeps gap_o = (distance to its nearest agent then)
            // - R, and every pair of o is safe while D_o + D_j <= gap_o.  Sufficient for the step: every agent has D_i <= skin / 2, or
            // D_i + max_j D_j <= gap_i (gap = 0 for agents that had candidates).
            // (raised priority: a hundred instructions on a SIMD it shares with three busy waves -- as the youngest wave it finished
            //  last, 3.3k cycles into the phase, and any instruction added to it lengthened the step)
            __builtin_amdgcn_s_setprio(3);
            double* vst = reinterpret_cast<double*>(vflag) + 2;    // frame offset (x, y), frame velocity (x, y): state of this wave, kept in LDS
            const double hacc = fabs(p.max_accel * p.action_gain) * p.dt * p.dt * 0.5;
            const float skin = RO_VSKIN * Rf;
            const float vmarg = 1.e-3f * Rf;                  // margin: the fp32 / fp64 roundings of these bounds are ~1e-6 R
            bool vl_ok = false;
            float dd[2] = {0.f, 0.f};                         // upper bounds of |D_i| after the next integration, agents lane, lane + 64
            if (vmode == RO_VM_REBUILD) {                     // S1 of this step listed the candidates of THESE positions:
                // they become the reference, the frame moves with the flock's mean velocity (any common vector is sound)
                double sx = 0.0, sy = 0.0;
                for (int i = lane; i < N; i += 64) { sx += svx[i]; sy += svy[i]; }
                const double fvx = wave_sum_d(sx) * p.dt / (double)N, fvy = wave_sum_d(sy) * p.dt / (double)N;
                unsigned long long iso[2];
#pragma unroll
                for (int a_ = 0; a_ < 2; ++a_) {
                    const int i = lane + 64 * a_;
                    bool lonely = false;
                    if (i < N) {
                        const double xi = spx[i], yi = spy[i];
                        vpref[i] = xi; vpref[N + i] = yi;
                        const double ex = fabs(svx[i] * p.dt - fvx) + hacc, ey = fabs(svy[i] * p.dt - fvy) + hacc;
                        dd[a_] = sqrtf((float)(ex * ex + ey * ey) * 1.000001f) * 1.000001f;
                        vgap[i] = 0.f;
                        lonely = vcnt[i] == 0;
                    }
                    iso[a_] = __builtin_amdgcn_ballot_w64(lonely);
                }
                // (wave-uniform loops over at most six such agents -- a flock hundreds of steps past the time limit has dozens, and
                //  this wave must not become the phase's tail; fp32 distances on the coordinates S1 used, relative to the reference
                //  point: each within 2^-24 of its magnitude, charged to the gap below)
                for (int a_ = 0, budget = 6; a_ < 2; ++a_) {
                    unsigned long long mk = iso[a_];
                    while (mk != 0ull && budget > 0) {
                        const int o = 64 * a_ + __builtin_ctzll(mk);
                        mk &= mk - 1ull; --budget;
                        const float2 so = *reinterpret_cast<const float2*>(&sxy[o]);
                        float nm = __builtin_huge_valf();
                        for (int j = lane; j < N; j += 64) {
                            const float2 sj = *reinterpret_cast<const float2*>(&sxy[j]);
                            const float dx = so.x - sj.x, dy = so.y - sj.y;
                            const float r2 = fmaf(dy, dy, dx * dx);
                            if (j != o) nm = fminf(nm, r2);
                        }
                        nm = -wave_max_to_last(-nm);
                        if (lane == 63) {
                            const float dn = sqrtf(nm);
                            vgap[o] = dn * 0.999998f - 1.e-6f * (fabsf(so.x) + fabsf(so.y) + dn) - Rf - vmarg;
                        }
                    }
                }
                if (lane == 0) { vst[0] = fvx; vst[1] = fvy; vst[2] = fvx; vst[3] = fvy; }
                vl_ok = true;
            } else if (vmode == RO_VM_CHEAP) {
                const double fvx = vst[2], fvy = vst[3];
                const double cx = vst[0] + fvx, cy = vst[1] + fvy;
#pragma unroll
                for (int a_ = 0; a_ < 2; ++a_) {
                    const int i = lane + 64 * a_;
                    if (i < N) {
                        const double ex = fabs((spx[i] + svx[i] * p.dt) - vpref[i] - cx) + hacc;
                        const double ey = fabs((spy[i] + svy[i] * p.dt) - vpref[N + i] - cy) + hacc;
                        dd[a_] = sqrtf((float)(ex * ex + ey * ey) * 1.000001f) * 1.000001f;
                    }
                }
                if (lane == 0) { vst[0] = cx; vst[1] = cy; }
                vl_ok = true;
            }
            const float dmx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wave_max_to_last(fmaxf(dd[0], dd[1]))), 63));
            bool bad = false;                                 // (comparisons written so that a NaN -- diverged episode -- falls to the exact pass)
#pragma unroll
            for (int a_ = 0; a_ < 2; ++a_) {
                const int i = lane + 64 * a_;
                if (i < N) bad = bad || !((dd[a_] <= 0.5f * (skin - vmarg)) || (dd[a_] + dmx <= vgap[i]));
            }
            const bool cheap_ok = vl_ok && __builtin_amdgcn_ballot_w64(bad) == 0ull;
            // Next step's mode.  A list that did not serve RO_VMINLIFE steps was not worth its rebuild (~0.4k cycles on top of the
            // exact pass): the flock moves too fast for this skin -- exact passes for a while, then another try.
            // (the back-off doubles, 8 .. 256 steps, while lists keep dying young, and starts over once one has served eight steps)
            int age = (vmode == RO_VM_CHEAP) ? vflag[1] + 1 : 0, backoff = vflag[2], bnext = vflag[3];
            if (age >= 8) bnext = 8;
            int next = RO_VM_CHEAP;
            if (!cheap_ok) {                                  // (wave-uniform)
                if (vl_ok && age < RO_VMINLIFE) { backoff = bnext; bnext = min(2 * bnext, 256); }
                const bool rebuild = backoff == 0 && (T - (t + 1) >= RO_VMINLIFE);
                if (backoff > 0) --backoff;
                next = rebuild ? RO_VM_REBUILD : RO_VM_FULL;
            }
            if (lane == 0) { vflag[1] = age; vflag[2] = backoff; vflag[3] = bnext; }
            if (lane == 0) vflag[0] = next;
            __builtin_amdgcn_s_setprio(0);
            RO_STAMP(24);
#ifdef MGP_RO_PROFILE
            if (blockIdx.x < 4096 && lane == 0) atomicAdd(&mgp_ro_vstat[blockIdx.x * 4 + next], 1u);
#endif
        }
        // (T512: one group of threads runs the feature pass, then the gather stage -- they are independent of each other)
        const int gtid = RO_T512 ? tid : tid - grp;           // (the groups swapped -- gather on the older waves -- measured the same)
        if (tid < grp) {
            double f0 = 0, f1 = 0, f2 = 0, f3 = 0, f4 = 0, f5 = 0;
            int cnt = 0;
            if (fr < N) {
                cnt = rc_new[fr];
                const unsigned char* lp = rl_new + fr * RS;
                const double xi = spx[fr], yi = spy[fr], vxi = svx[fr], vyi = svy[fr];
                // one neighbour's terms.  q = 1 / r2 to within an ulp: fp32 reciprocal seed (1 ulp of fp32), two Newton steps in fp64
                // -- five instructions where the correctly rounded division takes eleven (measured: 1 % of the step); the feature
                // sums already differ from the oracle's by their summation order (1e-11), the tests hold them to 1e-6
                auto term = [&](const double xj, const double yj, const double vxj, const double vyj) {
                    const double dx = xi - xj, dy = yi - yj;
                    const double r2 = dx * dx + dy * dy;
                    double q = (double)__builtin_amdgcn_rcpf((float)r2);
                    q = __builtin_fma(q, __builtin_fma(-r2, q, 1.0), q);
                    q = __builtin_fma(q, __builtin_fma(-r2, q, 1.0), q);
                    const double qq = q * q;
#if RO_FEAT_FMA
                    // [r5] fused multiply-adds for the four potential terms, and the velocity term as deg v_i - sum v_j (formed behind
                    // the loop): 23 instead of 29 fp64 instructions per neighbour; the sums move by ~1e-16 of their terms (tests: 1e-6)
                    f0 += vxj;
                    f1 = __builtin_fma(dx, qq, f1);
                    f2 = __builtin_fma(dx, q, f2);
                    f3 += vyj;
                    f4 = __builtin_fma(dy, qq, f4);
                    f5 = __builtin_fma(dy, q, f5);
#else
                    f0 += vxi - vxj;
                    f1 += dx * qq;
                    f2 += dx * q;
                    f3 += vyi - vyj;
                    f4 += dy * qq;
                    f5 += dy * q;
#endif
                };
                int e = fq;
#if RO_FEAT_PAIR
                // [r5] two entries per trip with both entries' LDS reads in flight together (a trip is a chain of two LDS round trips
                // and thirteen dependent fp64 instructions: the phase waits on latency as much as on issue); the lane's entries are
                // still added in list order
                for (; e + 4 < cnt; e += 8) {
                    const int ja = lp[e], jb = lp[e + 4];
                    const double xa = spx[ja], ya = spy[ja], vxa = svx[ja], vya = svy[ja];
                    const double xb2 = spx[jb], yb2 = spy[jb], vxb = svx[jb], vyb = svy[jb];
                    term(xa, ya, vxa, vya);
                    term(xb2, yb2, vxb, vyb);
                }
#endif
                for (; e < cnt; e += 4) {
                    const int j = lp[e];
                    term(spx[j], spy[j], svx[j], svy[j]);
                }
#if RO_FEAT_FMA
                {   // this lane's share of sum_j (v_i - v_j): (its entries) v_i - sum v_j
                    const double mine = (double)((cnt - fq + 3) >> 2);          // entries fq, fq + 4, ... < cnt
                    f0 = __builtin_fma(mine, vxi, -f0);
                    f3 = __builtin_fma(mine, vyi, -f3);
                }
#endif
            }
            RO_STAMP(8);
            f0 += dpp_d<0xB1>(f0); f1 += dpp_d<0xB1>(f1); f2 += dpp_d<0xB1>(f2);
            f3 += dpp_d<0xB1>(f3); f4 += dpp_d<0xB1>(f4); f5 += dpp_d<0xB1>(f5);
            f0 += dpp_d<0x4E>(f0); f1 += dpp_d<0x4E>(f1); f2 += dpp_d<0x4E>(f2);
            f3 += dpp_d<0x4E>(f3); f4 += dpp_d<0x4E>(f4); f5 += dpp_d<0x4E>(f5);
            if (fq == 0 && fr < N) {
                float* xn = XT + ((size_t)curn * Np + fr) * 8;     // overwrites the oldest tap (tap K - 1 of step t: not a source of stage 1)
                *reinterpret_cast<float4*>(xn) = make_float4((float)f0, (float)f1, (float)f2, (float)f3);
                *reinterpret_cast<float4*>(xn + 4) = make_float4((float)f4, (float)f5, 0.f, 0.f);
                if (CL) {
                    // expert action of the NEW state, closed form of its observation (FLOCK-SPEC section 5; the expression of
                    // flock.hip): the label of the next transition, and what drives the next step when the coin says so
                    double tvx = f0, tvy = f3;
                    if (p.centralized) { tvx = (double)N * svx[fr] - vtot[0]; tvy = (double)N * svy[fr] - vtot[1]; }
                    uexp[fr] = (float)(clipd(-tvx - (2.0 * f2 - 2.0 * f1), -p.ctrl_clip, p.ctrl_clip) * p.ctrl_gain);
                    uexp[N + fr] = (float)(clipd(-tvy - (2.0 * f5 - 2.0 * f4), -p.ctrl_clip, p.ctrl_clip) * p.ctrl_gain);
                }
                float* y0 = act + fr * RO_CS;                 // tap 0 of the next step's aggregation: G_0 = I  =>  y_0 = X_0
                y0[rpos(0 * K)] = (float)f0; y0[rpos(1 * K)] = (float)f1; y0[rpos(2 * K)] = (float)f2;
                y0[rpos(3 * K)] = (float)f3; y0[rpos(4 * K)] = (float)f4; y0[rpos(5 * K)] = (float)f5;
            }
            RO_STAMP(22);
        }
        if (do_s1 && gtid >= 0 && gtid < grp && (RO_T512 || tid >= grp)) {
#if RO_S2_PRIO
            // the gather group is dispatched behind the feature group and, as the younger half of every SIMD, ends the phase 0.5k
            // cycles after it: raised priority hands that delay to the feature waves
            __builtin_amdgcn_s_setprio(RO_S2_PRIO);
#endif
            gather_stage1(gtid, curn, rc_new, rl_new, w_new);
#if RO_S2_PRIO
            __builtin_amdgcn_s_setprio(0);
#endif
            RO_STAMP(20);
            RO_STAMP(23);
        }
        }
        if (RO_XS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the streamed block(s) of the next step have landed)
        __syncthreads();
        RO_STAMP(4);
        cur = curn;
        hs = hsn;
        s1_ready = do_s1;
        RO_STAMP(5);
        if (t < 3) RO_WALL(2 + t);
    }
    RO_WALL(5);

    // ------------------------------------------------------------------ exit: LDS -> the caller's buffers
    // Dense operator slices of the final state, one row per wave at a time:
    //     row i of G_j(T) = e_i . A_T . A_{T-1} ... (min(j, hv) networks of this launch)  [ . G_{j-hv}(t0) when j > hv ]
    // Slices are produced in descending j (a slice that is still an input -- j - hv < j -- is overwritten later), with a
    // workgroup barrier between slices.  Row vectors ping-pong in the activation area (no longer needed).
    RO_STAMPX(10);
    if (VL && T > 0 && K >= 2 && !(flags & MGP_RO_SKIP_DENSE)) {
        // the cheap pass of S1 writes lists only: fold the newest network's bits back from its list for the dense exit below
        for (int row = tid; row < N; row += RO_THREADS) {
            const unsigned char* lq_ = rlist + ((size_t)hs * N + row) * RS;
            const int cq_ = rcnt[hs * N + row];
            unsigned long long lo = 0ull, hi = 0ull;
            for (int e = 0; e < cq_; ++e) { const int m = lq_[e]; if (m < 64) lo |= 1ull << m; else hi |= 1ull << (m - 64); }
            rowmask[2 * row] = lo; rowmask[2 * row + 1] = hi;
        }
        __syncthreads();
    }
    // Factored hand-over (MGP_RO_EXIT_CARRY): bits + row weights of the last H networks, newest first.  The newest network's
    // bits are still in rowmask (cleared in phase B only); older ones are folded back from their lists.
    if (flags & MGP_RO_EXIT_CARRY) {
        unsigned long long* cb = carry + (size_t)b * cwords;
        float* cw = reinterpret_cast<float*>(cb + (size_t)H * N * 2);
        for (int it = tid; it < H * N; it += RO_THREADS) {
            const int q = it / N, row = it - q * N;
            int hq = hs - q; hq = hq < 0 ? hq + H : hq;
            unsigned long long lo = 0ull, hi = 0ull;
            if (q == 0 && !VL) { lo = rowmask[2 * row]; hi = rowmask[2 * row + 1]; }
            else {                                            // (VL: the newest network's bits exist as its list only)
                const unsigned char* lp = rlist + ((size_t)hq * N + row) * RS;
                const int cnt = rcnt[hq * N + row];
                for (int e = 0; e < cnt; ++e) {
                    const int m = lp[e];
                    if (m < 64) lo |= 1ull << m; else hi |= 1ull << (m - 64);
                }
            }
            cb[(size_t)it * 2] = lo; cb[(size_t)it * 2 + 1] = hi;
            cw[it] = wrow[hq * N + row];
        }
    }
    if (T > 0 && K >= 2 && !(flags & MGP_RO_SKIP_DENSE)) {
        const int hv = min(T + t_off, K - 1);
        float* rbuf = act + wave * 2 * Np;                    // [2][Np] per wave (16 x 2 x Np floats fit the activation area)
        for (int j = K - 1; j >= 1; --j) {
            const int nsp = min(j, hv);
            for (int i = wave; i < N; i += RO_WAVES) {
                float* r0 = rbuf;
                float* r1 = rbuf + Np;
                // e_i . A_T = row i of the newest network: w_T(i) on its membership bits (still set: cleared in phase B only)
                const float wi = wrow[hs * N + i];
                for (int n = lane; n < N; n += 64)
                    r0[n] = ((rowmask[2 * i + (n >> 6)] >> (n & 63)) & 1ull) ? wi : 0.f;
                for (int q = 2; q <= nsp; ++q) {              // . A_{T-q+1}: gather along the (symmetric) lists, source weights
                    int hq = hs - (q - 1); hq = hq < 0 ? hq + H : hq;
                    const float* wq = wrow + hq * N;
                    for (int n = lane; n < N; n += 64) {
                        const int cnt = rcnt[hq * N + n];
                        const unsigned char* lp = rlist + ((size_t)hq * N + n) * RS;
                        float s = 0.f;
                        for (int e = 0; e < cnt; ++e) { const int m = lp[e]; s = fmaf(r0[m], wq[m], s); }
                        r1[n] = s;
                    }
                    float* tsw = r0; r0 = r1; r1 = tsw;
                }
                float* grow = Gb + (size_t)j * NN + (size_t)i * N;
                if (j > hv) {                                 // dense tail with the caller's slice j - hv (rows with r0[m] != 0 only)
                    const float* gsrc = Gb + (size_t)(j - hv) * NN;
                    float s0 = 0.f, s1 = 0.f;
                    for (int m = 0; m < N; ++m) {
                        const float rv = r0[m];               // wave-uniform
                        if (rv != 0.f) {
                            if (lane < N) s0 = fmaf(rv, gsrc[(size_t)m * N + lane], s0);
                            if (lane + 64 < N) s1 = fmaf(rv, gsrc[(size_t)m * N + lane + 64], s1);
                        }
                    }
                    if (lane < N) grow[lane] = s0;
                    if (lane + 64 < N) grow[lane + 64] = s1;
                } else {
                    for (int n = lane; n < N; n += 64) grow[n] = r0[n];
                }
            }
            __syncthreads();
            RO_STAMPX(10 + (K - j));
        }
    }
    for (int e = tid; e < K * 6 * N; e += RO_THREADS) {
        const int k = e / (6 * N), r1 = e - k * 6 * N, f = r1 / N, n = r1 - f * N;
        Xb[e] = XT[((size_t)ro_slot(cur, k, K) * Np + n) * 8 + f];
    }
    for (int i = tid; i < N; i += RO_THREADS) {
        xb[i * 4 + 0] = spx[i]; xb[i * 4 + 1] = spy[i]; xb[i * 4 + 2] = svx[i]; xb[i * 4 + 3] = svy[i];
    }
    if (action != nullptr)
        for (int e = tid; e < 2 * N; e += RO_THREADS) action[(size_t)b * 2 * N + e] = uact[e];
    if (CL)
        for (int e = tid; e < 2 * N; e += RO_THREADS) cl.expert_io[(size_t)b * 2 * N + e] = uexp[e];
    RO_STAMPX(2);
    RO_WALL(6);
}

// ---------------------------------------------------------------------------------------------------------------------
// The same rollout for 128 < N <= 256 (the reference's n_twoflocks / transfer sweeps: N = 150, 200, 250; BASELINE
// configs[4]: N = 200, K = 4).  Differences from rollout_kernel, all forced by size: the networks of the last K - 1 steps
// are kept as membership BITS only (four 64-bit words per row; byte lists of three networks would be 125 KB at N = 200)
// and every consumer -- gather stages, feature pass, exit -- walks bits; rows / gather items / pair offsets are looped
// over instead of mapped one to a thread; everything is run-time sized.  Phases, barriers and arithmetic are the same.
#undef ro_fresh_tid_ph
// layers of the N > 128 kernel: split-bf16 where a layer's input is ONE K block (widths <= 32, the 128-wide build); the 64-wide
// build keeps fp32 fragments here (rollout_common.h: ro_wfs)
constexpr bool RB_BF = RO_BF16_CHAIN && RO_KB == 1;
constexpr int RB_WFS = ro_wfs(RB_BF);
struct RbOff { int pos, bits, wrow, uact, xt, vb, act, sxy, mmax, uexp, wl; };
constexpr int RB_MAXN = 256;
constexpr int RB_NW = 4;

__host__ __device__ inline RbOff rb_offsets(int N, int K)
{
    RbOff c = {};
    int off = 0;
    const int Np = (N + 3) & ~3, H = ro_hist(K);
    c.pos = ro_take(off, (4 * N + 2) * 8);
    c.bits = ro_take(off, H * N * RB_NW * 8);
    c.wrow = ro_take(off, H * N * 4);
    c.uact = ro_take(off, 2 * N * 4);
    c.xt = ro_take(off, K * Np * 8 * 4);
    c.vb = ro_take(off, 2 * (K > 2 ? K - 2 : 0) * Np * 8 * 4);
    c.act = ro_take(off, ((N + 15) & ~15) * RO_CS * 4);
    c.sxy = ro_take(off, N * 8);
    c.mmax = ro_take(off, 16);
    c.uexp = ro_take(off, 2 * N * 4 + 16);                    // expert action of the current state (collection) + velocity sums
    c.wl = off;
    return c;
}

// sum over the set bits m of `w` (base index `base`) of wq[m] * src[m][0..5]
__device__ __forceinline__ void rb_gather_word(unsigned long long w, int base, const float* wq, const float* src, float (&sa)[6])
{
    while (w) {
        const int m = base + __builtin_ctzll(w);
        w &= w - 1ull;
        const float gv = wq[m];
        const float4 x0 = *reinterpret_cast<const float4*>(src + m * 8);
        const float2 x1 = *reinterpret_cast<const float2*>(src + m * 8 + 4);
        sa[0] = fmaf(x0.x, gv, sa[0]); sa[1] = fmaf(x0.y, gv, sa[1]); sa[2] = fmaf(x0.z, gv, sa[2]);
        sa[3] = fmaf(x0.w, gv, sa[3]); sa[4] = fmaf(x1.x, gv, sa[4]); sa[5] = fmaf(x1.y, gv, sa[5]);
    }
}

// CN / CK: compile-time (N, K) of a sized instantiation (0 = run-time arguments), as in rollout_kernel: BASELINE configs[4]'s
// shape (N = 200, K = 4) runs with constant LDS addresses, loop bounds and divisors.
template <bool FD, bool CL, int CN = 0, int CK = 0>
__global__ __launch_bounds__(RO_THREADS)
void rollout_big_kernel(double* __restrict__ x, float* __restrict__ G, float* __restrict__ Xd, float* __restrict__ action,
                        double* __restrict__ rewards, RoParams P, MgpFlockParams p, int K_arg, int N_arg, int T,
                        unsigned long long dimsA, unsigned int dims8, unsigned long long woffA, unsigned long long woffB,
                        int n_layers, const float* __restrict__ image, int image_floats, unsigned long long* __restrict__ carry,
                        int flags, MgpCollect cl)
{
    const int N = CN ? CN : N_arg, K = CK ? CK : K_arg;
    const RbOff cv = rb_offsets(N, K);
    const int H = ro_hist(K);
    extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
    double* spx = reinterpret_cast<double*>(smraw + cv.pos);
    double* spy = spx + N; double* svx = spx + 2 * N; double* svy = spx + 3 * N;
    double* cref = spx + 4 * N;
    unsigned long long* bits = reinterpret_cast<unsigned long long*>(smraw + cv.bits);
    float* wrow = reinterpret_cast<float*>(smraw + cv.wrow);
    float* uact = reinterpret_cast<float*>(smraw + cv.uact);
    float* XT = reinterpret_cast<float*>(smraw + cv.xt);
    float* VB = reinterpret_cast<float*>(smraw + cv.vb);
    float* wl = reinterpret_cast<float*>(smraw + cv.wl);
    float* act = reinterpret_cast<float*>(smraw + cv.act);
    float2* sxy = reinterpret_cast<float2*>(smraw + cv.sxy);
    unsigned int* mmax = reinterpret_cast<unsigned int*>(smraw + cv.mmax);
    float* uexp = reinterpret_cast<float*>(smraw + cv.uexp);                 // [2][N]
    double* vtot = reinterpret_cast<double*>(smraw + cv.uexp + ((2 * N * 4 + 7) & ~7));

    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Np = (N + 3) & ~3;
    const int NN = N * N, FK = 6 * K;
    const int ncols16 = pad16(N), NT = ncols16 / 16;
    double* xb = x + (size_t)b * N * 4;
    float* Gb = G + (size_t)b * K * NN;
    float* Xb = Xd + (size_t)b * K * 6 * N;

    // ------------------------------------------------------------------ entry
    for (int i = tid; i < H * N * RB_NW; i += RO_THREADS) bits[i] = 0ull;
    for (int i = tid; i < H * N; i += RO_THREADS) wrow[i] = 0.f;
    for (int e = tid; e < K * Np * 8; e += RO_THREADS) {
        const int f = e & 7, mk = e >> 3, k = mk / Np, m = mk - k * Np;
        const int slot = (k == 0) ? 0 : K - k;
        XT[(slot * Np + m) * 8 + f] = (f < 6 && m < N) ? Xb[((size_t)k * 6 + f) * N + m] : 0.f;
    }
    for (int i = tid; i < N; i += RO_THREADS) {
        spx[i] = xb[i * 4 + 0]; spy[i] = xb[i * 4 + 1]; svx[i] = xb[i * 4 + 2]; svy[i] = xb[i * 4 + 3];
    }
    if (tid == 0) { cref[0] = xb[0]; cref[1] = xb[1]; mmax[0] = 0u; }
    unsigned long long coin_thr = 0ull;
    unsigned int coin_ep = 0u;
    if (CL) {                                                 // data collection (see rollout_kernel)
        for (int e = tid; e < 2 * N; e += RO_THREADS) uexp[e] = cl.expert_io[(size_t)b * 2 * N + e];
        const double bq = floor((double)cl.beta[b] * 4294967296.0);
        coin_thr = bq <= 0.0 ? 0ull : (bq >= 4294967296.0 ? 4294967296ull : (unsigned long long)bq);
        coin_ep = cl.episode[b];
    }
    if (image != nullptr) {                                   // prebuilt weight image (mgp_rollout_image): flat copy
        const float4* src4 = reinterpret_cast<const float4*>(image);
        float4* dst4 = reinterpret_cast<float4*>(wl);
        for (int e = tid; e < image_floats / 4; e += RO_THREADS) dst4[e] = src4[e];
    } else {
        for (int l = 0; l < P.n_layers; ++l) {
            const int cin = (l == 0) ? FK : P.dims[l];
            const int cout = P.dims[l + 1];
            const bool last = l == P.n_layers - 1;
            const int tot = last ? ro_weight_image_size(cout, true) : ro_chain_image_size(cout, false, RB_BF);
            float* dst = wl + P.woff[l];
            for (int e = tid; e < tot; e += RO_THREADS)
                dst[e] = last ? ro_weight_image_elem(P.W[l], P.b[l], cin, cout, true, e) : ro_chain_image_elem(P.W[l], P.b[l], cin, cout, l, false, e, RB_BF);
        }
    }
    // factored hand-over of the history networks (see rollout_kernel): carry slot q -> ring slot (H - q) % H, hs = 0
    int t_off = 0;
    const size_t cwords = ro_carry_words(K, N);
    if (flags & MGP_RO_ENTER_CARRY) {
        t_off = K - 1;
        __syncthreads();                                      // (the zero fill of bits / wrow above)
        const unsigned long long* cb = carry + (size_t)b * cwords;
        const float* cw = reinterpret_cast<const float*>(cb + (size_t)H * N * RB_NW);
        for (int it = tid; it < H * N * RB_NW; it += RO_THREADS) {
            const int wd = it % RB_NW, rq = it / RB_NW, q = rq / N, row = rq - q * N;
            const int slot = (q == 0) ? 0 : H - q;
            bits[((size_t)slot * N + row) * RB_NW + wd] = cb[it];
        }
        for (int it = tid; it < H * N; it += RO_THREADS) {
            const int q = it / N, row = it - q * N;
            wrow[((q == 0) ? 0 : H - q) * N + row] = cw[it];
        }
    }
    {
        float4* za = reinterpret_cast<float4*>(act);
        for (int i = tid; i < ncols16 * RO_CS / 4; i += RO_THREADS) za[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    for (int e = tid; e < N * 8; e += RO_THREADS) {
        const int f = e & 7, n = e >> 3;
        if (f < 6) act[n * RO_CS + rpos(f * K)] = XT[e];
    }
    __syncthreads();

    // thread roles, re-derived at the top of every phase from an opaque thread index (see rollout_kernel: kept across the
    // step loop they are spilled and come back from scratch memory in every phase)
    int tidv, lanev, wavev, gq, li, lq, piece, fr, fq;
#define RB_ROLES() do { tidv = ro_fresh_tid_<true>(); lanev = tidv & 63; wavev = tidv >> 6; gq = tidv & 1; li = lanev & 15; \
                        lq = lanev >> 4; piece = tidv & 7; fr = tidv >> 2; fq = tidv & 3; } while (0)
    const int half = N >> 1, dh = (half + RO_PIECES - 1) / RO_PIECES;          // offsets 1..N/2, up to 16 per piece
    const double R2 = p.comm_radius2;
    const float R2f = (float)R2, Rf = sqrtf(R2f);
    const int nitems = N * (K - 1);
    int cur = 0, hs = 0;

    for (int t = 0; t < T; ++t) {
        RB_ROLES();
        if (CL) {
            // file the state this step starts from: features = tap 0 of the delay line, the bit rows of its network (ring
            // slot hs, intact until phase A's end), the label phase D3 of the previous step left in uexp, its age.  All of
            // it was written before the previous step's closing barrier (or at entry): no barrier needed here.
            const size_t fs = (size_t)((cl.ring_step0 + t) % cl.ring_steps) * gridDim.x + b;
            float* ff = cl.feat + fs * 6 * N;
            for (int e = tidv; e < 6 * N; e += RO_THREADS) { const int f = e / N, n = e - f * N; ff[e] = XT[((size_t)cur * Np + n) * 8 + f]; }
            unsigned long long* fb = cl.bits + fs * RB_NW * N;
            const unsigned long long* cb_ = bits + (size_t)hs * N * RB_NW;
            for (int i = tidv; i < RB_NW * N; i += RO_THREADS) fb[i] = cb_[i];
            float* fl = cl.label + fs * 2 * N;
            for (int e = tidv; e < 2 * N; e += RO_THREADS) fl[e] = uexp[e];
            if (tidv == 0) cl.age[fs] = cl.age0 + t;
        }
        const int hsn = (hs + 1 == H) ? 0 : hs + 1;
        unsigned long long* rm_new = bits + (size_t)hsn * N * RB_NW;
        float* w_new = wrow + hsn * N;
        // -------------------------------------------------------------- A: aggregation, power-iterated along the bit rows
        const int hv = min(t + t_off, K - 1);
        for (int q = 1; q <= hv; ++q) {
            int hq = hs - (q - 1); hq = hq < 0 ? hq + H : hq;
            const unsigned long long* bq = bits + (size_t)hq * N * RB_NW;
            const float* wq = wrow + hq * N;
            for (int it0 = 0; it0 < nitems; it0 += RO_THREADS / 2) {
                const int item = it0 + (tidv >> 1);
                const int gt = item / N, gn = item - gt * N, j = q + gt;
                const bool on = item < nitems && j <= K - 1;
                float sa[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                if (on) {
                    const float* src = (q == 1) ? XT + (size_t)ro_slot(cur, j, K) * Np * 8
                                                : VB + ((size_t)((q - 1) & 1) * (K - 2) + (j - 2)) * Np * 8;
                    const unsigned long long* row = bq + (size_t)gn * RB_NW + 2 * gq;      // this lanev's two words
                    rb_gather_word(row[0], 128 * gq, wq, src, sa);
                    rb_gather_word(row[1], 128 * gq + 64, wq, src, sa);
                }
#pragma unroll
                for (int f = 0; f < 6; ++f) sa[f] += dpp_f<0xB1>(sa[f]);
                if (on && gq == 0) {
                    if (j == q) {
#pragma unroll
                        for (int f = 0; f < 6; ++f) act[gn * RO_CS + rpos(f * K + j)] = sa[f];
                    } else {
                        float* dst = VB + ((size_t)(q & 1) * (K - 2) + (j - 2)) * Np * 8 + gn * 8;
                        *reinterpret_cast<float4*>(dst) = make_float4(sa[0], sa[1], sa[2], sa[3]);
                        *reinterpret_cast<float2*>(dst + 4) = make_float2(sa[4], sa[5]);
                    }
                }
            }
            if (q < hv || hv < K - 1) __syncthreads();
        }
        if (hv < K - 1) {
            for (int it0 = 0; it0 < nitems; it0 += RO_THREADS / 2) {
                const int item = it0 + (tidv >> 1);
                const int gt = item / N, gn = item - gt * N, j = gt + 1;
                const bool on = item < nitems && j > hv;
                float sa[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                if (on) {
                    const float* src = (hv == 0) ? XT + (size_t)ro_slot(cur, j, K) * Np * 8
                                                 : VB + ((size_t)(hv & 1) * (K - 2) + (j - 2)) * Np * 8;
                    const float* gcol = Gb + (size_t)(j - hv) * NN + gn;
                    for (int m = gq; m < N; m += 20) {
                        float gv[10];
#pragma unroll
                        for (int u = 0; u < 10; ++u) gv[u] = (m + 2 * u < N) ? gcol[(size_t)(m + 2 * u) * N] : 0.f;
#pragma unroll
                        for (int u = 0; u < 10; ++u) {
                            const int mm = min(m + 2 * u, N - 1);
                            const float4 x0 = *reinterpret_cast<const float4*>(src + mm * 8);
                            const float2 x1 = *reinterpret_cast<const float2*>(src + mm * 8 + 4);
                            sa[0] = fmaf(x0.x, gv[u], sa[0]); sa[1] = fmaf(x0.y, gv[u], sa[1]); sa[2] = fmaf(x0.z, gv[u], sa[2]);
                            sa[3] = fmaf(x0.w, gv[u], sa[3]); sa[4] = fmaf(x1.x, gv[u], sa[4]); sa[5] = fmaf(x1.y, gv[u], sa[5]);
                        }
                    }
                }
#pragma unroll
                for (int f = 0; f < 6; ++f) sa[f] += dpp_f<0xB1>(sa[f]);
                if (on && gq == 0) {
#pragma unroll
                    for (int f = 0; f < 6; ++f) act[gn * RO_CS + rpos(f * K + j)] = sa[f];
                }
            }
        }
        if (tidv == RO_THREADS - 1) mmax[0] = 0u;
        __syncthreads();
        RB_ROLES();
        // the slot of the oldest network was read for the last time above: empty rows for this step's pairwise pass
        for (int i = tidv; i < N * RB_NW; i += RO_THREADS) rm_new[i] = 0ull;
        // -------------------------------------------------------------- B: filter GEMM + MLP on MFMA
        // Layer metadata comes from bit-packed scalar kernel arguments: P.dims[l] indexed dynamically is re-fetched from
        // the kernel-argument segment every layer of every step (~700 cycles each).
        if (wavev < NT) {                                      // wavev w owns columns 16 w .. 16 w + 15 through every layer
            const int col = wavev * 16 + li;
            // hidden layers on MFMA, activations chained through registers (rollout_common.h): only the first layer reads its B
            // operand (the aggregation result) from LDS, only the last one stores its activations there (for the output layer)
            float zc[RO_MAXMT][4];
#pragma unroll
            for (int a_ = 0; a_ < RO_MAXMT; ++a_)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) zc[a_][rr] = 0.f;
            int mtp = 0;
            for (int l = 0; l < n_layers - 1; ++l) {
                const int cout = ro_dim(dimsA, dims8, l + 1);
                const int MT = ro_mt(cout);
                const float* wfrag = wl + (int)((((l < 4) ? woffA : woffB) >> (16 * (l & 3))) & 0xFFFFull);
                float fb[RO_KS];
                int ksteps;
                if (l == 0) {
                    const float4* pb = reinterpret_cast<const float4*>(act + col * RO_CS + lq * RO_KS);
#pragma unroll
                    for (int i = 0; i < RO_KS / 4; ++i) { const float4 tq = pb[i]; fb[4 * i] = tq.x; fb[4 * i + 1] = tq.y; fb[4 * i + 2] = tq.z; fb[4 * i + 3] = tq.w; }
                    ksteps = pad4(FK) / 4;
                } else {
#pragma unroll
                    for (int s_ = 0; s_ < RO_KS; ++s_) fb[s_] = zc[s_ >> 2][s_ & 3];
                    ksteps = 4 * mtp;
                }
                const float* pw = wfrag + lanev * RB_WFS;
                const float* pbias = wfrag + MT * 64 * RB_WFS + lq * 4;
                if constexpr (RB_BF) {
                    // widths <= 32: split-bf16 MFMA (rollout_common.h), the whole K = 32 in one instruction per product (no k-step count)
                    const int nkb = (RO_KB == 2 && l > 0 && ro_dim(dimsA, dims8, l) > 32) ? 2 : 1;   // K blocks of this layer's input
                    if (RO_MAXMT >= 8 && MT == 8) ro_layer_bf16<(RO_MAXMT >= 8 ? 8 : 1), true>(fb, pw, pbias, zc, nkb);
                    else if (RO_MAXMT >= 4 && MT == 4) ro_layer_bf16<(RO_MAXMT >= 4 ? 4 : 1), true>(fb, pw, pbias, zc, nkb);
                    else if (MT == 2) ro_layer_bf16<(RO_MAXMT >= 2 ? 2 : 1), true>(fb, pw, pbias, zc, nkb);
                    else ro_layer_bf16<1, true>(fb, pw, pbias, zc, nkb);
                } else {
                if (RO_MAXMT >= 8 && MT == 8) ro_layer_regs<(RO_MAXMT >= 8 ? 8 : 1), true>(fb, pw, pbias, ksteps, zc);
                else if (MT == 4) ro_layer_regs<(RO_MAXMT >= 4 ? 4 : 1), true>(fb, pw, pbias, ksteps, zc);
                else if (MT == 2) ro_layer_regs<2, true>(fb, pw, pbias, ksteps, zc);
                else ro_layer_regs<1, true>(fb, pw, pbias, ksteps, zc);
                }
                mtp = MT;
                RO_STAMP(12 + l);
            }
            if (n_layers > 1) {                               // last hidden layer -> LDS, channel c at slot rpos(c)
                float* pcol = act + col * RO_CS;
#pragma unroll
                for (int a_ = 0; a_ < RO_MAXMT; ++a_)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr)
                        if (a_ < mtp) pcol[rr * RO_KS + a_ * 4 + lq] = zc[a_][rr];
            }
            // ---------------------------------------------------------- C: output layer (VALU) + integrate, same wavev
            // The 2-wide output layer is a packed-FMA chain (as one zero-padded MFMA m-tile fed from the registers it measured
            // 1.5k cycles against 0.85k: eight dependent MFMAs on one accumulator).  For this part the lanes are regrouped:
            // lanev L takes agent column L >> 2 of the wavev's tile and the 8 channels c = 4 s + (L & 3) (contiguous in the
            // B-fragment layout), so the four partial sums of a column sit in one quad and are added by DPP.  The first lanev of
            // the quad then integrates the agent (spec section 1, fp64: bit-exact given the action) and publishes its fp32
            // coordinates for D1.  No workgroup barrier since the hidden layers: the wavev only reads activations it wrote
            // itself (LDS operations of one wavev are ordered).
            const int lo_ = n_layers - 1;
            const float* w2 = wl + (int)((((lo_ < 4) ? woffA : woffB) >> (16 * (lo_ & 3))) & 0xFFFFull);
            const int ccol = wavev * 16 + (lanev >> 2), cg = lanev & 3;
            const float* zsrc = act + ccol * RO_CS + cg * RO_KS;
            float zo[RO_KS];
#pragma unroll
            for (int i = 0; i < RO_KS / 4; ++i) {
                const float4 zq = *reinterpret_cast<const float4*>(zsrc + 4 * i);
                zo[4 * i] = zq.x; zo[4 * i + 1] = zq.y; zo[4 * i + 2] = zq.z; zo[4 * i + 3] = zq.w;
            }
            double px = 0.0, py = 0.0, vx = 0.0, vy = 0.0, cx = 0.0, cy = 0.0;
            const bool agent = (cg == 0) && ccol < N;
            if (agent) { px = spx[ccol]; py = spy[ccol]; vx = svx[ccol]; vy = svy[ccol]; cx = cref[0]; cy = cref[1]; }
            f32x2 u2 = {0.f, 0.f}, u2b = {0.f, 0.f};
#pragma unroll
            for (int s_ = 0; s_ < RO_KS; s_ += 2) {           // channel c = 4 s + cg: weights (W[0][c], W[1][c]) at w2[2 c]
                const float2 wa = *reinterpret_cast<const float2*>(w2 + 2 * (4 * s_ + cg));
                const float2 wb = *reinterpret_cast<const float2*>(w2 + 2 * (4 * (s_ + 1) + cg));
                u2 = __builtin_elementwise_fma((f32x2){zo[s_], zo[s_]}, (f32x2){wa.x, wa.y}, u2);
                u2b = __builtin_elementwise_fma((f32x2){zo[s_ + 1], zo[s_ + 1]}, (f32x2){wb.x, wb.y}, u2b);
            }
            u2 = u2 + u2b;
            float ux = u2.x, uy = u2.y;
            ux += dpp_f<0xB1>(ux); uy += dpp_f<0xB1>(uy);
            ux += dpp_f<0x4E>(ux); uy += dpp_f<0x4E>(uy);
            RO_STAMP(14);
            float m = 0.f;
            if (agent) {
                const float2 bb = *reinterpret_cast<const float2*>(w2 + 2 * RO_OUTC);
                ux += bb.x; uy += bb.y;
                if (CL && (unsigned long long)dagger_coin(cl.seed, coin_ep, (unsigned int)(cl.age0 + t)) < coin_thr) {
                    ux = uexp[ccol]; uy = uexp[N + ccol];      // the expert drives this step (gnn_dagger.py:157-158)
                }
                uact[ccol] = ux; uact[N + ccol] = uy;
                const float ub[2] = {ux, uy};
                integrate_one(px, py, vx, vy, ub, 1, ccol < p.n_leaders, p);
                spx[ccol] = px; spy[ccol] = py; svx[ccol] = vx; svy[ccol] = vy;
                const float sx = (float)(px - cx), sy = (float)(py - cy);     // fp32 coordinates relative to cref
                sxy[ccol] = make_float2(sx, sy);
                m = fmaxf(fabsf(sx), fabsf(sy));
            }
            RO_STAMP(15);
            m = wave_max_to_last(m);
            if (lanev == 63) atomicMax(mmax, __float_as_uint(m));
            RO_STAMP(9);  // non-negative floats order like their bit patterns
        }
        __syncthreads();
        RB_ROLES();
        // -------------------------------------------------------------- D1: membership bits, every unordered pair once
        if (wavev == RO_WAVES - 1 && (rewards != nullptr || CL)) {     // reward: one wavev, no workgroup barrier
            double sx = 0.0, sy = 0.0;
            for (int i = lanev; i < N; i += 64) { sx += svx[i]; sy += svy[i]; }
            sx = mgp_wave_sum(sx); sy = mgp_wave_sum(sy);
            if (CL && lanev == 0) { vtot[0] = sx; vtot[1] = sy; }
            const double mx = sx / (double)N, my = sy / (double)N;
            double dv = 0.0;
            for (int i = lanev; i < N; i += 64) {
                const double ex = svx[i] - mx, ey = svy[i] - my;
                dv += ex * ex + ey * ey;
            }
            const double var = mgp_wave_sum(dv) / (double)N;
            if (lanev == 0 && rewards != nullptr) rewards[(size_t)b * T + t] = -1.0 * var * p.reward_scale;
        }
        {
            const float M = __uint_as_float(mmax[0]);
            const float band = Rf * (16.f * M + 16.f * Rf) * 5.9604645e-8f + R2f * 1.1920929e-7f;
            const float t_in = R2f - band, t_out = R2f + band;
            for (int pi = tidv >> 3; pi < N; pi += RO_THREADS / 8) {
                const float2 si = sxy[pi];
                for (int ob = 0; ob < dh; ob += 8) {
                    const int d0 = 1 + piece * dh + ob;
                    const int nd = min(min(8, dh - ob), half - d0 + 1);              // offsets d0 .. d0 + nd - 1
                    if (nd <= 0) continue;
                    unsigned int in_m = 0u, out_m = 0u;
                    float2 sj[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        int j = pi + d0 + min(q, nd - 1);
                        j = (j >= N) ? j - N : j;
                        sj[q] = sxy[j];
                    }
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const float dx = si.x - sj[q].x, dy = si.y - sj[q].y;
                        const float r2 = fmaf(dy, dy, dx * dx);
                        in_m = __builtin_amdgcn_alignbit(in_m, __float_as_uint(r2 - t_in), 31);
                        out_m = __builtin_amdgcn_alignbit(out_m, __float_as_uint(t_out - r2), 31);
                    }
                    in_m = __builtin_bitreverse32(in_m) >> 24;
                    out_m = __builtin_bitreverse32(out_m) >> 24;
                    const unsigned int valid = (1u << nd) - 1u;
                    in_m &= valid;
                    unsigned int unc_m = ~out_m & ~in_m & valid;
                    while (unc_m) {                           // rare: the spec's own fp64 expression decides
                        const int q = __builtin_ctz(unc_m);
                        unc_m &= unc_m - 1u;
                        int j = pi + d0 + q;
                        j = (j >= N) ? j - N : j;
                        const double dx = spx[pi] - spx[j], dy = spy[pi] - spy[j];
                        const double r2 = dx * dx + dy * dy;
                        if (r2 < R2) in_m |= 1u << q;
                    }
                    while (in_m) {
                        const int q = __builtin_ctz(in_m);
                        in_m &= in_m - 1u;
                        int j = pi + d0 + q;
                        j = (j >= N) ? j - N : j;
                        if (FD && p.link_drop != 0u &&
                            !link_up(p, pi, j, N, fade_word(spx[pi], spy[pi]), fade_word(spx[j], spy[j]))) continue;
                        atomicOr(&rm_new[RB_NW * pi + (j >> 6)], 1ull << (j & 63));
                        atomicOr(&rm_new[RB_NW * j + (pi >> 6)], 1ull << (pi & 63));
                    }
                }
            }
        }
        __syncthreads();
        RB_ROLES();
        // -------------------------------------------------------------- D2/D3: fp64 feature terms along the bit rows
        if (tidv < 4 * ncols16) {                              // 4 lanes per row (one 64-bit word each), whole waves
            double f0 = 0, f1 = 0, f2 = 0, f3 = 0, f4 = 0, f5 = 0;
            int cnt = 0;
            if (fr < N) {
                unsigned long long w = rm_new[RB_NW * fr + fq];
                cnt = __popcll(w);
                const double xi = spx[fr], yi = spy[fr], vxi = svx[fr], vyi = svy[fr];
                while (w) {
                    const int j = 64 * fq + __builtin_ctzll(w);
                    w &= w - 1ull;
                    const double dx = xi - spx[j], dy = yi - spy[j];
                    const double r2 = dx * dx + dy * dy;
                    const double q = 1.0 / r2;
                    const double qq = q * q;
                    f0 += vxi - svx[j];
                    f1 += dx * qq;
                    f2 += dx * q;
                    f3 += vyi - svy[j];
                    f4 += dy * qq;
                    f5 += dy * q;
                }
            }
            f0 += dpp_d<0xB1>(f0); f1 += dpp_d<0xB1>(f1); f2 += dpp_d<0xB1>(f2);
            f3 += dpp_d<0xB1>(f3); f4 += dpp_d<0xB1>(f4); f5 += dpp_d<0xB1>(f5);
            f0 += dpp_d<0x4E>(f0); f1 += dpp_d<0x4E>(f1); f2 += dpp_d<0x4E>(f2);
            f3 += dpp_d<0x4E>(f3); f4 += dpp_d<0x4E>(f4); f5 += dpp_d<0x4E>(f5);
            cnt += __builtin_amdgcn_update_dpp(0, cnt, 0xB1, 0xF, 0xF, true);
            cnt += __builtin_amdgcn_update_dpp(0, cnt, 0x4E, 0xF, 0xF, true);
            if (fq == 0 && fr < N) {
                const double deg = (double)cnt;
                const double w = p.mean_pooling ? 1.0 / (deg == 0.0 ? 1.0 : deg) : 1.0;
                w_new[fr] = (float)w;
                if (CL) {                                      // expert action of the NEW state (see rollout_kernel)
                    double tvx = f0, tvy = f3;
                    if (p.centralized) { tvx = (double)N * svx[fr] - vtot[0]; tvy = (double)N * svy[fr] - vtot[1]; }
                    uexp[fr] = (float)(clipd(-tvx - (2.0 * f2 - 2.0 * f1), -p.ctrl_clip, p.ctrl_clip) * p.ctrl_gain);
                    uexp[N + fr] = (float)(clipd(-tvy - (2.0 * f5 - 2.0 * f4), -p.ctrl_clip, p.ctrl_clip) * p.ctrl_gain);
                }
                float* xn = XT + ((size_t)(cur + 1 == K ? 0 : cur + 1) * Np + fr) * 8;
                *reinterpret_cast<float4*>(xn) = make_float4((float)f0, (float)f1, (float)f2, (float)f3);
                *reinterpret_cast<float4*>(xn + 4) = make_float4((float)f4, (float)f5, 0.f, 0.f);
                float* y0 = act + fr * RO_CS;
                y0[rpos(0 * K)] = (float)f0; y0[rpos(1 * K)] = (float)f1; y0[rpos(2 * K)] = (float)f2;
                y0[rpos(3 * K)] = (float)f3; y0[rpos(4 * K)] = (float)f4; y0[rpos(5 * K)] = (float)f5;
            }
        }
        __syncthreads();
        RB_ROLES();
        if (tidv == 0) { cref[0] = spx[0]; cref[1] = spy[0]; }
        cur = (cur + 1 == K) ? 0 : cur + 1;
        hs = hsn;
    }

#undef RB_ROLES
    // ------------------------------------------------------------------ exit (see rollout_kernel)
    if (flags & MGP_RO_EXIT_CARRY) {
        unsigned long long* cb = carry + (size_t)b * cwords;
        float* cw = reinterpret_cast<float*>(cb + (size_t)H * N * RB_NW);
        for (int it = tid; it < H * N * RB_NW; it += RO_THREADS) {
            const int wd = it % RB_NW, rq = it / RB_NW, q = rq / N, row = rq - q * N;
            int hq = hs - q; hq = hq < 0 ? hq + H : hq;
            cb[it] = bits[((size_t)hq * N + row) * RB_NW + wd];
        }
        for (int it = tid; it < H * N; it += RO_THREADS) {
            const int q = it / N, row = it - q * N;
            int hq = hs - q; hq = hq < 0 ? hq + H : hq;
            cw[it] = wrow[hq * N + row];
        }
    }
    if (T > 0 && K >= 2 && !(flags & MGP_RO_SKIP_DENSE)) {
        const int hv = min(T + t_off, K - 1);
        float* rbuf = act + wave * 2 * Np;
        for (int j = K - 1; j >= 1; --j) {
            const int nsp = min(j, hv);
            for (int i = wave; i < N; i += RO_WAVES) {
                float* r0 = rbuf;
                float* r1 = rbuf + Np;
                const float wi = wrow[hs * N + i];
                const unsigned long long* rowT = bits + ((size_t)hs * N + i) * RB_NW;
                for (int n = lane; n < N; n += 64) r0[n] = ((rowT[n >> 6] >> (n & 63)) & 1ull) ? wi : 0.f;
                for (int q = 2; q <= nsp; ++q) {
                    int hq = hs - (q - 1); hq = hq < 0 ? hq + H : hq;
                    const float* wq = wrow + hq * N;
                    for (int n = lane; n < N; n += 64) {
                        const unsigned long long* rw = bits + ((size_t)hq * N + n) * RB_NW;
                        float s = 0.f;
                        for (int wd = 0; wd < RB_NW; ++wd) {
                            unsigned long long w = rw[wd];
                            while (w) { const int m = 64 * wd + __builtin_ctzll(w); w &= w - 1ull; s = fmaf(r0[m], wq[m], s); }
                        }
                        r1[n] = s;
                    }
                    float* tsw = r0; r0 = r1; r1 = tsw;
                }
                float* grow = Gb + (size_t)j * NN + (size_t)i * N;
                if (j > hv) {
                    const float* gsrc = Gb + (size_t)(j - hv) * NN;
                    float sacc[RB_MAXN / 64] = {0.f, 0.f, 0.f, 0.f};
                    for (int m = 0; m < N; ++m) {
                        const float rv = r0[m];               // wave-uniform
                        if (rv != 0.f) {
#pragma unroll
                            for (int u = 0; u < RB_MAXN / 64; ++u)
                                if (lane + 64 * u < N) sacc[u] = fmaf(rv, gsrc[(size_t)m * N + lane + 64 * u], sacc[u]);
                        }
                    }
#pragma unroll
                    for (int u = 0; u < RB_MAXN / 64; ++u)
                        if (lane + 64 * u < N) grow[lane + 64 * u] = sacc[u];
                } else {
                    for (int n = lane; n < N; n += 64) grow[n] = r0[n];
                }
            }
            __syncthreads();
        }
    }
    for (int e = tid; e < K * 6 * N; e += RO_THREADS) {
        const int k = e / (6 * N), r1 = e - k * 6 * N, f = r1 / N, n = r1 - f * N;
        Xb[e] = XT[((size_t)ro_slot(cur, k, K) * Np + n) * 8 + f];
    }
    for (int i = tid; i < N; i += RO_THREADS) {
        xb[i * 4 + 0] = spx[i]; xb[i * 4 + 1] = spy[i]; xb[i * 4 + 2] = svx[i]; xb[i * 4 + 3] = svy[i];
    }
    if (action != nullptr)
        for (int e = tid; e < 2 * N; e += RO_THREADS) action[(size_t)b * 2 * N + e] = uact[e];
    if (CL)
        for (int e = tid; e < 2 * N; e += RO_THREADS) cl.expert_io[(size_t)b * 2 * N + e] = uexp[e];
}

// coverage check + weight image plan; returns false when the shape is outside the kernel's coverage
bool make_carve(const int* dims, int n_layers, int K, int N, RoParams* P, int* lds_bytes)
{
    if (dims == nullptr || n_layers < 1 || n_layers > MGP_MAX_LAYERS) return false;
    if (K < 1 || K > 5 || N < 4 || N > RB_MAXN) return false;                   // N <= 128: 2 N (K - 1) gather threads <= 1024
    if (dims[0] != 6 || dims[n_layers] != 2) return false;                      // simulator: 6 features in, 2-D action out
#ifdef MGP_RO_XD
    {   // this build: three or more hidden layers of up to 128 channels, at least one of them wider than 64, N = 100, K = 3
        if (n_layers < 4 || N != 100 || K != 3) return false;
        int widest = 0;
        for (int l = 1; l < n_layers; ++l) { if (dims[l] < 1 || dims[l] > 128) return false; widest = dims[l] > widest ? dims[l] : widest; }
        if (widest <= 64) return false;
        const int total = ro_offsets(N, K).wl + (RO_X2_L0 + 128 * (n_layers - 2) + RO_X2_OUT + 3 * RO_X2_BLK) * 4;
        if (total > RO_LDS_LIMIT) return false;
        if (P) {
            for (int l = 0; l < n_layers; ++l) P->woff[l] = (l == 0) ? 0 : RO_X2_L0 + (l - 1) * RO_X2_L1;
            for (int l = 0; l <= n_layers; ++l) P->dims[l] = dims[l];
            P->n_layers = n_layers; P->wtot = RO_X2_L0 + (n_layers - 2) * RO_X2_L1 + RO_X2_OUT; P->bf = 1;
        }
        if (lds_bytes) *lds_bytes = total;
        return true;
    }
#endif
#ifdef MGP_RO_X2
    {   // this build: two hidden layers, at least one of them wider than 64 (the 64-wide build takes the rest), N = 100, K = 3; fixed
        // image layout (rollout_common.h RO_X2_*), its LDS copy is one K block shorter than the image
        if (n_layers != 3 || N != 100 || K != 3 || dims[1] < 1 || dims[2] < 1 || dims[1] > 128 || dims[2] > 128 ||
            (dims[1] <= 64 && dims[2] <= 64)) return false;
        const int total = ro_offsets(N, K).wl + RO_X2_LDS * 4;
        if (total > RO_LDS_LIMIT) return false;
        if (P) {
            P->woff[0] = 0; P->woff[1] = RO_X2_L0; P->woff[2] = RO_X2_L0 + RO_X2_L1;
            for (int l = 0; l <= n_layers; ++l) P->dims[l] = dims[l];
            P->n_layers = n_layers; P->wtot = RO_X2_IMAGE; P->bf = 1;
        }
        if (lds_bytes) *lds_bytes = total;
        return true;
    }
#endif
#ifdef MGP_RO_X128
    if (n_layers != 2 || N > RO_MAXN) return false;                             // this build: ONE hidden layer (up to 128 wide), N <= 128
#endif
    // layout of the hidden layers' blocks: split-bf16 pieces wherever this build runs them (RB_BF beyond N = 128); the 64-wide
    // build retries with fp32 fragments when the larger piece image does not fit
    bool bf = N > RO_MAXN ? RB_BF : RO_BF16_CHAIN;
    int wtot = 0, total = 0;
    for (int attempt = 0; attempt < 2; ++attempt) {
        wtot = 0;
        for (int l = 0; l < n_layers; ++l) {
            const int cin = (l == 0) ? 6 * K : dims[l], cout = dims[l + 1];
            const bool last = l == n_layers - 1;
            // a hidden layer reads <= RO_KS k-steps (its input comes from the aggregation tile or the previous layer's <= 4 RO_KS
            // accumulator rows) and produces <= RO_MAXMT m-tiles; the output layer reads every row of the last hidden layer
            if (cin < 1 || cout < 1 || cin > (last ? RO_OUTC : 4 * RO_KS) || cout > 16 * RO_MAXMT) return false;
            if (l == 0 && cin > 4 * RO_KS) return false;
            if (P) { P->woff[l] = wtot; P->dims[l] = dims[l]; }
            wtot += last ? ((2 * RO_OUTC + 2 + 15) & ~15) : ro_mt(cout) * 64 * ro_wfs(bf) + ro_mt(cout) * 16;
        }
        total = (N > RO_MAXN ? rb_offsets(N, K).wl : ro_offsets(N, K).wl) + wtot * 4;   // N > 128: rollout_big_kernel
        if (total <= RO_LDS_LIMIT || !(bf && RO_KB == 2 && N <= RO_MAXN)) break;
        bf = false;
    }
    if (P) { P->dims[n_layers] = dims[n_layers]; P->n_layers = n_layers; P->wtot = wtot; P->bf = bf ? 1 : 0; }
    if (total > RO_LDS_LIMIT) return false;
    if (lds_bytes) *lds_bytes = total;
    return true;
}

// Dense slices of G from a carry (the lazy half of MGP_RO_SKIP_DENSE): row i of G_j = e_i . A_t . A_{t-1} ... A_{t-j+1},
// left to right along the bit rows, the very loop (and summation order: ascending m) of the kernels' exit section.
template <int NW>
__global__ __launch_bounds__(RO_THREADS)
void carry_to_dense_kernel(const unsigned long long* __restrict__ carry, float* __restrict__ G, int K, int N)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
    const int H = ro_hist(K), Np = (N + 3) & ~3;
    unsigned long long* bits = reinterpret_cast<unsigned long long*>(smraw);                 // [H][N][NW]
    float* wr = reinterpret_cast<float*>(bits + (size_t)H * N * NW);                        // [H][N]
    float* rball = wr + ((H * N + 3) & ~3);                                                 // [waves][2][Np]
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned long long* cb = carry + (size_t)b * ro_carry_words(K, N);
    const float* cw = reinterpret_cast<const float*>(cb + (size_t)H * N * NW);
    for (int i = tid; i < H * N * NW; i += RO_THREADS) bits[i] = cb[i];
    for (int i = tid; i < H * N; i += RO_THREADS) wr[i] = cw[i];
    __syncthreads();
    float* Gb = G + (size_t)b * K * N * N;
    float* rbuf = rball + wave * 2 * Np;
    for (int j = K - 1; j >= 1; --j) {
        for (int i = wave; i < N; i += RO_WAVES) {
            float* r0 = rbuf;
            float* r1 = rbuf + Np;
            const float wi = wr[i];
            const unsigned long long* rowT = bits + (size_t)i * NW;
            for (int n = lane; n < N; n += 64) r0[n] = ((rowT[n >> 6] >> (n & 63)) & 1ull) ? wi : 0.f;
            for (int q = 1; q < j; ++q) {                     // . A_{t-q}: gather along the (symmetric) bit rows, source weights
                const float* wq = wr + q * N;
                for (int n = lane; n < N; n += 64) {
                    const unsigned long long* rw = bits + ((size_t)q * N + n) * NW;
                    float sacc = 0.f;
#pragma unroll
                    for (int wd = 0; wd < NW; ++wd) {
                        unsigned long long w = rw[wd];
                        while (w) { const int m = 64 * wd + __builtin_ctzll(w); w &= w - 1ull; sacc = fmaf(r0[m], wq[m], sacc); }
                    }
                    r1[n] = sacc;
                }
                float* tsw = r0; r0 = r1; r1 = tsw;
            }
            float* grow = Gb + (size_t)j * N * N + (size_t)i * N;
            for (int n = lane; n < N; n += 64) grow[n] = r0[n];
        }
    }
}

__global__ void rollout_image_kernel(RoParams P, int K, float* __restrict__ image, bool bf)
{
    for (int l = 0; l < P.n_layers; ++l) {
        const int cin = (l == 0) ? 6 * K : P.dims[l], cout = P.dims[l + 1];
        const bool last = l == P.n_layers - 1;
        const int tot = last ? ro_weight_image_size(cout, true) : ro_chain_image_size(cout, false, bf);
        const int span = (l + 1 < P.n_layers ? P.woff[l + 1] : P.wtot) - P.woff[l];
        for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < span; e += gridDim.x * blockDim.x)
            image[P.woff[l] + e] = (RO_XS && l == 0) ? ro_x2_l0_elem(P.W[l], P.b[l], cin, cout, e)
                                   : (RO_XS && l >= 1 && !last) ? ro_x2_l1_elem(P.W[l], P.b[l], cin, cout, e)
                                   : (e >= tot) ? 0.f : (last ? ro_weight_image_elem(P.W[l], P.b[l], cin, cout, true, e)
                                                             : ro_chain_image_elem(P.W[l], P.b[l], cin, cout, l, false, e, bf));
    }
}

// mgp_set_launch_events (capi.hip): events the NEXT resident launch of this thread stamps with the kernel's own begin / end
// (hipExtLaunchKernel: no marker packets in front of or behind the kernel, nothing for the host to wait for before it can
// enqueue the launch); consumed by that launch.
static void take_launch_events(hipEvent_t* start, hipEvent_t* stop)
{
    *start = static_cast<hipEvent_t>(mgp_tls_launch_events[0]);
    *stop = static_cast<hipEvent_t>(mgp_tls_launch_events[1]);
    mgp_tls_launch_events[0] = mgp_tls_launch_events[1] = nullptr;
}

template <int CN, int CK, bool FD, bool CL, bool CM = false, bool WBF = RO_BF16_CHAIN, bool VL = false>
int launch_rollout(double* x, float* G, float* Xd, float* action, double* rewards, const RoParams& P,
                   const MgpFlockParams* p, int B, int K, int N, int T, unsigned long long dimsA, unsigned int dims8,
                   unsigned long long woffA, unsigned long long woffB, int n_layers, int lds, hipStream_t st,
                   const float* image, int image_floats, unsigned long long* carry, int flags, const MgpCollect* cl)
{
    if (VL) lds = ro_offsets(N, K).wl + ((image_floats * 4 + 15) & ~15) + ro_voffsets(N).total;   // candidate lists behind the image
    // (the attribute sticks to the function object of the CURRENT device: cached per (device, kernel), mgp_common.h)
    if (mgp_allow_dyn_lds(reinterpret_cast<const void*>(rollout_kernel<CN, CK, FD, CL, CM, WBF, VL>), (size_t)lds) != hipSuccess) return MGP_ELAUNCH;
    MgpCollect none = {};
    hipEvent_t ev0, ev1;
    take_launch_events(&ev0, &ev1);
    if (ev0 != nullptr || ev1 != nullptr)
        hipExtLaunchKernelGGL((rollout_kernel<CN, CK, FD, CL, CM, WBF, VL>), dim3(B), dim3(RO_THREADS), lds, st, ev0, ev1, 0, x, G, Xd, action,
                              rewards, P, *p, K, N, T, dimsA, dims8, woffA, woffB, n_layers, image, image_floats, carry, flags,
                              cl ? *cl : none);
    else
        hipLaunchKernelGGL((rollout_kernel<CN, CK, FD, CL, CM, WBF, VL>), dim3(B), dim3(RO_THREADS), lds, st, x, G, Xd, action, rewards, P, *p, K,
                           N, T, dimsA, dims8, woffA, woffB, n_layers, image, image_floats, carry, flags, cl ? *cl : none);
    return mgp_launch_status();
}

template <bool FD, bool CL, int CN = 0, int CK = 0>
int launch_rollout_big(double* x, float* G, float* Xd, float* action, double* rewards, const RoParams& P,
                       const MgpFlockParams* p, int B, int K, int N, int T, unsigned long long dimsA, unsigned int dims8,
                       unsigned long long woffA, unsigned long long woffB, int n_layers, int lds, hipStream_t st,
                       const float* image, int image_floats, unsigned long long* carry, int flags, const MgpCollect* cl)
{
    if (mgp_allow_dyn_lds(reinterpret_cast<const void*>(rollout_big_kernel<FD, CL, CN, CK>), (size_t)lds) != hipSuccess) return MGP_ELAUNCH;
    MgpCollect none = {};
    hipEvent_t ev0, ev1;
    take_launch_events(&ev0, &ev1);
    if (ev0 != nullptr || ev1 != nullptr)
        hipExtLaunchKernelGGL((rollout_big_kernel<FD, CL, CN, CK>), dim3(B), dim3(RO_THREADS), lds, st, ev0, ev1, 0, x, G, Xd, action, rewards,
                              P, *p, K, N, T, dimsA, dims8, woffA, woffB, n_layers, image, image_floats, carry, flags,
                              cl ? *cl : none);
    else
        hipLaunchKernelGGL((rollout_big_kernel<FD, CL, CN, CK>), dim3(B), dim3(RO_THREADS), lds, st, x, G, Xd, action, rewards, P, *p, K, N, T,
                           dimsA, dims8, woffA, woffB, n_layers, image, image_floats, carry, flags, cl ? *cl : none);
    return mgp_launch_status();
}

}  // namespace

// This file is compiled twice: as is (layer widths <= 32, MGP_RO_KS = 8: half the activation tile and weight image), and
// through rollout_wide.hip with MGP_RO_KS = 16 for widths up to 64 (cfg/hidden_size.cfg).  The narrow build owns the
// public entry points and forwards the shapes only the wide build covers.
// Build levels: 0 = this file as is (public entry points), 1 = rollout_wide.hip, 2 = rollout_w128.hip (ONE hidden layer up to
// 128 wide: cfg/hidden_size.cfg:58).  A level forwards the shapes it does not cover to the next one.
#if defined(MGP_RO_T512)
// rollout_t512.hip: the headline instantiation as 512-thread workgroups (two episodes per CU); reached from the base build's
// dispatch when a launch has more episodes than the device has CUs; library-internal entry points, nothing forwarded
#define MGP_RO_SUPPORTED mgp_rollout_t512_supported_
#define MGP_RO_STEPS_EX mgp_rollout_t512_steps_ex_
#define MGP_RO_COLLECT mgp_rollout_t512_collect_
#define MGP_RO_IMAGE_FLOATS mgp_rollout_t512_image_floats_
#define MGP_RO_IMAGE mgp_rollout_t512_image_
#elif defined(MGP_RO_F32REF)
// rollout_f32ref.hip: this build once more with the hidden layers on fp32 MFMA 16x16x4 (MGP_RO_BF16 = 0) -- the arithmetic the
// split-bf16 layers stand in for; entry points of their own (include/mgp.h), nothing forwarded
#define MGP_RO_SUPPORTED mgp_rollout_f32ref_supported
#define MGP_RO_STEPS_EX mgp_rollout_f32ref_steps_ex
#define MGP_RO_COLLECT mgp_rollout_f32ref_collect_
#define MGP_RO_IMAGE_FLOATS mgp_rollout_f32ref_image_floats
#define MGP_RO_IMAGE mgp_rollout_f32ref_image
#elif defined(MGP_RO_XD)
#define MGP_RO_SUPPORTED mgp_rollout_xd_supported_
#define MGP_RO_STEPS_EX mgp_rollout_xd_steps_ex_
#define MGP_RO_COLLECT mgp_rollout_xd_collect_
#define MGP_RO_IMAGE_FLOATS mgp_rollout_xd_image_floats_
#define MGP_RO_IMAGE mgp_rollout_xd_image_
#elif defined(MGP_RO_X2)
#define MGP_RO_SUPPORTED mgp_rollout_x2_supported_
#define MGP_RO_STEPS_EX mgp_rollout_x2_steps_ex_
#define MGP_RO_COLLECT mgp_rollout_x2_collect_
#define MGP_RO_IMAGE_FLOATS mgp_rollout_x2_image_floats_
#define MGP_RO_IMAGE mgp_rollout_x2_image_
#define MGP_RO_NEXT(name) mgp_rollout_xd_##name##_
#elif defined(MGP_RO_X128)
#define MGP_RO_SUPPORTED mgp_rollout_x128_supported_
#define MGP_RO_STEPS_EX mgp_rollout_x128_steps_ex_
#define MGP_RO_COLLECT mgp_rollout_x128_collect_
#define MGP_RO_IMAGE_FLOATS mgp_rollout_x128_image_floats_
#define MGP_RO_IMAGE mgp_rollout_x128_image_
#define MGP_RO_NEXT(name) mgp_rollout_x2_##name##_
#elif defined(MGP_RO_WIDE)
#define MGP_RO_SUPPORTED mgp_rollout_wide_supported_
#define MGP_RO_STEPS_EX mgp_rollout_wide_steps_ex_
#define MGP_RO_COLLECT mgp_rollout_wide_collect_
#define MGP_RO_IMAGE_FLOATS mgp_rollout_wide_image_floats_
#define MGP_RO_IMAGE mgp_rollout_wide_image_
#define MGP_RO_NEXT(name) mgp_rollout_x128_##name##_
#else
#define MGP_RO_SUPPORTED mgp_rollout_supported
#define MGP_RO_STEPS_EX mgp_rollout_steps_ex
#define MGP_RO_COLLECT mgp_rollout_collect
#define MGP_RO_IMAGE_FLOATS mgp_rollout_image_floats
#define MGP_RO_IMAGE mgp_rollout_image
#define MGP_RO_NEXT(name) mgp_rollout_wide_##name##_
#endif
#ifdef MGP_RO_NEXT
extern "C" int MGP_RO_NEXT(supported)(const int* dims, int n_layers, int K, int N);
extern "C" int MGP_RO_NEXT(steps_ex)(double* x, float* G, float* Xd, const float* const* W, const float* const* b,
                                     const int* dims, int n_layers, float* action, double* rewards,
                                     const MgpFlockParams* p, int B, int K, int N, int T, const float* image,
                                     void* carry, int flags, void* stream);
extern "C" int MGP_RO_NEXT(collect)(double* x, float* G, float* Xd, const float* const* W, const float* const* b,
                                    const int* dims, int n_layers, double* rewards, const MgpFlockParams* p, int B, int K,
                                    int N, int T, const float* image, void* carry, int flags, const MgpCollect* cl,
                                    void* stream);
extern "C" long MGP_RO_NEXT(image_floats)(const int* dims, int n_layers, int K, int N);
extern "C" int MGP_RO_NEXT(image)(const float* const* W, const float* const* b, const int* dims, int n_layers, int K, int N,
                                  float* image, void* stream);
#endif

extern "C" int MGP_RO_SUPPORTED(const int* dims, int n_layers, int K, int N)
{
    if (make_carve(dims, n_layers, K, N, nullptr, nullptr)) return 1;
#ifdef MGP_RO_NEXT
    return MGP_RO_NEXT(supported)(dims, n_layers, K, N);
#else
    return 0;
#endif
}

extern "C" long MGP_RO_IMAGE_FLOATS(const int* dims, int n_layers, int K, int N)
{
    RoParams P;
    if (make_carve(dims, n_layers, K, N, &P, nullptr)) return P.wtot;
#ifdef MGP_RO_NEXT
    return MGP_RO_NEXT(image_floats)(dims, n_layers, K, N);
#else
    return 0;
#endif
}

extern "C" int MGP_RO_IMAGE(const float* const* W, const float* const* b, const int* dims, int n_layers, int K, int N,
                            float* image, void* stream)
{
    if (W == nullptr || b == nullptr) return MGP_EINVAL;
    RoParams P;
    if (!make_carve(dims, n_layers, K, N, &P, nullptr)) {
#ifdef MGP_RO_NEXT
        return MGP_RO_NEXT(image)(W, b, dims, n_layers, K, N, image, stream);
#else
        return MGP_EUNSUPPORTED;
#endif
    }
    MGP_CHECK_PTR(image);
    if (!mgp_aligned16(image)) return MGP_EALIGN;
    for (int l = 0; l < n_layers; ++l) {
        MGP_CHECK_PTR(W[l]);
        MGP_CHECK_PTR(b[l]);
        P.W[l] = W[l]; P.b[l] = b[l];
    }
    mgp_clear_error();
    hipLaunchKernelGGL(rollout_image_kernel, dim3(8), dim3(256), 0, static_cast<hipStream_t>(stream), P, K, image, P.bf != 0);
    return mgp_launch_status();
}

#if defined(MGP_RO_BASE) && !defined(MGP_RO_F32REF)
extern "C" int mgp_rollout_t512_steps_ex_(double* x, float* G, float* Xd, const float* const* W, const float* const* b,
                                          const int* dims, int n_layers, float* action, double* rewards,
                                          const MgpFlockParams* p, int B, int K, int N, int T, const float* image, void* carry,
                                          int flags, void* stream);
extern "C" int mgp_rollout_t512_collect_(double* x, float* G, float* Xd, const float* const* W, const float* const* b,
                                         const int* dims, int n_layers, double* rewards, const MgpFlockParams* p, int B, int K, int N,
                                         int T, const float* image, void* carry, int flags, const MgpCollect* cl, void* stream);
namespace {
// Two episodes per CU (rollout_t512.hip) pay when a launch has more episodes than the device has CUs: one 1024-thread workgroup
// fills a CU, so beyond that the launch time is linear in the episode count, while two 512-thread workgroups overlap one
// episode's dependent phases with the other's.  MGP_RO_T512 = 0 / 1 forces the choice (tests, A/B).
bool ro_use_t512(int B)
{
    const char* env = getenv("MGP_RO_T512");                 // (read on every launch: tests switch builds inside one process)
    if (env != nullptr && env[0] != 0) return atoi(env) != 0;
    static thread_local int cus_dev = -1, cus = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    if (dev != cus_dev) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
        cus = v; cus_dev = dev;
    }
    return cus > 0 && B > cus;
}
}  // namespace
#endif

namespace {
int ro_run(double* x, float* G, float* Xd, const float* const* W, const float* const* b,
           const int* dims, int n_layers, float* action, double* rewards,
           const MgpFlockParams* p, int B, int K, int N, int T, const float* image, void* carry_v,
           int flags, const MgpCollect* cl, void* stream)
{
    if (B < 0 || T < 0 || p == nullptr) return MGP_EINVAL;
    if (image == nullptr && (W == nullptr || b == nullptr)) return MGP_EINVAL;
    if (!(p->comm_radius2 > 0.0) || !(p->dt > 0.0) || p->n_leaders < 0) return MGP_EINVAL;
    RoParams P;
    int lds = 0;
    if (!make_carve(dims, n_layers, K, N, &P, &lds)) {
#ifdef MGP_RO_NEXT
        if (cl != nullptr)
            return MGP_RO_NEXT(collect)(x, G, Xd, W, b, dims, n_layers, rewards, p, B, K, N, T, image, carry_v, flags, cl, stream);
        return MGP_RO_NEXT(steps_ex)(x, G, Xd, W, b, dims, n_layers, action, rewards, p, B, K, N, T, image, carry_v, flags, stream);
#else
        return MGP_EUNSUPPORTED;
#endif
    }
    unsigned long long* carry = static_cast<unsigned long long*>(carry_v);
    if (cl != nullptr) {
        // data collection starts at a reset observation (all-zero carry) or continues a collecting launch: the frame of the
        // launch's first state takes its network bits from the carry
        if (!(flags & MGP_RO_ENTER_CARRY) || !(flags & MGP_RO_EXIT_CARRY)) return MGP_EINVAL;
        if (cl->ring_steps < 1 || cl->ring_step0 < 0 || cl->age0 < 0) return MGP_EINVAL;
        MGP_CHECK_PTR(cl->feat); MGP_CHECK_PTR8(cl->bits); MGP_CHECK_PTR(cl->label); MGP_CHECK_PTR(cl->age);
        MGP_CHECK_PTR(cl->expert_io); MGP_CHECK_PTR(cl->beta); MGP_CHECK_PTR(cl->episode);
    }
    if (flags & ~(MGP_RO_ENTER_CARRY | MGP_RO_EXIT_CARRY | MGP_RO_SKIP_DENSE)) return MGP_EINVAL;
    if ((flags & (MGP_RO_ENTER_CARRY | MGP_RO_EXIT_CARRY)) && carry == nullptr) return MGP_EINVAL;
    // a launch that starts from dense slices knows only the networks it produces itself: it can hand over a complete
    // history (and skip the dense rebuild) only if it runs at least K - 1 steps
    if (!(flags & MGP_RO_ENTER_CARRY) && T < K - 1 && (flags & (MGP_RO_EXIT_CARRY | MGP_RO_SKIP_DENSE))) return MGP_EINVAL;
    if ((flags & MGP_RO_SKIP_DENSE) && !(flags & MGP_RO_EXIT_CARRY)) return MGP_EINVAL;
    if (B == 0 || T == 0) return MGP_OK;
    MGP_CHECK_PTR8(x);
    MGP_CHECK_PTR(G);
    MGP_CHECK_PTR(Xd);
    if (carry != nullptr && (reinterpret_cast<uintptr_t>(carry) & 7u)) return MGP_EALIGN;
    if (image != nullptr && !mgp_aligned16(image)) return MGP_EALIGN;
    if (action != nullptr && (reinterpret_cast<uintptr_t>(action) & 3u)) return MGP_EALIGN;
    if (rewards != nullptr && (reinterpret_cast<uintptr_t>(rewards) & 7u)) return MGP_EALIGN;
    for (int l = 0; l < n_layers; ++l) {
        P.W[l] = nullptr; P.b[l] = nullptr;
        if (image == nullptr) {
            MGP_CHECK_PTR(W[l]);
            MGP_CHECK_PTR(b[l]);
            P.W[l] = W[l]; P.b[l] = b[l];
        }
    }
    unsigned long long dimsA = 0ull, woffA = 0ull, woffB = 0ull;
    unsigned int dims8 = 0u;
    for (int l = 0; l <= n_layers; ++l) {
        if (l < 8) dimsA |= (unsigned long long)(dims[l] & 255) << (8 * l);
        else dims8 = (unsigned int)dims[l];
    }
    for (int l = 0; l < n_layers; ++l) {
        if (RO_XS) continue;                                 // (the streaming builds have a fixed image layout: offsets are not passed)
        if (P.woff[l] > 0xFFFF) return MGP_EUNSUPPORTED;
        if (l < 4) woffA |= (unsigned long long)P.woff[l] << (16 * l);
        else woffB |= (unsigned long long)P.woff[l] << (16 * (l - 4));
    }
    mgp_clear_error();
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int wt = P.wtot;
    const bool fade = p->link_drop != 0u;   // FlockingStochastic-v0: the generic builds carry the fade hash, the others do not
#if defined(MGP_RO_X2) || defined(MGP_RO_XD)
    // one instantiation: (N, K) = (100, 3), no link fading, no data collection, the weight image prebuilt (blocks 2 and 3 of the
    // second layer are streamed from it every step: it cannot be built inside the launch).  Anything else: the caller's
    // two-launch path (ops.rollout_steps builds the image and retries when only that was missing).
    if (cl != nullptr || fade || image == nullptr || !mgp_aligned16(image)) return MGP_EUNSUPPORTED;
    return launch_rollout<100, 3, false, false, false, RO_BF16_CHAIN, false>(x, G, Xd, action, rewards, P, p, B, K, N, T, dimsA, dims8, woffA, woffB,
                                                                             n_layers, lds, st, image, wt, carry, flags, cl);
#elif defined(MGP_RO_T512)
    // this build is one instantiation: the reference's policy shape at the headline (N, K), plain and collecting
    if (!(N == 100 && K == 3 && !fade && n_layers == 3 && dims[1] == 32 && dims[2] == 32 && P.woff[1] == 2 * 64 * RO_WFS + 32 &&
          P.woff[2] == 2 * (2 * 64 * RO_WFS + 32)) || !(RO_VERLET != 0))
        return MGP_EUNSUPPORTED;
    if (cl != nullptr)
        return launch_rollout<100, 3, false, true, true, RO_BF16_CHAIN, true>(x, G, Xd, action, rewards, P, p, B, K, N, T, dimsA, dims8, woffA, woffB,
                                                                              n_layers, lds, st, image, wt, carry, flags, cl);
    return launch_rollout<100, 3, false, false, true, RO_BF16_CHAIN, true>(x, G, Xd, action, rewards, P, p, B, K, N, T, dimsA, dims8, woffA, woffB,
                                                                           n_layers, lds, st, image, wt, carry, flags, cl);
#else
#define RB_LAUNCH(FD_, CL_) launch_rollout_big<FD_, CL_>(x, G, Xd, action, rewards, P, p, B, K, N, T, dimsA, dims8, woffA, woffB, n_layers, lds, st, image, wt, carry, flags, cl)
#ifndef MGP_RO_X128
    if (N > RO_MAXN) {
#ifdef MGP_RO_BASE
        if (N == 200 && K == 4 && !fade) {                   // BASELINE configs[4] (cfg/n_twoflocks.cfg:114-116): sized instantiation
            if (cl != nullptr)
                return launch_rollout_big<false, true, 200, 4>(x, G, Xd, action, rewards, P, p, B, K, N, T, dimsA, dims8, woffA, woffB,
                                                               n_layers, lds, st, image, wt, carry, flags, cl);
            return launch_rollout_big<false, false, 200, 4>(x, G, Xd, action, rewards, P, p, B, K, N, T, dimsA, dims8, woffA, woffB,
                                                            n_layers, lds, st, image, wt, carry, flags, cl);
        }
#endif
        if (cl != nullptr) return fade ? RB_LAUNCH(true, true) : RB_LAUNCH(false, true);
        return fade ? RB_LAUNCH(true, false) : RB_LAUNCH(false, false);
    }
#endif
#undef RB_LAUNCH
#define RO_LAUNCH__(CN_, CK_, FD_, CL_, WBF_, VL_) launch_rollout<CN_, CK_, FD_, CL_, false, WBF_, ((CN_) != 0 && !(FD_) && RO_VERLET && MGP_RO_VL_BUILD && (VL_))>(x, G, Xd, action, rewards, P, p, B, K, N, T, dimsA, dims8, woffA, woffB, n_layers, lds, st, image, wt, carry, flags, cl)
    // sized instantiations keep Verlet candidate lists behind the weight image (S1): where they would not fit the LDS, the
    // same instantiation without lists (64- / 128-wide builds) or the run-time sized build takes the shape
    const bool vroom = lds + ro_voffsets(N).total + 16 <= RO_LDS_LIMIT;
    const bool vfit = !(RO_VERLET && MGP_RO_VL_BUILD) || MGP_RO_VL_BOTH || vroom;
#if MGP_RO_VL_BOTH
#define RO_LAUNCH_(CN_, CK_, FD_, CL_, WBF_) (vroom ? RO_LAUNCH__(CN_, CK_, FD_, CL_, WBF_, true) : RO_LAUNCH__(CN_, CK_, FD_, CL_, WBF_, false))
#else
#define RO_LAUNCH_(CN_, CK_, FD_, CL_, WBF_) RO_LAUNCH__(CN_, CK_, FD_, CL_, WBF_, true)
#endif
#ifdef MGP_RO_WIDE
#define RO_LAUNCH(CN_, CK_, FD_, CL_) (P.bf ? RO_LAUNCH_(CN_, CK_, FD_, CL_, RO_BF16_CHAIN) : RO_LAUNCH_(CN_, CK_, FD_, CL_, false))
#else
#define RO_LAUNCH(CN_, CK_, FD_, CL_) RO_LAUNCH_(CN_, CK_, FD_, CL_, RO_BF16_CHAIN)
#endif
    if (cl != nullptr) {               // the data-collection builds (DAGGER rollouts)
#ifdef MGP_RO_BASE
        if (N == 100 && K == 3 && !fade && n_layers == 3 && dims[1] == 32 && dims[2] == 32 && P.woff[1] == 2 * 64 * RO_WFS + 32 &&
            P.woff[2] == 2 * (2 * 64 * RO_WFS + 32)) {                                // cfg/dagger.cfg, policy shape compiled in
#ifndef MGP_RO_F32REF
            if (RO_VERLET != 0 && ro_use_t512(B)) {
                const int rc = mgp_rollout_t512_collect_(x, G, Xd, W, b, dims, n_layers, rewards, p, B, K, N, T, image, carry_v, flags, cl, stream);
                if (rc != MGP_EUNSUPPORTED) return rc;
            }
#endif
            return launch_rollout<100, 3, false, true, true, RO_BF16_CHAIN, RO_VERLET != 0>(x, G, Xd, action, rewards, P, p, B, K, N, T, dimsA, dims8, woffA, woffB,
                                                             n_layers, lds, st, image, wt, carry, flags, cl);
        }
#endif
        if (N == 100 && K == 3 && !fade && vfit) return RO_LAUNCH(100, 3, false, true);
        return fade ? RO_LAUNCH(0, 0, true, true) : RO_LAUNCH(0, 0, false, true);
    }
#ifdef MGP_RO_BASE
    // the reference's own policy shape at the headline (N, K): everything compile-time (cfg/dagger.cfg; BASELINE.json configs[0..1])
    if (N == 100 && K == 3 && !fade && n_layers == 3 && dims[1] == 32 && dims[2] == 32 && P.woff[1] == 2 * 64 * RO_WFS + 32 &&
        P.woff[2] == 2 * (2 * 64 * RO_WFS + 32)) {
#ifndef MGP_RO_F32REF
        if (RO_VERLET != 0 && ro_use_t512(B)) {
            const int rc = mgp_rollout_t512_steps_ex_(x, G, Xd, W, b, dims, n_layers, action, rewards, p, B, K, N, T, image, carry_v, flags, stream);
            if (rc != MGP_EUNSUPPORTED) return rc;
        }
#endif
        return launch_rollout<100, 3, false, false, true, RO_BF16_CHAIN, RO_VERLET != 0>(x, G, Xd, action, rewards, P, p, B, K, N, T, dimsA, dims8, woffA, woffB, n_layers,
                                                          lds, st, image, wt, carry, flags, cl);
    }
#endif
    if (N == 100 && K == 3 && !fade && vfit)   // the headline (N, K) with any covered policy, every build: compile-time addresses
        return RO_LAUNCH(100, 3, false, false);
#ifdef MGP_RO_BASE
    if (N == 100 && K == 2 && !fade && vfit)   // cfg/default.cfg, cloning.cfg, dagger_twoflocks.cfg
        return RO_LAUNCH(100, 2, false, false);
    if (N == 100 && K == 4 && !fade && vfit)   // cfg/k.cfg
        return RO_LAUNCH(100, 4, false, false);
    if (N == 100 && K == 1 && !fade && vfit)   // cfg/dagger_leader.cfg, k.cfg
        return RO_LAUNCH(100, 1, false, false);
#endif
    return fade ? RO_LAUNCH(0, 0, true, false) : RO_LAUNCH(0, 0, false, false);
#undef RO_LAUNCH
#undef RO_LAUNCH_
#undef RO_LAUNCH__
#endif
}
}  // namespace

extern "C" int MGP_RO_STEPS_EX(double* x, float* G, float* Xd, const float* const* W, const float* const* b,
                               const int* dims, int n_layers, float* action, double* rewards,
                               const MgpFlockParams* p, int B, int K, int N, int T, const float* image, void* carry,
                               int flags, void* stream)
{
    return ro_run(x, G, Xd, W, b, dims, n_layers, action, rewards, p, B, K, N, T, image, carry, flags, nullptr, stream);
}

extern "C" int MGP_RO_COLLECT(double* x, float* G, float* Xd, const float* const* W, const float* const* b,
                              const int* dims, int n_layers, double* rewards, const MgpFlockParams* p, int B, int K, int N,
                              int T, const float* image, void* carry, int flags, const MgpCollect* cl, void* stream)
{
    if (cl == nullptr) return MGP_EINVAL;
    return ro_run(x, G, Xd, W, b, dims, n_layers, nullptr, rewards, p, B, K, N, T, image, carry, flags, cl, stream);
}

#if defined(MGP_RO_BASE) && !defined(MGP_RO_F32REF)
// the original entry point: dense state in, dense state out, weight image built inside the launch
extern "C" int mgp_rollout_steps(double* x, float* G, float* Xd, const float* const* W, const float* const* b,
                                 const int* dims, int n_layers, float* action, double* rewards,
                                 const MgpFlockParams* p, int B, int K, int N, int T, void* stream)
{
    if (W == nullptr || b == nullptr) return MGP_EINVAL;
    return mgp_rollout_steps_ex(x, G, Xd, W, b, dims, n_layers, action, rewards, p, B, K, N, T, nullptr, nullptr, 0, stream);
}

extern "C" long mgp_rollout_carry_bytes(int K, int N)
{
    if (K < 1 || K > 5 || N < 4 || N > RB_MAXN) return 0;
    return (long)(ro_carry_words(K, N) * 8);
}

extern "C" int mgp_rollout_carry_to_dense(const void* carry, float* G, int B, int K, int N, void* stream)
{
    if (B < 0 || K < 1 || K > 5 || N < 4 || N > RB_MAXN) return MGP_EINVAL;
    if (B == 0 || K == 1) return MGP_OK;
    MGP_CHECK_PTR8(carry);
    MGP_CHECK_PTR(G);
    const int H = ro_hist(K), Np = (N + 3) & ~3, NW = ro_carry_nw(N);
    const int lds = H * N * NW * 8 + ((H * N + 3) & ~3) * 4 + RO_WAVES * 2 * Np * 4;
    mgp_clear_error();
    hipStream_t st = static_cast<hipStream_t>(stream);
    const unsigned long long* c = static_cast<const unsigned long long*>(carry);
    if (NW == 2) {
        if (mgp_allow_dyn_lds(reinterpret_cast<const void*>(carry_to_dense_kernel<2>), (size_t)lds) != hipSuccess) return MGP_ELAUNCH;
        hipLaunchKernelGGL((carry_to_dense_kernel<2>), dim3(B), dim3(RO_THREADS), lds, st, c, G, K, N);
    } else {
        if (mgp_allow_dyn_lds(reinterpret_cast<const void*>(carry_to_dense_kernel<4>), (size_t)lds) != hipSuccess) return MGP_ELAUNCH;
        hipLaunchKernelGGL((carry_to_dense_kernel<4>), dim3(B), dim3(RO_THREADS), lds, st, c, G, K, N);
    }
    return mgp_launch_status();
}
#endif
