// The factored-state rollout (N > 256; sparse_sim.hip + sparse_policy.hip) as ONE launch of persistent workgroups.
//
// mgp_sparse_rollout's K launches per step (gather stage, policy tail, cell-list simulator) each start by staging the
// episode's rows again: every one of an episode's four workgroups (N = 1000) copies all source rows of the stage into its LDS
// (68 KB / 45 KB / 40 KB per workgroup and launch), waits 5-9k cycles for rows the previous launch wrote, and pays a launch
// boundary: 33 us per step for 36 MB of state that has to move, 65 MB moved (profiles/r06_pmc_hbm_traffic_factored.txt).
//
// Here an episode's workgroups stay resident for the T steps of the call and keep in LDS what the launches re-read:
//     ring   [2][N][6]      feature rows x_{t-1}, x_{t-2} of EVERY agent (the sources of gather stage 1)
//     lw     [2][N]         row weights of A_t, A_{t-1}
//     lists  [2][256][16]   compact neighbour rows of A_t, A_{t-1} for the OWN 256 columns
//     weight image, and every thread its agent's fp64 state in registers (each workgroup integrates the whole episode:
//     one agent per thread, so the positions never travel).
// What one workgroup produces and its siblings need goes through the caller's state buffers in HBM/L2 -- the same rings,
// same layout, same values as the K-launch path writes -- with write-through (agent-scope, `sc1`) stores, one arrival
// counter per exchange and episode, and L1-bypassing (`sc1`) loads behind it (MI355X_MICROARCH.md, "Workgroup dispatch, XCD
// placement & inter-workgroup visibility": sc1 payload -> s_waitcnt vmcnt(0) -> flag; one relaxed poll -> barrier -> sc1
// loads).  Three exchanges per step, each only what the siblings lack:
//     E_a  after gather stage 1:  tap 2's running product  x_{t-2} A_t   (6 floats per own row; read: the other rows)
//     E_b  after the policy tail: the action                              (2 floats per own agent)
//     E_c  after the simulator:   row weight + feature row of the new state (7 floats per own row)
// ~21 KB written and ~53 KB read per workgroup and step instead of 153 KB staged.  The arithmetic of every phase is the
// K-launch path's, term for term in the same order: actions, states, rings and rewards are BIT-IDENTICAL to it
// (tests/test_gpu_sparse.py::test_persistent_factored_rollout_is_bit_identical).  Bit rows / list rows / the fp64 state go
// to HBM where somebody can read them: in the last H = 2 steps of the call (the networks that outlive it) and for rows
// whose list overflows (their own gathers fall back to the bit row).
//
// Covered: K = 3, N <= 1024, <= 4 layers (policy rollouts and DAGGER collection, every environment variant); everything else stays on the K-launch path.
// Residency: a workgroup needs a CU to itself (156 KB of LDS), an episode's workgroups spin on each other, so a launch
// holds at most (CUs / tiles) episodes -- more episodes run as further launches -- and every poll gives up after
// MGP_SP_PERSIST_TIMEOUT_MS, default 3 s (another process holding CUs): the episode's outputs are poisoned with NaN and the error word is set
// (mgp_sparse_rollout_status).
#include <math.h>
#include "mgp_common.h"
#include "mgp_device.h"
#include "rollout_common.h"
#include "sparse_common.h"

namespace {

#ifdef MGP_SP_PROFILE
__device__ unsigned long long mgp_pp_stamps[16 * 32];     // [wave][stamp], workgroup (tile 1, episode 3), last step
#define PP_STAMP(i) do { if (stamp_on && (threadIdx.x & 63) == 0) mgp_pp_stamps[(threadIdx.x >> 6) * 32 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define PP_STAMP(i) do { } while (0)
#endif

constexpr int PP_THREADS = 1024;
constexpr int PP_ROWS = 256;               // rows / columns per workgroup: four lanes each
constexpr int PP_SUBCAP = SS_SUBCAP;       // hits a lane of the row search can note (a row's list holds 15)
constexpr int PP_K = 3;
constexpr long long PP_TICKS_PER_MS = 100000ll;  // wall_clock64 runs at 100 MHz

struct PpArgs {
    unsigned long long* bits; float* wrow; float* feat; unsigned short* nbr;      // the rings; batch strides in elements:
    long sBb, sWb, sFb, sNb;
    const float* image; int wtot;
    float* vbuf;                   // (B,N,8): tap 2's running product between gather stage 1 and the tail
    float* action;                 // (B,2,N)
    const double* x_in; double* x_out;
    double* rewards;               // (T,B) or NULL
    float* expert;                 // (B,N,2) or NULL
    unsigned int* ctrl;            // [B][16]: arrival counters 0..2, error word 3
    int B, N, NW, T, cur, hs, b0, Bc, n_layers;
    int tiles;                     // workgroups per episode: ceil(N / 256) at least; more when the call has fewer episodes than the device CUs
    MgpSparseCollect col;          // DAGGER collection (col.feat != NULL): frames of the FIRST step's ring_step / age_now onwards
    int allow_near;                // 0: always the write-through exchange (MGP_SP_PERSIST_NEAR=0)
    int timeout_ms;                // a poll gives up after this long (MGP_SP_PERSIST_TIMEOUT_MS, default 3000)
    int fault_episode;             // test hook (MGP_SP_PERSIST_FAULT): this episode's tile-1 workgroup stops arriving after step 0; -1: none
    unsigned long long dimsA, woffA;
    MgpFlockParams p;
};

// agent-scope relaxed accesses: global_load / global_store ... sc1 (bypass this CU's L1, write through the XCD's L2)
// `near`: every workgroup of the episode runs on ONE XCD (checked at the launch's start, see the kernel): a plain store -- written
// through this CU's L1 into the XCD's L2, where the line STAYS -- is what a sibling's L1-bypassing load finds there, and the
// arrival's s_waitcnt returns when the L2 has it.  Otherwise the write-through form (`sc1`: the line leaves the L2 for memory).
__device__ __forceinline__ void pp_st2(float* p, float a, float b, bool near)
{
    if (near) { *reinterpret_cast<float2*>(p) = make_float2(a, b); return; }
    const unsigned long long v = ((unsigned long long)__float_as_uint(b) << 32) | __float_as_uint(a);
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void pp_st1(float* p, float a, bool near)
{
    if (near) { *p = a; return; }
    __hip_atomic_store(reinterpret_cast<unsigned int*>(p), __float_as_uint(a), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float2 pp_ld2(const float* p)
{
    const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_float2(__uint_as_float((unsigned int)v), __uint_as_float((unsigned int)(v >> 32)));
}
__device__ __forceinline__ float pp_ld1(const float* p)
{
    return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned int*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ unsigned long long pp_ldw(const unsigned long long* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// every payload store of the workgroup has been acknowledged, then ONE arrival
__device__ __forceinline__ void pp_arrive(unsigned int* ctr, bool mute)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0 && !mute) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// thread 0 polls until all `target` arrivals are in (or somebody gave up); returns false when the episode is dead
__device__ __forceinline__ bool pp_wait(const unsigned int* ctr, unsigned int target, unsigned int* err, int* s_dead, long long timeout)
{
    if (threadIdx.x == 0) {
        const long long t0 = wall_clock64();
        int spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 63) == 0 &&
                (wall_clock64() - t0 > timeout || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *s_dead = 1;
                break;
            }
        }
    }
    __syncthreads();
    return *s_dead == 0;
}

// one neighbour entry m: lw[m] * src_t[m][0..5] for the NT taps of the stage (rows of 6 floats in LDS)
template <int NT>
__device__ __forceinline__ void pp_gather_entry(int m, const float* lw, const float* const (&src)[NT], float (&sa)[NT][6])
{
    const float gv = lw[m];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const float* r = src[t] + m * 6;
        const float2 x0 = *reinterpret_cast<const float2*>(r);
        const float2 x1 = *reinterpret_cast<const float2*>(r + 2);
        const float2 x2 = *reinterpret_cast<const float2*>(r + 4);
        sa[t][0] = fmaf(x0.x, gv, sa[t][0]); sa[t][1] = fmaf(x0.y, gv, sa[t][1]); sa[t][2] = fmaf(x1.x, gv, sa[t][2]);
        sa[t][3] = fmaf(x1.y, gv, sa[t][3]); sa[t][4] = fmaf(x2.x, gv, sa[t][4]); sa[t][5] = fmaf(x2.y, gv, sa[t][5]);
    }
}

// spl_gather_list (sparse_policy.hip) on the LDS-resident rows: the column's list entries part, part + 4, ... of this lane;
// count 0xFFFF: the bit row (own row: read back from global memory past the L1)
template <int NT>
__device__ __forceinline__ void pp_gather_list(uint2 lst, int part, const unsigned long long* __restrict__ brow, int wpl,
                                               const float* lw, const float* const (&src)[NT], bool live, float (&sa)[NT][6])
{
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int f = 0; f < 6; ++f) sa[t][f] = 0.f;
    const unsigned int cntw = dpp_u<0xFF>(lst.y) >> 16;
    if (live) {
        if (cntw != 0xFFFFu) {
            const int mine = ((int)cntw - part + 3) >> 2;
            for (int k = 0; k < mine; ++k) {
                const unsigned int pair = (k < 2) ? lst.x : lst.y;
                pp_gather_entry<NT>((int)((k & 1) ? (pair >> 16) : (pair & 0xFFFFu)), lw, src, sa);
            }
        } else {
            for (int q = 0; q < wpl; ++q) {
                unsigned long long w = pp_ldw(brow + q);
                const int base = 64 * (part * wpl + q);
                while (w) {
                    const int m = base + __builtin_ctzll(w);
                    w &= w - 1ull;
                    pp_gather_entry<NT>(m, lw, src, sa);
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int f = 0; f < 6; ++f) { sa[t][f] += dpp_f<0xB1>(sa[t][f]); sa[t][f] += dpp_f<0x4E>(sa[t][f]); }
}

// grid: x = tile * Bc + episode-of-the-chunk (an episode's tiles land on one XCD when Bc is a multiple of 8).
// LDS: ring | lw | lists | image | scratch = max( simulator: px py vx vy [N] f64, start, cursor, cid tmp sorted, sub, posf ;
//                                                 policy: act [256][RO_CS], vst [N][6] )
// CL: DAGGER collection compiled in; FD: link fading compiled in (the plain rollout pays for neither: the kernel sits at its
// 128-VGPR limit and every path compiled in costs the others spills -- 26.2 -> 28.4 us per step measured for 40 lines of
// frame filing)
template <bool CL, bool FD>
__global__ __launch_bounds__(PP_THREADS)
void spp_rollout_kernel(PpArgs A)
{
    extern __shared__ __attribute__((aligned(16))) float ppm[];
    __shared__ double red[SS_WAVES][2];
    __shared__ float4 redf[SS_WAVES];
    __shared__ double red2[SS_WAVES];
    __shared__ int shi[SS_WAVES];
    __shared__ int s_dead;
    __shared__ int s_near;
    const int N = A.N, NW = A.NW, Np = (N + 3) & ~3, N4 = Np, wt4 = (A.wtot + 3) & ~3;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles = A.tiles;
    // rows / columns of a tile: the episode's N dealt evenly over its workgroups in whole 16-column MFMA tiles (N = 300: 2 x 160
    // instead of 256 + 44; N = 520: 3 x 176 instead of 256 + 256 + 8): a step lasts as long as its slowest tile
    const int rpt = min(PP_ROWS, (((N + tiles - 1) / tiles) + 15) & ~15);
    const int ep = (int)(blockIdx.x % (unsigned int)A.Bc), tile = (int)(blockIdx.x / (unsigned int)A.Bc), b = A.b0 + ep;
    const int i0 = tile * rpt;
    const int K = PP_K, H = 2;
    // ---- LDS plan
    float* ring = ppm;                                                           // [2][N][6]
    float* lwr = ring + (size_t)2 * N * 6;                                       // [2][Np]
    unsigned short* lists = reinterpret_cast<unsigned short*>(lwr + 2 * Np);     // [2][256][16]
    float* wl = reinterpret_cast<float*>(lists + 2 * PP_ROWS * 16);              // weight image
    float* scr = wl + wt4;
    double* spx = reinterpret_cast<double*>(scr); double* spy = spx + N; double* svx = spx + 2 * (size_t)N; double* svy = spx + 3 * (size_t)N;
    int* start = reinterpret_cast<int*>(spx + 4 * (size_t)N);                    // [G*G + 2]
    int* cursor = start + SS_G * SS_G + 2;                                       // [G*G]
    unsigned short* cid = reinterpret_cast<unsigned short*>(cursor + SS_G * SS_G);
    unsigned short* tmp = cid + N4;
    unsigned short* sorted = tmp + N4;
    unsigned short* sub = reinterpret_cast<unsigned short*>((reinterpret_cast<uintptr_t>(sorted + N4) + 7) & ~(uintptr_t)7);   // [1024][PP_SUBCAP]
    float2* posf = reinterpret_cast<float2*>((reinterpret_cast<uintptr_t>(sub + (size_t)PP_THREADS * PP_SUBCAP) + 7) & ~(uintptr_t)7);
    float* act = scr;                                                            // [256][RO_CS]
    float* vst = act + PP_ROWS * RO_CS;                                          // [N][6]
    // ---- this episode's planes
    unsigned long long* bits_b = A.bits + (size_t)b * A.sBb;
    float* wrow_b = A.wrow + (size_t)b * A.sWb;
    float* feat_b = A.feat + (size_t)b * A.sFb;
    unsigned short* nbr_b = A.nbr + (size_t)b * A.sNb;
    float* vbuf_b = A.vbuf + (size_t)b * N * 8;
    float* act_b = A.action + (size_t)b * 2 * N;
    unsigned int* ctr = A.ctrl + (size_t)b * 16;
    unsigned int* err = ctr + 3;
    const MgpFlockParams& p = A.p;
    const int wpl = NW >> 2;                                   // words of a bit row per lane of its quad (NW is a multiple of 8)
    const int gc = tid >> 2, part = tid & 3, gn = i0 + gc;     // column / row of this quad
    const bool live = gc < rpt && gn < N;
    const bool in = tid < N;                                   // this thread's agent (the same in every workgroup of the episode)
#ifdef MGP_SP_PROFILE
    bool stamp_on = false;
#endif

    // ---- entry: what the K-launch path would find in HBM
    if (tid == 0) s_dead = 0;
    double px = 0.0, py = 0.0, vx = 0.0, vy = 0.0;
    if (in) {
        const double* xr = A.x_in + ((size_t)b * N + tid) * 4;
        const double2 a01 = *reinterpret_cast<const double2*>(xr), a23 = *reinterpret_cast<const double2*>(xr + 2);
        px = a01.x; py = a01.y; vx = a23.x; vy = a23.y;
    }
    for (int i = tid; i < wt4 / 4; i += PP_THREADS) reinterpret_cast<float4*>(wl)[i] = reinterpret_cast<const float4*>(A.image)[i];
    int cur = A.cur, hs = A.hs;
    for (int t = 0; t < 2; ++t) {                              // ring slot t <- x_{t-1-t}; lw / lists slot t <- A_{t-t}
        const float* src = feat_b + (size_t)ro_slot(cur, t + 1, K) * N * 8;
        float* dst = ring + (size_t)t * N * 6;
        for (int i = tid; i < 3 * N; i += PP_THREADS) {
            const int m = i / 3, h2 = 2 * (i - 3 * m);
            *reinterpret_cast<float2*>(dst + m * 6 + h2) = *reinterpret_cast<const float2*>(src + (size_t)m * 8 + h2);
        }
        const int hq = ro_slot(hs, t, H);
        for (int i = tid; i < N; i += PP_THREADS) lwr[t * Np + i] = wrow_b[(size_t)hq * N + i];
        if (tid < 2 * PP_ROWS) {                               // 256 rows x 32 bytes
            const int r = tid >> 1, hf = tid & 1;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (r < rpt && i0 + r < N) v = *reinterpret_cast<const uint4*>(nbr_b + ((size_t)hq * N + i0 + r) * 16 + 8 * hf);
            *reinterpret_cast<uint4*>(lists + ((size_t)t * PP_ROWS + r) * 16 + 8 * hf) = v;
        }
    }
    for (int i = tid; i < PP_ROWS * RO_CS / 4; i += PP_THREADS) reinterpret_cast<float4*>(act)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    const long long timeout = (long long)A.timeout_ms * PP_TICKS_PER_MS;
    // ---- where do the siblings run?  Observed: block b runs on XCD b % 8, and the grid puts an episode's tiles Bc blocks apart;
    // nothing promises it, so every workgroup files its XCC id (HW_REG_XCC_ID) and the episode agrees on `near` behind one
    // exchange of its own (counter 4, mask word 5).  near = 0 costs speed, never correctness (MGP_SP_PERSIST_NEAR=0 forces it).
    bool near = false, alive_entry = true;
    {
        if (tid == 0) {
            const unsigned int xcc = (unsigned int)__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u;     // hwreg(HW_REG_XCC_ID, 0, 4)
            __hip_atomic_fetch_or(ctr + 5, 1u << xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_near = 0;
        }
        pp_arrive(ctr + 4, false);
        if (!pp_wait(ctr + 4, (unsigned int)tiles, err, &s_dead, timeout)) alive_entry = false;
        if (tid == 0) {
            const unsigned int mask = __hip_atomic_load(ctr + 5, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_near = (A.allow_near && (mask & (mask - 1u)) == 0u) ? 1 : 0;
        }
        __syncthreads();
        near = s_near != 0;
    }
    int rs1 = 0;                                               // ring slot of tap 1 (x_{t-1}); tap 2 in the other
    int ws = 0;                                                // lw / lists slot of A_t; A_{t-1} in the other
    bool alive = alive_entry;
    for (int s = 0; s < A.T && alive; ++s) {
#ifdef MGP_SP_PROFILE
        stamp_on = (s == (A.T > 4 ? A.T - 4 : 0)) && tile == 1 && ep == 3;
#endif
        const int nc = (cur + 1) % K, nh = (hs + 1) % H;
        const unsigned int target = (unsigned int)(tiles * (s + 1));
        const bool collecting = CL && A.col.feat != nullptr;
        const bool last_h = s >= A.T - H || collecting;        // this step's network outlives the call (or is filed as a frame)
        const bool mute = b == A.fault_episode && tile == 1 && s >= 1;      // (test hook: a workgroup that never arrives)
        PP_STAMP(0);
        // ================= gather stage 1: taps 1, 2 times A_t =================
        __syncthreads();                                        // act is zero; lw / lists / ring of this step are complete
        {
            const uint2 lst = *reinterpret_cast<const uint2*>(lists + ((size_t)ws * PP_ROWS + gc) * 16 + 4 * part);
            const float* const src[2] = {ring + (size_t)rs1 * N * 6, ring + (size_t)(rs1 ^ 1) * N * 6};
            float sa[2][6];
            pp_gather_list<2>(lst, part, bits_b + ((size_t)hs * N + min(gn, N - 1)) * NW + part * wpl, wpl, lwr + ws * Np, src, live, sa);
            if (live) {                                         // (every lane of the quad holds the sums)
                if (part == 0) {
#pragma unroll
                    for (int f = 0; f < 6; ++f) act[gc * RO_CS + rpos(f * K + 1)] = sa[0][f];
                } else {
                    const int f = 2 * (part - 1);
                    *reinterpret_cast<float2*>(vst + (size_t)gn * 6 + f) = make_float2(sa[1][f], sa[1][f + 1]);
                    pp_st2(vbuf_b + (size_t)gn * 8 + f, sa[1][f], sa[1][f + 1], near);
                }
            }
        }
        PP_STAMP(1);
        pp_arrive(ctr + 0, mute);
        {   // ring: x_t (published one exchange round ago, or by the previous launch) replaces x_{t-2}.  (Requested in front of
            // the arrival -- one round trip for the store drain and these loads -- or behind exchange (c) and held in registers
            // through the gather: both measured SLOWER, 26.5 -> 30.4 us per step; the kernel sits at its 128-VGPR limit.)
            const float* src = feat_b + (size_t)cur * N * 8;
            float* dst = ring + (size_t)(rs1 ^ 1) * N * 6;
            for (int i = tid; i < 3 * N; i += PP_THREADS) {
                const int m = i / 3, h2 = 2 * (i - 3 * m);
                *reinterpret_cast<float2*>(dst + m * 6 + h2) = pp_ld2(src + (size_t)m * 8 + h2);
            }
        }
        PP_STAMP(2);
        alive = pp_wait(ctr + 0, target, err, &s_dead, timeout);
        if (!alive) break;
        PP_STAMP(3);
        for (int i = tid; i < 3 * N; i += PP_THREADS) {         // the siblings' rows of tap 2's running product
            const int m = i / 3, h2 = 2 * (i - 3 * m);
            if (m < i0 || m >= i0 + rpt) *reinterpret_cast<float2*>(vst + m * 6 + h2) = pp_ld2(vbuf_b + (size_t)m * 8 + h2);
        }
        __syncthreads();
        PP_STAMP(4);
        // ================= policy tail: tap 0, last factor of tap 2, MLP =================
        bool expert_drives = false;
        {
            const float* xt = ring + (size_t)(rs1 ^ 1) * N * 6;
            for (int i = tid; i < PP_ROWS * 6; i += PP_THREADS) {
                const int c = i / 6, f = i - 6 * c;
                if (c < rpt && i0 + c < N) act[c * RO_CS + rpos(f * K + 0)] = xt[(size_t)(i0 + c) * 6 + f];
            }
            if (collecting) {
                // DAGGER collection (gnn_dagger.py:154-178; spl_policy_kernel<CL>): the frame of the state the step starts from
                // -- features x_t, bit rows and row weights of A_t, the expert's action for x_t, the age -- own columns / rows
                const size_t fr = (size_t)((A.col.ring_step + s) % A.col.ring_steps) * A.B + b;
                const int age_now = A.col.age_now + s;
                const double bq = floor((double)A.col.beta[b] * 4294967296.0);            // P(expert drives) in units of 2^-32
                const unsigned long long thr = bq <= 0.0 ? 0ull : (bq >= 4294967296.0 ? 4294967296ull : (unsigned long long)bq);
                expert_drives = (unsigned long long)dagger_coin(A.col.seed, A.col.episode[b], (unsigned int)age_now) < thr;
                const int cols = max(0, min(rpt, N - i0));
                float* ff = A.col.feat + fr * 6 * N;
                for (int i = tid; i < 6 * PP_ROWS; i += PP_THREADS) {
                    const int f = i >> 8, c = i & 255;
                    if (c < cols) ff[(size_t)f * N + i0 + c] = xt[(size_t)(i0 + c) * 6 + f];
                }
                const float* ex = A.expert + (size_t)b * N * 2;
                float* lb = A.col.label + fr * 2 * N;
                for (int i = tid; i < 2 * PP_ROWS; i += PP_THREADS) {
                    const int a = i >> 8, c = i & 255;
                    if (c < cols) lb[(size_t)a * N + i0 + c] = pp_ld1(ex + (size_t)(i0 + c) * 2 + a);
                }
                const unsigned long long* nr = bits_b + ((size_t)hs * N + i0) * NW;
                unsigned long long* fb = A.col.bits + (fr * N + i0) * NW;
                for (int i = tid; i < cols * NW; i += PP_THREADS) fb[i] = pp_ldw(nr + i);
                if (tid < cols) A.col.wrow[fr * N + i0 + tid] = lwr[ws * Np + i0 + tid];
                if (tile == 0 && tid == 0) A.col.age[fr] = age_now;
            }
            const int hq = ro_slot(hs, 1, H);
            const uint2 lst = *reinterpret_cast<const uint2*>(lists + ((size_t)(ws ^ 1) * PP_ROWS + gc) * 16 + 4 * part);
            const float* const src[1] = {vst};
            float sa[1][6];
            pp_gather_list<1>(lst, part, bits_b + ((size_t)hq * N + min(gn, N - 1)) * NW + part * wpl, wpl, lwr + (ws ^ 1) * Np, src, live, sa);
            if (part == 0 && live) {
#pragma unroll
                for (int f = 0; f < 6; ++f) act[gc * RO_CS + rpos(f * K + K - 1)] = sa[0][f];
            }
        }
        __syncthreads();
        PP_STAMP(5);
        if (wave * 16 < rpt && i0 + wave * 16 < N) {            // whole waves: wave w owns columns 16 w .. 16 w + 15
            const int li = lane & 15, lq = lane >> 4;
            float* pcol = act + (wave * 16 + li) * RO_CS;
            for (int l = 0; l < A.n_layers - 1; ++l) {
                const int cin = (l == 0) ? 6 * K : (int)((A.dimsA >> (8 * l)) & 255ull);
                const int cout = (int)((A.dimsA >> (8 * (l + 1))) & 255ull);
                const int MT = mtiles(cout);
                const float* wfrag = wl + (int)((A.woffA >> (16 * l)) & 0xFFFFull);
                if (MT == 2) ro_mlp_cols<2>(pcol, wfrag + lane * RO_WFS, wfrag + 2 * 64 * RO_WFS + lq * 4, pad4(cin) / 4, lq);
                else ro_mlp_cols<1>(pcol, wfrag + lane * RO_WFS, wfrag + 64 * RO_WFS + lq * 4, pad4(cin) / 4, lq);
            }
            const int lo_ = A.n_layers - 1;
            const float* w2 = wl + (int)((A.woffA >> (16 * lo_)) & 0xFFFFull);
            const int ccol = wave * 16 + (lane >> 2), cg = lane & 3;
            const float* zsrc = act + ccol * RO_CS + cg * RO_KS;
            const float4 z0 = *reinterpret_cast<const float4*>(zsrc);
            const float4 z1 = *reinterpret_cast<const float4*>(zsrc + 4);
            const float zc[8] = {z0.x, z0.y, z0.z, z0.w, z1.x, z1.y, z1.z, z1.w};
            f32x2 u2 = {0.f, 0.f}, u2b = {0.f, 0.f};
#pragma unroll
            for (int s_ = 0; s_ < RO_KS; s_ += 2) {
                const float2 wa = *reinterpret_cast<const float2*>(w2 + 2 * (4 * s_ + cg));
                const float2 wb = *reinterpret_cast<const float2*>(w2 + 2 * (4 * (s_ + 1) + cg));
                u2 = __builtin_elementwise_fma((f32x2){zc[s_], zc[s_]}, (f32x2){wa.x, wa.y}, u2);
                u2b = __builtin_elementwise_fma((f32x2){zc[s_ + 1], zc[s_ + 1]}, (f32x2){wb.x, wb.y}, u2b);
            }
            u2 = u2 + u2b;
            float ux = u2.x, uy = u2.y;
            ux += dpp_f<0xB1>(ux); uy += dpp_f<0xB1>(uy);
            ux += dpp_f<0x4E>(ux); uy += dpp_f<0x4E>(uy);
            if (cg == 0 && i0 + ccol < N) {
                const float2 bb = *reinterpret_cast<const float2*>(w2 + 2 * 4 * RO_KS);
                float ax = ux + bb.x, ay = uy + bb.y;
                if (expert_drives) {                            // gnn_dagger.py:157-161: the stored label drives the step
                    const float2 e2 = pp_ld2(A.expert + ((size_t)b * N + i0 + ccol) * 2);
                    ax = e2.x; ay = e2.y;
                }
                pp_st1(act_b + i0 + ccol, ax, near);
                pp_st1(act_b + N + i0 + ccol, ay, near);
            }
        }
        PP_STAMP(6);
        pp_arrive(ctr + 1, mute);
        PP_STAMP(7);
        alive = pp_wait(ctr + 1, target, err, &s_dead, timeout);
        if (!alive) break;
        PP_STAMP(8);
        // ================= simulator (sp_sim_kernel, the whole episode from registers) =================
        {
#pragma clang fp contract(off)
            float aux = 0.f, auy = 0.f;
            if (in && tid >= p.n_leaders) { aux = pp_ld1(act_b + tid); auy = pp_ld1(act_b + N + tid); }
            PP_STAMP(14);
            for (int c = tid; c < SS_G * SS_G + 2; c += PP_THREADS) start[c] = 0;
            double sum_vx = 0.0, sum_vy = 0.0;
            float bnx = -3e38f, bxx = -3e38f, bny = -3e38f, bxy = -3e38f;
            if (in) {                                           // spec section 1
                const double ux = clipd((double)aux, -p.max_accel, p.max_accel) * p.action_gain;
                const double uy = clipd((double)auy, -p.max_accel, p.max_accel) * p.action_gain;
                px = (px + vx * p.dt) + ((ux * p.dt) * p.dt) * 0.5;
                py = (py + vy * p.dt) + ((uy * p.dt) * p.dt) * 0.5;
                vx = vx + ux * p.dt;
                vy = vy + uy * p.dt;
                spx[tid] = px; spy[tid] = py; svx[tid] = vx; svy[tid] = vy;
                sum_vx += vx; sum_vy += vy;
                bnx = fmaxf(bnx, -(float)px); bxx = fmaxf(bxx, (float)px); bny = fmaxf(bny, -(float)py); bxy = fmaxf(bxy, (float)py);
            }
            {
                const double s0 = wave_sum_d(sum_vx), s1 = wave_sum_d(sum_vy);
                const float m0 = wave_max_to_last(bnx), m1 = wave_max_to_last(bxx), m2 = wave_max_to_last(bny), m3 = wave_max_to_last(bxy);
                if (lane == 63) { red[wave][0] = s0; red[wave][1] = s1; redf[wave] = make_float4(m0, m1, m2, m3); }
            }
            PP_STAMP(15);
            __syncthreads();
            double tot_vx, tot_vy, mnx, mxx, mny, mxy;
            {
                const int wl_ = lane & 15;
                double t0 = red[wl_][0], t1 = red[wl_][1];
                float4 m = redf[wl_];
                t0 += dpp_d<0xB1>(t0); t0 += dpp_d<0x4E>(t0); t0 += dpp_d<0x141>(t0); t0 += dpp_d<0x140>(t0);
                t1 += dpp_d<0xB1>(t1); t1 += dpp_d<0x4E>(t1); t1 += dpp_d<0x141>(t1); t1 += dpp_d<0x140>(t1);
                m.x = fmaxf(m.x, dpp_f<0xB1>(m.x)); m.x = fmaxf(m.x, dpp_f<0x4E>(m.x)); m.x = fmaxf(m.x, dpp_f<0x141>(m.x)); m.x = fmaxf(m.x, dpp_f<0x140>(m.x));
                m.y = fmaxf(m.y, dpp_f<0xB1>(m.y)); m.y = fmaxf(m.y, dpp_f<0x4E>(m.y)); m.y = fmaxf(m.y, dpp_f<0x141>(m.y)); m.y = fmaxf(m.y, dpp_f<0x140>(m.y));
                m.z = fmaxf(m.z, dpp_f<0xB1>(m.z)); m.z = fmaxf(m.z, dpp_f<0x4E>(m.z)); m.z = fmaxf(m.z, dpp_f<0x141>(m.z)); m.z = fmaxf(m.z, dpp_f<0x140>(m.z));
                m.w = fmaxf(m.w, dpp_f<0xB1>(m.w)); m.w = fmaxf(m.w, dpp_f<0x4E>(m.w)); m.w = fmaxf(m.w, dpp_f<0x141>(m.w)); m.w = fmaxf(m.w, dpp_f<0x140>(m.w));
                tot_vx = ss_first_lane(t0); tot_vy = ss_first_lane(t1);
                const float nx = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, m.x)));
                const float xx = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, m.y)));
                const float ny = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, m.z)));
                const float xy = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, m.w)));
                mnx = -((double)nx + 2.4e-7 * fabs((double)nx) + 1e-30); mxx = (double)xx + 2.4e-7 * fabs((double)xx) + 1e-30;
                mny = -((double)ny + 2.4e-7 * fabs((double)ny) + 1e-30); mxy = (double)xy + 2.4e-7 * fabs((double)xy) + 1e-30;
            }
            PP_STAMP(16);
            const bool does_reward = A.rewards != nullptr && tile == 0;
            if (does_reward) {                                  // spec section 4: population variance, two passes
                const double mvx = tot_vx / (double)N, mvy = tot_vy / (double)N;
                double dv = 0.0;
                if (in) {
                    const double ex = svx[tid] - mvx, ey = svy[tid] - mvy;
                    dv += ex * ex + ey * ey;
                }
                dv = wave_sum_d(dv);
                if (lane == 0) red2[wave] = dv;
            }
            // ---- cell grid: width >= R (1 + 1e-9) on each axis, at most 32 x 32 cells
            const double R = sqrt(p.comm_radius2) * (1.0 + 1e-9);
            const double ex_ = mxx - mnx, ey_ = mxy - mny;
            const int gx = max(1, (int)fmin((double)SS_G, floor(ex_ / R)));
            const int gy = max(1, (int)fmin((double)SS_G, floor(ey_ / R)));
            const double iwx = (ex_ > 0.0) ? (double)gx / ex_ : 0.0, iwy = (ey_ > 0.0) ? (double)gy / ey_ : 0.0;
            const float cwf = (float)fmax(ex_ / (double)gx, ey_ / (double)gy);
            int myc = 0;
            if (in) {
                int cx = (int)((px - mnx) * iwx), cy = (int)((py - mny) * iwy);
                cx = min(max(cx, 0), gx - 1); cy = min(max(cy, 0), gy - 1);
                myc = cy * gx + cx;
                cid[tid] = (unsigned short)myc;
                atomicAdd(&start[myc + 1], 1);
            }
            PP_STAMP(17);
            __syncthreads();
            if (does_reward && tid == 0) {
                double var = 0.0;
#pragma unroll
                for (int w = 0; w < SS_WAVES; ++w) var += red2[w];
                A.rewards[(size_t)s * A.B + b] = -1.0 * (var / (double)N) * p.reward_scale;
            }
            {
                const int cnt = start[tid + 1];
                const int inc = ss_wave_scan(cnt);
                if (lane == 63) shi[wave] = inc;
                __syncthreads();
                int base = (lane < wave) ? shi[lane & 15] : 0;
                base += (int)dpp_u<0xB1>((unsigned int)base); base += (int)dpp_u<0x4E>((unsigned int)base);
                base += (int)dpp_u<0x141>((unsigned int)base); base += (int)dpp_u<0x140>((unsigned int)base);
                base = __builtin_amdgcn_readfirstlane(base);
                start[tid + 1] = base + inc;
                cursor[tid] = base + inc - cnt;
            }
            PP_STAMP(18);
            __syncthreads();
            if (in) tmp[atomicAdd(&cursor[myc], 1)] = (unsigned short)tid;
            __syncthreads();
            PP_STAMP(19);
            if (in) {
                const int s0 = start[myc], s1 = start[myc + 1];
                int rank = 0;
                for (int a = s0; a < s1; ++a) rank += (tmp[a] < tid) ? 1 : 0;
                sorted[s0 + rank] = (unsigned short)tid;
                posf[s0 + rank] = make_float2((float)px, (float)py);
            }
            __syncthreads();
            PP_STAMP(9);
            // ---- row r = tid >> 2 on four lanes (sp_sim_kernel's search, fp32 pre-filter form)
            const int i = gn;
            if (live) {
                const double xi = spx[i], yi = spy[i], vxi = svx[i], vyi = svy[i];
                const double R2 = p.comm_radius2;
                const int c = cid[i], cy = c / gx, cx = c - cy * gx;
                int deg = 0;
                double f0 = 0, f1 = 0, f2 = 0, f3 = 0, f4 = 0, f5 = 0;
                const int xa = max(cx - 1, 0), xz = min(cx + 1, gx - 1);
                auto terms = [&](int j, double dx, double dyy, double r2) {
                    const double q = 1.0 / r2;
                    const double qq = q * q;
                    deg += 1;
                    f0 += vxi - svx[j];
                    f1 += dx * qq;
                    f2 += dx * q;
                    f3 += vyi - svy[j];
                    f4 += dyy * qq;
                    f5 += dyy * q;
                };
                unsigned short* mine = sub + (size_t)tid * PP_SUBCAP;
                int cnt = 0;
                const bool fading = FD && p.link_drop != 0u;    // FLOCK-SPEC item 8: a radius pair is connected iff its hash says so
                const unsigned int wi = fading ? fade_word(xi, yi) : 0u;
                const float xif = (float)xi, yif = (float)yi, R2f = (float)R2;
                const float Rf = (float)R;
                const float band = 2.384185791015625e-07f * 4.f * Rf * (2.f * fmaxf(fabsf(xif), fabsf(yif)) + 3.f * cwf + Rf) + 1e-30f;
                for (int dy = -1; dy <= 1; ++dy) {
                    const int yy = cy + dy;
                    if (yy < 0 || yy >= gy) continue;
                    const int s0 = start[yy * gx + xa], s1 = start[yy * gx + xz + 1];
                    for (int a = s0 + part; a < s1; a += 4) {
                        const int j = sorted[a];
                        const float2 pj = posf[a];
                        const float dxf = xif - pj.x, dyf = yif - pj.y;
                        const float r2f = dxf * dxf + dyf * dyf;
                        if (j == i || !(r2f < R2f + band)) continue;
                        if (r2f > R2f - band) {
                            const double dx = xi - spx[j], dyy = yi - spy[j];
                            if (!(dx * dx + dyy * dyy < R2)) continue;
                        }
                        if (fading && !link_up(p, i, j, N, wi, fade_word(spx[j], spy[j]))) continue;
                        if (cnt < PP_SUBCAP) mine[cnt] = (unsigned short)j;
                        ++cnt;
                    }
                }
                const int c0 = (int)dpp_u<0x00>((unsigned int)cnt), c1 = (int)dpp_u<0x55>((unsigned int)cnt);
                const int c2 = (int)dpp_u<0xAA>((unsigned int)cnt), c3 = (int)dpp_u<0xFF>((unsigned int)cnt);
                const int tot = c0 + c1 + c2 + c3, cmax = max(max(c0, c1), max(c2, c3));
                const bool fits = tot <= 15 && cmax <= PP_SUBCAP;
                const unsigned short* rowsub = sub + (size_t)(tid & ~3) * PP_SUBCAP;
                // the row as a compact list into the resident slot of A_{t+1} (and to HBM where it outlives the call)
                {
                    unsigned short* lrow = lists + ((size_t)(ws ^ 1) * PP_ROWS + gc) * 16;
                    *reinterpret_cast<uint2*>(lrow + 4 * part) = make_uint2(0u, part == 3 ? ((fits ? (unsigned int)tot : 0xFFFFu) << 16) : 0u);
                    __builtin_amdgcn_s_waitcnt(0xC07F);
                    __builtin_amdgcn_wave_barrier();
                    if (fits) {
                        const int off = part == 0 ? 0 : (part == 1 ? c0 : (part == 2 ? c0 + c1 : c0 + c1 + c2));
                        for (int k = 0; k < cnt; ++k) { const int e = off + k; lrow[(e & 3) * 4 + (e >> 2)] = mine[k]; }
                    }
                    __builtin_amdgcn_s_waitcnt(0xC07F);
                    __builtin_amdgcn_wave_barrier();
                    if (last_h)
                        *reinterpret_cast<uint2*>(nbr_b + ((size_t)nh * N + i) * 16 + 4 * part) = *reinterpret_cast<const uint2*>(lrow + 4 * part);
                }
                unsigned long long* gb = bits_b + ((size_t)nh * N + i) * NW;
                if ((last_h || !fits) && cmax <= PP_SUBCAP) {   // the bit row from the lanes' lists: this lane's quarter of its words
                    unsigned long long bw[4] = {0ull, 0ull, 0ull, 0ull};
                    const int w0 = part * wpl;
                    for (int e = 0; e < tot; ++e) {
                        int k = e, sl = 0;
                        if (k >= c0) { k -= c0; sl = 1; if (k >= c1) { k -= c1; sl = 2; if (k >= c2) { k -= c2; sl = 3; } } }
                        const int j = rowsub[sl * PP_SUBCAP + k];
                        const int wd = (j >> 6) - w0;
                        const unsigned long long bit = 1ull << (j & 63);
#pragma unroll
                        for (int q = 0; q < 4; ++q) bw[q] |= (wd == q) ? bit : 0ull;
                    }
                    *reinterpret_cast<ulonglong2*>(gb + w0) = make_ulonglong2(bw[0], bw[1]);
                    if (wpl > 2) *reinterpret_cast<ulonglong2*>(gb + w0 + 2) = make_ulonglong2(bw[2], bw[3]);
                }
                // Pass 2: the row's hits -- the four lists one after the other -- dealt round-robin to the four lanes
                if (cmax <= PP_SUBCAP) {
                    for (int e = part; e < tot; e += 4) {
                        int k = e, sl = 0;
                        if (k >= c0) { k -= c0; sl = 1; if (k >= c1) { k -= c1; sl = 2; if (k >= c2) { k -= c2; sl = 3; } } }
                        const int j = rowsub[sl * PP_SUBCAP + k];
                        const double dx = xi - spx[j], dyy = yi - spy[j];
                        terms(j, dx, dyy, dx * dx + dyy * dyy);
                    }
                } else {
                    // a lane's list overflowed (a dense flock): every lane walks its candidates again -- for the feature terms
                    // and for the bit row (such a row always needs one: its gathers read it), OR-ed into the zeroed row in HBM
                    *reinterpret_cast<ulonglong2*>(gb + part * wpl) = make_ulonglong2(0ull, 0ull);
                    if (wpl > 2) *reinterpret_cast<ulonglong2*>(gb + part * wpl + 2) = make_ulonglong2(0ull, 0ull);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the quad's zeros have landed before anybody's bit
                    for (int dy = -1; dy <= 1; ++dy) {
                        const int yy = cy + dy;
                        if (yy < 0 || yy >= gy) continue;
                        const int s0 = start[yy * gx + xa], s1 = start[yy * gx + xz + 1];
                        for (int a = s0 + part; a < s1; a += 4) {
                            const int j = sorted[a];
                            const double dx = xi - spx[j], dyy = yi - spy[j];
                            const double r2 = dx * dx + dyy * dyy;
                            if (j == i || !(r2 < R2)) continue;
                            if (fading && !link_up(p, i, j, N, wi, fade_word(spx[j], spy[j]))) continue;
                            terms(j, dx, dyy, r2);
                            __hip_atomic_fetch_or(gb + (j >> 6), 1ull << (j & 63), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
                }
                deg += (int)dpp_u<0xB1>((unsigned int)deg); deg += (int)dpp_u<0x4E>((unsigned int)deg);
                f0 += dpp_d<0xB1>(f0); f0 += dpp_d<0x4E>(f0); f1 += dpp_d<0xB1>(f1); f1 += dpp_d<0x4E>(f1);
                f2 += dpp_d<0xB1>(f2); f2 += dpp_d<0x4E>(f2); f3 += dpp_d<0xB1>(f3); f3 += dpp_d<0x4E>(f3);
                f4 += dpp_d<0xB1>(f4); f4 += dpp_d<0x4E>(f4); f5 += dpp_d<0xB1>(f5); f5 += dpp_d<0x4E>(f5);
                // every lane of the quad holds the row's sums: the stores are dealt over the four lanes
                float* ft = feat_b + ((size_t)nc * N + i) * 8;
                if (part == 0) {
                    const double dg = (double)deg;
                    const double w = p.mean_pooling ? 1.0 / (dg == 0.0 ? 1.0 : dg) : 1.0;
                    lwr[(ws ^ 1) * Np + i] = (float)w;
                    pp_st1(wrow_b + (size_t)nh * N + i, (float)w, near);
                    if (A.expert != nullptr) {                  // spec section 5
                        double tvx = f0, tvy = f3;
                        if (p.centralized) { tvx = (double)N * vxi - tot_vx; tvy = (double)N * vyi - tot_vy; }
                        const double ux = clipd(-tvx - (2.0 * f2 - 2.0 * f1), -p.ctrl_clip, p.ctrl_clip) * p.ctrl_gain;
                        const double uy = clipd(-tvy - (2.0 * f5 - 2.0 * f4), -p.ctrl_clip, p.ctrl_clip) * p.ctrl_gain;
                        *reinterpret_cast<float2*>(A.expert + ((size_t)b * N + i) * 2) = make_float2((float)ux, (float)uy);
                    }
                } else if (part == 1) pp_st2(ft, (float)f0, (float)f1, near);
                else if (part == 2) pp_st2(ft + 2, (float)f2, (float)f3, near);
                else pp_st2(ft + 4, (float)f4, (float)f5, near);
            }
        }
        PP_STAMP(10);
        pp_arrive(ctr + 2, mute);
        PP_STAMP(11);
        // (in the shadow of the exchange: the next step's activation tile; the simulator's scratch is dead behind the arrival's barrier)
        for (int i = tid; i < PP_ROWS * RO_CS / 4; i += PP_THREADS) reinterpret_cast<float4*>(act)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        alive = pp_wait(ctr + 2, target, err, &s_dead, timeout);
        if (!alive) break;
        PP_STAMP(12);
        if (in && (tid < i0 || tid >= i0 + rpt))                 // the siblings' row weights of A_{t+1}
            lwr[(ws ^ 1) * Np + tid] = pp_ld1(wrow_b + (size_t)nh * N + tid);
        cur = nc; hs = nh; ws ^= 1; rs1 ^= 1;
        PP_STAMP(13);
    }
    // ---- exit: the state of the own rows; a dead episode poisons what its caller will read
    if (in && tid >= i0 && tid < i0 + rpt) {
        double* xr = A.x_out + ((size_t)b * N + tid) * 4;
        if (!alive) px = py = vx = vy = __builtin_nan("");
        *reinterpret_cast<double2*>(xr) = make_double2(px, py);
        *reinterpret_cast<double2*>(xr + 2) = make_double2(vx, vy);
        if (!alive) { act_b[tid] = __builtin_nanf(""); act_b[N + tid] = __builtin_nanf(""); }
    }
    if (!alive && tile == 0 && A.rewards != nullptr)
        for (int s = tid; s < A.T; s += PP_THREADS) A.rewards[(size_t)s * A.B + b] = __builtin_nan("");
}

size_t spp_lds_bytes(int N, int wtot)
{
    const size_t Np = (size_t)((N + 3) & ~3), wt4 = (size_t)((wtot + 3) & ~3);
    const size_t persist = ((size_t)2 * N * 6 + 2 * Np + wt4) * 4 + (size_t)2 * PP_ROWS * 16 * 2;
    const size_t sim = (size_t)4 * N * 8 + (size_t)(2 * SS_G * SS_G + 2) * 4 + 3 * Np * 2 + 8 + (size_t)PP_THREADS * PP_SUBCAP * 2 + 8
                       + (size_t)N * 8;
    const size_t pol = ((size_t)PP_ROWS * RO_CS + (size_t)N * 6) * 4;
    return persist + (sim > pol ? sim : pol);
}

}  // namespace

namespace {
// MGP_OK when the persistent form covers the shape (woff / wtot / lds filled), MGP_EUNSUPPORTED otherwise
int spp_covered(const int* dims, int n_layers, int K, int N, const MgpFlockParams* p, int* woff, int* wtot, size_t* lds)
{
    const char* env = getenv("MGP_SP_PERSIST");                // (read on every call: tests switch forms inside one process)
    if (env != nullptr && env[0] != 0 && atoi(env) == 0) return MGP_EUNSUPPORTED;
    if (K != PP_K || N < 1 || N > PP_THREADS || p == nullptr) return MGP_EUNSUPPORTED;
    if (n_layers < 1 || n_layers > 4) return MGP_EUNSUPPORTED;
    if (sp_plan(dims, n_layers, K, woff, wtot) != MGP_OK) return MGP_EUNSUPPORTED;
    for (int l = 0; l < n_layers; ++l) if (woff[l] > 0xFFFF) return MGP_EUNSUPPORTED;
    *lds = spp_lds_bytes(N, *wtot);
    if (*lds + 1024 > (size_t)160 * 1024) return MGP_EUNSUPPORTED;
    return MGP_OK;
}
}  // namespace

/* 1 when mgp_sparse_rollout runs this shape as one launch of persistent workgroups (given neighbour lists). */
extern "C" int mgp_sparse_rollout_persistent(const int* dims, int n_layers, int K, int N, const MgpFlockParams* p)
{
    int woff[MGP_MAX_LAYERS], wtot = 0;
    size_t lds = 0;
    return spp_covered(dims, n_layers, K, N, p, woff, &wtot, &lds) == MGP_OK ? 1 : 0;
}

int spp_rollout(unsigned long long* bits, float* wrow, float* feat, const float* image, const int* dims, int n_layers,
                float* scratch, float* action, double* x_a, double* x_b, double* rewards, float* expert,
                const MgpFlockParams* p, int B, int K, int N, int T, int cur, int hs, unsigned short* nbr,
                const MgpSparseCollect* collect, hipStream_t st)
{
    int woff[MGP_MAX_LAYERS], wtot = 0;
    size_t lds = 0;
    if (T < 1 || B < 1 || nbr == nullptr) return MGP_EUNSUPPORTED;
    if (collect != nullptr && expert == nullptr) return MGP_EINVAL;
    if (spp_covered(dims, n_layers, K, N, p, woff, &wtot, &lds) != MGP_OK) return MGP_EUNSUPPORTED;
    static thread_local int cus_dev = -1, cus = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return MGP_ENODEV;
    if (dev != cus_dev) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return MGP_ENODEV;
        cus = v; cus_dev = dev;
    }
    // workgroups per episode: ceil(N / 256) at least -- and as many more as the device has CUs for when the call is short of
    // episodes (8 x 1000: 32 tiles of 32 rows instead of 4 of 250 on 32 of 256 CUs): the row search, the gathers and the MLP
    // shrink with the rows of a tile, the per-workgroup set-up (integration, cell list) does not
    int tiles = mgp_ceil_div(N, PP_ROWS);
    if (tiles > cus) return MGP_EUNSUPPORTED;
    {
        const int most = mgp_ceil_div(N, 16);                   // one 16-column MFMA tile per workgroup at the very least
        int want = cus / (B < 1 ? 1 : B);
        const char* tl = getenv("MGP_SP_PERSIST_TILES");        // (A/B and tests: forces the tile count where valid)
        if (tl != nullptr && atoi(tl) > 0) want = atoi(tl);
        if (want > most) want = most;
        if (want > cus) want = cus;
        if (want > tiles) tiles = want;
        const int rpt = ((mgp_ceil_div(N, tiles) + 15) & ~15) < PP_ROWS ? ((mgp_ceil_div(N, tiles) + 15) & ~15) : PP_ROWS;
        tiles = mgp_ceil_div(N, rpt);                           // (no tile without rows: the kernel derives the same rpt from this count)
    }
    int bc_max = cus / tiles;                                   // one workgroup per CU: the episodes that can be resident together
    if (bc_max < 1) return MGP_EUNSUPPORTED;
    if (bc_max >= 8 && B > bc_max) bc_max &= ~7;                // several launches: whole XCD rounds, an episode's tiles share an L2
    MGP_CHECK_PTR8(bits); MGP_CHECK_PTR(wrow); MGP_CHECK_PTR(feat); MGP_CHECK_PTR(image); MGP_CHECK_PTR(action);
    MGP_CHECK_PTR(scratch); MGP_CHECK_PTR8(x_a); MGP_CHECK_PTR8(x_b);
    if (!mgp_aligned16(feat) || !mgp_aligned16(image) || !mgp_aligned16(scratch) || !mgp_aligned16(x_a) || !mgp_aligned16(x_b)
        || !mgp_aligned16(bits) || !mgp_aligned16(nbr)) return MGP_EALIGN;
    if (expert != nullptr && (reinterpret_cast<uintptr_t>(expert) & 7u)) return MGP_EALIGN;
    const int NW = mgp_sparse_words(N), H = 2;
    PpArgs A = {};
    A.bits = bits; A.wrow = wrow; A.feat = feat; A.nbr = nbr;
    A.sBb = (long)H * N * NW; A.sWb = (long)H * N; A.sFb = (long)K * N * 8; A.sNb = (long)H * N * 16;
    A.image = image; A.wtot = wtot;
    A.vbuf = scratch;
    A.ctrl = reinterpret_cast<unsigned int*>(scratch + (size_t)B * N * 8);     // (scratch holds 4 B N 8 floats at K = 3)
    A.action = action;
    A.x_in = x_a; A.x_out = (T & 1) ? x_b : x_a;
    A.rewards = rewards; A.expert = expert;
    A.B = B; A.N = N; A.NW = NW; A.T = T; A.cur = cur; A.hs = hs; A.n_layers = n_layers; A.tiles = tiles;
    for (int l = 0; l <= n_layers; ++l) A.dimsA |= (unsigned long long)(dims[l] & 255) << (8 * l);
    for (int l = 0; l < n_layers; ++l) A.woffA |= (unsigned long long)woff[l] << (16 * l);
    A.p = *p;
    if (collect != nullptr) A.col = *collect;
    const char* tmo = getenv("MGP_SP_PERSIST_TIMEOUT_MS");
    A.timeout_ms = (tmo != nullptr && atoi(tmo) > 0) ? atoi(tmo) : 3000;
    const char* nr = getenv("MGP_SP_PERSIST_NEAR");
    A.allow_near = (nr != nullptr && nr[0] != 0 && atoi(nr) == 0) ? 0 : 1;
    const char* flt = getenv("MGP_SP_PERSIST_FAULT");
    A.fault_episode = (flt != nullptr && flt[0] != 0) ? atoi(flt) : -1;
    mgp_clear_error();
    auto go = [&](auto kern) -> int {
        if (mgp_allow_dyn_lds(reinterpret_cast<const void*>(kern), lds) != hipSuccess) return MGP_ELAUNCH;
        if (hipMemsetAsync(A.ctrl, 0, (size_t)B * 16 * sizeof(unsigned int), st) != hipSuccess) return MGP_ELAUNCH;
        for (int b0 = 0; b0 < B; b0 += bc_max) {
            A.b0 = b0; A.Bc = (B - b0 < bc_max) ? B - b0 : bc_max;
            hipLaunchKernelGGL(kern, dim3((unsigned int)(tiles * A.Bc)), dim3(PP_THREADS), lds, st, A);
        }
        return mgp_launch_status();
    };
    const bool fd = p->link_drop != 0u;
    if (collect != nullptr) return fd ? go(spp_rollout_kernel<true, true>) : go(spp_rollout_kernel<true, false>);
    return fd ? go(spp_rollout_kernel<false, true>) : go(spp_rollout_kernel<false, false>);
}

/* Error word of the persistent form of mgp_sparse_rollout: synchronises `stream`, then MGP_OK, or MGP_ELAUNCH when an episode's
 * workgroups gave up waiting for each other (its outputs are NaN).  scratch / B / K / N as passed to mgp_sparse_rollout. */
extern "C" int mgp_sparse_rollout_status(const float* scratch, int B, int K, int N, void* stream)
{
    if (scratch == nullptr || B < 1 || N < 1) return MGP_EINVAL;
    if (K != PP_K) return MGP_OK;
    if (hipStreamSynchronize(static_cast<hipStream_t>(stream)) != hipSuccess) return MGP_ELAUNCH;
    unsigned int* host = static_cast<unsigned int*>(malloc((size_t)B * 16 * sizeof(unsigned int)));
    if (host == nullptr) return MGP_EINVAL;
    int rc = MGP_OK;
    if (hipMemcpy(host, scratch + (size_t)B * N * 8, (size_t)B * 16 * sizeof(unsigned int), hipMemcpyDeviceToHost) != hipSuccess) rc = MGP_ELAUNCH;
    else for (int b = 0; b < B; ++b) if (host[(size_t)b * 16 + 3] != 0u) rc = MGP_ELAUNCH;
    free(host);
    return rc;
}
