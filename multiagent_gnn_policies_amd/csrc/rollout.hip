// Episode-resident closed-loop rollout: T x { Actor forward -> simulator step -> delayed-GSO / delay-line transition }
// in ONE launch, one 1024-thread workgroup per episode, the episode's whole state in LDS.
//
// Reference loop being replaced (one episode, test_model.py:38-44 / gnn_dagger.py:194-203):
//     action = learner.select_action(state)                      -> actor.py:45-86
//     next_state, reward, done, _ = env.step(action)             -> gym_flock (FLOCK-SPEC v1)
//     state = MultiAgentStateWithDelay(..., prev_state=state)    -> state_with_delay.py:44-53
// The two-launch form of one step (mgp_actor_fwd + mgp_flock_step_advance) re-reads the dense delayed operator
// G (B,K,N,N) from HBM twice per step and rewrites it once; that traffic, not arithmetic, is what a step costs at
// N = 100.  Here G slices 1..K-1 ((K-1)*N*N*4 = 80 KB at N = 100, K = 3), the delay line, the fp64 agent states, the
// MFMA weight fragments and the activation tile all stay in the CU's 160 KB LDS for the whole launch; HBM sees the
// state once on entry and once on exit (plus one reward per step).  Slice 0 of G is the identity by construction
// (state_with_delay.py:44) and is neither read nor written: tap 0 of the aggregation is X_0 itself.
//
// Per step (barrier-separated phases, all arithmetic identical in kind to the stand-alone kernels):
//   A  aggregation  y[(f,k), n] = sum_m X_k[f, m] * G_k[m, n]  from LDS: thread = (tap, column n, row phase r),
//      row phases (lanes of one wave) combined by xor-shuffles; result stored in MFMA B-fragment order.
//   B  filter GEMM + tanh MLP on fp32 MFMA 16x16x4 (k-ordered fmaf chain, 1e-5 budget), wave = 16 agent columns
//      through all layers, activations in place in LDS; the last layer leaves the action in LDS.
//   C  fp64 integration of every agent (same expression tree as flock.hip / the oracle: bit-exact given the action).
//   D  pairwise pass, thread = (agent i, piece of 1/8 of the j range): membership bits, then the fp64 feature terms
//      for actual neighbours only; the 8 pieces of a row are adjacent lanes and are combined by shuffles.  One
//      otherwise idle wave computes the reward (velocity variance) of the step.
//   E  G_j <- A_t . G_{j-1} for j = K-1 .. 2 (row gathers in LDS, ascending neighbour order, fmaf chain: the same
//      arithmetic as gso.hip), then G_1 <- A_t expanded from the membership bits.  The delay line is a ring: the new
//      features overwrite the oldest tap, nothing is shifted.
// This translation unit is built with -ffp-contract=off (fp64 spec arithmetic); fp32 fused multiply-adds are explicit.
#include "mgp_device.h"

namespace {

constexpr int RO_THREADS = 1024;
constexpr int RO_WAVES = RO_THREADS / 64;
constexpr int RO_PIECES = 8;              // j-range pieces per agent row in the pairwise pass (adjacent lanes)
constexpr int RO_MAXN = 128;              // RO_THREADS / RO_PIECES rows; membership bits of a row fit 2 x u64
constexpr int RO_LDS_LIMIT = 160 * 1024;

#ifdef MGP_RO_PROFILE
__device__ unsigned long long mgp_ro_stamps[64];
#define RO_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0 && t == 1) mgp_ro_stamps[i] = __builtin_readcyclecounter(); } while (0)
#else
#define RO_STAMP(i) do { } while (0)
#endif

struct RoParams {
    const float* W[MGP_MAX_LAYERS];
    const float* b[MGP_MAX_LAYERS];
    int dims[MGP_MAX_LAYERS + 1];         // 6, h1, ..., 2
    int woff[MGP_MAX_LAYERS];             // offset (floats) of layer l's fragment block inside the weight image
    int n_layers;
};

struct RoCarve {                          // byte offsets into the dynamic LDS
    int pos;                              // double px, py, vx, vy [4][N]
    int mask;                             // u64 [N][2] membership bits of the current network
    int wrow;                             // float [N]  network weight of row i (1/deg or 1)
    int uact;                             // float [2][N] action (the Actor output layout (nA, N))
    int xt;                               // float [K][N][8] delay line, ring over taps, transposed (6 features + 2 pad)
    int gd;                               // float [K-1][N][N] delayed operator, slices 1..K-1
    int wl;                               // float weight image: per layer fragments [MT][64][AF_WFS] + bias [MT*16]
    int act;                              // float [ncols16][AF_CS] activations (in place through the layers)
    int total;
    int rps;                              // log2 of the aggregation's row phases
};

struct RoMlp {
    float* buf; const float* wfrag; float* uact;
    int ksteps, cout, N, nt, lane; bool last;
};

// One layer for the 16 agent columns of n-tile a.nt, in place: the wave reads its B fragments completely before it
// stores the first output (LDS accesses of one wave are ordered), and no other wave touches these columns.
template <int MT>
__device__ __forceinline__ void ro_mlp_layer(const RoMlp& a)
{
    const int li = a.lane & 15, lq = a.lane >> 4;
    const int col = a.nt * 16 + li;
    float fb[16];
    {
        const float4* pb = reinterpret_cast<const float4*>(a.buf + col * AF_CS + lq * 16);
#pragma unroll
        for (int i = 0; i < 4; ++i) { const float4 t = pb[i]; fb[4 * i] = t.x; fb[4 * i + 1] = t.y; fb[4 * i + 2] = t.z; fb[4 * i + 3] = t.w; }
    }
    float fa[MT][16];
    f32x4 acc[MT];
    const float* bias = a.wfrag + MT * 64 * AF_WFS;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const float4* pa = reinterpret_cast<const float4*>(a.wfrag + (mt * 64 + a.lane) * AF_WFS);
#pragma unroll
        for (int i = 0; i < 4; ++i) { const float4 t = pa[i]; fa[mt][4 * i] = t.x; fa[mt][4 * i + 1] = t.y; fa[mt][4 * i + 2] = t.z; fa[mt][4 * i + 3] = t.w; }
        const float4 bv = *reinterpret_cast<const float4*>(bias + mt * 16 + lq * 4);
        acc[mt][0] = bv.x; acc[mt][1] = bv.y; acc[mt][2] = bv.z; acc[mt][3] = bv.w;
    }
    // k-steps in groups of four; surplus steps multiply stale-but-finite activations by zero-padded weights
#pragma unroll
    for (int sg = 0; sg < 4; ++sg) {
        if (4 * sg < a.ksteps) {
#pragma unroll
            for (int s = 4 * sg; s < 4 * sg + 4; ++s)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[mt][s], fb[s], acc[mt], 0, 0, 0);
        }
    }
    if (a.last) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int c = mt * 16 + lq * 4 + rr;
                if (c < a.cout && col < a.N) a.uact[c * a.N + col] = acc[mt][rr];
            }
        return;
    }
    float z[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) z[mt][rr] = tanh_fast(acc[mt][rr]);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) a.buf[col * AF_CS + rr * 16 + mt * 4 + lq] = z[mt][rr];     // == bpos(c)
}

__device__ __forceinline__ int ro_slot(int cur, int k, int K) { int s = cur - k; return s < 0 ? s + K : s; }

__global__ __launch_bounds__(RO_THREADS)
void rollout_kernel(double* __restrict__ x, float* __restrict__ G, float* __restrict__ Xd, float* __restrict__ action,
                    double* __restrict__ rewards, RoParams P, RoCarve cv, MgpFlockParams p, int K, int N, int T)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
    double* spx = reinterpret_cast<double*>(smraw + cv.pos);
    double* spy = spx + N; double* svx = spx + 2 * N; double* svy = spx + 3 * N;
    unsigned long long* rowmask = reinterpret_cast<unsigned long long*>(smraw + cv.mask);
    float* wrow = reinterpret_cast<float*>(smraw + cv.wrow);
    float* uact = reinterpret_cast<float*>(smraw + cv.uact);
    float* XT = reinterpret_cast<float*>(smraw + cv.xt);
    float* Gd = reinterpret_cast<float*>(smraw + cv.gd);
    float* wl = reinterpret_cast<float*>(smraw + cv.wl);
    float* act = reinterpret_cast<float*>(smraw + cv.act);

    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NN = N * N, n4 = N >> 2, FK = 6 * K;
    const int ncols16 = pad16(N), NT = ncols16 / 16;
    double* xb = x + (size_t)b * N * 4;
    float* Gb = G + (size_t)b * K * NN;
    float* Xb = Xd + (size_t)b * K * 6 * N;

    // ------------------------------------------------------------------ entry: the episode's state -> LDS
    {
        const float4* gsrc = reinterpret_cast<const float4*>(Gb + NN);        // slices 1..K-1
        float4* gdst = reinterpret_cast<float4*>(Gd);
        const int tot4 = (K - 1) * NN / 4;
#pragma unroll 8
        for (int e = tid; e < tot4; e += RO_THREADS) gdst[e] = gsrc[e];
    }
    for (int e = tid; e < K * N * 8; e += RO_THREADS) {                        // tap k -> ring slot (K - k) % K, cur = 0
        const int f = e & 7, mk = e >> 3, k = mk / N, m = mk - k * N;
        const int slot = (k == 0) ? 0 : K - k;
        XT[(slot * N + m) * 8 + f] = (f < 6) ? Xb[((size_t)k * 6 + f) * N + m] : 0.f;
    }
    for (int i = tid; i < N; i += RO_THREADS) {
        spx[i] = xb[i * 4 + 0]; spy[i] = xb[i * 4 + 1]; svx[i] = xb[i * 4 + 2]; svy[i] = xb[i * 4 + 3];
    }
    // weights in MFMA A-fragment order (see actor_fused.hip): wfrag[mt][lane][AF_WFS], lane = (c & 3) * 16 + (o & 15),
    // slot s = c >> 2, zero padded; then the bias of the layer's MT*16 rows
    for (int l = 0; l < P.n_layers; ++l) {
        const int cin = (l == 0) ? FK : P.dims[l];
        const int cout = P.dims[l + 1];
        const int MT = mtiles(cout);
        const int tot = MT * 64 * AF_WFS;
        float* dst = wl + P.woff[l];
        const float* src = P.W[l];
        for (int e = tid; e < tot; e += RO_THREADS) {
            const int mt = e / (64 * AF_WFS), r1 = e - mt * (64 * AF_WFS);
            const int ln = r1 / AF_WFS, sl = r1 - ln * AF_WFS;
            const int c = 4 * sl + (ln >> 4), o = mt * 16 + (ln & 15);
            dst[e] = (sl < 16 && o < cout && c < cin) ? src[(size_t)o * cin + c] : 0.f;
        }
        for (int o = tid; o < MT * 16; o += RO_THREADS) dst[tot + o] = (o < cout) ? P.b[l][o] : 0.f;
    }
    {
        float4* za = reinterpret_cast<float4*>(act);
        for (int i = tid; i < ncols16 * AF_CS / 4; i += RO_THREADS) za[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();

    // thread roles that do not change over the steps
    const int rps = cv.rps, RPn = 1 << rps, per_k = N << rps;
    const int ak = tid / per_k, arem = tid - ak * per_k;
    const int an = arem >> rps, ar = arem & (RPn - 1);
    const bool agg_active = ak < K - 1;                       // tap ak + 1, column an, row phase ar
    const int pi = tid >> 3, piece = tid & 7;                 // pairwise: agent row pi, j piece
    const int jh = (N + RO_PIECES - 1) / RO_PIECES;           // <= 16 j's per piece
    const int hw = tid >> 5, hl = tid & 31;                   // operator rows: half-wave per row, 4 columns per lane
    const double R2 = p.comm_radius2;
    int cur = 0;

    for (int t = 0; t < T; ++t) {
        RO_STAMP(0);
        // -------------------------------------------------------------- A: aggregation from LDS
        {
            float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (agg_active) {
                const float* g = Gd + (size_t)ak * NN + an;
                const float* xt = XT + (size_t)ro_slot(cur, ak + 1, K) * N * 8;
#pragma unroll 5
                for (int m = ar; m < N; m += RPn) {
                    const float gv = g[m * N];
                    const float4 x0 = *reinterpret_cast<const float4*>(xt + m * 8);
                    const float2 x1 = *reinterpret_cast<const float2*>(xt + m * 8 + 4);
                    acc[0] = fmaf(x0.x, gv, acc[0]); acc[1] = fmaf(x0.y, gv, acc[1]); acc[2] = fmaf(x0.z, gv, acc[2]);
                    acc[3] = fmaf(x0.w, gv, acc[3]); acc[4] = fmaf(x1.x, gv, acc[4]); acc[5] = fmaf(x1.y, gv, acc[5]);
                }
            }
            for (int s = 1; s < RPn; s <<= 1) {
#pragma unroll
                for (int f = 0; f < 6; ++f) acc[f] += __shfl_xor(acc[f], s, MGP_WAVE);
            }
            if (agg_active && ar == 0) {
#pragma unroll
                for (int f = 0; f < 6; ++f) act[an * AF_CS + bpos(f * K + ak + 1)] = acc[f];
            }
            const float* x0t = XT + (size_t)cur * N * 8;      // tap 0: G_0 = I  =>  y_0 = X_0
            for (int e = tid; e < N * 8; e += RO_THREADS) {
                const int f = e & 7, n = e >> 3;
                if (f < 6) act[n * AF_CS + bpos(f * K)] = x0t[e];
            }
        }
        __syncthreads();
        RO_STAMP(1);
        // -------------------------------------------------------------- B: filter GEMM + MLP on MFMA
        if (wave < NT) {
            for (int l = 0; l < P.n_layers; ++l) {
                const int cin = (l == 0) ? FK : P.dims[l];
                const int cout = P.dims[l + 1];
                RoMlp ma = {act, wl + P.woff[l], uact, pad4(cin) / 4, cout, N, wave, lane, l == P.n_layers - 1};
                if (mtiles(cout) == 1) ro_mlp_layer<1>(ma);
                else ro_mlp_layer<2>(ma);
            }
        }
        __syncthreads();
        RO_STAMP(2);
        // -------------------------------------------------------------- C: integrate (fp64, spec section 1)
        if (tid < N) {
            double px = spx[tid], py = spy[tid], vx = svx[tid], vy = svy[tid];
            integrate_one(px, py, vx, vy, uact + tid, N, tid < p.n_leaders, p);
            spx[tid] = px; spy[tid] = py; svx[tid] = vx; svy[tid] = vy;
        }
        __syncthreads();
        RO_STAMP(3);
        // -------------------------------------------------------------- D: reward (one wave) + pairwise pass
        if (wave == RO_WAVES - 1 && rewards != nullptr) {
            double sx = 0.0, sy = 0.0;
            for (int i = lane; i < N; i += 64) { sx += svx[i]; sy += svy[i]; }
            sx = mgp_wave_sum(sx); sy = mgp_wave_sum(sy);
            const double mx = sx / (double)N, my = sy / (double)N;
            double dv = 0.0;
            for (int i = lane; i < N; i += 64) {
                const double ex = svx[i] - mx, ey = svy[i] - my;
                dv += ex * ex + ey * ey;
            }
            const double var = mgp_wave_sum(dv) / (double)N;
            if (lane == 0) rewards[(size_t)b * T + t] = -1.0 * var * p.reward_scale;
        }
        {
            double deg = 0.0, f0 = 0, f1 = 0, f2 = 0, f3 = 0, f4 = 0, f5 = 0;
            unsigned long long lo = 0ull, hi = 0ull;
            if (pi < N) {
                const double xi = spx[pi], yi = spy[pi], vxi = svx[pi], vyi = svy[pi];
                const int j0 = piece * jh, j1 = min(N, j0 + jh);
                unsigned int mask = 0u;
                for (int j = j0; j < j1; j += 4) {
                    double ox[4], oy[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) { const int jj = min(j + q, j1 - 1); ox[q] = spx[jj]; oy[q] = spy[jj]; }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const double dx = xi - ox[q], dy = yi - oy[q];
                        const double r2 = dx * dx + dy * dy;
                        if (j + q < j1 && j + q != pi && r2 < R2) mask |= 1u << (j + q - j0);
                    }
                }
                unsigned int m2 = mask;
                while (m2) {                                  // ascending j: division + feature terms for neighbours only
                    const int j = j0 + __builtin_ctz(m2);
                    m2 &= m2 - 1u;
                    const double dx = xi - spx[j], dy = yi - spy[j];
                    const double r2 = dx * dx + dy * dy;
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
                if (j0 < 64) {
                    lo = (unsigned long long)mask << j0;
                    if (j0 + jh > 64) hi = (unsigned long long)mask >> (64 - j0);     // j0 >= 49 here
                } else {
                    hi = (unsigned long long)mask << (j0 - 64);
                }
            }
#pragma unroll
            for (int s = 1; s < RO_PIECES; s <<= 1) {         // the 8 pieces of a row are adjacent lanes
                deg += __shfl_xor(deg, s, MGP_WAVE);
                f0 += __shfl_xor(f0, s, MGP_WAVE); f1 += __shfl_xor(f1, s, MGP_WAVE); f2 += __shfl_xor(f2, s, MGP_WAVE);
                f3 += __shfl_xor(f3, s, MGP_WAVE); f4 += __shfl_xor(f4, s, MGP_WAVE); f5 += __shfl_xor(f5, s, MGP_WAVE);
                lo |= __shfl_xor(lo, s, MGP_WAVE); hi |= __shfl_xor(hi, s, MGP_WAVE);
            }
            if (piece == 0 && pi < N) {
                const double w = p.mean_pooling ? 1.0 / (deg == 0.0 ? 1.0 : deg) : 1.0;
                wrow[pi] = (float)w;
                rowmask[2 * pi] = lo; rowmask[2 * pi + 1] = hi;
                float* xn = XT + ((size_t)(cur + 1 == K ? 0 : cur + 1) * N + pi) * 8;     // overwrites the oldest tap
                *reinterpret_cast<float4*>(xn) = make_float4((float)f0, (float)f1, (float)f2, (float)f3);
                *reinterpret_cast<float4*>(xn + 4) = make_float4((float)f4, (float)f5, 0.f, 0.f);
            }
        }
        __syncthreads();
        RO_STAMP(4);
        // -------------------------------------------------------------- E: operator transition
        for (int j = K - 1; j >= 2; --j) {                    // G_j <- A_t . G_{j-1}   (slice j lives at index j - 1)
            float* dst = Gd + (size_t)(j - 1) * NN;
            const float* src = Gd + (size_t)(j - 2) * NN + hl * 4;
            for (int i = hw; i < N; i += RO_THREADS / 32) {
                unsigned long long mlo = rowmask[2 * i], mhi = rowmask[2 * i + 1];
                const float w = wrow[i];
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                while (mlo | mhi) {                           // chunks of 8 source rows, loads first
                    int idx[8]; float wv[8];
#pragma unroll
                    for (int d = 0; d < 8; ++d) {
                        const bool ok = (mlo | mhi) != 0ull;
                        int l = 0;
                        if (mlo) { l = __builtin_ctzll(mlo); mlo &= mlo - 1ull; }
                        else if (mhi) { l = 64 + __builtin_ctzll(mhi); mhi &= mhi - 1ull; }
                        idx[d] = l; wv[d] = ok ? w : 0.f;
                    }
                    if (hl < n4) {
                        float4 g[8];
#pragma unroll
                        for (int d = 0; d < 8; ++d) g[d] = *reinterpret_cast<const float4*>(src + idx[d] * N);
#pragma unroll
                        for (int d = 0; d < 8; ++d) {
                            acc.x = fmaf(wv[d], g[d].x, acc.x); acc.y = fmaf(wv[d], g[d].y, acc.y);
                            acc.z = fmaf(wv[d], g[d].z, acc.z); acc.w = fmaf(wv[d], g[d].w, acc.w);
                        }
                    }
                }
                if (hl < n4) *reinterpret_cast<float4*>(dst + i * N + hl * 4) = acc;
            }
            __syncthreads();
        }
        if (K >= 2) {                                         // G_1 <- A_t from the membership bits
            float4* d4 = reinterpret_cast<float4*>(Gd);
            const float inv_n4 = 1.0f / (float)n4;
            for (int e = tid; e < N * n4; e += RO_THREADS) {
                const int i = (int)(((float)e + 0.5f) * inv_n4);               // exact floor(e / n4)
                const int c0 = (e - i * n4) << 2;
                const unsigned int nib = (unsigned int)(rowmask[2 * i + (c0 >> 6)] >> (c0 & 63)) & 15u;
                const float w = wrow[i];
                d4[e] = make_float4((nib & 1u) ? w : 0.f, (nib & 2u) ? w : 0.f, (nib & 4u) ? w : 0.f, (nib & 8u) ? w : 0.f);
            }
        }
        cur = (cur + 1 == K) ? 0 : cur + 1;
        __syncthreads();
        RO_STAMP(5);
    }

    // ------------------------------------------------------------------ exit: LDS -> the caller's buffers
    {
        const float4* gsrc = reinterpret_cast<const float4*>(Gd);
        float4* gdst = reinterpret_cast<float4*>(Gb + NN);
        const int tot4 = (K - 1) * NN / 4;
#pragma unroll 8
        for (int e = tid; e < tot4; e += RO_THREADS) gdst[e] = gsrc[e];
    }
    for (int e = tid; e < K * 6 * N; e += RO_THREADS) {
        const int k = e / (6 * N), r1 = e - k * 6 * N, f = r1 / N, n = r1 - f * N;
        Xb[e] = XT[((size_t)ro_slot(cur, k, K) * N + n) * 8 + f];
    }
    for (int i = tid; i < N; i += RO_THREADS) {
        xb[i * 4 + 0] = spx[i]; xb[i * 4 + 1] = spy[i]; xb[i * 4 + 2] = svx[i]; xb[i * 4 + 3] = svy[i];
    }
    if (action != nullptr)
        for (int e = tid; e < 2 * N; e += RO_THREADS) action[(size_t)b * 2 * N + e] = uact[e];
}

// LDS plan; returns false when the shape is outside the kernel's coverage
bool make_carve(const int* dims, int n_layers, int K, int N, RoParams* P, RoCarve* cv)
{
    if (dims == nullptr || n_layers < 1 || n_layers > MGP_MAX_LAYERS) return false;
    if (K < 1 || K > 8 || N < 4 || N > RO_MAXN || (N & 3)) return false;
    if (dims[0] != 6 || dims[n_layers] != 2) return false;                      // simulator: 6 features in, 2-D action out
    int wtot = 0;
    for (int l = 0; l < n_layers; ++l) {
        const int cin = (l == 0) ? 6 * K : dims[l], cout = dims[l + 1];
        if (cin < 1 || cout < 1 || cin > AF_MAXW || cout > 32) return false;   // MT <= 2: fits 128 VGPRs at 16 waves
        if (P) { P->woff[l] = wtot; P->dims[l] = dims[l]; }
        wtot += mtiles(cout) * 64 * AF_WFS + mtiles(cout) * 16;
    }
    if (P) { P->dims[n_layers] = dims[n_layers]; P->n_layers = n_layers; }
    int off = 0;
    auto take = [&off](int bytes) { const int o = off; off += (bytes + 15) & ~15; return o; };
    RoCarve c;
    c.pos = take(4 * N * 8);
    c.mask = take(2 * N * 8);
    c.wrow = take(N * 4);
    c.uact = take(2 * N * 4);
    c.xt = take(K * N * 8 * 4);
    c.gd = take((K - 1) * N * N * 4);
    c.wl = take(wtot * 4);
    c.act = take(pad16(N) * AF_CS * 4);
    c.total = off;
    int rp = 1;
    while (K > 1 && rp < 8 && 2 * rp * (K - 1) * N <= RO_THREADS) rp *= 2;
    if (K > 1 && (K - 1) * N > RO_THREADS) return false;
    c.rps = (rp == 1) ? 0 : (rp == 2) ? 1 : (rp == 4) ? 2 : 3;
    if (c.total > RO_LDS_LIMIT) return false;
    if (cv) *cv = c;
    return true;
}

}  // namespace

extern "C" int mgp_rollout_supported(const int* dims, int n_layers, int K, int N)
{
    return make_carve(dims, n_layers, K, N, nullptr, nullptr) ? 1 : 0;
}

extern "C" int mgp_rollout_steps(double* x, float* G, float* Xd, const float* const* W, const float* const* b,
                                 const int* dims, int n_layers, float* action, double* rewards,
                                 const MgpFlockParams* p, int B, int K, int N, int T, void* stream)
{
    if (B < 0 || T < 0 || p == nullptr || W == nullptr || b == nullptr) return MGP_EINVAL;
    if (!(p->comm_radius2 > 0.0) || !(p->dt > 0.0) || p->n_leaders < 0) return MGP_EINVAL;
    RoParams P;
    RoCarve cv;
    if (!make_carve(dims, n_layers, K, N, &P, &cv)) return MGP_EUNSUPPORTED;
    if (B == 0 || T == 0) return MGP_OK;
    MGP_CHECK_PTR8(x);
    MGP_CHECK_PTR(G);
    MGP_CHECK_PTR(Xd);
    if (!mgp_aligned16(G)) return MGP_EALIGN;
    if (action != nullptr && (reinterpret_cast<uintptr_t>(action) & 3u)) return MGP_EALIGN;
    if (rewards != nullptr && (reinterpret_cast<uintptr_t>(rewards) & 7u)) return MGP_EALIGN;
    for (int l = 0; l < n_layers; ++l) {
        MGP_CHECK_PTR(W[l]);
        MGP_CHECK_PTR(b[l]);
        P.W[l] = W[l]; P.b[l] = b[l];
    }
    mgp_clear_error();
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(rollout_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            cv.total) != hipSuccess)
        return MGP_ELAUNCH;
    hipLaunchKernelGGL(rollout_kernel, dim3(B), dim3(RO_THREADS), cv.total, static_cast<hipStream_t>(stream), x, G, Xd,
                       action, rewards, P, cv, *p, K, N, T);
    return mgp_launch_status();
}
