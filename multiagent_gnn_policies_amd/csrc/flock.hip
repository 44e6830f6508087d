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
#include <cstdlib>
#include "mgp_device.h"

namespace {

// Optional in-kernel phase timestamps (tools/harness/flock_phase_prof.hip defines MGP_FL_PROFILE; never in the product).
#ifdef MGP_FL_PROFILE
__device__ unsigned long long mgp_fl_stamps[32];
#define FL_STAMP(i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) mgp_fl_stamps[i] = __builtin_readcyclecounter(); } while (0)
#else
#define FL_STAMP(i) do { } while (0)
#endif

constexpr int FL_THREADS = 1024;                // in-place mode: one workgroup of 1024 threads per 128 rows
constexpr int FL_ROWS = 128;
constexpr int FP_THREADS = 256;                 // ping-pong mode (x_out != x): 256 threads per 32 rows -> 4 workgroups per
constexpr int FP_ROWS = 32;
constexpr int FP_PIECES = 8;                    // 8 j-pieces per row for the pairwise phases (256 threads); a wider
                                                // workgroup adds half-waves for the fused delayed-GSO rows
// N = 100 episode, 16 threads per row, several workgroups per CU


template <int WAVES>
__device__ __forceinline__ double block_sum(double v, double* sh /* [WAVES] */)
{
    v = mgp_wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) t += sh[w];
    return t;
}

struct FlockOut {
    float* A; double* A64; float* feat; double* feat64; double* reward;
    float* expert; double* expert64; int centralized;
    int sep_reward;         // 1: an extra workgroup per episode computes the reward (illegal while integrating in place)
    long sAb, sFb;          // batch strides (elements) of A and feat: lets the sim write straight into delay_gso[:,1] / delay_state[:,0]
    // fused state transition (mgp_flock_step_advance): A = Gn slice 1, feat = Xn tap 0, then
    // Gn[b,j] = A_t . Gp[b,j-1] (j >= 2) from the membership bits, Xn[b,j] = Xp[b,j-1] (j >= 1)
    int adv, K, has_prev; const float* Gp; float* Gn; const float* Xp; float* Xn;
    int vecA;               // 1: fp32 network rows may be written with 16-byte stores (N % 4 == 0, aligned, no fp64 copy)
    // sparse outputs (mgp_flock_step_sparse): the network as membership bit rows + row weights, features as (N, 8) rows
    unsigned long long* bits; float* wq; float* featT; long sBb, sWb, sTb;     // batch strides in words / floats / floats
};


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

// grid: x = row chunk, y = b.  LDS (doubles): px,py,vx,vy [N] | part [SPLIT][8][ROWS] | wrow [ROWS] | adjacency bits
// FUSE_INTEGRATE: the workgroup integrates the whole episode into LDS itself.  In-place mode (xo == x) that is only
// legal with ONE workgroup per episode; ping-pong mode (xo != x) lets every workgroup of the episode do it redundantly
// (reads x, writes only its own rows of xo), which is what allows small row tiles and many workgroups per episode.
template <bool FUSE_INTEGRATE, int THREADS, int ROWS, int PIECES>
__global__ __launch_bounds__(THREADS)
void flock_step_kernel(const double* __restrict__ x, double* __restrict__ xo, const float* __restrict__ u,
                       long su_agent, long su_axis, FlockOut o, MgpFlockParams p, int N)
{
    constexpr int FL_SPLIT = PIECES;                       // pairwise threads = ROWS * PIECES (<= THREADS); the rest of
                                                           // the workgroup only helps with loads, the row sweep and the gso rows
    constexpr int FL_WAVES = THREADS / 64;
    constexpr int FL_ROWS = ROWS;
    constexpr int FL_THREADS = THREADS;
    extern __shared__ __attribute__((aligned(16))) double sm[];
    __shared__ double sh[FL_WAVES];
    double* spx = sm; double* spy = sm + N; double* svx = sm + 2 * (size_t)N; double* svy = sm + 3 * (size_t)N;
    double* part = sm + 4 * (size_t)N;                     // [FL_SPLIT][FL_ROWS][8]: deg,f0..f5 of each j piece
    double* wrow = part + FL_SPLIT * FL_ROWS * 8;                     // [FL_ROWS] network weight of the row (fp64)
    unsigned long long* adjw = reinterpret_cast<unsigned long long*>(wrow + FL_ROWS);   // [FL_ROWS][FL_SPLIT][nch]
    // Workgroup -> (episode b, tile bx).  The dispatcher hands workgroup L = blockIdx.y * gridDim.x + blockIdx.x to XCD
    // L % 8 (observed; used for speed only), and each XCD has its own L2.  With the plain mapping the tiles of one episode
    // sit on different XCDs and every one of them pulls the source rows of G_prev[b] (re-read deg times in the fused
    // transition below) and the episode's state from HBM separately: PMC showed 3.2x the algorithmic read bytes.  With
    // b % 8 == L % 8 all tiles of an episode share one L2 (same remap as gso.hip).
    int b = blockIdx.y, bx = blockIdx.x;
    if ((gridDim.y & 7u) == 0u) {
        const int slots = gridDim.x;
        const int L = blockIdx.y * gridDim.x + blockIdx.x;
        const int grp = L / (8 * slots), rem8 = L - grp * (8 * slots);
        bx = rem8 >> 3; b = grp * 8 + (rem8 & 7);
    }
    const int tid = threadIdx.x;
    const int ntiles = (N + FL_ROWS - 1) / FL_ROWS;
    // an extra workgroup (bx == ntiles) only computes the episode's reward, so that no row workgroup has the
    // three block reductions (3.7k cycles) on its critical path
    const bool reward_wg = (o.sep_reward || o.adv) && bx == ntiles;
    const bool does_reward = o.reward != nullptr && (o.sep_reward ? reward_wg : bx == 0);
    const int i0 = reward_wg ? N : bx * FL_ROWS;
    const int rows = reward_wg ? 0 : min(FL_ROWS, N - i0);
    const double* xb = x + (size_t)b * N * 4;
    double* xob = xo + (size_t)b * N * 4;
    const bool all_rows = (xo == x);                       // in-place: this workgroup owns the whole episode's state

    FL_STAMP(0);
    // ---- load (and, when fused, integrate) every agent of the episode into LDS
    double sum_vx = 0.0, sum_vy = 0.0;
    for (int i = tid; i < N; i += FL_THREADS) {
        double px = xb[i * 4 + 0], py = xb[i * 4 + 1], vx = xb[i * 4 + 2], vy = xb[i * 4 + 3];
        if (FUSE_INTEGRATE && u != nullptr) {
            integrate_one(px, py, vx, vy, u + (size_t)b * N * 2 + (size_t)i * su_agent, su_axis, i < p.n_leaders, p);
            if (all_rows || (i >= i0 && i < i0 + rows)) {
                xob[i * 4 + 0] = px; xob[i * 4 + 1] = py; xob[i * 4 + 2] = vx; xob[i * 4 + 3] = vy;
            }
        }
        spx[i] = px; spy[i] = py; svx[i] = vx; svy[i] = vy;
        sum_vx += vx; sum_vy += vy;
    }
    __syncthreads();
    FL_STAMP(1);
    // ---- episode-level sums (reward, centralised controller): every workgroup of the episode recomputes them
    double tot_vx = 0.0, tot_vy = 0.0;
    const bool need_cent = o.centralized && (o.expert != nullptr || o.expert64 != nullptr);
    if (does_reward || need_cent) {                                 // workgroup-uniform condition
        tot_vx = block_sum<FL_WAVES>(sum_vx, sh);
        tot_vy = block_sum<FL_WAVES>(sum_vy, sh);
        if (does_reward) {
            const double mx = tot_vx / (double)N, my = tot_vy / (double)N;
            double dv = 0.0;
            for (int i = tid; i < N; i += FL_THREADS) {
                const double ex = svx[i] - mx, ey = svy[i] - my;
                dv += ex * ex + ey * ey;
            }
            const double var = block_sum<FL_WAVES>(dv, sh) / (double)N;
            if (tid == 0) o.reward[b] = -1.0 * var * p.reward_scale;
        }
    }
    if (reward_wg) {
        if (o.adv && o.K > 1) {                              // delay line: taps 1..K-1 <- previous taps 0..K-2
            const long tap = 6L * N, per = (long)o.K * tap;
            float* xn = o.Xn + (long)b * per;
            const float* xp = o.Xp + (long)b * per;
            for (long e = tap + tid; e < per; e += FL_THREADS) xn[e] = o.has_prev ? xp[e - tap] : 0.f;
        }
        return;
    }
    FL_STAMP(2);
    // ---- pairwise pass: thread = (row, j-half)
    const bool pair_active = tid < FL_ROWS * FL_SPLIT;
    const int rl = tid % FL_ROWS, half = pair_active ? tid / FL_ROWS : 0;
    const int i = i0 + rl;
    const double R2 = p.comm_radius2;
    double deg = 0.0, f0 = 0, f1 = 0, f2 = 0, f3 = 0, f4 = 0, f5 = 0;
    // j's per piece; sparse outputs: a multiple of 64, so that the (row, piece, word) masks below ARE the row's bit words
    const int jh = (o.bits != nullptr) ? ((((N + FL_SPLIT - 1) / FL_SPLIT) + 63) & ~63) : (N + FL_SPLIT - 1) / FL_SPLIT;
    const int nch = (jh + 63) / 64;                        // 64-bit adjacency words per (row, piece)
    if (pair_active && rl < rows) {
        const double xi = spx[i], yi = spy[i], vxi = svx[i], vyi = svy[i];
        const unsigned int wi = p.link_drop != 0u ? fade_word(xi, yi) : 0u;
        const int j0 = half * jh, j1 = min(N, j0 + jh);
        for (int c = 0; c < nch; ++c) {
            // phase 1: cheap membership test for up to 64 j's -> bit mask (the only fp64 work every pair pays)
            const int ja = j0 + 64 * c, jb = min(j1, ja + 64);
            unsigned long long mask = 0ull;
            for (int j = ja; j < jb; j += 4) {                 // explicit batches of 4: the LDS reads of a batch are
                double ox[4], oy[4];                           // issued together instead of one dependent chain per j
#pragma unroll
                for (int q = 0; q < 4; ++q) { const int jj = min(j + q, jb - 1); ox[q] = spx[jj]; oy[q] = spy[jj]; }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const double dx = xi - ox[q], dy = yi - oy[q];
                    const double r2 = dx * dx + dy * dy;
                    if (j + q < jb && j + q != i && r2 < R2) mask |= 1ull << (j + q - ja);
                }
            }
            if (p.link_drop != 0u) {                           // FlockingStochastic-v0: faded links leave the mask
                unsigned long long m = mask;
                while (m) {
                    const int j = ja + __builtin_ctzll(m);
                    m &= m - 1ull;
                    if (!link_up(p, i, j, N, wi, fade_word(spx[j], spy[j]))) mask &= ~(1ull << (j - ja));
                }
            }
            adjw[((size_t)rl * FL_SPLIT + half) * nch + c] = mask;
            // phase 2: the division and the six feature terms only for actual neighbours, ascending j.  A wave runs
            // this body max-over-lanes(popcount) times (~2-3) instead of once per j (13), which is where the
            // divergent fp64 division used to dominate.
            while (mask) {
                const int j = ja + __builtin_ctzll(mask);
                mask &= mask - 1ull;
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
        }
        FL_STAMP(3);
        if (half > 0) {
            // value-major layout part[h][k][row]: consecutive lanes (rows) hit consecutive banks (the row-major
            // [row][8] layout was a 16-way bank conflict on every one of these stores and of the reads below)
            double* pr = part + (size_t)half * 8 * FL_ROWS + rl;
            pr[0 * FL_ROWS] = deg; pr[1 * FL_ROWS] = f0; pr[2 * FL_ROWS] = f1; pr[3 * FL_ROWS] = f2;
            pr[4 * FL_ROWS] = f3; pr[5 * FL_ROWS] = f4; pr[6 * FL_ROWS] = f5;
        }
    }
    __syncthreads();
    if (pair_active && half == 0 && rl < rows) {
#pragma unroll 2      // NOT fully: 15 pieces x 7 doubles in flight cost 222 VGPRs and the second resident workgroup
        for (int h = 1; h < FL_SPLIT; ++h) {                 // ascending j pieces: deterministic
            const double* pr = part + (size_t)h * 8 * FL_ROWS + rl;
            deg += pr[0 * FL_ROWS]; f0 += pr[1 * FL_ROWS]; f1 += pr[2 * FL_ROWS]; f2 += pr[3 * FL_ROWS];
            f3 += pr[4 * FL_ROWS]; f4 += pr[5 * FL_ROWS]; f5 += pr[6 * FL_ROWS];
        }
        wrow[rl] = p.mean_pooling ? 1.0 / (deg == 0.0 ? 1.0 : deg) : 1.0;
        if (o.feat != nullptr) {
            float* fb = o.feat + (size_t)b * o.sFb + i;
            fb[0 * (size_t)N] = (float)f0; fb[1 * (size_t)N] = (float)f1; fb[2 * (size_t)N] = (float)f2;
            fb[3 * (size_t)N] = (float)f3; fb[4 * (size_t)N] = (float)f4; fb[5 * (size_t)N] = (float)f5;
        }
        if (o.bits != nullptr) {
            o.wq[(size_t)b * o.sWb + i] = (float)wrow[rl];
            float* ft = o.featT + (size_t)b * o.sTb + (size_t)i * 8;
            *reinterpret_cast<float4*>(ft) = make_float4((float)f0, (float)f1, (float)f2, (float)f3);
            *reinterpret_cast<float4*>(ft + 4) = make_float4((float)f4, (float)f5, 0.f, 0.f);
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
    FL_STAMP(4);
    if (o.bits != nullptr) {                              // the masks of this workgroup's rows, [row][piece][word] = [row][nwords]
        const int nwords = FL_SPLIT * nch;
        unsigned long long* gb = o.bits + (size_t)b * o.sBb + (size_t)i0 * nwords;
        for (int idx = tid; idx < rows * nwords; idx += FL_THREADS) gb[idx] = adjw[idx];
    }
    if (o.A == nullptr && o.A64 == nullptr) return;
    __syncthreads();
    FL_STAMP(5);
    // ---- fused state transition only: every row's ascending neighbour list, filled by all pairwise threads at once
    //      (thread (row, piece) knows its own membership word; its write position is the popcount of the row's
    //      earlier words).  rlist[row][N] ints live behind the membership words.
    // (one byte per entry: the fused path covers N <= 128; 12.8 KB of int lists cost the sixth resident workgroup per CU)
    unsigned char* rlist = reinterpret_cast<unsigned char*>(adjw + (size_t)FL_ROWS * FL_SPLIT * nch);
    if (o.adv && o.K > 2 && pair_active && rl < rows) {
        const unsigned long long* wr_ = adjw + (size_t)rl * FL_SPLIT * nch;
        int pos = 0;
        for (int t = 0; t < half * nch; ++t) pos += __popcll(wr_[t]);
        for (int c = 0; c < nch; ++c) {
            unsigned long long m = wr_[half * nch + c];
            const int jbase = half * jh + 64 * c;
            while (m) { rlist[(size_t)rl * N + pos++] = (unsigned char)(jbase + __builtin_ctzll(m)); m &= m - 1ull; }
        }
    }
    // The product rows come BEFORE the network rows: loads and stores of a wave complete in issue order, so gathers issued behind
    // the 12.8 KB store burst of the network rows would wait for its acknowledgements as well; the network rows need nothing back
    // and drain while the workgroup retires.
    if (o.adv && o.K > 2) {
    __syncthreads();                                              // neighbour lists complete
    // ---- fused delayed-GSO product for this workgroup's rows: Gn[b,j,i,:] = sum_{p in N(i)} w_i * Gp[b,j-1,p,:], j >= 2.
    //      Same arithmetic, order and weights as gso_rows_half_kernel (bit-identical), but the neighbour list comes from
    //      the membership bits instead of re-reading and compacting the dense row of A.  One half-wave per row.
    {
        constexpr int HW = FL_THREADS / 32;                       // half-waves in the workgroup
        const int lane = tid & 63, hw = tid >> 5, hl = lane & 31;
        const size_t NN = (size_t)N * N;
        const int nwords = FL_SPLIT * nch, n4 = N / 4;
        // a half-wave's rows (<= FL_ROWS / HW) are all accumulated before the first of them is stored: a gather issued behind
        // a store would wait for that store's acknowledgement too
        constexpr int RPH = (FL_ROWS + HW - 1) / HW;
        for (int j = 2; j < o.K; ++j) {
            float4 accs[RPH];
#pragma unroll
            for (int ri = 0; ri < RPH; ++ri) {
                const int r0 = hw + ri * HW;
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                if (r0 < rows && hl < n4 && o.has_prev) {
                    const unsigned char* mylist = rlist + (size_t)r0 * N;
                    int cnt = 0;
                    for (int t = 0; t < nwords; ++t) cnt += __popcll(adjw[(size_t)r0 * nwords + t]);
                    const float w = (float)wrow[r0];
                    const float* sj = o.Gp + (size_t)b * o.K * NN + (size_t)(j - 1) * NN + hl * 4;
                    // chunks of 8 source rows, all 8 loads issued before the first FMA (one L2 round trip per chunk,
                    // typical degree <= 8); entries past the list end re-read row 0 with weight 0 (adds exact zeros)
                    for (int e = 0; e < cnt; e += 8) {
                        float4 g[8];
                        float wv[8];
#pragma unroll
                        for (int d = 0; d < 8; ++d) {
                            const bool ok = (e + d) < cnt;
                            const int m = ok ? mylist[e + d] : 0;
                            g[d] = *reinterpret_cast<const float4*>(sj + (size_t)m * N);
                            wv[d] = ok ? w : 0.f;
                        }
#pragma unroll
                        for (int d = 0; d < 8; ++d) {
                            acc.x = fmaf(wv[d], g[d].x, acc.x); acc.y = fmaf(wv[d], g[d].y, acc.y);
                            acc.z = fmaf(wv[d], g[d].z, acc.z); acc.w = fmaf(wv[d], g[d].w, acc.w);
                        }
                    }
                }
                accs[ri] = acc;
            }
#pragma unroll
            for (int ri = 0; ri < RPH; ++ri) {
                const int r0 = hw + ri * HW;
                if (r0 < rows && hl < n4)
                    *reinterpret_cast<float4*>(o.Gn + (size_t)b * o.K * NN + (size_t)j * NN + (size_t)(i0 + r0) * N + hl * 4) = accs[ri];
            }
        }
    }
    }
    FL_STAMP(6);
    // ---- network rows i0..i0+rows-1: one flat coalesced sweep; membership comes from the phase-1 bit masks
    const size_t base = ((size_t)b * N + i0) * N;
    const size_t baseA = (size_t)b * o.sAb + (size_t)i0 * N;
    const float inv_jh = 1.0f / (float)jh, inv_n = 1.0f / (float)N;
    if (o.vecA) {
        // float4 form: thread -> (row, group of 4 columns); 4x fewer stores and index computations than the scalar sweep
        const int n4 = N >> 2, total4 = rows * n4;
        const float inv_n4 = 1.0f / (float)n4;
        const bool zero_all = o.adv && !o.has_prev;
        for (int idx0 = tid; idx0 < total4; idx0 += 4 * FL_THREADS) {
            float4 outv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int idx = min(idx0 + q * FL_THREADS, total4 - 1);
                const int ri = (int)(((float)idx + 0.5f) * inv_n4);       // exact floor(idx / n4)
                const int j0 = (idx - ri * n4) << 2;
                const float w = zero_all ? 0.f : (float)wrow[ri];
                float v[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int j = j0 + c;
                    const int piece = (int)(((float)j + 0.5f) * inv_jh);   // exact floor(j / jh)
                    const int off = j - piece * jh;
                    const unsigned long long wbits = adjw[((size_t)ri * FL_SPLIT + piece) * nch + (off >> 6)];
                    v[c] = ((wbits >> (off & 63)) & 1ull) ? w : 0.f;
                }
                outv[q] = make_float4(v[0], v[1], v[2], v[3]);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int idx = idx0 + q * FL_THREADS;
                if (idx < total4) *reinterpret_cast<float4*>(o.A + baseA + (size_t)idx * 4) = outv[q];
            }
        }
    } else {
    const int total = rows * N;
    for (int idx0 = tid; idx0 < total; idx0 += 4 * FL_THREADS) {        // batches of 4 independent elements per thread
        unsigned long long wb[4]; double wr[4]; int off[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int idx = min(idx0 + q * FL_THREADS, total - 1);
            const int ri = (int)(((float)idx + 0.5f) * inv_n);         // exact floor(idx / N) for idx < 2^20
            const int j = idx - ri * N;
            const int piece = (int)(((float)j + 0.5f) * inv_jh);       // exact floor(j / jh)
            off[q] = j - piece * jh;
            wb[q] = adjw[((size_t)ri * FL_SPLIT + piece) * nch + (off[q] >> 6)];
            wr[q] = wrow[ri];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int idx = idx0 + q * FL_THREADS;
            if (idx < total) {
                const double w = (((wb[q] >> (off[q] & 63)) & 1ull) && !(o.adv && !o.has_prev)) ? wr[q] : 0.0;
                if (o.A != nullptr) o.A[baseA + idx] = (float)w;
                if (o.A64 != nullptr) o.A64[base + idx] = w;
            }
        }
    }
    }
    FL_STAMP(7);
}

int check_params(const MgpFlockParams* p)
{
    if (p == nullptr) return MGP_EINVAL;
    if (!(p->comm_radius2 > 0.0) || !(p->dt > 0.0)) return MGP_EINVAL;
    if (p->n_leaders < 0) return MGP_EINVAL;
    return MGP_OK;
}

template <bool FUSE, int THREADS, int ROWS, int PIECES>
int launch_step(const double* x, double* xo, const float* u, long su_agent, long su_axis, const FlockOut& o,
                const MgpFlockParams* p, int B, int N, hipStream_t st)
{
    const int jh = (o.bits != nullptr) ? ((((N + PIECES - 1) / PIECES) + 63) & ~63) : (N + PIECES - 1) / PIECES;
    const size_t lds = ((size_t)4 * N + PIECES * ROWS * 8 + ROWS + (size_t)ROWS * PIECES * ((jh + 63) / 64)) *
                           sizeof(double) + (o.adv ? (((size_t)ROWS * N + 15) & ~(size_t)15) : 0);     // + one byte-indexed neighbour list per row
    if (mgp_allow_dyn_lds(reinterpret_cast<const void*>(flock_step_kernel<FUSE, THREADS, ROWS, PIECES>), lds) != hipSuccess)
        return MGP_ELAUNCH;
    dim3 grid(mgp_ceil_div(N, ROWS) + ((o.sep_reward || o.adv) ? 1 : 0), B);
    FlockOut ov = o;
    ov.vecA = (o.A != nullptr && o.A64 == nullptr && (N & 3) == 0 && (o.sAb & 3) == 0 && mgp_aligned16(o.A)) ? 1 : 0;
    hipLaunchKernelGGL((flock_step_kernel<FUSE, THREADS, ROWS, PIECES>), grid, dim3(THREADS), lds, st, x, xo, u, su_agent,
                       su_axis, ov, *p, N);
    return mgp_launch_status();
}

int launch_flock(double* x, double* x_out, const float* u, long su_agent, long su_axis, const FlockOut& o,
                 const MgpFlockParams* p, int B, int N, hipStream_t st)
{
    mgp_clear_error();
    FlockOut os = o;
    if (x_out != nullptr && x_out != x && u != nullptr) {
        // ping-pong: every workgroup integrates the episode redundantly from x and writes its rows of x_out
        os.sep_reward = o.reward != nullptr;
        return launch_step<true, FP_THREADS, FP_ROWS, FP_PIECES>(x, x_out, u, su_agent, su_axis, os, p, B, N, st);
    }
    if (N <= FL_ROWS) {
        os.sep_reward = (o.reward != nullptr) && (u == nullptr);       // nothing is integrated: x is read-only
        return launch_step<true, FL_THREADS, FL_ROWS, FL_THREADS / FL_ROWS>(x, x, u, su_agent, su_axis, os, p, B, N, st);
    }
    if (u != nullptr) {
        hipLaunchKernelGGL(flock_integrate_kernel, dim3(B), dim3(FL_THREADS), 0, st, x, u, su_agent, su_axis, *p, N);
        const int rc = mgp_launch_status();
        if (rc != MGP_OK) return rc;
    }
    os.sep_reward = o.reward != nullptr;                               // x was integrated by the kernel above
    return launch_step<false, FP_THREADS, FP_ROWS, FP_PIECES>(x, x, u, su_agent, su_axis, os, p, B, N, st);
}

// ---------------------------------------------------------------------------------------------------------------------
// [r5] mgp_flock_step_advance for N <= 128: ONE 1024-thread workgroup per episode.
// The row-tiled kernel above runs five workgroups per episode (four row tiles + the reward), each of which loads and integrates
// the whole episode, tests its rows against every agent in fp64, meets its j-pieces in LDS and then gathers the product rows
// G_j = A_t . G_{j-1}(prev) from L2 one dependent round trip per four rows: 21 us per 256 episodes at N = 100, 0.21 of the HBM
// rate on its 35.8 MB, for four rounds.  Here the episode's chain runs once -- integration, the membership test of the resident
// kernel (an fp32 test on coordinates relative to agent 0 decides every pair that clears the radius by a proven error band, the
// spec's fp64 expression the rest: the bits are the oracle's), fp64 feature terms of actual neighbours -- while the source
// slice G_{j-1}(prev) streams into LDS with coalesced 16-byte loads; the product rows are then gathered from LDS, and all
// stores (network rows, product rows, features, delay line, agent states) are flat coalesced sweeps.
// Bit-identical to the row-tiled kernel (tests/test_gpu_kernels.py::test_fused_sim_state_kernel_equals_two_kernel_protocol):
// same per-axis integration, the same j-pieces (eight per row, ceil(N / 8) candidates each) summed in ascending order, one true
// division per neighbour, the same reduction tree for the velocity sums, product rows accumulated along the ascending list.
// (Measured and dropped, LAB_NOTES.md round 5: two 512-thread workgroups per episode, each with half the rows -- both stage the
//  whole source slice, 20 MB instead of 10 through the CUs' memory pipelines in front of the agent states: 15.9 us against 13.4.)
constexpr int FA_THREADS = 1024;
constexpr int FA_WAVES = FA_THREADS / 64;
constexpr int FA_DELAY_ELEMS = 3072;                        // delay-line elements the kernel copies, (K - 1) 6 N: 4 taps of 6 x 128

struct FaOff { int pos, sxy, bits, wrow, rcnt, rlist, stage, total; };
__host__ __device__ inline int fa_take(int& off, int bytes) { const int o = off; off += (bytes + 15) & ~15; return o; }
__host__ __device__ inline int fa_list_stride(int N) { const int w = (N + 3) >> 2; return 4 * (w | 1); }
__host__ __device__ inline FaOff fa_offsets(int N, int K)
{
    FaOff c = {};
    int off = 0;
    c.pos = fa_take(off, (4 * N + 2) * 8);                  // double px, py, vx, vy [4][N] + reference point
    c.sxy = fa_take(off, N * 8);                            // float2 [N] coordinates relative to the reference point
    c.bits = fa_take(off, N * 16);                          // u64 [N][2] membership bits of the new network
    c.wrow = fa_take(off, N * 8);                           // double [N] row weights
    c.rcnt = fa_take(off, N * 4);
    c.rlist = fa_take(off, N * fa_list_stride(N));          // u8 [N][RS] ascending neighbour lists
    c.stage = fa_take(off, K > 2 ? N * N * 4 : 0);          // float [N][N]: one source slice G_{j-1}(prev)
    c.total = off;
    return c;
}

// 16 bytes per active lane straight into LDS: LDS[dst_uniform + 16 * lane] = *src (global_load_lds_dwordx4; M0 carries the LDS
// base; asm rather than the builtin, see actor_fused.hip).  hipcc does not count it: the caller waits (s_waitcnt vmcnt(0)) before
// the barrier in front of the first read, and issues it BEHIND the last load whose data it waits for itself (loads return in order).
__device__ __forceinline__ void fa_lds_dma16(const void* src, const void* dst_uniform)
{
    const unsigned int base = __builtin_amdgcn_readfirstlane((unsigned int)reinterpret_cast<uintptr_t>(dst_uniform));
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(base), "v"(src) : "memory", "m0");
}

// four entries per trip: the list bytes, then the four source rows in flight together, then the multiply-adds in list order
// (entries past the end re-read a valid row with weight 0: fma(0, g, acc) == acc; a row's list bytes are valid row indices)
__device__ __forceinline__ void fa_gather(float4& acc, const float w, const int cnt, const unsigned char* lp, const float4* stage4,
                                          const int n4, const int c4)
{
    for (int q = 0; q < cnt; q += 4) {
        int m[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) m[d] = (q + d < cnt) ? (int)lp[q + d] : 0;
        float4 g[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) g[d] = stage4[m[d] * n4 + c4];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const float wd = (q + d < cnt) ? w : 0.f;
            acc.x = fmaf(wd, g[d].x, acc.x); acc.y = fmaf(wd, g[d].y, acc.y); acc.z = fmaf(wd, g[d].z, acc.z); acc.w = fmaf(wd, g[d].w, acc.w);
        }
    }
}

__global__ __launch_bounds__(FA_THREADS)
void flock_advance_kernel(const double* __restrict__ x, double* __restrict__ xo, const float* __restrict__ u, long su_agent,
                          long su_axis, FlockOut o, MgpFlockParams p, int N)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char fsm[];
    __shared__ double tot_sh[2];
    const int K = o.K;
    const FaOff cv = fa_offsets(N, K);
    double* spx = reinterpret_cast<double*>(fsm + cv.pos);
    double* spy = spx + N; double* svx = spx + 2 * N; double* svy = spx + 3 * N;
    float2* sxy = reinterpret_cast<float2*>(fsm + cv.sxy);
    unsigned long long* rowmask = reinterpret_cast<unsigned long long*>(fsm + cv.bits);
    double* wrow = reinterpret_cast<double*>(fsm + cv.wrow);
    int* rcnt = reinterpret_cast<int*>(fsm + cv.rcnt);
    unsigned char* rlist = fsm + cv.rlist;
    unsigned char* stage = fsm + cv.stage;
    const float4* stage4 = reinterpret_cast<const float4*>(stage);
    const int RS = fa_list_stride(N);
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t NN = (size_t)N * N;
    const int n4 = N >> 2, nn4 = N * n4;                    // float4 per row / per slice
    const double* xb = x + (size_t)b * N * 4;
    double* xob = xo + (size_t)b * N * 4;
    const float* Gp = o.Gp + (size_t)b * K * NN;
    float* Gn = o.Gn + (size_t)b * K * NN;
    const bool prod = K > 2 && o.has_prev;
    // source slice `sl` of G_prev -> LDS, 1 KB per wave and request, no registers, nothing to wait for until it is read.  hipcc does
    // not count these requests but the hardware does, and loads return in order: EVERY s_waitcnt vmcnt a DMA wave executes
    // afterwards -- its own lanes masked off or not -- blocks it until its share of the slice has landed.  [r6] So the requests
    // are issued by waves that have nothing to do until the slice is needed: waves 13 and 14 where the membership phase runs on
    // thirteen row waves (8 N <= 832: N <= 104), the upper eight waves otherwise (they then join the membership phase late).
    constexpr int FA_LOAD_WAVES = 8;                        // waves 0 .. 7 load the agent states and copy the delay line
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const bool spare = 8 * N <= 13 * 64;
    const int dma_w0 = spare ? 13 : FA_WAVES - 8, dma_n = spare ? 2 : 8;
    auto stage_slice = [&](const int sl) {
        const unsigned char* src = reinterpret_cast<const unsigned char*>(Gp + (size_t)sl * NN);
        const int bytes = (int)NN * 4;
        if (wave_u >= dma_w0 && wave_u < dma_w0 + dma_n)
            for (int c = (wave_u - dma_w0) * 1024; c < bytes; c += dma_n * 1024)
                if (c + lane * 16 < bytes) fa_lds_dma16(src + c + lane * 16, stage + c);
    };

    FL_STAMP(8);
    // Issuing is not free either: the CU's load path takes a 1 KB request every ~100 cycles, so the forty requests of a slice keep
    // their waves busy for ~4k cycles -- in front of the first barrier that held every wave back (stamp 9 at 4.3k cycles, and later
    // the later the DMA started: profiles/r06_flock_advance_stamps.txt).  The spare waves therefore issue BEHIND that barrier; the
    // slice lands ~4k cycles into the 9.5k-cycle membership phase.
    if (prod && !spare) stage_slice(1);                     // (N > 104: no spare wave -- first thing, as in round 5)
    // ---- agent states and the delay line's taps are requested together (the lower eight waves); thread i owns agent i (the
    //      expression tree of integrate_one: bit-exact given the action)
    // [r6] Behind a SCALAR branch on the wave index: the waits hipcc puts in front of the first use of these loads are then
    // executed by the loading waves only (with the DMA on the upper eight waves and this section executed by all sixteen, the first
    // barrier stood at 4.9k cycles -- the slice's arrival -- instead of the states' ~2k: tools/harness/flock_phase_prof.hip stamp 9).
    if (wave_u < FA_LOAD_WAVES) {
    double px = 0.0, py = 0.0, vx = 0.0, vy = 0.0, cx = 0.0, cy = 0.0;
    if (tid < N) {
        const double2 pa = *reinterpret_cast<const double2*>(xb + tid * 4), pb = *reinterpret_cast<const double2*>(xb + tid * 4 + 2);
        px = pa.x; py = pa.y; vx = pb.x; vy = pb.y;
        // reference point of the fp32 membership test: agent 0's position BEFORE the step (any point is valid; this one needs no
        // exchange between threads: rollout.hip "Exact membership from an fp32 test")
        const double2 pc = *reinterpret_cast<const double2*>(xb);
        cx = pc.x; cy = pc.y;
    }
    const long tap = 6L * N, per = (long)K * tap;
    constexpr int XT_ = 64 * FA_LOAD_WAVES;                 // threads that copy the delay line
    constexpr int XP = FA_DELAY_ELEMS / XT_;                // elements per thread: (K - 1) 6 N <= 4 * 6 * 128 (checked by the dispatch)
    float xpv[XP];
#pragma unroll
    for (int r = 0; r < XP; ++r) {
        const long e = tap + tid + (long)r * XT_;
        xpv[r] = (K > 1 && o.has_prev && tid < XT_ && e < per) ? o.Xp[(long)b * per + e - tap] : 0.f;
    }
    if (tid < N) {
        integrate_one(px, py, vx, vy, u + (size_t)b * N * 2 + (size_t)tid * su_agent, su_axis, tid < p.n_leaders, p);
        xob[tid * 4 + 0] = px; xob[tid * 4 + 1] = py; xob[tid * 4 + 2] = vx; xob[tid * 4 + 3] = vy;
        spx[tid] = px; spy[tid] = py; svx[tid] = vx; svy[tid] = vy;
        sxy[tid] = make_float2((float)(px - cx), (float)(py - cy));
    }
    if (K > 1) {                                            // delay line: taps 1..K-1 <- previous taps 0..K-2
#pragma unroll
        for (int r = 0; r < XP; ++r) {
            const long e = tap + tid + (long)r * XT_;
            if (tid < XT_ && e < per) o.Xn[(long)b * per + e] = xpv[r];
        }
    }
    }
    __syncthreads();
    FL_STAMP(9);
    if (prod && spare) stage_slice(1);                      // waves 13, 14: under the membership phase of waves 0 .. 12
    FL_STAMP(10);
    // ---- membership + features: eight lanes per row, lane `piece` owns candidates j0 .. j0 + jh - 1 (the row-tiled kernel's
    //      pieces).  Meanwhile the last wave forms the episode sums in the reduction tree of the row-tiled kernel (256 threads
    //      there: agent i in lane i % 64 of wave i / 64, butterfly inside the wave, wave sums added in order).
    const int pi = tid >> 3, piece = tid & 7;
    const int jh = (N + 7) >> 3;
    const double R2 = p.comm_radius2;
    const float R2f = (float)R2, Rf = sqrtf(R2f);
    const bool need_cent = o.centralized && o.expert != nullptr;
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0;    // the row's feature sums (piece-0 lanes)
    if (wave == FA_WAVES - 1 && (o.reward != nullptr || need_cent)) {
        double sx[4], sy[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const int i = lane + 64 * w;
            sx[w] = mgp_wave_sum(i < N ? svx[i] : 0.0);
            sy[w] = mgp_wave_sum(i < N ? svy[i] : 0.0);
        }
        double tvx = 0.0, tvy = 0.0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { tvx += sx[w]; tvy += sy[w]; }
        if (lane == 0) { tot_sh[0] = tvx; tot_sh[1] = tvy; }
        if (o.reward != nullptr) {
            const double mx = tvx / (double)N, my = tvy / (double)N;
            double dsum = 0.0;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const int i = lane + 64 * w;
                double dv = 0.0;
                if (i < N) { const double ex = svx[i] - mx, ey = svy[i] - my; dv += ex * ex + ey * ey; }
                dsum += mgp_wave_sum(dv);
            }
            if (lane == 0) o.reward[b] = -1.0 * (dsum / (double)N) * p.reward_scale;
        }
    }
    if (pi < N) {
        const float2 si = sxy[pi];
        const int j0 = piece * jh, nd = max(0, min(jh, N - j0));
        const float M = fmaxf(fabsf(si.x), fabsf(si.y)) + 2.0f * Rf;
        const float band = Rf * (16.f * M + 16.f * Rf) * 5.9604645e-8f + R2f * 1.1920929e-7f;
        const float t_in = R2f - band, t_out = R2f + band;
        // sign bits instead of compare / select pairs: the sign of r2 - t_in says "clearly inside", the sign of t_out - r2 "clearly
        // outside"; v_alignbit shifts each into a mask (test k ends in bit 15 - k of a 16-test pass; rollout.hip ro_s1_masks)
        unsigned int im = 0u, om = 0u;
#pragma unroll
        for (int c0 = 0; c0 < 16; c0 += 8) {                // jh <= 16: two groups of eight, every read of a group in flight together
            float2 sj[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) sj[q] = sxy[min(j0 + c0 + q, N - 1)];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float dx = si.x - sj[q].x, dy = si.y - sj[q].y;
                const float r2 = fmaf(dy, dy, dx * dx);
                im = __builtin_amdgcn_alignbit(im, __float_as_uint(r2 - t_in), 31);
                om = __builtin_amdgcn_alignbit(om, __float_as_uint(t_out - r2), 31);
            }
        }
        unsigned int in_m = __builtin_bitreverse32(im) >> 16;                 // test k -> bit k
        unsigned int unc_m = ~(__builtin_bitreverse32(om) >> 16) & ~in_m & 0xFFFFu;
        unsigned int valid = (1u << nd) - 1u;               // (candidates beyond the piece re-tested row N - 1)
        const int self = pi - j0;
        if (self >= 0 && self < nd) valid &= ~(1u << self);
        in_m &= valid; unc_m &= valid;
        const double xi = spx[pi], yi = spy[pi], vxi = svx[pi], vyi = svy[pi];
        while (unc_m) {                                     // rare: the spec's own fp64 expression decides
            const int q = __builtin_ctz(unc_m);
            unc_m &= unc_m - 1u;
            const double dx = xi - spx[j0 + q], dy = yi - spy[j0 + q];
            if (dx * dx + dy * dy < R2) in_m |= 1u << q;
        }
        if (p.link_drop != 0u) {                            // FlockingStochastic-v0: faded links leave the mask
            unsigned int mq = in_m;
            const unsigned int wi = fade_word(xi, yi);
            while (mq) {
                const int q = __builtin_ctz(mq);
                mq &= mq - 1u;
                if (!link_up(p, pi, j0 + q, N, wi, fade_word(spx[j0 + q], spy[j0 + q]))) in_m &= ~(1u << q);
            }
        }
        // the piece's feature terms, ascending j, one true division per neighbour (the row-tiled kernel's expression)
        double f0 = 0, f1 = 0, f2 = 0, f3 = 0, f4 = 0, f5 = 0;
        {
            unsigned int mq = in_m;
            while (mq) {
                const int j = j0 + __builtin_ctz(mq);
                mq &= mq - 1u;
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
        // the row's 128-bit word, OR-combined over its eight lanes on the DPP path (every lane gets it)
        unsigned long long lo = 0ull, hi = 0ull;
        if (j0 < 64) {
            lo = (unsigned long long)in_m << j0;
            if (j0 > 32) hi = (unsigned long long)in_m >> (64 - j0);
        } else {
            hi = (unsigned long long)in_m << (j0 - 64);
        }
        unsigned int w0 = (unsigned int)lo, w1 = (unsigned int)(lo >> 32), w2 = (unsigned int)hi, w3 = (unsigned int)(hi >> 32);
#define FA_OR(c) w0 |= (unsigned int)__builtin_amdgcn_update_dpp(0, (int)w0, c, 0xF, 0xF, true); w1 |= (unsigned int)__builtin_amdgcn_update_dpp(0, (int)w1, c, 0xF, 0xF, true); \
                 w2 |= (unsigned int)__builtin_amdgcn_update_dpp(0, (int)w2, c, 0xF, 0xF, true); w3 |= (unsigned int)__builtin_amdgcn_update_dpp(0, (int)w3, c, 0xF, 0xF, true);
        FA_OR(0xB1) FA_OR(0x4E) FA_OR(0x141)
#undef FA_OR
        const unsigned long long flo = ((unsigned long long)w1 << 32) | w0, fhi = ((unsigned long long)w3 << 32) | w2;
        int pos;
        if (j0 < 64) pos = __popcll(flo & ((1ull << j0) - 1ull));
        else pos = __popcll(flo) + __popcll(fhi & ((1ull << (j0 - 64)) - 1ull));
        unsigned char* lp = rlist + pi * RS;
        {
            unsigned int mq = in_m;
            while (mq) { lp[pos++] = (unsigned char)(j0 + __builtin_ctz(mq)); mq &= mq - 1u; }
        }
        // pieces 1 .. 7 are added to piece 0's sums in ascending order (row_shl:s brings lane + s's value; the sums of the
        // other lanes are not used)
#define FA_DSHL(v, c) __builtin_bit_cast(double, ((unsigned long long)(unsigned int)__builtin_amdgcn_update_dpp(0, (int)(unsigned int)(__builtin_bit_cast(unsigned long long, v) >> 32), c, 0xF, 0xF, true) << 32) | \
                                                  (unsigned int)__builtin_amdgcn_update_dpp(0, (int)(unsigned int)__builtin_bit_cast(unsigned long long, v), c, 0xF, 0xF, true))
#define FA_CHAIN(a, f) a = f; a += FA_DSHL(f, 0x101); a += FA_DSHL(f, 0x102); a += FA_DSHL(f, 0x103); a += FA_DSHL(f, 0x104); \
                       a += FA_DSHL(f, 0x105); a += FA_DSHL(f, 0x106); a += FA_DSHL(f, 0x107);
        FA_CHAIN(a0, f0) FA_CHAIN(a1, f1) FA_CHAIN(a2, f2) FA_CHAIN(a3, f3) FA_CHAIN(a4, f4) FA_CHAIN(a5, f5)
#undef FA_CHAIN
#undef FA_DSHL
        if (piece == 0) {
            const int cnt = __popcll(flo) + __popcll(fhi);
            const double deg = (double)cnt;
            wrow[pi] = p.mean_pooling ? 1.0 / (deg == 0.0 ? 1.0 : deg) : 1.0;
            rcnt[pi] = cnt;
            rowmask[2 * pi] = flo; rowmask[2 * pi + 1] = fhi;
            if (o.feat != nullptr) {
                float* fb = o.feat + (size_t)b * o.sFb + pi;
                fb[0 * (size_t)N] = (float)a0; fb[1 * (size_t)N] = (float)a1; fb[2 * (size_t)N] = (float)a2;
                fb[3 * (size_t)N] = (float)a3; fb[4 * (size_t)N] = (float)a4; fb[5 * (size_t)N] = (float)a5;
            }
        }
    }
    if (prod) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of the source slice has landed
    __syncthreads();
    FL_STAMP(11);
    // expert action, a closed form of the observation (the centralised form needs the episode's velocity sums: behind the barrier)
    if (o.expert != nullptr && pi < N && piece == 0) {
        double tvx = a0, tvy = a3;
        if (o.centralized) {
            tvx = (double)N * svx[pi] - tot_sh[0];
            tvy = (double)N * svy[pi] - tot_sh[1];
        }
        const double ux = clipd(-tvx - (2.0 * a2 - 2.0 * a1), -p.ctrl_clip, p.ctrl_clip) * p.ctrl_gain;
        const double uy = clipd(-tvy - (2.0 * a5 - 2.0 * a4), -p.ctrl_clip, p.ctrl_clip) * p.ctrl_gain;
        o.expert[((size_t)b * N + pi) * 2 + 0] = (float)ux; o.expert[((size_t)b * N + pi) * 2 + 1] = (float)uy;
    }
    // ---- network rows (slice 1 of G_next) from the bit words, and the product rows G_next[j] = A_t . G_prev[j - 1], j >= 2:
    //      row i = w_i x (sum over i's ascending list of the source rows), one fused multiply-add per entry in list order (the
    //      arithmetic of gso_rows_half_kernel), source rows in LDS.  One flat sweep of float4 items (row, four columns) per slice.
    for (int j = 1; j < K; ++j) {
        float* Gj = Gn + (size_t)j * NN;
        for (int e = tid; e < nn4; e += FA_THREADS) {
            const int ri = e / n4, c4 = e - ri * n4;
            const float w = o.has_prev ? (float)wrow[ri] : 0.f;    // (an episode's first state: taps >= 1 read zero)
            if (j == 1) {
                const int c0 = c4 << 2;
                const unsigned long long wb = rowmask[2 * ri + (c0 >> 6)];
                const unsigned int nib = (unsigned int)(wb >> (c0 & 63)) & 0xFu;     // (c0 is a multiple of 4: the nibble does not straddle words)
                *reinterpret_cast<float4*>(Gj + (size_t)e * 4) = make_float4((nib & 1u) ? w : 0.f, (nib & 2u) ? w : 0.f, (nib & 4u) ? w : 0.f, (nib & 8u) ? w : 0.f);
                if (K > 2) {                                // the same item of the first product row right behind it
                    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (prod) fa_gather(acc, w, rcnt[ri], rlist + ri * RS, stage4, n4, c4);
                    *reinterpret_cast<float4*>(Gj + NN + (size_t)e * 4) = acc;
                }
            } else if (j > 2) {
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                if (prod) fa_gather(acc, w, rcnt[ri], rlist + ri * RS, stage4, n4, c4);
                *reinterpret_cast<float4*>(Gj + (size_t)e * 4) = acc;
            }
        }
        if (j == 1) FL_STAMP(12);
        if (prod && j >= 2 && j + 1 < K) {                  // the next source slice (K >= 4): product j + 1 reads G_prev[j]
            __syncthreads();                                // every gather of this slice done
            stage_slice(j);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }
    FL_STAMP(13);
}

int launch_advance_episode(const double* x, double* xo, const float* u, long su_agent, long su_axis, const FlockOut& o,
                           const MgpFlockParams* p, int B, int N, hipStream_t st)
{
    const int lds = fa_offsets(N, o.K).total;
    if (mgp_allow_dyn_lds(reinterpret_cast<const void*>(flock_advance_kernel), (size_t)lds) != hipSuccess) return MGP_ELAUNCH;
    hipLaunchKernelGGL(flock_advance_kernel, dim3(B), dim3(FA_THREADS), lds, st, x, xo, u, su_agent, su_axis, o, *p, N);
    return mgp_launch_status();
}

// Acceptance test of reset candidates (FLOCK-SPEC v1 section 3: an episode starts from the first draw whose minimum degree is
// >= min_degree and whose closest pair is >= min_dist_thresh apart -- at N = 100 about one draw in 140 passes, and the host's
// numpy test of one draw, a 100 x 100 fp64 distance matrix, took 130 us: 19 ms per episode reset, 8.4 of the 10.9 s of a whole
// training run).  One workgroup per candidate: min over rows of |{j != i : r2_ij < R^2}| and min over pairs of r2, with
// r2 = dx dx + dy dy evaluated exactly as numpy does (two rounded products, one rounded sum: this file is built with
// -ffp-contract=off), so the host's decision -- deg >= min_degree and sqrt(r2_min) >= thresh -- is the sequential sampler's.
constexpr int RC_THREADS = 256;
__global__ __launch_bounds__(RC_THREADS)
void reset_check_kernel(const double* __restrict__ pos, int N, double R2, int* __restrict__ min_degree, double* __restrict__ r2_min)
{
    extern __shared__ __attribute__((aligned(16))) double rcs[];   // px [N] | py [N]
    __shared__ int sdeg[RC_THREADS / 64];
    __shared__ double smin[RC_THREADS / 64];
    double* px = rcs;
    double* py = rcs + N;
    const double* pc = pos + (size_t)blockIdx.x * N * 2;
    for (int i = threadIdx.x; i < N; i += RC_THREADS) { px[i] = pc[2 * i]; py[i] = pc[2 * i + 1]; }
    __syncthreads();
    int dmin = 0x7FFFFFFF;
    double rmin = __builtin_huge_val();
    for (int i = threadIdx.x; i < N; i += RC_THREADS) {
        const double xi = px[i], yi = py[i];
        int deg = 0;
        for (int j = 0; j < N; ++j) {
            const double dx = xi - px[j], dy = yi - py[j];
            const double r2 = dx * dx + dy * dy;
            if (j != i) {
                deg += (r2 < R2) ? 1 : 0;
                rmin = fmin(rmin, r2);
            }
        }
        dmin = min(dmin, deg);
    }
    for (int o = 32; o > 0; o >>= 1) {
        dmin = min(dmin, __shfl_xor(dmin, o));
        rmin = fmin(rmin, __shfl_xor(rmin, o));
    }
    if ((threadIdx.x & 63) == 0) { sdeg[threadIdx.x >> 6] = dmin; smin[threadIdx.x >> 6] = rmin; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < RC_THREADS / 64; ++w) { dmin = min(dmin, sdeg[w]); rmin = fmin(rmin, smin[w]); }
        min_degree[blockIdx.x] = dmin;
        r2_min[blockIdx.x] = rmin;
    }
}

}  // namespace

extern "C" int mgp_flock_reset_check(const double* pos, int M, int N, double comm_radius2, int* min_degree, double* r2_min,
                                     void* stream)
{
    if (M < 0 || N < 2 || N > 8192) return MGP_EINVAL;
    if (M == 0) return MGP_OK;
    MGP_CHECK_PTR8(pos); MGP_CHECK_PTR(min_degree); MGP_CHECK_PTR8(r2_min);
    const size_t lds = (size_t)2 * N * sizeof(double);
    mgp_clear_error();
    if (mgp_allow_dyn_lds(reinterpret_cast<const void*>(reset_check_kernel), lds) != hipSuccess) return MGP_ELAUNCH;
    hipLaunchKernelGGL(reset_check_kernel, dim3((unsigned)M), dim3(RC_THREADS), lds, static_cast<hipStream_t>(stream), pos, N,
                       comm_radius2, min_degree, r2_min);
    return mgp_launch_status();
}

extern "C" int mgp_flock_step(double* x, double* x_out, const float* u, long su_agent, long su_axis,
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
    if (x_out != nullptr && (reinterpret_cast<uintptr_t>(x_out) & 7u)) return MGP_EALIGN;
    if (sAb < 0 || sFb < 0) return MGP_EINVAL;
    FlockOut o = {A, A64, feat, feat64, reward, expert, nullptr, p->centralized ? 1 : 0, 0, sAb ? sAb : (long)N * N,
                  sFb ? sFb : 6L * N,
                  0, 0, 0, nullptr, nullptr, nullptr, nullptr, 0};
    return launch_flock(x, x_out, u, su_agent, su_axis, o, p, B, N, static_cast<hipStream_t>(stream));
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
    FlockOut o = {nullptr, nullptr, nullptr, nullptr, nullptr, u, u64, centralized ? 1 : 0, 0, 0, 0,
                  0, 0, 0, nullptr, nullptr, nullptr, nullptr, 0};
    // no action => the state is only read
    return launch_flock(const_cast<double*>(x), nullptr, nullptr, 2, 1, o, p, B, N, static_cast<hipStream_t>(stream));
}

/* Simulator step + delayed-GSO / delay-line transition in ONE launch (device-resident rollouts).  Equivalent to
 * mgp_flock_step(x, x_out, u, ..., A = G_next + N*N, sAb = K*N*N, feat = Xd_next, sFb = K*6*N, ...) followed by
 * mgp_gso_advance(G_prev, G_next, Xd_prev, Xd_next, ...), bit-for-bit, without the second kernel's re-read and
 * re-compaction of the dense network rows. */
extern "C" int mgp_flock_step_advance(double* x, double* x_out, const float* u, long su_agent, long su_axis,
                                      const float* G_prev, float* G_next, const float* Xd_prev, float* Xd_next,
                                      double* reward, float* expert, const MgpFlockParams* p,
                                      int B, int K, int N, int has_prev, void* stream)
{
    if (B < 0 || N <= 0 || K <= 0) return MGP_EINVAL;
    int rc = check_params(p);
    if (rc != MGP_OK) return rc;
    if (B == 0) return MGP_OK;
    if (B > 65535) return MGP_EINVAL;
    // covered: ping-pong state, an action, N a multiple of 4 up to 128 (a row = <= 32 float4 lanes), K >= 2
    if (N > 128 || (N & 3) || K < 2 || u == nullptr || x_out == nullptr || x_out == x) return MGP_EUNSUPPORTED;
    MGP_CHECK_PTR8(x); MGP_CHECK_PTR8(x_out); MGP_CHECK_PTR(G_next); MGP_CHECK_PTR(Xd_next);
    if (!mgp_aligned16(G_next)) return MGP_EUNSUPPORTED;
    if (has_prev) {
        MGP_CHECK_PTR(G_prev); MGP_CHECK_PTR(Xd_prev);
        if (!mgp_aligned16(G_prev)) return MGP_EUNSUPPORTED;
        if (G_prev == G_next || Xd_prev == Xd_next) return MGP_EINVAL;
    }
    const long NN = (long)N * N;
    FlockOut o = {G_next + NN, nullptr, Xd_next, nullptr, reward, expert, nullptr, p->centralized ? 1 : 0,
                  reward != nullptr ? 1 : 0,
                  (long)K * NN, (long)K * 6 * N, 1, K, has_prev ? 1 : 0, G_prev, G_next, Xd_prev, Xd_next, 0};
    mgp_clear_error();
    // one workgroup per episode (flock_advance_kernel) unless MGP_FLOCK_ADVANCE_TILED asks for the row-tiled kernel of rounds 1-4
    static const bool tiled = getenv("MGP_FLOCK_ADVANCE_TILED") != nullptr && getenv("MGP_FLOCK_ADVANCE_TILED")[0] == '1';
    // flock_advance_kernel copies the delay line through a fixed register array: (K - 1) 6 N <= FA_DELAY_ELEMS (K <= 5 at N = 128);
    // longer delay lines run the row-tiled kernel, which loops over any K
    if (!tiled && fa_offsets(N, K).total <= 160 * 1024 && (long)(K - 1) * 6 * N <= FA_DELAY_ELEMS &&
        mgp_aligned16(x))                                                         // (the agent states are read as 16-byte pairs)
        return launch_advance_episode(x, x_out, u, su_agent, su_axis, o, p, B, N, static_cast<hipStream_t>(stream));
    return launch_step<true, FP_THREADS, FP_ROWS, FP_PIECES>(x, x_out, u, su_agent, su_axis, o, p, B, N,
                                                  static_cast<hipStream_t>(stream));
}

/* Words per membership bit row written by mgp_flock_step_sparse: 8 row pieces, each a whole number of 64-bit words. */
extern "C" int mgp_sparse_words(int N)
{
    if (N <= 0) return 0;
    return FP_PIECES * (((((N + FP_PIECES - 1) / FP_PIECES) + 63) & ~63) / 64);
}

extern "C" int mgp_flock_step_sparse(const double* x, double* x_out, const float* u, long su_agent, long su_axis,
                                     unsigned long long* bits, long sBb, float* wrow, long sWb, float* featT, long sTb,
                                     double* reward, float* expert, const MgpFlockParams* p, int B, int N, void* stream)
{
    if (B < 0 || N <= 0) return MGP_EINVAL;
    int rc = check_params(p);
    if (rc != MGP_OK) return rc;
    if (B == 0) return MGP_OK;
    if (B > 65535 || N > 4096) return MGP_EUNSUPPORTED;
    MGP_CHECK_PTR8(x); MGP_CHECK_PTR8(x_out); MGP_CHECK_PTR8(bits); MGP_CHECK_PTR(wrow); MGP_CHECK_PTR(featT);
    if (x_out == x) return MGP_EINVAL;                     // ping-pong state only
    if (!mgp_aligned16(featT) || (sTb & 3)) return MGP_EALIGN;
    FlockOut o = {};
    o.reward = reward; o.expert = expert; o.centralized = p->centralized;
    o.bits = bits; o.wq = wrow; o.featT = featT; o.sBb = sBb; o.sWb = sWb; o.sTb = sTb;
    o.sep_reward = reward != nullptr;
    mgp_clear_error();
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (u != nullptr)
        return launch_step<true, FP_THREADS, FP_ROWS, FP_PIECES>(x, x_out, u, su_agent, su_axis, o, p, B, N, st);
    // no action: observations of x itself (x_out is not written)
    return launch_step<false, FP_THREADS, FP_ROWS, FP_PIECES>(x, const_cast<double*>(x), nullptr, su_agent, su_axis, o, p, B, N, st);
}
