// Simulator step for large flocks on the factored state: FLOCK-SPEC sections 1-5, the arithmetic of
// flock.hip, with a cell list instead of the all-pairs sweep.  At N = 1000 a row has ~10 radius neighbours among 999
// candidates; flock.hip's kernel tests all of them (64 M ordered pair tests per step for 64 episodes, 55 us).  Here every
// workgroup bins the episode's agents into square cells no smaller than the communication radius (counting sort in LDS,
// cell members in ascending index order so that every sum has a fixed order), and a row only visits the 3 x 3 cells around
// its own: the membership test r2 < R^2 is still the spec's own fp64 expression on every candidate, so the bit rows are
// the all-pairs kernel's (and the oracle's) bit rows exactly; only the order of the fp64 feature sums differs (1e-16
// relative; the tests allow 1e-11).  The cell width is chosen per episode and step as extent / g with g = min(64,
// floor(extent / (R (1 + 1e-9)))) >= 1, i.e. never below R: a spread-out flock gets 64 x 64 cells, a collapsed one
// degenerates gracefully to all pairs.
// Outputs are those of mgp_flock_step_sparse: bit rows, row weights, (N, 8) feature rows, reward, expert action.
// Built with -ffp-contract=off (fp64 spec arithmetic).
#include <math.h>
#include "mgp_common.h"
#include "mgp_device.h"

namespace {

constexpr int SS_THREADS = 256;            // = rows per workgroup
constexpr int SS_WAVES = SS_THREADS / 64;
constexpr int SS_G = 64;                   // cells per axis, at most
constexpr int SS_MAXN = 2048;              // LDS plan: 32 B of state + 4 B of lists per agent, NW + 1 words of bits per row

template <typename T, typename OP>
__device__ __forceinline__ T ss_block_reduce(T v, T* sh /* [SS_WAVES] */, OP op)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = op(v, __shfl_xor(v, off, MGP_WAVE));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    T t = sh[0];
#pragma unroll
    for (int w = 1; w < SS_WAVES; ++w) t = op(t, sh[w]);
    return t;
}

struct SsOut {
    unsigned long long* bits; float* wq; float* featT; long sBb, sWb, sTb;
    double* reward; float* expert;
};

// grid: x = tile of 256 rows, y = b.
// LDS: px, py, vx, vy [N] f64 | start [G*G + 1] int | cid [N] u16 | sorted [N] u16 | rowbits [256][NW + 1] u64
template <bool FD>
__global__ __launch_bounds__(SS_THREADS)
void sp_sim_kernel(const double* __restrict__ x, double* __restrict__ xo, const float* __restrict__ u, long su_agent,
                   long su_axis, SsOut o, MgpFlockParams p, int N, int NW)
{
    extern __shared__ __attribute__((aligned(16))) double ssm[];
    __shared__ double shd[SS_WAVES];
    __shared__ int shi[SS_WAVES];
    double* spx = ssm; double* spy = ssm + N; double* svx = ssm + 2 * (size_t)N; double* svy = ssm + 3 * (size_t)N;
    int* start = reinterpret_cast<int*>(ssm + 4 * (size_t)N);                   // [G*G + 1] cell -> first entry of `sorted`
    unsigned short* cid = reinterpret_cast<unsigned short*>(start + SS_G * SS_G + 1 + 1);
    unsigned short* sorted = cid + ((N + 3) & ~3);
    unsigned long long* rowbits = reinterpret_cast<unsigned long long*>(
        (reinterpret_cast<uintptr_t>(sorted + ((N + 3) & ~3)) + 7) & ~(uintptr_t)7);
    const int tid = threadIdx.x, b = blockIdx.y;
    const int i0 = blockIdx.x * SS_THREADS;
    const double* xb = x + (size_t)b * N * 4;
    double* xob = xo + (size_t)b * N * 4;

    // ---- every agent of the episode -> LDS, integrated (spec section 1) when an action is given; own rows -> x_out
    double sum_vx = 0.0, sum_vy = 0.0, mnx = 1e300, mxx = -1e300, mny = 1e300, mxy = -1e300;
    for (int i = tid; i < N; i += SS_THREADS) {
        double px = xb[i * 4 + 0], py = xb[i * 4 + 1], vx = xb[i * 4 + 2], vy = xb[i * 4 + 3];
        if (u != nullptr) {
            integrate_one(px, py, vx, vy, u + (size_t)b * N * 2 + (size_t)i * su_agent, su_axis, i < p.n_leaders, p);
            if (i >= i0 && i < i0 + SS_THREADS) {
                xob[i * 4 + 0] = px; xob[i * 4 + 1] = py; xob[i * 4 + 2] = vx; xob[i * 4 + 3] = vy;
            }
        }
        spx[i] = px; spy[i] = py; svx[i] = vx; svy[i] = vy;
        sum_vx += vx; sum_vy += vy;
        mnx = fmin(mnx, px); mxx = fmax(mxx, px); mny = fmin(mny, py); mxy = fmax(mxy, py);
    }
    for (int c = tid; c <= SS_G * SS_G; c += SS_THREADS) start[c] = 0;
    auto add = [](double a, double c) { return a + c; };
    auto mn = [](double a, double c) { return fmin(a, c); };
    auto mx = [](double a, double c) { return fmax(a, c); };
    const double tot_vx = ss_block_reduce(sum_vx, shd, add);
    const double tot_vy = ss_block_reduce(sum_vy, shd, add);
    mnx = ss_block_reduce(mnx, shd, mn); mxx = ss_block_reduce(mxx, shd, mx);
    mny = ss_block_reduce(mny, shd, mn); mxy = ss_block_reduce(mxy, shd, mx);
    if (o.reward != nullptr && blockIdx.x == 0) {           // spec section 4: population variance, two passes
        const double mvx = tot_vx / (double)N, mvy = tot_vy / (double)N;
        double dv = 0.0;
        for (int i = tid; i < N; i += SS_THREADS) {
            const double ex = svx[i] - mvx, ey = svy[i] - mvy;
            dv += ex * ex + ey * ey;
        }
        const double var = ss_block_reduce(dv, shd, add) / (double)N;
        if (tid == 0) o.reward[b] = -1.0 * var * p.reward_scale;
    }
    // ---- cell grid: width >= R (1 + 1e-9) on each axis, at most 64 x 64 cells
    const double R = sqrt(p.comm_radius2) * (1.0 + 1e-9);
    const double ex_ = mxx - mnx, ey_ = mxy - mny;
    const int gx = max(1, (int)fmin((double)SS_G, floor(ex_ / R)));
    const int gy = max(1, (int)fmin((double)SS_G, floor(ey_ / R)));
    const double iwx = (ex_ > 0.0) ? (double)gx / ex_ : 0.0, iwy = (ey_ > 0.0) ? (double)gy / ey_ : 0.0;
    // NaN positions (a diverged episode) land in cell 0: every index stays valid, the outputs are garbage as they would be
    for (int i = tid; i < N; i += SS_THREADS) {
        int cx = (int)((spx[i] - mnx) * iwx), cy = (int)((spy[i] - mny) * iwy);
        cx = min(max(cx, 0), gx - 1); cy = min(max(cy, 0), gy - 1);
        const int c = cy * gx + cx;
        cid[i] = (unsigned short)c;
        atomicAdd(&start[c + 1], 1);                        // histogram, shifted by one: the scan below turns it into starts
    }
    __syncthreads();
    // ---- inclusive scan of the counts (16 cells per thread, then across threads)
    const int ncell = gx * gy;
    {
        int loc[16], run = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) { const int c = tid * 16 + q + 1; loc[q] = (c <= ncell) ? start[c] : 0; run += loc[q]; }
        // exclusive prefix of `run` over the 256 threads
        int inc = run;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(inc, off, MGP_WAVE); if ((tid & 63) >= off) inc += t; }
        if ((tid & 63) == 63) shi[tid >> 6] = inc;
        __syncthreads();
        int base = 0;
        for (int w = 0; w < (tid >> 6); ++w) base += shi[w];
        int acc = base + inc - run;
#pragma unroll
        for (int q = 0; q < 16; ++q) { const int c = tid * 16 + q + 1; acc += loc[q]; if (c <= ncell) start[c] = acc; }
    }
    __syncthreads();
    // ---- scatter by cell (arrival order), then every cell's members into ascending index order (fixed summation order)
    {
        int* cursor = reinterpret_cast<int*>(rowbits);       // [ncell] running fill of each cell (rowbits is not live yet)
        for (int c = tid; c < ncell; c += SS_THREADS) cursor[c] = start[c];
        __syncthreads();
        for (int i = tid; i < N; i += SS_THREADS) sorted[atomicAdd(&cursor[cid[i]], 1)] = (unsigned short)i;
        __syncthreads();
        for (int c = tid; c < ncell; c += SS_THREADS) {
            const int s0 = start[c], s1 = start[c + 1];
            for (int a = s0 + 1; a < s1; ++a) {               // insertion sort: cells hold a handful of agents
                const unsigned short v = sorted[a];
                int k = a - 1;
                while (k >= s0 && sorted[k] > v) { sorted[k + 1] = sorted[k]; --k; }
                sorted[k + 1] = v;
            }
        }
        __syncthreads();
    }
    // ---- this thread's row: the 3 x 3 cells around its own
    const int i = i0 + tid;
    const int RSW = NW + 1;                                   // odd word stride per row
    unsigned long long* myrow = rowbits + (size_t)tid * RSW;
    for (int wd = 0; wd < NW; ++wd) myrow[wd] = 0ull;
    if (i < N) {
        const double xi = spx[i], yi = spy[i], vxi = svx[i], vyi = svy[i];
        const double R2 = p.comm_radius2;
        const unsigned int wi = (FD && p.link_drop != 0u) ? fade_word(xi, yi) : 0u;
        const int c = cid[i], cy = c / gx, cx = c - cy * gx;
        double deg = 0.0, f0 = 0, f1 = 0, f2 = 0, f3 = 0, f4 = 0, f5 = 0;
        for (int dy = -1; dy <= 1; ++dy) {
            const int yy = cy + dy;
            if (yy < 0 || yy >= gy) continue;
            const int xa = max(cx - 1, 0), xz = min(cx + 1, gx - 1);
            // the (up to) three cells of a grid row are contiguous in `sorted`
            const int s0 = start[yy * gx + xa], s1 = start[yy * gx + xz + 1];
            for (int a = s0; a < s1; ++a) {
                const int j = sorted[a];
                const double dx = xi - spx[j], dyy = yi - spy[j];
                const double r2 = dx * dx + dyy * dyy;
                if (j == i || !(r2 < R2)) continue;
                if (FD && p.link_drop != 0u && !link_up(p, i, j, N, wi, fade_word(spx[j], spy[j]))) continue;
                myrow[j >> 6] |= 1ull << (j & 63);
                const double q = 1.0 / r2;
                const double qq = q * q;
                deg += 1.0;
                f0 += vxi - svx[j];
                f1 += dx * qq;
                f2 += dx * q;
                f3 += vyi - svy[j];
                f4 += dyy * qq;
                f5 += dyy * q;
            }
        }
        const double w = p.mean_pooling ? 1.0 / (deg == 0.0 ? 1.0 : deg) : 1.0;
        o.wq[(size_t)b * o.sWb + i] = (float)w;
        float* ft = o.featT + (size_t)b * o.sTb + (size_t)i * 8;
        *reinterpret_cast<float4*>(ft) = make_float4((float)f0, (float)f1, (float)f2, (float)f3);
        *reinterpret_cast<float4*>(ft + 4) = make_float4((float)f4, (float)f5, 0.f, 0.f);
        if (o.expert != nullptr) {                            // spec section 5
            double tvx = f0, tvy = f3;
            if (p.centralized) { tvx = (double)N * vxi - tot_vx; tvy = (double)N * vyi - tot_vy; }
            const double ux = clipd(-tvx - (2.0 * f2 - 2.0 * f1), -p.ctrl_clip, p.ctrl_clip) * p.ctrl_gain;
            const double uy = clipd(-tvy - (2.0 * f5 - 2.0 * f4), -p.ctrl_clip, p.ctrl_clip) * p.ctrl_gain;
            o.expert[((size_t)b * N + i) * 2 + 0] = (float)ux;
            o.expert[((size_t)b * N + i) * 2 + 1] = (float)uy;
        }
    }
    __syncthreads();
    // ---- bit rows of this tile -> HBM, coalesced
    {
        const int rows = min(SS_THREADS, N - i0);
        unsigned long long* gb = o.bits + (size_t)b * o.sBb + (size_t)i0 * NW;
        for (int idx = tid; idx < rows * NW; idx += SS_THREADS) {
            const int r = idx / NW, wd = idx - r * NW;
            gb[idx] = rowbits[(size_t)r * RSW + wd];
        }
    }
}

}  // namespace

/* mgp_flock_step_sparse with a cell list (see the top of this file).  Same arguments and outputs; N <= 2048. */
extern "C" int mgp_flock_step_cells(const double* x, double* x_out, const float* u, long su_agent, long su_axis,
                                    unsigned long long* bits, long sBb, float* wrow, long sWb, float* featT, long sTb,
                                    double* reward, float* expert, const MgpFlockParams* p, int B, int N, void* stream)
{
    if (B < 0 || N <= 0 || p == nullptr) return MGP_EINVAL;
    if (!(p->comm_radius2 > 0.0) || !(p->dt > 0.0) || p->n_leaders < 0) return MGP_EINVAL;
    if (B == 0) return MGP_OK;
    if (B > 65535 || N > SS_MAXN) return MGP_EUNSUPPORTED;
    MGP_CHECK_PTR8(x); MGP_CHECK_PTR8(x_out); MGP_CHECK_PTR8(bits); MGP_CHECK_PTR(wrow); MGP_CHECK_PTR(featT);
    if (x_out == x) return MGP_EINVAL;
    if (!mgp_aligned16(featT) || (sTb & 3)) return MGP_EALIGN;
    const int NW = mgp_sparse_words(N);
    const size_t lds = (size_t)4 * N * 8 + (size_t)(SS_G * SS_G + 2) * 4 + (size_t)2 * ((N + 3) & ~3) * 2 + 8
                       + (size_t)SS_THREADS * (NW + 1) * 8;
    SsOut o = {bits, wrow, featT, sBb, sWb, sTb, reward, expert};
    mgp_clear_error();
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid(mgp_ceil_div(N, SS_THREADS), B);
    const bool fade = p->link_drop != 0u;
    const void* fn = fade ? reinterpret_cast<const void*>(sp_sim_kernel<true>) : reinterpret_cast<const void*>(sp_sim_kernel<false>);
    if (lds > 48 * 1024 && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return MGP_ELAUNCH;
    if (fade)
        hipLaunchKernelGGL(sp_sim_kernel<true>, grid, dim3(SS_THREADS), lds, st, x, x_out, u, su_agent, su_axis, o, *p, N, NW);
    else
        hipLaunchKernelGGL(sp_sim_kernel<false>, grid, dim3(SS_THREADS), lds, st, x, x_out, u, su_agent, su_axis, o, *p, N, NW);
    return mgp_launch_status();
}
