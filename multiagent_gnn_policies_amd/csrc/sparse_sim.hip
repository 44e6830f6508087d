// Simulator step for large flocks on the factored state: FLOCK-SPEC sections 1-5, the arithmetic of
// flock.hip, with a cell list instead of the all-pairs sweep.  At N = 1000 a row has ~10 radius neighbours among 999
// candidates; flock.hip's kernel tests all of them (64 M ordered pair tests per step for 64 episodes, 55 us).  Here every
// workgroup bins the episode's agents into square cells no smaller than the communication radius (counting sort in LDS,
// cell members in ascending index order so that every sum has a fixed order), and a row only visits the 3 x 3 cells around
// its own: the membership test r2 < R^2 is still the spec's own fp64 expression on every candidate, so the bit rows are
// the all-pairs kernel's (and the oracle's) bit rows exactly; only the order of the fp64 feature sums differs (1e-16
// relative; the tests allow 1e-11).  The cell width is chosen per episode and step as extent / g with g = min(32,
// floor(extent / (R (1 + 1e-9)))) >= 1, i.e. never below R: a spread-out flock gets 32 x 32 cells (about one agent per cell
// at the N <= 2048 this kernel covers), a collapsed one degenerates gracefully to all pairs.
//
// [r3] One workgroup = 1024 threads = 256 rows x FOUR lanes per row (round 2: 256 threads, one per row, one wave per SIMD:
// 53k cycles per launch at 64 x 1000 -- 8k loading the episode in four dependent round trips, 5.6k in six block reductions
// one after the other, 7k in a per-cell insertion sort, 19-21k in the row search, 5k copying the bit rows out with an integer
// division per word).  Now: every request of the load phase is issued before the first use; ONE block reduction of six
// values; members ranked inside their cell in parallel (rank = members with a smaller index) instead of sorted serially; the
// row search splits a row's candidates over its four lanes (partial fp64 sums added in a fixed order by DPP); bit words
// are OR-ed into the LDS row by LDS atomics and leave as 32 contiguous bytes per lane.
// Outputs are those of mgp_flock_step_sparse: bit rows, row weights, (N, 8) feature rows, reward, expert action.
// Built with -ffp-contract=off (fp64 spec arithmetic).
#include <math.h>
#include "mgp_common.h"
#include "mgp_device.h"
#include "rollout_common.h"
#include "sparse_common.h"

namespace {

#ifdef MGP_SP_PROFILE
__device__ unsigned long long mgp_ss_stamps[16 * 16];     // [wave][stamp]
#define SS_STAMP(i) do { if (blockIdx.x == 1 && blockIdx.y == 3 && (threadIdx.x & 63) == 0) mgp_ss_stamps[(threadIdx.x >> 6) * 16 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define SS_STAMP(i) do { } while (0)
#endif

struct SsOut {
    unsigned long long* bits; float* wq; float* featT; long sBb, sWb, sTb;
    double* reward; float* expert;
    unsigned short* nbr; long sNb;          // optional compact neighbour rows (see mgp_flock_step_cells_nbr), batch stride in u16
    int nbl_on;                             // LDS room for assembling them
};

// grid: x = tile of 256 rows, y = b.
// LDS: px, py, vx, vy [N] f64 | start [G*G + 2] int | cursor [G*G] int | cid, tmp, sorted [N4] u16 | rowbits [256][NW + 1] u64 | sub [1024][subcap] u16
//      | PRE: posf [N] float2 (fp32 positions in `sorted` order)
template <bool FD, bool PRE>
__global__ __launch_bounds__(SS_THREADS)
void sp_sim_kernel(const double* __restrict__ x, double* __restrict__ xo, const float* __restrict__ u, long su_agent,
                   long su_axis, SsOut o, MgpFlockParams p, int N, int NW, int subcap)
{
    extern __shared__ __attribute__((aligned(16))) double ssm[];
    __shared__ double red[SS_WAVES][2];
    __shared__ float4 redf[SS_WAVES];
    __shared__ double red2[SS_WAVES];
    __shared__ int shi[SS_WAVES];
    const int N4 = (N + 3) & ~3;
    double* spx = ssm; double* spy = ssm + N; double* svx = ssm + 2 * (size_t)N; double* svy = ssm + 3 * (size_t)N;
    int* start = reinterpret_cast<int*>(ssm + 4 * (size_t)N);                   // [G*G + 2] cell -> first entry of `sorted`
    int* cursor = start + SS_G * SS_G + 2;                                      // [G*G] running fill of each cell
    unsigned short* cid = reinterpret_cast<unsigned short*>(cursor + SS_G * SS_G);
    unsigned short* tmp = cid + N4;                                             // cell members in arrival order
    unsigned short* sorted = tmp + N4;                                          // ... in ascending index order
    unsigned long long* rowbits = reinterpret_cast<unsigned long long*>(
        (reinterpret_cast<uintptr_t>(sorted + N4) + 7) & ~(uintptr_t)7);
    unsigned short* sub = reinterpret_cast<unsigned short*>(rowbits + (size_t)SS_ROWS * (NW + 1));   // [1024][subcap] hits of a lane
    float2* posf = reinterpret_cast<float2*>((reinterpret_cast<uintptr_t>(sub + (size_t)SS_THREADS * subcap) + 7) & ~(uintptr_t)7);
    unsigned short* nbl = reinterpret_cast<unsigned short*>(posf + (PRE ? N : 0));                    // [256][16] list rows being assembled
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.y;
    const int i0 = blockIdx.x * SS_ROWS;
    const int RSW = NW + 1;                                   // odd word stride per row
    const double* xb = x + (size_t)b * N * 4;
    double* xob = xo + (size_t)b * N * 4;

    SS_STAMP(0);
    // ---- every agent of the episode -> LDS, integrated (spec section 1) when an action is given; own rows -> x_out.
    // Two agents per thread at most (N <= 2048); every request is issued before the first use.
    double2 a01[2], a23[2];
    float aux[2], auy[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int i = tid + q * SS_THREADS;
        const bool in = i < N;
        a01[q] = in ? *reinterpret_cast<const double2*>(xb + (size_t)i * 4) : make_double2(0.0, 0.0);
        a23[q] = in ? *reinterpret_cast<const double2*>(xb + (size_t)i * 4 + 2) : make_double2(0.0, 0.0);
        aux[q] = auy[q] = 0.f;
        if (u != nullptr && in && i >= p.n_leaders) {
            const float* ub = u + (size_t)b * N * 2 + (size_t)i * su_agent;
            aux[q] = ub[0]; auy[q] = ub[su_axis];
        }
    }
    for (int c = tid; c < SS_G * SS_G + 2; c += SS_THREADS) start[c] = 0;
    for (int e = tid; e < SS_ROWS * RSW; e += SS_THREADS) rowbits[e] = 0ull;
    // bounding box in fp32 (minima negated: one kind of reduction for all four), widened below: the cell grid is private to
    // this workgroup and only has to CONTAIN every agent -- the membership test stays the spec's fp64 expression
    double sum_vx = 0.0, sum_vy = 0.0;
    float bnx = -3e38f, bxx = -3e38f, bny = -3e38f, bxy = -3e38f;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int i = tid + q * SS_THREADS;
        if (i < N) {
            double px = a01[q].x, py = a01[q].y, vx = a23[q].x, vy = a23[q].y;
            if (u != nullptr) {                               // spec section 1 (integrate_one, the action already in registers)
                const double ux = clipd((double)aux[q], -p.max_accel, p.max_accel) * p.action_gain;
                const double uy = clipd((double)auy[q], -p.max_accel, p.max_accel) * p.action_gain;
                px = (px + vx * p.dt) + ((ux * p.dt) * p.dt) * 0.5;
                py = (py + vy * p.dt) + ((uy * p.dt) * p.dt) * 0.5;
                vx = vx + ux * p.dt;
                vy = vy + uy * p.dt;
                if (i >= i0 && i < i0 + SS_ROWS) {
                    *reinterpret_cast<double2*>(xob + (size_t)i * 4) = make_double2(px, py);
                    *reinterpret_cast<double2*>(xob + (size_t)i * 4 + 2) = make_double2(vx, vy);
                }
            }
            spx[i] = px; spy[i] = py; svx[i] = vx; svy[i] = vy;
            sum_vx += vx; sum_vy += vy;
            bnx = fmaxf(bnx, -(float)px); bxx = fmaxf(bxx, (float)px); bny = fmaxf(bny, -(float)py); bxy = fmaxf(bxy, (float)py);
        }
    }
    SS_STAMP(1);
    // ---- one block reduction of the six values (fixed order: lanes by DPP, then the sixteen wave partials over a DPP row)
    {
        const double s0 = wave_sum_d(sum_vx), s1 = wave_sum_d(sum_vy);
        const float m0 = wave_max_to_last(bnx), m1 = wave_max_to_last(bxx), m2 = wave_max_to_last(bny), m3 = wave_max_to_last(bxy);
        if (lane == 63) { red[wave][0] = s0; red[wave][1] = s1; redf[wave] = make_float4(m0, m1, m2, m3); }
    }
    __syncthreads();
    double tot_vx, tot_vy, mnx, mxx, mny, mxy;
    {
        const int wl = lane & 15;
        double t0 = red[wl][0], t1 = red[wl][1];
        float4 m = redf[wl];
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
        // widened by more than the fp32 rounding of the conversions: every fp64 position lies inside
        mnx = -((double)nx + 2.4e-7 * fabs((double)nx) + 1e-30); mxx = (double)xx + 2.4e-7 * fabs((double)xx) + 1e-30;
        mny = -((double)ny + 2.4e-7 * fabs((double)ny) + 1e-30); mxy = (double)xy + 2.4e-7 * fabs((double)xy) + 1e-30;
    }
    SS_STAMP(2);
    const bool does_reward = o.reward != nullptr && blockIdx.x == 0;
    if (does_reward) {                                        // spec section 4: population variance, two passes
        const double mvx = tot_vx / (double)N, mvy = tot_vy / (double)N;
        double dv = 0.0;
#pragma unroll
        for (int q = 0; q < 2; ++q)
            if (tid + q * SS_THREADS < N) {
                const double ex = svx[tid + q * SS_THREADS] - mvx, ey = svy[tid + q * SS_THREADS] - mvy;
                dv += ex * ex + ey * ey;
            }
        dv = wave_sum_d(dv);
        if (lane == 0) red2[wave] = dv;                       // summed by thread 0 after the next barrier
    }
    SS_STAMP(3);
    // ---- cell grid: width >= R (1 + 1e-9) on each axis, at most 32 x 32 cells
    const double R = sqrt(p.comm_radius2) * (1.0 + 1e-9);
    const double ex_ = mxx - mnx, ey_ = mxy - mny;
    const int gx = max(1, (int)fmin((double)SS_G, floor(ex_ / R)));
    const int gy = max(1, (int)fmin((double)SS_G, floor(ey_ / R)));
    const double iwx = (ex_ > 0.0) ? (double)gx / ex_ : 0.0, iwy = (ey_ > 0.0) ? (double)gy / ey_ : 0.0;
    // NaN positions (a diverged episode) land in cell 0: every index stays valid, the outputs are garbage as they would be
    int myc[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int i = tid + q * SS_THREADS;
        myc[q] = 0;
        if (i < N) {
            int cx = (int)((spx[i] - mnx) * iwx), cy = (int)((spy[i] - mny) * iwy);
            cx = min(max(cx, 0), gx - 1); cy = min(max(cy, 0), gy - 1);
            const int c = cy * gx + cx;
            myc[q] = c;
            cid[i] = (unsigned short)c;
            atomicAdd(&start[c + 1], 1);                    // histogram, shifted by one: the scan below turns it into starts
        }
    }
    __syncthreads();
    SS_STAMP(4);
    if (does_reward && tid == 0) {
        double var = 0.0;
#pragma unroll
        for (int w = 0; w < SS_WAVES; ++w) var += red2[w];
        o.reward[b] = -1.0 * (var / (double)N) * p.reward_scale;
    }
    // ---- inclusive scan of the counts: thread t owns index t + 1 (= the count of cell t)
    {
        const int cnt = start[tid + 1];                        // (zero beyond the grid)
        const int inc = ss_wave_scan(cnt);
        if (lane == 63) shi[wave] = inc;
        __syncthreads();
        int base = (lane < wave) ? shi[lane & 15] : 0;          // waves before this one, summed over lanes 0..15
        base += (int)dpp_u<0xB1>((unsigned int)base); base += (int)dpp_u<0x4E>((unsigned int)base);
        base += (int)dpp_u<0x141>((unsigned int)base); base += (int)dpp_u<0x140>((unsigned int)base);
        base = __builtin_amdgcn_readfirstlane(base);
        start[tid + 1] = base + inc;                           // cells <= t hold this many agents
        cursor[tid] = base + inc - cnt;                        // first slot of cell t
    }
    __syncthreads();
    SS_STAMP(5);
    // ---- scatter by cell (arrival order), then every member to its rank inside the cell: the number of members with a
    // smaller index -- ascending index order (a fixed summation order) without a serial sort
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int i = tid + q * SS_THREADS;
        if (i < N) tmp[atomicAdd(&cursor[myc[q]], 1)] = (unsigned short)i;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int i = tid + q * SS_THREADS;
        if (i < N) {
            const int s0 = start[myc[q]], s1 = start[myc[q] + 1];
            int rank = 0;
            for (int a = s0; a < s1; ++a) rank += (tmp[a] < i) ? 1 : 0;
            sorted[s0 + rank] = (unsigned short)i;
            if (PRE) posf[s0 + rank] = make_float2((float)spx[i], (float)spy[i]);
        }
    }
    __syncthreads();
    SS_STAMP(6);
    // ---- row r = tid >> 2 on four lanes: the candidates of the 3 x 3 cells around its own, every fourth one per lane
    const int r = tid >> 2, part = tid & 3;
    const int i = i0 + r;
    if (i < N) {
        const double xi = spx[i], yi = spy[i], vxi = svx[i], vyi = svy[i];
        const double R2 = p.comm_radius2;
        const unsigned int wi = (FD && p.link_drop != 0u) ? fade_word(xi, yi) : 0u;
        const int c = cid[i], cy = c / gx, cx = c - cy * gx;
        unsigned long long* myrow = rowbits + (size_t)r * RSW;
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
        // Pass 1: the membership tests only; a lane notes its hits in its own short list.  (With the feature terms inside
        // this loop a wave executes them on EVERY candidate -- some lane always has a hit -- ~10 times per row; the hits
        // themselves are ~2 per lane.)
        SS_STAMP(11);
        unsigned short* mine = sub + (size_t)tid * subcap;
        int cnt = 0;
        // PRE: the candidates' positions in fp32, in candidate order (contiguous reads, no dependent index load -- this loop
        // was bound by the LDS pipe: three scattered reads per candidate and lane).  fp32 decides wherever it provably
        // agrees with the spec's fp64 test -- |r2_f32 - R^2| beyond `band`, four times the worst-case rounding of the fp32
        // evaluation for candidates of the 3 x 3 cells -- and the fp64 expression decides inside the band.
        const float xif = (float)xi, yif = (float)yi, R2f = (float)R2;
        const float cwf = (float)fmax(ex_ / (double)gx, ey_ / (double)gy);
        const float Rf = (float)R;
        const float band = 2.384185791015625e-07f * 4.f * Rf * (2.f * fmaxf(fabsf(xif), fabsf(yif)) + 3.f * cwf + Rf) + 1e-30f;
        for (int dy = -1; dy <= 1; ++dy) {
            const int yy = cy + dy;
            if (yy < 0 || yy >= gy) continue;
            // the (up to) three cells of a grid row are contiguous in `sorted`
            const int s0 = start[yy * gx + xa], s1 = start[yy * gx + xz + 1];
            for (int a = s0 + part; a < s1; a += 4) {
                const int j = sorted[a];
                if (PRE) {
                    const float2 pj = posf[a];
                    const float dxf = xif - pj.x, dyf = yif - pj.y;
                    const float r2f = dxf * dxf + dyf * dyf;
                    if (j == i || !(r2f < R2f + band)) continue;
                    if (r2f > R2f - band) {                       // too close to call in fp32
                        const double dx = xi - spx[j], dyy = yi - spy[j];
                        if (!(dx * dx + dyy * dyy < R2)) continue;
                    }
                } else {
                    const double dx = xi - spx[j], dyy = yi - spy[j];
                    const double r2 = dx * dx + dyy * dyy;
                    if (j == i || !(r2 < R2)) continue;
                }
                if (FD && p.link_drop != 0u && !link_up(p, i, j, N, wi, fade_word(spx[j], spy[j]))) continue;
                atomicOr(&myrow[j >> 6], 1ull << (j & 63));
                if (cnt < subcap) mine[cnt] = (unsigned short)j;
                ++cnt;
            }
        }
        // the row's bit words are final once its four lanes -- one wave -- have issued their LDS atomics: lane `part` writes its
        // quarter (NW / 4 contiguous words) to HBM now, under pass 2, without a workgroup barrier
        {
            __builtin_amdgcn_s_waitcnt(0xC07F);                   // lgkmcnt(0): the wave's LDS atomics have completed
            __builtin_amdgcn_wave_barrier();
            const int wpl = NW >> 2;                              // NW is a multiple of 8
            unsigned long long* gb = o.bits + (size_t)b * o.sBb + (size_t)i * NW + part * wpl;
            const unsigned long long* lr = rowbits + (size_t)r * RSW + part * wpl;
            for (int k = 0; k < wpl; k += 2)
                *reinterpret_cast<ulonglong2*>(gb + k) = make_ulonglong2(lr[k], lr[k + 1]);
        }
        SS_STAMP(10);
        // Pass 2: the row's hits -- the four lists one after the other -- dealt round-robin to the four lanes
        const int c0 = (int)dpp_u<0x00>((unsigned int)cnt), c1 = (int)dpp_u<0x55>((unsigned int)cnt);
        const int c2 = (int)dpp_u<0xAA>((unsigned int)cnt), c3 = (int)dpp_u<0xFF>((unsigned int)cnt);
        if (o.nbr != nullptr) {
            // the row as a compact LIST for the gather launches of the policy step (32 bytes to stage instead of the 128-byte
            // bit row at N = 1000, and evenly dealt entries): entry e of the concatenated lists at u16 position
            // (e & 3) * 4 + (e >> 2) -- lane q of a gathering quad reads entries q, q + 4, q + 8, q + 12 as one 8-byte word --
            // and the count at position 15; 0xFFFF there: more than 15 neighbours (or no list memory), use the bit row
            const int tot = c0 + c1 + c2 + c3;
            const bool fits = tot <= 15 && max(max(c0, c1), max(c2, c3)) <= subcap;
            if (!o.nbl_on) {                                      // no LDS left for the assembly (N near 2048): bit rows only
                if (part == 3) o.nbr[(size_t)b * o.sNb + (size_t)i * 16 + 15] = (unsigned short)0xFFFF;
            } else {
            unsigned short* lrow = nbl + (size_t)r * 16;
            *reinterpret_cast<uint2*>(lrow + 4 * part) = make_uint2(0u, part == 3 ? ((fits ? (unsigned int)tot : 0xFFFFu) << 16) : 0u);
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_wave_barrier();
            if (fits) {
                const int off = part == 0 ? 0 : (part == 1 ? c0 : (part == 2 ? c0 + c1 : c0 + c1 + c2));
                for (int k = 0; k < cnt; ++k) { const int e = off + k; lrow[(e & 3) * 4 + (e >> 2)] = mine[k]; }
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_wave_barrier();
            *reinterpret_cast<uint2*>(o.nbr + (size_t)b * o.sNb + (size_t)i * 16 + 4 * part) = *reinterpret_cast<const uint2*>(lrow + 4 * part);
            }
        }
        if (max(max(c0, c1), max(c2, c3)) <= subcap) {
            const int tot = c0 + c1 + c2 + c3;
            const unsigned short* rowsub = sub + (size_t)(tid & ~3) * subcap;
            for (int e = part; e < tot; e += 4) {
                int k = e, sl = 0;
                if (k >= c0) { k -= c0; sl = 1; if (k >= c1) { k -= c1; sl = 2; if (k >= c2) { k -= c2; sl = 3; } } }
                const int j = rowsub[sl * subcap + k];
                const double dx = xi - spx[j], dyy = yi - spy[j];
                terms(j, dx, dyy, dx * dx + dyy * dyy);
            }
        } else {
            // a list overflowed (a dense flock, or no room for lists at this N): every lane walks its candidates again
            for (int dy = -1; dy <= 1; ++dy) {
                const int yy = cy + dy;
                if (yy < 0 || yy >= gy) continue;
                const int s0 = start[yy * gx + xa], s1 = start[yy * gx + xz + 1];
                for (int a = s0 + part; a < s1; a += 4) {
                    const int j = sorted[a];
                    const double dx = xi - spx[j], dyy = yi - spy[j];
                    const double r2 = dx * dx + dyy * dyy;
                    if (j == i || !(r2 < R2)) continue;
                    if (FD && p.link_drop != 0u && !link_up(p, i, j, N, wi, fade_word(spx[j], spy[j]))) continue;
                    terms(j, dx, dyy, r2);
                }
            }
        }
        // the four partial sums of the row, fixed order (lanes 0+1, 2+3, then the pairs)
        deg += (int)dpp_u<0xB1>((unsigned int)deg); deg += (int)dpp_u<0x4E>((unsigned int)deg);
        f0 += dpp_d<0xB1>(f0); f0 += dpp_d<0x4E>(f0); f1 += dpp_d<0xB1>(f1); f1 += dpp_d<0x4E>(f1);
        f2 += dpp_d<0xB1>(f2); f2 += dpp_d<0x4E>(f2); f3 += dpp_d<0xB1>(f3); f3 += dpp_d<0x4E>(f3);
        f4 += dpp_d<0xB1>(f4); f4 += dpp_d<0x4E>(f4); f5 += dpp_d<0xB1>(f5); f5 += dpp_d<0x4E>(f5);
        SS_STAMP(7);
        if (part == 0) {
            const double dg = (double)deg;
            const double w = p.mean_pooling ? 1.0 / (dg == 0.0 ? 1.0 : dg) : 1.0;
            o.wq[(size_t)b * o.sWb + i] = (float)w;
            float* ft = o.featT + (size_t)b * o.sTb + (size_t)i * 8;
            *reinterpret_cast<float4*>(ft) = make_float4((float)f0, (float)f1, (float)f2, (float)f3);
            *reinterpret_cast<float4*>(ft + 4) = make_float4((float)f4, (float)f5, 0.f, 0.f);
            if (o.expert != nullptr) {                        // spec section 5
                double tvx = f0, tvy = f3;
                if (p.centralized) { tvx = (double)N * vxi - tot_vx; tvy = (double)N * vyi - tot_vy; }
                const double ux = clipd(-tvx - (2.0 * f2 - 2.0 * f1), -p.ctrl_clip, p.ctrl_clip) * p.ctrl_gain;
                const double uy = clipd(-tvy - (2.0 * f5 - 2.0 * f4), -p.ctrl_clip, p.ctrl_clip) * p.ctrl_gain;
                *reinterpret_cast<float2*>(o.expert + ((size_t)b * N + i) * 2) = make_float2((float)ux, (float)uy);
            }
        }
    }
    SS_STAMP(8);
    SS_STAMP(9);
}

}  // namespace

/* mgp_flock_step_cells that also writes every row as a compact neighbour list (nbr: (N,16) u16 per episode, batch stride sNb):
 * up to 15 neighbour indices -- entry e at position (e & 3) * 4 + (e >> 2) -- and the count at position 15 (0xFFFF: use the bit
 * row).  The order of a row's entries is a deterministic function of the state.  nbr may be NULL. */
extern "C" int mgp_flock_step_cells_nbr(const double* x, double* x_out, const float* u, long su_agent, long su_axis,
                                        unsigned long long* bits, long sBb, float* wrow, long sWb, float* featT, long sTb,
                                        unsigned short* nbr, long sNb, double* reward, float* expert, const MgpFlockParams* p,
                                        int B, int N, void* stream)
{
    if (B < 0 || N <= 0 || p == nullptr) return MGP_EINVAL;
    if (!(p->comm_radius2 > 0.0) || !(p->dt > 0.0) || p->n_leaders < 0) return MGP_EINVAL;
    if (B == 0) return MGP_OK;
    if (B > 65535 || N > SS_MAXN) return MGP_EUNSUPPORTED;
    MGP_CHECK_PTR8(x); MGP_CHECK_PTR8(x_out); MGP_CHECK_PTR8(bits); MGP_CHECK_PTR(wrow); MGP_CHECK_PTR(featT);
    if (x_out == x) return MGP_EINVAL;
    if (!mgp_aligned16(featT) || (sTb & 3)) return MGP_EALIGN;
    if (!mgp_aligned16(x) || !mgp_aligned16(x_out) || !mgp_aligned16(bits) || (sBb & 1)) return MGP_EALIGN;
    if (expert != nullptr && (reinterpret_cast<uintptr_t>(expert) & 7u)) return MGP_EALIGN;
    const int NW = mgp_sparse_words(N);
    const int N4 = (N + 3) & ~3;
    const size_t lds0 = (size_t)4 * N * 8 + (size_t)(SS_G * SS_G + 2 + SS_G * SS_G) * 4 + (size_t)3 * N4 * 2 + 8
                        + (size_t)SS_ROWS * (NW + 1) * 8;
    // per-lane hit lists of the row search: SS_SUBCAP = 8 entries where the LDS has room (a row's list holds 15), else none
    // (single-pass fallback); fp32 positions for the pre-filter of the membership tests where they fit besides.
    // [r6] 8, not 16, wherever the lists exist: which rows fall back -- to the second walk for their feature sums, to the bit
    // row in the gathers -- depends on this number, and the persistent form (sparse_persist.hip: 8, its LDS is full) must
    // make the same choice row for row to stay bit-identical.
    const size_t cap_lds = 160 * 1024;
    const int subcap = (lds0 + (size_t)SS_THREADS * SS_SUBCAP * 2 <= cap_lds) ? SS_SUBCAP : 0;
    const size_t lds1a = lds0 + (size_t)SS_THREADS * subcap * 2 + 8;
    const bool nbl_on = nbr != nullptr && subcap > 0 && lds1a + (size_t)SS_ROWS * 16 * 2 <= cap_lds;
    const size_t lds1 = lds1a + (nbl_on ? (size_t)SS_ROWS * 16 * 2 : 0);
    const bool pre = lds1 + (size_t)N * 8 <= cap_lds;
    const size_t lds = lds1 + (pre ? (size_t)N * 8 : 0);
    if (nbr != nullptr && ((reinterpret_cast<uintptr_t>(nbr) & 7u) || (sNb & 3))) return MGP_EALIGN;
    SsOut o = {bits, wrow, featT, sBb, sWb, sTb, reward, expert, nbr, sNb, nbl_on ? 1 : 0};
    mgp_clear_error();
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid(mgp_ceil_div(N, SS_ROWS), B);
    const bool fade = p->link_drop != 0u;
    auto go = [&](auto kern) -> int {
        if (mgp_allow_dyn_lds(reinterpret_cast<const void*>(kern), lds) != hipSuccess) return MGP_ELAUNCH;
        hipLaunchKernelGGL(kern, grid, dim3(SS_THREADS), lds, st, x, x_out, u, su_agent, su_axis, o, *p, N, NW, subcap);
        return MGP_OK;
    };
    int rc;
    if (fade) rc = pre ? go(sp_sim_kernel<true, true>) : go(sp_sim_kernel<true, false>);
    else rc = pre ? go(sp_sim_kernel<false, true>) : go(sp_sim_kernel<false, false>);
    if (rc != MGP_OK) return rc;
    return mgp_launch_status();
}

/* mgp_flock_step_sparse with a cell list (see the top of this file).  Same arguments and outputs; N <= 2048. */
extern "C" int mgp_flock_step_cells(const double* x, double* x_out, const float* u, long su_agent, long su_axis,
                                    unsigned long long* bits, long sBb, float* wrow, long sWb, float* featT, long sTb,
                                    double* reward, float* expert, const MgpFlockParams* p, int B, int N, void* stream)
{
    return mgp_flock_step_cells_nbr(x, x_out, u, su_agent, su_axis, bits, sBb, wrow, sWb, featT, sTb, nullptr, 0, reward, expert,
                                    p, B, N, stream);
}
