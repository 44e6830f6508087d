// Fused Actor forward / backward for ind_agg == 0 (reference learner/actor.py:45-86; the only configuration
// train.py reaches: gnn_dagger.py:43, gnn_cloning.py:41).
//
// Forward, ONE launch, one workgroup (512 threads = 8 waves) per (episode b, tile of <=128 agent columns):
//   phase 1  aggregation  Y[(f,k), n] = sum_m X[b,k,f,m] * G[b,k,m,n]       (HBM-bound: G is read exactly once)
//            G[b,k] rows are the contraction index, so a workgroup that owns whole rows streams the operator as
//            a flat float4 array (N <= 128) / 512-byte row segments (N > 128).  A thread owns 4 adjacent columns
//            and CT channel accumulators and walks rows with stride R = 512/colgroups; the X[b,k] tile sits
//            TRANSPOSED in LDS ([m][c]) so a row's CT multipliers are one or two wide, mostly-broadcast ds_reads.
//            The loads of tap k+1 are issued before the LDS combine of tap k, so HBM latency hides behind it.
//            Row phases are combined through LDS in fixed order (deterministic, no atomics).
//   phase 2  filter GEMM (H x F*K)(F*K x n) and the hidden layers on fp32 MFMA (v_mfma_f32_16x16x4_f32: exact
//            fp32 k-ordered fmaf chain -- keeps the 1e-5 budget; gfx950 has no xf32/TF32), bias + tanh on the
//            accumulator registers, activations ping-pong in LDS.  Weights live in LDS, zero padded to 16-row
//            m-tiles / 4-column k-steps, so no tail code in the MFMA loop.
//   Y and Z never touch HBM unless `saved` is requested (training).
// Backward: one workgroup per 64-column tile walks the layers in LDS and emits per-tile parameter-gradient
// partials; a second kernel adds the partials in tile order (deterministic).
#include "mgp_common.h"

namespace {

constexpr int AF_THREADS = 512;
constexpr int AF_WAVES = AF_THREADS / 64;
constexpr int AF_TILE = 128;              // max agent columns per workgroup
constexpr int AF_U = 20;                  // G rows in flight per thread (one batch covers N = 100)
constexpr int AF_MAXW = 64;               // max layer width covered by the fused kernel
constexpr int AF_LDS_LIMIT = 150 * 1024;

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Optional in-kernel phase timestamps (scratch/af_prof.hip defines MGP_AF_PROFILE; never in the product build).
#ifdef MGP_AF_PROFILE
__device__ unsigned long long mgp_af_stamps[64];
#define AF_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) mgp_af_stamps[i] = __builtin_readcyclecounter(); } while (0)
#else
#define AF_STAMP(i) do { } while (0)
#endif

struct ActorParams {
    const float* W[MGP_MAX_LAYERS];
    const float* b[MGP_MAX_LAYERS];
    int dims[MGP_MAX_LAYERS + 1];         // F, h1, ..., nA
    int woff[MGP_MAX_LAYERS];             // LDS offset (floats) of layer l's padded weight block
    int n_layers;
};

__host__ __device__ inline int pad4(int x) { return (x + 3) & ~3; }
__host__ __device__ inline int pad16(int x) { return (x + 15) & ~15; }
// LDS row stride of a padded weight block with `cin` input channels (odd => spread over banks)
__host__ __device__ inline int wstride(int cin) { return pad4(cin) + 1; }

template <int V> struct GLoad;
template <> struct GLoad<4> {
    static __device__ __forceinline__ void ld(const float* p, float (&g)[4]) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        g[0] = t.x; g[1] = t.y; g[2] = t.z; g[3] = t.w;
    }
};
template <> struct GLoad<1> {
    static __device__ __forceinline__ void ld(const float* p, float (&g)[1]) { g[0] = *p; }
};

// LDS carve-up (floats).  `red` (aggregation partials) and the activation ping-pong buffers alias: the MLP
// phase starts only after the combine.
struct Carve {
    int xs;            // X tile, all taps: [K][MC][CT]
    int ys;            // aggregated features  [pad16(F*K)][ncp]
    int w;             // padded weights (+ bias in the spare column of each row)
    int un;            // union: red [R][F*K][twp]  |  act0,act1 [pad16(maxw)][ncp] each
    int act_stride;    // floats per activation buffer
    int wtot;          // floats in the padded weight image
    int total;
};

// Thread -> (tap k, row phase r, column group cg).  Each thread walks rows r, r+R, ... of G[b,k] for its V
// columns with up to AF_U rows in flight; for N = 100, K = 3 (R = 6, 17 rows per thread) the whole 120 KB
// operator of the episode is requested in ONE batch before anything else happens, and the weight / X staging,
// the FMAs and the combine all run in the shadow of that single HBM round trip.  Three barriers in phase 1.
template <int CT, int V>
__global__ __launch_bounds__(AF_THREADS)
void actor_fwd_kernel(const float* __restrict__ X, const float* __restrict__ G, float* __restrict__ out,
                      float* __restrict__ saved, ActorParams P, Carve cv,
                      int B, int K, int N, int tw, int ntiles, int R, int MC, int ncp)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = blockIdx.x % ntiles, b = blockIdx.x / ntiles;
    const int F = P.dims[0];
    const int n0 = tile * tw;
    const int cols = min(tw, N - n0);
    const int cgt = (cols + V - 1) / V;
    const int twp = cgt * V;
    const int FK = F * K;
    const int per_k = R * cgt;
    const int kk = tid / per_k;
    const int rem = tid - kk * per_k;
    const int r = rem / cgt, cg = rem - r * cgt;
    const bool active = kk < K;

    float* xs = smem + cv.xs;
    float* ys = smem + cv.ys;
    float* wl = smem + cv.w;
    float* red = smem + cv.un;
    AF_STAMP(0);

    const size_t NN = (size_t)N * N;
    const float* Gk = G + ((size_t)b * K + (active ? kk : 0)) * NN + n0 + cg * V;
    const float* Xb = X + (size_t)b * K * F * N;

    float acc[CT][V];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int v = 0; v < V; ++v) acc[c][v] = 0.f;

    for (int m0 = 0; m0 < N; m0 += MC) {
        const int mc = min(MC, N - m0);
        // ---- (a) X tile of this row chunk, all taps: loads first (registers), stores after the G batch is issued
        constexpr int XU = 4;
        const int nx = FK * mc;                              // elements (q = k*F + c, mm) of the chunk
        float xv[XU];
        int xdst[XU];
#pragma unroll
        for (int j = 0; j < XU; ++j) {
            const int e = tid + AF_THREADS * j;
            const int ec = min(e, nx - 1);
            const int q = ec / mc, mm = ec - q * mc;
            const int k = q / F, c = q - k * F;
            xv[j] = Xb[(size_t)q * N + m0 + mm];
            xdst[j] = (e < nx) ? ((k * MC + mm) * CT + c) : -1;
        }
        // ---- (b) first batch of G rows: straight-line clamped loads (no branches => exact vmcnt bookkeeping)
        float g[AF_U][V];
        int mm = r;
#pragma unroll
        for (int u = 0; u < AF_U; ++u) {
            const int row = min(mm + u * R, mc - 1);
            GLoad<V>::ld(Gk + (size_t)(m0 + row) * N, g[u]);
        }
        // ---- (c) LDS staging while the G batch is in flight
        if (m0 > 0) __syncthreads();                         // readers of the previous chunk's xs are done
#pragma unroll
        for (int j = 0; j < XU; ++j)
            if (xdst[j] >= 0) xs[xdst[j]] = xv[j];
        for (int e = tid + AF_THREADS * XU; e < nx; e += AF_THREADS) {   // only for very large tiles
            const int q = e / mc, mx = e - q * mc;
            const int k = q / F, c = q - k * F;
            xs[(k * MC + mx) * CT + c] = Xb[(size_t)q * N + m0 + mx];
        }
        if (m0 == 0) {
            if (CT > F) {                                     // padded channels: zero once
                for (int i = tid; i < K * MC * (CT - F); i += AF_THREADS) {
                    const int row = i / (CT - F), c = F + i - row * (CT - F);
                    xs[row * CT + c] = 0.f;
                }
            }
            // padded weight image: rows up to pad16(cout), columns up to pad4(cin), bias in column ws-1
            for (int l = 0; l < P.n_layers; ++l) {
                const int cin = (l == 0) ? FK : P.dims[l];
                const int cout = P.dims[l + 1];
                const int ws = wstride(cin), tot = pad16(cout) * ws;
                float* dst = wl + P.woff[l];
                const float* src = P.W[l];
                const float* bsrc = P.b[l];
                constexpr int WU = 5;
                for (int base = 0; base < tot; base += AF_THREADS * WU) {
                    float wv[WU];
#pragma unroll
                    for (int j = 0; j < WU; ++j) {
                        const int e = base + tid + AF_THREADS * j;
                        const int o = e / ws, c = e - o * ws;
                        float v = 0.f;
                        if (e < tot && o < cout) {
                            if (c < cin) v = src[(size_t)o * cin + c];
                            else if (c == ws - 1) v = bsrc[o];
                        }
                        wv[j] = v;
                    }
#pragma unroll
                    for (int j = 0; j < WU; ++j) {
                        const int e = base + tid + AF_THREADS * j;
                        if (e < tot) dst[e] = wv[j];
                    }
                }
            }
            for (int i = tid; i < (pad16(FK) - FK) * ncp; i += AF_THREADS) ys[FK * ncp + i] = 0.f;
        }
        __syncthreads();
        AF_STAMP(1);
        // ---- (d) consume: acc[c][v] += x[k, m, c] * G[k, m, n]
        const float* xk = xs + (size_t)(active ? kk : 0) * MC * CT;
        while (true) {
#pragma unroll
            for (int u = 0; u < AF_U; ++u) {
                const int row = mm + u * R;
                const float keep = (active && row < mc) ? 1.f : 0.f;
                const float* xr = xk + min(row, mc - 1) * CT;
#pragma unroll
                for (int c = 0; c < CT; ++c) {
                    const float x = xr[c] * keep;
#pragma unroll
                    for (int v = 0; v < V; ++v) acc[c][v] = fmaf(x, g[u][v], acc[c][v]);
                }
            }
            mm += R * AF_U;
            if (mm >= mc) break;                              // uniform per (k, r) group, not per workgroup: no barrier inside
#pragma unroll
            for (int u = 0; u < AF_U; ++u) {
                const int row = min(mm + u * R, mc - 1);
                GLoad<V>::ld(Gk + (size_t)(m0 + row) * N, g[u]);
            }
        }
    }
    AF_STAMP(2);
    // ---- combine the R row phases: red[r][c*K + k][col], fixed order r = 0..R-1 (deterministic)
    if (R == 1) {
        if (active) {
#pragma unroll
            for (int c = 0; c < CT; ++c)
                if (c < F) {
#pragma unroll
                    for (int v = 0; v < V; ++v) ys[(c * K + kk) * ncp + cg * V + v] = acc[c][v];
                }
        }
    } else {
        if (active) {
#pragma unroll
            for (int c = 0; c < CT; ++c)
                if (c < F) {
#pragma unroll
                    for (int v = 0; v < V; ++v) red[((size_t)r * FK + c * K + kk) * twp + cg * V + v] = acc[c][v];
                }
        }
        __syncthreads();
        AF_STAMP(3);
        for (int i = tid; i < FK * cgt; i += AF_THREADS) {
            const int q = i / cgt, cgi = i - q * cgt;
            float s[V];
#pragma unroll
            for (int v = 0; v < V; ++v) s[v] = 0.f;
            for (int rr = 0; rr < R; ++rr) {
                const float* p = red + ((size_t)rr * FK + q) * twp + cgi * V;
#pragma unroll
                for (int v = 0; v < V; ++v) s[v] += p[v];
            }
#pragma unroll
            for (int v = 0; v < V; ++v) ys[q * ncp + cgi * V + v] = s[v];
        }
    }
    __syncthreads();
    AF_STAMP(4);
    // columns beyond `cols` inside the last 16-wide n-tile must be finite for the MFMA: zero them
    const int ncols16 = pad16(cols);
    for (int i = tid; i < FK * (ncols16 - cols); i += AF_THREADS) {
        const int row = i / (ncols16 - cols), col = cols + i % (ncols16 - cols);
        ys[row * ncp + col] = 0.f;
    }
    if (saved != nullptr) {
        float* sy = saved + (size_t)b * FK * N;
        for (int i = tid; i < FK * cols; i += AF_THREADS) {
            const int row = i / cols, col = i - row * cols;
            sy[(size_t)row * N + n0 + col] = ys[row * ncp + col];
        }
    }
    __syncthreads();
    AF_STAMP(5);

    // ---- phase 2: per-agent MLP on fp32 MFMA --------------------------------------------------------------
    const int NT = ncols16 / 16;
    const float* actin = ys;
    float* act0 = smem + cv.un;
    float* act1 = act0 + cv.act_stride;
    size_t soff = (size_t)B * FK * N;                   // running offset into `saved`
    const int li = lane & 15, lq = lane >> 4;
    for (int l = 0; l < P.n_layers; ++l) {
        const int cin = (l == 0) ? FK : P.dims[l];
        const int cout = P.dims[l + 1];
        const int ksteps = pad4(cin) / 4;
        const int ws = wstride(cin);
        const float* wb = wl + P.woff[l];
        const bool last = (l == P.n_layers - 1);
        float* actout = (l & 1) ? act1 : act0;
        const int MT = pad16(cout) / 16;
        // tile pairs (same m-tile, two n-tiles) share the A fragment and give two independent accumulators
        const int NTP = (NT + 1) / 2;
        for (int t = wave; t < MT * NTP; t += AF_WAVES) {
            const int mt = t / NTP, ntp = t - mt * NTP;
            const int nt0 = ntp * 2, nt1 = nt0 + 1;
            const bool two = nt1 < NT;
            // accumulators start at the bias of their rows (row = mt*16 + lq*4 + rr); padded rows hold 0
            f32x4 acc0;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) acc0[rr] = wb[(mt * 16 + lq * 4 + rr) * ws + ws - 1];
            f32x4 acc1 = acc0;
            const float* ap = wb + (mt * 16 + li) * ws + lq;
            const float* bp0 = actin + lq * ncp + nt0 * 16 + li;
            const float* bp1 = actin + lq * ncp + (two ? nt1 : nt0) * 16 + li;
            // fragments of up to 16 k-steps (cin <= 64) are fetched before the MFMA chain starts
            constexpr int KS = AF_MAXW / 4;
            float fa[KS], fb0[KS], fb1[KS];
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int sc = min(s, ksteps - 1);
                fa[s] = ap[4 * sc];
                fb0[s] = bp0[(size_t)4 * sc * ncp];
                fb1[s] = bp1[(size_t)4 * sc * ncp];
            }
            if (t == 0) AF_STAMP(10 + 4 * l);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                if (s < ksteps) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[s], fb0[s], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[s], fb1[s], acc1, 0, 0, 0);
                }
            }
            if (t == 0) { asm volatile("" :: "v"(acc0[0]), "v"(acc1[0])); AF_STAMP(11 + 4 * l); }
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                if (half == 1 && !two) break;
                const f32x4 acc = half ? acc1 : acc0;
                const int col = (half ? nt1 : nt0) * 16 + li;
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int row = mt * 16 + lq * 4 + rr;
                    float v = acc[rr];                    // rows >= cout: zero weights and zero bias -> 0
                    if (!last) v = tanhf(v);
                    if (last) {
                        if (row < cout && col < cols) out[((size_t)b * cout + row) * N + n0 + col] = v;
                    } else {
                        actout[row * ncp + col] = v;
                        if (saved != nullptr && row < cout && col < cols)
                            saved[soff + ((size_t)b * cout + row) * N + n0 + col] = v;
                    }
                }
            }
        }
        AF_STAMP(12 + 4 * l);
        soff += (size_t)B * cout * N;
        actin = actout;
        __syncthreads();
        AF_STAMP(6 + l);
    }
}

// ---------------------------------------------------------------------------------------------------------
// Backward (parameters only).  Workgroup per (b, 64-column tile).  LDS: delta ping-pong [maxw][65],
// input tile [maxin][65].  Partials: part[tile][P] with P = sum_l cout*cin + cout, layer-major (W then b).
constexpr int AB_THREADS = 256;
constexpr int AB_COLS = 64;

struct BwdParams {
    const float* W[MGP_MAX_LAYERS];
    int dims[MGP_MAX_LAYERS + 1];
    long poff[MGP_MAX_LAYERS];            // offset of layer l's (dW, db) block inside one partial
    long soff[MGP_MAX_LAYERS];            // offset (floats) of layer l's INPUT inside `saved` (per whole batch)
    int n_layers;
};

__global__ __launch_bounds__(AB_THREADS)
void actor_bwd_kernel(const float* __restrict__ dOut, const float* __restrict__ saved, float* __restrict__ part,
                      BwdParams P, long Ptot, int K, int N, int maxw, int maxin)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* d0 = smem;                                   // [maxw][65]
    float* d1 = d0 + (size_t)maxw * 65;                 // [maxw][65]
    float* ins = d1 + (size_t)maxw * 65;                // [maxin][65]
    const int tid = threadIdx.x;
    const int n0 = blockIdx.x * AB_COLS, b = blockIdx.y;
    const int ntx = gridDim.x;
    float* my = part + ((size_t)b * ntx + blockIdx.x) * Ptot;
    const int L = P.n_layers;
    const int nA = P.dims[L];

    float* dcur = d0;
    float* dnext = d1;
    for (int i = tid; i < nA * AB_COLS; i += AB_THREADS) {
        const int o = i >> 6, cl = i & 63;
        dcur[o * 65 + cl] = (n0 + cl < N) ? dOut[((size_t)b * nA + o) * N + n0 + cl] : 0.f;
    }
    for (int l = L - 1; l >= 0; --l) {
        const int cin = (l == 0) ? P.dims[0] * K : P.dims[l];
        const int cout = P.dims[l + 1];
        const float* inb = saved + P.soff[l] + (size_t)b * cin * N;
        __syncthreads();                                // dcur complete; previous users of ins/dnext done
        for (int i = tid; i < cin * AB_COLS; i += AB_THREADS) {
            const int c = i >> 6, cl = i & 63;
            ins[c * 65 + cl] = (n0 + cl < N) ? inb[(size_t)c * N + n0 + cl] : 0.f;
        }
        __syncthreads();
        float* myl = my + P.poff[l];
        // db
        for (int o = tid; o < cout; o += AB_THREADS) {
            float s = 0.f;
            for (int cl = 0; cl < AB_COLS; ++cl) s += dcur[o * 65 + cl];
            myl[(size_t)cout * cin + o] = s;
        }
        // dW[o][c] = sum_cols delta[o][col] * in[c][col]
        for (int p = tid; p < cout * cin; p += AB_THREADS) {
            const int o = p / cin, c = p - o * cin;
            float s = 0.f;
#pragma unroll 8
            for (int cl = 0; cl < AB_COLS; ++cl) s = fmaf(dcur[o * 65 + cl], ins[c * 65 + cl], s);
            myl[p] = s;
        }
        // delta_{l-1}[c][col] = (sum_o W[o][c] delta[o][col]) * (1 - in[c][col]^2)     (inputs of l>=1 are tanh outputs)
        if (l > 0) {
            const float* Wl = P.W[l];
            const int col = tid & 63, cgp = tid >> 6;
            for (int c = cgp; c < cin; c += AB_THREADS / 64) {
                float s = 0.f;
                for (int o = 0; o < cout; ++o) s = fmaf(Wl[(size_t)o * cin + c], dcur[o * 65 + col], s);
                const float z = ins[c * 65 + col];
                dnext[c * 65 + col] = s * (1.f - z * z);
            }
            float* t = dcur; dcur = dnext; dnext = t;
        }
    }
}

// scatter the reduced flat block into the caller's dW[l] / db[l] buffers
struct ScatterParams {
    float* dW[MGP_MAX_LAYERS];
    float* db[MGP_MAX_LAYERS];
    long poff[MGP_MAX_LAYERS];
    long wsz[MGP_MAX_LAYERS];
    int bsz[MGP_MAX_LAYERS];
    int n_layers;
};
__global__ __launch_bounds__(256)
void actor_bwd_scatter_kernel(const float* __restrict__ part, ScatterParams S, long Ptot, long ntiles)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= Ptot) return;
    float s = 0.f;
    for (long t = 0; t < ntiles; ++t) s += part[t * Ptot + i];
    int l = 0;
    while (l + 1 < S.n_layers && i >= S.poff[l + 1]) ++l;
    const long j = i - S.poff[l];
    if (j < S.wsz[l]) S.dW[l][j] = s; else S.db[l][j - S.wsz[l]] = s;
}

// ------------------------------------------------------------------------------------------ host helpers
struct Plan {
    int V, CT, tw, ntiles, MC, ncp, R;
    Carve cv;
    int woff[MGP_MAX_LAYERS];
};

bool make_plan(const int* dims, int n_layers, int K, int N, bool vec_ok, Plan* pl)
{
    if (n_layers <= 0 || n_layers > MGP_MAX_LAYERS || K <= 0 || N <= 0) return false;
    const int F = dims[0];
    if (F <= 0 || F > 8) return false;
    for (int i = 1; i <= n_layers; ++i) if (dims[i] <= 0 || dims[i] > AF_MAXW) return false;
    const int FK = F * K;
    if (FK > AF_MAXW) return false;
    pl->V = (vec_ok && N % 4 == 0) ? 4 : 1;
    pl->CT = F <= 4 ? 4 : (F <= 6 ? 6 : 8);
    pl->tw = N <= AF_TILE ? N : AF_TILE;
    pl->ntiles = (N + pl->tw - 1) / pl->tw;
    const int cgt = (pl->tw + pl->V - 1) / pl->V;
    if (K * cgt > AF_THREADS) return false;
    pl->R = AF_THREADS / (K * cgt);
    const int twp = cgt * pl->V;
    // rows of X staged per chunk, all taps at once: K * MC * CT floats <= 12288 (48 KB)
    pl->MC = 12288 / (K * pl->CT);
    if (pl->MC > N) pl->MC = N;
    if (pl->MC < 1) return false;
    // ncp = 16 mod 32 keeps the two k-rows an MFMA B-read touches per half-wave on disjoint banks
    int ncp = pad16(pl->tw);
    if (ncp % 32 == 0) ncp += 16;
    pl->ncp = ncp;
    int maxw = 0, wtot = 0;
    for (int l = 0; l < n_layers; ++l) {
        const int cin = (l == 0) ? FK : dims[l];
        pl->woff[l] = wtot;
        wtot += pad16(dims[l + 1]) * wstride(cin);
        if (l < n_layers - 1 && dims[l + 1] > maxw) maxw = dims[l + 1];
    }
    Carve& cv = pl->cv;
    int off = 0;
    cv.xs = off;  off += K * pl->MC * pl->CT; off = (off + 3) & ~3;
    cv.ys = off;  off += pad16(FK) * ncp;
    cv.w = off;   off += wtot; off = (off + 3) & ~3;
    cv.wtot = wtot;
    cv.un = off;
    cv.act_stride = pad16(maxw > 0 ? maxw : 16) * ncp;
    const int red = (pl->R > 1) ? pl->R * FK * twp : 0;
    const int act = 2 * cv.act_stride;
    off += red > act ? red : act;
    cv.total = off;
    return (size_t)off * sizeof(float) <= AF_LDS_LIMIT;
}

template <int CT, int V>
int launch_fwd(const float* X, const float* G, float* out, float* saved, const ActorParams& P, const Plan& pl,
               int B, int K, int N, hipStream_t st)
{
    const size_t lds = (size_t)pl.cv.total * sizeof(float);
    if (lds > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(actor_fwd_kernel<CT, V>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return MGP_ELAUNCH;
    hipLaunchKernelGGL((actor_fwd_kernel<CT, V>), dim3((unsigned)B * pl.ntiles), dim3(AF_THREADS), lds, st,
                       X, G, out, saved, P, pl.cv, B, K, N, pl.tw, pl.ntiles, pl.R, pl.MC, pl.ncp);
    return mgp_launch_status();
}

}  // namespace

extern "C" long mgp_actor_saved_floats(const int* dims, int n_layers, int B, int K, int N)
{
    if (dims == nullptr || n_layers <= 0 || n_layers > MGP_MAX_LAYERS || B <= 0 || K <= 0 || N <= 0) return 0;
    long tot = (long)B * dims[0] * K * N;
    for (int i = 1; i < n_layers; ++i) tot += (long)B * dims[i] * N;
    return tot;
}

extern "C" int mgp_actor_supported(const int* dims, int n_layers, int K, int N)
{
    if (dims == nullptr) return 0;
    Plan pl;
    return make_plan(dims, n_layers, K, N, true, &pl) ? 1 : 0;
}

extern "C" int mgp_actor_fwd(const float* X, const float* G, const float* const* W, const float* const* b,
                             const int* dims, int n_layers, float* out, float* saved,
                             int B, int K, int N, void* stream)
{
    if (dims == nullptr || W == nullptr || b == nullptr) return MGP_EINVAL;
    if (B < 0 || K <= 0 || N <= 0 || n_layers <= 0 || n_layers > MGP_MAX_LAYERS) return MGP_EINVAL;
    if (B == 0) return MGP_OK;
    MGP_CHECK_PTR(X); MGP_CHECK_PTR(G); MGP_CHECK_PTR(out);
    if (saved != nullptr && (reinterpret_cast<uintptr_t>(saved) & 3u)) return MGP_EALIGN;
    Plan pl;
    if (!make_plan(dims, n_layers, K, N, mgp_aligned16(G), &pl)) return MGP_EUNSUPPORTED;
    if ((long)B * pl.ntiles > 2147483647L) return MGP_EINVAL;
    ActorParams P;
    P.n_layers = n_layers;
    for (int i = 0; i <= n_layers; ++i) P.dims[i] = dims[i];
    for (int l = 0; l < n_layers; ++l) {
        MGP_CHECK_PTR(W[l]); MGP_CHECK_PTR(b[l]);
        P.W[l] = W[l]; P.b[l] = b[l]; P.woff[l] = pl.woff[l];
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    mgp_clear_error();
#define MGP_AF_CASE(CT, V) return launch_fwd<CT, V>(X, G, out, saved, P, pl, B, K, N, st)
    if (pl.V == 4) {
        if (pl.CT == 4) MGP_AF_CASE(4, 4);
        if (pl.CT == 6) MGP_AF_CASE(6, 4);
        MGP_AF_CASE(8, 4);
    } else {
        if (pl.CT == 4) MGP_AF_CASE(4, 1);
        if (pl.CT == 6) MGP_AF_CASE(6, 1);
        MGP_AF_CASE(8, 1);
    }
#undef MGP_AF_CASE
}

static long bwd_param_count(const int* dims, int n_layers, int K)
{
    long P = 0;
    for (int l = 0; l < n_layers; ++l) {
        const long cin = (l == 0) ? (long)dims[0] * K : dims[l];
        P += (long)dims[l + 1] * cin + dims[l + 1];
    }
    return P;
}

extern "C" long mgp_actor_bwd_workspace(const int* dims, int n_layers, int B, int K, int N)
{
    if (dims == nullptr || n_layers <= 0 || n_layers > MGP_MAX_LAYERS || B <= 0 || K <= 0 || N <= 0) return 0;
    const long ntiles = (long)B * ((N + AB_COLS - 1) / AB_COLS);
    return ntiles * bwd_param_count(dims, n_layers, K);
}

extern "C" int mgp_actor_bwd(const float* dOut, const float* saved, const float* const* W, const int* dims,
                             int n_layers, float* const* dW, float* const* db, int B, int K, int N,
                             float* workspace, void* stream)
{
    if (dims == nullptr || W == nullptr || dW == nullptr || db == nullptr) return MGP_EINVAL;
    if (B <= 0 || K <= 0 || N <= 0 || n_layers <= 0 || n_layers > MGP_MAX_LAYERS) return MGP_EINVAL;
    MGP_CHECK_PTR(dOut); MGP_CHECK_PTR(saved); MGP_CHECK_PTR(workspace);
    if (B > 65535) return MGP_EINVAL;
    BwdParams P;
    ScatterParams S;
    P.n_layers = S.n_layers = n_layers;
    long poff = 0, soff = 0;
    int maxw = dims[n_layers], maxin = 0;
    for (int i = 0; i <= n_layers; ++i) P.dims[i] = dims[i];
    for (int l = 0; l < n_layers; ++l) {
        MGP_CHECK_PTR(W[l]); MGP_CHECK_PTR(dW[l]); MGP_CHECK_PTR(db[l]);
        const int cin = (l == 0) ? dims[0] * K : dims[l];
        const int cout = dims[l + 1];
        P.W[l] = W[l];
        P.poff[l] = S.poff[l] = poff;
        P.soff[l] = soff;
        S.dW[l] = dW[l]; S.db[l] = db[l];
        S.wsz[l] = (long)cout * cin; S.bsz[l] = cout;
        poff += (long)cout * cin + cout;
        soff += (long)B * cin * N;
        if (cout > maxw) maxw = cout;
        if (cin > maxin) maxin = cin;
        if (cin > maxw && l > 0) maxw = cin;
    }
    const long Ptot = poff;
    const size_t lds = ((size_t)2 * maxw * 65 + (size_t)maxin * 65) * sizeof(float);
    if (lds > AF_LDS_LIMIT) return MGP_EUNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    mgp_clear_error();
    if (lds > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(actor_bwd_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return MGP_ELAUNCH;
    const int ntx = (N + AB_COLS - 1) / AB_COLS;
    hipLaunchKernelGGL(actor_bwd_kernel, dim3(ntx, B), dim3(AB_THREADS), lds, st, dOut, saved, workspace, P, Ptot,
                       K, N, maxw, maxin);
    int rc = mgp_launch_status();
    if (rc != MGP_OK) return rc;
    hipLaunchKernelGGL(actor_bwd_scatter_kernel, dim3((unsigned)((Ptot + 255) / 256)), dim3(256), 0, st, workspace, S,
                       Ptot, (long)B * ntx);
    return mgp_launch_status();
}
