// Device helpers shared by the episode-resident rollout kernels (rollout.hip) and the sparse policy kernels
// (sparse_policy.hip): MFMA operand layout of the <= 32-wide tanh MLP, the per-wave hidden-layer routine, DPP moves.
#ifndef MGP_ROLLOUT_COMMON_H
#define MGP_ROLLOUT_COMMON_H
#include "mgp_device.h"

namespace {

// MLP operand layout: the resident kernel covers layer widths <= 32, i.e. <= 8 MFMA k-steps, so an agent column of the
// activation buffer holds 4 x 8 floats (+4 pad) and a weight fragment lane 8 floats (+4 pad) -- half of actor_fused.hip's
// 64-wide layout.  36 and 12 words per lane keep a 16-lane ds_read_b128 group on disjoint banks.
#ifndef MGP_RO_KS
#define MGP_RO_KS 8                      // k-steps of an activation column: layer widths <= 4 * MGP_RO_KS (rollout_wide.hip: 16)
#endif
constexpr int RO_KS = MGP_RO_KS;
constexpr int RO_CS = 4 * RO_KS + 4;
#ifndef MGP_RO_BF16
#define MGP_RO_BF16 1                    // 0: the fp32-MFMA form everywhere (rollout_f32ref.hip, A/B builds of the harness)
#endif
// Hidden layers on split-bf16 MFMA (ro_layer_bf16 below): layer inputs of up to 32 channels are ONE K block of
// v_mfma_f32_16x16x32_bf16 (RO_KS = 8: rollout.hip, rollout_w128.hip), up to 64 channels TWO (RO_KS = 16: rollout_wide.hip).
constexpr bool RO_BF16_CHAIN = MGP_RO_BF16 && (RO_KS == 8 || RO_KS == 16);
constexpr int RO_KB = RO_KS / 8;          // K blocks of 32 input channels
// floats per lane in a weight fragment block: fp32 k-steps + pad, or [K block][piece][8 bf16] = 12 floats per block; 12 and 28
// words per lane keep a 16-lane ds_read_b128 group on disjoint banks (24 would put lanes 0 and 8 on the same four)
constexpr int RO_WFS = (RO_BF16_CHAIN && RO_KB == 2) ? 28 : RO_KS + 4;
// the N > 128 kernel of the 64-wide build keeps fp32 fragments (its state leaves no room for the 40 % larger image of the
// two-block form: N = 200 with [64, 64] would no longer fit 160 KB): `bf` selects the layout of an image
constexpr int RO_WFS_F32 = RO_KS + 4;
__host__ __device__ constexpr int ro_wfs(bool bf) { return bf ? RO_WFS : RO_WFS_F32; }
__host__ __device__ inline int rpos(int c) { return (c & 3) * RO_KS + (c >> 2); }   // channel -> slot (B-fragment order)
typedef float f32x2 __attribute__((ext_vector_type(2)));

// One hidden layer for the 16 agent columns a wave owns: D[mt] (16 x 16) = W[mt] (16 x cin) . Act (cin x 16) on fp32
// 16x16x4 MFMAs, both m-tiles of a 32-wide layer as two independent accumulator chains, bias preloaded into the
// accumulators, tanh on the accumulator registers.  The wave reads all its B fragments before it stores anything and
// therefore works in place; no other wave touches these columns, so hidden layers need no workgroup barrier between
// them.  (Splitting the m-tiles over two waves with ping-pong buffers and a barrier per layer measured the same.)
template <int MT> __device__ __forceinline__ bool ro_mlp_cols_bf16(float* pcol, const float* pw, const float* pbias, int lq);

template <int MT>
__device__ __forceinline__ void ro_mlp_cols(float* pcol, const float* pw, const float* pbias, int ksteps, int lq)
{
    if (ro_mlp_cols_bf16<MT>(pcol, pw, pbias, lq)) return;    // builds with <= 32 input channels per layer: split-bf16 MFMA (below)
    constexpr int CH = MT > 2 ? 2 : MT;                       // m-tiles in flight: two chains hide the MFMA latency, four spill
    float fb[RO_KS];
    const float4* pb = reinterpret_cast<const float4*>(pcol + lq * RO_KS);
#pragma unroll
    for (int i = 0; i < RO_KS / 4; ++i) { const float4 t = pb[i]; fb[4 * i] = t.x; fb[4 * i + 1] = t.y; fb[4 * i + 2] = t.z; fb[4 * i + 3] = t.w; }
#pragma unroll
    for (int h = 0; h < MT; h += CH) {                        // (every B fragment is in registers before the first store)
        float fa[CH][RO_KS];
        f32x4 acc[CH];
#pragma unroll
        for (int mt = 0; mt < CH; ++mt) {
            const float4* pa = reinterpret_cast<const float4*>(pw + (h + mt) * 64 * RO_WFS);
#pragma unroll
            for (int i = 0; i < RO_KS / 4; ++i) { const float4 u = pa[i]; fa[mt][4 * i] = u.x; fa[mt][4 * i + 1] = u.y; fa[mt][4 * i + 2] = u.z; fa[mt][4 * i + 3] = u.w; }
            const float4 bv = *reinterpret_cast<const float4*>(pbias + (h + mt) * 16);
            acc[mt][0] = bv.x; acc[mt][1] = bv.y; acc[mt][2] = bv.z; acc[mt][3] = bv.w;
        }
#pragma unroll
        for (int sg = 0; sg < RO_KS / 2; ++sg) {
            if (2 * sg < ksteps) {
#pragma unroll
                for (int s_ = 2 * sg; s_ < 2 * sg + 2; ++s_)
#pragma unroll
                    for (int mt = 0; mt < CH; ++mt)
                        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[mt][s_], fb[s_], acc[mt], 0, 0, 0);
            }
        }
        float z[CH][4];
#pragma unroll
        for (int mt = 0; mt < CH; ++mt)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) z[mt][rr] = tanh_fast(acc[mt][rr]);
#pragma unroll
        for (int mt = 0; mt < CH; ++mt)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) pcol[rr * RO_KS + (h + mt) * 4 + lq] = z[mt][rr];     // slot rpos(16 (h + mt) + 4 lq + rr)
    }
}

// ---- register-chained MLP (the episode-resident rollout kernels).  Layer l's 16x16 accumulator tile leaves lane (li, lq)
// holding rows 16 mt + 4 lq + rr (rr = 0..3) of column li -- which IS a valid B operand layout for layer l + 1 if its K index
// is enumerated as k-step s = 4 mt + rr, k-lane lq  <->  channel 16 mt + 4 lq + rr.  The weight image of every layer after the
// first stores its A fragments in that order (ro_chain_image_elem), so activations go from one layer's tanh straight into
// the next layer's MFMAs: no LDS round trip between layers, and the 2-wide output layer runs as one zero-padded m-tile on the
// same operands (rows 0, 1 of lanes lq == 0) instead of a separate VALU pass over LDS.
#ifndef MGP_RO_MAXMT
#define MGP_RO_MAXMT (MGP_RO_KS / 4)
#endif
constexpr int RO_MAXMT = MGP_RO_MAXMT;                         // m-tiles of the widest hidden layer (2: widths <= 32, 4: <= 64;
                                                               // rollout_w128.hip: 8 with RO_KS = 8 -- ONE hidden layer up to 128 wide)
constexpr int RO_OUTC = 16 * RO_MAXMT;                         // channels the output layer can read (= 4 RO_KS in the chained builds)
// m-tiles a hidden layer of `cout` rows is run with: 1, 2, 4 or 8 (padded up: few MLP code instances)
__host__ __device__ inline int ro_mt(int cout) { const int m = pad16(cout) / 16; return m <= 2 ? m : (m <= 4 ? 4 : 8); }

template <int MT, bool TANH, int WFS = RO_WFS_F32>            // WFS: floats per lane and m-tile of the fp32 fragment image
__device__ __forceinline__ void ro_layer_regs(const float (&fb)[RO_KS], const float* pw, const float* pbias, int ksteps,
                                              float (&zn)[RO_MAXMT][4])
{
    constexpr int CH = MT > 2 ? 2 : MT;                       // m-tiles in flight: two chains hide the MFMA latency, four spill
#pragma unroll
    for (int h = 0; h < MT; h += CH) {
        float fa[CH][RO_KS];
        f32x4 acc[CH];
#pragma unroll
        for (int mt = 0; mt < CH; ++mt) {
            const float4* pa = reinterpret_cast<const float4*>(pw + (h + mt) * 64 * WFS);
#pragma unroll
            for (int i = 0; i < RO_KS / 4; ++i) { const float4 u = pa[i]; fa[mt][4 * i] = u.x; fa[mt][4 * i + 1] = u.y; fa[mt][4 * i + 2] = u.z; fa[mt][4 * i + 3] = u.w; }
            const float4 bv = *reinterpret_cast<const float4*>(pbias + (h + mt) * 16);
            acc[mt][0] = bv.x; acc[mt][1] = bv.y; acc[mt][2] = bv.z; acc[mt][3] = bv.w;
        }
#pragma unroll
        for (int sg = 0; sg < RO_KS / 2; ++sg) {
            if (2 * sg < ksteps) {
#pragma unroll
                for (int s_ = 2 * sg; s_ < 2 * sg + 2; ++s_)
#pragma unroll
                    for (int mt = 0; mt < CH; ++mt)
                        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[mt][s_], fb[s_], acc[mt], 0, 0, 0);
            }
        }
#pragma unroll
        for (int mt = 0; mt < CH; ++mt)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) zn[h + mt][rr] = TANH ? tanh_fast(acc[mt][rr]) : acc[mt][rr];
    }
}

// ---- split-bf16 form of a 32-wide hidden layer (the compiled-in policy shape of the headline build).  The fp32 matrix pipe
// multiplies 16x16x4 per 8 passes; v_mfma_f32_16x16x32_bf16 multiplies 16x16x32 per 4.  With every fp32 operand written as
// x = x1 + x2 + x3 (three bf16 pieces: 24 significand bits, the split is exact) the products that matter at fp32 accuracy are
// w1 x1, w1 x2, w2 x1, w2 x2, w1 x3, w3 x1 (each exact in the fp32 accumulator's inputs; what is dropped is below 2^-24 of the
// product): six instructions of 16 cycles per m-tile cover K = 32, against five / eight of 32 cycles.  A lane's block record is
// the same 12 floats (48 bytes) as the fp32 fragments: [piece][8 bf16]; element j of k-group lq <-> channel 4 j + lq for the
// first layer (its B operand is the aggregation's LDS tile), 4 lq + j | 16 + 4 lq + (j - 4) for the second (B operand = the
// first layer's accumulator registers).
typedef __attribute__((ext_vector_type(8))) __bf16 ro_bf16x8;
#ifndef RO_MFMA_LARGEST_FIRST
#define RO_MFMA_LARGEST_FIRST 0           // order of a layer's six products: 0 = smallest first (the product build), 1 = largest first (A/B)
#endif

// Two values per v_cvt_pk_bf16_f32 (round to nearest even, the rounding of the scalar (__bf16) conversion): the packed word is
// unpacked with one shift and one mask, so a pair costs 3 conversions + 4 unpacks + 4 subtractions = 11 instructions.  (Written
// value by value, hipcc converted every piece alone AND again in pairs for the operand words: 15 per pair, 7 of them conversions
// at half the plain VALU rate -- profiles/r06_valu_rate.txt.)  Same pieces, bit for bit.
__device__ __forceinline__ unsigned int ro_cvt_pk_bf16(float lo, float hi)
{
    unsigned int r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}

__device__ __forceinline__ void ro_split3(const float* x /* [8] */, ro_bf16x8& h1, ro_bf16x8& h2, ro_bf16x8& h3)
{
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    u32x4 w1, w2, w3;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float x0 = x[2 * j], x1 = x[2 * j + 1];
        const unsigned int p = ro_cvt_pk_bf16(x0, x1);
        const float r0 = x0 - __uint_as_float(p << 16), r1 = x1 - __uint_as_float(p & 0xFFFF0000u);
        const unsigned int q = ro_cvt_pk_bf16(r0, r1);
        const float s0 = r0 - __uint_as_float(q << 16), s1 = r1 - __uint_as_float(q & 0xFFFF0000u);
        w1[j] = p; w2[j] = q; w3[j] = ro_cvt_pk_bf16(s0, s1);
    }
    h1 = __builtin_bit_cast(ro_bf16x8, w1); h2 = __builtin_bit_cast(ro_bf16x8, w2); h3 = __builtin_bit_cast(ro_bf16x8, w3);
}

// Two K blocks (layer inputs of up to 64 channels: rollout_wide.hip).  x[8 kb + j] <-> channel 16 (2 kb) + 4 lq + j for j < 4,
// 16 (2 kb + 1) + 4 lq + (j - 4) beyond (the accumulator registers of m-tiles 2 kb, 2 kb + 1 of the previous layer; layer 0
// reads its <= 32 aggregation channels as block 0 alone: nkb = 1).  One m-tile at a time, the two blocks as two independent
// accumulator chains (six dependent MFMAs each, smallest products first) added at the end: 24 registers of A pieces in flight.
// 12 MFMAs of 16 cycles per m-tile against 16 fp32 k-steps of 32.
template <int MT, bool TANH>
__device__ __forceinline__ void ro_layer_bf16_kb2(const float* x /* [16] */, const float* pw, const float* pbias,
                                                  float (&zn)[RO_MAXMT][4], int nkb)
{
    static_assert(MT <= RO_MAXMT, "m-tiles");
    ro_bf16x8 b1[2], b2[2], b3[2];
    ro_split3(x, b1[0], b2[0], b3[0]);
    if (nkb > 1) ro_split3(x + 8, b1[1], b2[1], b3[1]);       // (uniform)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const float4* pa = reinterpret_cast<const float4*>(pw + mt * 64 * RO_WFS);
        const float4 bv = *reinterpret_cast<const float4*>(pbias + mt * 16);
        f32x4 acc0 = f32x4{bv.x, bv.y, bv.z, bv.w}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
        {
            const float4 u1 = pa[0], u2 = pa[1], u3 = pa[2];
            const ro_bf16x8 a1 = *reinterpret_cast<const ro_bf16x8*>(&u1), a2 = *reinterpret_cast<const ro_bf16x8*>(&u2),
                            a3 = *reinterpret_cast<const ro_bf16x8*>(&u3);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b3[0], acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3, b1[0], acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b2[0], acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b2[0], acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b1[0], acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1[0], acc0, 0, 0, 0);
        }
        if (nkb > 1) {
            const float4 u1 = pa[3], u2 = pa[4], u3 = pa[5];
            const ro_bf16x8 a1 = *reinterpret_cast<const ro_bf16x8*>(&u1), a2 = *reinterpret_cast<const ro_bf16x8*>(&u2),
                            a3 = *reinterpret_cast<const ro_bf16x8*>(&u3);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b3[1], acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3, b1[1], acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b2[1], acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b2[1], acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b1[1], acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1[1], acc1, 0, 0, 0);
            acc0 += acc1;
        }
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) zn[mt][rr] = TANH ? tanh_fast(acc0[rr]) : acc0[rr];
    }
#pragma unroll
    for (int mt = MT; mt < RO_MAXMT; ++mt)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) zn[mt][rr] = 0.f;
}

template <int MT, bool TANH>
__device__ __forceinline__ void ro_layer_bf16(const float* x /* [8 RO_KB] */, const float* pw /* this lane's record of m-tile 0 */,
                                              const float* pbias, float (&zn)[RO_MAXMT][4], int nkb = 1)
{
    static_assert(MT <= RO_MAXMT, "m-tiles");
    if constexpr (RO_KB == 2) { ro_layer_bf16_kb2<MT, TANH>(x, pw, pbias, zn, nkb); return; }
    constexpr int CH = MT > 2 ? 2 : MT;                       // m-tiles in flight (two accumulator chains; their A pieces: 24 registers)
    ro_bf16x8 b1, b2, b3;
    ro_split3(x, b1, b2, b3);
#pragma unroll
    for (int h = 0; h < MT; h += CH) {
        f32x4 acc[CH];
        ro_bf16x8 a1[CH], a2[CH], a3[CH];
#pragma unroll
        for (int mt = 0; mt < CH; ++mt) {
            const float4* pa = reinterpret_cast<const float4*>(pw + (h + mt) * 64 * RO_WFS);
            const float4 u1 = pa[0], u2 = pa[1], u3 = pa[2];
            a1[mt] = *reinterpret_cast<const ro_bf16x8*>(&u1);
            a2[mt] = *reinterpret_cast<const ro_bf16x8*>(&u2);
            a3[mt] = *reinterpret_cast<const ro_bf16x8*>(&u3);
            const float4 bv = *reinterpret_cast<const float4*>(pbias + (h + mt) * 16);
            acc[mt][0] = bv.x; acc[mt][1] = bv.y; acc[mt][2] = bv.z; acc[mt][3] = bv.w;
        }
#if RO_MFMA_LARGEST_FIRST
        // (experiment: the product that needs only the FIRST pieces opens the chain, so that it can issue while the later pieces
        //  are still being split off)
#pragma unroll
        for (int mt = 0; mt < CH; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[mt], b1, acc[mt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < CH; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2[mt], b1, acc[mt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < CH; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3[mt], b1, acc[mt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < CH; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[mt], b2, acc[mt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < CH; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2[mt], b2, acc[mt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < CH; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[mt], b3, acc[mt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < CH; ++mt)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) zn[h + mt][rr] = TANH ? tanh_fast(acc[mt][rr]) : acc[mt][rr];
        continue;
#endif
        // smallest products first; the m-tiles of a chunk alternate (independent accumulator chains)
#pragma unroll
        for (int mt = 0; mt < CH; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[mt], b3, acc[mt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < CH; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3[mt], b1, acc[mt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < CH; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2[mt], b2, acc[mt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < CH; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[mt], b2, acc[mt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < CH; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2[mt], b1, acc[mt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < CH; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[mt], b1, acc[mt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < CH; ++mt)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) zn[h + mt][rr] = TANH ? tanh_fast(acc[mt][rr]) : acc[mt][rr];
    }
#pragma unroll
    for (int mt = MT; mt < RO_MAXMT; ++mt)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) zn[mt][rr] = 0.f;
}

// element e (a float slot = two bf16) of a hidden layer's block in that form; same block size as the fp32 fragments
__device__ __forceinline__ float ro_chain_image_elem_bf(const float* __restrict__ src, const float* __restrict__ bias, int cin,
                                                        int cout, int layer, int e)
{
    const int MT = ro_mt(cout), tot = MT * 64 * RO_WFS;
    if (e >= tot) { const int o = e - tot; return (o < cout) ? bias[o] : 0.f; }
    const int mt = e / (64 * RO_WFS), r1 = e - mt * (64 * RO_WFS);
    const int ln = r1 / RO_WFS, sl = r1 - ln * RO_WFS;
    const int kb = sl / 12, sb = sl - kb * 12;                // K block (32 input channels), slot inside its 12-float record
    const int piece = sb >> 2, pr = sb & 3;
    const int lqq = ln >> 4, o = mt * 16 + (ln & 15);
    unsigned int word = 0u;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int j = 2 * pr + t;
        // element j of k-group lq of block kb: the first layer reads the aggregation tile (channel 4 j + lq, block 0 only),
        // the others the accumulator registers of m-tiles 2 kb (j < 4) and 2 kb + 1 of the previous layer
        const int c = (layer == 0) ? (kb == 0 ? 4 * j + lqq : cin)
                                   : (j < 4 ? 32 * kb + 4 * lqq + j : 32 * kb + 16 + 4 * lqq + (j - 4));
        const float w = (kb < RO_KB && piece < 3 && o < cout && c < cin) ? src[(size_t)o * cin + c] : 0.f;
        const __bf16 a = (__bf16)w;
        const float r = w - (float)a;
        const __bf16 b = (__bf16)r;
        const float r2 = r - (float)b;
        const __bf16 h = piece == 0 ? a : (piece == 1 ? b : (__bf16)r2);
        unsigned short bits;
        __builtin_memcpy(&bits, &h, 2);
        word |= (unsigned int)bits << (16 * t);
    }
    return __uint_as_float(word);
}

// ---- [r6] two hidden layers of up to 128 channels on the resident kernel (rollout_w128x2.hip, MGP_RO_X2).  The second layer's
// piece image is 4 K blocks (32 input channels each) x 8 m-tiles x 64 lanes x 48 bytes = 96 KB -- with the first layer's 24 KB
// and 59 KB of episode state more than a CU's 160 KB.  Blocks 0 and 1 stay in LDS; blocks 2 and 3 share ONE 24 KB buffer and
// are streamed into it from the caller's image (L2) by LDS-DMA, each once per step, under compute: block 3 while the tile
// waves multiply blocks 0 and 1, block 2 (for the next step) under the simulator phases.  Image layout of that layer:
// [kb][mt][lane][piece][8 bf16] (a K block is one contiguous 24,576-byte span), then the bias [128]; element j of k-group lq of
// block kb <-> input channel 32 kb + (j < 4 ? 4 lq + j : 16 + 4 lq + (j - 4)): the accumulator registers of m-tiles 2 kb,
// 2 kb + 1 of the first layer.
constexpr int RO_X2_KB = 4;                                   // K blocks of the second layer
constexpr int RO_X2_BLK = 8 * 64 * 12;                        // floats per K block: 8 m-tiles x 64 lanes x 12
constexpr int RO_X2_L0 = 8 * 64 * 12 + 128;                   // first layer: 8 m-tiles of records + 128 bias values
constexpr int RO_X2_L1 = RO_X2_KB * RO_X2_BLK + 128;          // second layer: 4 K blocks + bias
constexpr int RO_X2_OUT = (2 * 128 + 2 + 15) & ~15;           // output layer: channel-ordered pairs + bias pair, padded
constexpr int RO_X2_IMAGE = RO_X2_L0 + RO_X2_L1 + RO_X2_OUT;  // floats of the image in HBM
constexpr int RO_X2_LDS = RO_X2_L0 + 2 * RO_X2_BLK + 128 + RO_X2_OUT + RO_X2_BLK;   // floats in LDS: blocks 0, 1 + the stream buffer

__device__ __forceinline__ float ro_x2_l1_elem(const float* __restrict__ src, const float* __restrict__ bias, int cin, int cout, int e)
{
    if (e >= RO_X2_KB * RO_X2_BLK) { const int o = e - RO_X2_KB * RO_X2_BLK; return (o < cout) ? bias[o] : 0.f; }
    const int kb = e / RO_X2_BLK, r0 = e - kb * RO_X2_BLK;
    const int mt = r0 / (64 * 12), r1 = r0 - mt * (64 * 12);
    const int ln = r1 / 12, sl = r1 - ln * 12;
    const int piece = sl >> 2, pr = sl & 3;
    const int lqq = ln >> 4, o = mt * 16 + (ln & 15);
    unsigned int word = 0u;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int j = 2 * pr + t;
        const int c = 32 * kb + (j < 4 ? 4 * lqq + j : 16 + 4 * lqq + (j - 4));
        const float w = (o < cout && c < cin) ? src[(size_t)o * cin + c] : 0.f;
        const __bf16 a = (__bf16)w;
        const float r = w - (float)a;
        const __bf16 b = (__bf16)r;
        const float r2 = r - (float)b;
        const __bf16 h = piece == 0 ? a : (piece == 1 ? b : (__bf16)r2);
        unsigned short bits;
        __builtin_memcpy(&bits, &h, 2);
        word |= (unsigned int)bits << (16 * t);
    }
    return __uint_as_float(word);
}

// first layer of that build: always eight m-tiles of records + 128 bias values (rows beyond cout are zero), the first-layer
// channel enumeration of ro_chain_image_elem_bf (element j of k-group lq <-> aggregation channel 4 j + lq)
__device__ __forceinline__ float ro_x2_l0_elem(const float* __restrict__ src, const float* __restrict__ bias, int cin, int cout, int e)
{
    if (e >= 8 * 64 * 12) { const int o = e - 8 * 64 * 12; return (o < cout) ? bias[o] : 0.f; }
    const int mt = e / (64 * 12), r1 = e - mt * (64 * 12);
    const int ln = r1 / 12, sl = r1 - ln * 12;
    const int piece = sl >> 2, pr = sl & 3;
    const int lqq = ln >> 4, o = mt * 16 + (ln & 15);
    unsigned int word = 0u;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int c = 4 * (2 * pr + t) + lqq;
        const float w = (o < cout && c < cin) ? src[(size_t)o * cin + c] : 0.f;
        const __bf16 a = (__bf16)w;
        const float r = w - (float)a;
        const __bf16 b = (__bf16)r;
        const float r2 = r - (float)b;
        const __bf16 h = piece == 0 ? a : (piece == 1 ? b : (__bf16)r2);
        unsigned short bits;
        __builtin_memcpy(&bits, &h, 2);
        word |= (unsigned int)bits << (16 * t);
    }
    return __uint_as_float(word);
}

// 16 bytes per active lane straight into LDS: LDS[dst_uniform + 16 * lane] = *src (global_load_lds_dwordx4; M0 carries the LDS
// base).  hipcc does not count the request; the hardware does: the issuing wave waits with s_waitcnt vmcnt(0) before the barrier
// in front of the first read, and every vmcnt wait it executes in between blocks on it.
__device__ __forceinline__ void ro_lds_dma16(const void* src, const void* dst_uniform)
{
    const unsigned int base = __builtin_amdgcn_readfirstlane((unsigned int)reinterpret_cast<uintptr_t>(dst_uniform));
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(base), "v"(src) : "memory", "m0");
}

// one K block of the second layer: acc[mt] += W[mt][kb] . x for the eight m-tiles, two at a time (independent accumulator
// chains), the six products of ro_layer_bf16 smallest first
template <int CH = 2>                                         // m-tiles in flight (a dependent chain of 16x16x32 MFMAs issues at the independent rate:
__device__ __forceinline__ void ro_x2_block(const float* blk /* this lane's record of m-tile 0 */, const ro_bf16x8& b1,   // profiles/r06_valu_rate.txt)
                                            const ro_bf16x8& b2, const ro_bf16x8& b3, f32x4 (&acc)[8])
{
#pragma unroll
    for (int h = 0; h < 8; h += CH) {
        ro_bf16x8 a1[CH], a2[CH], a3[CH];
#pragma unroll
        for (int mt = 0; mt < CH; ++mt) {
            const float4* pa = reinterpret_cast<const float4*>(blk + (h + mt) * 64 * 12);
            const float4 u1 = pa[0], u2 = pa[1], u3 = pa[2];
            a1[mt] = *reinterpret_cast<const ro_bf16x8*>(&u1);
            a2[mt] = *reinterpret_cast<const ro_bf16x8*>(&u2);
            a3[mt] = *reinterpret_cast<const ro_bf16x8*>(&u3);
        }
#pragma unroll
        for (int mt = 0; mt < CH; ++mt) acc[h + mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[mt], b3, acc[h + mt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < CH; ++mt) acc[h + mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3[mt], b1, acc[h + mt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < CH; ++mt) acc[h + mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2[mt], b2, acc[h + mt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < CH; ++mt) acc[h + mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[mt], b2, acc[h + mt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < CH; ++mt) acc[h + mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2[mt], b1, acc[h + mt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < CH; ++mt) acc[h + mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[mt], b1, acc[h + mt], 0, 0, 0);
    }
}

// the LDS-tile form of a hidden layer (ro_mlp_cols: every layer reads its B operand from the wave's activation columns, slot
// j of k-lane lq <-> channel 4 j + lq, and writes its output there) on the same instructions; its weight blocks are the
// first-layer enumeration of ro_chain_image_elem_bf for every layer (ro_weight_image_elem)
template <int MT>
__device__ __forceinline__ bool ro_mlp_cols_bf16(float* pcol, const float* pw, const float* pbias, int lq)
{
    if constexpr (RO_BF16_CHAIN && MT <= RO_MAXMT) {
        float fb[8];
        const float4* pb = reinterpret_cast<const float4*>(pcol + lq * RO_KS);
#pragma unroll
        for (int i = 0; i < 2; ++i) { const float4 t = pb[i]; fb[4 * i] = t.x; fb[4 * i + 1] = t.y; fb[4 * i + 2] = t.z; fb[4 * i + 3] = t.w; }
        float zn[RO_MAXMT][4];
        ro_layer_bf16<MT, true>(fb, pw, pbias, zn);           // (every B fragment is in registers before the first store)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) pcol[rr * RO_KS + mt * 4 + lq] = zn[mt][rr];     // slot rpos(16 mt + 4 lq + rr)
        return true;
    } else {
        return false;
    }
}

// element e of layer `layer`'s block in the chained weight image: A fragments [MT][64][RO_WFS] (lane = lq * 16 + (o & 15), slot
// s) + bias [MT * 16]; channel of (slot s, k-lane lq): 4 s + lq for the first layer (its B operand comes from the aggregation's
// LDS tile), 16 (s >> 2) + 4 lq + (s & 3) for the others (B operand = the previous layer's accumulator registers).  The output
// layer is one m-tile padded with zero rows.
__device__ __forceinline__ float ro_chain_image_elem(const float* __restrict__ src, const float* __restrict__ bias, int cin,
                                                     int cout, int layer, bool last, int e, bool bf = RO_BF16_CHAIN)
{
    if (bf && !last) return ro_chain_image_elem_bf(src, bias, cin, cout, layer, e);
    const int WFS = ro_wfs(false);
    const int MT = last ? 1 : ro_mt(cout), tot = MT * 64 * WFS;
    if (e >= tot) { const int o = e - tot; return (o < cout) ? bias[o] : 0.f; }
    const int mt = e / (64 * WFS), r1 = e - mt * (64 * WFS);
    const int ln = r1 / WFS, sl = r1 - ln * WFS;
    const int lqq = ln >> 4, o = mt * 16 + (ln & 15);
    const int c = (layer == 0) ? 4 * sl + lqq : 16 * (sl >> 2) + 4 * lqq + (sl & 3);
    return (sl < RO_KS && o < cout && c < cin) ? src[(size_t)o * cin + c] : 0.f;
}
__host__ __device__ inline int ro_chain_image_size(int cout, bool last, bool bf = RO_BF16_CHAIN)
{
    const int MT = last ? 1 : ro_mt(cout);
    return MT * 64 * ro_wfs(bf) + MT * 16;
}

__device__ __forceinline__ int ro_dim(unsigned long long dimsA, unsigned int dims8, int l)
{
    return (l < 8) ? (int)((dimsA >> (8 * l)) & 255ull) : (int)dims8;
}

// the thread index as a value the optimiser cannot trace back to threadIdx.x: whatever is derived from it is computed where
// it is used instead of being hoisted out of the step loop (and, under register pressure, spilled and reloaded every phase)
template <bool OPAQUE>
__device__ __forceinline__ int ro_fresh_tid_()
{
    int t = (int)threadIdx.x;
    if (OPAQUE) asm volatile("" : "+v"(t));
    return t;
}

__device__ __forceinline__ int ro_slot(int cur, int k, int K) { int s = cur - k; return s < 0 ? s + K : s; }

// Cross-lane adds on the DPP path (one VALU instruction per move; __shfl_xor compiles to ds_bpermute + address math).
// 0xB1 = quad_perm [1,0,3,2] (lane ^ 1), 0x4E = quad_perm [2,3,0,1] (lane ^ 2), 0x141 = row_half_mirror (i -> 7 - i).
template <int CTRL> __device__ __forceinline__ float dpp_f(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int CTRL> __device__ __forceinline__ unsigned int dpp_u(unsigned int v)
{
    return (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
template <int CTRL> __device__ __forceinline__ double dpp_d(double v)
{
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned int lo = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)(unsigned int)b, CTRL, 0xF, 0xF, true);
    const unsigned int hi = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)(unsigned int)(b >> 32), CTRL, 0xF, 0xF, true);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
// max over the 64 lanes, valid in lane 63: four DPP steps inside each row of 16, then row_bcast15 / row_bcast31
__device__ __forceinline__ float wave_max_to_last(float v)
{
    v = fmaxf(v, dpp_f<0xB1>(v));
    v = fmaxf(v, dpp_f<0x4E>(v));
    v = fmaxf(v, dpp_f<0x141>(v));                                   // row_half_mirror
    v = fmaxf(v, dpp_f<0x140>(v));                                   // row_mirror: every lane holds its row's max
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xA, 0xF, true)));   // row_bcast15 -> rows 1, 3
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xC, 0xF, true)));   // row_bcast31 -> rows 2, 3
    return v;
}


// sum of v over the four 16-lane rows of the wave (lanes l, l ^ 16, l ^ 32, l ^ 48), the same bits in every lane: rows 0 + 1
// and 2 + 3 with v_permlane16_swap (odd rows of one copy <-> even rows of the other), then the halves with
// v_permlane32_swap.  Inline asm with two distinct registers (the builtin folds identical operands and adds a copy to itself);
// s_nop 1 = the two wait states the swap needs after a VALU write of its operands.
__device__ __forceinline__ float rows_sum4(float v)
{
    unsigned int a = __float_as_uint(v), b = __float_as_uint(v);
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    v = __uint_as_float(a) + __uint_as_float(b);
    a = __float_as_uint(v); b = a;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return __uint_as_float(a) + __uint_as_float(b);
}

// rows_sum4 for six values at once (one pair of wait states per swap kind)
__device__ __forceinline__ void rows_sum4x6(float (&v)[6])
{
    unsigned int a[6], b[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) { a[i] = __float_as_uint(v[i]); b[i] = a[i]; }
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\tv_permlane16_swap_b32 %4, %5\n\t"
                 "v_permlane16_swap_b32 %6, %7\n\tv_permlane16_swap_b32 %8, %9\n\tv_permlane16_swap_b32 %10, %11"
                 : "+v"(a[0]), "+v"(b[0]), "+v"(a[1]), "+v"(b[1]), "+v"(a[2]), "+v"(b[2]), "+v"(a[3]), "+v"(b[3]), "+v"(a[4]), "+v"(b[4]),
                   "+v"(a[5]), "+v"(b[5]));
#pragma unroll
    for (int i = 0; i < 6; ++i) { v[i] = __uint_as_float(a[i]) + __uint_as_float(b[i]); a[i] = __float_as_uint(v[i]); b[i] = a[i]; }
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\tv_permlane32_swap_b32 %4, %5\n\t"
                 "v_permlane32_swap_b32 %6, %7\n\tv_permlane32_swap_b32 %8, %9\n\tv_permlane32_swap_b32 %10, %11"
                 : "+v"(a[0]), "+v"(b[0]), "+v"(a[1]), "+v"(b[1]), "+v"(a[2]), "+v"(b[2]), "+v"(a[3]), "+v"(b[3]), "+v"(a[4]), "+v"(b[4]),
                   "+v"(a[5]), "+v"(b[5]));
#pragma unroll
    for (int i = 0; i < 6; ++i) v[i] = __uint_as_float(a[i]) + __uint_as_float(b[i]);
}

// sum over the 64 lanes of a wave on the VALU cross-lane paths (DPP inside rows of 16, lane swaps across rows), every lane
// gets the total; fixed order.  (__shfl_xor compiles to ds_bpermute: 12 dependent LDS-pipe round trips per fp64 sum.)
__device__ __forceinline__ double wave_sum_d(double v)
{
    v += dpp_d<0xB1>(v);                                             // lane ^ 1
    v += dpp_d<0x4E>(v);                                             // lane ^ 2
    v += dpp_d<0x141>(v);                                            // row_half_mirror
    v += dpp_d<0x140>(v);                                            // row_mirror: every lane holds its row's sum
    unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    unsigned int a0 = (unsigned int)b, b0 = a0, a1 = (unsigned int)(b >> 32), b1 = a1;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3" : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1));
    v = __builtin_bit_cast(double, ((unsigned long long)a1 << 32) | a0) + __builtin_bit_cast(double, ((unsigned long long)b1 << 32) | b0);
    b = __builtin_bit_cast(unsigned long long, v);
    a0 = (unsigned int)b; b0 = a0; a1 = (unsigned int)(b >> 32); b1 = a1;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3" : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1));
    return __builtin_bit_cast(double, ((unsigned long long)a1 << 32) | a0) + __builtin_bit_cast(double, ((unsigned long long)b1 << 32) | b0);
}

// Weight image of a policy whose layers are <= 32 wide: hidden layers as MFMA A-fragments [MT][64][RO_WFS] + bias [MT*16]
// (lane = (c & 3) * 16 + (o & 15), slot s = c >> 2, zero padded), the 2-wide output layer as channel-ordered pairs
// (W[0][c], W[1][c]) padded to 32 channels + the bias pair.  `e` indexes the image of layer l (woff[l] excluded).
__device__ __forceinline__ float ro_weight_image_elem(const float* __restrict__ src, const float* __restrict__ bias, int cin,
                                                      int cout, bool last, int e)
{
    if (last) {
        const int c = e >> 1, o = e & 1;
        return (c < RO_OUTC) ? ((c < cin) ? src[(size_t)o * cin + c] : 0.f) : bias[o];
    }
    if (RO_BF16_CHAIN) return ro_chain_image_elem_bf(src, bias, cin, cout, 0, e);       // (every layer: the LDS-tile enumeration)
    const int MT = ro_mt(cout), tot = MT * 64 * RO_WFS;
    if (e >= tot) { const int o = e - tot; return (o < cout) ? bias[o] : 0.f; }
    const int mt = e / (64 * RO_WFS), r1 = e - mt * (64 * RO_WFS);
    const int ln = r1 / RO_WFS, sl = r1 - ln * RO_WFS;
    const int c = 4 * sl + (ln >> 4), o = mt * 16 + (ln & 15);
    return (sl < RO_KS && o < cout && c < cin) ? src[(size_t)o * cin + c] : 0.f;
}
__host__ __device__ inline int ro_weight_image_size(int cout, bool last)
{
    return last ? 2 * RO_OUTC + 2 : ro_mt(cout) * 64 * RO_WFS + ro_mt(cout) * 16;
}

}  // namespace
#endif
