// VALU issue-rate microbenchmark for gfx950 (MI355X): what a wave64 vector instruction costs its SIMD, per class.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -o scratch/valu_rate tools/harness/valu_rate.hip && ./scratch/valu_rate
//   (counter pass: rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -- ./scratch/valu_rate pmc)
//
// One 64 x W thread workgroup per CU (256 workgroups), W = 4 / 8 / 16 waves = 1 / 2 / 4 waves per SIMD.  Every wave runs
// REPS x UNROLL copies of one instruction between two s_memtime reads (shader cycles), either on eight independent
// registers (issue rate) or on one register (dependent-chain latency).  Reported per class and occupancy:
//   cyc/inst/SIMD  = (latest end - earliest start of the workgroup's waves) / (instructions per wave x waves per SIMD)
//   cyc/inst/wave  = one wave's own (end - start) / its instructions
// `bench.py` prices the resident kernel's instruction mix with the 4-waves-per-SIMD column (profiles/<round>_valu_rate.txt).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <algorithm>

#define UNROLL 64
#define REPS 64

// eight independent destinations v[0..7]; sources a, b are loop-invariant
#define R8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define R64(OP) R8(OP) R8(OP) R8(OP) R8(OP) R8(OP) R8(OP) R8(OP) R8(OP)

enum Cls { FMA_F32, MUL_F32, PK_FMA_F32, FMA_F64, MUL_F64, ADD_F64, RCP_F32, EXP_F32, RCP_F64, CVT_PK_BF16, ALIGNBIT, DPP_MOV, DPP_ADD,
           CMP_F32, CNDMASK, AND_B32, ADD_U32, MUL_LO_U32, MAD_U24, CVT_F64_F32, CVT_F32_F64, BCNT, PERMLANE32, LSHL_B64, READLANE,
           DS_READ_B32, DS_READ_B128, PERM_B32, AND_OR, LSHL_ADD, LSHLREV, MOV_B32, CMP_F64, MAD_U64, FMA_NOP, MFMA_BF16, MAX_F32,
           CVT_F32_U32, DS_WRITE_B32, DS_WRITE_B64, DS_READ_U8, N_CLS };
static const char* cls_name[N_CLS] = {
    "v_fma_f32", "v_mul_f32", "v_pk_fma_f32", "v_fma_f64", "v_mul_f64", "v_add_f64", "v_rcp_f32", "v_exp_f32", "v_rcp_f64",
    "v_cvt_pk_bf16_f32", "v_alignbit_b32", "v_mov_b32 dpp row_shl:1", "v_add_f32 dpp row_shr:1", "v_cmp_lt_f32 (sgpr pair)",
    "v_cndmask_b32", "v_and_b32", "v_add_u32", "v_mul_lo_u32", "v_mad_u32_u24", "v_cvt_f64_f32", "v_cvt_f32_f64", "v_bcnt_u32_b32",
    "v_permlane32_swap", "v_lshlrev_b64", "v_readlane_b32", "ds_read_b32", "ds_read_b128",
    "v_perm_b32", "v_and_or_b32", "v_lshl_add_u32", "v_lshlrev_b32", "v_mov_b32", "v_cmp_lt_f64 (sgpr pair)", "v_mad_u64_u32",
    "v_fma_f32 + s_nop 0", "v_mfma_f32_16x16x32_bf16", "v_max_f32", "v_cvt_f32_u32", "ds_write_b32", "ds_write_b64", "ds_read_u8"};

template <int CLS, bool DEP>
__global__ void __launch_bounds__(1024) rate_kernel(unsigned long long* stamps, float* sink, int reps) {
    __shared__ float lds[4096];
    float v[8]; double d[8]; float2 p[8];
    const float a = 1.0f + 1e-7f * threadIdx.x, b = 1e-9f * threadIdx.x;
    const double da = 1.0 + 1e-9 * threadIdx.x, db = 1e-12 * threadIdx.x;
    const float2 pa = {a, a}, pb = {b, b};
    for (int i = 0; i < 8; ++i) { v[i] = 0.5f + i; d[i] = 0.5 + i; p[i] = {v[i], v[i]}; }
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = i;
    unsigned ldsaddr = (threadIdx.x * 16) & 16383;
    float4 q[8];
    for (int i = 0; i < 8; ++i) q[i] = {0, 0, 0, 0};
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int r = 0; r < reps; ++r) {
#define IX(i) (DEP ? 0 : (i))
        if constexpr (CLS == FMA_F32) {
#define OP(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[IX(i)]) : "v"(a), "v"(b));
            R64(OP)
#undef OP
        } else if constexpr (CLS == MUL_F32) {
#define OP(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[IX(i)]) : "v"(a));
            R64(OP)
#undef OP
        } else if constexpr (CLS == PK_FMA_F32) {
#define OP(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[IX(i)]) : "v"(pa), "v"(pb));
            R64(OP)
#undef OP
        } else if constexpr (CLS == FMA_F64) {
#define OP(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[IX(i)]) : "v"(da), "v"(db));
            R64(OP)
#undef OP
        } else if constexpr (CLS == MUL_F64) {
#define OP(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[IX(i)]) : "v"(da));
            R64(OP)
#undef OP
        } else if constexpr (CLS == ADD_F64) {
#define OP(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[IX(i)]) : "v"(db));
            R64(OP)
#undef OP
        } else if constexpr (CLS == RCP_F32) {
#define OP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[IX(i)]));
            R64(OP)
#undef OP
        } else if constexpr (CLS == EXP_F32) {
#define OP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[IX(i)]));
            R64(OP)
#undef OP
        } else if constexpr (CLS == RCP_F64) {
#define OP(i) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[IX(i)]));
            R64(OP)
#undef OP
        } else if constexpr (CLS == CVT_PK_BF16) {
#define OP(i) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v[IX(i)]) : "v"(a));
            R64(OP)
#undef OP
        } else if constexpr (CLS == ALIGNBIT) {
#define OP(i) asm volatile("v_alignbit_b32 %0, %0, %1, 16" : "+v"(v[IX(i)]) : "v"(a));
            R64(OP)
#undef OP
        } else if constexpr (CLS == DPP_MOV) {
            // a DPP read of a register the previous VALU wrote needs two wait states (the compiler does not see into inline asm)
            if constexpr (DEP) {
#define OP(i) asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %0 row_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(v[0]));
                R64(OP)
#undef OP
            } else {
#define OP(i) asm volatile("v_mov_b32_dpp %0, %0 row_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(v[i]));
                R64(OP)
#undef OP
            }
        } else if constexpr (CLS == DPP_ADD) {
            if constexpr (DEP) {
#define OP(i) asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(v[0]) : "v"(a));
                R64(OP)
#undef OP
            } else {
#define OP(i) asm volatile("v_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(v[i]) : "v"(a));
                R64(OP)
#undef OP
            }
        } else if constexpr (CLS == CMP_F32) {
            unsigned long long m[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define OP(i) asm volatile("v_cmp_lt_f32 %0, %1, %2" : "=s"(m[i]) : "v"(v[i]), "v"(a));
            R64(OP)
#undef OP
            for (int i = 0; i < 8; ++i) v[0] += (float)(unsigned)m[i];
        } else if constexpr (CLS == CNDMASK) {
            const unsigned long long msk = 0x5555555555555555ull + reps;
#define OP(i) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(v[IX(i)]) : "v"(a), "s"(msk));
            R64(OP)
#undef OP
        } else if constexpr (CLS == AND_B32) {
#define OP(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(v[IX(i)]) : "v"(a));
            R64(OP)
#undef OP
        } else if constexpr (CLS == ADD_U32) {
#define OP(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[IX(i)]) : "v"(a));
            R64(OP)
#undef OP
        } else if constexpr (CLS == MUL_LO_U32) {
#define OP(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(v[IX(i)]) : "v"(a));
            R64(OP)
#undef OP
        } else if constexpr (CLS == MAD_U24) {
#define OP(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(v[IX(i)]) : "v"(a), "v"(b));
            R64(OP)
#undef OP
        } else if constexpr (CLS == CVT_F64_F32) {
#define OP(i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[IX(i)]) : "v"(v[IX(i)]));
            R64(OP)
#undef OP
        } else if constexpr (CLS == CVT_F32_F64) {
#define OP(i) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(v[IX(i)]) : "v"(d[IX(i)]));
            R64(OP)
#undef OP
        } else if constexpr (CLS == BCNT) {
#define OP(i) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(v[IX(i)]) : "v"(a));
            R64(OP)
#undef OP
        } else if constexpr (CLS == PERMLANE32) {
            if constexpr (DEP) {
#define OP(i) asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(v[0]), "+v"(v[1]));
                R64(OP)
#undef OP
            } else {                                             // pairs (0,1) (2,3) (4,5) (6,7): four independent swaps in rotation
#define OP(i) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(v[((i) & 3) * 2]), "+v"(v[((i) & 3) * 2 + 1]));
                R64(OP)
#undef OP
            }
        } else if constexpr (CLS == LSHL_B64) {
#define OP(i) asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(d[IX(i)]));
            R64(OP)
#undef OP
        } else if constexpr (CLS == READLANE) {
            unsigned s[8];
#define OP(i) asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s[i]) : "v"(v[i]));
            R64(OP)
#undef OP
            for (int i = 0; i < 8; ++i) v[0] += (float)s[i];
        } else if constexpr (CLS == DS_READ_B32) {
            // DEP: the address of the next read is the previous read's result (pointer chase: issue -> use latency)
            if constexpr (DEP) {
                unsigned ad = (threadIdx.x * 4) & 16383;
#define OP(i) asm volatile("ds_read_b32 %0, %0\n\ts_waitcnt lgkmcnt(0)\n\tv_and_b32 %0, 0x3ffc, %0" : "+v"(ad) :: "memory");
                R64(OP)
#undef OP
                v[0] += (float)ad;
            } else {
#define OP(i) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v[i]) : "v"((threadIdx.x * 4u) & 16383u), "n"(256 * (i)) : "memory");
                R64(OP)
#undef OP
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        } else if constexpr (CLS == PERM_B32) {
#define OP(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(v[IX(i)]) : "v"(a), "v"(b));
            R64(OP)
#undef OP
        } else if constexpr (CLS == AND_OR) {
#define OP(i) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(v[IX(i)]) : "v"(a), "v"(b));
            R64(OP)
#undef OP
        } else if constexpr (CLS == LSHL_ADD) {
#define OP(i) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(v[IX(i)]) : "v"(a));
            R64(OP)
#undef OP
        } else if constexpr (CLS == LSHLREV) {
#define OP(i) asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(v[IX(i)]));
            R64(OP)
#undef OP
        } else if constexpr (CLS == MOV_B32) {
#define OP(i) asm volatile("v_mov_b32 %0, %1" : "=v"(v[IX(i)]) : "v"(v[(IX(i) + 1) & 7]));
            R64(OP)
#undef OP
        } else if constexpr (CLS == CMP_F64) {
            unsigned long long m[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define OP(i) asm volatile("v_cmp_lt_f64 %0, %1, %2" : "=s"(m[i]) : "v"(d[i]), "v"(da));
            R64(OP)
#undef OP
            for (int i = 0; i < 8; ++i) v[0] += (float)(unsigned)m[i];
        } else if constexpr (CLS == MAD_U64) {
#define OP(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(d[IX(i)]) : "v"(a), "v"(b) : "vcc");
            R64(OP)
#undef OP
        } else if constexpr (CLS == FMA_NOP) {
#define OP(i) asm volatile("v_fma_f32 %0, %0, %1, %2\n\ts_nop 0" : "+v"(v[IX(i)]) : "v"(a), "v"(b));
            R64(OP)
#undef OP
        } else if constexpr (CLS == MFMA_BF16) {
            // independent: eight accumulator tiles in rotation; dependent: one accumulator
            typedef float f4 __attribute__((ext_vector_type(4)));
            f4 acc[8], opa = {a, b, a, b}, opb = {b, a, b, a};
            for (int i = 0; i < 8; ++i) acc[i] = f4{v[i], v[i], v[i], v[i]};
#define OP(i) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[IX(i)]) : "v"(opa), "v"(opb));
            R64(OP)
#undef OP
            for (int i = 0; i < 8; ++i) v[i] = acc[i][0] + acc[i][3];
        } else if constexpr (CLS == MAX_F32) {
#define OP(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[IX(i)]) : "v"(a));
            R64(OP)
#undef OP
        } else if constexpr (CLS == CVT_F32_U32) {
#define OP(i) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(v[IX(i)]));
            R64(OP)
#undef OP
        } else if constexpr (CLS == DS_WRITE_B32) {
            unsigned wa = (threadIdx.x * 4) & 16383;             // conflict-free: consecutive dwords
#define OP(i) asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"(wa), "v"(v[i]), "n"(256 * (i)) : "memory");
            R64(OP)
#undef OP
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if constexpr (CLS == DS_WRITE_B64) {
            unsigned wa = (threadIdx.x * 8) & 16383;
#define OP(i) asm volatile("ds_write_b64 %0, %1" :: "v"(wa), "v"(d[i]) : "memory");
            R64(OP)
#undef OP
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if constexpr (CLS == DS_READ_U8) {
            unsigned ra = (threadIdx.x * 4) & 16383;             // conflict-free
#define OP(i) asm volatile("ds_read_u8 %0, %1 offset:%2" : "=v"(v[i]) : "v"(ra), "n"(256 * (i) + 1) : "memory");
            R64(OP)
#undef OP
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if constexpr (CLS == DS_READ_B128) {
#define OP(i) asm volatile("ds_read_b128 %0, %1" : "=v"(q[i]) : "v"(ldsaddr) : "memory");
            R64(OP)
#undef OP
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float acc = 0.f;
    for (int i = 0; i < 8; ++i) acc += v[i] + (float)d[i] + p[i].x + p[i].y + q[i].x + q[i].w;
    if (acc == 12345.678f) sink[threadIdx.x] = acc + lds[threadIdx.x & 4095];
    if ((threadIdx.x & 63) == 0) {
        const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
        stamps[((size_t)blockIdx.x * nw + w) * 2 + 0] = t0;
        stamps[((size_t)blockIdx.x * nw + w) * 2 + 1] = t1;
    }
}

struct Row { double simd[3], wave1, dep; };

template <int CLS, bool DEP>
static void run_one(int waves, unsigned long long* dstamps, float* sink, double& per_simd, double& per_wave, double& wall_us, int blocks) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    rate_kernel<CLS, DEP><<<blocks, 64 * waves>>>(dstamps, sink, REPS);          // warm-up (instruction cache, clocks)
    hipDeviceSynchronize();
    hipEventRecord(e0);
    rate_kernel<CLS, DEP><<<blocks, 64 * waves>>>(dstamps, sink, REPS);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1); wall_us = 1e3 * ms;
    std::vector<unsigned long long> h((size_t)blocks * waves * 2);
    hipMemcpy(h.data(), dstamps, h.size() * 8, hipMemcpyDeviceToHost);
    const double n_inst = (double)REPS * UNROLL;
    std::vector<double> ps, pw;
    for (int b = 0; b < blocks; ++b) {
        unsigned long long lo = ~0ull, hi = 0;
        for (int w = 0; w < waves; ++w) {
            const unsigned long long a = h[((size_t)b * waves + w) * 2], z = h[((size_t)b * waves + w) * 2 + 1];
            lo = std::min(lo, a); hi = std::max(hi, z);
            pw.push_back((double)(z - a) / n_inst);
        }
        ps.push_back((double)(hi - lo) / (n_inst * (waves / 4.0)));
    }
    std::sort(ps.begin(), ps.end()); std::sort(pw.begin(), pw.end());
    per_simd = ps[ps.size() / 2]; per_wave = pw[pw.size() / 2];
    hipEventDestroy(e0); hipEventDestroy(e1);
}

template <int CLS>
static void run_cls(unsigned long long* dstamps, float* sink, int blocks, bool pmc) {
    double s, w, us;
    if (pmc) {                                                   // one launch per form: the counter pass reads them per dispatch
        run_one<CLS, false>(16, dstamps, sink, s, w, us, blocks);
        printf("%-26s 4 waves/SIMD independent: %.2f cyc/inst/SIMD\n", cls_name[CLS], s);
        return;
    }
    double simd[3], wave[3], dep;
    const int ws[3] = {4, 8, 16};
    for (int i = 0; i < 3; ++i) run_one<CLS, false>(ws[i], dstamps, sink, simd[i], wave[i], us, blocks);
    double ds, dw; run_one<CLS, true>(4, dstamps, sink, ds, dw, us, blocks); dep = dw;
    printf("%-26s | %6.2f %6.2f %6.2f | %6.2f %6.2f %6.2f | %6.2f\n", cls_name[CLS], simd[0], simd[1], simd[2], wave[0], wave[1], wave[2], dep);
}

int main(int argc, char** argv) {
    const bool pmc = argc > 1 && !strcmp(argv[1], "pmc");
    const int blocks = 256;
    unsigned long long* dstamps; float* sink;
    hipMalloc(&dstamps, (size_t)blocks * 16 * 2 * 8); hipMalloc(&sink, 4096 * 4);
    if (!pmc) {
        printf("# tools/harness/valu_rate.hip on MI355X (gfx950): %d workgroups (one per CU), %d instructions per wave between two s_memtime reads (shader cycles)\n", blocks, REPS * UNROLL);
        printf("# independent = eight destination registers round-robin; dependent = one register, one wave per SIMD\n");
        printf("# %-24s | cyc/inst/SIMD at 1 2 4 waves/SIMD | cyc/inst/wave at 1 2 4 waves/SIMD | dependent cyc/inst\n", "instruction (wave64)");
    }
#define RUN(C) run_cls<C>(dstamps, sink, blocks, pmc);
    RUN(FMA_F32) RUN(MUL_F32) RUN(PK_FMA_F32) RUN(FMA_F64) RUN(MUL_F64) RUN(ADD_F64) RUN(RCP_F32) RUN(EXP_F32) RUN(RCP_F64)
    RUN(CVT_PK_BF16) RUN(ALIGNBIT) RUN(DPP_MOV) RUN(DPP_ADD) RUN(CMP_F32) RUN(CNDMASK) RUN(AND_B32) RUN(ADD_U32) RUN(MUL_LO_U32)
    RUN(MAD_U24) RUN(CVT_F64_F32) RUN(CVT_F32_F64) RUN(BCNT) RUN(PERMLANE32) RUN(LSHL_B64) RUN(READLANE) RUN(DS_READ_B32) RUN(DS_READ_B128)
    RUN(PERM_B32) RUN(AND_OR) RUN(LSHL_ADD) RUN(LSHLREV) RUN(MOV_B32) RUN(CMP_F64) RUN(MAD_U64) RUN(FMA_NOP) RUN(MFMA_BF16) RUN(MAX_F32)
    RUN(CVT_F32_U32) RUN(DS_WRITE_B32) RUN(DS_WRITE_B64) RUN(DS_READ_U8)
    return 0;
}
