// Device helpers shared by the MFMA GEMM kernels of libdgq_hip.so (gemm_wxa8.hip, linear_fused.hip).
#pragma once
#include "dgq_common.h"

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// LDS-DMA of 16 B per lane: LDS[lds_addr + lane*16] <- *gsrc.  Issued from inline asm on purpose: with the builtin
// hipcc treats the DMA as an LDS store that every later ds_read may alias and drains it with s_waitcnt vmcnt(0)
// before the first ds_read of each K step; here the ring is ordered by hand (counted vmcnt + barrier below).
// M0 carries the wave-uniform LDS base: it is passed as an INPUT OPERAND bound to the physical register ("{m0}"), so
// hipcc emits the s_mov_b32 m0 itself and tracks the register like any other (defined behaviour; the round-1 form wrote
// M0 inside the asm and listed it as a clobber, which clang rejects as a reserved register and does not honour).  The
// s_nop covers the M0-write -> LDS-DMA wait state, which the hazard recogniser does not see inside an asm statement.
// Measured on 8192^3: each DMA piece costs ~4.5 % of the loop (skipping the two weight pieces of the six per wave per K
// tile: 521 -> 474 us) — the largest non-MFMA cost, ~100 cycles per piece against 512 cycles of MFMA per wave per K tile.
__device__ __forceinline__ void glds16(const void* gsrc, uint32_t lds_addr) {
    asm volatile("s_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gsrc), "{m0}"(lds_addr) : "memory");
}

// y -> [aqtizer_{q,k,v}(y)] -> [+ residual]; element (m, n) of the output
__device__ __forceinline__ float dgq_extra(const dgq_gemm_extra_t& ex, float y, int m, int n) {
    if (ex.fq_mode) {                                   // quant_layer.py:295-299 on the projection output (sd.py:174-182,199)
        const int t = m % ex.fq_T;
        if (t >= ex.fq_skip) {
            const int idx = ex.fq_mode == 1 ? 0 : (ex.fq_mode == 2 ? t - ex.fq_skip : n % ex.fq_D);
            const float d = ex.fq_delta[idx], z = ex.fq_zp[idx];
            y = d * (dgq_affine_code(y, d, z, ex.fq_qmax) - z);
        }
    }
    if (ex.residual) {
        const int64_t i = (int64_t)(m / ex.res_div) * ex.ldr + n;
        if (ex.res_dtype == DGQ_F16) y += __half2float(reinterpret_cast<const __half*>(ex.residual)[i]);
        else if (ex.res_dtype == DGQ_BF16) y += __bfloat162float(reinterpret_cast<const __hip_bfloat16*>(ex.residual)[i]);
        else y += reinterpret_cast<const float*>(ex.residual)[i];
    }
    return y;
}


// Running totals of the per-K W4 kernels as floats without v_cvt_f32_i32: their int32 accumulators start at (and are cleared to)
// DGQ_ACC_BIAS_I = bits(1.5·2^23); while |T| < 2^22 (dgq_amd/plan.py:seg_limit) the bits of bias + T ARE the float 1.5·2^23 + T, and
// subtracting 1.5·2^23 is exact: float(T) by one packed add per two totals instead of two conversions.  The K loop of the short
// per-K launches is VALU-issue bound (cvt + fma per accumulator register and chunk: profiles/r05_small_launch_timeline.txt).
constexpr int DGQ_ACC_BIAS_I = 0x4B400000;
constexpr float DGQ_ACC_BIAS_F = 12582912.0f;
template <bool BIASED>
__device__ __forceinline__ float dgq_total_to_float(int t) {
    if constexpr (BIASED) return __int_as_float(t) - DGQ_ACC_BIAS_F;
    else return (float)t;
}

// Dequantising epilogue of one output element, shared by every GEMM-family kernel so that their results agree bit for bit:
//   y = alpha·(R0·acc − zw·R1 + R2·vn) + gamma     per-K: R0 = 1, R1 = Σ_k δ_k s, R2 = 0;  per-M: R0 = δ_m, R1 = δ_m·Σ_k s,
//   R2 = δ_m·(offset − z_m)
// as explicit FMAs (the build uses -ffp-contract=off: written with * and + this was 7 VALU per output, and on short-K wide-N
// layers the epilogue issues as many VALU as the K loop — PMC: 2.9 of the 4.4 non-MFMA VALU per MFMA of the per-M GEGLU GEMM).
template <bool PER_M>
__device__ __forceinline__ float dgq_dequant(float acc, float r0, float r1, float r2, float al, float zw, float ga, float vn) {
    if (PER_M) {
        float t = __builtin_fmaf(r0, acc, -(zw * r1));
        t = __builtin_fmaf(r2, vn, t);
        return __builtin_fmaf(al, t, ga);
    }
    return __builtin_fmaf(al, __builtin_fmaf(-zw, r1, acc), ga);
}
