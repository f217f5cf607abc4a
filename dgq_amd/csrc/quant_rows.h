// Row-level building blocks of the quantise-on-load kernels (quant_act.hip) that the row-owning GEMM (gemm_wxa8.hip: a workgroup
// that has just stored whole rows of its output quantises them for the layer that consumes them) runs as well: the parameter
// block, the exact-rounding quantiser's helpers, the LayerNorm statistics of a row, and the two one-wave-per-row quantisers of a
// Linear input — natural order (scalar / per-token tables) and LDS scatter (per-K tables).  Same arithmetic in the same order
// as the stand-alone kernels: codes AND row sums are bit-identical (tests/test_gpu_kernels.py).
#pragma once
#include "dgq_common.h"

struct QuantActParams {
    const void* x;
    int B, H, W, C, kh, kw, stride, pad, Ho, Wo;
    const int32_t* ksrc;      // [Kp] (dh<<24 | dw<<16 | c) or -1, or NULL (natural order kp = tap*C + c)
    const int32_t* koff;      // optional [Kp]: (dh*W + dw)*ldc + c for THIS geometry, -1 for padding (interior rows)
    const int32_t* klds;      // optional [Kp]: (dh*kw + dw)*C + c, -1 for padding (LDS-staged conv path)
    const int32_t* kdst;      // optional [taps*C]: packed position kp of element (tap, c) — the inverse of ksrc (scatter path)
    const int32_t* kpat;      // optional [Kp]: (dh*PW + dw)*C + c inside the input patch of a conv tile (PW of dgq_quant_act_conv_tile), -1 padding
    int Kp, K;
    const float* delta;       // per_m: [L]; else [Kp/32]
    const float* zp;
    int L;
    float qmax, offset;
    int8_t* codes;
    float* rowsum;            // [ksplits][M] partial sums (the GEMM epilogue adds them in a fixed order)
    int M;
    int kp_per_split;         // multiple of 256
    const float* pre_scale;   // optional [B][C]: v = x*scale + shift (fused GroupNorm), then pre_act
    const float* pre_shift;
    int pre_act;              // 0 none, 1 SiLU, 2 GEGLU: value = x[c]·gelu(x[C + c]) on rows of 2C elements
    int ldc;                  // elements per input pixel/row (C, or 2C for GEGLU)
    const float* ln_gamma;    // optional [C]: LayerNorm over the C elements of the row, v = (x − μ)·rstd·γ + β (1x1 only)
    const float* ln_beta;
    float ln_eps;
};

// Up to DGQ_QA_BATCH problems of ONE kernel variant and the same row count in one launch (blockIdx.z = problem): the
// q / k / v projections of an attention quantise the same input three ways, the to_k / to_v of every cross-attention
// quantise the same text context — one launch each instead of one per layer.
#define DGQ_QA_BATCH 8
struct QuantActBatch {
    QuantActParams p[DGQ_QA_BATCH];
};



template <typename TIn>
__device__ __forceinline__ void load4(const TIn* p, float (&v)[4]);
template <>
__device__ __forceinline__ void load4<float>(const float* p, float (&v)[4]) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
template <>
__device__ __forceinline__ void load4<__half>(const __half* p, float (&v)[4]) {
    const uint2 t = *reinterpret_cast<const uint2*>(p);
    const __half* h = reinterpret_cast<const __half*>(&t);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = __half2float(h[j]);
}
template <>
__device__ __forceinline__ void load4<__hip_bfloat16>(const __hip_bfloat16* p, float (&v)[4]) {
    const uint2 t = *reinterpret_cast<const uint2*>(p);
    const uint16_t* h = reinterpret_cast<const uint16_t*>(&t);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = __uint_as_float(((uint32_t)h[j]) << 16);
}

// LayerNorm statistics of one row of C <= 2048 elements (C % 4 == 0), computed by the wave that quantises the row: the
// row is read ONCE into registers (8 float4 per lane), mean first, then Σ(x − mean)² from the registers; biased
// variance, rstd = 1/sqrt(var + eps) as nn.LayerNorm.
#define DGQ_LN_MAX_C 2048
template <typename TIn>
__device__ __forceinline__ void row_layernorm_stats(const TIn* xr, int C, float eps, int lane, float& mu, float& rstd) {
    float v[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = lane * 4 + 256 * i;
        if (c < C) load4<TIn>(xr + c, v[i]);
        else v[i][0] = v[i][1] = v[i][2] = v[i][3] = 0.0f;
    }
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    mu = s / (float)C;
    float q = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = lane * 4 + 256 * i;
        if (c < C) {
#pragma unroll
            for (int j = 0; j < 4; ++j) q += (v[i][j] - mu) * (v[i][j] - mu);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    rstd = 1.0f / sqrtf(q / (float)C + eps);
}

// Four codes q_j ∈ [0, 2^b−1] (floats) -> one dword of centred int8 codes s_j = q_j − off, 0 for padding:
// v_cvt_pk_u8_f32 inserts u8(q − off + 128) per byte, and u8(x + 128) ^ 0x80 is the two's-complement byte of x.
// `biased[j]` = valid ? q_j − off + 128 : 128 ; returns the dword, adds Σ biased to `fsum` (exact small integers).
__device__ __forceinline__ uint32_t dgq_pack4(const float (&biased)[4], float& fsum) {
    uint32_t w = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        w = __builtin_amdgcn_cvt_pk_u8_f32(biased[j], j, w);
        fsum += biased[j];
    }
    return w ^ 0x80808080u;
}


// argument check + parameter block of one dgq_quant_act problem (quant_act.hip)
int dgq_fill_quant_act(const dgq_quant_act_args_t& a, QuantActParams& p);

// ---- one wave, one row of a Linear input (kh = kw = 1), natural K order (no table): quant_act_kernel<TIn, false, PER_M>'s row,
// unsplit (the whole K range).  Writes the row's codes and its row sum.
template <typename TIn, bool PER_M>
__device__ __forceinline__ void qa_row_natural(const QuantActParams& p, int row, int lane) {
    const TIn* xr = reinterpret_cast<const TIn*>(p.x) + (int64_t)row * p.ldc;
    float md = 1.0f, mz = 0.0f, minv = 1.0f;
    if (PER_M) {
        const int li = row % p.L;
        md = p.delta[li];
        mz = p.zp[li];
        minv = dgq_rcp(md);
    }
    float partial = 0.0f;
    float ln_mu = 0.0f, ln_rstd = 1.0f;
    if (p.ln_gamma) row_layernorm_stats<TIn>(xr, p.C, p.ln_eps, lane, ln_mu, ln_rstd);
    uint32_t* out = reinterpret_cast<uint32_t*>(p.codes + (int64_t)row * p.Kp);
    const float bias = 128.0f - p.offset;
#pragma unroll 2
    for (int kp0 = lane * 4; kp0 < p.Kp; kp0 += 256) {
        float d = md, z = mz, inv = minv;
        if (!PER_M) {
            d = p.delta[kp0 >> 5];
            z = p.zp[kp0 >> 5];
            inv = dgq_rcp(d);
        }
        float v[4];
        const bool in_k = kp0 < p.K;
        if (in_k) {
            load4<TIn>(xr + kp0, v);
            if (p.ln_gamma) {
                const float4 ga = *reinterpret_cast<const float4*>(p.ln_gamma + kp0);
                const float4 be = *reinterpret_cast<const float4*>(p.ln_beta + kp0);
                v[0] = (v[0] - ln_mu) * ln_rstd * ga.x + be.x; v[1] = (v[1] - ln_mu) * ln_rstd * ga.y + be.y;
                v[2] = (v[2] - ln_mu) * ln_rstd * ga.z + be.z; v[3] = (v[3] - ln_mu) * ln_rstd * ga.w + be.w;
            }
            if (p.pre_act == 1) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = dgq_silu(v[j]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = 0.0f;
        }
        float biased[4], fsum = 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float q = dgq_affine_code_fast(v[j], d, inv, z, p.qmax);
            biased[j] = in_k ? q + bias : 128.0f;
        }
        out[kp0 >> 2] = dgq_pack4(biased, fsum);
        fsum -= 512.0f;
        partial += PER_M ? fsum : d * fsum;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) partial += __shfl_down(partial, o, 64);
    if (lane == 0) p.rowsum[row] = partial;
}

// ---- one wave, one row of a Linear input, per-K table: quant_act_scatter_kernel<TIn, 4>'s row.  tdelta / tinv / tzp: the
// per-chunk tables staged in LDS by the caller; image: Kp bytes of LDS owned by this wave (any content: zeroed here).
template <typename TIn>
__device__ __forceinline__ void qa_row_scatter(const QuantActParams& p, int row, int lane, const float* tdelta, const float* tinv,
                                               const float* tzp, uint8_t* image) {
    for (int i = lane; i < (p.Kp >> 4); i += 64) reinterpret_cast<uint4*>(image)[i] = make_uint4(0, 0, 0, 0);   // padding = code 0
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    const TIn* xr = reinterpret_cast<const TIn*>(p.x) + (int64_t)row * p.ldc;
    float partial = 0.0f;
    float ln_mu = 0.0f, ln_rstd = 1.0f;
    if (p.ln_gamma) row_layernorm_stats<TIn>(xr, p.C, p.ln_eps, lane, ln_mu, ln_rstd);
    for (int c = lane * 4; c < p.C; c += 256) {
        float v[4];
        load4<TIn>(xr + c, v);
        if (p.ln_gamma) {
            const float4 ga = *reinterpret_cast<const float4*>(p.ln_gamma + c);
            const float4 be = *reinterpret_cast<const float4*>(p.ln_beta + c);
            v[0] = (v[0] - ln_mu) * ln_rstd * ga.x + be.x; v[1] = (v[1] - ln_mu) * ln_rstd * ga.y + be.y;
            v[2] = (v[2] - ln_mu) * ln_rstd * ga.z + be.z; v[3] = (v[3] - ln_mu) * ln_rstd * ga.w + be.w;
        }
        if (p.pre_act == 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = dgq_silu(v[j]);
        }
        const int4 d4 = *reinterpret_cast<const int4*>(p.kdst + c);
        const int dst[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ch = dst[j] >> 5;
            const float d = tdelta[ch];
            const float sc = dgq_affine_code_fast(v[j], d, tinv[ch], tzp[ch], p.qmax) - p.offset;
            image[dst[j]] = (uint8_t)(int)sc;
            partial += d * sc;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) partial += __shfl_down(partial, o, 64);
    // the wave's own byte stores precede its reads of the image in LDS order; the fence keeps the compiler from reordering them
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    uint4* out = reinterpret_cast<uint4*>(p.codes + (int64_t)row * p.Kp);
    const uint4* im = reinterpret_cast<const uint4*>(image);
    for (int i = lane; i < (p.Kp >> 4); i += 64) out[i] = im[i];
    if (lane == 0) p.rowsum[row] = partial;
}
