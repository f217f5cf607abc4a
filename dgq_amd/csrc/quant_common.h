// Element loads and code packing shared by the quantise-on-load kernels (quant_act.hip) and the GEMM that quantises its own operand
// panel (gemm_panel.hip, FUSE) — one implementation, so that both produce the same codes.
#pragma once
#include "dgq_common.h"

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

// Four codes q_j ∈ [0, 2^b−1] (floats) -> one dword of centred int8 codes s_j = q_j − off, 0 for padding:
// v_cvt_pk_u8_f32 inserts u8(q − off + 128) per byte, and u8(x + 128) ^ 0x80 is the two's-complement byte of x.
// `biased[j]` = valid ? q_j − off + 128 : 128 ; returns the dword, adds Σ biased to `fsum` (exact small integers).
__device__ __forceinline__ uint32_t dgq_pack4(const float (&biased)[4], float& fsum) {
    uint32_t w = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) w = __builtin_amdgcn_cvt_pk_u8_f32(biased[j], j, w);
    fsum += (float)__builtin_amdgcn_sad_u8(w, 0u, 0u);       // Σ of the four bytes (the values are integers in [0, 255]: the bytes ARE the values)
    return w ^ 0x80808080u;
}


// dgq_affine_code_fast (dgq_common.h) for four elements, each with its own (δ, 1/δ, z): the same codes bit for bit, but the IEEE-division
// fallback of the tie band is ONE wave-uniform, rarely taken branch (any lane, any of the four: ~2.5 % of the calls) instead of a
// divergent branch per element — in a loop that quantises dozens of elements per lane the per-element form compiles to two or three
// taken branches per element (~100 cycles each, profiles/r05_small_launch_timeline.txt).
__device__ __forceinline__ void dgq_affine_code4_fast(const float (&x)[4], const float (&d)[4], const float (&inv)[4], const float (&z)[4],
                                                      float qmax, float (&q)[4]) {
    float r[4];
    bool nearj[4];
    bool near = false;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        // the tie band of dgq_affine_code_fast, |t − r| >= 0.5 − |t|·4e-7, as one fma + one compare (|t| >= 1.25e6, infinities and NaN
        // fall inside it by themselves)
        const float t = x[j] * inv[j];
        r[j] = rintf(t);
        nearj[j] = !(__builtin_fmaf(fabsf(t), 4.0e-7f, fabsf(t - r[j])) < 0.5f);
        near = near || nearj[j];
    }
    if (__builtin_amdgcn_ballot_w64(near) != 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float rr = rintf(__fdiv_rn(x[j], d[j]));
            r[j] = nearj[j] ? rr : r[j];
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) q[j] = __builtin_amdgcn_fmed3f(r[j] + z[j], 0.0f, qmax);
}

// ... four elements under ONE (δ, 1/δ, z)
__device__ __forceinline__ void dgq_affine_code4_fast(const float (&x)[4], float d, float inv, float z, float qmax, float (&q)[4]) {
    const float d4[4] = {d, d, d, d}, i4[4] = {inv, inv, inv, inv}, z4[4] = {z, z, z, z};
    dgq_affine_code4_fast(x, d4, i4, z4, qmax, q);
}
