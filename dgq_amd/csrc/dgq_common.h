// Internal helpers shared by the HIP translation units of libdgq_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include "../../include/dgq_hip.h"

void dgq_set_error(const char* fmt, ...);

#define DGQ_CHECK_ARG(cond, ...)            \
    do {                                    \
        if (!(cond)) {                      \
            dgq_set_error(__VA_ARGS__);     \
            return DGQ_EINVAL;              \
        }                                   \
    } while (0)

static inline int dgq_launch_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        dgq_set_error("%s: %s", what, hipGetErrorString(e));
        return DGQ_ELAUNCH;
    }
    return DGQ_OK;
}

typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float dgq_to_float(float v) { return v; }
__device__ __forceinline__ float dgq_to_float(__half v) { return __half2float(v); }
__device__ __forceinline__ float dgq_to_float(__hip_bfloat16 v) { return __bfloat162float(v); }

template <typename T> __device__ __forceinline__ T dgq_from_float(float v);
template <> __device__ __forceinline__ float dgq_from_float<float>(float v) { return v; }
template <> __device__ __forceinline__ __half dgq_from_float<__half>(float v) { return __float2half(v); }
template <> __device__ __forceinline__ __hip_bfloat16 dgq_from_float<__hip_bfloat16>(float v) { return __float2bfloat16(v); }

// One affine quantisation step exactly as the reference evaluates it in fp32
// (quant_layer.py:297): true division, round-half-even, add zero point, clamp.
__device__ __forceinline__ float dgq_affine_code(float x, float delta, float zp, float qmax) {
    float t = __fdiv_rn(x, delta);
    float r = rintf(t);
    float u = r + zp;
    return fminf(fmaxf(u, 0.0f), qmax);
}

// SiLU x/(1 + e^-x) to ~1 ulp in 13 VALU operations (libm expf + IEEE division: ~30): e^-x = 2^t with t = −x·log2e carried
// as a rounded product plus its exact residual (fma) and the low part of log2e, v_exp_f32 on the rounded part, first-
// order correction for the rest; 1/(1 + e) by v_rcp_f32 + one Newton step.  t is capped at 126 so that 1 + e stays
// finite (x < −87: the result is −0 … −1e-36 either way).
__device__ __forceinline__ float dgq_silu(float x) {
    const float L_HI = -1.44269502162933349609375f, L_LO = -1.925963033500011e-8f;   // −log2(e) = L_HI + L_LO
    float t = x * L_HI;
    float r = fmaf(x, L_HI, -t) + x * L_LO;
    t = fminf(t, 126.0f);
    float e = __builtin_amdgcn_exp2f(t);
    e = fmaf(e, r * 0.693147180559945f, e);
    const float den = 1.0f + e;
    float q = __builtin_amdgcn_rcpf(den);
    q = fmaf(fmaf(-den, q, 1.0f), q, q);
    return x * q;
}

// The same code, bit for bit, at the cost of a multiply: with inv = v_rcp_f32(δ) (1 ulp), t = fl(x·inv) differs from the
// real quotient q by < |q|·1.8e-7, and the correctly rounded fl(q) by < |q|·0.6e-7; so whenever t is farther than
// |t|·4e-7 from every half-integer, rint(t) == rint(fl(x/δ)).  Only values inside that band (probability ~1e-4 per
// element), or too large for the argument, take the IEEE division.
__device__ __forceinline__ float dgq_rcp(float d) { return __builtin_amdgcn_rcpf(d); }
__device__ __forceinline__ float dgq_affine_code_fast(float x, float delta, float inv_delta, float zp, float qmax) {
    const float t = x * inv_delta;
    float r = rintf(t);
    const float dist = fabsf(fabsf(t - r) - 0.5f);
    if (dist <= fabsf(t) * 4.0e-7f + 1e-30f || !(fabsf(t) < 3.0e6f)) r = rintf(__fdiv_rn(x, delta));
    return __builtin_amdgcn_fmed3f(r + zp, 0.0f, qmax);
}
