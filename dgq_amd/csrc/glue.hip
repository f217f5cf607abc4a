// The elementwise glue of a denoise step as single launches (each replaces a chain of eager / captured torch kernels that cost a
// dependent launch apiece, profiles/r05_step_dispatch_trace.tsv):
//   dgq_timestep_embedding : Timesteps.forward (diffusers_rewrite/sd.py get_timestep_embedding with flip_sin_to_cos=True,
//                            downscale_freq_shift=0): arange, mul, div, exp, mul, cos, sin, cat -> one launch
//   dgq_cfg_ddim_step      : classifier-free guidance + the DDIM update (pipeline_stable_diffusion.py:1037-1044, scheduling_ddim.py
//                            step with eta = 0): chunk, sub, mul, add, mul, sub, div, mul, mul, add -> one launch
// Both follow the operation ORDER of the torch formulation in fp32 (the build has -ffp-contract=off), including torch's division by a
// host scalar as a multiplication by its fp32 reciprocal, so the results are those of the eager chain bit for bit where the
// transcendental functions are the same ocml ones (tests/test_gpu_kernels.py).
#include <math.h>
#include "dgq_common.h"

template <typename TT, typename TOut>
__global__ __launch_bounds__(256) void timestep_embedding_kernel(const TT* __restrict__ t, int64_t t_stride, int rows, int half, float coef,
                                                                 float inv_half, TOut* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * half) return;
    const int r = i / half, j = i - r * half;
    // freqs = exp(-ln(10000) * arange(half) / half): (j * coef) * fl(1 / half), as torch evaluates tensor * scalar / scalar
    const float f = expf(((float)j * coef) * inv_half);
    const float a = (float)t[(int64_t)r * t_stride] * f;
    out[(int64_t)r * 2 * half + j] = dgq_from_float<TOut>(cosf(a));
    out[(int64_t)r * 2 * half + half + j] = dgq_from_float<TOut>(sinf(a));
}

extern "C" int dgq_timestep_embedding(const void* t, int t_is_float, int64_t t_stride, int rows, int dim, void* out, int out_dtype,
                                      void* stream) {
    DGQ_CHECK_ARG(t && out && rows > 0 && dim > 0 && dim % 2 == 0, "dgq_timestep_embedding: null pointer or bad shape (dim even)");
    const int half = dim / 2;
    const float coef = (float)(-9.210340371976184);      // -math.log(10000), rounded to fp32 as torch rounds the python scalar
    const float inv_half = 1.0f / (float)half;
    const dim3 grid((rows * half + 255) / 256), block(256);
    hipStream_t st = (hipStream_t)stream;
#define DGQ_TE(TT, TO) hipLaunchKernelGGL((timestep_embedding_kernel<TT, TO>), grid, block, 0, st, (const TT*)t, t_stride, rows, half, coef, inv_half, (TO*)out)
    if (t_is_float) {
        switch (out_dtype) {
            case DGQ_F32: DGQ_TE(float, float); break;
            case DGQ_F16: DGQ_TE(float, __half); break;
            case DGQ_BF16: DGQ_TE(float, __hip_bfloat16); break;
            default: dgq_set_error("dgq_timestep_embedding: unknown dtype %d", out_dtype); return DGQ_EINVAL;
        }
    } else {
        switch (out_dtype) {
            case DGQ_F32: DGQ_TE(int64_t, float); break;
            case DGQ_F16: DGQ_TE(int64_t, __half); break;
            case DGQ_BF16: DGQ_TE(int64_t, __hip_bfloat16); break;
            default: dgq_set_error("dgq_timestep_embedding: unknown dtype %d", out_dtype); return DGQ_EINVAL;
        }
    }
#undef DGQ_TE
    return dgq_launch_status("dgq_timestep_embedding");
}

// eps = e_u + g·(e_c − e_u);  x' = s3·((x − s1·eps)·inv_s2) + s4·eps   with s1 = √(1−ᾱ_t), inv_s2 = 1/√ᾱ_t, s3 = √ᾱ_prev, s4 = √(1−ᾱ_prev).
// sample / out: [n][C][HW] contiguous (the pipeline's latents); the two eps halves: element (n, c, p) at n·C·HW + c·ec + p·ep — ec = HW,
// ep = 1 for the same layout, ec = 1, ep = C for the channels-last tensor the UNet returns.  e_c == nullptr: eps = e_u (no guidance).
// 16-bit tensors: every statement of the eager chain rounds its result to the tensor's type (torch evaluates each in fp32 and stores
// bf16 / fp16) — RT(x) below; fp32: the identity.
template <typename T>
__global__ __launch_bounds__(256) void cfg_ddim_step_kernel(const T* __restrict__ e_u, const T* __restrict__ e_c, const T* __restrict__ x,
                                                            T* __restrict__ out, int64_t n, int C, int HW, int64_t ec, int64_t ep, float g,
                                                            float s1, float inv_s2, float s3, float s4) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t img = i / ((int64_t)C * HW), r = i - img * C * HW;
    const int c = (int)(r / HW), p = (int)(r - (int64_t)c * HW);
    const int64_t ei = img * C * HW + c * ec + p * ep;
    // (the empty asm keeps each fp32 result in a register before its conversion: for fp16 the compiler otherwise folds
    // fptrunc(fmul) into v_fma_mixlo_f16, whose result differed from the fp32-then-fp16 rounding of the eager kernels in ~1e-3 of the elements)
    auto RT = [](float v) {
        if (sizeof(T) == 4) return v;
        asm volatile("" : "+v"(v));
        return dgq_to_float(dgq_from_float<T>(v));
    };
    float e = dgq_to_float(e_u[ei]);
    if (e_c) e = RT(e + RT(g * RT(dgq_to_float(e_c[ei]) - e)));
    const float p0 = RT(RT(dgq_to_float(x[i]) - RT(s1 * e)) * inv_s2);
    float o = RT(s3 * p0) + RT(s4 * e);
    if (sizeof(T) != 4) asm volatile("" : "+v"(o));
    out[i] = dgq_from_float<T>(o);
}

extern "C" int dgq_cfg_ddim_step(const void* eps_uncond, const void* eps_cond, const void* sample, void* out, int dtype, int64_t n, int C, int HW,
                                 int eps_channels_last, float guidance, float s1, float inv_s2, float s3, float s4, void* stream) {
    DGQ_CHECK_ARG(eps_uncond && sample && out && n > 0 && C > 0 && HW > 0 && n % ((int64_t)C * HW) == 0,
                  "dgq_cfg_ddim_step: null pointer or n not a whole number of [C][HW] images");
    const int64_t ec = eps_channels_last ? 1 : HW, ep = eps_channels_last ? C : 1;
#define DGQ_CFG(TT) hipLaunchKernelGGL(cfg_ddim_step_kernel<TT>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const TT*)eps_uncond, \
                                       (const TT*)eps_cond, (const TT*)sample, (TT*)out, n, C, HW, ec, ep, guidance, s1, inv_s2, s3, s4)
    switch (dtype) {
        case DGQ_F32: DGQ_CFG(float); break;
        case DGQ_F16: DGQ_CFG(__half); break;
        case DGQ_BF16: DGQ_CFG(__hip_bfloat16); break;
        default: dgq_set_error("dgq_cfg_ddim_step: unknown dtype %d", dtype); return DGQ_EINVAL;
    }
#undef DGQ_CFG
    return dgq_launch_status("dgq_cfg_ddim_step");
}
