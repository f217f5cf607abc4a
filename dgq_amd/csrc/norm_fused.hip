// GroupNorm statistics for channels-last tensors, emitted as per-(batch, channel) scale/shift so that the
// normalisation (and the SiLU after it) is applied inside dgq_quant_act's load instead of materialising
// GN(x) and SiLU(GN(x)) (QuantResnetBlock2D.forward: norm1 -> SiLU -> conv1, norm2 -> SiLU -> conv2,
// quant/quant_block.py:98-119):   GN(x)[b,c] = x·scale[b,c] + shift[b,c],
//   scale = rstd[b,g]·γ[c],  shift = β[c] − mean[b,g]·rstd[b,g]·γ[c].
// Two launches: partial Welford moments per (b, group, spatial slice), then a merge (Chan) + scale/shift kernel.
#include "dgq_common.h"

struct Moments { float n, mean, m2; };

__device__ __forceinline__ Moments merge(Moments a, Moments b) {
    if (b.n == 0.0f) return a;
    if (a.n == 0.0f) return b;
    Moments r;
    r.n = a.n + b.n;
    const float d = b.mean - a.mean;
    r.mean = a.mean + d * (b.n / r.n);
    r.m2 = a.m2 + b.m2 + d * d * (a.n * b.n / r.n);
    return r;
}

// grid (B*G, S): block handles spatial rows [s*rows_per, ...) of group g of image b
template <typename T>
__global__ __launch_bounds__(256) void gn_partial_kernel(const T* __restrict__ x, int HW, int C, int G, int rows_per,
                                                         float* __restrict__ part) {
    const int bg = blockIdx.x, b = bg / G, g = bg - b * G;
    const int Cg = C / G;
    const int r0 = blockIdx.y * rows_per, r1 = min(HW, r0 + rows_per);
    const T* base = x + ((int64_t)b * HW) * C + g * Cg;
    Moments m = {0.0f, 0.0f, 0.0f};
    const int total = (r1 - r0) * Cg;
    for (int i = threadIdx.x; i < total; i += 256) {
        const int r = i / Cg, c = i - r * Cg;
        const float v = dgq_to_float(base[(int64_t)(r0 + r) * C + c]);
        m.n += 1.0f;
        const float d = v - m.mean;
        m.mean += d / m.n;
        m.m2 += d * (v - m.mean);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        Moments other;
        other.n = __shfl_down(m.n, o, 64);
        other.mean = __shfl_down(m.mean, o, 64);
        other.m2 = __shfl_down(m.m2, o, 64);
        m = merge(m, other);
    }
    __shared__ Moments sm[4];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = merge(merge(sm[0], sm[1]), merge(sm[2], sm[3]));
        float* p = part + ((int64_t)bg * gridDim.y + blockIdx.y) * 3;
        p[0] = m.n; p[1] = m.mean; p[2] = m.m2;
    }
}

// grid (B*G): merges S partials, writes scale/shift for the group's channels
__global__ __launch_bounds__(64) void gn_finalize_kernel(const float* __restrict__ part, int S, int C, int G, float eps,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float* __restrict__ scale, float* __restrict__ shift) {
    const int bg = blockIdx.x, b = bg / G, g = bg - b * G;
    const int Cg = C / G;
    Moments m = {0.0f, 0.0f, 0.0f};
    for (int s = 0; s < S; ++s) {                         // fixed order: deterministic
        const float* p = part + ((int64_t)bg * S + s) * 3;
        Moments o = {p[0], p[1], p[2]};
        m = merge(m, o);
    }
    const float var = m.m2 / m.n;                         // biased variance, as F.group_norm
    const float rstd = rsqrtf(var + eps);
    for (int c = threadIdx.x; c < Cg; c += 64) {
        const int ch = g * Cg + c;
        const float sc = rstd * gamma[ch];
        scale[(int64_t)b * C + ch] = sc;
        shift[(int64_t)b * C + ch] = beta[ch] - m.mean * sc;
    }
}

extern "C" int dgq_groupnorm_scale_shift(const void* x, int x_dtype, int B, int HW, int C, int G, float eps,
                                         const float* gamma, const float* beta, float* scale, float* shift,
                                         float* partial_ws, int slices, void* stream) {
    DGQ_CHECK_ARG(x && gamma && beta && scale && shift && partial_ws, "dgq_groupnorm_scale_shift: null pointer");
    DGQ_CHECK_ARG(B > 0 && HW > 0 && C > 0 && G > 0 && C % G == 0 && slices >= 1 && slices <= 64,
                  "dgq_groupnorm_scale_shift: bad shape");
    hipStream_t st = (hipStream_t)stream;
    const int rows_per = (HW + slices - 1) / slices;
    const int S = (HW + rows_per - 1) / rows_per;
    dim3 grid(B * G, S), block(256);
    switch (x_dtype) {
        case DGQ_F32: hipLaunchKernelGGL(gn_partial_kernel<float>, grid, block, 0, st, (const float*)x, HW, C, G, rows_per, partial_ws); break;
        case DGQ_F16: hipLaunchKernelGGL(gn_partial_kernel<__half>, grid, block, 0, st, (const __half*)x, HW, C, G, rows_per, partial_ws); break;
        case DGQ_BF16: hipLaunchKernelGGL(gn_partial_kernel<__hip_bfloat16>, grid, block, 0, st, (const __hip_bfloat16*)x, HW, C, G, rows_per, partial_ws); break;
        default: dgq_set_error("dgq_groupnorm_scale_shift: unknown dtype %d", x_dtype); return DGQ_EINVAL;
    }
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(B * G), dim3(64), 0, st, partial_ws, S, C, G, eps, gamma, beta, scale, shift);
    return dgq_launch_status("dgq_groupnorm_scale_shift");
}
