// Attention-side quantizers as HBM-bound elementwise kernels: q/k/v affine fake-quant on the projection
// output, global max (real-time δ) and log2 fake-quant of softmax probabilities.
#include "dgq_common.h"

template <typename T>
__global__ __launch_bounds__(256) void fakequant_rows_kernel(const T* __restrict__ x, T* __restrict__ y, int rows, int C,
                                                             int Ttok, int D, int mode, const float* __restrict__ delta,
                                                             const float* __restrict__ zp, int skip, float qmax) {
    const int64_t total = (int64_t)rows * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int row = (int)(i / C);
        const int col = (int)(i - (int64_t)row * C);
        const int t = row % Ttok;
        float v = dgq_to_float(x[i]);
        if (t >= skip) {
            int idx = mode == 0 ? 0 : (mode == 1 ? t - skip : col % D);
            float d = delta[idx], z = zp[idx];
            float q = dgq_affine_code_fast(v, d, dgq_rcp(d), z, qmax);
            v = d * (q - z);
        }
        y[i] = dgq_from_float<T>(v);
    }
}

__global__ __launch_bounds__(256) void max_f32_kernel(const float* __restrict__ p, int64_t rows, int S, int skip_cols,
                                                      float* __restrict__ out) {
    const int64_t total = rows * S;
    float m = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        int col = (int)(i % S);
        if (col >= skip_cols) m = fmaxf(m, p[i]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_down(m, o, 64));
    __shared__ float sm[4];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
        // values are >= 0, so the int ordering of the bit patterns equals the float ordering
        atomicMax(reinterpret_cast<int*>(out), __float_as_int(m));
    }
}

__global__ __launch_bounds__(256) void logquant_f32_kernel(const float* __restrict__ p, float* __restrict__ y, int64_t rows,
                                                           int S, int skip_cols, const float* __restrict__ delta, float qmax) {
    const int64_t total = rows * S;
    const float d = delta[0];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        int col = (int)(i % S);
        float v = p[i];
        if (col >= skip_cols) {
            // quant_layer_text.py:101-105: -log2(x/δ) -> rne -> clamp [0, 2^b-1] -> 2^-q -> ·δ
            float q = -1.0f * log2f(__fdiv_rn(v, d));
            q = rintf(q);
            q = fminf(fmaxf(q, 0.0f), qmax);
            v = exp2f(-1.0f * q) * d;
        }
        y[i] = v;
    }
}

static inline int grid_for(int64_t total, int block) {
    int64_t g = (total + block - 1) / block;
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    return (int)g;
}

extern "C" int dgq_fakequant_rows(const void* x, void* y, int dtype, int rows, int C, int T, int D,
                                  int mode, const float* delta, const float* zp, int skip, int bits, void* stream) {
    DGQ_CHECK_ARG(x && y && delta && zp, "dgq_fakequant_rows: null pointer");
    DGQ_CHECK_ARG(rows > 0 && C > 0 && T > 0 && D > 0 && mode >= 0 && mode <= 2 && skip >= 0 && bits >= 2 && bits <= 8,
                  "dgq_fakequant_rows: bad argument");
    float qmax = (float)((1 << bits) - 1);
    dim3 grid(grid_for((int64_t)rows * C, 256)), block(256);
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
        case DGQ_F32:
            hipLaunchKernelGGL(fakequant_rows_kernel<float>, grid, block, 0, st, (const float*)x, (float*)y, rows, C, T, D,
                               mode, delta, zp, skip, qmax);
            break;
        case DGQ_F16:
            hipLaunchKernelGGL(fakequant_rows_kernel<__half>, grid, block, 0, st, (const __half*)x, (__half*)y, rows, C, T, D,
                               mode, delta, zp, skip, qmax);
            break;
        case DGQ_BF16:
            hipLaunchKernelGGL(fakequant_rows_kernel<__hip_bfloat16>, grid, block, 0, st, (const __hip_bfloat16*)x,
                               (__hip_bfloat16*)y, rows, C, T, D, mode, delta, zp, skip, qmax);
            break;
        default: dgq_set_error("dgq_fakequant_rows: unknown dtype %d", dtype); return DGQ_EINVAL;
    }
    return dgq_launch_status("dgq_fakequant_rows");
}

extern "C" int dgq_max_f32(const float* p, int64_t rows, int S, int skip_cols, float* out, void* stream) {
    DGQ_CHECK_ARG(p && out && rows > 0 && S > 0 && skip_cols >= 0, "dgq_max_f32: bad argument");
    hipLaunchKernelGGL(max_f32_kernel, dim3(grid_for(rows * S, 256)), dim3(256), 0, (hipStream_t)stream, p, rows, S,
                       skip_cols, out);
    return dgq_launch_status("dgq_max_f32");
}

extern "C" int dgq_logquant_f32(const float* p, float* y, int64_t rows, int S, int skip_cols, const float* delta,
                                int bits, void* stream) {
    DGQ_CHECK_ARG(p && y && delta && rows > 0 && S > 0 && skip_cols >= 0 && bits >= 2 && bits <= 8,
                  "dgq_logquant_f32: bad argument");
    hipLaunchKernelGGL(logquant_f32_kernel, dim3(grid_for(rows * S, 256)), dim3(256), 0, (hipStream_t)stream, p, y, rows,
                       S, skip_cols, delta, (float)((1 << bits) - 1));
    return dgq_launch_status("dgq_logquant_f32");
}
