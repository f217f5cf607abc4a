// Calibration-time statistics of the DGQ activation-calibration producer: the per-"in-channel" and
// per-"out-channel" minima / maxima UniformAffineQuantizer.record_min_max_ema collects for every activation quantizer
// and every calibration batch (quant/quant_layer.py:301-313): for a tensor viewed as [rows][C] these are the column-wise
// and the row-wise min / max (the caller folds batch / head indices, which only repeat rows or columns).
// One pass over x per statistic, HBM-bound, fp32 outputs; min / max are exact, so the results do not depend on the
// reduction order.
#include <cfloat>
#include "dgq_common.h"

// row statistics: one wave per row, 4 elements per lane per step (rows are contiguous: ldx == C or 16-byte friendly)
template <typename T>
__global__ __launch_bounds__(256) void row_minmax_kernel(const T* __restrict__ x, int rows, int C, int64_t ldx,
                                                         float* __restrict__ rmin, float* __restrict__ rmax) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const T* xr = x + (int64_t)row * ldx;
    float lo = FLT_MAX, hi = -FLT_MAX;
    for (int c = lane; c < C; c += 64) {
        const float v = dgq_to_float(xr[c]);
        lo = fminf(lo, v);
        hi = fmaxf(hi, v);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, o, 64));
        hi = fmaxf(hi, __shfl_xor(hi, o, 64));
    }
    if (lane == 0) {
        rmin[row] = lo;
        rmax[row] = hi;
    }
}

// column statistics, stage 1: thread = one column, block.y = a slice of the rows (coalesced across the 256 columns of a block)
template <typename T>
__global__ __launch_bounds__(256) void col_minmax_partial_kernel(const T* __restrict__ x, int rows, int C, int64_t ldx,
                                                                 int rows_per_slice, float* __restrict__ part) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const int r0 = blockIdx.y * rows_per_slice, r1 = min(rows, r0 + rows_per_slice);
    float lo = FLT_MAX, hi = -FLT_MAX;
    for (int r = r0; r < r1; ++r) {
        const float v = dgq_to_float(x[(int64_t)r * ldx + c]);
        lo = fminf(lo, v);
        hi = fmaxf(hi, v);
    }
    part[((int64_t)blockIdx.y * 2 + 0) * C + c] = lo;
    part[((int64_t)blockIdx.y * 2 + 1) * C + c] = hi;
}

__global__ __launch_bounds__(256) void col_minmax_final_kernel(const float* __restrict__ part, int slices, int C,
                                                               float* __restrict__ cmin, float* __restrict__ cmax) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float lo = FLT_MAX, hi = -FLT_MAX;
    for (int s = 0; s < slices; ++s) {
        lo = fminf(lo, part[((int64_t)s * 2 + 0) * C + c]);
        hi = fmaxf(hi, part[((int64_t)s * 2 + 1) * C + c]);
    }
    cmin[c] = lo;
    cmax[c] = hi;
}

template <typename T>
static void launch_minmax(const T* x, int rows, int C, int64_t ldx, float* rmin, float* rmax, float* cmin, float* cmax,
                          float* part, int slices, hipStream_t st) {
    if (rmin)
        hipLaunchKernelGGL((row_minmax_kernel<T>), dim3((rows + 3) / 4), dim3(256), 0, st, x, rows, C, ldx, rmin, rmax);
    if (cmin) {
        const int rps = (rows + slices - 1) / slices;
        const int used = (rows + rps - 1) / rps;
        hipLaunchKernelGGL((col_minmax_partial_kernel<T>), dim3((C + 255) / 256, used), dim3(256), 0, st, x, rows, C, ldx, rps, part);
        hipLaunchKernelGGL(col_minmax_final_kernel, dim3((C + 255) / 256), dim3(256), 0, st, part, used, C, cmin, cmax);
    }
}

extern "C" int dgq_minmax_rows_cols(const void* x, int dtype, int rows, int C, int64_t ldx,
                                    float* rowmin, float* rowmax, float* colmin, float* colmax,
                                    float* partial_ws, int slices, void* stream) {
    DGQ_CHECK_ARG(x && rows > 0 && C > 0 && ldx >= C, "dgq_minmax_rows_cols: bad shape rows=%d C=%d", rows, C);
    DGQ_CHECK_ARG((rowmin == nullptr) == (rowmax == nullptr) && (colmin == nullptr) == (colmax == nullptr) && (rowmin || colmin),
                  "dgq_minmax_rows_cols: outputs come in (min, max) pairs");
    DGQ_CHECK_ARG(!colmin || (partial_ws && slices >= 1 && slices <= 4096), "dgq_minmax_rows_cols: column statistics need partial_ws [2*slices*C]");
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
        case DGQ_F32: launch_minmax((const float*)x, rows, C, ldx, rowmin, rowmax, colmin, colmax, partial_ws, slices, st); break;
        case DGQ_F16: launch_minmax((const __half*)x, rows, C, ldx, rowmin, rowmax, colmin, colmax, partial_ws, slices, st); break;
        case DGQ_BF16: launch_minmax((const __hip_bfloat16*)x, rows, C, ldx, rowmin, rowmax, colmin, colmax, partial_ws, slices, st); break;
        default: dgq_set_error("dgq_minmax_rows_cols: unknown dtype %d", dtype); return DGQ_EINVAL;
    }
    return dgq_launch_status("dgq_minmax_rows_cols");
}
