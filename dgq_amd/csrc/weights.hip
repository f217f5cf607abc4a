// Load-time weight kernels: integer codes, int4/int8 packing in the GEMM's K order, exact unpack.
#include "dgq_common.h"

// codes[n][k] = clamp(rne(w/δ_n) + z_n, 0, 2^b-1)                (quant_layer.py:295-299)
//            or clamp(floor(w/δ_n) + (α>=0) + z_n, 0, 2^b-1)     (adaptive_rounding.py:51,58-70)
__global__ void quantize_weight_kernel(const float* __restrict__ w, const float* __restrict__ delta,
                                       const float* __restrict__ zp, const float* __restrict__ alpha,
                                       int N, int K, float qmax, uint8_t* __restrict__ codes) {
    int64_t total = (int64_t)N * K;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        int n = (int)(i / K);
        float d = delta[n], z = zp[n];
        float t = __fdiv_rn(w[i], d);
        float r = alpha ? (floorf(t) + (alpha[i] >= 0.0f ? 1.0f : 0.0f)) : rintf(t);
        float q = fminf(fmaxf(r + z, 0.0f), qmax);
        codes[i] = (uint8_t)q;
    }
}

// layout 1 (dgq_gemm_wxa8): rows with (n & 16) keep the two 8-byte halves of every 16 packed bytes exchanged — word w of a
// row sits at w ^ 2 (see include/dgq_hip.h)
// layout 2 (gemm_panel.hip): fragment-major — word w of row n goes to block (n / 32, chunk pair), lane (K half << 5) | (n & 31),
// word (chunk & 1)·2 + (w & 1) of the lane's 16 bytes
__device__ __forceinline__ int64_t w4_word_index(int64_t i, int wpr, int layout) {
    if (layout == 1 && (((i / wpr) >> 4) & 1)) return i ^ 2;
    if (layout == 2) {
        const int64_t n = i / wpr;
        const int w = (int)(i - n * wpr);
        const int c = w >> 2, pairs = wpr >> 3;
        return ((n >> 5) * pairs + (c >> 1)) * 256 + ((((w & 3) >> 1) * 32 + (int)(n & 31)) * 4 + (c & 1) * 2 + (w & 1));
    }
    return i;
}

// one thread per packed 32-bit word = 8 consecutive kp
__global__ void pack_w4_kernel(const uint8_t* __restrict__ codes, int N, int K, const int32_t* __restrict__ kperm,
                               int Kp, int layout, uint32_t* __restrict__ packed) {
    int wpr = Kp / 8;
    const int rows = layout == 2 ? (N + 31) / 32 * 32 : N;       // layout 2 pads N to whole 32-column tiles (zero rows)
    int64_t total = (int64_t)rows * wpr;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        int n = (int)(i / wpr);
        int kp0 = (int)(i % wpr) * 8;
        uint32_t word = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int kp = kp0 + j;
            int k = kperm ? kperm[kp] : (kp < K ? kp : -1);
            uint32_t c = (k >= 0 && n < N) ? (codes[(int64_t)n * K + k] & 0xF) : 0u;
            int byte = j & 3, hi = j >> 2;
            word |= c << (8 * byte + 4 * hi);
        }
        packed[w4_word_index(i, wpr, layout)] = word;
    }
}

__global__ void unpack_w4_kernel(const uint32_t* __restrict__ packed, int N, int Kp, int layout, uint8_t* __restrict__ out) {
    int wpr = Kp / 8;
    int64_t total = (int64_t)N * wpr;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t word = packed[w4_word_index(i, wpr, layout)];
        uint32_t lo = word & 0x0F0F0F0Fu, hi = (word >> 4) & 0x0F0F0F0Fu;   // the GEMM's unpack
        uint2 o = make_uint2(lo, hi);
        *reinterpret_cast<uint2*>(out + i * 8) = o;
    }
}

__global__ void pack_w8_kernel(const uint8_t* __restrict__ codes, int N, int K, const int32_t* __restrict__ kperm,
                               int Kp, int8_t* __restrict__ packed) {
    int64_t total = (int64_t)N * Kp;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        int n = (int)(i / Kp);
        int kp = (int)(i % Kp);
        int k = kperm ? kperm[kp] : (kp < K ? kp : -1);
        packed[i] = (k >= 0) ? (int8_t)((int)codes[(int64_t)n * K + k] - 128) : (int8_t)0;
    }
}

static inline int grid_for(int64_t total, int block) {
    int64_t g = (total + block - 1) / block;
    if (g > 8192) g = 8192;
    if (g < 1) g = 1;
    return (int)g;
}

extern "C" int dgq_quantize_weight(const float* w, const float* delta, const float* zp, const float* alpha,
                                   int N, int K, int bits, uint8_t* codes, void* stream) {
    DGQ_CHECK_ARG(w && delta && zp && codes, "dgq_quantize_weight: null pointer");
    DGQ_CHECK_ARG(N > 0 && K > 0 && bits >= 2 && bits <= 8, "dgq_quantize_weight: bad N=%d K=%d bits=%d", N, K, bits);
    int64_t total = (int64_t)N * K;
    hipLaunchKernelGGL(quantize_weight_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       w, delta, zp, alpha, N, K, (float)((1 << bits) - 1), codes);
    return dgq_launch_status("dgq_quantize_weight");
}

extern "C" int dgq_pack_w4(const uint8_t* codes, int N, int K, const int32_t* kperm, int Kp, int layout, uint8_t* packed,
                           void* stream) {
    DGQ_CHECK_ARG(codes && packed, "dgq_pack_w4: null pointer");
    DGQ_CHECK_ARG(layout >= 0 && layout <= 2, "dgq_pack_w4: layout=%d", layout);
    DGQ_CHECK_ARG(N > 0 && K > 0 && Kp > 0 && Kp % DGQ_KTILE == 0, "dgq_pack_w4: Kp=%d must be a multiple of %d", Kp, DGQ_KTILE);
    DGQ_CHECK_ARG(kperm || Kp >= K, "dgq_pack_w4: identity order needs Kp >= K");
    hipLaunchKernelGGL(pack_w4_kernel, dim3(grid_for((int64_t)((N + 31) / 32 * 32) * Kp / 8, 256)), dim3(256), 0, (hipStream_t)stream,
                       codes, N, K, kperm, Kp, layout, reinterpret_cast<uint32_t*>(packed));
    return dgq_launch_status("dgq_pack_w4");
}

extern "C" int dgq_unpack_w4(const uint8_t* packed, int N, int Kp, int layout, uint8_t* out, void* stream) {
    DGQ_CHECK_ARG(packed && out, "dgq_unpack_w4: null pointer");
    DGQ_CHECK_ARG(N > 0 && Kp > 0 && Kp % 8 == 0 && (layout == 0 || (layout == 1 && Kp % 32 == 0) || (layout == 2 && Kp % 64 == 0)),
                  "dgq_unpack_w4: bad shape / layout");
    hipLaunchKernelGGL(unpack_w4_kernel, dim3(grid_for((int64_t)N * Kp / 8, 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const uint32_t*>(packed), N, Kp, layout, out);
    return dgq_launch_status("dgq_unpack_w4");
}

extern "C" int dgq_pack_w8(const uint8_t* codes, int N, int K, const int32_t* kperm, int Kp, int8_t* packed,
                           void* stream) {
    DGQ_CHECK_ARG(codes && packed, "dgq_pack_w8: null pointer");
    DGQ_CHECK_ARG(N > 0 && K > 0 && Kp > 0 && Kp % DGQ_KTILE == 0, "dgq_pack_w8: Kp=%d must be a multiple of %d", Kp, DGQ_KTILE);
    DGQ_CHECK_ARG(kperm || Kp >= K, "dgq_pack_w8: identity order needs Kp >= K");
    hipLaunchKernelGGL(pack_w8_kernel, dim3(grid_for((int64_t)N * Kp, 256)), dim3(256), 0, (hipStream_t)stream,
                       codes, N, K, kperm, Kp, packed);
    return dgq_launch_status("dgq_pack_w8");
}
