// Weight-only state (use_wq = True, use_aq = False: BASELINE config 1, and the reference's --use_aq-less runs): the layer is
// y = x · ŵᵀ + b with UNQUANTISED activations and the dequantised weight ŵ = δw·(qw − zw) (quant_layer.py:642-659 feeds
// that to F.linear / F.conv2d).  Integer MFMA does not apply — the activations are fp32 — so this is an exact-fp32 GEMM on
// V_MFMA_F32_32X32X2_F32 (each product and add as in an fmaf chain) with the convolution's im2col folded into the A-tile
// load: rows = output positions, K = tap·C + c (natural order), out-of-image taps read 0 like F.conv2d's zero padding.
// Plumbing-grade tiling (64x64 block tile, 16-deep K tiles through LDS, no ring): this state is not on the timed path; it
// exists so that no state of a quantized model leaves this library on the GPU.
#include "quant_common.h"

typedef float v16f __attribute__((ext_vector_type(16)));

struct ConvF32Params {
    const void* x;       // [B][H][W][C] channels-last, x_dtype
    const float* w;      // [N][K] fp32, K = kh·kw·C in (tap, c) order
    const float* bias;   // [N] or nullptr
    const float* pre_scale;   // optional [B][C]: x·scale + shift (a folded GroupNorm) ...
    const float* pre_shift;
    int pre_act;              // ... followed by SiLU when 1; out-of-image taps stay 0 (the conv pads AFTER norm + activation)
    void* y;             // [M][ldy], y_dtype
    void* y2;            // (or nullptr) a second copy of the rows at pitch ldy2 (the output's slot in a concatenation buffer)
    int ldy2;
    float* gn_partial;   // (or nullptr; M % 16 == 0) [M/16][N][2] = (mean, M2) per 16-row block and column of the values as stored: dgq_gemm_extra_t.gn_partial
    int x_dtype, y_dtype;
    int B, H, W, C, kh, kw, stride, pad, Ho, Wo, N, K, M, ldy;
};

__device__ __forceinline__ float ld_any(const void* p, int dtype, int64_t i);
// the activation the convolution sees: optional per-(b, c) affine (GroupNorm) and SiLU applied on load
__device__ __forceinline__ float ld_pre(const ConvF32Params& p, int b, int64_t pix, int c) {
    float v = ld_any(p.x, p.x_dtype, pix * p.C + c);
    if (p.pre_scale) v = v * p.pre_scale[(int64_t)b * p.C + c] + p.pre_shift[(int64_t)b * p.C + c];
    if (p.pre_act == 1) v = dgq_silu(v);
    return v;
}

__device__ __forceinline__ void store_any(void* p, int dtype, int64_t i, float v) {
    if (dtype == DGQ_F16) reinterpret_cast<__half*>(p)[i] = __float2half(v);
    else if (dtype == DGQ_BF16) reinterpret_cast<__hip_bfloat16*>(p)[i] = __float2bfloat16(v);
    else reinterpret_cast<float*>(p)[i] = v;
}
__device__ __forceinline__ float ld_any(const void* p, int dtype, int64_t i) {
    if (dtype == DGQ_F16) return __half2float(reinterpret_cast<const __half*>(p)[i]);
    if (dtype == DGQ_BF16) return __bfloat162float(reinterpret_cast<const __hip_bfloat16*>(p)[i]);
    return reinterpret_cast<const float*>(p)[i];
}

#define CF_BM 64
#define CF_BN 64
#define CF_BK 16
#define CF_LD (CF_BK + 1)

__global__ __launch_bounds__(256) void conv_f32w_kernel(ConvF32Params p) {
    __shared__ float As[CF_BM][CF_LD];
    __shared__ float Bs[CF_BN][CF_LD];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int m0 = blockIdx.y * CF_BM, n0 = blockIdx.x * CF_BN;
    const int wm = (wid >> 1) * 32, wn = (wid & 1) * 32;     // 2 x 2 waves, 32 x 32 each
    // staging role of this thread: row tid / 4 of the tile, four consecutive k
    const int srow = tid >> 2, sk = (tid & 3) * 4;
    const int am = m0 + srow;
    int ab = 0, aho = 0, awo = 0;
    if (am < p.M) {
        const int L = p.Ho * p.Wo;
        ab = am / L;
        const int l = am - ab * L;
        aho = l / p.Wo;
        awo = l - aho * p.Wo;
    }
    v16f acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    for (int k0 = 0; k0 < p.K; k0 += CF_BK) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = k0 + sk + j;
            float a = 0.0f, b = 0.0f;
            if (k < p.K) {
                if (am < p.M) {
                    const int tap = k / p.C, c = k - tap * p.C;
                    const int dh = tap / p.kw, dw = tap - dh * p.kw;
                    const int hi = aho * p.stride - p.pad + dh, wi = awo * p.stride - p.pad + dw;
                    if (hi >= 0 && hi < p.H && wi >= 0 && wi < p.W)
                        a = ld_pre(p, ab, ((int64_t)ab * p.H + hi) * p.W + wi, c);
                }
                if (n0 + srow < p.N) b = p.w[(int64_t)(n0 + srow) * p.K + k];
            }
            As[srow][sk + j] = a;
            Bs[srow][sk + j] = b;
        }
        __syncthreads();
        // MFMA_F32_32X32X2: lane l supplies row / column l & 31 at k = l >> 5
#pragma unroll
        for (int kk = 0; kk < CF_BK; kk += 2) {
            const float av = As[wm + (lane & 31)][kk + (lane >> 5)];
            const float bv = Bs[wn + (lane & 31)][kk + (lane >> 5)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    // C/D layout: column = lane & 31, row = (r & 3) + 8·(r >> 2) + 4·(lane >> 5)
    const int n = n0 + wn + (lane & 31);
    if (n >= p.N) return;
    const float bias = p.bias ? p.bias[n] : 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m >= p.M) continue;
        const float v = acc[r] + bias;
        store_any(p.y, p.y_dtype, (int64_t)m * p.ldy + n, v);
        if (p.y2) store_any(p.y2, p.y_dtype, (int64_t)m * p.ldy2 + n, v);
    }
    if (p.gn_partial) {
        // GroupNorm partials of the output (conv_in feeds the first resnet's norm1 and, as a skip, the last up resnet's): a lane holds 8 of
        // the 16 rows of each of its two 16-row blocks (rows (r & 3) + 4·(lane >> 5) + 8·(r >> 2)), lane ^ 32 the other 8: two-pass
        // statistics of the 8 stored values, merged with the partner's by Chan's formula (equal counts)
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            float val[8], sum = 0.0f;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                float v = acc[blk * 8 + q] + bias;
                if (p.y_dtype == DGQ_F16) v = __half2float(__float2half(v));
                else if (p.y_dtype == DGQ_BF16) v = __bfloat162float(__float2bfloat16(v));
                val[q] = v;
                sum += v;
            }
            float mean = sum * 0.125f, m2 = 0.0f;
#pragma unroll
            for (int q = 0; q < 8; ++q) m2 += (val[q] - mean) * (val[q] - mean);
            const float om = __shfl_xor(mean, 32, 64), o2 = __shfl_xor(m2, 32, 64);
            const float dd = om - mean;
            m2 = m2 + o2 + dd * dd * 4.0f;
            mean = 0.5f * (mean + om);
            const int mb = m0 + wm + blk * 16;
            if ((lane >> 5) == 0 && mb < p.M) {
                float* q2 = p.gn_partial + ((int64_t)(mb >> 4) * p.N + n) * 2;
                q2[0] = mean; q2[1] = m2;
            }
        }
    }
}

// N <= 8 output channels (conv_out of the UNets: 320 -> 4): a 64-wide column tile would idle 94 % of its MFMAs and walk K in
// 180 barrier-separated steps; here one wave owns an output position, its lanes split K (coalesced along the channels of each
// tap), every lane keeps N partial sums, and the wave reduces them with shuffles.  fmaf chain per lane + a fixed reduction tree.
template <int NMAX>
__global__ __launch_bounds__(256) void conv_f32w_smalln_kernel(ConvF32Params p) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= p.M) return;
    const int L = p.Ho * p.Wo;
    const int b = m / L, l = m - b * L;
    const int ho = l / p.Wo, wo = l - ho * p.Wo;
    float acc[NMAX];
#pragma unroll
    for (int n = 0; n < NMAX; ++n) acc[n] = 0.0f;
    // fp32 tensors with C % 4 == 0 (conv_out of every UNet): tap by tap (the bounds test is wave-uniform, no divisions per element),
    // 16-byte loads of four channels, of their folded-GroupNorm scale / shift and of the N weight rows: conv_out of an SD step (8192
    // positions x 2880 -> 4, statistics pass included) 90 -> 67 us, at the 131072 positions of config C5 1.03 -> 0.66 ms; fetching all 18
    // (tap, step) inputs of a position before the first use measured slower (84 us): the unrolled predicates cost more than the overlap
    // (16-bit tensors: the same units from 8-byte loads — the element-by-element form below cost 140 us against 36 for conv_out of a
    // bf16 SD step)
    const bool vec = (p.C & 3) == 0 && (reinterpret_cast<uintptr_t>(p.x) & 15) == 0 &&
                     (reinterpret_cast<uintptr_t>(p.w) & 15) == 0 &&
                     (!p.pre_scale || (((reinterpret_cast<uintptr_t>(p.pre_scale) | reinterpret_cast<uintptr_t>(p.pre_shift)) & 15) == 0));
    if (vec) {
        // (tap, four channels) units of the position dealt round-robin over the lanes, four units per lane and round with all their loads
        // issued before the first use: C = 320 is 720 units = 11.25 per lane in three rounds (N <= 4; two units per round for 5..8 outputs).  (Tap by tap with 256-channel steps the
        // second step of every tap ran on 16 of the 64 lanes and a wave walked 18 dependent load rounds: 49 us for conv_out of an SD step.)
        const int c4n = p.C >> 2, units = p.kh * p.kw * c4n;
        constexpr int UB = NMAX <= 4 ? 4 : 2;                      // (registers: UB x NMAX float4 of weights in flight)
        for (int u0 = lane; u0 < units; u0 += 64 * UB) {
            float4 v[UB], sc[UB], sh[UB], w4[UB][NMAX];
            bool inb[UB];
#pragma unroll
            for (int j = 0; j < UB; ++j) {
                const int u = min(u0 + 64 * j, units - 1);
                const int tap = u / c4n, c = (u - tap * c4n) << 2;
                const int dh = tap / p.kw, dw = tap - dh * p.kw;
                const int hi = ho * p.stride - p.pad + dh, wi = wo * p.stride - p.pad + dw;
                inb[j] = (u0 + 64 * j < units) && hi >= 0 && hi < p.H && wi >= 0 && wi < p.W;
                const int hc = min(max(hi, 0), p.H - 1), wc = min(max(wi, 0), p.W - 1);       // clamped: the load is unconditional
                const int64_t xi = (((int64_t)b * p.H + hc) * p.W + wc) * p.C + c;
                float t4[4];
                if (p.x_dtype == DGQ_F32) load4<float>(reinterpret_cast<const float*>(p.x) + xi, t4);              // (kernel-uniform)
                else if (p.x_dtype == DGQ_BF16) load4<__hip_bfloat16>(reinterpret_cast<const __hip_bfloat16*>(p.x) + xi, t4);
                else load4<__half>(reinterpret_cast<const __half*>(p.x) + xi, t4);
                v[j] = make_float4(t4[0], t4[1], t4[2], t4[3]);
                if (p.pre_scale) {
                    sc[j] = *reinterpret_cast<const float4*>(p.pre_scale + (int64_t)b * p.C + c);
                    sh[j] = *reinterpret_cast<const float4*>(p.pre_shift + (int64_t)b * p.C + c);
                }
#pragma unroll
                for (int n = 0; n < NMAX; ++n)
                    if (n < p.N) w4[j][n] = *reinterpret_cast<const float4*>(p.w + (int64_t)n * p.K + (int64_t)tap * p.C + c);
            }
#pragma unroll
            for (int j = 0; j < UB; ++j) {
                float4 t = v[j];
                if (p.pre_scale) {
                    t.x = t.x * sc[j].x + sh[j].x; t.y = t.y * sc[j].y + sh[j].y; t.z = t.z * sc[j].z + sh[j].z; t.w = t.w * sc[j].w + sh[j].w;
                }
                if (p.pre_act == 1) { t.x = dgq_silu(t.x); t.y = dgq_silu(t.y); t.z = dgq_silu(t.z); t.w = dgq_silu(t.w); }
                if (!inb[j]) t = make_float4(0.0f, 0.0f, 0.0f, 0.0f);                       // a tap outside the image / past the last unit adds nothing
#pragma unroll
                for (int n = 0; n < NMAX; ++n)
                    if (n < p.N) {
                        acc[n] = __builtin_fmaf(t.x, w4[j][n].x, acc[n]);
                        acc[n] = __builtin_fmaf(t.y, w4[j][n].y, acc[n]);
                        acc[n] = __builtin_fmaf(t.z, w4[j][n].z, acc[n]);
                        acc[n] = __builtin_fmaf(t.w, w4[j][n].w, acc[n]);
                    }
            }
        }
    } else
    for (int k = lane; k < p.K; k += 64) {
        const int tap = k / p.C, c = k - tap * p.C;
        const int dh = tap / p.kw, dw = tap - dh * p.kw;
        const int hi = ho * p.stride - p.pad + dh, wi = wo * p.stride - p.pad + dw;
        if (hi < 0 || hi >= p.H || wi < 0 || wi >= p.W) continue;
        const float a = ld_pre(p, b, ((int64_t)b * p.H + hi) * p.W + wi, c);
#pragma unroll
        for (int n = 0; n < NMAX; ++n)
            if (n < p.N) acc[n] = __builtin_fmaf(a, p.w[(int64_t)n * p.K + k], acc[n]);
    }
#pragma unroll
    for (int n = 0; n < NMAX; ++n) {
        float v = acc[n];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0 && n < p.N) {
            v += p.bias ? p.bias[n] : 0.0f;
            store_any(p.y, p.y_dtype, (int64_t)m * p.ldy + n, v);
            if (p.y2) store_any(p.y2, p.y_dtype, (int64_t)m * p.ldy2 + n, v);
        }
    }
}

extern "C" int dgq_conv2d_f32w(const void* x, int x_dtype, int B, int H, int W, int C, int kh, int kw, int stride, int pad,
                               const float* w, const float* bias, int N, void* y, int y_dtype, int ldy,
                               const float* pre_scale, const float* pre_shift, int pre_act, void* y2, int ldy2, float* gn_partial, void* stream) {
    DGQ_CHECK_ARG(x && w && y, "dgq_conv2d_f32w: null pointer");
    DGQ_CHECK_ARG(B > 0 && H > 0 && W > 0 && C > 0 && kh > 0 && kw > 0 && stride > 0 && pad >= 0 && N > 0, "dgq_conv2d_f32w: bad geometry");
    DGQ_CHECK_ARG((x_dtype == DGQ_F32 || x_dtype == DGQ_F16 || x_dtype == DGQ_BF16) && (y_dtype == DGQ_F32 || y_dtype == DGQ_F16 || y_dtype == DGQ_BF16),
                  "dgq_conv2d_f32w: unknown dtype");
    DGQ_CHECK_ARG((pre_scale == nullptr) == (pre_shift == nullptr) && (pre_act == 0 || pre_act == 1), "dgq_conv2d_f32w: bad prologue");
    ConvF32Params p;
    p.pre_scale = pre_scale; p.pre_shift = pre_shift; p.pre_act = pre_act;
    p.x = x; p.w = w; p.bias = bias; p.y = y; p.x_dtype = x_dtype; p.y_dtype = y_dtype;
    p.B = B; p.H = H; p.W = W; p.C = C; p.kh = kh; p.kw = kw; p.stride = stride; p.pad = pad;
    p.Ho = (H + 2 * pad - kh) / stride + 1; p.Wo = (W + 2 * pad - kw) / stride + 1;
    DGQ_CHECK_ARG(p.Ho > 0 && p.Wo > 0, "dgq_conv2d_f32w: empty output");
    p.N = N; p.K = kh * kw * C; p.M = B * p.Ho * p.Wo; p.ldy = ldy;
    DGQ_CHECK_ARG(ldy >= N && (!y2 || ldy2 >= N), "dgq_conv2d_f32w: ldy < N");
    p.y2 = y2; p.ldy2 = ldy2;
    DGQ_CHECK_ARG(!gn_partial || (p.M % 16 == 0 && N > 8), "dgq_conv2d_f32w: GroupNorm partials need M %% 16 == 0 and the tiled kernel (N > 8)");
    p.gn_partial = gn_partial;
    if (N <= 8) {
        if (N <= 4) hipLaunchKernelGGL(conv_f32w_smalln_kernel<4>, dim3((p.M + 3) / 4), dim3(256), 0, (hipStream_t)stream, p);
        else hipLaunchKernelGGL(conv_f32w_smalln_kernel<8>, dim3((p.M + 3) / 4), dim3(256), 0, (hipStream_t)stream, p);
        return dgq_launch_status("dgq_conv2d_f32w");
    }
    dim3 grid((N + CF_BN - 1) / CF_BN, (p.M + CF_BM - 1) / CF_BM);
    DGQ_CHECK_ARG(grid.y <= 65535, "dgq_conv2d_f32w: M = %d rows exceed the grid", p.M);
    hipLaunchKernelGGL(conv_f32w_kernel, grid, dim3(256), 0, (hipStream_t)stream, p);
    return dgq_launch_status("dgq_conv2d_f32w");
}
