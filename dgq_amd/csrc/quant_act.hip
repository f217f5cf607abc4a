// Activation quantise-on-load pre-pass: channels-last fp tensor -> int8 codes of the (implicit) unfolded
// operand in the packed weight's K order, + one float per row.  HBM-bound; one wave per output row,
// 16 consecutive kp (one 16-byte store) per lane per iteration, exact fp32 division per element.
#include "dgq_common.h"

struct QuantActParams {
    const void* x;
    int B, H, W, C, kh, kw, stride, pad, Ho, Wo;
    const int32_t* ksrc;      // [Kp] (tap<<16 | c) or -1, or NULL (natural order kp = tap*C + c)
    int Kp, K;
    const float* delta;       // per_m: [L]; else [Kp/64]
    const float* zp;
    int L;
    float qmax, offset;
    int8_t* codes;
    float* rowsum;
    int M;
};

template <typename TIn>
__device__ __forceinline__ void load16(const TIn* p, float (&v)[16]);

template <>
__device__ __forceinline__ void load16<float>(const float* p, float (&v)[16]) {
    const float4* q = reinterpret_cast<const float4*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float4 t = q[i];
        v[4 * i] = t.x; v[4 * i + 1] = t.y; v[4 * i + 2] = t.z; v[4 * i + 3] = t.w;
    }
}
template <>
__device__ __forceinline__ void load16<__half>(const __half* p, float (&v)[16]) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        uint4 t = q[i];
        const __half* h = reinterpret_cast<const __half*>(&t);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[8 * i + j] = __half2float(h[j]);
    }
}
template <>
__device__ __forceinline__ void load16<__hip_bfloat16>(const __hip_bfloat16* p, float (&v)[16]) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        uint4 t = q[i];
        const uint16_t* h = reinterpret_cast<const uint16_t*>(&t);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[8 * i + j] = __uint_as_float(((uint32_t)h[j]) << 16);
    }
}

template <typename TIn, bool HAS_TABLE, bool PER_M>
__global__ __launch_bounds__(256) void quant_act_kernel(QuantActParams p) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.M) return;                                 // whole wave leaves; no barriers below
    const TIn* x = reinterpret_cast<const TIn*>(p.x);
    const int L = p.Ho * p.Wo;
    const int b = row / L;
    const int l = row - b * L;
    const int ho = l / p.Wo, wo = l - ho * p.Wo;
    const int hbase = ho * p.stride - p.pad, wbase = wo * p.stride - p.pad;
    const int64_t img = (int64_t)b * p.H * p.W;
    float md = 1.0f, mz = 0.0f;
    if (PER_M) {
        int li = row % p.L;
        md = p.delta[li];
        mz = p.zp[li];
    }
    float partial = 0.0f;
    int8_t* out = p.codes + (int64_t)row * p.Kp;
    for (int kp0 = lane * 16; kp0 < p.Kp; kp0 += 64 * 16) {
        float d = md, z = mz;
        if (!PER_M) {
            d = p.delta[kp0 >> 6];
            z = p.zp[kp0 >> 6];
        }
        float v[16];
        bool valid[16];
        if (!HAS_TABLE) {
            // natural order: 16 | C, so the 16 elements share one tap and are contiguous in c
            bool in_k = kp0 < p.K;
            int tap = in_k ? kp0 / p.C : 0;
            int c = kp0 - tap * p.C;
            int dh = tap / p.kw, dw = tap - dh * p.kw;
            int hi = hbase + dh, wi = wbase + dw;
            bool inb = in_k && hi >= 0 && hi < p.H && wi >= 0 && wi < p.W;
            if (inb) {
                load16<TIn>(x + ((img + (int64_t)hi * p.W + wi) * p.C + c), v);
            } else {
#pragma unroll
                for (int j = 0; j < 16; ++j) v[j] = 0.0f;
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) valid[j] = in_k;
        } else {
            int idx[16];
            const int4* t4 = reinterpret_cast<const int4*>(p.ksrc + kp0);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int4 t = t4[i];
                idx[4 * i] = t.x; idx[4 * i + 1] = t.y; idx[4 * i + 2] = t.z; idx[4 * i + 3] = t.w;
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                int e = idx[j];
                valid[j] = e >= 0;
                int tap = (e >> 16) & 0x7FFF, c = e & 0xFFFF;
                int dh = tap / p.kw, dw = tap - dh * p.kw;
                int hi = hbase + dh, wi = wbase + dw;
                bool inb = valid[j] && hi >= 0 && hi < p.H && wi >= 0 && wi < p.W;
                v[j] = inb ? dgq_to_float(x[(img + (int64_t)hi * p.W + wi) * p.C + c]) : 0.0f;
            }
        }
        uint32_t w4[4] = {0, 0, 0, 0};
        int ssum = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            float q = dgq_affine_code(v[j], d, z, p.qmax);
            int s = valid[j] ? ((int)q - (int)p.offset) : 0;
            ssum += s;
            w4[j >> 2] |= ((uint32_t)(s & 0xFF)) << (8 * (j & 3));
        }
        partial += PER_M ? (float)ssum : d * (float)ssum;
        *reinterpret_cast<uint4*>(out + kp0) = make_uint4(w4[0], w4[1], w4[2], w4[3]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) partial += __shfl_down(partial, o, 64);
    if (lane == 0) p.rowsum[row] = partial;
}

template <typename TIn>
static void launch_quant_act(const QuantActParams& p, bool table, bool per_m, hipStream_t st) {
    dim3 grid((p.M + 3) / 4), block(256);
    if (table) {
        if (per_m) hipLaunchKernelGGL((quant_act_kernel<TIn, true, true>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((quant_act_kernel<TIn, true, false>), grid, block, 0, st, p);
    } else {
        if (per_m) hipLaunchKernelGGL((quant_act_kernel<TIn, false, true>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((quant_act_kernel<TIn, false, false>), grid, block, 0, st, p);
    }
}

extern "C" int dgq_quant_act(const void* x, int x_dtype, int B, int H, int W, int C,
                             int kh, int kw, int stride, int pad,
                             const int32_t* ksrc, int Kp,
                             int per_m, const float* delta, const float* zp, int L,
                             int bits, int8_t* codes, float* rowsum, void* stream) {
    DGQ_CHECK_ARG(x && delta && zp && codes && rowsum, "dgq_quant_act: null pointer");
    DGQ_CHECK_ARG(B > 0 && H > 0 && W > 0 && C > 0 && kh > 0 && kw > 0 && stride > 0 && pad >= 0,
                  "dgq_quant_act: bad geometry");
    DGQ_CHECK_ARG(kh * kw <= 0x7FFF && C <= 0xFFFF, "dgq_quant_act: kernel/channel count out of range");
    DGQ_CHECK_ARG(Kp > 0 && Kp % DGQ_KTILE == 0, "dgq_quant_act: Kp=%d must be a multiple of %d", Kp, DGQ_KTILE);
    DGQ_CHECK_ARG(bits >= 2 && bits <= 8, "dgq_quant_act: bits=%d", bits);
    DGQ_CHECK_ARG(!per_m || L >= 1, "dgq_quant_act: per_m needs L >= 1");
    int K = C * kh * kw;
    if (!ksrc) {
        DGQ_CHECK_ARG(C % 16 == 0, "dgq_quant_act: natural K order needs C %% 16 == 0 (C=%d)", C);
        DGQ_CHECK_ARG(Kp >= K, "dgq_quant_act: natural K order needs Kp >= K");
    }
    int Ho = (H + 2 * pad - kh) / stride + 1, Wo = (W + 2 * pad - kw) / stride + 1;
    DGQ_CHECK_ARG(Ho > 0 && Wo > 0, "dgq_quant_act: empty output");
    QuantActParams p;
    p.x = x; p.B = B; p.H = H; p.W = W; p.C = C; p.kh = kh; p.kw = kw; p.stride = stride; p.pad = pad;
    p.Ho = Ho; p.Wo = Wo; p.ksrc = ksrc; p.Kp = Kp; p.K = K; p.delta = delta; p.zp = zp; p.L = per_m ? L : 1;
    p.qmax = (float)((1 << bits) - 1);
    p.offset = bits == 8 ? 128.0f : 0.0f;
    p.codes = codes; p.rowsum = rowsum; p.M = B * Ho * Wo;
    hipStream_t st = (hipStream_t)stream;
    switch (x_dtype) {
        case DGQ_F32: launch_quant_act<float>(p, ksrc != nullptr, per_m != 0, st); break;
        case DGQ_F16: launch_quant_act<__half>(p, ksrc != nullptr, per_m != 0, st); break;
        case DGQ_BF16: launch_quant_act<__hip_bfloat16>(p, ksrc != nullptr, per_m != 0, st); break;
        default: dgq_set_error("dgq_quant_act: unknown dtype %d", x_dtype); return DGQ_EINVAL;
    }
    return dgq_launch_status("dgq_quant_act");
}
