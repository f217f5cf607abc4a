// Fused quantised-softmax attention on the bf16 MFMA with fp32-equivalent accuracy ("bf16x3").
//
// Same two-pass algorithm and interface as attn_fused.hip (exact fp32 MFMA, 64 FLOP/clk/SIMD), but the matrix
// products run on V_MFMA_F32_32X32X16_BF16 (16x the rate):
//   * every fp32 operand x is split exactly into three bf16 terms x = h + m + l (24 significant bits);
//   * Q·K^T keeps the six products whose weight is >= 2^-16 of the leading one (hh, hm, mh, hl, lh, mm): each bf16 x
//     bf16 product is exact in the fp32 accumulator, the dropped terms are below fp32 rounding of the sum;
//   * the quantised probabilities are EXACT single bf16 numbers: log2 quantiser p̂/δ = 2^-code, uniform quantiser
//     p̂/δ = code ∈ [0,255]; so P̂·V needs only the three V terms, and δ multiplies the output once;
//   * the start-peak column (an unquantised probability) is added as a rank-1 fp32 update.
// Work decomposition as attn_fused.hip: block = 4 waves = 128 query rows of one (batch, head), 32-key tiles, the
// score tile is computed transposed (S^T = K·Q^T) so softmax statistics are in-register and the S^T accumulator is
// directly the B operand of O^T = V^T·P̂^T (k order inside a 16-key step: element j of lane half h is key
// 16s + 8(j>>2) + 4h + (j&3) — the V tile is stored transposed in exactly that key order).
#include "dgq_common.h"

typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define QROWS 128
#define KT 32
#define LOG2E 1.4426950408889634f

struct AttnParams {
    const float* q;
    const float* k;
    const float* v;
    float* o;
    int B, H, T, S;
    float scale;
    int mode;              // 1: log2 real-time δ, 2: log2 static δ, 3: uniform (δ, z = 0)
    int skip;
    float qmax;
    float* stats;          // [B*H][T][2] : m (log2 units), l
    float* delta;
};

__device__ __forceinline__ unsigned short bf16_bits(float x) {
    return __builtin_bit_cast(unsigned short, __float2bfloat16(x));
}
__device__ __forceinline__ float bf16_to_f(unsigned short b) { return __uint_as_float(((unsigned)b) << 16); }

// exact three-way split
__device__ __forceinline__ void split3(float x, unsigned short& h, unsigned short& m, unsigned short& l) {
    h = bf16_bits(x);
    const float r1 = x - bf16_to_f(h);
    m = bf16_bits(r1);
    const float r2 = r1 - bf16_to_f(m);
    l = bf16_bits(r2);
}

template <int D> struct Geo {
    static constexpr int DP = (D + 15) / 16 * 16;      // K depth of the score product
    static constexpr int NKK = DP / 16;
    static constexpr int NDT = (D + 31) / 32;          // 32-wide d tiles of O^T
    static constexpr int KLD = DP + 8;                 // bf16 elements per K row (16-byte aligned, de-conflicted)
    static constexpr int VLD = KT + 8;                 // bf16 elements per V^T row
    static constexpr int K_ELEMS = 3 * KT * KLD;
    static constexpr int V_ELEMS = 3 * NDT * 32 * VLD;
};

// K tile -> LDS as three bf16 planes [3][32][KLD]
template <int D>
__device__ __forceinline__ void stage_k(unsigned short* kb, const float* src, int valid, int row_stride, int tid) {
    using G = Geo<D>;
    for (int i = tid; i < KT * G::DP / 2; i += 256) {
        const int r = i / (G::DP / 2), c = (i - r * (G::DP / 2)) * 2;
        float x0 = 0.0f, x1 = 0.0f;
        if (r < valid) {
            if (c < D) x0 = src[(int64_t)r * row_stride + c];
            if (c + 1 < D) x1 = src[(int64_t)r * row_stride + c + 1];
        }
        unsigned short h0, m0, l0, h1, m1, l1;
        split3(x0, h0, m0, l0);
        split3(x1, h1, m1, l1);
        unsigned* dst = reinterpret_cast<unsigned*>(kb + r * G::KLD + c);
        dst[0] = (unsigned)h0 | ((unsigned)h1 << 16);
        dst[(KT * G::KLD) / 2] = (unsigned)m0 | ((unsigned)m1 << 16);
        dst[KT * G::KLD] = (unsigned)l0 | ((unsigned)l1 << 16);
    }
}

// V tile -> LDS transposed, three planes [3][NDT*32][VLD]; key `kk` of the tile goes to slot 16s + 8h + 4a + b where
// kk = 16s + 8a + 4h + b (the k order of an accumulator tile used as the next MFMA's B operand)
template <int D>
__device__ __forceinline__ void stage_v(unsigned short* vt, const float* src, int valid, int row_stride, int tid) {
    using G = Geo<D>;
    constexpr int DV = G::NDT * 32;
    for (int i = tid; i < KT * DV; i += 256) {
        const int key = i / DV, d = i - key * DV;
        const float x = (key < valid && d < D) ? src[(int64_t)key * row_stride + d] : 0.0f;
        unsigned short h, m, l;
        split3(x, h, m, l);
        const int s = key >> 4, a = (key >> 3) & 1, hh = (key >> 2) & 1, b = key & 3;
        const int slot = 16 * s + 8 * hh + 4 * a + b;
        unsigned short* dst = vt + d * G::VLD + slot;
        dst[0] = h;
        dst[DV * G::VLD] = m;
        dst[2 * DV * G::VLD] = l;
    }
}

// Q rows of this lane as B-operand fragments: qf[split][kk] holds Q[t][16kk + 8h + j], j = 0..7
template <int D>
__device__ __forceinline__ void load_q(bf16x8 (&qf)[3][Geo<D>::NKK], const float* qrow, int h32) {
    using G = Geo<D>;
#pragma unroll
    for (int kk = 0; kk < G::NKK; ++kk) {
        unsigned short hs[8], ms[8], ls[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int d = 16 * kk + 8 * h32 + j;
            const float x = (d < D) ? qrow[d] : 0.0f;
            split3(x, hs[j], ms[j], ls[j]);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            qf[0][kk][j] = __builtin_bit_cast(__bf16, hs[j]);
            qf[1][kk][j] = __builtin_bit_cast(__bf16, ms[j]);
            qf[2][kk][j] = __builtin_bit_cast(__bf16, ls[j]);
        }
    }
}

// S^T tile: acc[r] = Σ_d K[key_of(r,h)][d]·Q[t][d] (unscaled), six bf16 products per 16-deep step
template <int D>
__device__ __forceinline__ v16f score_tile(const unsigned short* kb, const bf16x8 (&qf)[3][Geo<D>::NKK], int lane) {
    using G = Geo<D>;
    v16f acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    const unsigned short* kp = kb + (lane & 31) * G::KLD + 8 * (lane >> 5);
    constexpr int PL = KT * G::KLD;
#pragma unroll
    for (int kk = 0; kk < G::NKK; ++kk) {
        const bf16x8 kh = *reinterpret_cast<const bf16x8*>(kp + 16 * kk);
        const bf16x8 km = *reinterpret_cast<const bf16x8*>(kp + PL + 16 * kk);
        const bf16x8 kl = *reinterpret_cast<const bf16x8*>(kp + 2 * PL + 16 * kk);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl, qf[0][kk], acc, 0, 0, 0);    // smallest terms first
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qf[2][kk], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(km, qf[1][kk], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(km, qf[0][kk], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qf[1][kk], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qf[0][kk], acc, 0, 0, 0);
    }
    return acc;
}

__device__ __forceinline__ int key_of(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

template <int D>
__global__ __launch_bounds__(256) void attn3_stats_kernel(AttnParams p) {
    using G = Geo<D>;
    extern __shared__ __attribute__((aligned(16))) unsigned short lds16[];
    unsigned short* kb = lds16;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, h32 = lane >> 5;
    const int bh = blockIdx.y, b = bh / p.H, hd = bh - b * p.H;
    const int HD = p.H * D;
    const int t = blockIdx.x * QROWS + wid * 32 + (lane & 31);
    const int tq = min(t, p.T - 1);
    bf16x8 qf[3][G::NKK];
    load_q<D>(qf, p.q + ((int64_t)(b * p.T + tq) * p.H + hd) * D, h32);
    const float sl2 = p.scale * LOG2E;                   // scores in log2 units: p = 2^(s2 − m)/l
    float m = -INFINITY, l = 0.0f, m2 = -INFINITY;
    for (int s0 = 0; s0 < p.S; s0 += KT) {
        __syncthreads();
        stage_k<D>(kb, p.k + ((int64_t)(b * p.S + s0) * p.H + hd) * D, min(KT, p.S - s0), HD, tid);
        __syncthreads();
        v16f acc = score_tile<D>(kb, qf, lane);
        float tmax = -INFINITY, tmax2 = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int s = s0 + key_of(r, h32);
            const float sc = (s < p.S) ? acc[r] * sl2 : -INFINITY;
            acc[r] = sc;
            tmax = fmaxf(tmax, sc);
            if (s >= p.skip) tmax2 = fmaxf(tmax2, sc);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        tmax2 = fmaxf(tmax2, __shfl_xor(tmax2, 32, 64));
        const float mn = fmaxf(m, tmax);
        float part = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) part += exp2f(acc[r] - mn);
        part += __shfl_xor(part, 32, 64);
        l = l * exp2f(m - mn) + part;
        m = mn;
        m2 = fmaxf(m2, tmax2);
    }
    if (t < p.T && h32 == 0) {
        float* st = p.stats + ((int64_t)bh * p.T + t) * 2;
        st[0] = m;
        st[1] = l;
    }
    if (p.mode == 1) {
        float pm = (t < p.T) ? exp2f(m2 - m) / l : 0.0f;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) pm = fmaxf(pm, __shfl_xor(pm, o, 64));
        if (lane == 0) atomicMax(reinterpret_cast<int*>(p.delta), __float_as_int(pm));
    }
}

template <int D>
__global__ __launch_bounds__(256) void attn3_pv_kernel(AttnParams p) {
    using G = Geo<D>;
    extern __shared__ __attribute__((aligned(16))) unsigned short lds16[];
    unsigned short* kb = lds16;
    unsigned short* vt = lds16 + G::K_ELEMS;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, h32 = lane >> 5;
    const int bh = blockIdx.y, b = bh / p.H, hd = bh - b * p.H;
    const int HD = p.H * D;
    const int t = blockIdx.x * QROWS + wid * 32 + (lane & 31);
    const int tq = min(t, p.T - 1);
    bf16x8 qf[3][G::NKK];
    load_q<D>(qf, p.q + ((int64_t)(b * p.T + tq) * p.H + hd) * D, h32);
    const float m = p.stats[((int64_t)bh * p.T + tq) * 2], l = p.stats[((int64_t)bh * p.T + tq) * 2 + 1];
    const float delta = p.delta[0];
    const float nsl2 = -(p.scale * LOG2E);
    const float a0 = m + log2f(l) + log2f(delta);       // −log2(p/δ) = a0 − s2
    const float inv_l = 1.0f / l;
    float p_bypass = 0.0f;                               // unquantised probability of key 0 (start-peak)
    v16f oacc[G::NDT];
#pragma unroll
    for (int j = 0; j < G::NDT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[j][r] = 0.0f;
    constexpr int VPL = G::NDT * 32 * G::VLD;

    for (int s0 = 0; s0 < p.S; s0 += KT) {
        __syncthreads();
        const int valid = min(KT, p.S - s0);
        stage_k<D>(kb, p.k + ((int64_t)(b * p.S + s0) * p.H + hd) * D, valid, HD, tid);
        stage_v<D>(vt, p.v + ((int64_t)(b * p.S + s0) * p.H + hd) * D, valid, HD, tid);
        __syncthreads();
        v16f acc = score_tile<D>(kb, qf, lane);
        const bool edge = (s0 + KT > p.S) || (s0 < p.skip);      // block-uniform: tail tile or the bypassed column
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float ph;
            if (p.mode == 3) {
                const float pr = exp2f(fmaf(acc[r], -nsl2, -m)) * inv_l;     // uniform, always_zero: code = clamp(rne(p/δ))
                ph = fminf(fmaxf(rintf(__fdiv_rn(pr, delta)), 0.0f), p.qmax);
            } else {
                float code = rintf(fmaf(acc[r], nsl2, a0));
                code = fminf(fmaxf(code, 0.0f), p.qmax);
                // 2^-code as fp32 bits; codes > 126 (p̂ < 2^-126·δ) are below anything the fp32 sum can resolve
                const int e = 127 - (int)code;
                ph = e > 0 ? __int_as_float(e << 23) : 0.0f;
            }
            if (edge) {
                const int s = s0 + key_of(r, h32);
                if (s >= p.S) ph = 0.0f;
                else if (s < p.skip) {
                    p_bypass = exp2f(fmaf(acc[r], -nsl2, -m)) * inv_l;
                    ph = 0.0f;
                }
            }
            acc[r] = ph;
        }
        // B fragments of the two 16-key steps: exact bf16 = upper halves of the fp32 words
        bf16x8 pf[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            unsigned w[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                w[i] = (__float_as_uint(acc[8 * ks + 2 * i]) >> 16) | (__float_as_uint(acc[8 * ks + 2 * i + 1]) & 0xFFFF0000u);
            pf[ks] = __builtin_bit_cast(bf16x8, w);
        }
#pragma unroll
        for (int j = 0; j < G::NDT; ++j) {
            const unsigned short* vp = vt + (j * 32 + (lane & 31)) * G::VLD + 8 * h32;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x8 vh = *reinterpret_cast<const bf16x8*>(vp + 16 * ks);
                const bf16x8 vm = *reinterpret_cast<const bf16x8*>(vp + VPL + 16 * ks);
                const bf16x8 vl = *reinterpret_cast<const bf16x8*>(vp + 2 * VPL + 16 * ks);
                oacc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl, pf[ks], oacc[j], 0, 0, 0);
                oacc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vm, pf[ks], oacc[j], 0, 0, 0);
                oacc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, pf[ks], oacc[j], 0, 0, 0);
            }
        }
    }
    if (p.skip > 0) p_bypass = __shfl(p_bypass, lane & 31, 64);   // key 0 lives in the lower half-wave
    if (t < p.T) {
        float* op = p.o + ((int64_t)(b * p.T + t) * p.H + hd) * D;
        const float* v0 = p.v + ((int64_t)(b * p.S) * p.H + hd) * D;
#pragma unroll
        for (int j = 0; j < G::NDT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int d = j * 32 + key_of(r, h32);
                if (d < D) {
                    float o = delta * oacc[j][r];
                    if (p.skip > 0) o += p_bypass * v0[d];
                    op[d] = o;
                }
            }
    }
}

template <int D>
static int launch_attn3(const AttnParams& p, hipStream_t st) {
    using G = Geo<D>;
    dim3 grid((p.T + QROWS - 1) / QROWS, p.B * p.H), block(256);
    hipLaunchKernelGGL((attn3_stats_kernel<D>), grid, block, G::K_ELEMS * 2, st, p);
    hipLaunchKernelGGL((attn3_pv_kernel<D>), grid, block, (G::K_ELEMS + G::V_ELEMS) * 2, st, p);
    return dgq_launch_status("dgq_attention_f32(bf16x3)");
}

// called from dgq_attention_f32 (attn_fused.hip) for the quantised modes; returns 1 when D is not instantiated here
int dgq_attention_bf16x3(const float* q, const float* k, const float* v, float* o, int B, int H, int T, int S, int D,
                         float scale, int mode, int skip, float qmax, float* stats_ws, float* delta_ws, hipStream_t st) {
    AttnParams p;
    p.q = q; p.k = k; p.v = v; p.o = o; p.B = B; p.H = H; p.T = T; p.S = S; p.scale = scale; p.mode = mode; p.skip = skip;
    p.qmax = qmax; p.stats = stats_ws; p.delta = delta_ws;
    switch (D) {
        case 8: return launch_attn3<8>(p, st);
        case 16: return launch_attn3<16>(p, st);
        case 40: return launch_attn3<40>(p, st);
        case 64: return launch_attn3<64>(p, st);
        case 80: return launch_attn3<80>(p, st);
        default: return 1;
    }
}
