// Activation quantise-on-load pre-pass: channels-last fp tensor -> int8 codes of the (implicit) unfolded
// operand in the packed weight's K order, + one float per row.  HBM-bound; one wave per output row, 4 consecutive
// kp (one packed dword) per lane per 256-wide step; rounding bit-identical to fp32 true division (see below).
#include <algorithm>
#include "dgq_common.h"

struct QuantActParams {
    const void* x;
    int B, H, W, C, kh, kw, stride, pad, Ho, Wo;
    const int32_t* ksrc;      // [Kp] (dh<<24 | dw<<16 | c) or -1, or NULL (natural order kp = tap*C + c)
    const int32_t* koff;      // optional [Kp]: (dh*W + dw)*ldc + c for THIS geometry, -1 for padding (interior rows)
    const int32_t* klds;      // optional [Kp]: (dh*kw + dw)*C + c, -1 for padding (LDS-staged conv path)
    int Kp, K;
    const float* delta;       // per_m: [L]; else [Kp/64]
    const float* zp;
    int L;
    float qmax, offset;
    int8_t* codes;
    float* rowsum;            // [ksplits][M] partial sums (the GEMM epilogue adds them in a fixed order)
    int M;
    int kp_per_split;         // multiple of 256
    const float* pre_scale;   // optional [B][C]: v = x*scale + shift (fused GroupNorm), then pre_act
    const float* pre_shift;
    int pre_act;              // 0 none, 1 SiLU, 2 GEGLU: value = x[c]·gelu(x[C + c]) on rows of 2C elements
    int ldc;                  // elements per input pixel/row (C, or 2C for GEGLU)
    const float* ln_gamma;    // optional [C]: LayerNorm over the C elements of the row, v = (x − μ)·rstd·γ + β (1x1 only)
    const float* ln_beta;
    float ln_eps;
};



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

// LayerNorm statistics of one row of C <= 2048 elements (C % 4 == 0), computed by the wave that quantises the row: the
// row is read ONCE into registers (8 float4 per lane), mean first, then Σ(x − mean)² from the registers; biased
// variance, rstd = 1/sqrt(var + eps) as nn.LayerNorm.
#define DGQ_LN_MAX_C 2048
template <typename TIn>
__device__ __forceinline__ void row_layernorm_stats(const TIn* xr, int C, float eps, int lane, float& mu, float& rstd) {
    float v[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = lane * 4 + 256 * i;
        if (c < C) load4<TIn>(xr + c, v[i]);
        else v[i][0] = v[i][1] = v[i][2] = v[i][3] = 0.0f;
    }
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    mu = s / (float)C;
    float q = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = lane * 4 + 256 * i;
        if (c < C) {
#pragma unroll
            for (int j = 0; j < 4; ++j) q += (v[i][j] - mu) * (v[i][j] - mu);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    rstd = 1.0f / sqrtf(q / (float)C + eps);
}

// Four codes q_j ∈ [0, 2^b−1] (floats) -> one dword of centred int8 codes s_j = q_j − off, 0 for padding:
// v_cvt_pk_u8_f32 inserts u8(q − off + 128) per byte, and u8(x + 128) ^ 0x80 is the two's-complement byte of x.
// `biased[j]` = valid ? q_j − off + 128 : 128 ; returns the dword, adds Σ biased to `fsum` (exact small integers).
__device__ __forceinline__ uint32_t dgq_pack4(const float (&biased)[4], float& fsum) {
    uint32_t w = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        w = __builtin_amdgcn_cvt_pk_u8_f32(biased[j], j, w);
        fsum += biased[j];
    }
    return w ^ 0x80808080u;
}

// One wave per output row; each lane owns 4 consecutive kp per 256-wide step (one packed dword), so that the
// table read (int4), the gathered loads (lane stride 16 B within a (group, tap) run) and the code store (256 B per
// wave instruction) are all coalesced.  The 4 kp of a lane share one 64-wide chunk, hence one (δ, z).
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
    const TIn* img = x + (int64_t)b * p.H * p.W * p.ldc;
    const int rowoff = (hbase * p.W + wbase) * p.ldc;
    const bool interior = hbase >= 0 && wbase >= 0 && hbase + p.kh <= p.H && wbase + p.kw <= p.W;
    const float* pre_sc = p.pre_scale ? p.pre_scale + (int64_t)b * p.C : nullptr;
    const float* pre_sh = p.pre_shift ? p.pre_shift + (int64_t)b * p.C : nullptr;
    float md = 1.0f, mz = 0.0f, minv = 1.0f;
    if (PER_M) {
        const int li = row % p.L;
        md = p.delta[li];
        mz = p.zp[li];
        minv = dgq_rcp(md);
    }
    float partial = 0.0f;
    float ln_mu = 0.0f, ln_rstd = 1.0f;
    if (p.ln_gamma) row_layernorm_stats<TIn>(img + rowoff, p.C, p.ln_eps, lane, ln_mu, ln_rstd);   // 1x1: the row itself
    uint32_t* out = reinterpret_cast<uint32_t*>(p.codes + (int64_t)row * p.Kp);
    // K range of this wave (blockIdx.y): low-M layers would otherwise leave the chip empty (M=512: 2 waves per CU)
    const int k_begin = blockIdx.y * p.kp_per_split;
    const int k_end = min(p.Kp, k_begin + p.kp_per_split);
    // natural order: (tap, c) of the lane's first element, advanced by 256 per step without divisions
    int ntap = 0, nc = 0;
    if (!HAS_TABLE) {
        const int k0 = k_begin + lane * 4;
        ntap = k0 / p.C;
        nc = k0 - ntap * p.C;
    }
    const float bias = 128.0f - p.offset;                  // biased code = q − off + 128 ∈ [0,255]
    if (HAS_TABLE) {
        // gather path, 4 steps (1024 kp) per iteration: all table reads, then all 16 gathers, then quantise + store.
        // Interior rows (no tap outside the image: the great majority) read precomputed element offsets (koff) and skip
        // every bounds check; border rows decode (dh, dw, c) from ksrc.
        const TIn* __restrict__ imgr = img;
        const bool fast = interior && p.koff != nullptr;
        const int32_t* __restrict__ tab = fast ? p.koff : p.ksrc;
        for (int kb = k_begin + lane * 4; kb < k_end; kb += 1024) {
            int idx[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int kp0 = kb + 256 * u;
                int4 t = make_int4(-1, -1, -1, -1);
                if (kp0 < k_end) t = *reinterpret_cast<const int4*>(tab + kp0);
                idx[u][0] = t.x; idx[u][1] = t.y; idx[u][2] = t.z; idx[u][3] = t.w;
            }
            float v[4][4];
            if (fast) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int e = idx[u][j];
                        v[u][j] = dgq_to_float(imgr[rowoff + max(e, 0)]);        // padding reads element 0: value unused
                    }
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int e = idx[u][j];
                        const int dh = (e >> 24) & 0x7F, dw = (e >> 16) & 0xFF, c = e & 0xFFFF;
                        const int hi = hbase + dh, wi = wbase + dw;
                        const bool inb = e >= 0 && hi >= 0 && hi < p.H && wi >= 0 && wi < p.W;
                        const int off = rowoff + (dh * p.W + dw) * p.ldc + c;      // < 2^31 elements per image
                        v[u][j] = inb ? dgq_to_float(imgr[off]) : 0.0f;
                        idx[u][j] = e >= 0 ? (inb ? off - rowoff : -2) : -1;       // -2: out-of-image tap (value 0, no prologue)
                    }
            }
            if (p.pre_scale || p.pre_act || p.ln_gamma) {   // folded GroupNorm / LayerNorm / SiLU / GEGLU (wave-uniform branch)
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int e = idx[u][j];
                        if (e >= 0) {
                            const int c = e % p.ldc;                          // e = (dh·W + dw)·ldc + c
                            float val = v[u][j];
                            if (p.pre_scale) val = val * pre_sc[c] + pre_sh[c];
                            if (p.ln_gamma) val = (val - ln_mu) * ln_rstd * p.ln_gamma[c] + p.ln_beta[c];
                            if (p.pre_act == 1) val = dgq_silu(val);
                            else if (p.pre_act == 2) {
                                const float g = dgq_to_float(imgr[rowoff + e + p.C]);
                                val = val * (0.5f * g * (1.0f + erff(g * 0.70710678118654752f)));
                            }
                            v[u][j] = val;
                        }
                    }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int kp0 = kb + 256 * u;
                if (kp0 < k_end) {
                    float d = md, z = mz, inv = minv;
                    if (!PER_M) {
                        d = p.delta[kp0 >> 6];
                        z = p.zp[kp0 >> 6];
                        inv = dgq_rcp(d);
                    }
                    float biased[4], fsum = 0.0f;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float q = dgq_affine_code_fast(v[u][j], d, inv, z, p.qmax);
                        biased[j] = idx[u][j] != -1 ? q + bias : 128.0f;
                    }
                    out[kp0 >> 2] = dgq_pack4(biased, fsum);
                    fsum -= 512.0f;                                              // Σ (biased − 128) = Σ s
                    partial += PER_M ? fsum : d * fsum;
                }
            }
        }
    } else {
#pragma unroll 2
    for (int kp0 = k_begin + lane * 4; kp0 < k_end; kp0 += 256) {
        float d = md, z = mz, inv = minv;
        if (!PER_M) {
            d = p.delta[kp0 >> 6];
            z = p.zp[kp0 >> 6];
            inv = dgq_rcp(d);
        }
        float v[4];
        // natural order kp = tap*C + c ; 4 | C, so the lane's 4 elements are contiguous channels of one tap
        const bool in_k = kp0 < p.K;
        const int dh = ntap / p.kw, dw = ntap - dh * p.kw;
        const int hi = hbase + dh, wi = wbase + dw;
        const bool inb = in_k && hi >= 0 && hi < p.H && wi >= 0 && wi < p.W;
        if (inb) {
            load4<TIn>(img + ((int64_t)hi * p.W + wi) * p.ldc + nc, v);
            if (p.pre_scale) {
                const float4 sc = *reinterpret_cast<const float4*>(p.pre_scale + (int64_t)b * p.C + nc);
                const float4 sh = *reinterpret_cast<const float4*>(p.pre_shift + (int64_t)b * p.C + nc);
                v[0] = v[0] * sc.x + sh.x; v[1] = v[1] * sc.y + sh.y; v[2] = v[2] * sc.z + sh.z; v[3] = v[3] * sc.w + sh.w;
            }
            if (p.ln_gamma) {
                const float4 ga = *reinterpret_cast<const float4*>(p.ln_gamma + nc);
                const float4 be = *reinterpret_cast<const float4*>(p.ln_beta + nc);
                v[0] = (v[0] - ln_mu) * ln_rstd * ga.x + be.x; v[1] = (v[1] - ln_mu) * ln_rstd * ga.y + be.y;
                v[2] = (v[2] - ln_mu) * ln_rstd * ga.z + be.z; v[3] = (v[3] - ln_mu) * ln_rstd * ga.w + be.w;
            }
            if (p.pre_act == 1) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = dgq_silu(v[j]);
            } else if (p.pre_act == 2) {
                float g[4];
                load4<TIn>(img + ((int64_t)hi * p.W + wi) * p.ldc + p.C + nc, g);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = v[j] * (0.5f * g[j] * (1.0f + erff(g[j] * 0.70710678118654752f)));
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = 0.0f;
        }
        nc += 256;
        while (nc >= p.C) { nc -= p.C; ++ntap; }
        float biased[4], fsum = 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float q = dgq_affine_code_fast(v[j], d, inv, z, p.qmax);
            biased[j] = in_k ? q + bias : 128.0f;
        }
        out[kp0 >> 2] = dgq_pack4(biased, fsum);
        fsum -= 512.0f;
        partial += PER_M ? fsum : d * fsum;
    }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) partial += __shfl_down(partial, o, 64);
    if (lane == 0) p.rowsum[(int64_t)blockIdx.y * p.M + row] = partial;
}

// Per-K conv layers (the quantizer sees the unfolded operand, so each (c, tap) may carry its own group): the table
// gathers 4-byte elements scattered over the kh·kw pixel rows of this output position.  Straight from global memory
// that is one L1 access per element (measured: TCP_TOTAL_CACHE_ACCESSES = 32 per wave-instruction, the kernel runs at
// the L1's access rate, ~50 us for 8192 x 2880); here each wave first copies its taps — C contiguous floats each, with
// GroupNorm / SiLU applied and zeros for taps outside the image — into its own LDS strip with coalesced 16-byte loads
// and gathers from LDS.  No block-level sync: a wave only reads what it wrote.
template <typename TIn, bool PER_M>
__global__ __launch_bounds__(256) void quant_act_staged_kernel(QuantActParams p) {
    extern __shared__ __attribute__((aligned(16))) float strips[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int row = blockIdx.x * (blockDim.x >> 6) + wv;
    if (row >= p.M) return;                                 // whole wave leaves; no barriers below
    const TIn* x = reinterpret_cast<const TIn*>(p.x);
    const int L = p.Ho * p.Wo;
    const int b = row / L;
    const int l = row - b * L;
    const int ho = l / p.Wo, wo = l - ho * p.Wo;
    const int hbase = ho * p.stride - p.pad, wbase = wo * p.stride - p.pad;
    const TIn* img = x + (int64_t)b * p.H * p.W * p.ldc;
    const int taps = p.kh * p.kw;
    float* strip = strips + wv * taps * p.C;
    const float* pre_sc = p.pre_scale ? p.pre_scale + (int64_t)b * p.C : nullptr;
    const float* pre_sh = p.pre_shift ? p.pre_shift + (int64_t)b * p.C : nullptr;
    float ln_mu = 0.0f, ln_rstd = 1.0f;                 // Linear inputs (taps == 1): LayerNorm over the row, or GEGLU
    if (p.ln_gamma) row_layernorm_stats<TIn>(img + (int64_t)(hbase * p.W + wbase) * p.ldc, p.C, p.ln_eps, lane, ln_mu, ln_rstd);
    for (int tap = 0; tap < taps; ++tap) {
        const int dh = tap / p.kw, dw = tap - dh * p.kw;
        const int hi = hbase + dh, wi = wbase + dw;
        const bool inb = hi >= 0 && hi < p.H && wi >= 0 && wi < p.W;     // wave-uniform
        const TIn* src = img + ((int64_t)hi * p.W + wi) * p.ldc;
        float* dst = strip + tap * p.C;
        for (int c = lane * 4; c < p.C; c += 256) {
            float v[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            if (inb) {
                load4<TIn>(src + c, v);
                if (pre_sc) {
                    const float4 sc = *reinterpret_cast<const float4*>(pre_sc + c);
                    const float4 sh = *reinterpret_cast<const float4*>(pre_sh + c);
                    v[0] = v[0] * sc.x + sh.x; v[1] = v[1] * sc.y + sh.y; v[2] = v[2] * sc.z + sh.z; v[3] = v[3] * sc.w + sh.w;
                }
                if (p.ln_gamma) {
                    const float4 ga = *reinterpret_cast<const float4*>(p.ln_gamma + c);
                    const float4 be = *reinterpret_cast<const float4*>(p.ln_beta + c);
                    v[0] = (v[0] - ln_mu) * ln_rstd * ga.x + be.x; v[1] = (v[1] - ln_mu) * ln_rstd * ga.y + be.y;
                    v[2] = (v[2] - ln_mu) * ln_rstd * ga.z + be.z; v[3] = (v[3] - ln_mu) * ln_rstd * ga.w + be.w;
                }
                if (p.pre_act == 1) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = dgq_silu(v[j]);
                } else if (p.pre_act == 2) {
                    float g[4];
                    load4<TIn>(src + p.C + c, g);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = v[j] * (0.5f * g[j] * (1.0f + erff(g[j] * 0.70710678118654752f)));
                }
            }
            *reinterpret_cast<float4*>(dst + c) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
    float md = 1.0f, mz = 0.0f, minv = 1.0f;
    if (PER_M) {
        const int li = row % p.L;
        md = p.delta[li];
        mz = p.zp[li];
        minv = dgq_rcp(md);
    }
    float partial = 0.0f;
    uint32_t* out = reinterpret_cast<uint32_t*>(p.codes + (int64_t)row * p.Kp);
    const int k_begin = blockIdx.y * p.kp_per_split;
    const int k_end = min(p.Kp, k_begin + p.kp_per_split);
    const float bias = 128.0f - p.offset;
    const int32_t* __restrict__ tab = p.klds;
    for (int kb = k_begin + lane * 4; kb < k_end; kb += 1024) {
        int idx[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int kp0 = kb + 256 * u;
            int4 t = make_int4(-1, -1, -1, -1);
            if (kp0 < k_end) t = *reinterpret_cast<const int4*>(tab + kp0);
            idx[u][0] = t.x; idx[u][1] = t.y; idx[u][2] = t.z; idx[u][3] = t.w;
        }
        float v[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) v[u][j] = strip[max(idx[u][j], 0)];       // padding reads element 0: value unused
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int kp0 = kb + 256 * u;
            if (kp0 < k_end) {
                float d = md, z = mz, inv = minv;
                if (!PER_M) {
                    d = p.delta[kp0 >> 6];
                    z = p.zp[kp0 >> 6];
                    inv = dgq_rcp(d);
                }
                float biased[4], fsum = 0.0f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float q = dgq_affine_code_fast(v[u][j], d, inv, z, p.qmax);
                    biased[j] = idx[u][j] >= 0 ? q + bias : 128.0f;
                }
                out[kp0 >> 2] = dgq_pack4(biased, fsum);
                fsum -= 512.0f;
                partial += PER_M ? fsum : d * fsum;
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) partial += __shfl_down(partial, o, 64);
    if (lane == 0) p.rowsum[(int64_t)blockIdx.y * p.M + row] = partial;
}

template <typename TIn>
static void launch_quant_act(const QuantActParams& p, bool table, bool per_m, hipStream_t st) {
    const int ks = (p.Kp + p.kp_per_split - 1) / p.kp_per_split;
    const size_t strip_bytes = (size_t)p.kh * p.kw * p.C * sizeof(float);
    // measured (SD1.4 layers): staging wins at C = 320 (4 rows per block, 3 blocks per CU: 88 -> 64 us) and loses once the
    // strips cut occupancy (C >= 640 at 3x3) or the K range is split over blocks that would each re-stage every tap.
    // Linear inputs (one tap: the strip is the row itself, <= 16 KB) are always staged — the global gather pays one L1
    // access per code (8192 x 320 -> Kp 1024: 20 us) — together with their LayerNorm / GEGLU prologue.
    const int taps_ = p.kh * p.kw;
    const bool stage_conv = taps_ > 1 && p.pre_act != 2 && !p.ln_gamma && ks == 1;
    const bool stage_lin = taps_ == 1 && (!p.ln_gamma || p.C <= DGQ_LN_MAX_C);
    if (table && p.klds && p.C % 4 == 0 && strip_bytes <= 16 * 1024 && (stage_conv || stage_lin)) {
        const int nw = 4;                                                       // waves (= rows) per block, <= 64 KB of LDS
        dim3 sgrid((p.M + nw - 1) / nw, ks), sblock(64 * nw);
        if (per_m) hipLaunchKernelGGL((quant_act_staged_kernel<TIn, true>), sgrid, sblock, nw * strip_bytes, st, p);
        else hipLaunchKernelGGL((quant_act_staged_kernel<TIn, false>), sgrid, sblock, nw * strip_bytes, st, p);
        return;
    }
    dim3 grid((p.M + 3) / 4, ks), block(256);
    if (table) {
        if (per_m) hipLaunchKernelGGL((quant_act_kernel<TIn, true, true>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((quant_act_kernel<TIn, true, false>), grid, block, 0, st, p);
    } else {
        if (per_m) hipLaunchKernelGGL((quant_act_kernel<TIn, false, true>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((quant_act_kernel<TIn, false, false>), grid, block, 0, st, p);
    }
}

extern "C" int dgq_quant_act_parts(int Kp, int ksplits) {
    if (ksplits < 1) ksplits = 1;
    const int per = (((Kp + ksplits - 1) / ksplits) + 255) / 256 * 256;
    return (Kp + per - 1) / per;
}

extern "C" int dgq_quant_act(const void* x, int x_dtype, int B, int H, int W, int C,
                             int kh, int kw, int stride, int pad,
                             const int32_t* ksrc, const int32_t* koff, const int32_t* klds, int Kp,
                             int per_m, const float* delta, const float* zp, int L,
                             int bits, int8_t* codes, float* rowsum, int ksplits,
                             const float* pre_scale, const float* pre_shift, int pre_act,
                             const float* ln_gamma, const float* ln_beta, float ln_eps, void* stream) {
    DGQ_CHECK_ARG(x && delta && zp && codes && rowsum, "dgq_quant_act: null pointer");
    DGQ_CHECK_ARG(B > 0 && H > 0 && W > 0 && C > 0 && kh > 0 && kw > 0 && stride > 0 && pad >= 0,
                  "dgq_quant_act: bad geometry");
    DGQ_CHECK_ARG(kh <= 0x7F && kw <= 0xFF && C <= 0xFFFF, "dgq_quant_act: kernel/channel count out of range");
    DGQ_CHECK_ARG(Kp > 0 && Kp % DGQ_KTILE == 0, "dgq_quant_act: Kp=%d must be a multiple of %d", Kp, DGQ_KTILE);
    DGQ_CHECK_ARG(bits >= 2 && bits <= 8, "dgq_quant_act: bits=%d", bits);
    DGQ_CHECK_ARG(!per_m || L >= 1, "dgq_quant_act: per_m needs L >= 1");
    DGQ_CHECK_ARG(ksplits >= 1 && ksplits <= 64, "dgq_quant_act: ksplits=%d", ksplits);
    DGQ_CHECK_ARG((pre_scale == nullptr) == (pre_shift == nullptr) && pre_act >= 0 && pre_act <= 2, "dgq_quant_act: bad prologue");
    DGQ_CHECK_ARG(pre_act != 2 || (kh == 1 && kw == 1 && !pre_scale), "dgq_quant_act: GEGLU prologue is for Linear inputs");
    DGQ_CHECK_ARG((ln_gamma == nullptr) == (ln_beta == nullptr), "dgq_quant_act: LayerNorm prologue needs gamma and beta");
    DGQ_CHECK_ARG(!ln_gamma || (kh == 1 && kw == 1 && !pre_scale && pre_act == 0 && C % 4 == 0 && C <= DGQ_LN_MAX_C && ln_eps > 0.0f),
                  "dgq_quant_act: LayerNorm prologue is for Linear inputs (1x1, C %% 4 == 0, C <= 2048, no other prologue)");
    int K = C * kh * kw;
    if (!ksrc) {
        DGQ_CHECK_ARG(C % 4 == 0, "dgq_quant_act: natural K order needs C %% 4 == 0 (C=%d)", C);
        DGQ_CHECK_ARG(Kp >= K, "dgq_quant_act: natural K order needs Kp >= K");
    }
    int Ho = (H + 2 * pad - kh) / stride + 1, Wo = (W + 2 * pad - kw) / stride + 1;
    DGQ_CHECK_ARG(Ho > 0 && Wo > 0, "dgq_quant_act: empty output");
    QuantActParams p;
    p.x = x; p.B = B; p.H = H; p.W = W; p.C = C; p.kh = kh; p.kw = kw; p.stride = stride; p.pad = pad;
    p.Ho = Ho; p.Wo = Wo; p.ksrc = ksrc; p.koff = ksrc ? koff : nullptr; p.klds = ksrc ? klds : nullptr; p.Kp = Kp; p.K = K; p.delta = delta; p.zp = zp; p.L = per_m ? L : 1;
    p.qmax = (float)((1 << bits) - 1);
    p.offset = (float)(1 << (bits - 1));
    p.codes = codes; p.rowsum = rowsum; p.M = B * Ho * Wo;
    p.kp_per_split = (((Kp + ksplits - 1) / ksplits) + 255) / 256 * 256;
    p.pre_scale = pre_scale; p.pre_shift = pre_shift; p.pre_act = pre_act;
    p.ln_gamma = ln_gamma; p.ln_beta = ln_beta; p.ln_eps = ln_eps;
    p.ldc = pre_act == 2 ? 2 * C : C;
    hipStream_t st = (hipStream_t)stream;
    switch (x_dtype) {
        case DGQ_F32: launch_quant_act<float>(p, ksrc != nullptr, per_m != 0, st); break;
        case DGQ_F16: launch_quant_act<__half>(p, ksrc != nullptr, per_m != 0, st); break;
        case DGQ_BF16: launch_quant_act<__hip_bfloat16>(p, ksrc != nullptr, per_m != 0, st); break;
        default: dgq_set_error("dgq_quant_act: unknown dtype %d", x_dtype); return DGQ_EINVAL;
    }
    return dgq_launch_status("dgq_quant_act");
}
