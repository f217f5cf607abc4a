// Activation quantise-on-load pre-pass: channels-last fp tensor -> int8 codes of the (implicit) unfolded
// operand in the packed weight's K order, + one float per row.  HBM-bound; one wave per output row, 4 consecutive
// kp (one packed dword) per lane per 256-wide step; rounding bit-identical to fp32 true division (see below).
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include "dgq_common.h"
#include "quant_common.h"
#include "diag.h"

DGQ_DIAG_BUFFER(quant)

struct QuantActParams {
    const void* x;
    int B, H, W, C, kh, kw, stride, pad, Ho, Wo;
    const int32_t* ksrc;      // [Kp] (dh<<24 | dw<<16 | c) or -1, or NULL (natural order kp = tap*C + c)
    const int32_t* koff;      // optional [Kp]: (dh*W + dw)*ldc + c for THIS geometry, -1 for padding (interior rows)
    const int32_t* klds;      // optional [Kp]: (dh*kw + dw)*C + c, -1 for padding (LDS-staged conv path)
    const int32_t* kdst;      // optional [taps*C]: packed position kp of element (tap, c) — the inverse of ksrc (scatter path)
    const int32_t* kpat;      // optional [Kp]: (dh*PW + dw)*C + c inside the input patch of a conv tile (PW of dgq_quant_act_conv_tile), -1 padding
    int Kp, K;
    const float* delta;       // per_m: [L]; else [Kp/32]
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
    int ups;                  // 1: x holds (H/2) x (W/2) pixels per image and pixel (hi, wi) of the H x W input reads (hi/2, wi/2) — a 2x nearest
                              // upsample in front of the layer (Upsample2D) folded into the load; scatter and block-staged conv paths only
};

// Up to DGQ_QA_BATCH problems of ONE kernel variant and the same row count in one launch (blockIdx.z = problem): the
// q / k / v projections of an attention quantise the same input three ways, the to_k / to_v of every cross-attention
// quantise the same text context — one launch each instead of one per layer.
#define DGQ_QA_BATCH 8
struct QuantActBatch {
    QuantActParams p[DGQ_QA_BATCH];
};
// Shared-input form of the natural-order per-M kernel (NSH = 2..4 problems that differ ONLY in their quantiser tables and outputs —
// the q / k / v projections of a self-attention under scalar / per-token scales): one wave loads its row and applies the folded
// prologue once, then quantises and stores it NSH times, instead of NSH workgroups each reading the row again.
#define DGQ_QA_SHARE 4



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

// One wave per output row; each lane owns 4 consecutive kp per 256-wide step (one packed dword), so that the
// table read (int4), the gathered loads (lane stride 16 B within a (group, tap) run) and the code store (256 B per
// wave instruction) are all coalesced.  The 4 kp of a lane share one 32-wide chunk, hence one (δ, z).
template <typename TIn, bool HAS_TABLE, bool PER_M, int NSH = 1>
__global__ __launch_bounds__(256) void quant_act_kernel(QuantActBatch bt) {
    static_assert(NSH == 1 || (!HAS_TABLE && PER_M), "shared-input form: natural order, per-M tables");
    const QuantActParams& p = bt.p[NSH > 1 ? 0 : blockIdx.z];
    if ((int)blockIdx.y * p.kp_per_split >= p.Kp) return;     // this problem has fewer K splits than the widest of the batch
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.M) return;                                 // whole wave leaves; no barriers below
    DGQ_DIAG_DECL
    DGQ_STAMP(0); DGQ_STAMP_REAL(1); DGQ_STAMP_WHERE(2);
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
    // problems 1 .. NSH-1 of the shared-input form: their own table entry, code range and outputs
    float mdS[NSH], mzS[NSH], minvS[NSH], qmaxS[NSH], biasS[NSH], partS[NSH];
    uint32_t* outS[NSH];
#pragma unroll
    for (int q = 1; q < NSH; ++q) {
        const QuantActParams& ps = bt.p[q];
        const int li = row % ps.L;
        mdS[q] = ps.delta[li];
        mzS[q] = ps.zp[li];
        minvS[q] = dgq_rcp(mdS[q]);
        qmaxS[q] = ps.qmax;
        biasS[q] = 128.0f - ps.offset;
        partS[q] = 0.0f;
        outS[q] = reinterpret_cast<uint32_t*>(ps.codes + (int64_t)row * ps.Kp);
    }
    float partial = 0.0f;
    float ln_mu = 0.0f, ln_rstd = 1.0f;
    if (p.ln_gamma) row_layernorm_stats<TIn>(img + rowoff, p.C, p.ln_eps, lane, ln_mu, ln_rstd);   // 1x1: the row itself
    uint32_t* out = reinterpret_cast<uint32_t*>(p.codes + (int64_t)row * p.Kp);
    DGQ_STAMP(3);
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
                        d = p.delta[kp0 >> 5];
                        z = p.zp[kp0 >> 5];
                        inv = dgq_rcp(d);
                    }
                    float biased[4], qv[4], fsum = 0.0f;
                    dgq_affine_code4_fast(v[u], d, inv, z, p.qmax, qv);
#pragma unroll
                    for (int j = 0; j < 4; ++j) biased[j] = idx[u][j] != -1 ? qv[j] + bias : 128.0f;
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
            d = p.delta[kp0 >> 5];
            z = p.zp[kp0 >> 5];
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
        float biased[4], qv[4], fsum = 0.0f;
        dgq_affine_code4_fast(v, d, inv, z, p.qmax, qv);
#pragma unroll
        for (int j = 0; j < 4; ++j) biased[j] = in_k ? qv[j] + bias : 128.0f;
        out[kp0 >> 2] = dgq_pack4(biased, fsum);
        fsum -= 512.0f;
        partial += PER_M ? fsum : d * fsum;
#pragma unroll
        for (int q = 1; q < NSH; ++q) {
            float bq[4], qs[4], fs = 0.0f;
            dgq_affine_code4_fast(v, mdS[q], minvS[q], mzS[q], qmaxS[q], qs);
#pragma unroll
            for (int j = 0; j < 4; ++j) bq[j] = in_k ? qs[j] + biasS[q] : 128.0f;
            outS[q][kp0 >> 2] = dgq_pack4(bq, fs);
            partS[q] += fs - 512.0f;
        }
    }
    }
    DGQ_STAMP(4);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) partial += __shfl_down(partial, o, 64);
    if (lane == 0) p.rowsum[(int64_t)blockIdx.y * p.M + row] = partial;
#pragma unroll
    for (int q = 1; q < NSH; ++q) {
        float ps_ = partS[q];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ps_ += __shfl_down(ps_, o, 64);
        if (lane == 0) bt.p[q].rowsum[(int64_t)blockIdx.y * bt.p[q].M + row] = ps_;
    }
    DGQ_STAMP(9);
    DGQ_DIAG_DRAIN();
    DGQ_STAMP(10); DGQ_STAMP_REAL(11);
    DGQ_DIAG_FLUSH(quant, 4, threadIdx.x >> 6, lane);
}

// Per-K conv layers (the quantizer sees the unfolded operand, so each (c, tap) may carry its own group): the table
// gathers 4-byte elements scattered over the kh·kw pixel rows of this output position.  Straight from global memory
// that is one L1 access per element (measured: TCP_TOTAL_CACHE_ACCESSES = 32 per wave-instruction, the kernel runs at
// the L1's access rate, ~50 us for 8192 x 2880); here each wave first copies its taps — C contiguous floats each, with
// GroupNorm / SiLU applied and zeros for taps outside the image — into its own LDS strip with coalesced 16-byte loads
// and gathers from LDS.  No block-level sync: a wave only reads what it wrote.
template <typename TIn, bool PER_M>
__global__ __launch_bounds__(256) void quant_act_staged_kernel(QuantActBatch bt) {
    const QuantActParams& p = bt.p[blockIdx.z];
    if ((int)blockIdx.y * p.kp_per_split >= p.Kp) return;
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
                    d = p.delta[kp0 >> 5];
                    z = p.zp[kp0 >> 5];
                    inv = dgq_rcp(d);
                }
                float biased[4], qv[4], fsum = 0.0f;
                dgq_affine_code4_fast(v[u], d, inv, z, p.qmax, qv);
#pragma unroll
                for (int j = 0; j < 4; ++j) biased[j] = idx[u][j] >= 0 ? qv[j] + bias : 128.0f;
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

// Per-K conv layers with many input channels (3x3 on C >= 640, or any layer whose tap strips do not fit a wave's LDS
// share): the gather above reads one scattered 4-byte element per code straight from global memory and runs at the L1's
// access rate (8192 x 9088: 257 us, 0.3 TB/s).  Turned around: every tap's C contiguous channels are READ coalesced
// (16 bytes per lane, GroupNorm / SiLU applied on the way), quantised with the (δ, 1/δ, z) of the element's destination
// chunk (tables staged in LDS once per block), and the code byte is SCATTERED into the row's image in LDS at its packed
// position kdst[tap][c]; the finished row leaves LDS as 16-byte coalesced stores.  RPB rows per block: 4 (one wave per
// row) for large M; 1 (the four waves take the taps round-robin and share the row image) where M alone cannot fill the chip.
// (round 4) RPB == 1 runs NWV = 8 or 16 waves per row: with four, a wave walked up to 3 taps x C/256 dependent load rounds (16.6 us for
// 128 x 11904 codes against 6.0 us for the per-M form of the same layer); the (tap, 256-channel step) units are dealt round-robin instead.
template <typename TIn, int RPB, int NWV = 4>
__global__ __launch_bounds__(64 * NWV) void quant_act_scatter_kernel(QuantActBatch bt) {
    static_assert(RPB == 1 || NWV == 4 || NWV == 8, "four rows per block: one or two waves each");
    const QuantActParams& p = bt.p[blockIdx.z];
    extern __shared__ __attribute__((aligned(16))) uint8_t sc_smem[];
    constexpr int WPR = NWV / RPB;                           // waves per row
    constexpr int NT_ = 64 * NWV;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    DGQ_DIAG_DECL
    DGQ_STAMP(0); DGQ_STAMP_REAL(1); DGQ_STAMP_WHERE(2);
    const int nch = p.Kp >> 5;
    float* tdelta = reinterpret_cast<float*>(sc_smem);
    float* tinv = tdelta + nch;
    float* tzp = tinv + nch;
    float* psum = tzp + nch;                                 // [16] per-wave partial row sums (RPB == 1)
    uint8_t* images = sc_smem + (((3 * nch + 16) * 4 + 15) & ~15);
    for (int i = tid; i < nch; i += NT_) {
        const float d = p.delta[i];
        tdelta[i] = d; tinv[i] = dgq_rcp(d); tzp[i] = p.zp[i];
    }
    for (int i = tid; i < RPB * (p.Kp >> 4); i += NT_) reinterpret_cast<uint4*>(images)[i] = make_uint4(0, 0, 0, 0);   // padding = code 0
    __syncthreads();
    DGQ_STAMP(3);
    const int rslot = wv / WPR, wsub = wv % WPR;
    const int row = blockIdx.x * RPB + rslot;
    uint8_t* image = images + (size_t)rslot * p.Kp;
    float partial = 0.0f;
    if (row < p.M) {
        const TIn* x = reinterpret_cast<const TIn*>(p.x);
        const int L = p.Ho * p.Wo;
        const int b = row / L, l = row - b * L;
        const int ho = l / p.Wo, wo = l - ho * p.Wo;
        const int hbase = ho * p.stride - p.pad, wbase = wo * p.stride - p.pad;
        const int us = p.ups, Ws = p.W >> us;               // source geometry: (H >> ups) x (W >> ups) pixels per image
        const TIn* img = x + (int64_t)b * (p.H >> us) * Ws * p.ldc;
        const float* pre_sc = p.pre_scale ? p.pre_scale + (int64_t)b * p.C : nullptr;
        const float* pre_sh = p.pre_shift ? p.pre_shift + (int64_t)b * p.C : nullptr;
        const int taps = p.kh * p.kw;
        float ln_mu = 0.0f, ln_rstd = 1.0f;                 // Linear inputs (taps == 1, one wave per row): LayerNorm over the row
        if (p.ln_gamma) row_layernorm_stats<TIn>(img + (int64_t)(hbase * p.W + wbase) * p.ldc, p.C, p.ln_eps, lane, ln_mu, ln_rstd);
        DGQ_STAMP(4);
        // one (tap, 256-channel step) unit of the row
        auto unit = [&](int tap, int c) {
            const int dh = tap / p.kw, dw = tap - dh * p.kw;
            const int hi = hbase + dh, wi = wbase + dw;
            const bool inb = hi >= 0 && hi < p.H && wi >= 0 && wi < p.W;     // wave-uniform
            const TIn* src = img + ((int64_t)(hi >> us) * Ws + (wi >> us)) * p.ldc;
            const int32_t* kd = p.kdst + tap * p.C;
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
                const int4 d4 = *reinterpret_cast<const int4*>(kd + c);
                const int dst[4] = {d4.x, d4.y, d4.z, d4.w};
                float dd[4], di[4], dz[4], qv[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int ch = dst[j] >> 5;
                    dd[j] = tdelta[ch]; di[j] = tinv[ch]; dz[j] = tzp[ch];
                }
                dgq_affine_code4_fast(v, dd, di, dz, p.qmax, qv);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float sc = qv[j] - p.offset;
                    image[dst[j]] = (uint8_t)(int)sc;
                    partial += dd[j] * sc;
                }
        };
        if constexpr (WPR == 1) {                            // one wave per row: taps in order, channel steps inside (loads of a tap overlap)
            for (int tap = 0; tap < taps; ++tap)
                for (int c = lane * 4; c < p.C; c += 256) unit(tap, c);
        } else {                                             // several waves per row: the units dealt round-robin
            const int csteps = (p.C + 255) >> 8;
            for (int u = wsub; u < taps * csteps; u += WPR) {
                const int tap = u / csteps;
                const int c = ((u - tap * csteps) << 8) + lane * 4;
                if (c < p.C) unit(tap, c);
            }
        }
    }
    DGQ_STAMP(5);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) partial += __shfl_down(partial, o, 64);
    if ((RPB == 1 || WPR > 1) && lane == 0) psum[wv] = partial;
    __syncthreads();                                        // every wave's bytes of the shared image(s) are in LDS
    DGQ_STAMP(6);
    if (row < p.M) {
        uint4* out = reinterpret_cast<uint4*>(p.codes + (int64_t)row * p.Kp);
        const uint4* im = reinterpret_cast<const uint4*>(image);
        for (int i = wsub * 64 + lane; i < (p.Kp >> 4); i += 64 * WPR) out[i] = im[i];
        if (lane == 0 && wsub == 0) {
            float tot = partial;
            if (RPB == 1) {                                  // fixed order: pairs, then a chain
                tot = 0.0f;
#pragma unroll
                for (int w = 0; w < NWV; w += 2) tot += psum[w] + psum[w + 1];
            } else if (WPR > 1) {                            // the row's waves, in wave order
                tot = psum[rslot * WPR];
#pragma unroll
                for (int w = 1; w < WPR; ++w) tot += psum[rslot * WPR + w];
            }
            p.rowsum[row] = tot;
        }
    }
    DGQ_STAMP(9);
    DGQ_DIAG_DRAIN();
    DGQ_STAMP(10); DGQ_STAMP_REAL(11);
    DGQ_DIAG_FLUSH(quant, NWV, wv, lane);
}

// Convolutions (kh·kw > 1), block-staged.  The quantizer sees the UNFOLDED operand, so every input element is quantised once
// per tap that reads it — but the GroupNorm scale/shift and the SiLU in front of it depend on the element alone.  The per-row
// kernels above applied them once per (row, tap) (9x per element for a 3x3; the 64x64-level convs were VALU-bound: 55 us for
// 8192 x 2880 codes).  Here a workgroup owns a TH x TW tile of output positions: it stages the tile's input patch
// ((TH−1)·stride + kh) x ((TW−1)·stride + kw) pixels x C channels ONCE in LDS as fp32, prologue applied (zeros for pixels
// outside the image: F.unfold pads before the quantizer), 16-byte coalesced reads; then every wave takes output rows and
// gathers its codes from the patch through the kpat table (lane = 4 consecutive kp, one packed dword, 256-byte row
// segments per store), with the same exact-division quantiser.  Per-K (kpat in the weight's group-sorted K order, one (δ, z)
// per 32-chunk) and per-M / scalar (natural order, one (δ, z) per row) alike.
template <typename TIn, bool PER_M, int TH, int TW, int NW>
__global__ __launch_bounds__(64 * NW) void quant_act_conv_kernel(QuantActBatch bt) {
    const QuantActParams& p = bt.p[blockIdx.z];
    extern __shared__ __attribute__((aligned(16))) float patch[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_w = (p.Wo + TW - 1) / TW, tiles_h = (p.Ho + TH - 1) / TH;
    int t = blockIdx.x;
    const int b = t / (tiles_h * tiles_w);
    t -= b * tiles_h * tiles_w;
    const int th = t / tiles_w, tw = t - th * tiles_w;
    if (b >= p.B) return;
    const int ho0 = th * TH, wo0 = tw * TW;
    const int PH = (TH - 1) * p.stride + p.kh, PW = (TW - 1) * p.stride + p.kw;
    const int hi0 = ho0 * p.stride - p.pad, wi0 = wo0 * p.stride - p.pad;
    const int us = p.ups, Ws = p.W >> us;                   // source geometry: (H >> ups) x (W >> ups) pixels per image
    const TIn* img = reinterpret_cast<const TIn*>(p.x) + (int64_t)b * (p.H >> us) * Ws * p.C;
    const float* pre_sc = p.pre_scale ? p.pre_scale + (int64_t)b * p.C : nullptr;
    const float* pre_sh = p.pre_shift ? p.pre_shift + (int64_t)b * p.C : nullptr;
    // ---- stage the patch: wave w takes pixels w, w + NW, ...
    for (int pp = wv; pp < PH * PW; pp += NW) {
        const int ph = pp / PW, pw_ = pp - ph * PW;
        const int hi = hi0 + ph, wi = wi0 + pw_;
        const bool inb = hi >= 0 && hi < p.H && wi >= 0 && wi < p.W;       // wave-uniform
        const TIn* src = img + ((int64_t)(hi >> us) * Ws + (wi >> us)) * p.C;      // (ups: the four pixels of a 2 x 2 cell read one source pixel)
        float* dst = patch + pp * p.C;
        for (int c = lane * 4; c < p.C; c += 256) {
            float v[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            if (inb) {
                load4<TIn>(src + c, v);
                if (pre_sc) {
                    const float4 sc = *reinterpret_cast<const float4*>(pre_sc + c);
                    const float4 sh = *reinterpret_cast<const float4*>(pre_sh + c);
                    v[0] = v[0] * sc.x + sh.x; v[1] = v[1] * sc.y + sh.y; v[2] = v[2] * sc.z + sh.z; v[3] = v[3] * sc.w + sh.w;
                }
                if (p.pre_act == 1) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = dgq_silu(v[j]);
                }
            }
            *reinterpret_cast<float4*>(dst + c) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
    // the gather table (and the per-chunk quantiser tables) behind the patch: read once per workgroup, every row of the tile
    // then finds them at LDS latency (read from L1 per row, their ~1 us round trips were exposed at two waves per SIMD)
    // (16-bit entries: a patch holds < 2^16 floats; 0xFFFF = padding)
    uint16_t* tab = reinterpret_cast<uint16_t*>(patch + PH * PW * p.C);
    float* tdl = reinterpret_cast<float*>(tab + p.Kp);
    float* tzp = tdl + (p.Kp >> 5);
    for (int k = tid * 4; k < p.Kp; k += 4 * 64 * NW) {
        const int4 e = *reinterpret_cast<const int4*>(p.kpat + k);
        *reinterpret_cast<uint2*>(tab + k) = make_uint2(((uint32_t)e.x & 0xFFFFu) | ((uint32_t)e.y << 16), ((uint32_t)e.z & 0xFFFFu) | ((uint32_t)e.w << 16));
    }
    if (!PER_M)
        for (int c = tid; c < (p.Kp >> 5); c += 64 * NW) { tdl[c] = p.delta[c]; tzp[c] = p.zp[c]; }
    __syncthreads();
    // ---- gather + quantise: wave w takes output positions w, w + NW, ... of the tile; with more waves than positions (WPR waves per
    // position) a position's 1024-code steps are dealt round-robin over its waves and the row sum is the sum of their parts in wave order
    constexpr int WPR = (NW > TH * TW) ? NW / (TH * TW) : 1;
    static_assert(WPR == 1 || NW == WPR * TH * TW, "waves per position: a whole number");
    __shared__ float rs_part[WPR > 1 ? TH * TW * WPR : 1];
    const int part = WPR > 1 ? wv % WPR : 0;
    const float bias = 128.0f - p.offset;
    for (int r = wv / WPR; r < TH * TW; r += NW / WPR) {
        const int i = r / TW, j = r - i * TW;
        const int ho = ho0 + i, wo = wo0 + j;
        if (WPR == 1 && (ho >= p.Ho || wo >= p.Wo)) continue;                 // wave-uniform
        const bool live = ho < p.Ho && wo < p.Wo;                             // (WPR > 1: every wave reaches the barrier below)
        const int row = (b * p.Ho + min(ho, p.Ho - 1)) * p.Wo + min(wo, p.Wo - 1);
        const float* pr = patch + ((i * p.stride) * PW + j * p.stride) * p.C;
        float md = 1.0f, mz = 0.0f, minv = 1.0f;
        if (PER_M) {
            const int li = row % p.L;
            md = p.delta[li];
            mz = p.zp[li];
            minv = dgq_rcp(md);
        }
        float partial = 0.0f;
        uint32_t* out = reinterpret_cast<uint32_t*>(p.codes + (int64_t)row * p.Kp);
        for (int kb = lane * 4 + 1024 * part; kb < (live ? p.Kp : 0); kb += 1024 * WPR) {
            int idx[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int kp0 = kb + 256 * u;
                uint2 tt = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
                if (kp0 < p.Kp) tt = *reinterpret_cast<const uint2*>(tab + kp0);
                idx[u][0] = tt.x & 0xFFFF; idx[u][1] = tt.x >> 16; idx[u][2] = tt.y & 0xFFFF; idx[u][3] = tt.y >> 16;
            }
            float v[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int q = 0; q < 4; ++q) v[u][q] = pr[idx[u][q] == 0xFFFF ? 0 : idx[u][q]];   // padding reads element 0: value unused
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int kp0 = kb + 256 * u;
                if (kp0 < p.Kp) {
                    float d = md, z = mz, inv = minv;
                    if (!PER_M) {
                        d = tdl[kp0 >> 5];
                        z = tzp[kp0 >> 5];
                        inv = dgq_rcp(d);
                    }
                    float biased[4], qv[4], fsum = 0.0f;
                    dgq_affine_code4_fast(v[u], d, inv, z, p.qmax, qv);
#pragma unroll
                    for (int q = 0; q < 4; ++q) biased[q] = idx[u][q] != 0xFFFF ? qv[q] + bias : 128.0f;
                    out[kp0 >> 2] = dgq_pack4(biased, fsum);
                    fsum -= 512.0f;
                    partial += PER_M ? fsum : d * fsum;
                }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) partial += __shfl_down(partial, o, 64);
        if constexpr (WPR == 1) {
            if (lane == 0) p.rowsum[row] = partial;
        } else {
            if (lane == 0) rs_part[r * WPR + part] = partial;
            __syncthreads();                                                  // (one position per wave group: every wave gets here exactly once)
            if (lane == 0 && part == 0 && live) {
                float tot = rs_part[r * WPR];
#pragma unroll
                for (int q = 1; q < WPR; ++q) tot += rs_part[r * WPR + q];
                p.rowsum[row] = tot;
            }
        }
    }
}

// tile of the block-staged conv path for a geometry: the largest of 4x8 / 4x4 / 2x4 output positions whose input patch fits
// the LDS (two workgroups per CU for the first, one for the others); 0 = none (the per-row paths take the layer)
struct ConvTile { int id, th, tw, pw; size_t lds; };
static ConvTile conv_tile(int C, int kh, int kw, int stride, int Kp) {
    const int cand[3][2] = {{4, 8}, {4, 4}, {2, 4}};
    const size_t tables = (size_t)Kp * 2 + (size_t)(Kp >> 5) * 8 + 64;
    ConvTile fit = {0, 0, 0, 0, 0};
    for (int i = 0; i < 3; ++i) {
        const int th = cand[i][0], tw = cand[i][1];
        const int ph = (th - 1) * stride + kh, pw = (tw - 1) * stride + kw;
        const size_t lds = (size_t)ph * pw * C * sizeof(float) + tables;
        if (lds <= 78u * 1024) return {i + 1, th, tw, pw, lds};           // two workgroups per CU: staging of one overlaps the other's gather
        if (fit.id == 0 && lds <= 150u * 1024) fit = {i + 1, th, tw, pw, lds};
    }
    return fit;
}
// Where the path pays (tools/bench_qact_conv.py, profiles/r03_quant_act_conv_block.txt): the tile grid must cover the chip
// (>= 256 workgroups), i.e. the 64x64-level convolutions and the wide 32x32 ones; on smaller grids the per-row paths, whose
// workgroups hold four rows, keep more of the chip busy.
static long conv_tiles(const QuantActParams& p, const ConvTile& t) {
    return (long)p.B * ((p.Ho + t.th - 1) / t.th) * ((p.Wo + t.tw - 1) / t.tw);
}
// ... for a geometry: where the 4 x 4 tile leaves the grid short of the chip (2048 positions: 128 tiles) the 2 x 4 tile — the same patch
// width, hence the same kpat table — covers it (the 320 -> 640 convolution of SD's 32 x 32 level: 29.8 us on the global-gather path,
// which the tile rule used to leave it on, against 17 us here)
static ConvTile conv_tile_geo(const QuantActParams& p) {
    ConvTile t = conv_tile(p.C, p.kh, p.kw, p.stride, p.Kp);
    if (t.id == 2 && conv_tiles(p, t) < 256) {
        const int ph = (2 - 1) * p.stride + p.kh;
        ConvTile h = {3, 2, 4, t.pw, (size_t)ph * t.pw * p.C * sizeof(float) + ((size_t)p.Kp * 2 + (size_t)(p.Kp >> 5) * 8 + 64)};
        if (conv_tiles(p, h) >= 256) return h;
    }
    return t;
}
static bool conv_block_pays(const QuantActParams& p, const ConvTile& t) {
    if (t.id == 0) return false;
    return conv_tiles(p, t) >= 256 && p.M >= 2048;
}

extern "C" int dgq_quant_act_conv_tile(int C, int kh, int kw, int stride, int Kp, int* patch_w) {
    if (C <= 0 || kh <= 0 || kw <= 0 || stride <= 0 || kh * kw <= 1 || C % 4 != 0 || Kp <= 0) return 0;
    const ConvTile t = conv_tile(C, kh, kw, stride, Kp);
    if (patch_w) *patch_w = t.pw;
    return t.id;
}

// kernel variant of one problem: 0 = LDS-staged strips, 1 = table gather from global, 2 = natural order, 3 / 4 = scatter
// through an LDS row image with 4 rows / 1 row per block
static int quant_act_variant(const QuantActParams& p, bool table) {
    const int ks = (p.Kp + p.kp_per_split - 1) / p.kp_per_split;
    const size_t strip_bytes = (size_t)p.kh * p.kw * p.C * sizeof(float);
    // measured (SD1.4 layers): staging wins at C = 320 (4 rows per block, 3 blocks per CU: 88 -> 64 us) and loses once the
    // strips cut occupancy (C >= 640 at 3x3) or the K range is split over blocks that would each re-stage every tap.
    // Linear inputs (one tap: the strip is the row itself, <= 16 KB) are always staged — the global gather pays one L1
    // access per code (8192 x 320 -> Kp 1024: 20 us) — together with their LayerNorm / GEGLU prologue.
    const int taps_ = p.kh * p.kw;
    // per-K Linear inputs whose groups are short (K = 320 with 16 groups: Kp = 512, 38 % padding): the staged gather
    // evaluates the quantiser for every PACKED position, padding included; the scatter path evaluates it once per SOURCE
    // element (K instead of Kp codes, 16-byte coalesced reads with the LayerNorm / GEGLU prologue applied on the way) and
    // drops the byte into the row image in LDS.  Taken where the padding is at least a quarter of K.
    {
        const size_t tab_b = (((size_t)3 * (p.Kp >> 5) + 16) * 4 + 15) & ~(size_t)15;
        if (table && p.kdst && taps_ == 1 && p.C % 4 == 0 && ks == 1 && (!p.ln_gamma || p.C <= DGQ_LN_MAX_C) &&
            tab_b + 4 * (size_t)p.Kp <= 150 * 1024 && 4 * p.Kp >= 5 * p.K)
            return 3;
    }
    // convolutions whose input patch fits the LDS: the block-staged path
    {
        if (p.kpat && taps_ > 1 && p.C % 4 == 0 && ks == 1 && p.pre_act != 2 && !p.ln_gamma &&
            conv_block_pays(p, conv_tile_geo(p)))
            return 5;
    }
    const bool stage_conv = taps_ > 1 && p.pre_act != 2 && !p.ln_gamma && ks == 1;
    const bool stage_lin = taps_ == 1 && (!p.ln_gamma || p.C <= DGQ_LN_MAX_C);
    if (table && p.klds && p.C % 4 == 0 && strip_bytes <= 16 * 1024 && (stage_conv || stage_lin)) return 0;
    // scatter path: per-K table, more than one tap, the whole row in one wave/block (no K split), no LN / GEGLU prologue
    const size_t sc_tab = (((size_t)3 * (p.Kp >> 5) + 16) * 4 + 15) & ~(size_t)15;
    if (table && p.kdst && taps_ > 1 && p.C % 4 == 0 && ks == 1 && p.pre_act != 2 && !p.ln_gamma && sc_tab + (size_t)p.Kp <= 150 * 1024)
        return (p.M >= 4096 && sc_tab + 4 * (size_t)p.Kp <= 150 * 1024) ? 3 : 4;     // (2048 rows x 17536: 79 us with one wave per row, see below)
    // per-K Linear inputs too wide for a staged strip (K > 4096: SDXL's ff.net.2 at 5120): the multi-wave scatter form instead of the
    // global gather (one L1 access per code)
    {
        const size_t tab_b = (((size_t)3 * (p.Kp >> 5) + 16) * 4 + 15) & ~(size_t)15;
        if (table && p.kdst && taps_ == 1 && p.C % 4 == 0 && ks == 1 && !p.ln_gamma && p.pre_act != 2 && tab_b + (size_t)p.Kp <= 150 * 1024)
            return 4;
    }
    return table ? 1 : 2;
}

// all n problems: same variant, same per_m, same M (checked by the caller)
template <typename TIn>
static void launch_quant_act(const QuantActBatch& bt, int n, int variant, bool per_m, hipStream_t st) {
    const QuantActParams& p0 = bt.p[0];
    int ks = 1;
    size_t strip_bytes = 0;
    for (int i = 0; i < n; ++i) {
        const QuantActParams& p = bt.p[i];
        ks = std::max(ks, (p.Kp + p.kp_per_split - 1) / p.kp_per_split);
        strip_bytes = std::max(strip_bytes, (size_t)p.kh * p.kw * p.C * sizeof(float));
    }
    if (variant == 5) {
        const ConvTile t = conv_tile_geo(p0);
        const int tiles = p0.B * ((p0.Ho + t.th - 1) / t.th) * ((p0.Wo + t.tw - 1) / t.tw);
        static std::atomic<bool> attr5[64];
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (dev < 0 || dev >= 64 || !attr5[dev].load(std::memory_order_acquire)) {          // all six instantiations of this dtype, once
#define DGQ_QA_ATTR(PM, TH_, TW_) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&quant_act_conv_kernel<TIn, PM, TH_, TW_, 8>), \
                                                            hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024)
            DGQ_QA_ATTR(true, 4, 8); DGQ_QA_ATTR(false, 4, 8);
#undef DGQ_QA_ATTR
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&quant_act_conv_kernel<TIn, true, 4, 4, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&quant_act_conv_kernel<TIn, false, 4, 4, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&quant_act_conv_kernel<TIn, true, 2, 4, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&quant_act_conv_kernel<TIn, false, 2, 4, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
            if (dev >= 0 && dev < 64) attr5[dev].store(true, std::memory_order_release);
        }
#define DGQ_QA_CONV(PM, TH_, TW_) hipLaunchKernelGGL((quant_act_conv_kernel<TIn, PM, TH_, TW_, 8>), dim3(tiles, 1, n), dim3(512), t.lds, st, bt)
        if (t.id == 1) { if (per_m) DGQ_QA_CONV(true, 4, 8); else DGQ_QA_CONV(false, 4, 8); }
        else if (t.id == 2) {
            // 16 positions per tile: one wave per output position (16 waves) instead of two positions per wave — the gather / quantise phase is a
            // chain of dependent LDS reads per 1024 codes, and twice the waves hide twice the latency (same lanes, same order per row:
            // bit-identical; step +0.3-0.5 % same box)
            if (per_m) hipLaunchKernelGGL((quant_act_conv_kernel<TIn, true, 4, 4, 16>), dim3(tiles, 1, n), dim3(1024), t.lds, st, bt);
            else hipLaunchKernelGGL((quant_act_conv_kernel<TIn, false, 4, 4, 16>), dim3(tiles, 1, n), dim3(1024), t.lds, st, bt);
        }
        else {
            // 8 positions per tile, 16 waves: two waves per position (the per-K row sum then adds its two parts: last-bit differences
            // against the one-wave order; per-M sums are exact integers)
            if (per_m) hipLaunchKernelGGL((quant_act_conv_kernel<TIn, true, 2, 4, 16>), dim3(tiles, 1, n), dim3(1024), t.lds, st, bt);
            else hipLaunchKernelGGL((quant_act_conv_kernel<TIn, false, 2, 4, 16>), dim3(tiles, 1, n), dim3(1024), t.lds, st, bt);
        }
#undef DGQ_QA_CONV
        return;
    }
    if (variant == 3 || variant == 4) {
        size_t lds = 0;
        for (int i = 0; i < n; ++i) {
            const size_t tab = (((size_t)3 * (bt.p[i].Kp >> 5) + 16) * 4 + 15) & ~(size_t)15;
            lds = std::max(lds, tab + (variant == 3 ? 4 : 1) * (size_t)bt.p[i].Kp);
        }
        static std::atomic<bool> attr_set[64];
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (dev < 0 || dev >= 64 || !attr_set[dev].load(std::memory_order_acquire)) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&quant_act_scatter_kernel<TIn, 4, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&quant_act_scatter_kernel<TIn, 1, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&quant_act_scatter_kernel<TIn, 1, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
            if (dev >= 0 && dev < 64) attr_set[dev].store(true, std::memory_order_release);
        }
        // four rows per block: two waves per row (the (tap, 256-channel) units dealt round-robin; each wave forms the row's LayerNorm
        // statistics itself) — a 2048-row launch is 2048 waves otherwise, two per SIMD, each a chain of dependent loads.
        // Rows (x problems) up to 2048 get 16 waves each instead of 8 (256 -> 2048 measured +0.5 % on the SD step, same box).
        if (variant == 3) hipLaunchKernelGGL((quant_act_scatter_kernel<TIn, 4, 8>), dim3((p0.M + 3) / 4, 1, n), dim3(512), lds, st, bt);
        else if ((long)p0.M * n <= 2048L) hipLaunchKernelGGL((quant_act_scatter_kernel<TIn, 1, 16>), dim3(p0.M, 1, n), dim3(1024), lds, st, bt);
        else hipLaunchKernelGGL((quant_act_scatter_kernel<TIn, 1, 8>), dim3(p0.M, 1, n), dim3(512), lds, st, bt);
        return;
    }
    if (variant == 0) {
        const int nw = 4;                                                       // waves (= rows) per block, <= 64 KB of LDS
        dim3 sgrid((p0.M + nw - 1) / nw, ks, n), sblock(64 * nw);
        if (per_m) hipLaunchKernelGGL((quant_act_staged_kernel<TIn, true>), sgrid, sblock, nw * strip_bytes, st, bt);
        else hipLaunchKernelGGL((quant_act_staged_kernel<TIn, false>), sgrid, sblock, nw * strip_bytes, st, bt);
        return;
    }
    // shared-input form: 2..4 natural-order per-M problems that differ only in tables and outputs (q / k / v of a self-attention)
    if (variant == 2 && per_m && n >= 2 && n <= DGQ_QA_SHARE) {
        bool same = true;
        for (int i = 1; i < n && same; ++i) {
            const QuantActParams& p = bt.p[i];
            same = p.x == p0.x && p.B == p0.B && p.H == p0.H && p.W == p0.W && p.C == p0.C && p.kh == p0.kh && p.kw == p0.kw &&
                   p.stride == p0.stride && p.pad == p0.pad && p.Kp == p0.Kp && p.K == p0.K && p.M == p0.M &&
                   p.kp_per_split == p0.kp_per_split && p.pre_scale == p0.pre_scale && p.pre_shift == p0.pre_shift &&
                   p.pre_act == p0.pre_act && p.ldc == p0.ldc && p.ln_gamma == p0.ln_gamma && p.ln_beta == p0.ln_beta &&
                   p.ln_eps == p0.ln_eps && p.codes != p0.codes;
        }
        if (same) {                                            // measured: C5 step 93.5 -> 94.6 prompt-steps/s (60 + 10 q/k/v launches per step)
            dim3 g1((p0.M + 3) / 4, ks, 1), b1(256);
            if (n == 2) hipLaunchKernelGGL((quant_act_kernel<TIn, false, true, 2>), g1, b1, 0, st, bt);
            else if (n == 3) hipLaunchKernelGGL((quant_act_kernel<TIn, false, true, 3>), g1, b1, 0, st, bt);
            else hipLaunchKernelGGL((quant_act_kernel<TIn, false, true, 4>), g1, b1, 0, st, bt);
            return;
        }
    }
    dim3 grid((p0.M + 3) / 4, ks, n), block(256);
    if (variant == 1) {
        if (per_m) hipLaunchKernelGGL((quant_act_kernel<TIn, true, true>), grid, block, 0, st, bt);
        else hipLaunchKernelGGL((quant_act_kernel<TIn, true, false>), grid, block, 0, st, bt);
    } else {
        if (per_m) hipLaunchKernelGGL((quant_act_kernel<TIn, false, true>), grid, block, 0, st, bt);
        else hipLaunchKernelGGL((quant_act_kernel<TIn, false, false>), grid, block, 0, st, bt);
    }
}

extern "C" int dgq_quant_act_parts(int Kp, int ksplits) {
    if (ksplits < 1) ksplits = 1;
    const int per = (((Kp + ksplits - 1) / ksplits) + 255) / 256 * 256;
    return (Kp + per - 1) / per;
}

static int fill_quant_act(const dgq_quant_act_args_t& a, QuantActParams& p) {
    DGQ_CHECK_ARG(a.x && a.delta && a.zp && a.codes && a.rowsum, "dgq_quant_act: null pointer");
    DGQ_CHECK_ARG(a.B > 0 && a.H > 0 && a.W > 0 && a.C > 0 && a.kh > 0 && a.kw > 0 && a.stride > 0 && a.pad >= 0,
                  "dgq_quant_act: bad geometry");
    DGQ_CHECK_ARG(a.kh <= 0x7F && a.kw <= 0xFF && a.C <= 0xFFFF, "dgq_quant_act: kernel/channel count out of range");
    DGQ_CHECK_ARG(a.Kp > 0 && a.Kp % DGQ_KTILE == 0, "dgq_quant_act: Kp=%d must be a multiple of %d", a.Kp, DGQ_KTILE);
    DGQ_CHECK_ARG(a.bits >= 2 && a.bits <= 8, "dgq_quant_act: bits=%d", a.bits);
    DGQ_CHECK_ARG(!a.per_m || a.L >= 1, "dgq_quant_act: per_m needs L >= 1");
    DGQ_CHECK_ARG(a.ksplits >= 1 && a.ksplits <= 64, "dgq_quant_act: ksplits=%d", a.ksplits);
    DGQ_CHECK_ARG((a.pre_scale == nullptr) == (a.pre_shift == nullptr) && a.pre_act >= 0 && a.pre_act <= 2, "dgq_quant_act: bad prologue");
    DGQ_CHECK_ARG(a.pre_act != 2 || (a.kh == 1 && a.kw == 1 && !a.pre_scale), "dgq_quant_act: GEGLU prologue is for Linear inputs");
    DGQ_CHECK_ARG((a.ln_gamma == nullptr) == (a.ln_beta == nullptr), "dgq_quant_act: LayerNorm prologue needs gamma and beta");
    DGQ_CHECK_ARG(!a.ln_gamma || (a.kh == 1 && a.kw == 1 && !a.pre_scale && a.pre_act == 0 && a.C % 4 == 0 && a.C <= DGQ_LN_MAX_C && a.ln_eps > 0.0f),
                  "dgq_quant_act: LayerNorm prologue is for Linear inputs (1x1, C %% 4 == 0, C <= 2048, no other prologue)");
    const int K = a.C * a.kh * a.kw;
    if (!a.ksrc) {
        DGQ_CHECK_ARG(a.C % 4 == 0, "dgq_quant_act: natural K order needs C %% 4 == 0 (C=%d)", a.C);
        DGQ_CHECK_ARG(a.Kp >= K, "dgq_quant_act: natural K order needs Kp >= K");
    }
    const int Ho = (a.H + 2 * a.pad - a.kh) / a.stride + 1, Wo = (a.W + 2 * a.pad - a.kw) / a.stride + 1;
    DGQ_CHECK_ARG(Ho > 0 && Wo > 0, "dgq_quant_act: empty output");
    p.x = a.x; p.B = a.B; p.H = a.H; p.W = a.W; p.C = a.C; p.kh = a.kh; p.kw = a.kw; p.stride = a.stride; p.pad = a.pad;
    p.Ho = Ho; p.Wo = Wo; p.ksrc = a.ksrc; p.koff = a.ksrc ? a.koff : nullptr; p.klds = a.ksrc ? a.klds : nullptr;
    p.kdst = a.ksrc ? a.kdst : nullptr;
    p.kpat = a.kpat;
    p.Kp = a.Kp; p.K = K; p.delta = a.delta; p.zp = a.zp; p.L = a.per_m ? a.L : 1;
    p.qmax = (float)((1 << a.bits) - 1);
    p.offset = (float)(1 << (a.bits - 1));
    p.codes = a.codes; p.rowsum = a.rowsum; p.M = a.B * Ho * Wo;
    p.kp_per_split = (((a.Kp + a.ksplits - 1) / a.ksplits) + 255) / 256 * 256;
    p.pre_scale = a.pre_scale; p.pre_shift = a.pre_shift; p.pre_act = a.pre_act;
    p.ln_gamma = a.ln_gamma; p.ln_beta = a.ln_beta; p.ln_eps = a.ln_eps;
    p.ldc = a.pre_act == 2 ? 2 * a.C : a.C;
    p.ups = a.ups ? 1 : 0;
    DGQ_CHECK_ARG(!p.ups || (a.H % 2 == 0 && a.W % 2 == 0 && a.kh * a.kw > 1), "dgq_quant_act: ups needs even H, W and a convolution");
    return DGQ_OK;
}

extern "C" int dgq_quant_act_batch(int n, const dgq_quant_act_args_t* args, void* stream) {
    DGQ_CHECK_ARG(args && n >= 1 && n <= DGQ_QA_BATCH, "dgq_quant_act_batch: n=%d (1..%d)", n, DGQ_QA_BATCH);
    QuantActBatch bt;
    int variant = -1;
    for (int i = 0; i < n; ++i) {
        const int rc = fill_quant_act(args[i], bt.p[i]);
        if (rc != DGQ_OK) return rc;
        const int v = quant_act_variant(bt.p[i], args[i].ksrc != nullptr);
        if (i == 0) variant = v;
        DGQ_CHECK_ARG(!bt.p[i].ups || v == 3 || v == 4 || v == 5, "dgq_quant_act_batch: the folded 2x upsample exists on the scatter and block-staged "
                      "conv paths (dgq_quant_act_variant 3 / 4 / 5), this problem takes variant %d", v);
        DGQ_CHECK_ARG(v == variant && args[i].x_dtype == args[0].x_dtype && (args[i].per_m != 0) == (args[0].per_m != 0) &&
                      bt.p[i].M == bt.p[0].M,
                      "dgq_quant_act_batch: problem %d differs from problem 0 in kernel variant / dtype / scale mode / row count", i);
        // the block-staged conv path takes its tile shape, grid and LDS size from problem 0: one geometry per launch
        DGQ_CHECK_ARG(v != 5 || (args[i].B == args[0].B && args[i].H == args[0].H && args[i].W == args[0].W && args[i].C == args[0].C &&
                                 args[i].kh == args[0].kh && args[i].kw == args[0].kw && args[i].stride == args[0].stride &&
                                 args[i].pad == args[0].pad && args[i].Kp == args[0].Kp),
                      "dgq_quant_act_batch: block-staged conv problems of one launch must share their geometry (problem %d)", i);
    }
    hipStream_t st = (hipStream_t)stream;
    const bool per_m = args[0].per_m != 0;
    switch (args[0].x_dtype) {
        case DGQ_F32: launch_quant_act<float>(bt, n, variant, per_m, st); break;
        case DGQ_F16: launch_quant_act<__half>(bt, n, variant, per_m, st); break;
        case DGQ_BF16: launch_quant_act<__hip_bfloat16>(bt, n, variant, per_m, st); break;
        default: dgq_set_error("dgq_quant_act: unknown dtype %d", args[0].x_dtype); return DGQ_EINVAL;
    }
    return dgq_launch_status("dgq_quant_act");
}

// which problems may share a launch: the kernel variant the library would pick for this problem (0 staged, 1 gather, 2 natural,
// 3 / 4 LDS scatter, 5 block-staged conv)
extern "C" int dgq_quant_act_variant(const dgq_quant_act_args_t* a) {
    QuantActParams p;
    if (!a || fill_quant_act(*a, p) != DGQ_OK) return -1;
    return quant_act_variant(p, a->ksrc != nullptr);
}

extern "C" int dgq_quant_act(const void* x, int x_dtype, int B, int H, int W, int C,
                             int kh, int kw, int stride, int pad,
                             const int32_t* ksrc, const int32_t* koff, const int32_t* klds, int Kp,
                             int per_m, const float* delta, const float* zp, int L,
                             int bits, int8_t* codes, float* rowsum, int ksplits,
                             const float* pre_scale, const float* pre_shift, int pre_act,
                             const float* ln_gamma, const float* ln_beta, float ln_eps, void* stream) {
    dgq_quant_act_args_t a;
    a.x = x; a.x_dtype = x_dtype; a.B = B; a.H = H; a.W = W; a.C = C; a.kh = kh; a.kw = kw; a.stride = stride; a.pad = pad;
    a.ksrc = ksrc; a.koff = koff; a.klds = klds; a.kdst = nullptr; a.kpat = nullptr; a.Kp = Kp; a.per_m = per_m; a.delta = delta; a.zp = zp; a.L = L; a.bits = bits;
    a.codes = codes; a.rowsum = rowsum; a.ksplits = ksplits; a.pre_scale = pre_scale; a.pre_shift = pre_shift; a.pre_act = pre_act;
    a.ln_gamma = ln_gamma; a.ln_beta = ln_beta; a.ln_eps = ln_eps; a.ups = 0;
    return dgq_quant_act_batch(1, &a, stream);
}
