// Fused attention with quantised softmax — replaces, for one attention call of Attention.Attention_forward
// (diffusers_rewrite/sd.py:183-201):
//     scores = (q @ k^T) * scale ; p = softmax(scores) (fp32) ; p = aqtizer_w(p) [column 0 bypassed under
//     start_peak] ; o = p @ v
// without materialising the [B,H,T,S] probabilities (1 GB per SD self-attention at 64x64 in the reference).
//
// T2ILogQuantizer in real-time mode (quant_layer_text.py:96-105) needs δ = max over the WHOLE probability tensor
// before any element can be quantised, so the kernel pair is two-pass:
//   pass 1  attn_stats : per query row m = max_s score, l = Σ_s exp(score − m) (online), and the row's largest
//                        probability outside the bypassed column, exp(m' − m)/l; a grid-wide max of those is δ
//                        (atomicMax on the bit pattern: values are positive floats);
//   pass 2  attn_pv    : recomputes the scores tile by tile, forms p = exp(score − m)/l with the FINAL m, l (no online
//                        rescaling), quantises and accumulates o += p̂·v.
// log2 quantiser in the log domain:  −log2(p/δ) = (m − score)·log2(e) + log2(l) + log2(δ)  → code = clamp(rne(·), 0,
// 2^b−1), p̂ = ldexp(δ, −code) (exact power of two times δ, as the reference's 2**(−code)·δ).
//
// All arithmetic is fp32 like the reference: the matrix products run on the exact fp32 MFMA
// (V_MFMA_F32_32X32X2_F32 = a k-ordered fmaf chain, 64 FLOP/clk/SIMD).  Work decomposition (wave64):
//   block = 4 waves = 128 query rows of one (batch, head); each wave owns 32 rows; key/value tiles of 32 rows are
//   staged once per block in LDS.  The score tile is computed TRANSPOSED (S^T = K·Q^T: A = K tile, B = Q) so that a
//   lane's column is its query row: all softmax statistics are in-register (16 keys per lane; lanes l and l+32 hold
//   the two key halves of the same query), and the S^T accumulator is directly the B operand of O^T = V^T · P^T
//   (cdna guide §3 "accumulator tile as the next MFMA's operand"): register r holds key (r&3)+8(r>>2) in lanes 0-31
//   and that key + 4 in lanes 32-63 — exactly the k = 0/1 pair of the 32x32x2 instruction.
#include <cstdlib>
#include "dgq_common.h"

typedef float v16f __attribute__((ext_vector_type(16)));

#define ATT_QROWS 128      // query rows per block
#define ATT_KT 32          // keys per tile
#define LOG2E 1.4426950408889634f

struct AttnParams {
    const float* q;
    const float* k;
    const float* v;
    float* o;
    int B, H, T, S;
    float scale;
    int mode;              // 0: no quantiser, 1: log2 real-time δ, 2: log2 static δ, 3: uniform (δ, z = 0)
    int skip;              // leading key columns that bypass the quantiser (start-peak)
    float qmax;
    float* stats;          // [B*H][T][2] : m, l
    float* delta;          // device scalar
};

__device__ __forceinline__ int key_of(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

template <int D>
__device__ __forceinline__ void load_tile(float* dst, const float* src, int rows_valid, int row_stride, int tid) {
    // dst [ATT_KT][D+1] ; src row-major with `row_stride` floats between consecutive keys
    constexpr int LD = D + 1;
    for (int i = tid; i < ATT_KT * D; i += 256) {
        const int r = i / D, c = i - r * D;
        dst[r * LD + c] = (r < rows_valid) ? src[(int64_t)r * row_stride + c] : 0.0f;
    }
}

// S^T tile (32 keys x 32 queries per wave): acc[r] = score(key_of(r,h), query lane&31), unscaled
template <int D>
__device__ __forceinline__ v16f score_tile(const float* ks, const float (&qreg)[D / 2], int lane) {
    constexpr int LD = D + 1;
    v16f acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    const float* kp = ks + (lane & 31) * LD + (lane >> 5);
#pragma unroll
    for (int kk = 0; kk < D / 2; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(kp[2 * kk], qreg[kk], acc, 0, 0, 0);
    return acc;
}

template <int D>
__global__ __launch_bounds__(256) void attn_stats_kernel(AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* ks = lds;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, h32 = lane >> 5;
    const int bh = blockIdx.y, b = bh / p.H, hd = bh - b * p.H;
    const int HD = p.H * D;
    const int t = blockIdx.x * ATT_QROWS + wid * 32 + (lane & 31);
    const int tq = min(t, p.T - 1);
    float qreg[D / 2];
    {
        const float* qp = p.q + ((int64_t)(b * p.T + tq) * p.H + hd) * D + h32;
#pragma unroll
        for (int kk = 0; kk < D / 2; ++kk) qreg[kk] = qp[2 * kk];
    }
    float m = -INFINITY, l = 0.0f, m2 = -INFINITY;      // m2: max over keys >= skip
    for (int s0 = 0; s0 < p.S; s0 += ATT_KT) {
        __syncthreads();
        load_tile<D>(ks, p.k + ((int64_t)(b * p.S + s0) * p.H + hd) * D, min(ATT_KT, p.S - s0), HD, tid);
        __syncthreads();
        v16f acc = score_tile<D>(ks, qreg, lane);
        float tmax = -INFINITY, tmax2 = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int s = s0 + key_of(r, h32);
            const float sc = (s < p.S) ? acc[r] * p.scale : -INFINITY;
            acc[r] = sc;
            tmax = fmaxf(tmax, sc);
            if (s >= p.skip) tmax2 = fmaxf(tmax2, sc);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        tmax2 = fmaxf(tmax2, __shfl_xor(tmax2, 32, 64));
        const float mn = fmaxf(m, tmax);
        float part = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) part += expf(acc[r] - mn);
        part += __shfl_xor(part, 32, 64);
        l = l * expf(m - mn) + part;
        m = mn;
        m2 = fmaxf(m2, tmax2);
    }
    if (t < p.T && h32 == 0) {
        float* st = p.stats + ((int64_t)bh * p.T + t) * 2;
        st[0] = m;
        st[1] = l;
    }
    if (p.mode == 1) {
        float pm = (t < p.T) ? expf(m2 - m) / l : 0.0f;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) pm = fmaxf(pm, __shfl_xor(pm, o, 64));
        if (lane == 0) atomicMax(reinterpret_cast<int*>(p.delta), __float_as_int(pm));
    }
}

template <int D>
__global__ __launch_bounds__(256) void attn_pv_kernel(AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int LD = D + 1;
    constexpr int NDT = (D + 31) / 32;                 // 32-wide d tiles of O^T
    float* ks = lds;
    float* vs = lds + ATT_KT * LD;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, h32 = lane >> 5;
    const int bh = blockIdx.y, b = bh / p.H, hd = bh - b * p.H;
    const int HD = p.H * D;
    const int t = blockIdx.x * ATT_QROWS + wid * 32 + (lane & 31);
    const int tq = min(t, p.T - 1);
    float qreg[D / 2];
    {
        const float* qp = p.q + ((int64_t)(b * p.T + tq) * p.H + hd) * D + h32;
#pragma unroll
        for (int kk = 0; kk < D / 2; ++kk) qreg[kk] = qp[2 * kk];
    }
    const float m = p.stats[((int64_t)bh * p.T + tq) * 2], l = p.stats[((int64_t)bh * p.T + tq) * 2 + 1];
    const float delta = (p.mode != 0) ? p.delta[0] : 1.0f;
    const float c0 = log2f(l) + log2f(delta);           // −log2(p/δ) = (m − score)·log2e + c0
    const float inv_l = 1.0f / l;
    v16f oacc[NDT];
#pragma unroll
    for (int j = 0; j < NDT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[j][r] = 0.0f;

    for (int s0 = 0; s0 < p.S; s0 += ATT_KT) {
        __syncthreads();
        const int valid = min(ATT_KT, p.S - s0);
        load_tile<D>(ks, p.k + ((int64_t)(b * p.S + s0) * p.H + hd) * D, valid, HD, tid);
        load_tile<D>(vs, p.v + ((int64_t)(b * p.S + s0) * p.H + hd) * D, valid, HD, tid);
        __syncthreads();
        v16f acc = score_tile<D>(ks, qreg, lane);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int s = s0 + key_of(r, h32);
            const float sc = acc[r] * p.scale;
            float ph;
            if (s >= p.S) {
                ph = 0.0f;
            } else if (p.mode == 0 || s < p.skip) {
                ph = expf(sc - m) * inv_l;               // unquantised probability (bypassed column / FP attention)
            } else if (p.mode == 3) {
                const float pr = expf(sc - m) * inv_l;   // uniform, always_zero: δ·clamp(rne(p/δ), 0, 2^b−1)
                ph = delta * fminf(fmaxf(rintf(__fdiv_rn(pr, delta)), 0.0f), p.qmax);
            } else {
                float code = rintf((m - sc) * LOG2E + c0);
                code = fminf(fmaxf(code, 0.0f), p.qmax);
                ph = ldexpf(delta, -(int)code);
            }
            acc[r] = ph;
        }
        // O^T[d, t] += Σ_s V^T[d, s] · P^T[s, t]: register r of acc is the B operand of k-step r
#pragma unroll
        for (int j = 0; j < NDT; ++j) {
            const int d = j * 32 + (lane & 31);
            const float* vp = vs + (4 * h32) * LD + min(d, D - 1);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int srow = (r & 3) + 8 * (r >> 2);
                const float a = (d < D) ? vp[srow * LD] : 0.0f;
                oacc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, acc[r], oacc[j], 0, 0, 0);
            }
        }
    }
    // O^T tile j: lane holds query t = lane&31 (column), rows d = j*32 + (r&3) + 8(r>>2) + 4*h32
    if (t < p.T) {
        float* op = p.o + ((int64_t)(b * p.T + t) * p.H + hd) * D;
#pragma unroll
        for (int j = 0; j < NDT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int d = j * 32 + key_of(r, h32);
                if (d < D) op[d] = oacc[j][r];
            }
    }
}

template <int D>
static int launch_attn(const AttnParams& p, hipStream_t st) {
    dim3 grid((p.T + ATT_QROWS - 1) / ATT_QROWS, p.B * p.H), block(256);
    const int lds1 = ATT_KT * (D + 1) * sizeof(float), lds2 = 2 * lds1;
    hipLaunchKernelGGL((attn_stats_kernel<D>), grid, block, lds1, st, p);
    hipLaunchKernelGGL((attn_pv_kernel<D>), grid, block, lds2, st, p);
    return dgq_launch_status("dgq_attention_f32");
}

int dgq_attention_bf16x3(const void* q, const void* k, const void* v, void* o, int io_dtype, int B, int H, int T, int S, int D,
                         float scale, int mode, int skip, float qmax, float* stats_ws, float* delta_ws, void* planes,
                         float* qfq, float* o_part, const dgq_attn_fq_t* fq, hipStream_t st);
size_t dgq_attention_bf16x3_bytes(int B, int H, int S, int D);
size_t dgq_attention_qi8_bytes(int B, int H, int T, int D);

// workspace layout: [0,256) δ scalar | stats B·H·T·10 floats (2 merged + 2 x 4 per key half of a split launch; 256-byte aligned) | tile images of the bf16 split planes of
// K and V | fake-quantised copy of q (used when aqtizer_q is fused) | the second key half's part of o (split launches) | the first half's (16-bit tensors)
static size_t attn_stats_off() { return 8704; }   // (δ slots at 0, the single-launch form's 1024 exchange granules at 512: DELTA_AREA_BYTES of attn_bf16x3_dev.h)
static size_t attn_planes_off(int B, int H, int T) { return 8704 + ((((size_t)B * H * T * 10 + 4) * sizeof(float) + 255) / 256) * 256; }   // (+ 4: the partial area starts 16-byte aligned)

// query scratch: an fp32 (fake-quantised) copy of q, or the int8 codes + per-query table of the QI8 path
static size_t attn_q_scratch(int B, int H, int T, int D) {
    const size_t f32 = (size_t)B * T * H * D * sizeof(float), i8 = dgq_attention_qi8_bytes(B, H, T, D);
    return ((f32 > i8 ? f32 : i8) + 255) / 256 * 256;
}

extern "C" size_t dgq_attention_workspace_bytes(int B, int H, int T, int S, int D) {
    return attn_planes_off(B, H, T) + dgq_attention_bf16x3_bytes(B, H, S, D) + attn_q_scratch(B, H, T, D) +
           2 * (((size_t)B * T * H * D * sizeof(float) + 255) / 256 * 256);       // (16-bit tensors: both halves' parts are fp32 scratch)
}

extern "C" int dgq_attention_fuses_fakequant(int D, int mode) {
    static const bool force_fp32 = getenv("DGQ_ATTN_FP32") != nullptr;
    return (mode >= 1 && mode <= 3 && !force_fp32 && dgq_attention_bf16x3_bytes(1, 1, 32, D) > 0) ? 1 : 0;
}

static int attention_impl(const void* q_, const void* k_, const void* v_, void* o_, int dtype, int B, int H, int T, int S,
                          int D, float scale, int mode, int skip, const float* delta_in, int bits,
                          const dgq_attn_fq_t* fq, void* workspace, size_t workspace_bytes, void* stream) {
    DGQ_CHECK_ARG(dtype == DGQ_F32 || dtype == DGQ_F16 || dtype == DGQ_BF16, "dgq_attention: unknown dtype %d", dtype);
    if (dtype != DGQ_F32 && !dgq_attention_fuses_fakequant(D, mode)) {
        dgq_set_error("dgq_attention: fp16 / bf16 tensors are served by the quantised modes only (mode %d, head_dim %d)", mode, D);
        return DGQ_EUNSUPPORTED;
    }
    const float* q = reinterpret_cast<const float*>(q_);
    const float* k = reinterpret_cast<const float*>(k_);
    const float* v = reinterpret_cast<const float*>(v_);
    float* o = reinterpret_cast<float*>(o_);
    DGQ_CHECK_ARG(q && k && v && o && workspace, "dgq_attention_f32: null pointer");
    DGQ_CHECK_ARG((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, "dgq_attention_f32: workspace must be 256-byte aligned");
    DGQ_CHECK_ARG(workspace_bytes >= dgq_attention_workspace_bytes(B, H, T, S, D), "dgq_attention_f32: workspace too small");
    float* delta_ws = reinterpret_cast<float*>(workspace);
    float* stats_ws = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + attn_stats_off());
    void* planes = reinterpret_cast<char*>(workspace) + attn_planes_off(B, H, T);
    DGQ_CHECK_ARG(B > 0 && H > 0 && T > 0 && S > 0, "dgq_attention_f32: bad shape");
    DGQ_CHECK_ARG(mode >= 0 && mode <= 3 && skip >= 0 && skip < S && bits >= 2 && bits <= 8, "dgq_attention_f32: bad mode");
    DGQ_CHECK_ARG(mode < 2 || delta_in, "dgq_attention_f32: static modes need delta");
    hipStream_t st = (hipStream_t)stream;
    AttnParams p;
    p.q = q; p.k = k; p.v = v; p.o = o; p.B = B; p.H = H; p.T = T; p.S = S; p.scale = scale; p.mode = mode; p.skip = skip;
    p.qmax = (float)((1 << bits) - 1); p.stats = stats_ws; p.delta = delta_ws;
    // quantised modes: bf16x3 MFMA path (fp32-equivalent accuracy, 16x the matrix rate); DGQ_ATTN_FP32=1 forces the
    // exact-fp32 MFMA kernels below, which also serve mode 0 and head dims the bf16x3 file does not instantiate
    const bool use3 = dgq_attention_fuses_fakequant(D, mode) != 0;
    bool any_fq = false;
    if (fq) for (int i = 0; i < 3; ++i) {
        if (fq[i].mode < 0) continue;
        any_fq = true;
        DGQ_CHECK_ARG(fq[i].mode <= 2 && fq[i].delta && fq[i].zero_point && fq[i].skip >= 0 && fq[i].bits >= 2 && fq[i].bits <= 8,
                      "dgq_attention_f32: bad q/k/v quantizer descriptor");
    }
    if (any_fq && !use3) {
        dgq_set_error("dgq_attention_f32: q/k/v quantizers are fused only where dgq_attention_fuses_fakequant(D, mode) "
                      "is 1; run dgq_fakequant_rows first");
        return DGQ_EUNSUPPORTED;
    }
    if (mode == 1 && !use3) {                                  // (the bf16x3 pre-pass resets δ itself)
        if (hipMemsetAsync(delta_ws, 0, sizeof(float), st) != hipSuccess) { dgq_set_error("dgq_attention_f32: memset"); return DGQ_ELAUNCH; }
    } else if (mode >= 2 && !use3) {                           // (the bf16x3 kernels read the caller's δ in place)
        if (hipMemcpyAsync(delta_ws, delta_in, sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) {
            dgq_set_error("dgq_attention_f32: memcpy"); return DGQ_ELAUNCH;
        }
    }
    float* qfq = reinterpret_cast<float*>(reinterpret_cast<char*>(planes) + dgq_attention_bf16x3_bytes(B, H, S, D));
    if (use3) return dgq_attention_bf16x3(q_, k_, v_, o_, dtype, B, H, T, S, D, scale, mode, skip, p.qmax, stats_ws,
                                          mode >= 2 ? const_cast<float*>(delta_in) : delta_ws, planes, qfq,
                                          reinterpret_cast<float*>(reinterpret_cast<char*>(qfq) + attn_q_scratch(B, H, T, D)), fq, st);
    switch (D) {
        case 8: return launch_attn<8>(p, st);
        case 16: return launch_attn<16>(p, st);
        case 40: return launch_attn<40>(p, st);
        case 64: return launch_attn<64>(p, st);
        case 80: return launch_attn<80>(p, st);
        case 160: return launch_attn<160>(p, st);
        default: dgq_set_error("dgq_attention_f32: head_dim %d not instantiated (8,16,40,64,80,160)", D); return DGQ_EUNSUPPORTED;
    }
}

extern "C" int dgq_attention(const void* q, const void* k, const void* v, void* o, int dtype, int B, int H, int T, int S,
                             int D, float scale, int mode, int skip, const float* delta_in, int bits,
                             const dgq_attn_fq_t* fq, void* workspace, size_t workspace_bytes, void* stream) {
    return attention_impl(q, k, v, o, dtype, B, H, T, S, D, scale, mode, skip, delta_in, bits, fq, workspace, workspace_bytes, stream);
}


extern "C" int dgq_attention_f32(const float* q, const float* k, const float* v, float* o, int B, int H, int T, int S,
                                 int D, float scale, int mode, int skip, const float* delta_in, int bits,
                                 const dgq_attn_fq_t* fq, void* workspace, size_t workspace_bytes, void* stream) {
    return dgq_attention(q, k, v, o, DGQ_F32, B, H, T, S, D, scale, mode, skip, delta_in, bits, fq, workspace, workspace_bytes, stream);
}
