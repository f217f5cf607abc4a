// GroupNorm statistics for channels-last tensors, emitted as per-(batch, channel) scale/shift so that the
// normalisation (and the SiLU after it) is applied inside dgq_quant_act's load instead of materialising
// GN(x) and SiLU(GN(x)) (QuantResnetBlock2D.forward: norm1 -> SiLU -> conv1, norm2 -> SiLU -> conv2,
// quant/quant_block.py:98-119):   GN(x)[b,c] = x·scale[b,c] + shift[b,c],
//   scale = rstd[b,g]·γ[c],  shift = β[c] − mean[b,g]·rstd[b,g]·γ[c].
// One launch when a (batch, group) fits one block (slices == 1), else partial moments per (b, group, spatial slice)
// followed by a merge (Chan) + scale/shift kernel.
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

// grid (B*G, S): block handles spatial rows [s*rows_per, ...) of group g of image b.  Per thread: shifted sums
// Σ(x − K), Σ(x − K)² with K = the first element of the block's range (3 VALU per element, no cancellation worth
// mentioning because K sits inside the data), converted to (n, mean, M2) and merged pairwise (Chan) in a fixed order.
// S == 1 (FINAL): the block also writes the group's scale/shift, so the whole GroupNorm statistic is one launch.
template <typename T, bool FINAL>
__global__ __launch_bounds__(256) void gn_partial_kernel(const T* __restrict__ x, int HW, int C, int G, int rows_per,
                                                         float* __restrict__ part, float eps,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         float* __restrict__ scale, float* __restrict__ shift) {
    const int bg = blockIdx.x, b = bg / G, g = bg - b * G;
    const int Cg = C / G;
    const int r0 = blockIdx.y * rows_per, r1 = min(HW, r0 + rows_per);
    const T* base = x + ((int64_t)b * HW) * C + g * Cg;
    const float K = dgq_to_float(base[(int64_t)r0 * C]);
    const int total = (r1 - r0) * Cg;
    float s1 = 0.0f, s2 = 0.0f, cnt = 0.0f;
    int r = threadIdx.x / Cg, c = threadIdx.x - r * Cg;     // element i = r*Cg + c, advanced by 256 without divisions
    const int dr = 256 / Cg, dc = 256 - dr * Cg;
    for (int i = threadIdx.x; i < total; i += 256) {
        const float d = dgq_to_float(base[(int64_t)(r0 + r) * C + c]) - K;
        s1 += d;
        s2 += d * d;
        cnt += 1.0f;
        r += dr; c += dc;
        if (c >= Cg) { c -= Cg; ++r; }
    }
    Moments m = {cnt, 0.0f, 0.0f};
    if (cnt > 0.0f) {
        const float md = s1 / cnt;
        m.mean = K + md;
        m.m2 = fmaxf(s2 - s1 * md, 0.0f);
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
    __shared__ float fin[2];
    __shared__ int do_final;
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = merge(merge(sm[0], sm[1]), merge(sm[2], sm[3]));
        int fin_here = FINAL ? 1 : 0;
        if (!FINAL) {
            float* p = part + ((int64_t)bg * gridDim.y + blockIdx.y) * 3;
            p[0] = m.n; p[1] = m.mean; p[2] = m.m2;
        }
        if (fin_here) {
            fin[0] = m.mean;
            fin[1] = rsqrtf(m.m2 / m.n + eps);            // biased variance, as F.group_norm
        }
        do_final = fin_here;
    }
    __syncthreads();
    if (do_final) {
        const float mean = fin[0], rstd = fin[1];
        for (int cc = threadIdx.x; cc < Cg; cc += 256) {
            const int ch = g * Cg + cc;
            const float sc = rstd * gamma[ch];
            scale[(int64_t)b * C + ch] = sc;
            shift[(int64_t)b * C + ch] = beta[ch] - mean * sc;
        }
    }
}

// grid (B*G): merges S partials, writes scale/shift for the group's channels.  (A row-slice form of the partial pass —
// one workgroup per slice of whole pixel rows, 16-byte coalesced reads, all channels — measured 0.7 % SLOWER on the SD step
// than the per-(group, slice) kernel above, whose 40-byte runs are served from L2: dropped.)
__global__ __launch_bounds__(64) void gn_finalize_kernel(const float* __restrict__ part, int S, int C, int G, float eps,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float* __restrict__ scale, float* __restrict__ shift) {
    const int bg = blockIdx.x, b = bg / G, g = bg - b * G;
    const int Cg = C / G;
    // lane t merges slices t, t + 64, ... in order, then a fixed xor-tree over the 64 lanes: deterministic for any S
    Moments m = {0.0f, 0.0f, 0.0f};
    for (int s = threadIdx.x; s < S; s += 64) {
        const float* p = part + ((int64_t)bg * S + s) * 3;
        Moments o = {p[0], p[1], p[2]};
        m = merge(m, o);
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        Moments other;
        other.n = __shfl_xor(m.n, o, 64);
        other.mean = __shfl_xor(m.mean, o, 64);
        other.m2 = __shfl_xor(m.m2, o, 64);
        // both partners must form the SAME value: merge in lane order (lower lane first)
        m = (threadIdx.x & o) ? merge(other, m) : merge(m, other);
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

// grid (B*G), 256 threads: the (batch, group)'s statistic from the per-(16-row block, channel) partials written by
// dgq_gemm_wxa8's epilogue.  Every partial stands for 16 values, so the merge is plain sums: with K the group's first partial mean,
// mean = K + Σ(m_i − K)/E and M2 = Σ M2_i + 16·(Σ(m_i − K)² − (Σ(m_i − K))²/E) — independent loads, no division in the loop
// (a lane-serial Chan merge of the same entries took ~10 us per statistic: 40 dependent iterations with two divisions each).
// Fixed summation order (thread-strided, then a fixed shuffle / LDS tree): deterministic.  Channels [0, C1) come from `part`,
// [C1, C1 + C2) from `part2` (a channel concat).
__global__ __launch_bounds__(256) void gn_from_partials_kernel(const float* __restrict__ part, int C1, const float* __restrict__ part2,
                                                              int C2, int HW, int G, float eps, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float* __restrict__ scale,
                                                              float* __restrict__ shift) {
    const int bg = blockIdx.x, b = bg / G, g = bg - b * G;
    const int C = C1 + C2, Cg = C / G, RB = HW / 16;
    const int total = RB * Cg;
    auto entry = [&](int e) {
        const int rb = e / Cg, c = g * Cg + (e - rb * Cg);
        return (c < C1) ? part + ((int64_t)(b * RB + rb) * C1 + c) * 2 : part2 + ((int64_t)(b * RB + rb) * C2 + (c - C1)) * 2;
    };
    // every load of the launch goes out before the first wait: the shift K, this thread's first four partials and its channel's
    // (γ, β) — as written before (K, then the loop, then γ / β behind the reduction) the 5 us of this launch were three dependent
    // round trips to L2 in series
    const float* e0 = entry(0);
    float2 pre[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) pre[j] = *reinterpret_cast<const float2*>(entry(min((int)threadIdx.x + 256 * j, total - 1)));
    const int chq = g * Cg + min((int)threadIdx.x, Cg - 1);
    const float gam = gamma[chq], bet = beta[chq];
    const float K = e0[0];
    float s1 = 0.0f, s2 = 0.0f, sm = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if ((int)threadIdx.x + 256 * j < total) {
            const float d = pre[j].x - K;
            s1 += d; s2 += d * d; sm += pre[j].y;
        }
#pragma unroll 4
    for (int e = threadIdx.x + 1024; e < total; e += 256) {
        const float2 v = *reinterpret_cast<const float2*>(entry(e));
        const float d = v.x - K;
        s1 += d; s2 += d * d; sm += v.y;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s1 += __shfl_down(s1, o, 64); s2 += __shfl_down(s2, o, 64); sm += __shfl_down(sm, o, 64);
    }
    __shared__ float red[4][3];
    __shared__ float fin[2];
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = s1; red[threadIdx.x >> 6][1] = s2; red[threadIdx.x >> 6][2] = sm; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float t1 = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
        const float t2 = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
        const float tm = (red[0][2] + red[1][2]) + (red[2][2] + red[3][2]);
        const float E = (float)total;
        const float md = t1 / E;
        const float m2 = tm + 16.0f * fmaxf(t2 - t1 * md, 0.0f);
        fin[0] = K + md;
        fin[1] = rsqrtf(m2 / (16.0f * E) + eps);                      // biased variance, as F.group_norm
    }
    __syncthreads();
    const float mean = fin[0], rstd = fin[1];
    if ((int)threadIdx.x < Cg) {
        const float sc = rstd * gam;
        scale[(int64_t)b * C + chq] = sc;
        shift[(int64_t)b * C + chq] = bet - mean * sc;
    }
    for (int c = threadIdx.x + 256; c < Cg; c += 256) {      // (groups wider than 256 channels: not in these models)
        const int ch = g * Cg + c;
        const float sc = rstd * gamma[ch];
        scale[(int64_t)b * C + ch] = sc;
        shift[(int64_t)b * C + ch] = beta[ch] - mean * sc;
    }
}

extern "C" int dgq_groupnorm_from_partials(const float* partial, int C1, const float* partial2, int C2, int B, int HW, int G,
                                           float eps, const float* gamma, const float* beta, float* scale, float* shift,
                                           void* stream) {
    DGQ_CHECK_ARG(partial && gamma && beta && scale && shift && (C2 == 0 || partial2), "dgq_groupnorm_from_partials: null pointer");
    DGQ_CHECK_ARG(B > 0 && HW > 0 && HW % 16 == 0 && C1 > 0 && C2 >= 0 && G > 0 && (C1 + C2) % G == 0,
                  "dgq_groupnorm_from_partials: bad shape (HW %% 16 == 0, (C1 + C2) %% G == 0)");
    hipLaunchKernelGGL(gn_from_partials_kernel, dim3(B * G), dim3(256), 0, (hipStream_t)stream, partial, C1, partial2, C2, HW, G,
                       eps, gamma, beta, scale, shift);
    return dgq_launch_status("dgq_groupnorm_from_partials");
}

extern "C" int dgq_groupnorm_scale_shift(const void* x, int x_dtype, int B, int HW, int C, int G, float eps,
                                         const float* gamma, const float* beta, float* scale, float* shift,
                                         float* partial_ws, int slices, void* stream) {
    DGQ_CHECK_ARG(x && gamma && beta && scale && shift && partial_ws, "dgq_groupnorm_scale_shift: null pointer");
    DGQ_CHECK_ARG(B > 0 && HW > 0 && C > 0 && G > 0 && C % G == 0 && slices >= 1 && slices <= 256,
                  "dgq_groupnorm_scale_shift: bad shape");
    hipStream_t st = (hipStream_t)stream;
    const int rows_per = (HW + slices - 1) / slices;
    const int S = (HW + rows_per - 1) / rows_per;
    dim3 grid(B * G, S), block(256);
#define DGQ_GN_LAUNCH(TT, FIN) hipLaunchKernelGGL((gn_partial_kernel<TT, FIN>), grid, block, 0, st, (const TT*)x, HW, C, G, \
                                                  rows_per, partial_ws, eps, gamma, beta, scale, shift)
    switch (x_dtype) {
        case DGQ_F32: if (S == 1) DGQ_GN_LAUNCH(float, true); else DGQ_GN_LAUNCH(float, false); break;
        case DGQ_F16: if (S == 1) DGQ_GN_LAUNCH(__half, true); else DGQ_GN_LAUNCH(__half, false); break;
        case DGQ_BF16: if (S == 1) DGQ_GN_LAUNCH(__hip_bfloat16, true); else DGQ_GN_LAUNCH(__hip_bfloat16, false); break;
        default: dgq_set_error("dgq_groupnorm_scale_shift: unknown dtype %d", x_dtype); return DGQ_EINVAL;
    }
#undef DGQ_GN_LAUNCH
    if (S > 1)
        hipLaunchKernelGGL(gn_finalize_kernel, dim3(B * G), dim3(64), 0, st, partial_ws, S, C, G, eps, gamma, beta, scale, shift);
    return dgq_launch_status("dgq_groupnorm_scale_shift");
}
