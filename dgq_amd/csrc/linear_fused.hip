// Quantise-on-load INSIDE the GEMM for Linear / 1x1-conv layers: one launch per layer instead of dgq_quant_act +
// dgq_gemm_wxa8, and no int8 operand in HBM (reference operator: QuantLayer.forward,
// quant/quant_layer.py:626-661 with F.linear at :659).  OPT-IN (DGQ_FUSED_LINEAR): measured against the two launches it is
// faster only for per-M / scalar tables on narrow layers (8192 x 320 -> 320: 12.7 vs 20.2 us) and slower for DGQ's per-K
// tables (30.6 vs 26.1 us) and wide N — see DESIGN.md §4.
//
// A workgroup owns a PANEL of 16 activation rows and 320 output columns:
//   phase 1  the four waves read 4 rows each (16 lanes per row, 16-byte loads, the whole row in registers), apply the folded
//            prologue (GroupNorm scale/shift, LayerNorm with the row statistics of dgq_quant_act — same summation tree,
//            bit for bit — SiLU, GEGLU), quantise with clamp(rne(x/δ)+z) (exact-division semantics, dgq_common.h) and
//            write the centred int8 codes straight into the MFMA operand image in LDS:
//              per-K (DGQ groups): every SOURCE element is quantised once with the (δ, z) of its destination chunk and its
//                      byte scattered to its packed position kdst[c] (the group-sorted, 64-padded K order of the weight);
//              per-M / scalar: 4 codes per dword at the natural position.
//            The image is laid out as the GEMM's A stage ([K tile][row][128 B], 16-byte chunk c at c ^ ((row>>1)&7)).
//   phase 2  the waves split the COLUMNS — wave w owns 80 of the 320 columns (5 MFMA column tiles) for all 16 rows — and its
//            int4 weight fragments never touch LDS: each lane reads its 8 packed bytes per (column tile, K half) straight
//            from L2 into registers (ordinary loads the compiler counts itself, one K tile prefetched ahead).  After the single
//            barrier that publishes the panel the waves never synchronise again.  V_MFMA_I32_16X16X64_I8, per-group fp32
//            flush by summation by parts (per-K), the shared dequantising epilogue with the optional residual /
//            attention-side quantizer of dgq_gemm_wxa8, 16-byte stores through a per-wave LDS transpose.
// Layers wider than 320 columns use several workgroups per panel (each re-quantises it).
// (A first form — 32-row panels, 64-wide column tiles fed by a barrier-synchronised LDS-DMA weight ring — measured 15.8 us
// on the per-M case and 30 us per-K: profiles/r02_fused_linear_microbench.txt.)
#include <atomic>
#include <cstdlib>
#include "dgq_common.h"
#include "gemm_device.h"

#define FBK 128
#define DGQ_FUSED_BATCH 4

struct FusedParams {
    // activation side
    const void* x;
    int M, C, ldc, hw;                    // rows, channels, elements per row (C, or 2C for GEGLU), rows per image (pre_scale)
    const int32_t* kdst;                  // per-K: [C] packed position of channel c; NULL: natural order
    const float* qdelta;                  // per-K: [Kp/64]; per-M: [L]
    const float* qzp;
    int L;
    float qmax, offset;
    const float* pre_scale;               // optional [B][C]
    const float* pre_shift;
    int pre_act;
    const float* ln_gamma;
    const float* ln_beta;
    float ln_eps;
    // weight side / epilogue (as dgq_gemm_wxa8)
    int Kp, N;
    const uint8_t* wpacked;
    const uint8_t* cflush;
    const float* alpha;
    const float* zw;
    const float* gamma;
    const float* vn;
    void* y;
    int ldy;
    dgq_gemm_extra_t ex;
};

struct FusedBatch {
    FusedParams p[DGQ_FUSED_BATCH];
    int debug;                            // development (DGQ_FUSED_DEBUG, timing only): 1 = stop after phase 1, 2 = skip phase 1, 3 = skip the tile loop
};

template <typename TIn>
__device__ __forceinline__ void f_load4(const TIn* p, float (&v)[4]);
template <>
__device__ __forceinline__ void f_load4<float>(const float* p, float (&v)[4]) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
template <>
__device__ __forceinline__ void f_load4<__half>(const __half* p, float (&v)[4]) {
    const uint2 t = *reinterpret_cast<const uint2*>(p);
    const __half* h = reinterpret_cast<const __half*>(&t);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = __half2float(h[j]);
}
template <>
__device__ __forceinline__ void f_load4<__hip_bfloat16>(const __hip_bfloat16* p, float (&v)[4]) {
    const uint2 t = *reinterpret_cast<const uint2*>(p);
    const uint16_t* h = reinterpret_cast<const uint16_t*>(&t);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = __uint_as_float(((uint32_t)h[j]) << 16);
}

#define F2_BM 16
#define F2_NB 320
#define F2_TN 5
#define F2_ABYTES (F2_BM * FBK)

__device__ __forceinline__ int panel2_addr(int r, int kp) {
    return (kp >> 7) * F2_ABYTES + r * FBK + ((((kp & 127) >> 4) ^ ((r >> 1) & 7)) << 4) + (kp & 15);
}

// NV = float4 registers per lane for one row slice: 16 lanes x NV x 4 floats >= C
template <typename T, bool PER_M, int NV>
__global__ __launch_bounds__(256) void linear_fused2_kernel(FusedBatch bt) {
    const FusedParams& p = bt.p[blockIdx.z];
    const int m0 = blockIdx.x * F2_BM;
    const int nb0 = blockIdx.y * F2_NB;
    if (m0 >= p.M || nb0 >= p.N) return;
    const int nk = p.Kp / FBK, nch = p.Kp >> 6;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);

    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t* panel = smem;                                               // [nk][16][128]
    float* ep_all = reinterpret_cast<float*>(panel + (size_t)nk * F2_ABYTES);   // [4 waves][16][84]
    float* vtab = ep_all + 4 * 16 * 84;                                  // [3][16]
    float* vcol = vtab + 3 * F2_BM;                                      // [4][320]
    float* ctab = vcol + 4 * F2_NB;                                      // [nch]
    float* tdelta = ctab + nch;
    float* tinv = tdelta + nch;
    float* tzp = tinv + nch;
    int32_t* kd = reinterpret_cast<int32_t*>(tzp + nch);                 // [C]

    for (int i = tid; i < nk * (F2_ABYTES / 16); i += 256) reinterpret_cast<uint4*>(panel)[i] = make_uint4(0, 0, 0, 0);
    for (int c = tid; c < F2_NB; c += 256) {
        const int n = min(nb0 + c, p.N - 1);
        vcol[c] = p.alpha[n]; vcol[F2_NB + c] = p.zw[n]; vcol[2 * F2_NB + c] = p.gamma[n];
        vcol[3 * F2_NB + c] = PER_M ? p.vn[n] : 0.0f;
    }
    if (!PER_M) {
        for (int i = tid; i < nch; i += 256) {
            const float d = p.qdelta[i];
            tdelta[i] = d; tinv[i] = dgq_rcp(d); tzp[i] = p.qzp[i];
            ctab[i] = d - (i == nch - 1 ? 0.0f : p.qdelta[i + 1]);        // summation by parts, as dgq_gemm_wxa8
        }
        for (int i = tid; i < p.C; i += 256) kd[i] = p.kdst[i];
    }
    __syncthreads();

    // ---- phase 1: 4 rows per wave, 16 lanes per row ----------------------------------------------------------------------
    if (bt.debug != 2) {
        const int r = wid * 4 + (lane >> 4), sub = lane & 15;
        const int m = min(m0 + r, p.M - 1);
        const T* xr = reinterpret_cast<const T*>(p.x) + (int64_t)m * p.ldc;
        float v[NV][4];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = sub * 4 + 64 * i;
            if (c < p.C) f_load4<T>(xr + c, v[i]);
            else v[i][0] = v[i][1] = v[i][2] = v[i][3] = 0.0f;
        }
        if (p.pre_scale) {
            const int b = m / p.hw;
            const float* sc = p.pre_scale + (int64_t)b * p.C;
            const float* sh = p.pre_shift + (int64_t)b * p.C;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = sub * 4 + 64 * i;
                if (c < p.C) {
                    const float4 a = *reinterpret_cast<const float4*>(sc + c), s4 = *reinterpret_cast<const float4*>(sh + c);
                    v[i][0] = v[i][0] * a.x + s4.x; v[i][1] = v[i][1] * a.y + s4.y;
                    v[i][2] = v[i][2] * a.z + s4.z; v[i][3] = v[i][3] * a.w + s4.w;
                }
            }
        }
        if (p.ln_gamma) {
            // dgq_quant_act's row statistics, bit for bit: virtual lane L = (c/4) % 64 = sub + 16·(i % 4); the xor 32 / 16
            // steps of its butterfly pair registers of this lane, the xor 8 / 4 / 2 / 1 steps the 16 lanes of the row
            float t[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                t[q] = 0.0f;
#pragma unroll
                for (int u = q; u < NV; u += 4) t[q] += (v[u][0] + v[u][1]) + (v[u][2] + v[u][3]);
            }
            t[0] = t[0] + t[2]; t[1] = t[1] + t[3];
            float s = t[0] + t[1];
            s += __shfl_xor(s, 8, 64); s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 1, 64);
            const float mu = s / (float)p.C;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                t[q] = 0.0f;
#pragma unroll
                for (int u = q; u < NV; u += 4) {
                    if (sub * 4 + 64 * u < p.C) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) t[q] += (v[u][j] - mu) * (v[u][j] - mu);
                    }
                }
            }
            t[0] = t[0] + t[2]; t[1] = t[1] + t[3];
            float qv = t[0] + t[1];
            qv += __shfl_xor(qv, 8, 64); qv += __shfl_xor(qv, 4, 64); qv += __shfl_xor(qv, 2, 64); qv += __shfl_xor(qv, 1, 64);
            const float rstd = 1.0f / sqrtf(qv / (float)p.C + p.ln_eps);
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = sub * 4 + 64 * i;
                if (c < p.C) {
                    const float4 ga = *reinterpret_cast<const float4*>(p.ln_gamma + c);
                    const float4 be = *reinterpret_cast<const float4*>(p.ln_beta + c);
                    v[i][0] = (v[i][0] - mu) * rstd * ga.x + be.x; v[i][1] = (v[i][1] - mu) * rstd * ga.y + be.y;
                    v[i][2] = (v[i][2] - mu) * rstd * ga.z + be.z; v[i][3] = (v[i][3] - mu) * rstd * ga.w + be.w;
                }
            }
        }
        if (p.pre_act == 1) {
#pragma unroll
            for (int i = 0; i < NV; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) v[i][j] = dgq_silu(v[i][j]);
        } else if (p.pre_act == 2) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = sub * 4 + 64 * i;
                if (c < p.C) {
                    float g[4];
                    f_load4<T>(xr + p.C + c, g);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[i][j] = v[i][j] * (0.5f * g[j] * (1.0f + erff(g[j] * 0.70710678118654752f)));
                }
            }
        }
        float partial = 0.0f;
        if (PER_M) {
            const int li = m % p.L;
            const float md = p.qdelta[li], mz = p.qzp[li], minv = dgq_rcp(md);
            const float bias = 128.0f - p.offset;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = sub * 4 + 64 * i;
                if (c < p.C) {
                    uint32_t w = 0;
                    float fsum = 0.0f;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float b8 = dgq_affine_code_fast(v[i][j], md, minv, mz, p.qmax) + bias;
                        w = __builtin_amdgcn_cvt_pk_u8_f32(b8, j, w);
                        fsum += b8;
                    }
                    *reinterpret_cast<uint32_t*>(panel + panel2_addr(r, c)) = w ^ 0x80808080u;
                    partial += fsum - 512.0f;
                }
            }
            partial += __shfl_xor(partial, 8, 64); partial += __shfl_xor(partial, 4, 64);
            partial += __shfl_xor(partial, 2, 64); partial += __shfl_xor(partial, 1, 64);
            if (sub == 0) { vtab[r] = md; vtab[F2_BM + r] = md * partial; vtab[2 * F2_BM + r] = md * (p.offset - mz); }
        } else {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = sub * 4 + 64 * i;
                if (c < p.C) {
                    const int4 d4 = *reinterpret_cast<const int4*>(kd + c);
                    const int dst[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int ch = dst[j] >> 6;
                        const float d = tdelta[ch];
                        const float sc = dgq_affine_code_fast(v[i][j], d, tinv[ch], tzp[ch], p.qmax) - p.offset;
                        panel[panel2_addr(r, dst[j])] = (uint8_t)(int)sc;
                        partial += d * sc;
                    }
                }
            }
            partial += __shfl_xor(partial, 8, 64); partial += __shfl_xor(partial, 4, 64);
            partial += __shfl_xor(partial, 2, 64); partial += __shfl_xor(partial, 1, 64);
            if (sub == 0) { vtab[r] = 1.0f; vtab[F2_BM + r] = partial; vtab[2 * F2_BM + r] = 0.0f; }
        }
    }
    __syncthreads();                                                     // the panel is complete; no barrier after this one
    if (bt.debug == 1) return;

    // ---- phase 2: this wave's 80 columns x all K tiles, weights from L2 to registers ------------------------------------------
    const int fr = lane & 15, fq = lane >> 4;
    const int a_off0 = fr * FBK + (((0 + fq) ^ ((fr >> 1) & 7)) << 4);
    const int a_off1 = fr * FBK + (((4 + fq) ^ ((fr >> 1) & 7)) << 4);
    const int colw = wid * (F2_NB / 4);                                  // first column of this wave inside the workgroup's 320
    const uint8_t* wrow[F2_TN];
#pragma unroll
    for (int j = 0; j < F2_TN; ++j) {
        const int n = min(nb0 + colw + 16 * j + fr, p.N - 1);
        wrow[j] = p.wpacked + (int64_t)n * (p.Kp / 2) + fq * 8;
    }
    v4i acc[F2_TN];
    v4f accf[F2_TN];
#pragma unroll
    for (int j = 0; j < F2_TN; ++j) { acc[j] = (v4i){0, 0, 0, 0}; accf[j] = (v4f){0.f, 0.f, 0.f, 0.f}; }
    uint2 wc[F2_TN][2], wn[F2_TN][2];
#pragma unroll
    for (int j = 0; j < F2_TN; ++j) {
        wc[j][0] = *reinterpret_cast<const uint2*>(wrow[j]);
        wc[j][1] = *reinterpret_cast<const uint2*>(wrow[j] + 32);
    }
    for (int kt = 0; kt < (bt.debug == 3 ? 0 : nk); ++kt) {
        const int ktn = min(kt + 1, nk - 1);
#pragma unroll
        for (int j = 0; j < F2_TN; ++j) {                                // next K tile's fragments fly during this tile's MFMAs
            wn[j][0] = *reinterpret_cast<const uint2*>(wrow[j] + ktn * (FBK / 2));
            wn[j][1] = *reinterpret_cast<const uint2*>(wrow[j] + ktn * (FBK / 2) + 32);
        }
        const uint8_t* sa = panel + kt * F2_ABYTES;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const v4i af = *reinterpret_cast<const v4i*>(sa + (h ? a_off1 : a_off0));
#pragma unroll
            for (int j = 0; j < F2_TN; ++j) {
                const uint2 wv = wc[j][h];
                const v4i bf = (v4i){(int)(wv.x & 0x0F0F0F0Fu), (int)((wv.x >> 4) & 0x0F0F0F0Fu),
                                     (int)(wv.y & 0x0F0F0F0Fu), (int)((wv.y >> 4) & 0x0F0F0F0Fu)};
                acc[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af, bf, acc[j], 0, 0, 0);
            }
            if (!PER_M) {
                const float sc = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(
                    __builtin_bit_cast(int, ctab[2 * kt + h])));
                if (sc != 0.0f) {
#pragma unroll
                    for (int j = 0; j < F2_TN; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) accf[j][r] = __builtin_fmaf(sc, (float)acc[j][r], accf[j][r]);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < F2_TN; ++j) { wc[j][0] = wn[j][0]; wc[j][1] = wn[j][1]; }
    }

    // ---- epilogue: 16 x 80 outputs of this wave through its own LDS slab, 16-byte stores -----------------------------------------
    float* ep = ep_all + wid * 16 * 84;
#pragma unroll
    for (int j = 0; j < F2_TN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) ep[(fq * 4 + r) * 84 + j * 16 + fr] = PER_M ? (float)acc[j][r] : accf[j][r];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                  // same-wave LDS round trip
    T* y = reinterpret_cast<T*>(p.y);
    const bool has_extra = p.ex.fq_mode != 0 || p.ex.residual != nullptr;
    const bool st_al = ((p.ldy * (int)sizeof(T)) % 16 == 0) && ((reinterpret_cast<uintptr_t>(p.y) & 15) == 0) &&
                       (sizeof(T) == 4 || (p.ldy & 3) == 0);
#pragma unroll
    for (int pass = 0; pass < 5; ++pass) {
        const int idx = pass * 64 + lane;                                // 16 rows x 20 float4
        const int row = idx / 20, c4 = (idx - row * 20) * 4;
        const int m = m0 + row;
        const int nb = nb0 + colw + c4;
        if (m >= p.M || nb >= p.N) continue;
        const float4 v = *reinterpret_cast<const float4*>(ep + row * 84 + c4);
        const float r0 = vtab[row], r1 = vtab[F2_BM + row], r2 = vtab[2 * F2_BM + row];
        const float* vc = vcol + colw + c4;
        const float4 al = *reinterpret_cast<const float4*>(vc), zw = *reinterpret_cast<const float4*>(vc + F2_NB);
        const float4 ga = *reinterpret_cast<const float4*>(vc + 2 * F2_NB), vn = *reinterpret_cast<const float4*>(vc + 3 * F2_NB);
        float o[4];
        o[0] = dgq_dequant<PER_M>(v.x, r0, r1, r2, al.x, zw.x, ga.x, vn.x);
        o[1] = dgq_dequant<PER_M>(v.y, r0, r1, r2, al.y, zw.y, ga.y, vn.y);
        o[2] = dgq_dequant<PER_M>(v.z, r0, r1, r2, al.z, zw.z, ga.z, vn.z);
        o[3] = dgq_dequant<PER_M>(v.w, r0, r1, r2, al.w, zw.w, ga.w, vn.w);
        if (has_extra) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (nb + k < p.N) o[k] = dgq_extra(p.ex, o[k], m, nb + k);
        }
        T* dst = y + (int64_t)m * p.ldy + nb;
        if (st_al && nb + 3 < p.N) {
            if (sizeof(T) == 4) {
                *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
            } else {
                T t4[4] = {dgq_from_float<T>(o[0]), dgq_from_float<T>(o[1]), dgq_from_float<T>(o[2]), dgq_from_float<T>(o[3])};
                *reinterpret_cast<uint2*>(dst) = *reinterpret_cast<const uint2*>(t4);
            }
        } else {
            for (int k = 0; k < 4 && nb + k < p.N; ++k) dst[k] = dgq_from_float<T>(o[k]);
        }
    }
}

static size_t fused2_lds_bytes(int Kp, int C, bool per_m) {
    size_t b = (size_t)(Kp / FBK) * F2_ABYTES + 4 * 16 * 84 * 4 + 3 * F2_BM * 4 + (size_t)4 * F2_NB * 4;
    if (!per_m) b += (size_t)4 * (Kp >> 6) * 4 + (size_t)C * 4;
    return (b + 15) & ~(size_t)15;
}
static int fused2_nv(int C) { return C <= 320 ? 5 : (C <= 768 ? 12 : (C <= 1280 ? 20 : 0)); }

template <typename T, bool PER_M, int NV>
static void launch_fused2_nv(const FusedBatch& bt, dim3 grid, size_t lds, hipStream_t st) {
    static std::atomic<bool> attr_set[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !attr_set[dev].load(std::memory_order_acquire)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_fused2_kernel<T, PER_M, NV>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
        if (dev >= 0 && dev < 64) attr_set[dev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL((linear_fused2_kernel<T, PER_M, NV>), grid, dim3(256), lds, st, bt);
}
template <typename T, bool PER_M>
static void launch_fused2(const FusedBatch& bt, int nv, dim3 grid, size_t lds, hipStream_t st) {
    if (nv == 5) launch_fused2_nv<T, PER_M, 5>(bt, grid, lds, st);
    else if (nv == 12) launch_fused2_nv<T, PER_M, 12>(bt, grid, lds, st);
    else launch_fused2_nv<T, PER_M, 20>(bt, grid, lds, st);
}

// ----------------------------------------------------------------------------------------------------------------- host side
extern "C" int dgq_linear_fused_supported(int M, int C, int Kp, int N, int per_m, int w_bits) {
    if (w_bits != 4 || M < 1 || N < 1 || C % 4 != 0 || fused2_nv(C) == 0 || Kp % FBK != 0 || Kp > 2048 || Kp < C) return 0;
    return fused2_lds_bytes(Kp, C, per_m != 0) <= 150 * 1024 ? 1 : 0;
}

extern "C" int dgq_linear_fused_batch(int n, const dgq_fused_linear_args_t* args, void* stream) {
    DGQ_CHECK_ARG(args && n >= 1 && n <= DGQ_FUSED_BATCH, "dgq_linear_fused_batch: n=%d (1..%d)", n, DGQ_FUSED_BATCH);
    FusedBatch bt;
    const dgq_fused_linear_args_t& a0 = args[0];
    int maxN = 0;
    for (int i = 0; i < n; ++i) {
        const dgq_fused_linear_args_t& a = args[i];
        DGQ_CHECK_ARG(a.x && a.delta && a.zp && a.wpacked && a.alpha && a.zw && a.gamma && a.y, "dgq_linear_fused: null pointer");
        DGQ_CHECK_ARG(dgq_linear_fused_supported(a.M, a.C, a.Kp, a.N, a.per_m, a.w_bits),
                      "dgq_linear_fused: unsupported shape M=%d C=%d Kp=%d N=%d w_bits=%d", a.M, a.C, a.Kp, a.N, a.w_bits);
        DGQ_CHECK_ARG(a.M == a0.M && a.C == a0.C && a.x_dtype == a0.x_dtype && a.y_dtype == a0.x_dtype && (a.per_m != 0) == (a0.per_m != 0) &&
                      a.Kp == a0.Kp, "dgq_linear_fused_batch: problem %d differs from problem 0 (rows / channels / dtype / scale mode / Kp)", i);
        DGQ_CHECK_ARG(a.a_bits >= 2 && a.a_bits <= 8 && a.ldy >= a.N, "dgq_linear_fused: bad a_bits / ldy");
        DGQ_CHECK_ARG(a.per_m ? (a.vn && a.L >= 1) : (a.kdst && a.cflush), "dgq_linear_fused: per_m needs vn/L, per-K needs kdst/cflush");
        DGQ_CHECK_ARG((a.pre_scale == nullptr) == (a.pre_shift == nullptr) && a.pre_act >= 0 && a.pre_act <= 2 && (!a.pre_scale || a.hw >= 1),
                      "dgq_linear_fused: bad prologue");
        DGQ_CHECK_ARG((a.ln_gamma == nullptr) == (a.ln_beta == nullptr) && (!a.ln_gamma || (a.ln_eps > 0.0f && !a.pre_scale && a.pre_act == 0)),
                      "dgq_linear_fused: LayerNorm prologue excludes the others");
        DGQ_CHECK_ARG((reinterpret_cast<uintptr_t>(a.x) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.wpacked) & 15) == 0,
                      "dgq_linear_fused: x / wpacked must be 16-byte aligned");
        FusedParams& p = bt.p[i];
        p.x = a.x; p.M = a.M; p.C = a.C; p.ldc = a.pre_act == 2 ? 2 * a.C : a.C; p.hw = a.hw > 0 ? a.hw : 1;
        p.kdst = a.per_m ? nullptr : a.kdst; p.qdelta = a.delta; p.qzp = a.zp; p.L = a.per_m ? a.L : 1;
        p.qmax = (float)((1 << a.a_bits) - 1); p.offset = (float)(1 << (a.a_bits - 1));
        p.pre_scale = a.pre_scale; p.pre_shift = a.pre_shift; p.pre_act = a.pre_act;
        p.ln_gamma = a.ln_gamma; p.ln_beta = a.ln_beta; p.ln_eps = a.ln_eps;
        p.Kp = a.Kp; p.N = a.N; p.wpacked = reinterpret_cast<const uint8_t*>(a.wpacked); p.cflush = a.cflush;
        p.alpha = a.alpha; p.zw = a.zw; p.gamma = a.gamma; p.vn = a.vn; p.y = a.y; p.ldy = a.ldy;
        if (a.extra) {
            p.ex = *a.extra;
            DGQ_CHECK_ARG(p.ex.fq_mode >= 0 && p.ex.fq_mode <= 3, "dgq_linear_fused: bad fq_mode");
            DGQ_CHECK_ARG(p.ex.fq_mode == 0 || (p.ex.fq_delta && p.ex.fq_zp && p.ex.fq_T > 0 && p.ex.fq_D > 0), "dgq_linear_fused: fused quantizer needs tables");
            DGQ_CHECK_ARG(!p.ex.residual || (p.ex.ldr >= a.N && p.ex.res_div >= 1 && p.ex.res_dtype >= DGQ_F32 && p.ex.res_dtype <= DGQ_BF16),
                          "dgq_linear_fused: bad residual descriptor");
        } else {
            p.ex.residual = nullptr; p.ex.ldr = 0; p.ex.res_div = 1; p.ex.res_dtype = DGQ_F32; p.ex.fq_mode = 0; p.ex.fq_delta = nullptr;
            p.ex.fq_zp = nullptr; p.ex.fq_T = 1; p.ex.fq_D = 1; p.ex.fq_skip = 0; p.ex.fq_qmax = 255.0f;
        }
        maxN = a.N > maxN ? a.N : maxN;
    }
    { const char* e = getenv("DGQ_FUSED_DEBUG"); bt.debug = e && *e ? atoi(e) : 0; }
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = fused2_lds_bytes(a0.Kp, a0.C, a0.per_m != 0);
    DGQ_CHECK_ARG(lds <= 150 * 1024, "dgq_linear_fused_batch: %zu bytes of LDS", lds);
    dim3 grid((a0.M + F2_BM - 1) / F2_BM, (maxN + F2_NB - 1) / F2_NB, n);
    const int nv = fused2_nv(a0.C);
    const bool pm = a0.per_m != 0;
    switch (a0.x_dtype) {
        case DGQ_F32: pm ? launch_fused2<float, true>(bt, nv, grid, lds, st) : launch_fused2<float, false>(bt, nv, grid, lds, st); break;
        case DGQ_F16: pm ? launch_fused2<__half, true>(bt, nv, grid, lds, st) : launch_fused2<__half, false>(bt, nv, grid, lds, st); break;
        case DGQ_BF16: pm ? launch_fused2<__hip_bfloat16, true>(bt, nv, grid, lds, st) : launch_fused2<__hip_bfloat16, false>(bt, nv, grid, lds, st); break;
        default: dgq_set_error("dgq_linear_fused: unknown dtype %d", a0.x_dtype); return DGQ_EINVAL;
    }
    return dgq_launch_status("dgq_linear_fused_batch");
}
