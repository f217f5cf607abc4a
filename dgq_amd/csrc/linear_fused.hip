// Quantise-on-load INSIDE the GEMM for Linear / 1x1-conv layers: one launch per layer instead of dgq_quant_act +
// dgq_gemm_wxa8, and no int8 operand in HBM (reference operator: QuantLayer.forward,
// quant/quant_layer.py:626-661 with F.linear at :659).
//
// A workgroup owns a PANEL of 32 activation rows and a range of 64-wide column tiles:
//   phase 1  the four waves read their 8 rows each (8 lanes per row, 16-byte loads, the whole row in registers), apply the
//            folded prologue (GroupNorm scale/shift, LayerNorm with the row statistics of dgq_quant_act — same summation
//            tree, bit for bit — SiLU, GEGLU), quantise with clamp(rne(x/δ)+z) (exact-division semantics, dgq_common.h)
//            and write the centred int8 codes straight into the MFMA operand image in LDS:
//              per-K (DGQ groups): every SOURCE element is quantised once with the (δ, z) of its destination chunk and its
//                      byte scattered to its packed position kdst[c] (the group-sorted, 64-padded K order of the weight);
//              per-M / scalar: 4 codes per dword at the natural position.
//            The image is laid out as the GEMM's A stage ([K tile][row][128 B], 16-byte chunk c at c ^ ((row>>1)&7)), so
//            phase 2 reads fragments from it with the same conflict-free ds_read_b128.
//   phase 2  for each column tile, for each K tile: int4 weights by LDS-DMA through a 6-stage ring (the activation operand
//            is already resident, so the ring carries only 4 KB per stage and runs ACROSS column tiles without draining),
//            V_MFMA_I32_16X16X64_I8, per-group fp32 flush (per-K), dequantising epilogue with the optional residual /
//            attention-side quantizer of dgq_gemm_wxa8, 16-byte stores through an LDS transpose.
// The activations of a panel are quantised once per workgroup; `nsplit` workgroups share a panel when the row count alone
// cannot fill the chip (each re-quantises it: 1-3 us of VALU against a saved launch and a saved HBM round trip).
#include <atomic>
#include <cstdlib>
#include "dgq_common.h"
#include "gemm_device.h"

#define FBM 32
#define FBN 64
#define FBK 128
#define FSTAGES 6
#define F_WSTAGE (FBN * FBK / 2)          // 4 KiB of packed int4 per stage
#define F_ABYTES (FBM * FBK)              // 4 KiB of codes per K tile of the panel
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
    int tiles_per_block;                  // column tiles per workgroup (the same for every problem of a batch)
    int debug;                            // development: 1 = stop after phase 1, 2 = skip phase 1 (DGQ_FUSED_DEBUG; timing only)
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

// byte offset of code (row r, packed position kp) inside the panel image
__device__ __forceinline__ int panel_addr(int r, int kp) {
    return (kp >> 7) * F_ABYTES + r * FBK + ((((kp & 127) >> 4) ^ ((r >> 1) & 7)) << 4) + (kp & 15);
}

// NV = float4 registers per lane holding one row slice: 8 lanes x NV x 4 floats >= C
template <typename T, bool PER_M, int NV>
__global__ __launch_bounds__(256) void linear_fused_kernel(FusedBatch bt) {
    const FusedParams& p = bt.p[blockIdx.z];
    const int n_tiles = (p.N + FBN - 1) / FBN;
    const int nt_begin = blockIdx.y * bt.tiles_per_block;
    const int m0 = blockIdx.x * FBM;
    if (nt_begin >= n_tiles || m0 >= p.M) return;                        // whole workgroup
    const int nt_cnt = min(bt.tiles_per_block, n_tiles - nt_begin);
    const int nk = p.Kp / FBK;
    const int nch = p.Kp >> 6;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);

    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t* panel = smem;                                               // [nk][32][128]
    uint8_t* ring = panel + (size_t)nk * F_ABYTES;                       // [FSTAGES][64][64]
    float* ep_all = reinterpret_cast<float*>(ring + FSTAGES * F_WSTAGE); // [4 waves][16][36]
    float* vtab = ep_all + 4 * 16 * 36;                                  // [3][32]: R0 R1 R2 per row
    float* vcol = vtab + 3 * FBM;                                        // [4][tiles_per_block*64]: alpha zw gamma vn
    const int ncol = bt.tiles_per_block * FBN;
    float* ctab = vcol + 4 * ncol;                                       // [nch] signed per-chunk scale (per-K)
    float* tdelta = ctab + nch;                                          // [nch] x3 (per-K)
    float* tinv = tdelta + nch;
    float* tzp = tinv + nch;
    int32_t* kd = reinterpret_cast<int32_t*>(tzp + nch);                 // [C] (per-K)

    // ---- weight ring: issue the first tiles before anything else (they fly during phase 1) -------------------------------
    const uint32_t ring_lds = (uint32_t)(uintptr_t)(lptr_t)ring;
    const int wrow = wid * 16 + (lane >> 2);                             // this lane's weight row inside a column tile
    const int wchunk = ((lane & 3) ^ ((wrow >> 2) & 3)) * 16;
    const int total = nt_cnt * nk;
    auto issue_w = [&](int s) {
        const int j = s / nk, kt = s - j * nk;
        const int n = min((nt_begin + j) * FBN + wrow, p.N - 1);
        const uint8_t* src = p.wpacked + (int64_t)n * (p.Kp / 2) + (int64_t)kt * (FBK / 2) + wchunk;
        glds16(src, __builtin_amdgcn_readfirstlane(ring_lds + (s % FSTAGES) * F_WSTAGE + wid * 1024));
    };
#pragma unroll
    for (int s = 0; s < FSTAGES - 1; ++s)
        if (s < total) issue_w(s);

    // ---- staging of tables + zeroed panel ----------------------------------------------------------------------------------
    for (int i = tid; i < nk * (F_ABYTES / 16); i += 256) reinterpret_cast<uint4*>(panel)[i] = make_uint4(0, 0, 0, 0);
    for (int c = tid; c < nt_cnt * FBN; c += 256) {
        const int n = min(nt_begin * FBN + c, p.N - 1);
        vcol[c] = p.alpha[n]; vcol[ncol + c] = p.zw[n]; vcol[2 * ncol + c] = p.gamma[n];
        vcol[3 * ncol + c] = PER_M ? p.vn[n] : 0.0f;
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

    // ---- phase 1: quantise the panel ------------------------------------------------------------------------------------------
    if (bt.debug != 2) {
        const int r = wid * 8 + (lane >> 3), sub = lane & 7;
        const int m = min(m0 + r, p.M - 1);
        const T* xr = reinterpret_cast<const T*>(p.x) + (int64_t)m * p.ldc;
        float v[NV][4];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = sub * 4 + 32 * i;
            if (c < p.C) f_load4<T>(xr + c, v[i]);
            else v[i][0] = v[i][1] = v[i][2] = v[i][3] = 0.0f;
        }
        if (p.pre_scale) {                                               // folded GroupNorm: x·scale[b,c] + shift[b,c]
            const int b = m / p.hw;
            const float* sc = p.pre_scale + (int64_t)b * p.C;
            const float* sh = p.pre_shift + (int64_t)b * p.C;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = sub * 4 + 32 * i;
                if (c < p.C) {
                    const float4 a = *reinterpret_cast<const float4*>(sc + c), s4 = *reinterpret_cast<const float4*>(sh + c);
                    v[i][0] = v[i][0] * a.x + s4.x; v[i][1] = v[i][1] * a.y + s4.y;
                    v[i][2] = v[i][2] * a.z + s4.z; v[i][3] = v[i][3] * a.w + s4.w;
                }
            }
        }
        if (p.ln_gamma) {
            // Row statistics exactly as dgq_quant_act's row_layernorm_stats (64 virtual lanes L = (c/4) % 64 each summing
            // its 256-strided float4s in order, then a xor-butterfly over L): here L = sub + 8·(i % 8), so the xor 32/16/8
            // steps pair registers of this lane and the xor 4/2/1 steps pair the 8 lanes of the row.
            float t[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                t[i] = 0.0f;
#pragma unroll
                for (int u = i; u < NV; u += 8) t[i] += (v[u][0] + v[u][1]) + (v[u][2] + v[u][3]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) t[i] = t[i] + t[i + 4];
            t[0] = t[0] + t[2]; t[1] = t[1] + t[3];
            float s = t[0] + t[1];
            s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 1, 64);
            const float mu = s / (float)p.C;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                t[i] = 0.0f;
#pragma unroll
                for (int u = i; u < NV; u += 8) {
                    if (sub * 4 + 32 * u < p.C) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) t[i] += (v[u][j] - mu) * (v[u][j] - mu);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) t[i] = t[i] + t[i + 4];
            t[0] = t[0] + t[2]; t[1] = t[1] + t[3];
            float q = t[0] + t[1];
            q += __shfl_xor(q, 4, 64); q += __shfl_xor(q, 2, 64); q += __shfl_xor(q, 1, 64);
            const float rstd = 1.0f / sqrtf(q / (float)p.C + p.ln_eps);
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = sub * 4 + 32 * i;
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
                const int c = sub * 4 + 32 * i;
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
                const int c = sub * 4 + 32 * i;
                if (c < p.C) {
                    uint32_t w = 0;
                    float fsum = 0.0f;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float b8 = dgq_affine_code_fast(v[i][j], md, minv, mz, p.qmax) + bias;   // q − off + 128
                        w = __builtin_amdgcn_cvt_pk_u8_f32(b8, j, w);
                        fsum += b8;
                    }
                    *reinterpret_cast<uint32_t*>(panel + panel_addr(r, c)) = w ^ 0x80808080u;
                    partial += fsum - 512.0f;
                }
            }
            partial += __shfl_xor(partial, 4, 64); partial += __shfl_xor(partial, 2, 64); partial += __shfl_xor(partial, 1, 64);
            if (sub == 0) { vtab[r] = md; vtab[FBM + r] = md * partial; vtab[2 * FBM + r] = md * (p.offset - mz); }
        } else {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = sub * 4 + 32 * i;
                if (c < p.C) {
                    const int4 d4 = *reinterpret_cast<const int4*>(kd + c);
                    const int dst[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int ch = dst[j] >> 6;
                        const float d = tdelta[ch];
                        const float sc = dgq_affine_code_fast(v[i][j], d, tinv[ch], tzp[ch], p.qmax) - p.offset;
                        panel[panel_addr(r, dst[j])] = (uint8_t)(int)sc;
                        partial += d * sc;
                    }
                }
            }
            partial += __shfl_xor(partial, 4, 64); partial += __shfl_xor(partial, 2, 64); partial += __shfl_xor(partial, 1, 64);
            if (sub == 0) { vtab[r] = 1.0f; vtab[FBM + r] = partial; vtab[2 * FBM + r] = 0.0f; }
        }
    }
    // every DMA piece issued so far is older than anything below; the panel is complete after this barrier
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (bt.debug == 1) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); return; }

    // ---- phase 2: column tiles x K tiles --------------------------------------------------------------------------------------
    const int wave_m = wid >> 1, wave_n = wid & 1;                          // 2 x 2 waves, 16 x 32 outputs each
    const int fr = lane & 15, fq = lane >> 4;
    int a_off[2], w_off[2][2];
    {
        const int row = wave_m * 16 + fr;
#pragma unroll
        for (int h = 0; h < 2; ++h) a_off[h] = row * FBK + (((4 * h + fq) ^ ((row >> 1) & 7)) << 4);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int wr = wave_n * 32 + j * 16 + fr;
#pragma unroll
            for (int h = 0; h < 2; ++h) w_off[j][h] = wr * (FBK / 2) + (((4 * h + fq) ^ (((wr >> 2) & 3) << 1)) << 3);
        }
    }
    v4i acc[2];
    v4f accf[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) { acc[j] = (v4i){0, 0, 0, 0}; accf[j] = (v4f){0.f, 0.f, 0.f, 0.f}; }
    float* ep = ep_all + wid * 16 * 36;
    T* y = reinterpret_cast<T*>(p.y);
    const int c4 = (lane & 7) * 4, lrow = lane >> 3;                        // epilogue: 8 lanes per row, 8 rows per pass

    // the first tile must have landed before the loop's first read (this wave's piece: all but the younger ones done)
    {
        const int younger = min(FSTAGES - 2, max(0, total - 1));
        switch (younger) {
            case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        }
    }
    __builtin_amdgcn_s_barrier();

    for (int s = 0; s < total; ++s) {
        const int j = s / nk, kt = s - j * nk;
        if (s + FSTAGES - 1 < total) issue_w(s + FSTAGES - 1);              // into the stage read in iteration s − 1
        const uint8_t* sa = panel + kt * F_ABYTES;
        const uint8_t* sw = ring + (s % FSTAGES) * F_WSTAGE;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const v4i af = *reinterpret_cast<const v4i*>(sa + a_off[h]);
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const uint2 wv = *reinterpret_cast<const uint2*>(sw + w_off[jj][h]);
                const v4i bf = (v4i){(int)(wv.x & 0x0F0F0F0Fu), (int)((wv.x >> 4) & 0x0F0F0F0Fu),
                                     (int)(wv.y & 0x0F0F0F0Fu), (int)((wv.y >> 4) & 0x0F0F0F0Fu)};
                acc[jj] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af, bf, acc[jj], 0, 0, 0);
            }
            if (!PER_M) {
                const float sc = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(
                    __builtin_bit_cast(int, ctab[2 * kt + h])));
                if (sc != 0.0f) {
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) accf[jj][r] = __builtin_fmaf(sc, (float)acc[jj][r], accf[jj][r]);
                    }
                }
            }
        }
        if (kt == nk - 1) {
            // ---- epilogue of column tile nt_begin + j (same arithmetic as dgq_gemm_wxa8) ---------------------------------
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    ep[(fq * 4 + r) * 36 + jj * 16 + fr] = PER_M ? (float)acc[jj][r] : accf[jj][r];
                acc[jj] = (v4i){0, 0, 0, 0};
                accf[jj] = (v4f){0.f, 0.f, 0.f, 0.f};
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // same-wave LDS round trip
            const int nb = (nt_begin + j) * FBN + wave_n * 32 + c4;
            const float* vc = vcol + j * FBN + wave_n * 32 + c4;
            const float4 al = *reinterpret_cast<const float4*>(vc);
            const float4 zw = *reinterpret_cast<const float4*>(vc + ncol);
            const float4 ga = *reinterpret_cast<const float4*>(vc + 2 * ncol);
            const float4 vn = *reinterpret_cast<const float4*>(vc + 3 * ncol);
            const bool vec_ok = (nb + 3 < p.N);
            const bool st_vec = vec_ok && ((p.ldy * (int)sizeof(T)) % 16 == 0) && ((reinterpret_cast<uintptr_t>(p.y) & 15) == 0) &&
                                (sizeof(T) == 4 || (p.ldy & 3) == 0);
            const bool has_extra = p.ex.fq_mode != 0 || p.ex.residual != nullptr;
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const int row = rr * 8 + lrow;
                const int m = m0 + wave_m * 16 + row;
                if (m >= p.M || nb >= p.N) continue;
                const float4 v = *reinterpret_cast<const float4*>(ep + row * 36 + c4);
                const float* vr = vtab + wave_m * 16 + row;
                const float r0 = vr[0], r1 = vr[FBM], r2 = vr[2 * FBM];
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
                if (st_vec) {
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
        // tile s + 1 must have landed (this wave's piece) and this wave's reads of stage s must be complete before the
        // barrier; vmcnt counts DMA pieces AND the epilogue's loads / stores in issue order, so waiting for "all but the
        // FSTAGES − 2 youngest" is exact in the steady state and only conservative right after an epilogue.
        if (s + 1 < total) {
            const int younger = min(FSTAGES - 2, total - 2 - s);
            switch (younger) {
                case 0: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); break;
                case 1: asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory"); break;
                case 2: asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory"); break;
                case 3: asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory"); break;
                default: asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory"); break;
            }
            __builtin_amdgcn_s_barrier();
        }
    }
}

// ----------------------------------------------------------------------------------------------------------------- host side
static size_t fused_lds_bytes(int Kp, int C, int tiles_per_block, bool per_m) {
    size_t b = (size_t)(Kp / FBK) * F_ABYTES + (size_t)FSTAGES * F_WSTAGE + 4 * 16 * 36 * 4 + 3 * FBM * 4 +
               (size_t)4 * tiles_per_block * FBN * 4;
    if (!per_m) b += (size_t)4 * (Kp >> 6) * 4 + (size_t)C * 4;
    return (b + 15) & ~(size_t)15;
}

static int fused_nv(int C) { return C <= 320 ? 10 : (C <= 768 ? 24 : (C <= 1280 ? 40 : 0)); }

// column tiles per workgroup: all of them when the row panels alone fill the chip, otherwise split so that the grid reaches
// ~2 workgroups per CU (each split re-quantises its panel)
static int fused_tiles_per_block(int M, int N) {
    const char* e = getenv("DGQ_FUSED_NSPLIT");                // development hook, read per call
    const int forced = e && *e ? atoi(e) : 0;
    const int n_tiles = (N + FBN - 1) / FBN;
    const int panels = (M + FBM - 1) / FBM;
    int nsplit = forced > 0 ? forced : (512 + panels - 1) / panels;
    if (nsplit > n_tiles) nsplit = n_tiles;
    if (nsplit < 1) nsplit = 1;
    int tpb = (n_tiles + nsplit - 1) / nsplit;
    if (tpb > 12) tpb = 12;                                   // bounds the per-column tables in LDS and the serial tile loop
    return tpb;
}

extern "C" int dgq_linear_fused_supported(int M, int C, int Kp, int N, int per_m, int w_bits) {
    if (w_bits != 4 || M < 1 || C % 4 != 0 || fused_nv(C) == 0 || Kp % FBK != 0 || Kp > 2048 || Kp < C) return 0;
    const int tpb = fused_tiles_per_block(M, N);
    return fused_lds_bytes(Kp, C, tpb, per_m != 0) <= 150 * 1024 ? 1 : 0;
}

template <typename T, bool PER_M, int NV>
static void launch_fused_nv(const FusedBatch& bt, int n, dim3 grid, size_t lds, hipStream_t st) {
    static std::atomic<bool> attr_set[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !attr_set[dev].load(std::memory_order_acquire)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_fused_kernel<T, PER_M, NV>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
        if (dev >= 0 && dev < 64) attr_set[dev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL((linear_fused_kernel<T, PER_M, NV>), grid, dim3(256), lds, st, bt);
}

template <typename T, bool PER_M>
static void launch_fused(const FusedBatch& bt, int n, int nv, dim3 grid, size_t lds, hipStream_t st) {
    if (nv == 10) launch_fused_nv<T, PER_M, 10>(bt, n, grid, lds, st);
    else if (nv == 24) launch_fused_nv<T, PER_M, 24>(bt, n, grid, lds, st);
    else launch_fused_nv<T, PER_M, 40>(bt, n, grid, lds, st);
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
    bt.tiles_per_block = fused_tiles_per_block(a0.M, maxN);
    { const char* e = getenv("DGQ_FUSED_DEBUG"); bt.debug = e && *e ? atoi(e) : 0; }
    const int n_tiles = (maxN + FBN - 1) / FBN;
    const size_t lds = fused_lds_bytes(a0.Kp, a0.C, bt.tiles_per_block, a0.per_m != 0);
    DGQ_CHECK_ARG(lds <= 150 * 1024, "dgq_linear_fused_batch: %zu bytes of LDS", lds);
    dim3 grid((a0.M + FBM - 1) / FBM, (n_tiles + bt.tiles_per_block - 1) / bt.tiles_per_block, n);
    hipStream_t st = (hipStream_t)stream;
    const int nv = fused_nv(a0.C);
    const bool pm = a0.per_m != 0;
    switch (a0.x_dtype) {
        case DGQ_F32: pm ? launch_fused<float, true>(bt, n, nv, grid, lds, st) : launch_fused<float, false>(bt, n, nv, grid, lds, st); break;
        case DGQ_F16: pm ? launch_fused<__half, true>(bt, n, nv, grid, lds, st) : launch_fused<__half, false>(bt, n, nv, grid, lds, st); break;
        case DGQ_BF16: pm ? launch_fused<__hip_bfloat16, true>(bt, n, nv, grid, lds, st) : launch_fused<__hip_bfloat16, false>(bt, n, nv, grid, lds, st); break;
        default: dgq_set_error("dgq_linear_fused: unknown dtype %d", a0.x_dtype); return DGQ_EINVAL;
    }
    return dgq_launch_status("dgq_linear_fused_batch");
}
