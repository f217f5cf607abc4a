// Device-side pieces shared by the attention translation units (attn_bf16x3.hip: pre-pass, statistics, P·V; attn_one.hip: the
// single-launch form for short key ranges): operand geometry, LDS-DMA helpers, Q fragment loads and the three score-tile forms.
// See attn_bf16x3.hip for the algorithm.
#pragma once
#include <atomic>
#include <type_traits>
#include "dgq_common.h"
#include "diag.h"

typedef float v16f __attribute__((ext_vector_type(16)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define KT 32
#ifndef DGQ_ATTN_TPS_STATS
#define DGQ_ATTN_TPS_STATS 2   // key tiles per ring stage of the 8-wave statistics launches (see attn3_stats_kernel)
#endif
#ifndef DGQ_ATTN_TPS_PV
#define DGQ_ATTN_TPS_PV 2      // ... of the 8-wave P·V launches (where two stages of that many tile images fit the LDS)
#endif
#ifndef DGQ_PV_ST64
#define DGQ_PV_ST64 2          // P·V ring depth at D = 64 with small (int8 K + one-plane V) images
#endif
// real-time δ (the tensor-wide maximum probability): the statistics pass leaves one maximum per workgroup in slot
// (workgroup index mod 64) of the 256-byte δ area and the P̂·V pass takes the maximum of the 64 slots — 2048 waves hitting
// ONE address with atomicMax took 25 us of a 30 us launch (4096 queries x 77 keys)
#define DELTA_SLOTS 64
#define DELTA_GRANULES 1024     // behind the slots (byte offset 512): one 8-byte {tag, maximum} granule per workgroup of the single-launch form (attn_one.hip)
#define DELTA_AREA_BYTES (512 + 8 * DELTA_GRANULES)
typedef float f2 __attribute__((ext_vector_type(2)));      // operand pair of v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32
#ifndef DGQ_ATTN_PK
#define DGQ_ATTN_PK 1          // 1: v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 on score pairs; 0: scalar instructions (A/B builds)
#endif
#if DGQ_ATTN_PK
__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 pk_mul(f2 a, f2 b) { return a * b; }
__device__ __forceinline__ f2 pk_add(f2 a, f2 b) { return a + b; }
#else
__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return f2{fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y)}; }
__device__ __forceinline__ f2 pk_mul(f2 a, f2 b) { return f2{a.x * b.x, a.y * b.y}; }
__device__ __forceinline__ f2 pk_add(f2 a, f2 b) { return f2{a.x + b.x, a.y + b.y}; }
#endif
#define LOG2E 1.4426950408889634f

// optional UniformAffineQuantizer applied to q / k / v as they are loaded (aqtizer_q/k/v, sd.py:165-181): same
// addressing as dgq_fakequant_rows — mode 0 scalar, 1 per token (entry t − skip), 2 per head-dim element; tokens
// < skip pass through (start_peak).  mode < 0: none.
struct FqDesc {
    int mode;
    int skip;
    float qmax;
    const float* delta;
    const float* zp;
};

__device__ __forceinline__ float fq_apply(const FqDesc& f, float x, int t, int d) {
    if (f.mode < 0 || t < f.skip) return x;
    const int idx = f.mode == 0 ? 0 : (f.mode == 1 ? t - f.skip : d);
    const float dl = f.delta[idx], z = f.zp[idx];
    return dl * (dgq_affine_code_fast(x, dl, dgq_rcp(dl), z, f.qmax) - z);
}

// eight consecutive head-dim elements d0 .. d0+7 of token t (d0 % 8 == 0): the table entries are fetched once per
// token (modes 0/1) or as four 16-byte loads (mode 2) instead of two dependent loads per element
__device__ __forceinline__ void fq_apply8(const FqDesc& f, float (&x)[8], int t, int d0) {
    if (f.mode < 0 || t < f.skip) return;
    if (f.mode == 2) {
        const float4 da = *reinterpret_cast<const float4*>(f.delta + d0), db = *reinterpret_cast<const float4*>(f.delta + d0 + 4);
        const float4 za = *reinterpret_cast<const float4*>(f.zp + d0), zb = *reinterpret_cast<const float4*>(f.zp + d0 + 4);
        const float dl[8] = {da.x, da.y, da.z, da.w, db.x, db.y, db.z, db.w};
        const float zz[8] = {za.x, za.y, za.z, za.w, zb.x, zb.y, zb.z, zb.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = dl[j] * (dgq_affine_code_fast(x[j], dl[j], dgq_rcp(dl[j]), zz[j], f.qmax) - zz[j]);
    } else {
        const int idx = f.mode == 0 ? 0 : t - f.skip;
        const float dl = f.delta[idx], z = f.zp[idx], inv = dgq_rcp(dl);
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = dl * (dgq_affine_code_fast(x[j], dl, inv, z, f.qmax) - z);
    }
}

struct AttnParams {
    FqDesc fq[3];          // q, k, v
    const float* q;        // fp32 queries: the caller's tensor, or the scratch copy written by the pre-pass
    const void* k;         // k / v / o in the caller's dtype (io_dtype)
    const void* v;
    void* o;
    int io_dtype;
    int B, H, T, S;
    float scale;
    int mode;              // 1: log2 real-time δ, 2: log2 static δ, 3: uniform (δ, z = 0)
    int skip;
    float qmax;
    float* stats;          // [B*H][T][2] : m (log2 units), l
    float* stats_part;     // key-split launches (gridDim.z = 2): [2][B*H][T][4] = m, l, m2 (maximum without the bypassed keys) per key half
    float* o_part;         // key-split launches: the second key half's part of o ([B][T][H][D] fp32; attn3_add_kernel adds it to o)
    float* o_part0;        // key-split launches on 16-bit tensors: the FIRST half's part, fp32 like the second (attn3_add16_kernel stores o = round(part0 + part))
    float* delta;
    const unsigned char* planes;   // [B*H][NT] tile images of the bf16 split planes (Geo<D>::IMG_BYTES each)
    int NT;                        // 32-key tiles per (batch, head)
    const int8_t* qcodes;          // QI8: [B][T][H][DP32] centred int8 codes of aqtizer_q(q), zero padded
    const float* qtab;             // QI8: [B][T][H][4] = δq, z'q, Σ_d c'q − D·z'q, start-peak score / δq
    int img_bytes;                 // bytes of one tile image in global memory (depends on the K / V plane formats)
    int kskip;                     // QI8: leading keys that bypass aqtizer_k (start-peak key 0: exact fp32 rank-1 score)
    int xcd;                       // 1: workgroups of one (batch, head) share an XCD (its K/V tile images stay in one L2)
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

template <typename TIn> __device__ __forceinline__ void load8(const TIn* p, float (&x)[8]);
template <> __device__ __forceinline__ void load8<float>(const float* p, float (&x)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p), c = *reinterpret_cast<const float4*>(p + 4);
    x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = c.x; x[5] = c.y; x[6] = c.z; x[7] = c.w;
}
template <> __device__ __forceinline__ void load8<__half>(const __half* p, float (&x)[8]) {
    const uint4 t = *reinterpret_cast<const uint4*>(p);
    const __half* h = reinterpret_cast<const __half*>(&t);
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = __half2float(h[j]);
}
template <> __device__ __forceinline__ void load8<__hip_bfloat16>(const __hip_bfloat16* p, float (&x)[8]) {
    const uint4 t = *reinterpret_cast<const uint4*>(p);
    const unsigned short* h = reinterpret_cast<const unsigned short*>(&t);
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = __uint_as_float(((unsigned)h[j]) << 16);
}
__device__ __forceinline__ float load_any(const void* p, int dtype, int64_t i) {
    if (dtype == DGQ_F16) return __half2float(reinterpret_cast<const __half*>(p)[i]);
    if (dtype == DGQ_BF16) return __bfloat162float(reinterpret_cast<const __hip_bfloat16*>(p)[i]);
    return reinterpret_cast<const float*>(p)[i];
}
__device__ __forceinline__ void store_any(void* p, int dtype, int64_t i, float v) {
    if (dtype == DGQ_F16) reinterpret_cast<__half*>(p)[i] = __float2half(v);
    else if (dtype == DGQ_BF16) reinterpret_cast<__hip_bfloat16*>(p)[i] = __float2bfloat16(v);
    else reinterpret_cast<float*>(p)[i] = v;
}

// QM: how Q·K^T is formed — 0: three bf16 planes each (six products); 1 (QI8): int8 codes on both sides; 2 (Q1K3): ONE
// plane of centred Q codes (exact in bf16) against three K planes (pre-scaled by δq(d) for a per-head-dim aqtizer_q) with a
// rank-1 zero-point correction per key — three products, for every aqtizer_q the int8 path cannot take.
template <int D, int QM = 0, bool VINT = false> struct Geo {
    static constexpr bool QI8 = QM == 1 || QM == 3;     // 3: int8 scores with a SCALAR aqtizer_k (δk, z'k folded into per-query constants)
    static constexpr int DP = (D + 15) / 16 * 16;      // K depth of the score product
    static constexpr int NKK = DP / 16;
    static constexpr int NDT = (D + 31) / 32;          // 32-wide d tiles of O^T
    static constexpr int DV = NDT * 32;
    static constexpr int KLD = DP + 8;                 // bf16 elements per K row (16-byte aligned, de-conflicted)
    static constexpr int VLD = KT + 8;                 // bf16 elements per V^T row
    // QI8: int8 K codes [KT][K8_LD] (D zero-padded to a multiple of 32 = one MFMA_I32_32X32X32_I8 step, + 16 B of row
    // padding) followed by the per-key table [3][KT] floats (δk, −z'k, −Σ_d c'k)
    static constexpr int DP32 = (D + 31) / 32 * 32;
    static constexpr int NK32 = DP32 / 32;
    static constexpr int K8_LD = DP32 + 16;
    static constexpr int K8_BYTES = KT * K8_LD;
    static constexpr int KTAB_BYTES = 3 * KT * 4;
    // bf16-element count of the K part of the image (QI8: bytes / 2, a multiple of 8 so that V stays 16-byte aligned)
    // Q1K3 appends one float per key (Σ_d w(d)·K̃[s][d], the zero-point correction) to the three planes
    static constexpr int K_ELEMS = QI8 ? (K8_BYTES + KTAB_BYTES) / 2 : 3 * KT * KLD + (QM == 2 ? 2 * KT : 0);
    // VINT (scalar / per-head-dim aqtizer_v): ONE plane of centred integer codes c'v (exact bf16) instead of three planes
    // of the dequantised values — P̂·V is then a single bf16 product, δv(d) and the zero point move to the epilogue
    static constexpr int V_PLANES = VINT ? 1 : 3;
    // Q1K3 with >= 3 zero-padded depth slots (D = 8, 40): the per-key zero-point correction −zmul(t)·tv[s] rides in the
    // padding of the score product itself — K slots D..D+2 each hold the three-way split of tv[s], the query's slots hold
    // the three bf16 terms of −zmul(t) — instead of one fma per score behind it
    static constexpr bool FOLDZ = QM == 2 && (DP - D) >= 3;
    // VINT with a spare row in the last d tile (D = 8, 16, 40, 80): V^T row D is all ones, so Σ_s p̂/δ (the zero-point term
    // of V) comes out of the P̂·V product as O^T[D] instead of one add per score
    static constexpr bool VONES = VINT && (NDT * 32 > D);
    static constexpr int V_ELEMS = V_PLANES * DV * VLD;
    // One 32-key tile of a (batch, head) is ONE contiguous image in global memory, laid out exactly as it sits in LDS
    // (K planes [3][KT][KLD] then V^T planes [3][DV][VLD], padding included), so staging is a flat LDS-DMA copy in
    // 1-KB pieces (64 lanes x 16 B) with no registers in between.
    static constexpr int K_PIECES = (2 * K_ELEMS + 1023) / 1024;              // statistics pass: K part only
    static constexpr int IMG_PIECES = (2 * (K_ELEMS + V_ELEMS) + 1023) / 1024;
    static constexpr int IMG_BYTES = IMG_PIECES * 1024;
    // ring depths: prefetch distance STAGES-1 tiles; sized so that two blocks share a CU's 160 KB where the grid is
    // large (D = 40: 3 x 26 KB) and by what fits otherwise
    static constexpr int STATS_STAGES = D <= 80 ? 4 : 3;
    static constexpr int PV_STAGES = D == 64 ? (IMG_BYTES <= 16 * 1024 ? DGQ_PV_ST64 : 2) : ((D <= 40 || D == 80) ? 3 : 2);
};

template <int D> using GeoI8 = Geo<D, 1, false>;        // the K part does not depend on VINT

typedef __attribute__((address_space(3))) void* lptr_t;

// LDS-DMA of 16 B per lane: LDS[lds_addr + lane*16] <- *gsrc (inline asm: see gemm_wxa8.hip — the builtin form makes
// hipcc drain every DMA with vmcnt(0) before the next ds_read; the ring below is ordered by counted vmcnt + barrier).
__device__ __forceinline__ void glds16(const void* gsrc, uint32_t lds_addr) {
    // M0 (the wave-uniform LDS base of the DMA) is an INPUT OPERAND bound to the physical register with "{m0}": hipcc
    // materialises the s_mov_b32 m0 itself and tracks the register like any other — nothing is clobbered behind its back.
    asm volatile("s_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gsrc), "{m0}"(lds_addr) : "memory");
}

// this wave's share of one tile image: pieces wid, wid+NW, ... < NP (NW waves per block)
template <int NP, int NW>
__device__ __forceinline__ void issue_image(const unsigned char* img_lane, uint32_t lds_stage, int wid) {
#pragma unroll
    for (int i = 0; i < (NP + NW - 1) / NW; ++i) {
        const int j = wid + NW * i;
        if (j < NP) glds16(img_lane + 1024 * j, __builtin_amdgcn_readfirstlane(lds_stage + 1024 * j));
    }
}

// wait until all but the youngest YOUNGER tiles of this wave's DMA pieces have landed (vmcnt counts in issue order;
// a wave issues ceil or floor of NP/NW pieces per tile depending on its index)
template <int NP, int YOUNGER, int NW>
__device__ __forceinline__ void wait_image(int wid) {
    constexpr int HI = (NP + NW - 1) / NW, LO = NP / NW;
    static_assert(HI * YOUNGER <= 63, "vmcnt immediate");
    if (HI == LO || wid < NP % NW) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(HI * YOUNGER) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LO * YOUNGER) : "memory");
}

// One pre-pass per attention call: fake-quantise (optional) and split K and V exactly into three bf16 planes, written
// as the per-tile LDS images described above.  V is stored transposed with the keys of a tile permuted into the k
// order of an accumulator tile used as the next MFMA's B operand: key 16s + 8a + 4h + b -> slot 16s + 8h + 4a + b.
// Every Q block of a (batch, head) re-reads these images; splitting inside the main loop cost more than the MFMAs.
// centred integer code c' = clamp(rne(x/δ)+z, 0, qmax) − off of one element (the reference's code, bit for bit)
__device__ __forceinline__ float fq_code(float x, float dl, float inv, float z, float qmax, float off) {
    return dgq_affine_code_fast(x, dl, inv, z, qmax) - off;
}


// Q rows of this lane as B-operand fragments: qf[split][kk] holds Q[t][16kk + 8h + j], j = 0..7
template <int D>
__device__ __forceinline__ void load_q(bf16x8 (&qf)[3][Geo<D>::NKK], const float* qrow, int h32, const FqDesc& fq, int t) {
    using G = Geo<D>;
#pragma unroll
    for (int kk = 0; kk < G::NKK; ++kk) {
        const int d0 = 16 * kk + 8 * h32;                   // D % 8 == 0: the 8 elements are all inside D or all padding
        float x[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        if (d0 < D) {
            const float4 a = *reinterpret_cast<const float4*>(qrow + d0), c = *reinterpret_cast<const float4*>(qrow + d0 + 4);
            x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = c.x; x[5] = c.y; x[6] = c.z; x[7] = c.w;
            fq_apply8(fq, x, t, d0);
        }
        unsigned wh[4], wm[4], wl[4];                       // packed pairs (no sub-dword arrays: those go to scratch)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            unsigned short h0, m0, l0, h1, m1, l1;
            split3(x[2 * j], h0, m0, l0);
            split3(x[2 * j + 1], h1, m1, l1);
            wh[j] = (unsigned)h0 | ((unsigned)h1 << 16);
            wm[j] = (unsigned)m0 | ((unsigned)m1 << 16);
            wl[j] = (unsigned)l0 | ((unsigned)l1 << 16);
        }
        qf[0][kk] = __builtin_bit_cast(bf16x8, make_uint4(wh[0], wh[1], wh[2], wh[3]));
        qf[1][kk] = __builtin_bit_cast(bf16x8, make_uint4(wm[0], wm[1], wm[2], wm[3]));
        qf[2][kk] = __builtin_bit_cast(bf16x8, make_uint4(wl[0], wl[1], wl[2], wl[3]));
    }
}

// S^T tile: acc[r] = Σ_d K[key_of(r,h)][d]·Q[t][d] (unscaled), six bf16 products per 16-deep step
template <int D>
__device__ __forceinline__ v16f score_tile(const unsigned short* kb, const bf16x8 (&qf)[3][Geo<D>::NKK], int lane) {
    using G = Geo<D>;
    // D >= 64 (one block per CU, a lone wave per SIMD): two independent accumulator chains — a dependent MFMA cannot
    // issue until its predecessor has left the pipe and nothing else fills that gap (measured: D=160 86 -> 74 us, D=80
    // 121 -> 117).  D <= 40 runs two waves per SIMD that fill each other's gaps; there the 16 extra adds cost 5 %.
    constexpr bool DUAL = D >= 64;
    v16f acc, acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = acc2[r] = 0.0f;
    const unsigned short* kp = kb + (lane & 31) * G::KLD + 8 * (lane >> 5);
    constexpr int PL = KT * G::KLD;
#pragma unroll
    for (int kk = 0; kk < G::NKK; ++kk) {
        const bf16x8 kh = *reinterpret_cast<const bf16x8*>(kp + 16 * kk);
        const bf16x8 km = *reinterpret_cast<const bf16x8*>(kp + PL + 16 * kk);
        const bf16x8 kl = *reinterpret_cast<const bf16x8*>(kp + 2 * PL + 16 * kk);
        if (DUAL) {
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl, qf[0][kk], acc2, 0, 0, 0);   // smallest terms first
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(km, qf[0][kk], acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qf[2][kk], acc2, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qf[1][kk], acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(km, qf[1][kk], acc2, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qf[0][kk], acc, 0, 0, 0);
        } else {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl, qf[0][kk], acc, 0, 0, 0);     // smallest terms first
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qf[2][kk], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(km, qf[1][kk], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(km, qf[0][kk], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qf[1][kk], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qf[0][kk], acc, 0, 0, 0);
        }
    }
    if (DUAL) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += acc2[r];
    }
    return acc;
}

__device__ __forceinline__ int key_of(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// QI8: this lane's query as B-operand fragments of V_MFMA_I32_32X32X32_I8 (lane l: column l&31, k = 16·(l>>5) + byte j)
// and its table entry (δq, z'q, Σc'q − D·z'q, start-peak score / δq)
template <int D>
__device__ __forceinline__ void load_q_i8(v4i (&qc)[GeoI8<D>::NK32], float4& qt, const AttnParams& p, int64_t row, int h32) {
    using G = Geo<D, true>;
    const int8_t* src = p.qcodes + row * G::DP32 + 16 * h32;
#pragma unroll
    for (int kk = 0; kk < G::NK32; ++kk) qc[kk] = *reinterpret_cast<const v4i*>(src + 32 * kk);
    qt = *reinterpret_cast<const float4*>(p.qtab + row * 4);
}

// QI8 S^T tile in units of δq: acc[r] = δk(s)·(Σ_d c'k c'q − z'q Σ_d c'k − z'k (Σ_d c'q − D z'q)), s = key_of(r, h);
// the bypassed start-peak key (tile 0, key 0: r = 0 of the lower half-wave) gets its exact fp32 rank-1 score.
// KS (scalar aqtizer_k: one δk and one z'k for every key): the tile is returned in units of δq·δk and WITHOUT the constant
// cq = −z'k·(Σ_d c'q − D z'q) of its query — acc[r] = Σ_d c'k c'q − z'q Σ_d c'k — the caller folds δk into its log2 scale and
// cq into its offsets (the softmax is shift-invariant): one packed fma per score pair instead of two and a multiply.
template <int D, bool KS>
__device__ __forceinline__ v16f score_tile_i8(const unsigned char* kimg, const v4i (&qc)[GeoI8<D>::NK32], const float4& qt,
                                              int lane, bool bypass_key0, float inv_dk, float cq) {
    using G = Geo<D, true>;
    v16i acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0;
    const unsigned char* kp = kimg + (lane & 31) * G::K8_LD + 16 * (lane >> 5);
#pragma unroll
    for (int kk = 0; kk < G::NK32; ++kk) {
        const v4i kf = *reinterpret_cast<const v4i*>(kp + 32 * kk);
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(kf, qc[kk], acc, 0, 0, 0);
    }
    const float* ktab = reinterpret_cast<const float*>(kimg + G::K8_BYTES);
    const int h32 = lane >> 5;
    const f2 zq = {qt.y, qt.y}, zs = {qt.z, qt.z};
    v16f out;
#pragma unroll
    for (int g = 0; g < 4; ++g) {                          // keys 8g + 4h .. +3: four consecutive table entries
        const float4 ns = *reinterpret_cast<const float4*>(ktab + 2 * KT + 8 * g + 4 * h32);
        float4 dk, nz;
        if constexpr (!KS) {
            dk = *reinterpret_cast<const float4*>(ktab + 8 * g + 4 * h32);
            nz = *reinterpret_cast<const float4*>(ktab + KT + 8 * g + 4 * h32);
        }
#pragma unroll
        for (int e = 0; e < 4; e += 2) {
            const f2 a = {(float)acc[4 * g + e], (float)acc[4 * g + e + 1]};
            f2 v = pk_fma(zq, e ? f2{ns.z, ns.w} : f2{ns.x, ns.y}, a);        // Σ c'k c'q − z'q·Σ c'k
            if constexpr (!KS) {
                v = pk_fma(zs, e ? f2{nz.z, nz.w} : f2{nz.x, nz.y}, v);       // − z'k·(Σ c'q − D z'q)
                v = pk_mul(v, e ? f2{dk.z, dk.w} : f2{dk.x, dk.y});
            }
            out[4 * g + e] = v.x;
            out[4 * g + e + 1] = v.y;
        }
    }
    if (bypass_key0 && h32 == 0) out[0] = KS ? fmaf(qt.w, inv_dk, -cq) : qt.w;
    return out;
}

// Q1K3 S^T tile in units of the query scale: acc[r] = Σ_d c'q[t][d]·K̃[s][d] − zmul·tv[s] (three bf16 products per 16-deep
// step: the Q plane is exact, K̃ is split three ways)
// Q1K3: the query's bf16 code plane as B-operand fragments, straight from the pre-pass's image (no split)
template <int D>
__device__ __forceinline__ void load_q1(bf16x8 (&qf)[3][Geo<D>::NKK], const unsigned short* qrow, int h32) {
#pragma unroll
    for (int kk = 0; kk < Geo<D>::NKK; ++kk) {
        const int d0 = 16 * kk + 8 * h32;
        qf[0][kk] = d0 < D ? *reinterpret_cast<const bf16x8*>(qrow + d0) : __builtin_bit_cast(bf16x8, make_uint4(0u, 0u, 0u, 0u));
    }
}

// the query's −zmul(t) = zh + zm + zl in depth slots D, D+1, D+2 of its (single) code plane (Geo::FOLDZ)
template <int D>
__device__ __forceinline__ void fold_zmul(bf16x8 (&qf)[3][Geo<D>::NKK], float zmul, int h32) {
    unsigned short zh, zm, zl;
    split3(-zmul, zh, zm, zl);
    if (h32 == (D / 8) % 2)
        qf[0][D / 16] = __builtin_bit_cast(bf16x8, make_uint4((unsigned)zh | ((unsigned)zm << 16), (unsigned)zl, 0u, 0u));
}

template <int D>
__device__ __forceinline__ v16f score_tile_q1(const unsigned short* kb, const bf16x8 (&qf)[3][Geo<D>::NKK], float zmul, int lane) {
    using G = Geo<D, 2, false>;
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
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl, qf[0][kk], acc, 0, 0, 0);     // smallest terms first
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(km, qf[0][kk], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qf[0][kk], acc, 0, 0, 0);
    }
    if constexpr (!G::FOLDZ) {
        const float* tv = reinterpret_cast<const float*>(kb + 3 * PL);
        const int h32 = lane >> 5;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 t4 = *reinterpret_cast<const float4*>(tv + 8 * g + 4 * h32);
            acc[4 * g + 0] = fmaf(-zmul, t4.x, acc[4 * g + 0]);
            acc[4 * g + 1] = fmaf(-zmul, t4.y, acc[4 * g + 1]);
            acc[4 * g + 2] = fmaf(-zmul, t4.z, acc[4 * g + 2]);
            acc[4 * g + 3] = fmaf(-zmul, t4.w, acc[4 * g + 3]);
        }
    }
    return acc;
}

// Workgroup -> (query tile, batch·head).  Workgroups are dealt round-robin over the 8 XCDs, each with its own L2: in launch
// order the query tiles of one (batch, head) — which all stream the same K / V tile images — sit on 8 different XCDs and
// every L2 fetches every image.  Remapped (bijectively, any grid) so that XCD k owns a contiguous range of (batch·head)
// major tiles, like the GEMM's tile order.
__device__ __forceinline__ void attn_block_coords(int xcd_remap, int& bx, int& bh) {
    bx = blockIdx.x; bh = blockIdx.y;
    if (xcd_remap) {
        const int gx = gridDim.x, T = gridDim.x * gridDim.y;
        const int bid = blockIdx.x + gx * blockIdx.y;
        const int q = T >> 3, r = T & 7, xcd = bid & 7;
        const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
        bh = logical / gx;
        bx = logical - bh * gx;
    }
}

// ---- element loaders of the pre-pass code below: TypedLd<T> for a kernel templated on the tensor dtype (attn3_prep_kernel), RtLd for
// one that takes the dtype at run time (a wave-uniform branch per load instead of 3x the code)
template <typename TIn> struct TypedLd {
    __device__ __forceinline__ void load8(const void* base, int64_t i, float (&x)[8]) const { ::load8<TIn>(reinterpret_cast<const TIn*>(base) + i, x); }
    __device__ __forceinline__ float load1(const void* base, int64_t i) const { return dgq_to_float(reinterpret_cast<const TIn*>(base)[i]); }
};
struct RtLd {
    int dtype;
    __device__ __forceinline__ void load8(const void* base, int64_t i, float (&x)[8]) const {
        if (dtype == DGQ_F16) ::load8<__half>(reinterpret_cast<const __half*>(base) + i, x);
        else if (dtype == DGQ_BF16) ::load8<__hip_bfloat16>(reinterpret_cast<const __hip_bfloat16*>(base) + i, x);
        else ::load8<float>(reinterpret_cast<const float*>(base) + i, x);
    }
    __device__ __forceinline__ float load1(const void* base, int64_t i) const { return load_any(base, dtype, i); }
};

// The pre-pass's work on ONE 32-key tile of one (batch, head), by the 256 threads of a workgroup (tid): the K image (prep_k_tile) and
// the V image (prep_v_tile) laid out as Geo<D, QM, VINT> says — into global memory (attn3_prep_kernel: the tile images the main kernels
// stage by LDS-DMA) or, through a generic pointer, straight into LDS.  kp / vp: the k / v tensors; kbase / vbase: element index
// of key 0 of this (batch, head); HD = H·D (row stride of a key); s0 = first key of the tile.
template <int D, int QM, bool VINT, typename LD>
__device__ __forceinline__ void prep_k_tile(const LD& ld, const void* kp, int64_t kbase, int64_t HD, int S, int s0, const FqDesc& fk,
                                            const FqDesc& fqq, unsigned short* kimg, int tid) {
    using G = Geo<D, QM, VINT>;
    constexpr bool QI8 = QM == 1;
    if (QI8) {
        // int8 K codes + per-key table; 8 threads per key row like the Q rows above
        int8_t* k8 = reinterpret_cast<int8_t*>(kimg);
        float* ktab = reinterpret_cast<float*>(k8 + G::K8_BYTES);
        const int r = tid >> 3, part = tid & 7;
        const int sidx = s0 + r;
        const bool quant = sidx < S && sidx >= fk.skip;          // key 0 under start-peak: zero row, scale 0 (rank-1 path)
        float dl = 0.0f, z = 0.0f, inv = 0.0f;
        if (quant) {
            const int idx = fk.mode == 0 ? 0 : sidx - fk.skip;
            dl = fk.delta[idx]; z = fk.zp[idx]; inv = dgq_rcp(dl);
        }
        const float off = 0.5f * (fk.qmax + 1.0f);
        float csum = 0.0f;
        for (int c8 = part; c8 < G::K8_LD / 8; c8 += 8) {
            unsigned w0 = 0, w1 = 0;
            if (quant && 8 * c8 < D) {
                float x[8];
                ld.load8(kp, kbase + sidx * HD + 8 * c8, x);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float c = fq_code(x[j], dl, inv, z, fk.qmax, off);
                    csum += c;
                    const unsigned byte = ((unsigned)(int)c) & 0xFFu;
                    if (j < 4) w0 |= byte << (8 * j); else w1 |= byte << (8 * (j - 4));
                }
            }
            *reinterpret_cast<uint2*>(k8 + r * G::K8_LD + 8 * c8) = make_uint2(w0, w1);
        }
#pragma unroll
        for (int o = 4; o > 0; o >>= 1) csum += __shfl_xor(csum, o, 64);
        if (part == 0) {
            ktab[r] = dl;                                        // 0 for padding keys and the bypassed key
            ktab[KT + r] = quant ? -(z - off) : 0.0f;
            ktab[2 * KT + r] = -csum;
        }
    }
    constexpr int KC = G::KLD / 8;                       // 16-byte chunks per K row (padding chunks are zero)
    if (QM == 2) {
        // Q1K3: three planes of K̃[s][d] = aqtizer_k(k)[s][d]·(δq(d) for a per-head-dim aqtizer_q) and, per key,
        // Σ_d w(d)·K̃[s][d] with w = z'q(d) (per-head-dim) or 1; 8 threads per key row, deterministic shuffle reduction
        const int r = tid >> 3, part = tid & 7;
        const int sidx = s0 + r;
        const float offq = 0.5f * (fqq.qmax + 1.0f);
        float corr = 0.0f;
        constexpr int KIT = (KC + 7) / 8;
        float x[KIT][8];
#pragma unroll
        for (int it = 0; it < KIT; ++it)                     // all loads of the thread in flight together
            if (sidx < S && 8 * (part + 8 * it) < D) ld.load8(kp, kbase + sidx * HD + 8 * (part + 8 * it), x[it]);
#pragma unroll
        for (int it = 0; it < KIT; ++it) {
            const int c8 = part + 8 * it;
            if (c8 >= KC) continue;
            unsigned wh[4] = {0, 0, 0, 0}, wm[4] = {0, 0, 0, 0}, wl[4] = {0, 0, 0, 0};
            if (sidx < S && 8 * c8 < D) {
                fq_apply8(fk, x[it], sidx, 8 * c8);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (fqq.mode == 2) {
                        x[it][j] *= fqq.delta[8 * c8 + j];
                        corr += (fqq.zp[8 * c8 + j] - offq) * x[it][j];
                    } else {
                        corr += x[it][j];
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    unsigned short h0, m0, l0, h1, m1, l1;
                    split3(x[it][2 * j], h0, m0, l0);
                    split3(x[it][2 * j + 1], h1, m1, l1);
                    wh[j] = (unsigned)h0 | ((unsigned)h1 << 16);
                    wm[j] = (unsigned)m0 | ((unsigned)m1 << 16);
                    wl[j] = (unsigned)l0 | ((unsigned)l1 << 16);
                }
            }
            unsigned short* dst = kimg + r * G::KLD + 8 * c8;
            *reinterpret_cast<uint4*>(dst) = make_uint4(wh[0], wh[1], wh[2], wh[3]);
            *reinterpret_cast<uint4*>(dst + KT * G::KLD) = make_uint4(wm[0], wm[1], wm[2], wm[3]);
            *reinterpret_cast<uint4*>(dst + 2 * KT * G::KLD) = make_uint4(wl[0], wl[1], wl[2], wl[3]);
        }
#pragma unroll
        for (int o = 4; o > 0; o >>= 1) corr += __shfl_xor(corr, o, 64);
        if (part == 0) reinterpret_cast<float*>(kimg + 3 * KT * G::KLD)[r] = corr;
        if (G::FOLDZ && part == (D / 8) % 8) {               // (this thread wrote the zero chunk D/8 above: same-thread order)
            unsigned short th, tm, tl;
            split3(corr, th, tm, tl);
            unsigned short* dst = kimg + r * G::KLD + D;     // slots D, D+1, D+2 all hold tv[s] = th + tm + tl
            *reinterpret_cast<uint2*>(dst) = make_uint2((unsigned)th | ((unsigned)th << 16), (unsigned)th);
            *reinterpret_cast<uint2*>(dst + KT * G::KLD) = make_uint2((unsigned)tm | ((unsigned)tm << 16), (unsigned)tm);
            *reinterpret_cast<uint2*>(dst + 2 * KT * G::KLD) = make_uint2((unsigned)tl | ((unsigned)tl << 16), (unsigned)tl);
        }
    }
    for (int i = tid; QM == 0 && i < KT * KC; i += 256) {
        const int r = i / KC, c8 = i - r * KC;
        const int sidx = s0 + r;
        unsigned wh[4], wm[4], wl[4];
        const bool live = sidx < S && 8 * c8 < D;
        float x[8];
        if (live) {
            ld.load8(kp, kbase + sidx * HD + 8 * c8, x);
            fq_apply8(fk, x, sidx, 8 * c8);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            unsigned short h0 = 0, m0 = 0, l0 = 0, h1 = 0, m1 = 0, l1 = 0;
            if (live) {
                split3(x[2 * j], h0, m0, l0);
                split3(x[2 * j + 1], h1, m1, l1);
            }
            wh[j] = (unsigned)h0 | ((unsigned)h1 << 16);
            wm[j] = (unsigned)m0 | ((unsigned)m1 << 16);
            wl[j] = (unsigned)l0 | ((unsigned)l1 << 16);
        }
        unsigned short* dst = kimg + r * G::KLD + 8 * c8;
        *reinterpret_cast<uint4*>(dst) = make_uint4(wh[0], wh[1], wh[2], wh[3]);
        *reinterpret_cast<uint4*>(dst + KT * G::KLD) = make_uint4(wm[0], wm[1], wm[2], wm[3]);
        *reinterpret_cast<uint4*>(dst + 2 * KT * G::KLD) = make_uint4(wl[0], wl[1], wl[2], wl[3]);
    }
}

template <int D, int QM, bool VINT, typename LD>
__device__ __forceinline__ void prep_v_tile(const LD& ld, const void* vp, int64_t vbase, int64_t HD, int S, int s0, const FqDesc& fv,
                                            unsigned short* vimg, int tid) {
    using G = Geo<D, QM, VINT>;
    constexpr int VC = G::VLD / 8;                       // 4 chunks of 8 key slots + 1 padding chunk per V^T row
    constexpr int VIT = (G::DV * VC + 255) / 256;
    {
        float xv[VIT][8];
#pragma unroll
        for (int it = 0; it < VIT; ++it) {                   // all loads of the thread in flight together
            const int i = tid + 256 * it;
            const int c8 = i / G::DV, d = i - c8 * G::DV;    // lanes run over d: coalesced reads of every key row
#pragma unroll
            for (int s8 = 0; s8 < 8; ++s8) {                 // slot within the chunk = 4a + b
                const int sidx = s0 + 16 * (c8 >> 1) + 8 * (s8 >> 2) + 4 * (c8 & 1) + (s8 & 3);
                xv[it][s8] = (c8 < 4 && sidx < S && d < D) ? ld.load1(vp, vbase + sidx * HD + d) : 0.0f;
            }
        }
#pragma unroll
        for (int it = 0; it < VIT; ++it) {
            const int i = tid + 256 * it;
            if (i >= G::DV * VC) continue;
            const int c8 = i / G::DV, d = i - c8 * G::DV;
            unsigned wh[4], wm[4], wl[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                unsigned short hh[2] = {0, 0}, mm[2] = {0, 0}, ll[2] = {0, 0};
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int slot8 = 2 * j + e;
                    const int sidx = s0 + 16 * (c8 >> 1) + 8 * (slot8 >> 2) + 4 * (c8 & 1) + (slot8 & 3);
                    if (G::VONES && c8 < 4 && d == D) hh[e] = 0x3F80;          // the ones row (every slot: padding keys carry p̂ = 0)
                    if (c8 < 4 && sidx < S && d < D) {
                        const float x1 = xv[it][slot8];
                        if (VINT) {                                            // centred code (|c'| <= 128: exact in bf16)
                            const int idx = fv.mode == 0 ? 0 : d;
                            const float dl = fv.delta[idx];
                            hh[e] = bf16_bits(fq_code(x1, dl, dgq_rcp(dl), fv.zp[idx], fv.qmax, 0.5f * (fv.qmax + 1.0f)));
                        } else {
                            split3(fq_apply(fv, x1, sidx, d), hh[e], mm[e], ll[e]);
                        }
                    }
                }
                wh[j] = (unsigned)hh[0] | ((unsigned)hh[1] << 16);
                wm[j] = (unsigned)mm[0] | ((unsigned)mm[1] << 16);
                wl[j] = (unsigned)ll[0] | ((unsigned)ll[1] << 16);
            }
            unsigned short* dst = vimg + d * G::VLD + 8 * c8;
            *reinterpret_cast<uint4*>(dst) = make_uint4(wh[0], wh[1], wh[2], wh[3]);
            if (!VINT) {
                *reinterpret_cast<uint4*>(dst + G::DV * G::VLD) = make_uint4(wm[0], wm[1], wm[2], wm[3]);
                *reinterpret_cast<uint4*>(dst + 2 * G::DV * G::VLD) = make_uint4(wl[0], wl[1], wl[2], wl[3]);
            }
        }
    }
}

// attn_one.hip: statistics + δ exchange + P·V of one call in ONE launch (key ranges of <= 8 tiles); returns 1 when it does not take the
// call (then the three launches of attn_bf16x3.hip run), 0 / a negative DGQ_E* code otherwise.  sync: the words behind the δ slots (mode 1).
int dgq_attention_one_launch(const AttnParams& p, int D, int qm, bool vint, unsigned* sync, hipStream_t st);
