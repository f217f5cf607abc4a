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
#include <type_traits>
#include "dgq_common.h"

typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define QROWS 128
#define KT 32
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
    return dl * (dgq_affine_code(x, dl, z, f.qmax) - z);
}

struct AttnParams {
    FqDesc fq[3];          // q, k, v
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
    const unsigned short* kb;   // [B*H][3][Spad][DP] bf16 split planes of K
    const unsigned short* vt;   // [B*H][3][DV][Spad] bf16 split planes of V^T, keys permuted within each 32-key tile
    int Spad;
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

// One pre-pass per attention call: split K and V exactly into three bf16 planes, laid out so that a 32-key tile is a
// set of whole 16-byte chunks for the main kernels (K: rows of DP elements; V: transposed, key order inside a tile =
// the k order of an accumulator tile used as the next MFMA's B operand: key 16s + 8a + 4h + b -> slot 16s + 8h + 4a + b).
// Every Q block of a (batch, head) re-reads these planes; splitting them inside the main loop cost more than the MFMAs.
template <int D>
__global__ __launch_bounds__(256) void attn3_prep_kernel(const float* __restrict__ k, const float* __restrict__ v,
                                                         unsigned short* __restrict__ kb, unsigned short* __restrict__ vt,
                                                         int B, int H, int S, int Spad, FqDesc fk, FqDesc fv,
                                                         float* __restrict__ delta_reset) {
    using G = Geo<D>;
    constexpr int DV = G::NDT * 32;
    const int bh = blockIdx.y, b = bh / H, hd = bh - b * H;
    const int s0 = blockIdx.x * KT;
    if (delta_reset && blockIdx.x == 0 && bh == 0 && threadIdx.x == 0) *delta_reset = 0.0f;   // real-time δ: max starts at 0
    const float* kbase = k + ((int64_t)(b * S) * H + hd) * D;
    const float* vbase = v + ((int64_t)(b * S) * H + hd) * D;
    const int64_t HD = (int64_t)H * D;
    unsigned short* kdst = kb + (int64_t)bh * 3 * Spad * G::DP;
    unsigned short* vdst = vt + (int64_t)bh * 3 * DV * Spad;
    for (int i = threadIdx.x; i < KT * G::DP; i += 256) {
        const int r = i / G::DP, c = i - r * G::DP;
        const int sidx = s0 + r;
        const float x = (sidx < S && c < D) ? fq_apply(fk, kbase[sidx * HD + c], sidx, c) : 0.0f;
        unsigned short h, m, l;
        split3(x, h, m, l);
        const int64_t o = (int64_t)sidx * G::DP + c;
        kdst[o] = h;
        kdst[(int64_t)Spad * G::DP + o] = m;
        kdst[2 * (int64_t)Spad * G::DP + o] = l;
    }
    for (int i = threadIdx.x; i < KT * DV; i += 256) {
        const int key = i / DV, d = i - key * DV;
        const int sidx = s0 + key;
        const float x = (sidx < S && d < D) ? fq_apply(fv, vbase[sidx * HD + d], sidx, d) : 0.0f;
        unsigned short h, m, l;
        split3(x, h, m, l);
        const int sg = key >> 4, a = (key >> 3) & 1, hh = (key >> 2) & 1, bb = key & 3;
        const int slot = 16 * sg + 8 * hh + 4 * a + bb;
        const int64_t o = (int64_t)d * Spad + s0 + slot;
        vdst[o] = h;
        vdst[(int64_t)DV * Spad + o] = m;
        vdst[2 * (int64_t)DV * Spad + o] = l;
    }
}

// 16-byte chunk copies of one tile: global planes -> registers -> LDS (padded rows).  Written as macros over local
// arrays with compile-time trip counts: as struct members the register arrays were demoted to scratch.
template <int D> struct StageGeo {
    using G = Geo<D>;
    static constexpr int CPR = G::DP / 8;                       // 16-byte chunks per K row
    static constexpr int KCHUNKS = 3 * KT * CPR;
    static constexpr int KPER = (KCHUNKS + 255) / 256;
    static constexpr int DV = G::NDT * 32;
    static constexpr int VCHUNKS = 3 * DV * 4;                  // 4 chunks (32 keys) per V^T row
    static constexpr int VPER = (VCHUNKS + 255) / 256;
};

#define K_LOAD(kr, kb_bh, s0_)                                                                                        \
    _Pragma("unroll") for (int i_ = 0; i_ < SG::KPER; ++i_) {                                                          \
        const int id_ = min(tid + 256 * i_, SG::KCHUNKS - 1);      /* clamped: every register is always written */ \
        const int pl_ = id_ / (KT * SG::CPR), rem_ = id_ - pl_ * (KT * SG::CPR), row_ = rem_ / SG::CPR,                \
                  c_ = rem_ - row_ * SG::CPR;                                                                          \
        kr[i_] = *reinterpret_cast<const uint4*>((kb_bh) + ((int64_t)pl_ * p.Spad + (s0_) + row_) * G::DP + 8 * c_);   \
    }
#define K_STORE(kr, lds_)                                                                                             \
    _Pragma("unroll") for (int i_ = 0; i_ < SG::KPER; ++i_) {                                                          \
        const int id_ = tid + 256 * i_;                                                                                \
        if (id_ < SG::KCHUNKS) {                                                                                       \
            const int pl_ = id_ / (KT * SG::CPR), rem_ = id_ - pl_ * (KT * SG::CPR), row_ = rem_ / SG::CPR,            \
                      c_ = rem_ - row_ * SG::CPR;                                                                      \
            *reinterpret_cast<uint4*>((lds_) + (pl_ * KT + row_) * G::KLD + 8 * c_) = kr[i_];                          \
        }                                                                                                              \
    }
#define V_LOAD(vr, vt_bh, s0_)                                                                                        \
    _Pragma("unroll") for (int i_ = 0; i_ < SG::VPER; ++i_) {                                                          \
        const int id_ = min(tid + 256 * i_, SG::VCHUNKS - 1);                                                          \
        vr[i_] = *reinterpret_cast<const uint4*>((vt_bh) + (int64_t)(id_ >> 2) * p.Spad + (s0_) + 8 * (id_ & 3));      \
    }
#define V_STORE(vr, lds_)                                                                                             \
    _Pragma("unroll") for (int i_ = 0; i_ < SG::VPER; ++i_) {                                                          \
        const int id_ = tid + 256 * i_;                                                                                \
        if (id_ < SG::VCHUNKS) *reinterpret_cast<uint4*>((lds_) + (id_ >> 2) * G::VLD + 8 * (id_ & 3)) = vr[i_];       \
    }

// Q rows of this lane as B-operand fragments: qf[split][kk] holds Q[t][16kk + 8h + j], j = 0..7
template <int D>
__device__ __forceinline__ void load_q(bf16x8 (&qf)[3][Geo<D>::NKK], const float* qrow, int h32, const FqDesc& fq, int t) {
    using G = Geo<D>;
#pragma unroll
    for (int kk = 0; kk < G::NKK; ++kk) {
        unsigned wh[4], wm[4], wl[4];                       // packed pairs (no sub-dword arrays: those go to scratch)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int d = 16 * kk + 8 * h32 + 2 * j;
            const float x0 = (d < D) ? fq_apply(fq, qrow[d], t, d) : 0.0f;
            const float x1 = (d + 1 < D) ? fq_apply(fq, qrow[d + 1], t, d + 1) : 0.0f;
            unsigned short h0, m0, l0, h1, m1, l1;
            split3(x0, h0, m0, l0);
            split3(x1, h1, m1, l1);
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
    load_q<D>(qf, p.q + ((int64_t)(b * p.T + tq) * p.H + hd) * D, h32, p.fq[0], tq);
    const float sl2 = p.scale * LOG2E;                   // scores in log2 units: p = 2^(s2 − m)/l
    float m = -INFINITY, l = 0.0f, m2 = -INFINITY;
    const unsigned short* kb_bh = p.kb + (int64_t)bh * 3 * p.Spad * G::DP;
    using SG = StageGeo<D>;
    uint4 kr[SG::KPER];
    K_LOAD(kr, kb_bh, 0)
    K_STORE(kr, kb)
    __syncthreads();
    int cur = 0;
    for (int s0 = 0; s0 < p.S; s0 += KT) {
        const bool more = s0 + KT < p.S;
        const int s_next = more ? s0 + KT : s0;                   // unconditional (a conditional load demotes kr to scratch)
        K_LOAD(kr, kb_bh, s_next)                                  // in flight during the MFMAs of this tile
        const unsigned short* kbc = kb + cur * G::K_ELEMS;
        v16f acc = score_tile<D>(kbc, qf, lane);
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
        if (more) { K_STORE(kr, kb + (cur ^ 1) * G::K_ELEMS) }    // the other buffer was last read one barrier ago
        __syncthreads();
        cur ^= 1;
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

template <int D, bool UNIFORM>
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
    load_q<D>(qf, p.q + ((int64_t)(b * p.T + tq) * p.H + hd) * D, h32, p.fq[0], tq);
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

    const unsigned short* kb_bh = p.kb + (int64_t)bh * 3 * p.Spad * G::DP;
    const unsigned short* vt_bh = p.vt + (int64_t)bh * 3 * (G::NDT * 32) * p.Spad;
    using SG = StageGeo<D>;
    uint4 kr[SG::KPER];
    uint4 vr[SG::VPER];
    K_LOAD(kr, kb_bh, 0)
    V_LOAD(vr, vt_bh, 0)
    K_STORE(kr, kb)
    V_STORE(vr, vt)
    __syncthreads();
    int cur = 0;
    constexpr int BUF = G::K_ELEMS + G::V_ELEMS;
    for (int s0 = 0; s0 < p.S; s0 += KT) {
        const bool more = s0 + KT < p.S;
        const int s_next = more ? s0 + KT : s0;                   // unconditional (a conditional load demotes kr/vr to scratch)
        K_LOAD(kr, kb_bh, s_next)
        V_LOAD(vr, vt_bh, s_next)
        const unsigned short* kbc = kb + cur * BUF;
        const unsigned short* vtc = vt + cur * BUF;
        v16f acc = score_tile<D>(kbc, qf, lane);
        // interior tiles carry no per-key conditions; only the first tile (bypassed column) and a partial last tile do
        const bool edge = (s0 + KT > p.S) || (s0 < p.skip);      // block-uniform
        auto quantise = [&](auto edge_tag) {
            constexpr bool EDGE = decltype(edge_tag)::value;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float ph;
                if (UNIFORM) {
                    const float pr = exp2f(fmaf(acc[r], -nsl2, -m)) * inv_l;     // always_zero: code = clamp(rne(p/δ), 0, 2^b−1)
                    ph = fminf(fmaxf(rintf(__fdiv_rn(pr, delta)), 0.0f), p.qmax);
                } else {
                    float code = rintf(fmaf(acc[r], nsl2, a0));
                    code = fminf(fmaxf(code, 0.0f), p.qmax);
                    // 2^-code as fp32 bits; codes > 126 (p̂ < 2^-126·δ) are below anything the fp32 sum can resolve
                    const int e = 127 - (int)code;
                    ph = e > 0 ? __int_as_float(e << 23) : 0.0f;
                }
                if (EDGE) {
                    const int s = s0 + key_of(r, h32);
                    if (s >= p.S) ph = 0.0f;
                    else if (s < p.skip) {
                        p_bypass = exp2f(fmaf(acc[r], -nsl2, -m)) * inv_l;
                        ph = 0.0f;
                    }
                }
                acc[r] = ph;
            }
        };
        if (edge) quantise(std::true_type{});
        else quantise(std::false_type{});
        // B fragments of the two 16-key steps: exact bf16 = upper halves of the fp32 words
        bf16x8 pf[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#define PK(i) ((__float_as_uint(acc[8 * ks + 2 * (i)]) >> 16) | (__float_as_uint(acc[8 * ks + 2 * (i) + 1]) & 0xFFFF0000u))
            pf[ks] = __builtin_bit_cast(bf16x8, make_uint4(PK(0), PK(1), PK(2), PK(3)));
#undef PK
        }
#pragma unroll
        for (int j = 0; j < G::NDT; ++j) {
            const unsigned short* vp = vtc + (j * 32 + (lane & 31)) * G::VLD + 8 * h32;
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
        if (more) {
            K_STORE(kr, kb + (cur ^ 1) * BUF)
            V_STORE(vr, vt + (cur ^ 1) * BUF)
        }
        __syncthreads();
        cur ^= 1;
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
                    if (p.skip > 0) o += p_bypass * fq_apply(p.fq[2], v0[d], 0, d);
                    op[d] = o;
                }
            }
    }
}

template <int D>
static int launch_attn3(AttnParams p, unsigned short* kb, unsigned short* vt, hipStream_t st) {
    using G = Geo<D>;
    p.kb = kb;
    p.vt = vt;
    hipLaunchKernelGGL((attn3_prep_kernel<D>), dim3(p.Spad / KT, p.B * p.H), dim3(256), 0, st, p.k, p.v, kb, vt, p.B, p.H,
                       p.S, p.Spad, p.fq[1], p.fq[2], p.mode == 1 ? p.delta : nullptr);
    dim3 grid((p.T + QROWS - 1) / QROWS, p.B * p.H), block(256);
    static const bool lds_ok = [] {                            // up to 141 KB of dynamic LDS (D = 160): opt in once
        const int bytes = 2 * (G::K_ELEMS + G::V_ELEMS) * 2;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn3_stats_kernel<D>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn3_pv_kernel<D, true>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn3_pv_kernel<D, false>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        return true;
    }();
    (void)lds_ok;
    hipLaunchKernelGGL((attn3_stats_kernel<D>), grid, block, 2 * G::K_ELEMS * 2, st, p);
    if (p.mode == 3) hipLaunchKernelGGL((attn3_pv_kernel<D, true>), grid, block, 2 * (G::K_ELEMS + G::V_ELEMS) * 2, st, p);
    else hipLaunchKernelGGL((attn3_pv_kernel<D, false>), grid, block, 2 * (G::K_ELEMS + G::V_ELEMS) * 2, st, p);
    return dgq_launch_status("dgq_attention_f32(bf16x3)");
}

// bytes of K/V split planes for one call (0 when D is not instantiated here)
size_t dgq_attention_bf16x3_bytes(int B, int H, int S, int D) {
    const size_t Spad = (size_t)(S + KT - 1) / KT * KT;
    size_t dp, dv;
    switch (D) {
        case 8: dp = Geo<8>::DP; dv = Geo<8>::NDT * 32; break;
        case 16: dp = Geo<16>::DP; dv = Geo<16>::NDT * 32; break;
        case 40: dp = Geo<40>::DP; dv = Geo<40>::NDT * 32; break;
        case 64: dp = Geo<64>::DP; dv = Geo<64>::NDT * 32; break;
        case 80: dp = Geo<80>::DP; dv = Geo<80>::NDT * 32; break;
        case 160: dp = Geo<160>::DP; dv = Geo<160>::NDT * 32; break;
        default: return 0;
    }
    return (size_t)B * H * 3 * Spad * (dp + dv) * sizeof(unsigned short);
}

// called from dgq_attention_f32 (attn_fused.hip) for the quantised modes; returns 1 when D is not instantiated here
int dgq_attention_bf16x3(const float* q, const float* k, const float* v, float* o, int B, int H, int T, int S, int D,
                         float scale, int mode, int skip, float qmax, float* stats_ws, float* delta_ws, void* planes,
                         const dgq_attn_fq_t* fq, hipStream_t st) {
    AttnParams p;
    for (int i = 0; i < 3; ++i) {
        p.fq[i].mode = -1; p.fq[i].skip = 0; p.fq[i].qmax = 0.0f; p.fq[i].delta = nullptr; p.fq[i].zp = nullptr;
        if (fq && fq[i].mode >= 0) {
            p.fq[i].mode = fq[i].mode; p.fq[i].skip = fq[i].skip; p.fq[i].qmax = (float)((1 << fq[i].bits) - 1);
            p.fq[i].delta = fq[i].delta; p.fq[i].zp = fq[i].zero_point;
        }
    }
    p.q = q; p.k = k; p.v = v; p.o = o; p.B = B; p.H = H; p.T = T; p.S = S; p.scale = scale; p.mode = mode; p.skip = skip;
    p.qmax = qmax; p.stats = stats_ws; p.delta = delta_ws;
    p.Spad = (S + KT - 1) / KT * KT;
    unsigned short* kb = reinterpret_cast<unsigned short*>(planes);
    switch (D) {
        case 8: return launch_attn3<8>(p, kb, kb + (size_t)B * H * 3 * p.Spad * Geo<8>::DP, st);
        case 16: return launch_attn3<16>(p, kb, kb + (size_t)B * H * 3 * p.Spad * Geo<16>::DP, st);
        case 40: return launch_attn3<40>(p, kb, kb + (size_t)B * H * 3 * p.Spad * Geo<40>::DP, st);
        case 64: return launch_attn3<64>(p, kb, kb + (size_t)B * H * 3 * p.Spad * Geo<64>::DP, st);
        case 80: return launch_attn3<80>(p, kb, kb + (size_t)B * H * 3 * p.Spad * Geo<80>::DP, st);
        case 160: return launch_attn3<160>(p, kb, kb + (size_t)B * H * 3 * p.Spad * Geo<160>::DP, st);
        default: return 1;
    }
}
