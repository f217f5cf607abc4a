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
//
// QI8 variant ("int8 attention"): when aqtizer_q and aqtizer_k are scalar or per-token quantizers the
// operands of Q·K^T are integer codes times one scale per token, q̂ = δq(t)(cq − zq(t)), k̂ = δk(s)(ck − zk(s)), so
//     Σ_d q̂k̂ = δq δk [ Σ_d c'q c'k  −  z'q Σ_d c'k  −  z'k (Σ_d c'q − D z'q) ]        (c' = c − 2^(b−1): centred int8)
// is ONE exact int8 contraction on V_MFMA_I32_32X32X32_I8 (2x the bf16 rate, one product instead of six) plus a rank-1
// correction per key and per query, applied in fp32 to the int32 accumulator (3 FMA/mul per score).  The K tile image
// then holds int8 codes + a (δk, −z'k, −Σc'k) table per key, the pre-pass writes int8 Q codes + (δq, z'q, Σc'q − D z'q,
// start-peak score) per query; δq is folded into the per-lane log2 scale.  Per-head-dim quantizers put a scale inside the
// sum and stay on the bf16x3 products.  P̂·V is unchanged.
// Work decomposition as attn_fused.hip: block = 4 waves = 128 query rows of one (batch, head), 32-key tiles, the
// score tile is computed transposed (S^T = K·Q^T) so softmax statistics are in-register and the S^T accumulator is
// directly the B operand of O^T = V^T·P̂^T (k order inside a 16-key step: element j of lane half h is key
// 16s + 8(j>>2) + 4h + (j&3) — the V tile is stored transposed in exactly that key order).
#include "attn_bf16x3_dev.h"

DGQ_DIAG_BUFFER(attn)


template <int D, typename TIn, int QM, bool VINT>
__global__ __launch_bounds__(256) void attn3_prep_kernel(const TIn* __restrict__ k, const TIn* __restrict__ v,
                                                         unsigned char* __restrict__ planes, int B, int H, int S, int NT,
                                                         FqDesc fk, FqDesc fv, float* __restrict__ delta_reset, int n_reset,
                                                         const TIn* __restrict__ q, float* __restrict__ qfq, int T, FqDesc fqq) {
    using G = Geo<D, QM, VINT>;
    constexpr bool QI8 = QM == 1;
    static_assert(D % 8 == 0, "head_dim must be a multiple of 8");
    const int bh = blockIdx.y, b = bh / H, hd = bh - b * H;
    DGQ_DIAG_DECL
    DGQ_STAMP(0); DGQ_STAMP_REAL(1); DGQ_STAMP_WHERE(2);
#ifdef DGQ_DIAG
    dg.t[13] = 1;                                           // kernel tag: 1 pre-pass, 2 statistics, 3 P·V
#define DGQ_ATTN_PREP_DONE() do { DGQ_STAMP(9); DGQ_DIAG_DRAIN(); DGQ_STAMP(10); DGQ_STAMP_REAL(11); DGQ_DIAG_FLUSH(attn, 4, (threadIdx.x >> 6) + 32768, threadIdx.x & 63); } while (0)
#else
#define DGQ_ATTN_PREP_DONE() do {} while (0)
#endif
    if (QM == 2 && (int)blockIdx.x >= 2 * NT) {
        // Q1K3: centred codes c'q = c − 2^(b−1) of aqtizer_q(q) as fp32 (exact), + (q scale, zero-point multiplier) per query
        const int t0 = ((int)blockIdx.x - 2 * NT) * 32;
        constexpr int QC = D / 8;
        unsigned short* qb = reinterpret_cast<unsigned short*>(qfq);          // [B][T][H][D] bf16 codes (|c'| <= 128: exact)
        float* qtab = reinterpret_cast<float*>(qb + (size_t)B * T * H * D);
        const float off = 0.5f * (fqq.qmax + 1.0f);
        constexpr int QIT = (32 * QC + 255) / 256;
        float x[QIT][8];
#pragma unroll
        for (int it = 0; it < QIT; ++it) {                                    // all loads of the thread in flight together
            const int i = threadIdx.x + 256 * it, r = i / QC, c8 = i - r * QC;
            if (i < 32 * QC && t0 + r < T) load8<TIn>(q + ((int64_t)(b * T + t0 + r) * H + hd) * D + 8 * c8, x[it]);
        }
#pragma unroll
        for (int it = 0; it < QIT; ++it) {
            const int i = threadIdx.x + 256 * it, r = i / QC, c8 = i - r * QC;
            const int t = t0 + r;
            if (i >= 32 * QC || t >= T) continue;
            const int64_t row = (int64_t)(b * T + t) * H + hd;
            unsigned w[4];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int idx = fqq.mode == 0 ? 0 : (fqq.mode == 1 ? t - fqq.skip : 8 * c8 + j);
                const float dl = fqq.delta[idx];
                const unsigned bits = bf16_bits(fq_code(x[it][j], dl, dgq_rcp(dl), fqq.zp[idx], fqq.qmax, off));
                if (j & 1) w[j >> 1] |= bits << 16; else w[j >> 1] = bits;
            }
            *reinterpret_cast<uint4*>(qb + row * D + 8 * c8) = make_uint4(w[0], w[1], w[2], w[3]);
            if (c8 == 0) {
                // per-d table: δq(d) lives in the K planes and z'q(d) in the per-key correction (multiplier 1);
                // scalar / per-token: scale δq(t) outside, correction z'q(t)·Σ_d K[s][d]
                const int idx = fqq.mode == 0 ? 0 : t - fqq.skip;
                qtab[row * 2] = fqq.mode == 2 ? 1.0f : fqq.delta[idx];
                qtab[row * 2 + 1] = fqq.mode == 2 ? 1.0f : fqq.zp[idx] - off;
            }
        }
        DGQ_ATTN_PREP_DONE(); return;
    }
    if (QI8 && (int)blockIdx.x >= 2 * NT) {
        // QI8: int8 codes of aqtizer_q(q) for 32 query rows + (δq, z'q, Σc'q − D·z'q, start-peak score/δq) per query.
        // One thread per query row (D <= 160 elements); the row is read as 8-element vectors.
        const int t = ((int)blockIdx.x - 2 * NT) * 32 + (threadIdx.x >> 3);
        const int part = threadIdx.x & 7;                       // 8 threads share a row: chunks part, part+8, ...
        int8_t* qc = reinterpret_cast<int8_t*>(qfq);             // [B][T][H][DP32] codes, then the float table
        float* qtab = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(qfq) + (size_t)B * T * H * G::DP32);
        if (t < T) {
            const int64_t row = (int64_t)(b * T + t) * H + hd;
            const int idx = fqq.mode == 0 ? 0 : t - fqq.skip;     // per-token / scalar only (host-checked); q has no skip
            const float dl = fqq.delta[idx], z = fqq.zp[idx], inv = dgq_rcp(dl);
            const float off = 0.5f * (fqq.qmax + 1.0f);
            const float zc = z - off;
            const TIn* k0 = k + ((int64_t)(b * S) * H + hd) * D;   // raw key 0 (start-peak bypass)
            float csum = 0.0f, sp = 0.0f;
            for (int c8 = part; c8 < G::DP32 / 8; c8 += 8) {
                float x[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                unsigned w0 = 0, w1 = 0;
                if (8 * c8 < D) {
                    load8<TIn>(q + row * D + 8 * c8, x);
                    float kk[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    if (fk.skip > 0) load8<TIn>(k0 + 8 * c8, kk);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float c = fq_code(x[j], dl, inv, z, fqq.qmax, off);
                        csum += c;
                        sp += (c - zc) * kk[j];
                        const unsigned byte = ((unsigned)(int)c) & 0xFFu;
                        if (j < 4) w0 |= byte << (8 * j); else w1 |= byte << (8 * (j - 4));
                    }
                }
                *reinterpret_cast<uint2*>(qc + row * G::DP32 + 8 * c8) = make_uint2(w0, w1);
            }
#pragma unroll
            for (int o = 4; o > 0; o >>= 1) {
                csum += __shfl_xor(csum, o, 64);
                sp += __shfl_xor(sp, o, 64);
            }
            if (part == 0) *reinterpret_cast<float4*>(qtab + row * 4) = make_float4(dl, zc, csum - (float)D * zc, sp);
        }
        DGQ_ATTN_PREP_DONE(); return;
    }
    if ((int)blockIdx.x >= 2 * NT) {
        // extra blocks (only when aqtizer_q is fused): fake-quantised copy of 32 query rows of this (batch, head), so
        // that the Q-fragment loads of the two main kernels stay plain loads
        const int t0 = ((int)blockIdx.x - 2 * NT) * 32;
        constexpr int QC = D / 8;
        for (int i = threadIdx.x; i < 32 * QC; i += 256) {
            const int r = i / QC, c8 = i - r * QC;
            const int t = t0 + r;
            if (t >= T) continue;
            const int64_t o = ((int64_t)(b * T + t) * H + hd) * D + 8 * c8;
            float x[8];
            load8<TIn>(q + o, x);
            fq_apply8(fqq, x, t, 8 * c8);
            *reinterpret_cast<float4*>(qfq + o) = make_float4(x[0], x[1], x[2], x[3]);
            *reinterpret_cast<float4*>(qfq + o + 4) = make_float4(x[4], x[5], x[6], x[7]);
        }
        DGQ_ATTN_PREP_DONE(); return;
    }
    // K and V images of a key tile are written by DIFFERENT workgroups (x < NT: K, NT <= x < 2·NT: V): each is a chain of
    // a few dependent load rounds, and run back to back in one workgroup they set the duration of this launch (18-23 us
    // for any token count; the data are a few hundred KB)
    const bool do_k = (int)blockIdx.x < NT;
    const int tile = do_k ? (int)blockIdx.x : (int)blockIdx.x - NT;
    const int s0 = tile * KT;
    // real-time δ: the maxima start at 0 — and with them the granules of the single-launch form's exchange (attn_one.hip) behind the slots
    if (delta_reset && blockIdx.x == 0 && bh == 0)
        for (int i = threadIdx.x; i < n_reset; i += 256) delta_reset[i] = 0.0f;
    const int64_t kbase = ((int64_t)(b * S) * H + hd) * D, vbase = kbase;     // element index of key 0 of this (batch, head)
    const int64_t HD = (int64_t)H * D;
    unsigned short* kimg = reinterpret_cast<unsigned short*>(planes + ((int64_t)bh * NT + tile) * G::IMG_BYTES);
    unsigned short* vimg = kimg + G::K_ELEMS;
    const TypedLd<TIn> ld{};
    if (do_k) prep_k_tile<D, QM, VINT>(ld, k, kbase, HD, S, s0, fk, fqq, kimg, threadIdx.x);
    else prep_v_tile<D, QM, VINT>(ld, v, vbase, HD, S, s0, fv, vimg, threadIdx.x);
    DGQ_ATTN_PREP_DONE();
}

// NW waves of 32 query rows per block.  NW = 8 (256 rows, one block per CU, two waves per SIMD) where the grid still
// fills the chip (T >= 2048 at B*H = 16): the K/V tile images are then staged once per 256 rows instead of once per
// 128 — half the LDS-DMA pieces per wave per tile, the largest non-MFMA cost of the loop.
// TPS: key tiles per ring stage.  The barrier that retires a stage keeps the two waves of a SIMD (same workgroup) in lockstep — both in
// their MFMA phase, then both in their VALU phase; with TPS = 2 a wave runs two tiles between barriers and its sibling may be a tile apart,
// so one wave's exponentials overlap the other's products (the long self-attention loops: 128 tiles at T = S = 4096).
template <int D, int NW, int QM, int TPS = 1>
__global__ __launch_bounds__(64 * NW) void attn3_stats_kernel(AttnParams p) {
    using G = Geo<D, QM>;
    constexpr bool QI8 = G::QI8, KS = QM == 3;
    constexpr int ST = TPS > 1 ? 3 : G::STATS_STAGES, NP = G::K_PIECES, TB = NP * 1024, SB = TPS * TB;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds8[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, h32 = lane >> 5;
    DGQ_DIAG_DECL
    DGQ_STAMP(0); DGQ_STAMP_REAL(1); DGQ_STAMP_WHERE(2);
#ifdef DGQ_DIAG
    dg.t[13] = 2;
#endif
    int bx, bh;
    attn_block_coords(p.xcd, bx, bh);
    const int b = bh / p.H, hd = bh - b * p.H;
    const int t = bx * (32 * NW) + wid * 32 + (lane & 31);
    const int tq = min(t, p.T - 1);
    const unsigned char* img_lane = p.planes + (int64_t)bh * p.NT * p.img_bytes + lane * 16;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(lptr_t)lds8;
    // ring invariant at the top of iteration i: tiles i .. i+ST-2 are issued (indices clamped to the last tile, so
    // the in-flight count is the same in every iteration), tile i has landed
    // key-split launches: workgroup z walks the key tiles [i0, i1) of its half (under-filled grids, see launch_attn3)
    const int nsp = gridDim.z, i0 = (int)((long)p.NT * blockIdx.z / nsp), i1 = (int)((long)p.NT * (blockIdx.z + 1) / nsp);
#pragma unroll
    for (int i = 0; i < ST - 1; ++i)
#pragma unroll
        for (int j = 0; j < TPS; ++j)
            issue_image<NP, NW>(img_lane + (int64_t)min(i0 + TPS * i + j, i1 - 1) * p.img_bytes, lds_base + i * SB + j * TB, wid);
    bf16x8 qf[3][QI8 ? 1 : G::NKK];
    v4i qc[G::NK32];
    float4 qt = make_float4(1.0f, 0.0f, 0.0f, 0.0f);
    if constexpr (QI8) load_q_i8<D>(qc, qt, p, (int64_t)(b * p.T + tq) * p.H + hd, h32);
    else if constexpr (QM == 2) load_q1<D>(qf, reinterpret_cast<const unsigned short*>(p.q) + ((int64_t)(b * p.T + tq) * p.H + hd) * D, h32);
    else load_q<D>(qf, p.q + ((int64_t)(b * p.T + tq) * p.H + hd) * D, h32, p.fq[0], tq);
    if constexpr (QM == 2) {                              // (query scale, zero-point multiplier) written by the pre-pass
        const float2 t2 = *reinterpret_cast<const float2*>(p.qtab + ((int64_t)(b * p.T + tq) * p.H + hd) * 2);
        qt.x = t2.x;
        qt.y = t2.y;
        if constexpr (G::FOLDZ) fold_zmul<D>(qf, qt.y, h32);
    }
    // KS: δk joins the log2 scale, cq = −z'k·(Σ c'q − D z'q) is a per-query shift of every score (added to the maxima at the end)
    float inv_dk = 1.0f, cq = 0.0f, dk = 1.0f;
    if constexpr (KS) {
        dk = p.fq[1].delta[0];
        inv_dk = 1.0f / dk;
        cq = -(p.fq[1].zp[0] - 0.5f * (p.fq[1].qmax + 1.0f)) * qt.z;
    }
    const float sl2 = p.scale * LOG2E * qt.x * dk;       // scores in log2 units: p = 2^(s2 − m)/l  (QI8: δq folded in, > 0)
    float mraw = -INFINITY, l = 0.0f, m2raw = -INFINITY; // running maxima of the UNSCALED scores (scale > 0)
    DGQ_STAMP(3);
    wait_image<NP, TPS * (ST - 2), NW>(wid);
    __builtin_amdgcn_s_barrier();
    DGQ_STAMP(4);
    int stage = 0, istage = ST - 1;
    // (TPS > 1: the host launches this form only for key ranges of a whole number of stages)
    for (int ib = i0; ib < i1; ib += TPS) {
#pragma unroll
      for (int jt = 0; jt < TPS; ++jt)
        issue_image<NP, NW>(img_lane + (int64_t)min(ib + TPS * (ST - 1) + jt, i1 - 1) * p.img_bytes, lds_base + istage * SB + jt * TB, wid);
#pragma unroll
      for (int jt = 0; jt < TPS; ++jt) {
        const int i = ib + jt;
        const unsigned char* tile_lds = lds8 + stage * SB + jt * TB;
        const int s0 = i * KT;
        v16f acc;
        if constexpr (QI8) acc = score_tile_i8<D, KS>(tile_lds, qc, qt, lane, i == 0 && p.kskip > 0, inv_dk, cq);
        else if constexpr (QM == 2) acc = score_tile_q1<D>(reinterpret_cast<const unsigned short*>(tile_lds), qf, qt.y, lane);
        else acc = score_tile<D>(reinterpret_cast<const unsigned short*>(tile_lds), qf, lane);
        const bool edge = (s0 + KT > p.S) || (s0 < p.skip);      // block-uniform: only the first / a partial last tile
        float tmax = -INFINITY, tmax2 = -INFINITY;
        if (edge) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int s = s0 + key_of(r, h32);
                if (s >= p.S) acc[r] = -INFINITY;
                tmax = fmaxf(tmax, acc[r]);
                if (s >= p.skip) tmax2 = fmaxf(tmax2, acc[r]);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, acc[r]);
            tmax2 = tmax;
        }
        // the two half-waves of a query keep their own (max, sum) over their own keys — no cross-lane traffic inside the
        // loop; merged once behind it.  A half-wave that has seen no key yet (S < 8) holds max = −inf: offset 0 then.
        const float mn = fmaxf(mraw, tmax);
        const float nb = (mn == -INFINITY) ? 0.0f : -(mn * sl2);
        f2 part = {0.0f, 0.0f};
        const f2 sl2v = {sl2, sl2}, nbv = {nb, nb};
#pragma unroll
        for (int r = 0; r < 16; r += 2) {                    // args <= 0: no range fix-up needed
            const f2 a = pk_fma(f2{acc[r], acc[r + 1]}, sl2v, nbv);
            part = pk_add(part, f2{__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)});
        }
        l = l * __builtin_amdgcn_exp2f(fmaf(mraw, sl2, nb)) + (part.x + part.y);
        mraw = mn;
        m2raw = fmaxf(m2raw, tmax2);
      }
        wait_image<NP, TPS * (ST - 2), NW>(wid);              // the next stage (this wave's pieces) has landed
        __builtin_amdgcn_s_barrier();                     // ... everyone's; and everyone is done reading `stage`
        stage = (stage + 1 == ST) ? 0 : stage + 1;
        istage = (istage + 1 == ST) ? 0 : istage + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // no DMA may outlive the block's LDS allocation
    DGQ_STAMP(5);
    {
        const float mo = __shfl_xor(mraw, 32, 64), lo = __shfl_xor(l, 32, 64);
        const float mm = fmaxf(mraw, mo), nb = -(mm * sl2);                   // finite: every key range holds >= 8 keys
        l = l * __builtin_amdgcn_exp2f(fmaf(mraw, sl2, nb)) + lo * __builtin_amdgcn_exp2f(fmaf(mo, sl2, nb));
        mraw = mm;
        m2raw = fmaxf(m2raw, __shfl_xor(m2raw, 32, 64));
    }
    mraw += cq;
    m2raw += cq;
    const float m = mraw * sl2;
    if (nsp > 1) {                                       // partial statistics of this key half; attn3_merge_kernel finishes them
        if (t < p.T && h32 == 0)
            *reinterpret_cast<float4*>(p.stats_part + (((int64_t)blockIdx.z * p.B * p.H + bh) * p.T + t) * 4) = make_float4(m, l, m2raw * sl2, 0.0f);
        DGQ_STAMP(9); DGQ_DIAG_DRAIN(); DGQ_STAMP(10); DGQ_STAMP_REAL(11);
        DGQ_DIAG_FLUSH(attn, NW, wid + 16384, lane);
        return;
    }
    if (t < p.T && h32 == 0) {
        float* st = p.stats + ((int64_t)bh * p.T + t) * 2;
        st[0] = m;
        st[1] = l;
    }
    if (p.mode == 1) {
        float pm = (t < p.T) ? exp2f(m2raw * sl2 - m) / l : 0.0f;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) pm = fmaxf(pm, __shfl_xor(pm, o, 64));
        float* wmax = reinterpret_cast<float*>(lds8);        // (the ring is idle: every DMA has landed, every wave is past its last read)
        __syncthreads();
        if (lane == 0) wmax[wid] = pm;
        __syncthreads();
        if (tid == 0) {
#pragma unroll
            for (int w = 1; w < NW; ++w) pm = fmaxf(pm, wmax[w]);
            const int slot = (blockIdx.x + gridDim.x * blockIdx.y) & (DELTA_SLOTS - 1);
            atomicMax(reinterpret_cast<int*>(p.delta) + slot, __float_as_int(pm));
        }
    }
    DGQ_STAMP(9); DGQ_DIAG_DRAIN(); DGQ_STAMP(10); DGQ_STAMP_REAL(11);
    DGQ_DIAG_FLUSH(attn, NW, wid + 16384, lane);
}

// 16-bit tensors: both key halves leave as fp32 parts; o = round(part0 + part1) — the fp32 call's sum, rounded once to the tensor's type
template <typename T>
__global__ __launch_bounds__(256) void attn3_add16_kernel(T* __restrict__ o, const float* __restrict__ part0, const float* __restrict__ part1, int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) {
        const float4 a = reinterpret_cast<const float4*>(part0)[i], c = reinterpret_cast<const float4*>(part1)[i];
        T t[4] = {dgq_from_float<T>(a.x + c.x), dgq_from_float<T>(a.y + c.y), dgq_from_float<T>(a.z + c.z), dgq_from_float<T>(a.w + c.w)};
        reinterpret_cast<uint2*>(o)[i] = *reinterpret_cast<const uint2*>(t);
    }
}

__global__ __launch_bounds__(256) void attn3_add_kernel(float* __restrict__ o, const float* __restrict__ part, int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) {
        float4 a = reinterpret_cast<float4*>(o)[i];
        const float4 c = reinterpret_cast<const float4*>(part)[i];
        a.x += c.x; a.y += c.y; a.z += c.z; a.w += c.w;
        reinterpret_cast<float4*>(o)[i] = a;
    }
}

// Key-split launches: merges the two halves' (m, l, m2) per query row into the statistics the P·V pass reads and takes the
// real-time δ maximum over the merged rows (mode 1).  One thread per (batch·head, query).  The P·V halves use the merged
// (m, l, δ), so their parts of o simply add (attn3_add_kernel).
__global__ __launch_bounds__(256) void attn3_merge_kernel(AttnParams p) {
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x, rows = (int64_t)p.B * p.H * p.T;
    float pm = 0.0f;
    if (row < rows) {
        const float4 a = *reinterpret_cast<const float4*>(p.stats_part + row * 4);
        const float4 c = *reinterpret_cast<const float4*>(p.stats_part + (rows + row) * 4);
        const float m = fmaxf(a.x, c.x);
        const float l = a.y * exp2f(a.x - m) + c.y * exp2f(c.x - m);
        p.stats[row * 2] = m;
        p.stats[row * 2 + 1] = l;
        if (p.mode == 1) pm = exp2f(fmaxf(a.z, c.z) - m) / l;
    }
    if (p.mode == 1) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) pm = fmaxf(pm, __shfl_xor(pm, o, 64));
        __shared__ float wmax[4];
        if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = pm;
        __syncthreads();
        if (threadIdx.x == 0) {
            pm = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
            atomicMax(reinterpret_cast<int*>(p.delta) + (blockIdx.x & (DELTA_SLOTS - 1)), __float_as_int(pm));
        }
    }
}

// TPS: key tiles per ring stage (see attn3_stats_kernel); TPS = 2 runs a two-stage ring — the same prefetch distance in tiles as the
// three-stage ring of single tiles, half the barriers.
template <int D, bool UNIFORM, int NW, int QM, bool VINT, int TPS = 1>
__global__ __launch_bounds__(64 * NW) void attn3_pv_kernel(AttnParams p) {
    using G = Geo<D, QM, VINT>;
    constexpr bool QI8 = G::QI8, KS = QM == 3;
    constexpr int ST = TPS > 1 ? 2 : G::PV_STAGES, NP = G::IMG_PIECES, TB = G::IMG_BYTES, SB = TPS * TB;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds8[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, h32 = lane >> 5;
    DGQ_DIAG_DECL
    DGQ_STAMP(0); DGQ_STAMP_REAL(1); DGQ_STAMP_WHERE(2);
#ifdef DGQ_DIAG
    dg.t[13] = 3;
#endif
    int bx, bh;
    attn_block_coords(p.xcd, bx, bh);
    const int b = bh / p.H, hd = bh - b * p.H;
    const int t = bx * (32 * NW) + wid * 32 + (lane & 31);
    const int tq = min(t, p.T - 1);
    const unsigned char* img_lane = p.planes + (int64_t)bh * p.NT * G::IMG_BYTES + lane * 16;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(lptr_t)lds8;
    const int nsp = gridDim.z, i0 = (int)((long)p.NT * blockIdx.z / nsp), i1 = (int)((long)p.NT * (blockIdx.z + 1) / nsp);   // key-split launches
#pragma unroll
    for (int i = 0; i < ST - 1; ++i)
#pragma unroll
        for (int j = 0; j < TPS; ++j)
            issue_image<NP, NW>(img_lane + (int64_t)min(i0 + TPS * i + j, i1 - 1) * G::IMG_BYTES, lds_base + i * SB + j * TB, wid);
    bf16x8 qf[3][QI8 ? 1 : G::NKK];
    v4i qc[G::NK32];
    float4 qt = make_float4(1.0f, 0.0f, 0.0f, 0.0f);
    if constexpr (QI8) load_q_i8<D>(qc, qt, p, (int64_t)(b * p.T + tq) * p.H + hd, h32);
    else if constexpr (QM == 2) load_q1<D>(qf, reinterpret_cast<const unsigned short*>(p.q) + ((int64_t)(b * p.T + tq) * p.H + hd) * D, h32);
    else load_q<D>(qf, p.q + ((int64_t)(b * p.T + tq) * p.H + hd) * D, h32, p.fq[0], tq);
    if constexpr (QM == 2) {                              // (query scale, zero-point multiplier) written by the pre-pass
        const float2 t2 = *reinterpret_cast<const float2*>(p.qtab + ((int64_t)(b * p.T + tq) * p.H + hd) * 2);
        qt.x = t2.x;
        qt.y = t2.y;
        if constexpr (G::FOLDZ) fold_zmul<D>(qf, qt.y, h32);
    }
    float inv_dk = 1.0f, cq = 0.0f, dk = 1.0f;              // KS: see attn3_stats_kernel
    if constexpr (KS) {
        dk = p.fq[1].delta[0];
        inv_dk = 1.0f / dk;
        cq = -(p.fq[1].zp[0] - 0.5f * (p.fq[1].qmax + 1.0f)) * qt.z;
    }
    const float2 ml = *reinterpret_cast<const float2*>(p.stats + ((int64_t)bh * p.T + tq) * 2);      // (m, l) of the row: ONE load, issued with the others
    const float l = ml.y;
    float delta;
    if (p.mode == 1) {                                   // real-time δ: maximum of the statistics pass's slots
        delta = p.delta[lane];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) delta = fmaxf(delta, __shfl_xor(delta, o, 64));
    } else {
        delta = p.delta[0];
    }
    const float sl2 = p.scale * LOG2E * qt.x * dk;
    const float nsl2 = -sl2;
    const float m = ml.x - cq * sl2;                      // the row maximum in the units of the score tiles
    const float a0 = m + log2f(l) + log2f(delta);       // −log2(p/δ) = a0 − s2
    const float inv_l = 1.0f / l;
    // log2 codes by the magic-number route: rne(x) = bits(x + 1.5·2^23) − bits(1.5·2^23); clamp as integers to
    // [0, min(2^b−1, 127)] (2^-127 and below is 0 to the fp32 sum), then 2^-code = bits(1.0) − (code << 23)
    constexpr float MAGIC = 12582912.0f;
    constexpr int MAGIC_I = 0x4B400000;
    const int cmax_i = MAGIC_I + min((int)p.qmax, 127);
    float p_bypass = 0.0f;                               // unquantised probability of key 0 (start-peak)
    float psum = 0.0f;                                   // VINT: Σ_s p̂/δ of this lane's keys (zero-point term of V)
    v16f oacc[G::NDT];
#pragma unroll
    for (int j = 0; j < G::NDT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[j][r] = 0.0f;
    constexpr int VPL = G::DV * G::VLD;
    DGQ_STAMP(3);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (the stats / δ loads above share the counter: drain once)
    __builtin_amdgcn_s_barrier();
    DGQ_STAMP(4);
    int stage = 0, istage = ST - 1;
    // (TPS > 1: the host launches this form only for key ranges of a whole number of stages)
    for (int ib = i0; ib < i1; ib += TPS) {
#pragma unroll
      for (int jt = 0; jt < TPS; ++jt)
        issue_image<NP, NW>(img_lane + (int64_t)min(ib + TPS * (ST - 1) + jt, i1 - 1) * G::IMG_BYTES, lds_base + istage * SB + jt * TB, wid);
#pragma unroll
      for (int jt = 0; jt < TPS; ++jt) {
        const int i = ib + jt;
        const int s0 = i * KT;
        const unsigned short* kbc = reinterpret_cast<const unsigned short*>(lds8 + stage * SB + jt * TB);
        const unsigned short* vtc = kbc + G::K_ELEMS;
        v16f acc;
        if constexpr (QI8) acc = score_tile_i8<D, KS>(lds8 + stage * SB + jt * TB, qc, qt, lane, i == 0 && p.kskip > 0, inv_dk, cq);
        else if constexpr (QM == 2) acc = score_tile_q1<D>(kbc, qf, qt.y, lane);
        else acc = score_tile<D>(kbc, qf, lane);
        // interior tiles carry no per-key conditions; only the first tile (bypassed column) and a partial last tile do
        const bool edge = (s0 + KT > p.S) || (s0 < p.skip);      // block-uniform
        auto quantise = [&](auto edge_tag) {
            constexpr bool EDGE = decltype(edge_tag)::value;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float ph;
                if (UNIFORM) {
                    const float pr = exp2f(fmaf(acc[r], sl2, -m)) * inv_l;     // always_zero: code = clamp(rne(p/δ), 0, 2^b−1)
                    ph = fminf(fmaxf(rintf(__fdiv_rn(pr, delta)), 0.0f), p.qmax);
                } else {
                    const float x = fmaf(acc[r], nsl2, a0);
                    int ci = __float_as_int(x + MAGIC);    // out-of-range x lands outside [MAGIC_I, cmax_i]: clamped below
                    ci = min(max(ci, MAGIC_I), cmax_i);
                    ph = __int_as_float(0x3F800000 - (ci << 23));              // (MAGIC_I << 23) wraps to 0
                }
                if (EDGE) {
                    const int s = s0 + key_of(r, h32);
                    if (s >= p.S) ph = 0.0f;
                    else if (s < p.skip) {
                        p_bypass = exp2f(fmaf(acc[r], sl2, -m)) * inv_l;
                        ph = 0.0f;
                    }
                }
                acc[r] = ph;
                if constexpr (VINT && !G::VONES) psum += ph;
            }
        };
        bf16x8 pf[2];                                    // B fragments of the two 16-key steps
        if (UNIFORM && !edge) {
            // interior tiles of the uniform (always_zero) quantiser: code = min(rne(p/δ), 2^b − 1), and p/δ is ONE exponential,
            // 2^(s2 − m − log2 l − log2 δ) (p >= 0: no lower clamp).  rne by the magic add, the clamp on the integer image, and
            // bits(x + MAGIC) − MAGIC is the code as an exact float whose upper half is its bf16: 5.5 VALU per score (the
            // generic form below — exp2f with its range fix-up, an IEEE division, rint, two clamps — is ~25)
            const int cmaxu_i = MAGIC_I + (int)p.qmax;
            const f2 sl2v = {sl2, sl2}, na0v = {-a0, -a0}, magic = {MAGIC, MAGIC}, nmagic = {-MAGIC, -MAGIC};
            f2 ps2 = {0.0f, 0.0f};
            unsigned w[8];
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const f2 a = pk_fma(f2{acc[r], acc[r + 1]}, sl2v, na0v);
                const f2 y = pk_add(f2{__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)}, magic);
                const f2 c = pk_add(f2{__int_as_float(min(__float_as_int(y.x), cmaxu_i)), __int_as_float(min(__float_as_int(y.y), cmaxu_i))}, nmagic);
                if constexpr (VINT && !G::VONES) ps2 = pk_add(ps2, c);
                w[r >> 1] = __builtin_amdgcn_perm(__float_as_uint(c.y), __float_as_uint(c.x), 0x07060302u);
            }
            if constexpr (VINT && !G::VONES) psum += ps2.x + ps2.y;
            pf[0] = __builtin_bit_cast(bf16x8, make_uint4(w[0], w[1], w[2], w[3]));
            pf[1] = __builtin_bit_cast(bf16x8, make_uint4(w[4], w[5], w[6], w[7]));
        } else if (!UNIFORM && !edge) {
            // interior tiles of the log2 quantiser never form p̂ as a float: the clamped magic-number integers of a key
            // pair are merged (their low halves = the two codes), and 2^-code as a bf16 is 0x3F80 − (code << 7), so the
            // packed pair is 0x3F803F80 − 128·(c0 | c1 << 16) — one 24-bit multiply-add (codes <= 127: no borrow)
            int ci[16];
            const f2 nsl2v = {nsl2, nsl2}, a0m = {a0, a0}, magic = {MAGIC, MAGIC};
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const f2 y = pk_add(pk_fma(f2{acc[r], acc[r + 1]}, nsl2v, a0m), magic);
                ci[r] = min(max(__float_as_int(y.x), MAGIC_I), cmax_i);
                ci[r + 1] = min(max(__float_as_int(y.y), MAGIC_I), cmax_i);
                if constexpr (VINT && !G::VONES) psum += __int_as_float(0x3F800000 - (ci[r] << 23)) + __int_as_float(0x3F800000 - (ci[r + 1] << 23));
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                unsigned w[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const unsigned pair = __builtin_amdgcn_perm((unsigned)ci[8 * ks + 2 * i + 1], (unsigned)ci[8 * ks + 2 * i], 0x05040100u);
                    w[i] = (unsigned)(__mul24((int)pair, -128) + 0x3F803F80);
                }
                pf[ks] = __builtin_bit_cast(bf16x8, make_uint4(w[0], w[1], w[2], w[3]));
            }
        } else {
            if (edge) quantise(std::true_type{});
            else quantise(std::false_type{});
            // exact bf16 = upper halves of the fp32 words
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#define PK(i) ((__float_as_uint(acc[8 * ks + 2 * (i)]) >> 16) | (__float_as_uint(acc[8 * ks + 2 * (i) + 1]) & 0xFFFF0000u))
                pf[ks] = __builtin_bit_cast(bf16x8, make_uint4(PK(0), PK(1), PK(2), PK(3)));
#undef PK
            }
        }
#pragma unroll
        for (int j = 0; j < G::NDT; ++j) {
            const unsigned short* vp = vtc + (j * 32 + (lane & 31)) * G::VLD + 8 * h32;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x8 vh = *reinterpret_cast<const bf16x8*>(vp + 16 * ks);
                if constexpr (!VINT) {
                    const bf16x8 vm = *reinterpret_cast<const bf16x8*>(vp + VPL + 16 * ks);
                    const bf16x8 vl = *reinterpret_cast<const bf16x8*>(vp + 2 * VPL + 16 * ks);
                    oacc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl, pf[ks], oacc[j], 0, 0, 0);
                    oacc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vm, pf[ks], oacc[j], 0, 0, 0);
                }
                oacc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, pf[ks], oacc[j], 0, 0, 0);
            }
        }
      }
        wait_image<NP, TPS * (ST - 2), NW>(wid);
        __builtin_amdgcn_s_barrier();
        stage = (stage + 1 == ST) ? 0 : stage + 1;
        istage = (istage + 1 == ST) ? 0 : istage + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // no DMA may outlive the block's LDS allocation
    DGQ_STAMP(5);
    if (p.skip > 0) p_bypass = __shfl(p_bypass, lane & 31, 64);   // key 0 lives in the lower half-wave
    if constexpr (G::VONES) {                            // Σ p̂/δ over ALL keys = O^T[D] (the ones row): d = D sits in one half-wave
        constexpr int kd = D % 32, rr = (kd & 3) + 4 * (kd >> 3), hh = (kd >> 2) & 1;
        psum = __shfl(oacc[D / 32][rr], (lane & 31) + 32 * hh, 64);
    } else if (VINT) {
        psum += __shfl_xor(psum, 32, 64);                // both half-waves hold keys of the same query
    }
    // VINT: (δv, z'v) per head-dim element staged once through the (now idle) LDS ring — read per output element from
    // global memory they were 5·16 dependent round trips between the stores (D = 160: +10 us on a 64-row call)
    // likewise the (fake-quantised) value row of the bypassed start-peak key
    float* vtab = reinterpret_cast<float*>(lds8);
    __syncthreads();                                     // every wave's DMA has landed (its own vmcnt(0) above): the ring is free
    // three tables of DV entries, always all three (an absent one holds its neutral value): the epilogue below is then free of
    // per-element conditions
    for (int d = tid; d < G::DV; d += 64 * NW) {
        float t0 = 1.0f, t1 = 0.0f, t2 = 0.0f;
        if (VINT) {
            const int idx = (p.fq[2].mode == 0 || d >= D) ? 0 : d;
            t0 = p.fq[2].delta[idx];
            t1 = p.fq[2].zp[idx] - 0.5f * (p.fq[2].qmax + 1.0f);
        }
        if (p.skip > 0 && d < D) t2 = fq_apply(p.fq[2], load_any(p.v, p.io_dtype, ((int64_t)(b * p.S) * p.H + hd) * D + d), 0, d);
        vtab[d] = t0; vtab[G::DV + d] = t1; vtab[2 * G::DV + d] = t2;
    }
    __syncthreads();
    DGQ_STAMP(6);
    // The O^T tiles (lane = query, registers = 16 head-dim elements in runs of 4) leave through the idle ring, one 32 x 32 block at a
    // time: dequantised with the block's table entries (read as 16-byte runs, all of a block's reads issued before its first
    // use), written as they stand (ds_write_b128 per run of 4), read back row-wise, stored 16 bytes per lane in 128-byte row
    // segments.  Before: per element two dependent LDS reads, a branch on the start-peak flag and a 4-byte store at a 32-row
    // stride — 41-57 % of this launch at one wave per SIMD (profiles/r05_attention_timeline.txt).  Same-wave LDS round trip: no
    // barrier.
    {
        constexpr int ELD = 36;
        float* ep = reinterpret_cast<float*>(lds8) + 3 * G::DV + wid * (32 * ELD);
        const int tw0 = bx * (32 * NW) + wid * 32;          // first query of this wave
        const int er = lane >> 3, ec = (lane & 7) * 4;
        // a split launch: the second key half's part goes to fp32 scratch (attn3_add_kernel adds it to an fp32 o); for 16-bit tensors so
        // does the first (attn3_add16_kernel rounds their sum into o)
        float* const fpart = blockIdx.z > 0 ? p.o_part : ((gridDim.z > 1 && p.io_dtype != DGQ_F32) ? p.o_part0 : nullptr);
        const float pb = p.skip > 0 ? p_bypass : 0.0f;
#pragma unroll
        for (int j = 0; j < G::NDT; ++j) {
            DGQ_STAMP_NOW(dg_e0);
            float4 t0[4], t1[4], t2[4];
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) {
                const int d = j * 32 + 8 * k4 + 4 * h32;
                t0[k4] = *reinterpret_cast<const float4*>(vtab + d);
                t1[k4] = *reinterpret_cast<const float4*>(vtab + G::DV + d);
                t2[k4] = *reinterpret_cast<const float4*>(vtab + 2 * G::DV + d);
            }
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) {
                const float a0[4] = {t0[k4].x, t0[k4].y, t0[k4].z, t0[k4].w}, a1[4] = {t1[k4].x, t1[k4].y, t1[k4].z, t1[k4].w};
                const float a2[4] = {t2[k4].x, t2[k4].y, t2[k4].z, t2[k4].w};
                float o4[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * k4 + e;                 // d = j·32 + key_of(r, h32) = j·32 + 8·k4 + 4·h32 + e
                    float o;
                    if (VINT) o = delta * (a0[e] * (oacc[j][r] - a1[e] * psum));     // o = δw·δv(d)·(Σ p̂'·c'v − z'v(d)·Σ p̂')
                    else o = delta * oacc[j][r];
                    o4[e] = o + pb * a2[e];                   // start-peak: + p(key 0)·v̂(key 0)   (pb = 0 otherwise)
                }
                *reinterpret_cast<float4*>(ep + (lane & 31) * ELD + 8 * k4 + 4 * h32) = make_float4(o4[0], o4[1], o4[2], o4[3]);
            }
            DGQ_STAMP_ACC(7, dg_e0);                           // (diagnostic) dequantise + ds_write
            DGQ_STAMP_NOW(dg_e1);
            float4 ov[4];
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) ov[ps] = *reinterpret_cast<const float4*>(ep + (er + 8 * ps) * ELD + ec);
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                const int row = er + 8 * ps, tt = tw0 + row, d0 = j * 32 + ec;
                const float4 v = ov[ps];
                if (tt < p.T && d0 < D) {
                    const int64_t oi = ((int64_t)(b * p.T + tt) * p.H + hd) * D + d0;
                    if (fpart) {
                        *reinterpret_cast<float4*>(fpart + oi) = v;
                    } else if (p.io_dtype == DGQ_F32) {
                        *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.o) + oi) = v;
                    } else if (p.io_dtype == DGQ_F16) {
                        __half hv[4] = {__float2half(v.x), __float2half(v.y), __float2half(v.z), __float2half(v.w)};
                        *reinterpret_cast<uint2*>(reinterpret_cast<__half*>(p.o) + oi) = *reinterpret_cast<const uint2*>(hv);
                    } else {
                        __hip_bfloat16 hv[4] = {__float2bfloat16(v.x), __float2bfloat16(v.y), __float2bfloat16(v.z), __float2bfloat16(v.w)};
                        *reinterpret_cast<uint2*>(reinterpret_cast<__hip_bfloat16*>(p.o) + oi) = *reinterpret_cast<const uint2*>(hv);
                    }
                }
            }
            DGQ_STAMP_ACC(12, dg_e1);                          // (diagnostic) ds_read + global store issue
        }
    }
    DGQ_STAMP(9); DGQ_DIAG_DRAIN(); DGQ_STAMP(10); DGQ_STAMP_REAL(11);
    DGQ_DIAG_FLUSH(attn, NW, wid, lane);
}

template <int D, int QM, bool VINT>
static int launch_attn3(AttnParams p, const void* q_raw, unsigned char* planes, float* qfq, hipStream_t st) {
    using G = Geo<D, QM, VINT>;
    constexpr bool QI8 = G::QI8;
    constexpr int PQM = QM == 3 ? 1 : QM;                  // the pre-pass writes the same images for both int8-score forms
    p.planes = planes;
    p.img_bytes = G::IMG_BYTES;
    constexpr int stats_lds = G::STATS_STAGES * G::K_PIECES * 1024;
    constexpr int TPS_S = (3 * DGQ_ATTN_TPS_STATS * G::K_PIECES * 1024 <= 160 * 1024) ? DGQ_ATTN_TPS_STATS : 2;                                // key tiles per stage of the wide statistics launches (three stages)
    constexpr int stats2_lds = D <= 64 ? 3 * TPS_S * G::K_PIECES * 1024 : 0;
    // (+ the P·V epilogue's scratch in the idle ring: 3·DV floats of V tables, then 32 x 36 floats per wave — 8 waves at most)
    constexpr int pv_ring = G::PV_STAGES * G::IMG_BYTES, pv_scratch = 3 * G::DV * 4 + 8 * 32 * 36 * 4;
    constexpr int pv_lds = pv_ring > pv_scratch ? pv_ring : pv_scratch;
    constexpr int TPS_P = (2 * DGQ_ATTN_TPS_PV * G::IMG_BYTES <= 160 * 1024) ? DGQ_ATTN_TPS_PV : 2;     // ... of the wide P·V launches (two stages)
    constexpr int pv2_ring = D <= 64 ? 2 * TPS_P * G::IMG_BYTES : 0, pv2_lds = pv2_ring > pv_scratch ? pv2_ring : pv_scratch;
    static_assert(stats_lds <= 160 * 1024 && pv_lds <= 160 * 1024 && stats2_lds <= 160 * 1024 && pv2_lds <= 160 * 1024, "LDS ring too large");
    // up to 138 KB of dynamic LDS (D = 160): opt in, once per device (the attribute is per device)
    static std::atomic<bool> attr_set[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !attr_set[dev].load(std::memory_order_acquire)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn3_stats_kernel<D, 4, QM>), hipFuncAttributeMaxDynamicSharedMemorySize, stats_lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn3_pv_kernel<D, true, 4, QM, VINT>), hipFuncAttributeMaxDynamicSharedMemorySize, pv_lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn3_pv_kernel<D, false, 4, QM, VINT>), hipFuncAttributeMaxDynamicSharedMemorySize, pv_lds);
        if constexpr (D <= 64) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn3_stats_kernel<D, 8, QM>), hipFuncAttributeMaxDynamicSharedMemorySize, stats_lds);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn3_stats_kernel<D, 8, QM, TPS_S>), hipFuncAttributeMaxDynamicSharedMemorySize, stats2_lds);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn3_pv_kernel<D, true, 8, QM, VINT>), hipFuncAttributeMaxDynamicSharedMemorySize, pv_lds);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn3_pv_kernel<D, false, 8, QM, VINT>), hipFuncAttributeMaxDynamicSharedMemorySize, pv_lds);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn3_pv_kernel<D, true, 8, QM, VINT, TPS_P>), hipFuncAttributeMaxDynamicSharedMemorySize, pv2_lds);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn3_pv_kernel<D, false, 8, QM, VINT, TPS_P>), hipFuncAttributeMaxDynamicSharedMemorySize, pv2_lds);
        }
        if (dev >= 0 && dev < 64) attr_set[dev].store(true, std::memory_order_release);
    }
    // the main kernels read fp32 queries: the caller's tensor when it is fp32 and aqtizer_q is not fused, else a scratch
    // copy (converted / fake-quantised) written by extra blocks of the pre-pass; QI8: int8 codes + per-query table
    const bool q_copy = QM != 0 || ((p.fq[0].mode >= 0 || p.io_dtype != DGQ_F32) && qfq != nullptr);
    const dim3 pgrid(2 * p.NT + (q_copy ? (p.T + 31) / 32 : 0), p.B * p.H);   // K tiles, V tiles, Q row blocks
    float* dreset = p.mode == 1 ? p.delta : nullptr;
    // (the granules of the single-launch form's exchange only where that form can take the call: one per 128-row workgroup)
    const long one_wgs = (long)((p.T + 127) / 128) * p.B * p.H;
    const int n_reset = DELTA_SLOTS + ((p.NT <= 8 && one_wgs <= DELTA_GRANULES) ? (512 - 4 * DELTA_SLOTS) / 4 + 2 * (int)one_wgs : 0);
#define DGQ_PREP(TT) hipLaunchKernelGGL((attn3_prep_kernel<D, TT, PQM, VINT>), pgrid, dim3(256), 0, st, (const TT*)p.k, (const TT*)p.v, planes, \
                                        p.B, p.H, p.S, p.NT, p.fq[1], p.fq[2], dreset, n_reset, (const TT*)q_raw, qfq, p.T, p.fq[0])
    if (p.io_dtype == DGQ_F16) DGQ_PREP(__half);
    else if (p.io_dtype == DGQ_BF16) DGQ_PREP(__hip_bfloat16);
    else DGQ_PREP(float);
#undef DGQ_PREP
    p.kskip = 0;
    p.qcodes = nullptr;
    p.qtab = nullptr;
    if (QI8) {
        p.qcodes = reinterpret_cast<const int8_t*>(qfq);
        p.qtab = reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(qfq) + (size_t)p.B * p.T * p.H * G::DP32);
        p.kskip = p.fq[1].skip;
        p.q = nullptr;
        p.fq[0].mode = -1;
    } else if (QM == 2) {
        p.q = qfq;                                             // centred codes as bf16
        p.qtab = reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(qfq) + (size_t)p.B * p.T * p.H * D);   // (query scale, zero-point multiplier) per query
        p.fq[0].mode = -1;
    } else if (q_copy) {
        p.q = qfq;
        p.fq[0].mode = -1;
    } else {
        p.q = reinterpret_cast<const float*>(q_raw);
    }
    // short key ranges: statistics, δ exchange and P·V in ONE launch (attn_one.hip) where that form takes the call
    {
        const int one = dgq_attention_one_launch(p, D, QM, VINT, p.mode == 1 ? reinterpret_cast<unsigned*>(p.delta) + 128 : nullptr, st);
        if (one <= 0) return one;
    }
    // 8-wave blocks (256 query rows) when they still give one block per CU; register budgets allow it for D <= 64
    // ... unless the 8-wave grid ends in a mostly empty round: two 8-wave workgroups share a CU (512 resident), and e.g. 640 of
    // them (T = 1024, B·H = 160) run as 1.25 rounds in the time of 2 — the 4-wave grid's last round costs a fraction of that
    // (its lone workgroups have their SIMDs to themselves): 165 -> 152 us for that call, while 4096 rows at B·H = 80 (2.5
    // rounds) stay on the wide form (762 against 813 us)
    const long nblk8 = (long)((p.T + 255) / 256) * p.B * p.H;
    const long rounds8 = (nblk8 + 511) / 512;
    const bool tail_ok = nblk8 <= 512 || 10 * nblk8 >= 8 * 512 * rounds8;
    const bool wide = D <= 64 && nblk8 >= 256 && tail_ok;
    // Key split: a grid that leaves the chip under-filled (fewer 128-row workgroups than ~0.9 per CU) runs each (query block,
    // batch·head) as TWO workgroups over the two halves of the key tiles.  The statistics halves are merged by attn3_merge_kernel
    // ; the P·V halves use the merged (m, l, δ), so their parts simply add (the second half's part goes to a scratch tensor,
    // attn3_add_kernel adds it — float atomics into o were 2x slower than the unsplit call; 16-bit tensors: both parts to scratch,
    // attn3_add16_kernel rounds the sum).  >= 4 key tiles per half, D % 4 == 0.
    // Measured: 1024 x 1024, D = 80, B·H = 16: 86 -> 65 us; SDXL 4096 x 4096 at B·H = 10 (320 workgroups): 333 -> 282 us; 8-wave
    // grids (one workgroup per CU already) gain nothing and stay unsplit.
    const char* se = getenv("DGQ_ATTN_SPLIT");                 // grids below this many 4-wave workgroups split (0: never); read per call: tests toggle it
    const long split_below = se ? atol(se) : 448L;
    const long nblk4 = (long)((p.T + 127) / 128) * p.B * p.H;
    const int nsp = (!wide && nblk4 < split_below && p.NT >= 8 && D % 4 == 0) ? 2 : 1;
    if (wide) {
        if constexpr (D <= 64) {
            dim3 grid((p.T + 255) / 256, p.B * p.H), block(512);
            if (p.NT >= 4 * TPS_S && p.NT % TPS_S == 0) hipLaunchKernelGGL((attn3_stats_kernel<D, 8, QM, TPS_S>), grid, block, stats2_lds, st, p);
            else hipLaunchKernelGGL((attn3_stats_kernel<D, 8, QM>), grid, block, stats_lds, st, p);
            if (p.NT >= 4 * TPS_P && p.NT % TPS_P == 0) {
                if (p.mode == 3) hipLaunchKernelGGL((attn3_pv_kernel<D, true, 8, QM, VINT, TPS_P>), grid, block, pv2_lds, st, p);
                else hipLaunchKernelGGL((attn3_pv_kernel<D, false, 8, QM, VINT, TPS_P>), grid, block, pv2_lds, st, p);
            } else if (p.mode == 3) hipLaunchKernelGGL((attn3_pv_kernel<D, true, 8, QM, VINT>), grid, block, pv_lds, st, p);
            else hipLaunchKernelGGL((attn3_pv_kernel<D, false, 8, QM, VINT>), grid, block, pv_lds, st, p);
        }
    } else {
        dim3 grid((p.T + 127) / 128, p.B * p.H, nsp), block(256);
        hipLaunchKernelGGL((attn3_stats_kernel<D, 4, QM>), grid, block, stats_lds, st, p);
        if (nsp > 1) hipLaunchKernelGGL(attn3_merge_kernel, dim3((unsigned)(((long)p.B * p.H * p.T + 255) / 256)), dim3(256), 0, st, p);
        if (p.mode == 3) hipLaunchKernelGGL((attn3_pv_kernel<D, true, 4, QM, VINT>), grid, block, pv_lds, st, p);
        else hipLaunchKernelGGL((attn3_pv_kernel<D, false, 4, QM, VINT>), grid, block, pv_lds, st, p);
    }
    if (nsp > 1) {
        const int64_t n4 = (int64_t)p.B * p.T * p.H * D / 4;
        const dim3 ag((unsigned)((n4 + 255) / 256));
        if (p.io_dtype == DGQ_F32) hipLaunchKernelGGL(attn3_add_kernel, ag, dim3(256), 0, st, reinterpret_cast<float*>(p.o), p.o_part, n4);
        else if (p.io_dtype == DGQ_BF16) hipLaunchKernelGGL(attn3_add16_kernel<__hip_bfloat16>, ag, dim3(256), 0, st, reinterpret_cast<__hip_bfloat16*>(p.o), p.o_part0, p.o_part, n4);
        else hipLaunchKernelGGL(attn3_add16_kernel<__half>, ag, dim3(256), 0, st, reinterpret_cast<__half*>(p.o), p.o_part0, p.o_part, n4);
    }
    return dgq_launch_status("dgq_attention_f32(bf16x3)");
}

// bytes of the int8 query codes + per-query table of the QI8 path (0 when D is not instantiated here)
size_t dgq_attention_qi8_bytes(int B, int H, int T, int D) {
    const size_t dp32 = (size_t)(D + 31) / 32 * 32;
    const size_t i8 = (size_t)B * T * H * (dp32 + 16), q1 = (size_t)B * T * H * (D + 2) * sizeof(float);
    return i8 > q1 ? i8 : q1;
}

// bytes of the K/V tile images for one call (0 when D is not instantiated here)
size_t dgq_attention_bf16x3_bytes(int B, int H, int S, int D) {
    const size_t NT = (size_t)(S + KT - 1) / KT;
    size_t img;
    // the largest image of any operand format (Q1K3 appends a per-key table to the three K planes)
#define DGQ_IMG(DD) case DD: img = Geo<DD, 2, false>::IMG_BYTES > Geo<DD, 0, false>::IMG_BYTES ? Geo<DD, 2, false>::IMG_BYTES : Geo<DD, 0, false>::IMG_BYTES; break
    switch (D) {
        DGQ_IMG(8);
        DGQ_IMG(16);
        DGQ_IMG(40);
        DGQ_IMG(64);
        DGQ_IMG(80);
        DGQ_IMG(160);
        default: return 0;
    }
#undef DGQ_IMG
    return (size_t)B * H * NT * img;
}

// called from dgq_attention_f32 (attn_fused.hip) for the quantised modes; returns 1 when D is not instantiated here
int dgq_attention_bf16x3(const void* q, const void* k, const void* v, void* o, int io_dtype, int B, int H, int T, int S, int D,
                         float scale, int mode, int skip, float qmax, float* stats_ws, float* delta_ws, void* planes,
                         float* qfq, float* o_part, const dgq_attn_fq_t* fq, hipStream_t st) {
    AttnParams p;
    p.stats_part = stats_ws + (((size_t)B * H * T * 2 + 3) & ~(size_t)3);      // (the statistics area holds 10 floats per row: 2 merged + 2 x 4 partial, read as float4: 16-byte aligned for an odd row count too)
    p.o_part = o_part;
    p.o_part0 = o_part + (((size_t)B * T * H * D * sizeof(float) + 255) / 256 * 256) / sizeof(float);
    for (int i = 0; i < 3; ++i) {
        p.fq[i].mode = -1; p.fq[i].skip = 0; p.fq[i].qmax = 0.0f; p.fq[i].delta = nullptr; p.fq[i].zp = nullptr;
        if (fq && fq[i].mode >= 0) {
            p.fq[i].mode = fq[i].mode; p.fq[i].skip = fq[i].skip; p.fq[i].qmax = (float)((1 << fq[i].bits) - 1);
            p.fq[i].delta = fq[i].delta; p.fq[i].zp = fq[i].zero_point;
        }
    }
    p.q = nullptr; p.k = k; p.v = v; p.o = o; p.io_dtype = io_dtype; p.B = B; p.H = H; p.T = T; p.S = S; p.scale = scale;
    p.mode = mode; p.skip = skip;
    p.qmax = qmax; p.stats = stats_ws; p.delta = delta_ws;
    p.NT = (S + KT - 1) / KT;
    p.xcd = 1;
    p.qcodes = nullptr; p.qtab = nullptr; p.kskip = 0;
    unsigned char* img = reinterpret_cast<unsigned char*>(planes);
    // int8 score path: aqtizer_q and aqtizer_k both fused and scalar / per-token (one scale per token outside the d sum);
    // DGQ_ATTN_I8=0 keeps every call on the bf16x3 products (A/B runs)
    const char* i8_env = getenv("DGQ_ATTN_I8");                     // read per call: tests toggle it in-process
    const bool i8_off = i8_env != nullptr && i8_env[0] == '0';
    const bool qi8 = !i8_off && qfq != nullptr && p.fq[0].mode >= 0 && p.fq[0].mode <= 1 && p.fq[1].mode >= 0 && p.fq[1].mode <= 1 &&
                     p.fq[0].skip == 0;
    // any other fused aqtizer_q (a per-head-dim table on q or k, or an unquantised k): one exact plane of centred Q codes
    // against three K planes — three products instead of six
    const int qm = qi8 ? 1 : ((!i8_off && qfq != nullptr && p.fq[0].mode >= 0 && p.fq[0].skip == 0) ? 2 : 0);
    const bool kscalar = qi8 && p.fq[1].mode == 0;         // one (δk, z'k) for every key: folded into per-query constants
    // single-plane integer V: aqtizer_v fused and scalar / per-head-dim (its scale is outside the sum over keys)
    const bool vint = !i8_off && (p.fq[2].mode == 0 || p.fq[2].mode == 2);
#define DGQ_ATTN_V(DD, QQ) (vint ? launch_attn3<DD, QQ, true>(p, q, img, qfq, st) : launch_attn3<DD, QQ, false>(p, q, img, qfq, st))
#define DGQ_ATTN_CASE(DD) case DD: return qm == 1 ? (kscalar ? DGQ_ATTN_V(DD, 3) : DGQ_ATTN_V(DD, 1)) : (qm == 2 ? DGQ_ATTN_V(DD, 2) : DGQ_ATTN_V(DD, 0))
    switch (D) {
        DGQ_ATTN_CASE(8);
        DGQ_ATTN_CASE(16);
        DGQ_ATTN_CASE(40);
        DGQ_ATTN_CASE(64);
        DGQ_ATTN_CASE(80);
        DGQ_ATTN_CASE(160);
        default: return 1;
    }
#undef DGQ_ATTN_CASE
#undef DGQ_ATTN_V
}
