// gemm_panel.hip — the quantise-on-load member of the W4A8 GEMM family (round 5): Linear / 1x1 layers whose whole padded K fits an LDS
// panel (SD1.4: K = 320 ... 1280, after DGQ padding Kp = 384 ... 1664) run their activation quantiser (UniformAffineQuantizer.forward on
// the layer input, quant/quant_layer.py:295-299, with the LayerNorm / GroupNorm / SiLU in front of it folded in) INSIDE the GEMM launch:
// no int8 code matrix, no row-sum vector, no second launch (QuantLayer.forward, quant_layer.py:626-661, as one kernel).  Same tables,
// epilogue (gemm_tile.h) and results as dgq_quant_act + gemm_wxa8_kernel.
//
// Why (profiles/r05_small_launch_timeline.txt, s_memtime stamps of the 32x64 tile kernel on 8192 x 320 x 320 per-K): the K loop was
// 62 % of a workgroup's life at 1681 cycles per K tile for TWO MFMAs per wave — five co-resident waves per SIMD, each paying per K
// tile two LDS-DMA pieces (60-100 issue cycles each), a barrier, a counted wait, fragment reads for both operands and the ring
// bookkeeping; and in front of every such launch sits a quantise-on-load launch of about the same length that exists only to turn
// the fp32 rows into int8 codes in HBM.  The weights of a 32-row tile are read by ONE wave (no reuse), so LDS staging buys nothing.
//   * A (activation codes): the workgroup's 32 rows x the WHOLE K sit in LDS (the "panel"), written by the workgroup quantising its
//     rows itself; ONE barrier, no ring.  Image per K tile as in gemm_wxa8_kernel (128-byte rows, 16-byte pieces XOR-swizzled).
//   * W (int4): never touches LDS.  dgq_pack_w4 layout 2 stores the weights FRAGMENT-MAJOR: for every 32-column tile and every pair
//     of 32-wide K chunks one 1-KiB block in which lane l finds, at l·16, the 8 bytes of its column (l & 31) and K half (l >> 5) of
//     both chunks — one perfectly coalesced global_load_dwordx4 per two MFMA B operands, prefetched DT K tiles ahead in registers.
//   * a wave owns a 32 x 32 output tile; NW waves side by side (BN = 32·NW columns) x KW waves along K (each a contiguous
//     range of the K tiles; their partial tiles meet in LDS behind the loop, in a fixed order).
// Grid = column blocks x row blocks x problems.  (The round-5 form that took a materialised code matrix lost to the tile family inside
// the step — profiles/r05_panel_plan_in_step_ab.txt — and was removed in round 6.)
#include "gemm_tile.h"
#include "quant_common.h"

DGQ_DIAG_BUFFER(panel)

namespace {

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int TM, int NW, int KW>
struct PanelCfg {
    static constexpr int BM = 32 * TM, BN = 32 * NW, NWT = NW * KW, NT = 64 * NWT;
    static constexpr int EP_BYTES = NW * BM * (32 + 4) * 4;          // the store epilogue's transposition scratch (gemm_store_tile)
    static constexpr int RED_BYTES = (KW - 1) * NW * BM * 32 * 4;    // partial tiles of the K waves 1 .. KW-1
    static constexpr int VEC_BYTES = (3 * BM + 4 * BN) * 4;
};

__host__ __device__ constexpr int align16(int v) { return (v + 15) & ~15; }
// LDS map: region 0 (the A panel; once it is idle the epilogue's scratch + the K waves' partial tiles) | vtab | vcol | ctab |
// (per-K) the chunks' δ and z
template <int TM, int NW, int KW>
__host__ __device__ constexpr int panel_region0(int nk) {
    using C = PanelCfg<TM, NW, KW>;
    const int a = nk * C::BM * BK, e = C::EP_BYTES + C::RED_BYTES;
    return align16(a > e ? a : e);
}
__host__ __device__ constexpr int panel_ctab_bytes(bool per_m, int nk) { return per_m ? 0 : align16((NCH + 1) * nk * 4); }
template <int TM, int NW, int KW>
constexpr int panel_lds(bool per_m, int nk) {
    return panel_region0<TM, NW, KW>(nk) + PanelCfg<TM, NW, KW>::VEC_BYTES + panel_ctab_bytes(per_m, nk) +
           (!per_m ? nk * NCH * 16 + nk * BK * 4 : 0);      // (δ, z, 1/δ, ·) per chunk | decoded destination per source channel
}

template <bool PER_M, typename TIO, int TM, int NW, int KW>
__global__ __launch_bounds__(64 * NW * KW) void gemm_panel_kernel(GemmBatch bt, int n_major) {
    using Cfg = PanelCfg<TM, NW, KW>;
    constexpr int BM = Cfg::BM, BN = Cfg::BN, NT = Cfg::NT, NWT = Cfg::NWT;
    constexpr int ACCS = (!PER_M && TM == 1) ? 2 : 1;        // two accumulator sets: a chunk's flush issues behind the next chunk's MFMA
    constexpr int DT = 4;                                    // W prefetch depth in K tiles (two 16-byte loads per lane each)
    static_assert(DT == 4, "the wait ladder of the K loop is written for DT = 4");
    static_assert(TM == 1, "quantise-on-load: one 32-row tile per workgroup");
    const GemmParams& p = bt.p[bt.n > 1 ? blockIdx.z : 0];
    const int zsplit = bt.n > 1 ? 0 : blockIdx.z;
    gemm_prefetch_params(p);
    // XCD-aware tile order (as gemm_wxa8_kernel): XCD k owns a contiguous range of tiles — row-block major when the activations are
    // the larger operand, column-block major (n_major) when the weights are: the big operand is then fetched by one XCD's L2 only.
    int tile_n, tile_m;
    {
        const int gx = gridDim.x, gy = gridDim.y, T = gx * gy;
        const int bid = blockIdx.x + gx * blockIdx.y;
        const int q = T >> 3, r = T & 7, xcd = bid & 7;
        const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
        if (n_major) { tile_n = logical / gy; tile_m = logical - tile_n * gy; }
        else { tile_m = logical / gx; tile_n = logical - tile_m * gx; }
    }
    if (tile_n * BN >= p.N || tile_m * BM >= p.M) return;   // batch: a narrower problem than the grid (whole block)
    DGQ_DIAG_DECL
    DGQ_STAMP(0); DGQ_STAMP_REAL(1); DGQ_STAMP_WHERE(2);
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nq = wid % NW, kq = wid / NW;                  // column tile / K range of this wave
    const int n0 = tile_n * BN, m0 = tile_m * BM;
    const int nk_total = p.Kp / BK;
    const int kt_begin = zsplit * p.tiles_per_split;
    const int kt_end = min(nk_total, kt_begin + p.tiles_per_split);
    const int nk = kt_end - kt_begin;
    // K range of this wave inside the slice: tiles [w_t0, w_t0 + w_nk)
    const int per_kw = (nk + KW - 1) / KW;
    const int w_t0 = kq * per_kw;
    const int w_nk = max(0, min(nk - w_t0, per_kw));

    // ---- W stream: this wave's 32 columns, fragment-major (layout 2): 2 x 16 bytes per lane per K tile, DT tiles ahead.
    // The loads are asm statements with hand-counted waits: inside a loop whose steps are guarded (t < nk), hipcc's own bookkeeping
    // merges the paths conservatively and drains the ring (s_waitcnt vmcnt(0)) at every step.  NS = DT + 1 register slots: the loads
    // of tile t + DT go to the slot tile t − 1 has just left, so DT tiles stay in flight while tile t computes.
    const int ntile32 = (p.N + 31) >> 5;
    const int jt = min(tile_n * NW + nq, ntile32 - 1);      // a wave past N recomputes the last column tile and stores nothing
    const uint4* wsrc = reinterpret_cast<const uint4*>(p.wfrag) + ((int64_t)jt * (nk_total * 2) + (kt_begin + w_t0) * 2) * 64 + lane;
    constexpr int NS = DT + 1;
    v4i wr[NS][2];
    auto wload = [&](int slot_t, v4i (&dst)[2]) {
        const uint4* q = wsrc + (slot_t * 2) * 64;
        asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:1024"
                     : "=&v"(dst[0]), "=&v"(dst[1]) : "v"(q) : "memory");
    };
#pragma unroll
    for (int d = 0; d < DT; ++d)
        if (d < w_nk) wload(d, wr[d]);
#pragma unroll
    for (int j = 0; j < 2; ++j) wr[DT][j] = (v4i){0, 0, 0, 0};

    const uint32_t lds_base = (uint32_t)(uintptr_t)(lptr_t)smem;
    (void)lds_base;

    // ---- tables (as gemm_wxa8_kernel: asm loads at clamped indices, retired by one wait)
    const int region0 = panel_region0<TM, NW, KW>(nk);
    float* vtab = reinterpret_cast<float*>(smem + region0);  // [3][BM]: R0 R1 R2 | [4][BN]: alpha zw gamma vn
    float* vcol = vtab + 3 * BM;
    float* ctab = vcol + 4 * BN;                              // [nk·4] flush coefficients | [nk] clear flags
    // per-K: [nk·NCH] float4 (δ, z, 1/δ, ·) of the chunks | [<= nk·BK] per source channel: panel offset of its code byte at row 0 with
    // swizzle 0 (bits 0-19) and its chunk (bits 20-31) — the per-element work of the scatter is then one xor-add and one table read
    float4* tq = reinterpret_cast<float4*>(reinterpret_cast<uint8_t*>(ctab) + panel_ctab_bytes(PER_M, nk));
    uint32_t* tdst = reinterpret_cast<uint32_t*>(tq + nk * NCH);
    static_assert(BM <= NT && BN <= NT, "one row / column of the epilogue vectors per thread");
    constexpr int MYCH = NCH;
    const int n_coef = PER_M ? 0 : nk * MYCH, n_tab = PER_M ? 0 : n_coef + nk;
    // coefficient / clear flag e of the slice.  A chunk belongs to the K wave whose tile range holds it and, inside that wave, to
    // sequence (chunk mod ACCS): coef = δ_c − δ_next-of-its-sequence; the last chunk of a sequence in its wave's range — and the one
    // in front of a clear mark — takes the full δ_c.
    struct CoefIdx { int g, gn, tl; bool is_coef, seq_last, tile_end, not_last_tile; };
    auto coef_idx = [&](int e) {
        CoefIdx x;
        x.is_coef = e < n_coef;
        const int ec = x.is_coef ? e : 0;
        const int tc = ec / MYCH, ci = ec - tc * MYCH;
        const int t = x.is_coef ? tc : e - n_coef;
        const int range_last = min((t / per_kw + 1) * per_kw, nk) - 1;          // last tile of the owning K wave's range
        x.g = (kt_begin + tc) * NCH + ci;
        x.tile_end = (ci + ACCS >= MYCH);                                    // last chunk of its sequence in the tile
        x.seq_last = x.tile_end && tc == range_last;
        x.gn = min(x.tile_end ? (kt_begin + tc + 1) * NCH + (ci + ACCS - MYCH) : x.g + ACCS, nk_total * NCH - 1);
        x.tl = (kt_begin + t) * NCH + NCH - 1;
        x.not_last_tile = t != range_last;
        return x;
    };
    auto coef_val = [&](const CoefIdx& x, float d, float dn, uint32_t cf) {
        const bool clr = (cf & 0xFF) == 2;
        const float coef = (x.seq_last || (x.tile_end && clr)) ? d : d - dn;
        const float flag = (x.not_last_tile && clr) ? 1.0f : 0.0f;
        return x.is_coef ? coef : flag;
    };
    const bool final_ep = (p.splits == 1);
    const bool has_col = final_ep && tid < BN;
    float c_vn = 0.0f, c_d = 0.0f, c_dn = 0.0f;
    uint32_t c_cf = 0;
    const int ncol = min(n0 + tid, p.N - 1);
    float c_al = gload_f32(p.alpha + ncol), c_zw = gload_f32(p.zw + ncol), c_ga = gload_f32(p.gamma + ncol);
    if constexpr (PER_M) c_vn = gload_f32(p.vn + ncol);
    CoefIdx cx = {};
    if constexpr (!PER_M) {
        cx = coef_idx(min(tid, n_tab - 1));
        c_d = gload_f32(p.cdelta + cx.g); c_dn = gload_f32(p.cdelta + cx.gn); c_cf = gload_u8(p.cflush + cx.tl);
    }
    DGQ_STAMP(3);

    // ---- the activation quantiser of the layer (dgq_quant_act's arithmetic) on this workgroup's BM rows, written into the
    // panel image.  QL lanes share a row (64 / QL rows per wave at a time); per-M / scalar scales: natural K order, four codes per
    // dword store; per-K: each source channel's code byte goes to its packed position kdst[c] (padding stays zero).
    // (every load below is an ordinary one: the asm loads above are all OLDER, so hipcc's counted waits for these stay correct)
    {
        const dgq_gemm_act_t& act = p.act;
        const TIO* x = reinterpret_cast<const TIO*>(act.x);
        const int K = act.K;
        const float qmax = (float)((1 << act.bits) - 1), aoff = (float)(1 << (act.bits - 1));
        if constexpr (!PER_M) {
            for (int i = tid; i < nk * NCH; i += NT) {
                const float dl = p.cdelta[kt_begin * NCH + i];
                tq[i] = make_float4(dl, act.czp[kt_begin * NCH + i], dgq_rcp(dl), 0.0f);
            }
            for (int c = tid; c < K; c += NT) {
                const int kp = act.kdst[c];
                tdst[c] = (uint32_t)((kp >> 7) * (BM * BK) + (kp & 127)) | ((uint32_t)(kp >> 5) << 20);
            }
            for (int i = tid * 16; i < nk * BM * BK; i += NT * 16) *reinterpret_cast<uint4*>(smem + i) = make_uint4(0, 0, 0, 0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        DGQ_STAMP(14);
        constexpr int QL = (NWT >= 8) ? 16 : 8;              // lanes per row
        constexpr int RPWV = 64 / QL, RP = NWT * RPWV;       // rows per wave / per pass of the workgroup
        constexpr int SEG = 4;                               // 16-byte loads per lane in flight per round (5 spills: 168 VGPRs + scratch)
        const int sl = lane % QL, rw = lane / QL;
        auto panel_addr = [&](int row, int kp) {
            return (kp >> 7) * (BM * BK) + row * BK + ((((kp & 127) >> 4) ^ ((row >> 1) & 7)) << 4) + (kp & 15);
        };
        for (int r0 = 0; r0 < BM; r0 += RP) {
            const int row = r0 + wid * RPWV + rw;
            const bool rv = row < BM;
            const int m = min(m0 + min(row, BM - 1), p.M - 1);
            const TIO* xr = x + (int64_t)m * act.ldx;
            float mu = 0.0f, rstd = 1.0f;
            // (every load of a round goes out unconditionally at a clamped address — a load under `if (c < K)` makes hipcc branch around
            // it and wait for each one before the next is issued: five dependent round trips per pass of a 320-wide row)
            if (act.ln_gamma) {
                // LayerNorm statistics of the row: two passes (mean, then Σ(x − mean)²) in the summation order of quant_act.hip's
                // row_layernorm_stats — that kernel gives a row to 64 lanes, lane v owning columns 4v + 256i; here a row's QL lanes
                // each stand for the 64/QL lanes v = sl + QL·u of it and merge them in the order of its xor-shuffle tree (offsets 32,
                // 16, (8)), then finish the tree among themselves: mean and rstd — and with them the codes — equal the two-launch
                // form's bit for bit (the second pass re-reads the row from L1)
                constexpr int NU = 64 / QL;
                float sv[NU];
#pragma unroll
                for (int u = 0; u < NU; ++u) sv[u] = 0.0f;
                for (int i0 = 0; i0 < K; i0 += 256) {
                    float v[NU][4];
#pragma unroll
                    for (int u = 0; u < NU; ++u) load4<TIO>(xr + min(i0 + (sl + QL * u) * 4, K - 4), v[u]);
#pragma unroll
                    for (int u = 0; u < NU; ++u) {
                        const float t4 = (v[u][0] + v[u][1]) + (v[u][2] + v[u][3]);
                        sv[u] += (i0 + (sl + QL * u) * 4 < K) ? t4 : 0.0f;
                    }
                }
                auto tree = [&](float (&a)[NU]) {
#pragma unroll
                    for (int h = NU / 2; h > 0; h >>= 1)
#pragma unroll
                        for (int u = 0; u < h; ++u) a[u] += a[u + h];
                    float r = a[0];
#pragma unroll
                    for (int o = QL / 2; o > 0; o >>= 1) r += __shfl_xor(r, o, 64);
                    return r;
                };
                mu = tree(sv) / (float)K;
#pragma unroll
                for (int u = 0; u < NU; ++u) sv[u] = 0.0f;
                for (int i0 = 0; i0 < K; i0 += 256) {
                    float v[NU][4];
#pragma unroll
                    for (int u = 0; u < NU; ++u) load4<TIO>(xr + min(i0 + (sl + QL * u) * 4, K - 4), v[u]);
#pragma unroll
                    for (int u = 0; u < NU; ++u) {
                        float q4 = sv[u];
#pragma unroll
                        for (int e = 0; e < 4; ++e) q4 += (v[u][e] - mu) * (v[u][e] - mu);
                        sv[u] = (i0 + (sl + QL * u) * 4 < K) ? q4 : sv[u];
                    }
                }
                rstd = 1.0f / sqrtf(tree(sv) / (float)K + act.ln_eps);
            }
            DGQ_STAMP(15);
            const int img = act.pre_scale ? m / act.rows_per_image : 0;
            const float* psc = act.pre_scale ? act.pre_scale + (int64_t)img * K : nullptr;
            const float* psh = act.pre_scale ? act.pre_shift + (int64_t)img * K : nullptr;
            float rmd = 1.0f, rmz = 0.0f, rinv = 1.0f;
            if constexpr (PER_M) {
                const int li = m % p.L;
                rmd = p.mdelta[li]; rmz = p.mzp[li]; rinv = dgq_rcp(rmd);
            }
            const float bias = 128.0f - aoff;
            float partial = 0.0f;
            const uint32_t row_base = (uint32_t)(row * BK), row_swz = (uint32_t)(((row >> 1) & 7) << 4);
            const int kend = PER_M ? nk * BK : K;            // per-M also writes the zero codes of the K padding
            for (int c0 = sl * 4; c0 < kend; c0 += 4 * QL * SEG) {
                float v[SEG][4];
                int4 kd[SEG];
                float4 f0[SEG], f1[SEG];                     // folded norm: scale / shift (GroupNorm) or gamma / beta (LayerNorm)
#pragma unroll
                for (int j = 0; j < SEG; ++j) {
                    const int cc = min(c0 + j * 4 * QL, K - 4);
                    load4<TIO>(xr + cc, v[j]);
                    if constexpr (!PER_M) kd[j] = *reinterpret_cast<const int4*>(tdst + cc);
                }
                DGQ_STAMP_NOW(dg_q0);
                if (psc) {
#pragma unroll
                    for (int j = 0; j < SEG; ++j) {
                        const int cc = min(c0 + j * 4 * QL, K - 4);
                        f0[j] = *reinterpret_cast<const float4*>(psc + cc);
                        f1[j] = *reinterpret_cast<const float4*>(psh + cc);
                    }
                } else if (act.ln_gamma) {
#pragma unroll
                    for (int j = 0; j < SEG; ++j) {
                        const int cc = min(c0 + j * 4 * QL, K - 4);
                        f0[j] = *reinterpret_cast<const float4*>(act.ln_gamma + cc);
                        f1[j] = *reinterpret_cast<const float4*>(act.ln_beta + cc);
                    }
                }
#ifdef DGQ_DIAG
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("" : "+v"(v[0][0]), "+v"(v[1][0]), "+v"(v[2][0]), "+v"(v[3][0]));
#endif
                DGQ_STAMP_ACC(12, dg_q0);                                  // (diagnostic) the round's loads
                DGQ_STAMP_NOW(dg_q1);
#pragma unroll
                for (int j = 0; j < SEG; ++j) {
                    const int c = c0 + j * 4 * QL;                     // (past K: the clamped load's values, computed and discarded)
                    // ... unless the whole wave is past the end (K = 320: the second round holds one segment, not four): wave-uniform
                    if (c - sl * 4 >= kend) break;
                    if (psc) {
                        v[j][0] = v[j][0] * f0[j].x + f1[j].x; v[j][1] = v[j][1] * f0[j].y + f1[j].y;
                        v[j][2] = v[j][2] * f0[j].z + f1[j].z; v[j][3] = v[j][3] * f0[j].w + f1[j].w;
                    } else if (act.ln_gamma) {
                        v[j][0] = (v[j][0] - mu) * rstd * f0[j].x + f1[j].x; v[j][1] = (v[j][1] - mu) * rstd * f0[j].y + f1[j].y;
                        v[j][2] = (v[j][2] - mu) * rstd * f0[j].z + f1[j].z; v[j][3] = (v[j][3] - mu) * rstd * f0[j].w + f1[j].w;
                    }
                    if (act.pre_act == 1) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[j][e] = dgq_silu(v[j][e]);
                    }
                    if constexpr (PER_M) {
                        {
                            const float d4[4] = {rmd, rmd, rmd, rmd}, i4[4] = {rinv, rinv, rinv, rinv}, z4[4] = {rmz, rmz, rmz, rmz};
                            float qv[4], biased[4], fsum = 0.0f;
                            dgq_affine_code4_fast(v[j], d4, i4, z4, qmax, qv);
#pragma unroll
                            for (int e = 0; e < 4; ++e) biased[e] = (c < K) ? qv[e] + bias : 128.0f;
                            const uint32_t w = dgq_pack4(biased, fsum);
                            if (rv && c < kend) *reinterpret_cast<uint32_t*>(smem + panel_addr(row, c)) = w;
                            partial += (c < kend) ? fsum - 512.0f : 0.0f;
                        }
                    } else {
                        {
                            const uint32_t dst[4] = {(uint32_t)kd[j].x, (uint32_t)kd[j].y, (uint32_t)kd[j].z, (uint32_t)kd[j].w};
                            float d4[4], i4[4], z4[4], qv[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float4 t4 = tq[dst[e] >> 20];
                                d4[e] = t4.x; z4[e] = t4.y; i4[e] = t4.z;
                            }
                            dgq_affine_code4_fast(v[j], d4, i4, z4, qmax, qv);
                            const bool live = rv && c < K;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float scode = qv[e] - aoff;
                                // panel_addr(row, kp) = row·BK + (offset at row 0 ^ the row's 16-byte swizzle)
                                if (live) smem[row_base + ((dst[e] & 0xFFFFFu) ^ row_swz)] = (uint8_t)(int)scode;
                                partial += (c < K) ? d4[e] * scode : 0.0f;
                            }
                        }
                    }
                }
                DGQ_STAMP_ACC(7, dg_q1);                                   // (diagnostic) the round's arithmetic + panel writes
            }
#pragma unroll
            for (int o = QL / 2; o > 0; o >>= 1) partial += __shfl_xor(partial, o, 64);
            if (rv && sl == 0) {                             // the row's epilogue constants: R0 R1 R2
                float r0v = 1.0f, r1v = partial, r2v = 0.0f;
                if (PER_M) { r0v = rmd; r1v = rmd * partial; r2v = rmd * (p.offset - rmz); }
                vtab[row] = r0v; vtab[BM + row] = r1v; vtab[2 * BM + row] = r2v;
            }
        }
    }

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the panel (DMA), the first W tiles and the tables
    DGQ_STAMP(4);
    if (PER_M) asm volatile("" : "+v"(c_al), "+v"(c_zw), "+v"(c_ga), "+v"(c_vn));
    else asm volatile("" : "+v"(c_al), "+v"(c_zw), "+v"(c_ga), "+v"(c_d), "+v"(c_dn), "+v"(c_cf));
    __builtin_amdgcn_sched_barrier(0);
    if (has_col) {
        vcol[tid] = c_al; vcol[BN + tid] = c_zw; vcol[2 * BN + tid] = c_ga; vcol[3 * BN + tid] = c_vn;
    }
    if constexpr (!PER_M) {
        if (tid < n_tab) ctab[tid] = coef_val(cx, c_d, c_dn, c_cf);
        for (int e = tid + NT; e < n_tab; e += NT) {
            const CoefIdx x = coef_idx(e);
            ctab[e] = coef_val(x, p.cdelta[x.g], p.cdelta[x.gn], p.cflush[x.tl]);
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    DGQ_STAMP(5);

    // ---- K loop: per chunk TM ds_read_b128 (A fragments), one int4 -> int8 widening, TM MFMAs, and (per-K) the flush that is due
    const int lr = lane & 31, hh = lane >> 5;
    int a_off[TM][NCH];
#pragma unroll
    for (int cg = 0; cg < NCH; ++cg)
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row = i * 32 + lr;
            a_off[i][cg] = row * BK + (((2 * cg + hh) ^ ((row >> 1) & 7)) << 4);
        }
    v16i acc[ACCS][TM][1];
    v16f accf[TM][1];
    constexpr bool BIASED = !PER_M;                         // per-K (W4): totals carry DGQ_ACC_BIAS_I (gemm_device.h)
    constexpr int ACC0 = BIASED ? DGQ_ACC_BIAS_I : 0;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int a = 0; a < ACCS; ++a) acc[a][i][0][r] = ACC0;
            accf[i][0][r] = 0.0f;
        }
    auto flush = [&](const v16i (&ac)[TM][1], float coef) {
        const float sc = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, coef)));
        if (sc != 0.0f) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) accf[i][0][r] = __builtin_fmaf(sc, dgq_total_to_float<BIASED>(ac[i][0][r]), accf[i][0][r]);
        }
    };
    float pend = 0.0f;
    typedef float cvec_t __attribute__((ext_vector_type(NCH)));
    const float* tclr = ctab + nk * MYCH;
    // one K tile (slice tile ts): four chunks = the two register pairs w0 (chunks 0, 1) and w1 (chunks 2, 3)
    auto tile = [&](int ts, const v4i& w0, const v4i& w1) {
        cvec_t cq;
        float tc = 0.0f;
        if (!PER_M) {
            cq = *reinterpret_cast<const cvec_t*>(ctab + ts * MYCH);
            tc = tclr[ts];
        }
        const uint8_t* sa = smem + ts * (BM * BK);
#pragma unroll
        for (int ci = 0; ci < NCH; ++ci) {
            v4i af[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const v4i*>(sa + a_off[i][ci]);
            const v4i& w = ci < 2 ? w0 : w1;
            const uint32_t x = (uint32_t)((ci & 1) ? w[2] : w[0]), y = (uint32_t)((ci & 1) ? w[3] : w[1]);
            const v4i bf = (v4i){(int)(x & 0x0F0F0F0Fu), (int)((x >> 4) & 0x0F0F0F0Fu), (int)(y & 0x0F0F0F0Fu), (int)((y >> 4) & 0x0F0F0F0Fu)};
            if constexpr (PER_M) {
#pragma unroll
                for (int i = 0; i < TM; ++i) acc[0][i][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[i], bf, acc[0][i][0], 0, 0, 0);
            } else if constexpr (ACCS == 1) {
#pragma unroll
                for (int i = 0; i < TM; ++i) acc[0][i][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[i], bf, acc[0][i][0], 0, 0, 0);
                flush(acc[0], cq[ci]);
            } else {
                if (ci & 1) {
                    acc[1][0][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[0], bf, acc[1][0][0], 0, 0, 0);
                    flush(acc[0], pend);
                } else {
                    acc[0][0][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[0], bf, acc[0][0][0], 0, 0, 0);
                    flush(acc[1], pend);
                }
                pend = cq[ci];
            }
        }
        if (!PER_M) {
            if (__builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, tc)) != 0) {     // rare: a segment of running totals ends
                if constexpr (ACCS == 2) { flush(acc[1], pend); pend = 0.0f; }
#pragma unroll
                for (int a = 0; a < ACCS; ++a)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[a][i][0][r] = ACC0;
            }
        }
    };
    for (int tb = 0; tb < w_nk; tb += NS) {
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) {
            const int t = tb + sl;
            if (t < w_nk) {                                   // wave-uniform
                if (t + DT < w_nk) wload(t + DT, wr[(sl + DT) % NS]);
                // tile t's two loads are the oldest in flight; 2·min(DT, w_nk − 1 − t) younger ones may stay
                const int young = min(DT, w_nk - 1 - t);
                if (young >= 4) wait_vmcnt<8>();
                else if (young == 3) wait_vmcnt<6>();
                else if (young == 2) wait_vmcnt<4>();
                else if (young == 1) wait_vmcnt<2>();
                else wait_vmcnt<0>();
                asm volatile("" : "+v"(wr[sl][0]), "+v"(wr[sl][1]));       // the slot's registers are defined HERE for the compiler
                __builtin_amdgcn_sched_barrier(0);
                tile(w_t0 + t, wr[sl][0], wr[sl][1]);
            }
        }
    }
    if constexpr (!PER_M && ACCS == 2) flush(acc[1], pend);
    DGQ_STAMP(6);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // this wave's panel reads are complete ...
    __builtin_amdgcn_s_barrier();                            // ... everyone's: region 0 becomes scratch

    // ---- the K waves' partial tiles meet in LDS (behind the epilogue's own scratch): wave (0, nq) adds those of (1, nq), (2, nq) ...
    // in that order — int32 for per-M (exact), fp32 for per-K — and stores the tile alone
    if constexpr (KW > 1) {
        uint8_t* red = smem + Cfg::EP_BYTES;
        if (kq > 0) {
            uint8_t* dst = red + ((kq - 1) * NW + nq) * (BM * 32 * 4) + lane * 16;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    uint8_t* q = dst + (i * 4 + r4) * 1024;
                    if constexpr (PER_M) *reinterpret_cast<v4i*>(q) = (v4i){acc[0][i][0][4 * r4], acc[0][i][0][4 * r4 + 1], acc[0][i][0][4 * r4 + 2], acc[0][i][0][4 * r4 + 3]};
                    else *reinterpret_cast<v4f*>(q) = (v4f){accf[i][0][4 * r4], accf[i][0][4 * r4 + 1], accf[i][0][4 * r4 + 2], accf[i][0][4 * r4 + 3]};
                }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kq > 0) {
            DGQ_STAMP(9); DGQ_STAMP(10); DGQ_STAMP_REAL(11);
            DGQ_DIAG_FLUSH(panel, NWT, wid, lane);
            return;
        }
#pragma unroll
        for (int k = 1; k < KW; ++k) {
            const uint8_t* src = red + ((k - 1) * NW + nq) * (BM * 32 * 4) + lane * 16;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const uint8_t* q = src + (i * 4 + r4) * 1024;
                    if constexpr (PER_M) {
                        const v4i u = *reinterpret_cast<const v4i*>(q);
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[0][i][0][4 * r4 + e] += u[e];
                    } else {
                        const v4f u = *reinterpret_cast<const v4f*>(q);
#pragma unroll
                        for (int e = 0; e < 4; ++e) accf[i][0][4 * r4 + e] += u[e];
                    }
                }
        }
    }

    gemm_store_tile<PER_M, TIO, BM, BN, 1, NW, 1, Cfg::EP_BYTES, TM, 1>(p, zsplit, smem, vtab, vcol, nq, lane, 0, nq, 0, m0, n0, acc[0],
                                                                       accf DGQ_DIAG_ARG);
    DGQ_STAMP(9);
    DGQ_DIAG_DRAIN();
    DGQ_STAMP(10); DGQ_STAMP_REAL(11);
#ifdef DGQ_DIAG
    dg.t[13] = (unsigned long long)nk | ((unsigned long long)tile_m << 16) | ((unsigned long long)tile_n << 32) | ((unsigned long long)wid << 48);
#endif
    DGQ_DIAG_FLUSH(panel, NWT, wid, lane);
}

template <bool PER_M, typename TIO, int TM, int NW, int KW>
void launch_panel(const GemmBatch& bt, hipStream_t st) {
    using Cfg = PanelCfg<TM, NW, KW>;
    const GemmParams& p = bt.p[0];
    int maxN = 0, maxM = 0, max_tps = 0;
    size_t a_bytes = 0, w_bytes = 0;
    for (int i = 0; i < bt.n; ++i) {
        maxN = bt.p[i].N > maxN ? bt.p[i].N : maxN;
        maxM = bt.p[i].M > maxM ? bt.p[i].M : maxM;
        max_tps = bt.p[i].tiles_per_split > max_tps ? bt.p[i].tiles_per_split : max_tps;
        a_bytes += (size_t)bt.p[i].M * bt.p[i].Kp;
        w_bytes += (size_t)bt.p[i].N * bt.p[i].Kp / 2;
    }
    const int lds = panel_lds<TM, NW, KW>(PER_M, max_tps);
    static std::atomic<bool> attr_set[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !attr_set[dev].load(std::memory_order_acquire)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_panel_kernel<PER_M, TIO, TM, NW, KW>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (dev >= 0 && dev < 64) attr_set[dev].store(true, std::memory_order_release);
    }
    dim3 grid((maxN + Cfg::BN - 1) / Cfg::BN, (maxM + Cfg::BM - 1) / Cfg::BM, bt.n > 1 ? bt.n : p.splits), block(Cfg::NT);
    hipLaunchKernelGGL((gemm_panel_kernel<PER_M, TIO, TM, NW, KW>), grid, block, lds, st, bt, w_bytes > a_bytes ? 1 : 0);
}

// configurations (TM, NW, KW): key = TM·1000 + NW·10 + KW.  (Round 5 also carried ten configurations that took a materialised code
// matrix — measured slower inside the step than the tile family, profiles/r05_panel_plan_in_step_ab.txt — and two 16-wave
// quantise-on-load ones that compiled with scratch beside the hand-counted W-stream waits: removed in round 6; the Makefile's
// `check-panel-spills` fails the build if any kernel of this file needs scratch.)
#define DGQ_PANEL_CONFIGS(X) X(1, 10, 1) X(1, 5, 1) X(1, 5, 2) X(1, 4, 2)

template <bool PER_M, typename TIO>
int launch_panel_cfg(const GemmBatch& bt, int tm, int nw, int kw, hipStream_t st) {
    const int key = tm * 1000 + nw * 10 + kw;
    switch (key) {
#define X(TM_, NW_, KW_) case (TM_) * 1000 + (NW_) * 10 + (KW_): launch_panel<PER_M, TIO, TM_, NW_, KW_>(bt, st); break;
        DGQ_PANEL_CONFIGS(X)
#undef X
        default: dgq_set_error("dgq_gemm_wxa8: no panel configuration TM=%d NW=%d KW=%d", tm, nw, kw); return DGQ_EINVAL;
    }
    return DGQ_OK;
}

}  // namespace

// does the configuration exist, and how much LDS does a launch with K slices of `tiles` K tiles need (0: no such configuration)
size_t dgq_gemm_panel_lds_bytes(int tm, int nw, int kw, bool per_m, int tiles) {
    const int key = tm * 1000 + nw * 10 + kw;
    switch (key) {
#define X(TM_, NW_, KW_) case (TM_) * 1000 + (NW_) * 10 + (KW_): return (size_t)panel_lds<TM_, NW_, KW_>(per_m, tiles);
        DGQ_PANEL_CONFIGS(X)
#undef X
        default: return 0;
    }
}

int dgq_launch_gemm_panel(const GemmBatch& bt, bool per_m, int y_dtype, int tm, int nw, int kw, hipStream_t st) {
    switch (y_dtype) {
        case DGQ_F32: return per_m ? launch_panel_cfg<true, float>(bt, tm, nw, kw, st) : launch_panel_cfg<false, float>(bt, tm, nw, kw, st);
        case DGQ_F16: return per_m ? launch_panel_cfg<true, __half>(bt, tm, nw, kw, st) : launch_panel_cfg<false, __half>(bt, tm, nw, kw, st);
        case DGQ_BF16: return per_m ? launch_panel_cfg<true, __hip_bfloat16>(bt, tm, nw, kw, st) : launch_panel_cfg<false, __hip_bfloat16>(bt, tm, nw, kw, st);
        default: dgq_set_error("dgq_gemm_wxa8: unknown y dtype %d", y_dtype); return DGQ_EINVAL;
    }
}
