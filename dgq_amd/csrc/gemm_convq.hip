// gemm_convq.hip — a 3x3 convolution layer of the quantized UNet as ONE launch: the activation quantiser of the unfolded operand
// INSIDE the contraction (round 6; BASELINE.json north_star: "activations group-quantized on the fly").
//
// Reference: QuantLayer.forward for a Conv2d under group quantisation (quant/quant_layer.py:626-661): F.unfold (:630-638) ->
// aqtizer on the [B, C·kh·kw, L] operand (:640-641, one (δ, z) per (channel, tap) or per output position) -> weight @ unfolded
// (input_unfolded_pseudo_conv2d, :526-574).  The two-launch form of this library materialises that operand as an int8 code matrix
// [M][Kp] in HBM (quant_act_conv_kernel: 25 MB written and read back for a 64 x 64 x 320 layer, more than its fp32 input) — the largest
// class of a step's traffic that the algorithm does not need ("what's missing" 1).  Here a workgroup owns a 4 x 8 tile of
// output positions (= the 32 rows of one MFMA tile) and ALL N <= 320 output channels (NW column waves):
//   1. the tile's (4 + kh − 1) x (8 + kw − 1) x C input patch goes to LDS once, as fp32, with the folded GroupNorm scale / shift and
//      SiLU applied (zeros outside the image: F.unfold pads before the quantiser sees the operand) — quant_act_conv_kernel's staging;
//   2. K is walked in slabs of `slab` K tiles (the A image of 32 rows x the whole Kp does not fit beside the patch): the waves gather
//      and quantise the slab's codes from the patch through the kpat table straight into the slab's LDS image — the same lane
//      mapping and arithmetic as quant_act_conv_kernel (dgq_affine_code4_fast, quant_common.h), so codes AND row sums are the
//      two-launch form's bit for bit — and
//   3. run the slab's K tiles of gemm_panel_kernel's loop: A fragments from LDS, int4 weights streamed fragment-major into registers
//      (hand-counted waits; the stream runs on across the slab switches), per-K group flushes by summation by parts;
//   4. the dequantising store epilogue of the family (gemm_tile.h, TILED row mapping: residual / temb rows / GroupNorm partials).
// No code matrix, no row-sum vector, one launch instead of two.  LDS (C = 320, Kp = 3072): patch 75 KB + tables 7 KB + slab 40 KB +
// epilogue vectors 6 KB.  Layers it takes (dgq_gemm_conv_act_fuses): 3x3, stride 1, pad 1, W4, N % 32 == 0 and N <= 320, H % 4 == 0,
// W % 8 == 0, patch + tables + a 10-tile slab within 160 KB (C <= 340) — the C = 320 convolutions of the 64 x 64 level of SD.  Layers with
// more input channels (the concatenated up-path inputs, the 32 x 32 level) keep the two-launch form: their patch alone exceeds the LDS.
#include "gemm_tile.h"
#include "quant_common.h"

DGQ_DIAG_BUFFER(convq)

namespace {

template <int N>
__device__ __forceinline__ void cq_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

constexpr int CQ_TH = 4, CQ_TW = 8;                      // output positions per workgroup: one 32-row MFMA tile
__host__ __device__ constexpr int cq_align16(int v) { return (v + 15) & ~15; }

struct ConvqLds {                                        // byte offsets of the LDS regions
    int patch, tab, tdl, rowt, slab, vtab, total;
};
__host__ __device__ inline ConvqLds convq_lds(int C, int kh, int kw, int stride, int Kp, int nw, int slab_tiles, bool per_m) {
    const int PH = (CQ_TH - 1) * stride + kh, PW = (CQ_TW - 1) * stride + kw;
    const int ep = nw * 32 * (32 + 4) * 4;               // gemm_store_tile's transposition scratch: over the (then idle) patch
    const int patch = PH * PW * C * 4;
    ConvqLds l;
    l.patch = 0;
    l.tab = cq_align16(patch > ep ? patch : ep);
    l.tdl = l.tab + cq_align16(Kp * 2);
    l.rowt = l.tdl + cq_align16((Kp >> 5) * 8);
    l.slab = l.rowt + 3 * 32 * 4;
    l.vtab = l.slab + slab_tiles * 32 * BK;
    l.total = l.vtab + (3 * 32 + 4 * 32 * nw) * 4 + (per_m ? 0 : cq_align16((NCH + 1) * (Kp / BK) * 4));
    return l;
}

template <bool PER_M, typename TIO, int NW>
__global__ __launch_bounds__(64 * NW) void gemm_convq_kernel(GemmBatch bt, int slab_tiles) {
    constexpr int BM = 32, BN = 32 * NW, NT = 64 * NW;
    constexpr int ACCS = PER_M ? 1 : 2;
    constexpr int DT = 4, NS = DT + 1;
    const GemmParams& p = bt.p[0];
    gemm_prefetch_params(p);
    DGQ_DIAG_DECL
    DGQ_STAMP(0); DGQ_STAMP_REAL(1); DGQ_STAMP_WHERE(2);
    const dgq_gemm_act_t& act = p.act;
    const int C = act.K, H = act.H, W = act.W, kh = act.kh, kw = act.kw, stride = act.stride, pad = act.pad;
    const int Ho = (H + 2 * pad - kh) / stride + 1, Wo = (W + 2 * pad - kw) / stride + 1;
    const int tiles_w = Wo / CQ_TW, tiles_h = Ho / CQ_TH;
    // XCD-aware tile order: XCD k owns a contiguous range of tiles (neighbours share their halo rows in one L2)
    int tile;
    {
        const int T = gridDim.x, bid = blockIdx.x;
        const int q = T >> 3, r = T & 7, xcd = bid & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int b = tile / (tiles_h * tiles_w);
    const int trem = tile - b * tiles_h * tiles_w;
    const int th = trem / tiles_w, tw = trem - th * tiles_w;
    const int ho0 = th * CQ_TH, wo0 = tw * CQ_TW;
    const int m0 = (b * Ho + ho0) * Wo + wo0;              // the tile's first output position (row r = m0 + (r >> 3)·Wo + (r & 7))
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nk = p.Kp / BK;
    const ConvqLds L = convq_lds(C, kh, kw, stride, p.Kp, NW, slab_tiles, PER_M);

    // ---- W stream (gemm_panel_kernel's): this wave's 32 columns, fragment-major, DT K tiles ahead in NS register slots
    const int ntile32 = (p.N + 31) >> 5;
    const int jt = min(wid, ntile32 - 1);                  // a wave past N recomputes the last column tile and stores nothing
    const uint4* wsrc = reinterpret_cast<const uint4*>(p.wfrag) + ((int64_t)jt * (nk * 2)) * 64 + lane;
    v4i wr[NS][2];
    auto wload = [&](int t, v4i (&dst)[2]) {
        const uint4* q = wsrc + (t * 2) * 64;
        asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:1024"
                     : "=&v"(dst[0]), "=&v"(dst[1]) : "v"(q) : "memory");
    };
#pragma unroll
    for (int d = 0; d < DT; ++d)
        if (d < nk) wload(d, wr[d]);
#pragma unroll
    for (int j = 0; j < 2; ++j) wr[DT][j] = (v4i){0, 0, 0, 0};

    // ---- epilogue vectors and flush coefficients (gemm_panel_kernel's tables, KW = 1)
    float* vtab = reinterpret_cast<float*>(smem + L.vtab);   // [3][32]: R0 R1 R2 | [4][BN]: alpha zw gamma vn
    float* vcol = vtab + 3 * BM;
    float* ctab = vcol + 4 * BN;                              // per-K: [nk·4] flush coefficients | [nk] clear flags
    constexpr int MYCH = NCH;
    const int n_coef = PER_M ? 0 : nk * MYCH, n_tab = PER_M ? 0 : n_coef + nk;
    struct CoefIdx { int g, gn, tl; bool is_coef, seq_last, tile_end, not_last_tile; };
    auto coef_idx = [&](int e) {
        CoefIdx x;
        x.is_coef = e < n_coef;
        const int ec = x.is_coef ? e : 0;
        const int tc = ec / MYCH, ci = ec - tc * MYCH;
        const int t = x.is_coef ? tc : e - n_coef;
        x.g = tc * NCH + ci;
        x.tile_end = (ci + ACCS >= MYCH);
        x.seq_last = x.tile_end && tc == nk - 1;
        x.gn = min(x.tile_end ? (tc + 1) * NCH + (ci + ACCS - MYCH) : x.g + ACCS, nk * NCH - 1);
        x.tl = t * NCH + NCH - 1;
        x.not_last_tile = t != nk - 1;
        return x;
    };
    auto coef_val = [&](const CoefIdx& x, float d, float dn, uint32_t cf) {
        const bool clr = (cf & 0xFF) == 2;
        const float coef = (x.seq_last || (x.tile_end && clr)) ? d : d - dn;
        const float flag = (x.not_last_tile && clr) ? 1.0f : 0.0f;
        return x.is_coef ? coef : flag;
    };
    const bool has_col = tid < BN;
    float c_vn = 0.0f, c_d = 0.0f, c_dn = 0.0f;
    uint32_t c_cf = 0;
    const int ncol = min(tid, p.N - 1);
    float c_al = gload_f32(p.alpha + ncol), c_zw = gload_f32(p.zw + ncol), c_ga = gload_f32(p.gamma + ncol);
    if constexpr (PER_M) c_vn = gload_f32(p.vn + ncol);
    CoefIdx cx = {};
    if constexpr (!PER_M) {
        cx = coef_idx(min(tid, n_tab - 1));
        c_d = gload_f32(p.cdelta + cx.g); c_dn = gload_f32(p.cdelta + cx.gn); c_cf = gload_u8(p.cflush + cx.tl);
    }

    // ---- 1. the input patch (quant_act_conv_kernel's staging: wave w takes pixels w, w + NW, ...)
    // (every load below is an ordinary one: the asm loads above are all OLDER, so hipcc's counted waits for these stay correct)
    float* patch = reinterpret_cast<float*>(smem + L.patch);
    const int PH = (CQ_TH - 1) * stride + kh, PW = (CQ_TW - 1) * stride + kw;
    const int hi0 = ho0 * stride - pad, wi0 = wo0 * stride - pad;
    {
        const TIO* img = reinterpret_cast<const TIO*>(act.x) + (int64_t)b * H * W * act.ldx;
        const float* pre_sc = act.pre_scale ? act.pre_scale + (int64_t)b * C : nullptr;
        const float* pre_sh = act.pre_scale ? act.pre_shift + (int64_t)b * C : nullptr;
        // a wave's pixels in batches of PB: every 16-byte load of a batch is issued before the first is used (one memory latency per
        // batch — pixel by pixel, each load → SiLU → LDS store chain paid its own)
        constexpr int PB = 6;
        const int npx = PH * PW;
        for (int p0 = wid; p0 < npx; p0 += NW * PB) {
            float v[PB][2][4];
            bool inb[PB];
#pragma unroll
            for (int i = 0; i < PB; ++i) {
                const int pp = min(p0 + NW * i, npx - 1);
                const int ph = pp / PW, pw_ = pp - ph * PW;
                const int hi = hi0 + ph, wi = wi0 + pw_;
                inb[i] = hi >= 0 && hi < H && wi >= 0 && wi < W;                 // wave-uniform
                const TIO* src = img + ((int64_t)min(max(hi, 0), H - 1) * W + min(max(wi, 0), W - 1)) * act.ldx;   // (clamped: loaded, then zeroed)
#pragma unroll
                for (int h = 0; h < 2; ++h) load4<TIO>(src + min(lane * 4 + 256 * h, C - 4), v[i][h]);
            }
#pragma unroll
            for (int i = 0; i < PB; ++i) {
                const int pp = p0 + NW * i;
                if (pp >= npx) break;                                            // wave-uniform
                float* dst = patch + pp * C;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int c = lane * 4 + 256 * h;
                    if (c >= C) continue;
                    float o[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                    if (inb[i]) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) o[j] = v[i][h][j];
                        if (pre_sc) {
                            const float4 sc = *reinterpret_cast<const float4*>(pre_sc + c);
                            const float4 sh = *reinterpret_cast<const float4*>(pre_sh + c);
                            o[0] = o[0] * sc.x + sh.x; o[1] = o[1] * sc.y + sh.y; o[2] = o[2] * sc.z + sh.z; o[3] = o[3] * sc.w + sh.w;
                        }
                        if (act.pre_act == 1) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) o[j] = dgq_silu(o[j]);
                        }
                    }
                    *reinterpret_cast<float4*>(dst + c) = make_float4(o[0], o[1], o[2], o[3]);
                }
            }
        }
    }
    DGQ_STAMP(3);                                          // (diagnostic) patch staged
    // the gather table, the chunks' (δ, z) and the rows' (δ, z, 1/δ) into LDS
    uint16_t* tab = reinterpret_cast<uint16_t*>(smem + L.tab);
    float* tdl = reinterpret_cast<float*>(smem + L.tdl);
    float* tzp = tdl + (p.Kp >> 5);
    float* rowt = reinterpret_cast<float*>(smem + L.rowt);     // per-M: [3][32] δ, z, 1/δ of the tile's rows
    for (int k = tid * 4; k < p.Kp; k += 4 * NT) {
        const int4 e = *reinterpret_cast<const int4*>(act.kpat + k);
        *reinterpret_cast<uint2*>(tab + k) = make_uint2(((uint32_t)e.x & 0xFFFFu) | ((uint32_t)e.y << 16), ((uint32_t)e.z & 0xFFFFu) | ((uint32_t)e.w << 16));
    }
    if constexpr (!PER_M) {
        for (int c = tid; c < (p.Kp >> 5); c += NT) { tdl[c] = p.cdelta[c]; tzp[c] = act.czp[c]; }
    } else if (tid < 32) {
        const int li = (m0 + (tid >> 3) * Wo + (tid & 7)) % p.L;
        const float md = p.mdelta[li];
        rowt[tid] = md; rowt[32 + tid] = p.mzp[li]; rowt[64 + tid] = dgq_rcp(md);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the first W tiles and the tables
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
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // (long K only: ordinary loads of the loop above)
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    DGQ_STAMP(4);                                          // (diagnostic) tables staged, everyone's patch rows visible

    // ---- 2 + 3. slabs of the K range: quantise into the slab image, then its K tiles
    uint8_t* slab = smem + L.slab;
    const int lr = lane & 31, hh = lane >> 5;
    int a_off[NCH];
#pragma unroll
    for (int cg = 0; cg < NCH; ++cg) a_off[cg] = lr * BK + (((2 * cg + hh) ^ ((lr >> 1) & 7)) << 4);
    v16i acc[ACCS][1][1];
    v16f accf[1][1];
    constexpr bool BIASED = !PER_M;
    constexpr int ACC0 = BIASED ? DGQ_ACC_BIAS_I : 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
        for (int a = 0; a < ACCS; ++a) acc[a][0][0][r] = ACC0;
        accf[0][0][r] = 0.0f;
    }
    auto flush = [&](const v16i (&ac)[1][1], float coef) {
        const float sc = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, coef)));
        if (sc != 0.0f) {
#pragma unroll
            for (int r = 0; r < 16; ++r) accf[0][0][r] = __builtin_fmaf(sc, dgq_total_to_float<BIASED>(ac[0][0][r]), accf[0][0][r]);
        }
    };
    float pend = 0.0f;
    typedef float cvec_t __attribute__((ext_vector_type(NCH)));
    const float* tclr = ctab + nk * MYCH;
    auto tile_fn = [&](int t, int ts, const v4i& w0, const v4i& w1) {      // t: K tile, ts: its index inside the slab image
        cvec_t cq;
        float tc = 0.0f;
        if (!PER_M) {
            cq = *reinterpret_cast<const cvec_t*>(ctab + t * MYCH);
            tc = tclr[t];
        }
        const uint8_t* sa = slab + ts * (BM * BK);
#pragma unroll
        for (int ci = 0; ci < NCH; ++ci) {
            const v4i af = *reinterpret_cast<const v4i*>(sa + a_off[ci]);
            const v4i& w = ci < 2 ? w0 : w1;
            const uint32_t x = (uint32_t)((ci & 1) ? w[2] : w[0]), y = (uint32_t)((ci & 1) ? w[3] : w[1]);
            const v4i bf = (v4i){(int)(x & 0x0F0F0F0Fu), (int)((x >> 4) & 0x0F0F0F0Fu), (int)(y & 0x0F0F0F0Fu), (int)((y >> 4) & 0x0F0F0F0Fu)};
            if constexpr (PER_M) {
                acc[0][0][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af, bf, acc[0][0][0], 0, 0, 0);
            } else {
                if (ci & 1) {
                    acc[1][0][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af, bf, acc[1][0][0], 0, 0, 0);
                    flush(acc[0], pend);
                } else {
                    acc[0][0][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af, bf, acc[0][0][0], 0, 0, 0);
                    flush(acc[1], pend);
                }
                pend = cq[ci];
            }
        }
        if constexpr (!PER_M) {
            if (__builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, tc)) != 0) {     // rare: a segment of running totals ends
                flush(acc[1], pend);
                pend = 0.0f;
#pragma unroll
                for (int a = 0; a < ACCS; ++a)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[a][0][0][r] = ACC0;
            }
        }
    };
    // the rows this wave quantises: wid, wid + NW, ... < 32; their running row sums live across the slabs
    constexpr int RQ = (32 + NW - 1) / NW;
    float partial[RQ];
#pragma unroll
    for (int q = 0; q < RQ; ++q) partial[q] = 0.0f;
    const float qmax = (float)((1 << act.bits) - 1), bias = 128.0f - p.offset;
    auto quantise_slab = [&](int k0, int k1) {
#pragma unroll
        for (int q = 0; q < RQ; ++q) {
            const int r = wid + NW * q;
            if (r >= 32) continue;                           // wave-uniform
            const int i = r >> 3, j = r & 7;
            const float* pr = patch + ((i * stride) * PW + j * stride) * C;
            float md = 1.0f, mz = 0.0f, minv = 1.0f;
            if (PER_M) { md = rowt[r]; mz = rowt[32 + r]; minv = rowt[64 + r]; }
            const uint32_t row_base = (uint32_t)(r * BK), row_swz = (uint32_t)(((r >> 1) & 7) << 4);
            float part = partial[q];
            // QU = 5 steps of 256 codes per round: a 10-tile slab (1280 codes) is ONE round — every table read and every gather of the row's
            // slab in flight together.  (The lane -> code mapping and each lane's ascending order are quant_act_conv_kernel's, whose rounds
            // are 4 steps: the row sums add up in the same order.)
            constexpr int QU = 5;
            // (every condition of the round is wave-uniform — a step of 256 codes lies inside the slab as a whole, but for the upper half
            // of the last step when Kp is an odd multiple of 128: those lanes compute on a clamped position and write nothing — so the
            // compiler emits scalar branches, not exec-mask regions)
            for (int kbu = k0; kbu < k1; kbu += 256 * QU) {   // wave-uniform round base
                int idx[QU][4];
#pragma unroll
                for (int u = 0; u < QU; ++u) {
                    const int kpc = min(kbu + 256 * u + lane * 4, p.Kp - 4);          // (clamped: a step past the slab reads, computes, discards)
                    const uint2 tt = *reinterpret_cast<const uint2*>(tab + kpc);
                    idx[u][0] = tt.x & 0xFFFF; idx[u][1] = tt.x >> 16; idx[u][2] = tt.y & 0xFFFF; idx[u][3] = tt.y >> 16;
                }
                float v[QU][4];
#pragma unroll
                for (int u = 0; u < QU; ++u)
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[u][e] = pr[idx[u][e] == 0xFFFF ? 0 : idx[u][e]];   // padding reads element 0: value unused
#pragma unroll
                for (int u = 0; u < QU; ++u) {
                    const int ku = kbu + 256 * u;             // wave-uniform
                    if (ku >= k1) break;
                    const int kp0 = ku + lane * 4;
                    const bool in = kp0 < k1;                 // false only in the upper half-wave of a last half step
                    const int kpc = min(kp0, p.Kp - 4);
                    float d = md, z = mz, inv = minv;
                    if (!PER_M) {
                        d = tdl[kpc >> 5];
                        z = tzp[kpc >> 5];
                        inv = dgq_rcp(d);
                    }
                    float biased[4], qv[4], fsum = 0.0f;
                    dgq_affine_code4_fast(v[u], d, inv, z, qmax, qv);
#pragma unroll
                    for (int e = 0; e < 4; ++e) biased[e] = idx[u][e] != 0xFFFF ? qv[e] + bias : 128.0f;
                    const uint32_t w = dgq_pack4(biased, fsum);
                    const int ks = kpc - k0;                  // position inside the slab image
                    if (in) *reinterpret_cast<uint32_t*>(slab + (ks >> 7) * (BM * BK) + row_base + ((((uint32_t)ks & 127u) & ~15u) ^ row_swz) + (ks & 15)) = w;
                    fsum -= 512.0f;
                    part += in ? (PER_M ? fsum : d * fsum) : 0.0f;
                }
            }
            partial[q] = part;
        }
    };
    for (int tb = 0; tb < nk; tb += NS) {
        if (tb % slab_tiles == 0) {                          // (slab_tiles is a multiple of NS: a slab starts at sl == 0 only)
            DGQ_STAMP_NOW(dg_s0);
            if (tb > 0) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's reads of the previous slab are complete ...
                __builtin_amdgcn_s_barrier();                        // ... everyone's
            }
            DGQ_STAMP_ACC(12, dg_s0);                                // (diagnostic) waiting for the slab image to be free
            DGQ_STAMP_NOW(dg_s1);
            quantise_slab(tb * BK, min(nk, tb + slab_tiles) * BK);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            DGQ_STAMP_ACC(7, dg_s1);                                 // (diagnostic) this wave's rows of the slab
            DGQ_STAMP_NOW(dg_s2);
            __builtin_amdgcn_s_barrier();
            DGQ_STAMP_ACC(13, dg_s2);                                // (diagnostic) waiting for the other waves' rows
        }
        const int ts0 = tb % slab_tiles;
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) {
            const int t = tb + sl;
            if (t < nk) {                                     // wave-uniform
                if (t + DT < nk) wload(t + DT, wr[(sl + DT) % NS]);
                const int young = min(DT, nk - 1 - t);
                if (young >= 4) cq_wait_vmcnt<8>();
                else if (young == 3) cq_wait_vmcnt<6>();
                else if (young == 2) cq_wait_vmcnt<4>();
                else if (young == 1) cq_wait_vmcnt<2>();
                else cq_wait_vmcnt<0>();
                asm volatile("" : "+v"(wr[sl][0]), "+v"(wr[sl][1]));       // the slot's registers are defined HERE for the compiler
                __builtin_amdgcn_sched_barrier(0);
                tile_fn(t, ts0 + sl, wr[sl][0], wr[sl][1]);
            }
        }
    }
    if constexpr (!PER_M) flush(acc[1], pend);
    DGQ_STAMP(6);
    // the rows' epilogue constants R0 R1 R2 (row sums complete: every slab has been quantised)
#pragma unroll
    for (int q = 0; q < RQ; ++q) {
        const int r = wid + NW * q;
        float part = partial[q];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) part += __shfl_down(part, o, 64);
        if (r < 32 && lane == 0) {
            float r0v = 1.0f, r1v = part, r2v = 0.0f;
            if (PER_M) { const float md = rowt[r]; r0v = md; r1v = md * part; r2v = md * (p.offset - rowt[32 + r]); }
            vtab[r] = r0v; vtab[BM + r] = r1v; vtab[2 * BM + r] = r2v;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // this wave's slab reads are complete, its vtab rows written ...
    __builtin_amdgcn_s_barrier();                            // ... everyone's: the patch region becomes the epilogue's scratch
    gemm_store_tile<PER_M, TIO, BM, BN, 1, NW, 1, NW * 32 * (32 + 4) * 4, 1, 1, true>(p, 0, smem + L.patch, vtab, vcol, wid, lane, 0, wid, 0, m0, 0,
                                                                                      acc[0], accf DGQ_DIAG_ARG, Wo, tile * 2);
    DGQ_STAMP(9);
    DGQ_DIAG_DRAIN();
    DGQ_STAMP(10); DGQ_STAMP_REAL(11);
    DGQ_DIAG_FLUSH(convq, NW, wid, lane);
}

template <bool PER_M, typename TIO, int NW>
int launch_convq(const GemmBatch& bt, int slab_tiles, int lds, hipStream_t st) {
    const dgq_gemm_act_t& a = bt.p[0].act;
    const int Ho = (a.H + 2 * a.pad - a.kh) / a.stride + 1, Wo = (a.W + 2 * a.pad - a.kw) / a.stride + 1;
    static std::atomic<bool> attr_set[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !attr_set[dev].load(std::memory_order_acquire)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_convq_kernel<PER_M, TIO, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (dev >= 0 && dev < 64) attr_set[dev].store(true, std::memory_order_release);
    }
    const dim3 grid(a.B * (Ho / CQ_TH) * (Wo / CQ_TW)), block(64 * NW);
    hipLaunchKernelGGL((gemm_convq_kernel<PER_M, TIO, NW>), grid, block, lds, st, bt, slab_tiles);
    return DGQ_OK;
}

}  // namespace

// The slab length (K tiles, a multiple of 5 = the weight ring's slots) and LDS bytes of a convolution in this form, or 0 where it does
// not take the layer.  Shapes only; the caller checks pointers and dtypes.
int dgq_gemm_convq_plan(int B, int H, int W, int C, int kh, int kw, int stride, int pad, int N, int Kp, int w_bits, bool per_m, int* lds_bytes) {
    if (kh != 3 || kw != 3 || stride != 1 || pad != 1 || w_bits != 4 || B < 1 || C < 4 || C % 4 != 0 || C > 512 || H % CQ_TH != 0 || W % CQ_TW != 0 ||
        (N != 160 && N != 320) || Kp % BK != 0 || Kp < C * kh * kw)
        return 0;
    const int nk = Kp / BK, nw = N / 32;
    if ((CQ_TH + kh - 1) * (CQ_TW + kw - 1) * C >= 0xFFFF) return 0;      // 16-bit patch indices
    // a slab is a multiple of 5 K tiles (the weight ring's slots: a slab starts at slot 0) and — unless it is the only one — of 2 (its first
    // code a multiple of 256: every lane keeps the code positions quant_act_conv_kernel gives it, so the row sums add up in that order)
    const int one = (nk + 4) / 5 * 5;
    const int cand[3] = {one <= 20 ? one : 20, 20, 10};
    for (int i = 0; i < 3; ++i) {
        const int s = cand[i] < one ? cand[i] : one;
        const ConvqLds l = convq_lds(C, kh, kw, stride, Kp, nw, s, per_m);
        if (l.total <= 160 * 1024 - 64) {
            if (lds_bytes) *lds_bytes = l.total;
            return s;
        }
    }
    return 0;
}

int dgq_launch_gemm_convq(const GemmBatch& bt, bool per_m, int y_dtype, int slab_tiles, int lds, hipStream_t st) {
    const int nw = bt.p[0].N / 32;
#define CQ_NW(PM, T) (nw == 10 ? launch_convq<PM, T, 10>(bt, slab_tiles, lds, st) : launch_convq<PM, T, 5>(bt, slab_tiles, lds, st))
#define CQ_PM(T) (per_m ? CQ_NW(true, T) : CQ_NW(false, T))
    switch (y_dtype) {
        case DGQ_F32: return CQ_PM(float);
        case DGQ_F16: return CQ_PM(__half);
        case DGQ_BF16: return CQ_PM(__hip_bfloat16);
        default: dgq_set_error("dgq_gemm_wxa8: unknown y dtype %d", y_dtype); return DGQ_EINVAL;
    }
#undef CQ_PM
#undef CQ_NW
}
