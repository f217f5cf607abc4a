// Single-launch form of the fused quantised-softmax attention for SHORT key ranges (S <= 256: every cross-attention over the 77
// text tokens, the 16x16 / 8x8 self-attentions of SD) — the attention core of Attention.Attention_forward, diffusers_rewrite/sd.py:183-201,
// with the T2ILogQuantizer real-time δ of quant/quant_layer_text.py:96-105 — behind the pre-pass launch of attn_bf16x3.hip.
//
// Why: those calls are 1-2 us of arithmetic behind dependent launches (pre-pass -> statistics -> [merge] -> P̂·V [-> add]) of
// 8-18 us each (profiles/r05_attention_shapes.txt: 24-48 us per call, 27 of the 32 attentions of an SD step).  With at most NTM = 8
// key tiles the score tiles of a wave's 32 queries fit its registers (16 per tile), so ONE kernel runs
//   phase 1  = the statistics loop of attn3_stats_kernel, keeping every S^T tile;
//   exchange = (real-time δ only) every workgroup publishes its maximum as one 8-byte {tag, value} granule (a write-through sc1 store)
//              and one wave per workgroup sweeps the grid's granules with agent-scope loads until every tag is set — the max over
//              <= 1024 co-resident workgroups, not a grid barrier (cdna_hip_programming.md Guideline 16, R2: the data is the flag; the
//              words are only ever touched by sc1 stores / loads inside the launch; the pre-pass zeroes them);
//   phase 2  = the P̂·V loop and store epilogue of attn3_pv_kernel on the kept tiles (no second Q·K^T, no statistics round trip).
// Per-row arithmetic, tile order and the δ maximum are those of the three-launch form without a key split: outputs are equal bit for
// bit (tests/test_gpu_kernels.py::test_attention_one_launch_is_bit_identical).  The static-δ modes (2, 3) need no exchange.
// Residency: the sweep requires every workgroup of the grid to be resident — the host admits the form for mode 1 only when the grid
// is within the occupancy query's capacity less a margin (MI355X_MICROARCH.md, "Residency and cooperative launch"), else the three
// launches run; the sweep is bounded by the 100 MHz clock and counts a give-up in dgq_attn_sync_timeouts_dev (read by
// dgq_attention_sync_timeouts(); never expected to be non-zero).
// Measured (profiles/r06_attention_one_launch.txt, hipGraph replay per call, B·H = 16): static δ −3.5 … −6 us per call; real-time δ
// −1 us at 512 workgroups (4096 x 77), −2 … −4 us below; a first exchange by atomicMax + an arrival counter that every workgroup polled
// cost +20 us at 512 workgroups.  Also built and measured there, NOT kept: the pre-pass inside the same kernel (every workgroup
// quantising its own queries into fragments and building the K / V tile images of its (batch, head) in LDS: one launch per
// cross-attention) — bit-identical, and slower: 30.5 us against 27.8 at D = 40, 42-59 against 23-28 at D = 160; a workgroup walks six
// tile images one after the other where the pre-pass launch spreads them over 96 workgroups.
#include "attn_bf16x3_dev.h"

#define NTM 8                                   // key tiles kept in registers (S <= 256)
#define ONE_TIMEOUT_TICKS 5000000ull            // 50 ms of s_memrealtime (100 MHz)

__device__ unsigned dgq_attn_sync_timeouts_dev;
namespace {

// one S^T tile's contribution to the running softmax statistics of this lane's query (attn3_stats_kernel's loop body)
__device__ __forceinline__ void stats_tile(v16f acc, int s0, int S, int skip, int h32, float sl2, float& mraw, float& l, float& m2raw) {
    const bool edge = (s0 + KT > S) || (s0 < skip);
    float tmax = -INFINITY, tmax2 = -INFINITY;
    if (edge) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int s = s0 + key_of(r, h32);
            if (s >= S) acc[r] = -INFINITY;
            tmax = fmaxf(tmax, acc[r]);
            if (s >= skip) tmax2 = fmaxf(tmax2, acc[r]);
        }
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, acc[r]);
        tmax2 = tmax;
    }
    const float mn = fmaxf(mraw, tmax);
    const float nb = (mn == -INFINITY) ? 0.0f : -(mn * sl2);
    f2 part = {0.0f, 0.0f};
    const f2 sl2v = {sl2, sl2}, nbv = {nb, nb};
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        const f2 a = pk_fma(f2{acc[r], acc[r + 1]}, sl2v, nbv);
        part = pk_add(part, f2{__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)});
    }
    l = l * __builtin_amdgcn_exp2f(fmaf(mraw, sl2, nb)) + (part.x + part.y);
    mraw = mn;
    m2raw = fmaxf(m2raw, tmax2);
}

struct PvCtx { float sl2, nsl2, m, a0, inv_l, delta, qmax; int cmax_i, S, skip; };

// one kept S^T tile through the softmax quantiser and into O^T += V^T·P̂^T (attn3_pv_kernel's loop body); vtc: the tile's V image in LDS
template <int D, bool UNIFORM, int QM, bool VINT>
__device__ __forceinline__ void pv_tile(v16f acc, const unsigned short* vtc, int s0, const PvCtx& cx, v16f (&oacc)[Geo<D, QM, VINT>::NDT],
                                        float& p_bypass, float& psum, int lane, int h32) {
    using G = Geo<D, QM, VINT>;
    constexpr float MAGIC = 12582912.0f;
    constexpr int MAGIC_I = 0x4B400000;
    constexpr int VPL = G::DV * G::VLD;
    const bool edge = (s0 + KT > cx.S) || (s0 < cx.skip);
    auto quantise = [&](auto edge_tag) {
        constexpr bool EDGE = decltype(edge_tag)::value;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float ph;
            if (UNIFORM) {
                const float pr = exp2f(fmaf(acc[r], cx.sl2, -cx.m)) * cx.inv_l;
                ph = fminf(fmaxf(rintf(__fdiv_rn(pr, cx.delta)), 0.0f), cx.qmax);
            } else {
                const float x = fmaf(acc[r], cx.nsl2, cx.a0);
                int ci = __float_as_int(x + MAGIC);
                ci = min(max(ci, MAGIC_I), cx.cmax_i);
                ph = __int_as_float(0x3F800000 - (ci << 23));
            }
            if (EDGE) {
                const int s = s0 + key_of(r, h32);
                if (s >= cx.S) ph = 0.0f;
                else if (s < cx.skip) {
                    p_bypass = exp2f(fmaf(acc[r], cx.sl2, -cx.m)) * cx.inv_l;
                    ph = 0.0f;
                }
            }
            acc[r] = ph;
            if constexpr (VINT && !G::VONES) psum += ph;
        }
    };
    bf16x8 pf[2];
    if (UNIFORM && !edge) {
        const int cmaxu_i = MAGIC_I + (int)cx.qmax;
        const f2 sl2v = {cx.sl2, cx.sl2}, na0v = {-cx.a0, -cx.a0}, magic = {MAGIC, MAGIC}, nmagic = {-MAGIC, -MAGIC};
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
        int ci[16];
        const f2 nsl2v = {cx.nsl2, cx.nsl2}, a0m = {cx.a0, cx.a0}, magic = {MAGIC, MAGIC};
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const f2 y = pk_add(pk_fma(f2{acc[r], acc[r + 1]}, nsl2v, a0m), magic);
            ci[r] = min(max(__float_as_int(y.x), MAGIC_I), cx.cmax_i);
            ci[r + 1] = min(max(__float_as_int(y.y), MAGIC_I), cx.cmax_i);
            if constexpr (VINT && !G::VONES) psum += __int_as_float(0x3F800000 - (ci[r] << 23)) + __int_as_float(0x3F800000 - (ci[r + 1] << 23));
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            unsigned w[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned pair = __builtin_amdgcn_perm((unsigned)ci[8 * ks + 2 * e + 1], (unsigned)ci[8 * ks + 2 * e], 0x05040100u);
                w[e] = (unsigned)(__mul24((int)pair, -128) + 0x3F803F80);
            }
            pf[ks] = __builtin_bit_cast(bf16x8, make_uint4(w[0], w[1], w[2], w[3]));
        }
    } else {
        if (edge) quantise(std::true_type{});
        else quantise(std::false_type{});
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#define PK(e) ((__float_as_uint(acc[8 * ks + 2 * (e)]) >> 16) | (__float_as_uint(acc[8 * ks + 2 * (e) + 1]) & 0xFFFF0000u))
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

// merge of the two half-waves of a query, the statistics pass's stored (m, l), and this wave's candidate for the real-time δ
__device__ __forceinline__ void finish_stats(float sl2, float cq, float& mraw, float& l, float& m2raw) {
    const float mo = __shfl_xor(mraw, 32, 64), lo = __shfl_xor(l, 32, 64);
    const float mm = fmaxf(mraw, mo), nb = -(mm * sl2);
    l = l * __builtin_amdgcn_exp2f(fmaf(mraw, sl2, nb)) + lo * __builtin_amdgcn_exp2f(fmaf(mo, sl2, nb));
    mraw = mm;
    m2raw = fmaxf(m2raw, __shfl_xor(m2raw, 32, 64));
    mraw += cq;
    m2raw += cq;
}

// wave 0 sweeps the grid's granules until every one carries `tag`; returns the maximum of their values in every lane
__device__ __forceinline__ float sweep_granules(const unsigned long long* gran, int nwg, unsigned tag, int lane) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    float dmax = 0.0f;
    for (;;) {
        bool ok = true;
        dmax = 0.0f;
        for (int g = lane; g < nwg; g += 64) {
            const unsigned long long x = __hip_atomic_load(gran + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = ok && (unsigned)(x >> 32) == tag;
            dmax = fmaxf(dmax, __uint_as_float((unsigned)x));
        }
        if (__all(ok)) break;
        __builtin_amdgcn_s_sleep(1);
        if (__builtin_amdgcn_s_memrealtime() - t0 > ONE_TIMEOUT_TICKS) {   // never expected: a non-resident workgroup
            if (lane == 0) atomicAdd(&dgq_attn_sync_timeouts_dev, 1u);
            break;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dmax = fmaxf(dmax, __shfl_xor(dmax, o, 64));
    return dmax;
}

// the store epilogue of attn3_pv_kernel (O^T tiles through the idle LDS, 16 bytes per lane in 128-byte row segments); lds8: free LDS of
// at least 3·DV floats + 4 x 32 x 36 floats
template <int D, int QM, bool VINT>
__device__ __forceinline__ void store_epilogue(const AttnParams& p, unsigned char* lds8, v16f (&oacc)[Geo<D, QM, VINT>::NDT], float p_bypass,
                                               float psum, float delta, int b, int hd, int bx, int tid, int lane, int wid, int h32) {
    using G = Geo<D, QM, VINT>;
    constexpr int NW = 4;
    if (p.skip > 0) p_bypass = __shfl(p_bypass, lane & 31, 64);
    if constexpr (G::VONES) {
        constexpr int kd = D % 32, rr = (kd & 3) + 4 * (kd >> 3), hh = (kd >> 2) & 1;
        psum = __shfl(oacc[D / 32][rr], (lane & 31) + 32 * hh, 64);
    } else if (VINT) {
        psum += __shfl_xor(psum, 32, 64);
    }
    // ---------------------------------------------------------------- store epilogue (attn3_pv_kernel's, through the idle ring)
    float* vtab = reinterpret_cast<float*>(lds8);
    __syncthreads();
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
    {
        constexpr int ELD = 36;
        float* ep = reinterpret_cast<float*>(lds8) + 3 * G::DV + wid * (32 * ELD);
        const int tw0 = bx * (32 * NW) + wid * 32;
        const int er = lane >> 3, ec = (lane & 7) * 4;
        const float pb = p.skip > 0 ? p_bypass : 0.0f;
#pragma unroll
        for (int j = 0; j < G::NDT; ++j) {
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
                const float a0v[4] = {t0[k4].x, t0[k4].y, t0[k4].z, t0[k4].w}, a1v[4] = {t1[k4].x, t1[k4].y, t1[k4].z, t1[k4].w};
                const float a2v[4] = {t2[k4].x, t2[k4].y, t2[k4].z, t2[k4].w};
                float o4[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * k4 + e;
                    float o;
                    if (VINT) o = delta * (a0v[e] * (oacc[j][r] - a1v[e] * psum));
                    else o = delta * oacc[j][r];
                    o4[e] = o + pb * a2v[e];
                }
                *reinterpret_cast<float4*>(ep + (lane & 31) * ELD + 8 * k4 + 4 * h32) = make_float4(o4[0], o4[1], o4[2], o4[3]);
            }
            float4 ov[4];
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) ov[ps] = *reinterpret_cast<const float4*>(ep + (er + 8 * ps) * ELD + ec);
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                const int row = er + 8 * ps, tt = tw0 + row, d0 = j * 32 + ec;
                const float4 v = ov[ps];
                if (tt < p.T && d0 < D) {
                    const int64_t oi = ((int64_t)(b * p.T + tt) * p.H + hd) * D + d0;
                    if (p.io_dtype == DGQ_F32) {
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
        }
    }
}


}  // namespace

// ===================================================================================================== behind the pre-pass (ring staging)
template <int D, bool UNIFORM, int QM, bool VINT>
__global__ __launch_bounds__(256) void attn3_one_kernel(AttnParams p, unsigned long long* __restrict__ gran, int nwg) {
    using G = Geo<D, QM, VINT>;
    constexpr bool QI8 = G::QI8, KS = QM == 3;
    constexpr int NW = 4;
    constexpr int ST1 = G::STATS_STAGES, NP1 = G::K_PIECES, TB1 = NP1 * 1024;
    constexpr int ST2 = G::PV_STAGES, NP2 = G::IMG_PIECES, TB2 = G::IMG_BYTES;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds8[];
    __shared__ float wmax_s[NW];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, h32 = lane >> 5;
    int bx, bh;
    attn_block_coords(p.xcd, bx, bh);
    const int b = bh / p.H, hd = bh - b * p.H;
    const int t = bx * (32 * NW) + wid * 32 + (lane & 31);
    const int tq = min(t, p.T - 1);
    const int NT = p.NT;
    const unsigned char* img_lane = p.planes + (int64_t)bh * NT * G::IMG_BYTES + lane * 16;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(lptr_t)lds8;

    // ---------------------------------------------------------------- phase 1: statistics (attn3_stats_kernel<D, 4, QM, 1>)
#pragma unroll
    for (int i = 0; i < ST1 - 1; ++i)
        issue_image<NP1, NW>(img_lane + (int64_t)min(i, NT - 1) * G::IMG_BYTES, lds_base + i * TB1, wid);
    bf16x8 qf[3][QI8 ? 1 : G::NKK];
    v4i qc[G::NK32];
    float4 qt = make_float4(1.0f, 0.0f, 0.0f, 0.0f);
    if constexpr (QI8) load_q_i8<D>(qc, qt, p, (int64_t)(b * p.T + tq) * p.H + hd, h32);
    else if constexpr (QM == 2) load_q1<D>(qf, reinterpret_cast<const unsigned short*>(p.q) + ((int64_t)(b * p.T + tq) * p.H + hd) * D, h32);
    else load_q<D>(qf, p.q + ((int64_t)(b * p.T + tq) * p.H + hd) * D, h32, p.fq[0], tq);
    if constexpr (QM == 2) {
        const float2 t2 = *reinterpret_cast<const float2*>(p.qtab + ((int64_t)(b * p.T + tq) * p.H + hd) * 2);
        qt.x = t2.x;
        qt.y = t2.y;
        if constexpr (G::FOLDZ) fold_zmul<D>(qf, qt.y, h32);
    }
    float inv_dk = 1.0f, cq = 0.0f, dk = 1.0f;
    if constexpr (KS) {
        dk = p.fq[1].delta[0];
        inv_dk = 1.0f / dk;
        cq = -(p.fq[1].zp[0] - 0.5f * (p.fq[1].qmax + 1.0f)) * qt.z;
    }
    const float sl2 = p.scale * LOG2E * qt.x * dk;
    float mraw = -INFINITY, l = 0.0f, m2raw = -INFINITY;
    wait_image<NP1, ST1 - 2, NW>(wid);
    __builtin_amdgcn_s_barrier();
    v16f sc[NTM];
    {
        int stage = 0, istage = ST1 - 1;
#pragma unroll
        for (int i = 0; i < NTM; ++i) {
            if (i < NT) {                                    // block-uniform
                issue_image<NP1, NW>(img_lane + (int64_t)min(i + ST1 - 1, NT - 1) * G::IMG_BYTES, lds_base + istage * TB1, wid);
                const unsigned char* tile_lds = lds8 + stage * TB1;
                v16f acc;
                if constexpr (QI8) acc = score_tile_i8<D, KS>(tile_lds, qc, qt, lane, i == 0 && p.kskip > 0, inv_dk, cq);
                else if constexpr (QM == 2) acc = score_tile_q1<D>(reinterpret_cast<const unsigned short*>(tile_lds), qf, qt.y, lane);
                else acc = score_tile<D>(reinterpret_cast<const unsigned short*>(tile_lds), qf, lane);
                sc[i] = acc;                                 // kept as computed: phase 2 masks the edge tiles itself
                stats_tile(acc, i * KT, p.S, p.skip, h32, sl2, mraw, l, m2raw);
                wait_image<NP1, ST1 - 2, NW>(wid);
                __builtin_amdgcn_s_barrier();
                stage = (stage + 1 == ST1) ? 0 : stage + 1;
                istage = (istage + 1 == ST1) ? 0 : istage + 1;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's phase-1 DMA has landed ...
    __builtin_amdgcn_s_barrier();                         // ... everyone's: the ring may be refilled
    // ---------------------------------------------------------------- phase 2 prologue: the first image stages fly during the exchange
#pragma unroll
    for (int i = 0; i < ST2 - 1; ++i)
        issue_image<NP2, NW>(img_lane + (int64_t)min(i, NT - 1) * G::IMG_BYTES, lds_base + i * TB2, wid);
    finish_stats(sl2, cq, mraw, l, m2raw);
    const float m_st = mraw * sl2;                        // what the statistics pass stores as m
    float delta;
    if (p.mode == 1) {
        float pm = (t < p.T) ? exp2f(m2raw * sl2 - m_st) / l : 0.0f;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) pm = fmaxf(pm, __shfl_xor(pm, o, 64));
        if (lane == 0) wmax_s[wid] = pm;
        __syncthreads();
        // (512 arrivals on one atomic counter plus 512 pollers of it cost 20 us: profiles/r06_attention_one_launch.txt; the granule sweep 3-4)
        if (tid == 0) {
#pragma unroll
            for (int w = 1; w < NW; ++w) pm = fmaxf(pm, wmax_s[w]);
            __hip_atomic_store(gran + (blockIdx.x + gridDim.x * blockIdx.y), (1ull << 32) | (unsigned long long)__float_as_uint(pm),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (wid == 0) {
            const float dmax = sweep_granules(gran, nwg, 1u, lane);
            if (lane == 0) wmax_s[0] = dmax;
        }
        __syncthreads();
        delta = wmax_s[0];
    } else {
        delta = p.delta[0];
    }
    // ---------------------------------------------------------------- phase 2: P̂·V (attn3_pv_kernel<D, UNIFORM, 4, QM, VINT, 1>)
    PvCtx cx;
    cx.sl2 = sl2; cx.nsl2 = -sl2; cx.m = m_st - cq * sl2; cx.a0 = cx.m + log2f(l) + log2f(delta); cx.inv_l = 1.0f / l; cx.delta = delta;
    cx.qmax = p.qmax; cx.cmax_i = 0x4B400000 + min((int)p.qmax, 127); cx.S = p.S; cx.skip = p.skip;
    float p_bypass = 0.0f;
    float psum = 0.0f;
    v16f oacc[G::NDT];
#pragma unroll
    for (int j = 0; j < G::NDT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[j][r] = 0.0f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    {
        int stage = 0, istage = ST2 - 1;
#pragma unroll
        for (int i = 0; i < NTM; ++i) {
            if (i < NT) {
                issue_image<NP2, NW>(img_lane + (int64_t)min(i + ST2 - 1, NT - 1) * G::IMG_BYTES, lds_base + istage * TB2, wid);
                const unsigned short* vtc = reinterpret_cast<const unsigned short*>(lds8 + stage * TB2) + G::K_ELEMS;
                pv_tile<D, UNIFORM, QM, VINT>(sc[i], vtc, i * KT, cx, oacc, p_bypass, psum, lane, h32);
                wait_image<NP2, ST2 - 2, NW>(wid);
                __builtin_amdgcn_s_barrier();
                stage = (stage + 1 == ST2) ? 0 : stage + 1;
                istage = (istage + 1 == ST2) ? 0 : istage + 1;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // no DMA may outlive the block's LDS allocation
    store_epilogue<D, QM, VINT>(p, lds8, oacc, p_bypass, psum, delta, b, hd, bx, tid, lane, wid, h32);
}

namespace {

// per device and kernel: the dynamic-LDS opt-in and how many workgroups are resident at once (0: not known yet)
static int resident_capacity(const void* fn, int lds, std::atomic<int>& slot) {
    int cap = slot.load(std::memory_order_acquire);
    if (cap != 0) return cap;
    int dev = 0, per_cu = 0, cus = 0;
    (void)hipGetDevice(&dev);
    (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, lds) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || per_cu < 1 || cus < 1)
        return -1;
    // the occupancy query can be one workgroup per CU high where registers, not LDS, bind (MI355X_MICROARCH.md): keep one in hand
    if (per_cu > 2) per_cu -= 1;
    if (per_cu > 4) per_cu = 4;
    cap = per_cu * cus;
    slot.store(cap, std::memory_order_release);
    return cap;
}

template <int D, bool UNIFORM, int QM, bool VINT>
int launch_one(const AttnParams& p, unsigned long long* gran, hipStream_t st) {
    using G = Geo<D, QM, VINT>;
    constexpr int ring1 = G::STATS_STAGES * G::K_PIECES * 1024, ring2 = G::PV_STAGES * G::IMG_BYTES;
    constexpr int scratch = 3 * G::DV * 4 + 4 * 32 * 36 * 4;
    constexpr int lds = ring1 > ring2 ? (ring1 > scratch ? ring1 : scratch) : (ring2 > scratch ? ring2 : scratch);
    static_assert(lds + 64 <= 160 * 1024, "LDS ring too large");
    static std::atomic<int> capacity[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64) return 1;
    const int cap = resident_capacity(reinterpret_cast<const void*>(&attn3_one_kernel<D, UNIFORM, QM, VINT>), lds, capacity[dev]);
    const dim3 grid((p.T + 127) / 128, p.B * p.H), block(256);
    const long nwg = (long)grid.x * grid.y;
    if (cap < 0 || (p.mode == 1 && (nwg > cap || nwg > DELTA_GRANULES))) return 1;   // the δ sweep needs every workgroup resident
    hipLaunchKernelGGL((attn3_one_kernel<D, UNIFORM, QM, VINT>), grid, block, lds, st, p, gran, (int)nwg);
    return dgq_launch_status("dgq_attention(one launch)");
}

}  // namespace

#define ONE_DISPATCH(FN, ...)                                                                                                       \
    switch (D) {                                                                                                                    \
        ONE_CASE(FN, 40, __VA_ARGS__);                                                                                              \
        ONE_CASE(FN, 64, __VA_ARGS__);                                                                                              \
        ONE_CASE(FN, 80, __VA_ARGS__);                                                                                              \
        ONE_CASE(FN, 160, __VA_ARGS__);                                                                                             \
        default: return 1;                                                                                                          \
    }
#define ONE_U(FN, DD, QQ, VV, ...) (p.mode == 3 ? FN<DD, true, QQ, VV>(__VA_ARGS__) : FN<DD, false, QQ, VV>(__VA_ARGS__))
#define ONE_V(FN, DD, QQ, ...) (vint ? ONE_U(FN, DD, QQ, true, __VA_ARGS__) : ONE_U(FN, DD, QQ, false, __VA_ARGS__))
#define ONE_CASE(FN, DD, ...) case DD: return qm == 1 ? ONE_V(FN, DD, 1, __VA_ARGS__) : (qm == 2 ? ONE_V(FN, DD, 2, __VA_ARGS__) : ONE_V(FN, DD, 3, __VA_ARGS__))

// Behind the pre-pass: returns 1 when this form does not take the call (head dim / operand format not instantiated, key range too long,
// grid beyond the resident capacity under the real-time δ): the caller runs the three launches.
int dgq_attention_one_launch(const AttnParams& p, int D, int qm, bool vint, unsigned* sync, hipStream_t st) {
    const char* e = getenv("DGQ_ATTN_ONE");                // read per call: tests toggle it in-process
    if (e && e[0] == '0') return 1;
    if (p.NT > NTM || qm < 1 || qm > 3) return 1;
    unsigned long long* gran = reinterpret_cast<unsigned long long*>(sync);
    ONE_DISPATCH(launch_one, p, gran, st)
}

#undef ONE_CASE
#undef ONE_V
#undef ONE_U
#undef ONE_DISPATCH

extern "C" int dgq_attention_sync_timeouts(void) {
    unsigned v = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(dgq_attn_sync_timeouts_dev), sizeof(v)) != hipSuccess) return -1;
    return (int)v;
}
