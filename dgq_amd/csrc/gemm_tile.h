// Shared by the W4A8 / W8A8 GEMM kernels of libdgq_hip.so (gemm_wxa8.hip: the tile family of the layer shapes of a UNet step;
// gemm_wxa8_big.hip: the 256-row ping-pong kernel for the compute-bound shapes): launch parameters, prologue helpers and the
// dequantising store epilogue — ONE implementation, so that every kernel's results agree bit for bit.
#pragma once
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include "dgq_common.h"
#include "gemm_device.h"
#include "diag.h"

#define BK 128
#define NCH 4                 // 32-wide chunks per K tile

typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));

struct GemmParams {
    const int8_t* codes;
    const float* rowsum;
    int rowsum_parts;
    int M, Kp, N;
    const uint8_t* wpacked;
    const float* cdelta;
    const uint8_t* cflush;
    const float* mdelta;
    const float* mzp;
    int L;
    float offset;
    const float* alpha;
    const float* zw;
    const float* gamma;
    const float* vn;
    void* y;
    int ldy;
    float* slab;          // split-K partials [S][M][N] (nullptr when S == 1)
    int splits;
    int tiles_per_split;  // K tiles (of BK) per split
    dgq_gemm_extra_t ex;  // optional epilogue extras (residual add, fused attention-side quantizer, GEGLU pairs)
    dgq_gemm_conv_t cv;   // implicit im2col A operand (cv.codes_in != nullptr); ex.conv is not used on the device
    const float* ccoef;   // per-K, unsplit: host-formed flush coefficients + clear flags (ex.flush_coef) or nullptr
    const uint8_t* wfrag; // the same W4 weights fragment-major (dgq_pack_w4 layout 2; ex.wfrag) or nullptr: what gemm_panel.hip reads
    dgq_gemm_act_t act;   // quantise-on-load inside the panel kernel (act.x != nullptr; ex.act), see include/dgq_hip.h
};

// Up to DGQ_GEMM_BATCH problems of one kernel instance (tile shape, weight bits, scale mode, output dtype; no K split) in
// ONE launch: blockIdx.z = problem.  Grid x / y cover the widest problem; blocks outside a problem's tile grid exit.
#define DGQ_GEMM_BATCH 8
struct GemmBatch {
    GemmParams p[DGQ_GEMM_BATCH];
    int n;                 // 1: p[0], blockIdx.z = K split;  > 1: blockIdx.z = problem, no split
};

// Every scalar a GEMM workgroup needs from its kernel arguments, fetched in ONE burst at kernel entry.  hipcc otherwise emits each
// s_load where the field is first used — behind early-exit branches, one `s_waitcnt lgkmcnt(0)` each: eight dependent scalar-cache
// round trips (1100-2600 cycles from entry to the first vector load, profiles/r05_small_launch_timeline.txt).  An asm statement that
// takes the values as inputs pins their loads in front of it; later reads of the same (invariant) fields reuse them.
__device__ __forceinline__ void gemm_prefetch_params(const GemmParams& p) {
    asm volatile("" ::"s"(p.M), "s"(p.N), "s"(p.Kp), "s"(p.tiles_per_split), "s"(p.splits), "s"(p.codes), "s"(p.wpacked), "s"(p.rowsum),
                 "s"(p.rowsum_parts), "s"(p.alpha), "s"(p.zw), "s"(p.gamma), "s"(p.vn), "s"(p.cdelta), "s"(p.cflush), "s"(p.mdelta),
                 "s"(p.mzp), "s"(p.L), "s"(p.offset), "s"(p.y), "s"(p.ldy), "s"(p.slab), "s"(p.ex.residual), "s"(p.ex.ldr),
                 "s"(p.ex.res_div), "s"(p.ex.res_dtype), "s"(p.ex.fq_mode), "s"(p.ex.geglu), "s"(p.ex.gn_partial), "s"(p.wfrag), "s"(p.ex.y2));
}

template <bool PER_M>
__device__ __forceinline__ float dgq_epilogue(const GemmParams& p, float acc, int m, int n, float al, float zw, float ga,
                                              float vn) {
    float rs = 0.0f;
    for (int j = 0; j < p.rowsum_parts; ++j) rs += p.rowsum[(int64_t)j * p.M + m];
    if (PER_M) {
        const int li = m % p.L;
        const float md = p.mdelta[li], mz = p.mzp[li];
        return dgq_dequant<true>(acc, md, md * rs, md * (p.offset - mz), al, zw, ga, vn);
    }
    return dgq_dequant<false>(acc, 1.0f, rs, 0.0f, al, zw, ga, vn);
}

// x·gelu(g), gelu by erf as F.gelu's default (FeedForward's GEGLU, diffusers_rewrite/sd.py:210-222)
__device__ __forceinline__ float dgq_geglu(float a, float g) { return a * (0.5f * g * (1.0f + erff(g * 0.70710678118654752f))); }

// LDS ring: NST stages (template parameter; 3 is what ships — see launch_ring).
constexpr int gemm_stage_bytes(int wbits, int bm, int bn) { return bm * BK + bn * (wbits == 4 ? BK / 2 : BK); }
// waves per SIMD the LDS ring of a tile shape allows (ring + tables), capped at 5 blocks per CU: the register budget follows
constexpr int gemm_waves_per_simd(int wbits, int bm, int bn, int nw, int nst) {
    const int per_block = nst * gemm_stage_bytes(wbits, bm, bn) + 6 * 1024;
    int o = (160 * 1024) / per_block;
    o = o > 5 ? 5 : (o < 1 ? 1 : o);
    const int w = o * nw / 4;
    return w > 8 ? 8 : w;
}

// prologue loads the compiler must not count: beside LDS-DMA in flight hipcc waits vmcnt(0) for any load of its own, which
// would drain the ring it is supposed to leave in flight (cdna guide §5 trap (b)); the wait is the counted one below
__device__ __forceinline__ float gload_f32(const float* q) {
    float v;
    asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(q) : "memory");
    return v;
}
__device__ __forceinline__ uint32_t gload_u8(const uint8_t* q) {
    uint32_t v;
    asm volatile("global_load_ubyte %0, %1, off" : "=v"(v) : "v"(q) : "memory");
    return v;
}
template <int N>
__device__ __forceinline__ void wait_vmcnt_lgkm0() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
}
// runtime count of DMA instructions that may stay in flight: `tiles` younger tiles of PER instructions each
template <int PER, int MAXT>
__device__ __forceinline__ void wait_ring(int tiles) {
    static_assert(PER * MAXT <= 63, "vmcnt is a 6-bit counter");
    if (tiles <= 0) wait_vmcnt_lgkm0<0>();
    else if (MAXT >= 1 && tiles == 1) wait_vmcnt_lgkm0<PER>();
    else if (MAXT >= 2 && tiles == 2) wait_vmcnt_lgkm0<PER * (MAXT >= 2 ? 2 : 0)>();
    else if (MAXT >= 3 && tiles == 3) wait_vmcnt_lgkm0<PER * (MAXT >= 3 ? 3 : 0)>();
    else wait_vmcnt_lgkm0<PER * (MAXT >= 4 ? 4 : 0)>();
}


// Store epilogue of one workgroup tile.  acc0 (per-M) / accf (per-K) hold the wave's (BM/WVM) x (BN/WVN) tile in the MFMA C/D
// layout; vtab [3][BM] = R0 R1 R2 per row and vcol [4][BN] = alpha zw gamma vn per column were staged into LDS by the prologue;
// smem .. smem + LDS_CAP is free for the transposition (the operand ring, idle by now: the CALLER has made sure — a barrier —
// that no wave still reads it).  Variants: split-K slab, GEGLU pairs, residual / fused quantizer, GroupNorm partials.
// TILED (gemm_convq.hip: the workgroup's 32 rows are a 4 x 8 tile of output positions of one image, not 32 consecutive rows): tile row r
// is output row m0 + (r >> 3)·tiled_w + (r & 7) (m0 = the tile's first position, tiled_w = Wo), and its two 16-row halves write the
// GroupNorm partial slots tiled_gn0, tiled_gn0 + 1 (any partition of an image's rows into 16-row blocks merges to the same statistics).
template <bool PER_M, typename TOut, int BM, int BN, int WVM, int WVN, int WVK, int LDS_CAP, int TM, int TN, bool TILED = false>
__device__ __forceinline__ void gemm_store_tile(const GemmParams& p, int zsplit, uint8_t* smem, const float* vtab, const float* vcol,
                                                int wid, int lane, int wave_m, int wave_n, int wave_k, int m0, int n0,
                                                const v16i (&acc0)[TM][TN], const v16f (&accf)[TM][TN] DGQ_DIAG_PARAM,
                                                int tiled_w = 0, int tiled_gn0 = 0) {
    static_assert(!TILED || (BM == 32 && WVM == 1 && WVK == 1), "tiled rows: one 32-row tile per workgroup");
    constexpr int NW = WVM * WVN * WVK;
    constexpr int WM = BM / WVM, WN = BN / WVN;
    static_assert(TM == WM / 32 && TN == WN / 32, "wave tile");
    const int lr = lane & 31, hh = lane >> 5;
    // epilogue.  The MFMA C/D layout (col = lane&31, row = (reg&3) + 8·(reg>>2) + 4·(lane>>5)) gives 4-byte stores in 128-byte
    // runs; 16 bytes per lane measured faster on every small-K layer (the stores, not the MFMAs, bound them).  Each wave
    // therefore transposes its WM x WN fp32 tile through the (now idle) LDS ring and the workgroup writes 16 bytes per lane,
    // WN·4 contiguous bytes per row.  Row stride WN + 4 floats: conflict-free ds_write_b32, near conflict-free ds_read_b128.
    // With WVK = 2 the two K halves of a tile sit in two regions; after a barrier each of the two waves adds them (k = 0
    // first: a fixed order) for half of the rows.
    constexpr int EP_LD = WN + 4;
    constexpr int LPR = WN / 4;                              // lanes per output row (4 consecutive n each)
    constexpr int RPP = 64 / LPR;                            // rows per pass of the wave
    constexpr int PASSES = WM / RPP / WVK;                   // passes of this wave
    // The 256-row kernel's eight wave tiles (128x64: 278 KB with the padding; 128x32 / 64x64: 139-147 KB) do not fit its ring: such
    // a wave stages its tile in EPH = 2 halves (whole MFMA row tiles), each a same-wave LDS round trip like the whole tile elsewhere.
    constexpr int EPH = (NW * WM * EP_LD * 4 <= LDS_CAP) ? 1 : 2;
    static_assert(EPH == 1 || (WVK == 1 && TM % 2 == 0 && PASSES % 2 == 0), "half-tile staging: whole MFMA row tiles per half");
    constexpr int HROWS = WM / EPH;                          // rows staged at a time
    constexpr int REGION = HROWS * EP_LD;                    // floats per wave region
    static_assert(NW * REGION * 4 <= LDS_CAP, "epilogue staging must fit the LDS ring");
    float* ep_all = reinterpret_cast<float*>(smem);
    float* ep = ep_all + wid * REGION;
    auto stage_half = [&](int h) {
#pragma unroll
        for (int i = 0; i < TM / EPH; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ti = h * (TM / EPH) + i;
                    ep[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh) * EP_LD + j * 32 + lr] = PER_M ? (float)acc0[ti][j][r] : accf[ti][j][r];
                }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    // before pass rr of a two-half tile: the second half replaces the first once this wave's reads of it have returned
    auto next_half = [&](int rr) {
        if (EPH == 2 && rr == PASSES / 2) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            stage_half(1);
        }
    };
    stage_half(0);
    if (WVK > 1) __builtin_amdgcn_s_barrier();              // WVK == 1: same-wave LDS round trip, no barrier needed
    DGQ_STAMP(8);
    const float* ep0 = ep_all + (wave_m * WVN + wave_n) * REGION;                     // k = 0 half
    const float* ep1 = ep0 + (WVM * WVN) * REGION;                                    // k = 1 half (WVK == 2)
    const int c4 = (lane % LPR) * 4;                         // 4 consecutive n per lane
    const int lrow = lane / LPR;
    const int nb = n0 + wave_n * WN + c4;
    const bool vec_ok = (nb + 3 < p.N);
    // the wave's rr-th pass: the two K halves of a WVK = 2 tile take the upper / lower half of the rows (contiguous row sets per
    // wave: the GroupNorm partials below are per 16-row block)
    auto tile_row = [&](int rr) { return wave_k * (WM / WVK) + rr * RPP + lrow; };
    auto row_m = [&](int row) { return TILED ? m0 + (row >> 3) * tiled_w + (row & 7) : m0 + wave_m * WM + row; };
    auto tile_val = [&](int row) {
        const int sr = (EPH == 2) ? (row & (HROWS - 1)) : row;                         // row inside the staged half
        float4 v = *reinterpret_cast<const float4*>(ep0 + sr * EP_LD + c4);
        if (WVK > 1) {
            const float4 u = *reinterpret_cast<const float4*>(ep1 + sr * EP_LD + c4);
            v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
        }
        return v;
    };
    if (p.splits > 1) {
        float* slab = p.slab + (int64_t)zsplit * p.M * p.N;
        const bool al16 = ((p.N & 3) == 0);
#pragma unroll 4
        for (int rr = 0; rr < PASSES; ++rr) {
            next_half(rr);
            const int row = tile_row(rr);
            const int m = row_m(row);
            if (m >= p.M || nb >= p.N) continue;
            const float4 v = tile_val(row);
            float* dst = slab + (int64_t)m * p.N + nb;
            if (vec_ok && al16) {
                *reinterpret_cast<float4*>(dst) = v;
            } else {
                const float e[4] = {v.x, v.y, v.z, v.w};
                for (int k = 0; k < 4 && nb + k < p.N; ++k) dst[k] = e[k];
            }
        }
        return;
    }
    TOut* y = reinterpret_cast<TOut*>(p.y);
    const float* vc = vcol + wave_n * WN + c4;
    const float4 al = *reinterpret_cast<const float4*>(vc);
    const float4 zw = *reinterpret_cast<const float4*>(vc + BN);
    const float4 ga = *reinterpret_cast<const float4*>(vc + 2 * BN);
    const float4 vn = *reinterpret_cast<const float4*>(vc + 3 * BN);
    if (p.ex.geglu) {
        // FeedForward's GEGLU in the epilogue of ff.net.0 (sd.py:210-236): the weight rows were interleaved at pack time
        // (row 2i = value half i, row 2i+1 = gate half i), so a lane's four columns are two (value, gate) pairs and it
        // writes columns nb/2, nb/2 + 1 of the [M][N/2] output — half the stores, and no GEGLU pass in front of ff.net.2.
        const int ob = nb >> 1;
        const bool st2 = vec_ok && (p.ldy % 2 == 0) && ((reinterpret_cast<uintptr_t>(p.y) & 7) == 0);
#pragma unroll
        for (int rr = 0; rr < PASSES; ++rr) {
            next_half(rr);
            const int row = tile_row(rr);
            const int m = row_m(row);
            if (m >= p.M || nb >= p.N) continue;
            const float4 v = tile_val(row);
            const float* vr = vtab + wave_m * WM + row;
            const float r0 = vr[0], r1 = vr[BM], r2 = vr[2 * BM];
            const float a0 = dgq_dequant<PER_M>(v.x, r0, r1, r2, al.x, zw.x, ga.x, vn.x);
            const float g0 = dgq_dequant<PER_M>(v.y, r0, r1, r2, al.y, zw.y, ga.y, vn.y);
            const float a1 = dgq_dequant<PER_M>(v.z, r0, r1, r2, al.z, zw.z, ga.z, vn.z);
            const float g1 = dgq_dequant<PER_M>(v.w, r0, r1, r2, al.w, zw.w, ga.w, vn.w);
            const float o0 = dgq_geglu(a0, g0), o1 = dgq_geglu(a1, g1);
            TOut* dst = y + (int64_t)m * p.ldy + ob;
            if (st2) {
                if (sizeof(TOut) == 4) {
                    *reinterpret_cast<float2*>(dst) = make_float2(o0, o1);
                } else {
                    TOut t[2] = {dgq_from_float<TOut>(o0), dgq_from_float<TOut>(o1)};
                    *reinterpret_cast<uint32_t*>(dst) = *reinterpret_cast<const uint32_t*>(t);
                }
            } else {
                dst[0] = dgq_from_float<TOut>(o0);
                if (nb + 3 < p.N) dst[1] = dgq_from_float<TOut>(o1);
            }
        }
        return;
    }
    const bool st_vec = vec_ok && ((p.ldy * (int)sizeof(TOut)) % 16 == 0) &&
                        ((reinterpret_cast<uintptr_t>(p.y) & 15) == 0) && (sizeof(TOut) == 4 || (p.ldy & 3) == 0);
    const bool st_vec2 = vec_ok && ((p.ex.ldy2 * (int)sizeof(TOut)) % 16 == 0) &&
                         ((reinterpret_cast<uintptr_t>(p.ex.y2) & 15) == 0) && (sizeof(TOut) == 4 || (p.ex.ldy2 & 3) == 0);
    // residual tile: all rows of this lane fetched up front as 16-byte loads, so the epilogue pays one memory latency
    // (fetched row by row inside the store loop, the dependent loads made the fused add slower than a separate kernel)
    const int res_es = p.ex.res_dtype == DGQ_F32 ? 4 : 2;
    const bool res_vec = p.ex.residual != nullptr && vec_ok && (p.ex.ldr & 3) == 0 &&
                         (reinterpret_cast<uintptr_t>(p.ex.residual) & (4 * res_es - 1)) == 0;
    // (a two-half tile prefetches per half — HP passes — right after the half is staged: a 128-row wave tile would otherwise
    // hold 128 registers of residual beside its accumulators)
    constexpr int HP = PASSES / EPH;
    float4 res[HP];
    auto load_res = [&](int h) {
        if (!res_vec) return;
#pragma unroll
        for (int q = 0; q < HP; ++q) {
            const int m = min(row_m(tile_row(h * HP + q)), p.M - 1);
            const int64_t i = (int64_t)(m / p.ex.res_div) * p.ex.ldr + nb;
            if (p.ex.res_dtype == DGQ_F32) {
                res[q] = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.ex.residual) + i);
            } else {
                const uint2 t = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(p.ex.residual) + i);
                if (p.ex.res_dtype == DGQ_F16) {
                    const __half* h16 = reinterpret_cast<const __half*>(&t);
                    res[q] = make_float4(__half2float(h16[0]), __half2float(h16[1]), __half2float(h16[2]), __half2float(h16[3]));
                } else {
                    res[q] = make_float4(__uint_as_float(t.x << 16), __uint_as_float(t.x & 0xFFFF0000u),
                                         __uint_as_float(t.y << 16), __uint_as_float(t.y & 0xFFFF0000u));
                }
            }
        }
    };
    load_res(0);
    dgq_gemm_extra_t exl = p.ex;
    if (res_vec) exl.residual = nullptr;                 // added below from the prefetched tile
    const bool has_extra = exl.fq_mode != 0 || exl.residual != nullptr;
    // GroupNorm partial statistics of the OUTPUT tensor (ex.gn_partial, M % 16 == 0 and N % 4 == 0 checked on the host): per
    // 16-row block and column the mean and the sum of squared deviations of the values this launch stores, so that the
    // GroupNorm in front of the next layer (QuantResnetBlock2D norm2 / the next block's norm1, quant_block.py:98-119) needs
    // no pass over the tensor: dgq_groupnorm_from_partials merges them.  A wave's rows are contiguous; a lane sums its
    // PPB passes of a block with the block's first value as the shift (no cancellation), the RPP lanes that share the
    // columns merge pairwise (equal counts: Chan's formula), lane row 0 writes.
    constexpr int PPB = (16 / RPP) < 1 ? 1 : (16 / RPP);  // passes per 16-row block
    static_assert(RPP <= 16 && (WM / WVK) % 16 == 0, "GroupNorm partials: 16-row blocks per wave");
    const bool gn = p.ex.gn_partial != nullptr;
    float gK[4] = {0.f, 0.f, 0.f, 0.f}, g1[4] = {0.f, 0.f, 0.f, 0.f}, g2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int rr = 0; rr < PASSES; ++rr) {
        next_half(rr);
        if (EPH == 2 && rr == HP) load_res(1);
        const int row = tile_row(rr);
        const int m = row_m(row);
        if (m >= p.M || nb >= p.N) continue;
        const float4 v = tile_val(row);
        const float* vr = vtab + wave_m * WM + row;
        const float r0 = vr[0], r1 = vr[BM], r2 = vr[2 * BM];
        // y = alpha·(R0·acc − zw·R1 + R2·vn) + gamma   (per-K: R0 = 1, R1 = rowsum, R2 = 0)
        float o[4];
        o[0] = dgq_dequant<PER_M>(v.x, r0, r1, r2, al.x, zw.x, ga.x, vn.x);
        o[1] = dgq_dequant<PER_M>(v.y, r0, r1, r2, al.y, zw.y, ga.y, vn.y);
        o[2] = dgq_dequant<PER_M>(v.z, r0, r1, r2, al.z, zw.z, ga.z, vn.z);
        o[3] = dgq_dequant<PER_M>(v.w, r0, r1, r2, al.w, zw.w, ga.w, vn.w);
        if (has_extra) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (nb + k < p.N) o[k] = dgq_extra(exl, o[k], m, nb + k);
        }
        if (res_vec) {
            o[0] += res[rr % HP].x; o[1] += res[rr % HP].y; o[2] += res[rr % HP].z; o[3] += res[rr % HP].w;
        }
        // ex.y2: a second copy of the rows (the skip tensor's slot in the concatenation buffer of the up path, sd.py:558-613): the same
        // values, its own row pitch — no torch.cat launch later
#pragma unroll
        for (int cp = 0; cp < 2; ++cp) {
            if (cp == 1 && p.ex.y2 == nullptr) break;                                  // (kernel-uniform)
            TOut* dst = cp == 0 ? y + (int64_t)m * p.ldy + nb : reinterpret_cast<TOut*>(p.ex.y2) + (int64_t)m * p.ex.ldy2 + nb;
            if (cp == 0 ? st_vec : st_vec2) {
                if (sizeof(TOut) == 4) {
                    *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
                } else {
                    TOut t[4] = {dgq_from_float<TOut>(o[0]), dgq_from_float<TOut>(o[1]), dgq_from_float<TOut>(o[2]),
                                 dgq_from_float<TOut>(o[3])};
                    *reinterpret_cast<uint2*>(dst) = *reinterpret_cast<const uint2*>(t);
                }
            } else {
                for (int k = 0; k < 4 && nb + k < p.N; ++k) dst[k] = dgq_from_float<TOut>(o[k]);
            }
        }
        if (gn) {                                        // wave-uniform; rows of a 16-row block are all valid or all past M
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float val = dgq_to_float(dgq_from_float<TOut>(o[k]));      // the value as stored
                if (rr % PPB == 0) { gK[k] = val; g1[k] = 0.0f; g2[k] = 0.0f; }
                else { const float dv = val - gK[k]; g1[k] += dv; g2[k] += dv * dv; }
            }
            if (rr % PPB == PPB - 1) {
                float mean[4], m2[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float md = g1[k] * (1.0f / PPB);
                    mean[k] = gK[k] + md;
                    m2[k] = fmaxf(g2[k] - g1[k] * md, 0.0f);
                }
                float cnt = (float)PPB;
#pragma unroll
                for (int off = LPR; off < 64; off <<= 1) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float om = __shfl_xor(mean[k], off, 64), o2 = __shfl_xor(m2[k], off, 64);
                        const float dd = om - mean[k];
                        m2[k] = m2[k] + o2 + dd * dd * (0.5f * cnt);
                        mean[k] = 0.5f * (mean[k] + om);
                    }
                    cnt *= 2.0f;
                }
                if (lrow == 0) {
                    float* q = p.ex.gn_partial + ((int64_t)(TILED ? tiled_gn0 + (row >> 4) : (m >> 4)) * p.N + nb) * 2;
                    *reinterpret_cast<float4*>(q) = make_float4(mean[0], m2[0], mean[1], m2[1]);
                    *reinterpret_cast<float4*>(q + 4) = make_float4(mean[2], m2[2], mean[3], m2[3]);
                }
            }
        }
    }
}

