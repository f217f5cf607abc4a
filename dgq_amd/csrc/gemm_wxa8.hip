// W4A8 / W8A8 GEMM on V_MFMA_I32_16X16X64_I8 with DGQ per-group dequantisation fused in.
//
//   A  = int8 activation codes [M][Kp]  (from dgq_quant_act; K already permuted so that each DGQ group is a run
//        of 64-wide chunks)
//   W  = int4 packed [N][Kp/2]  or int8 [N][Kp]
//   Y  = fp [M][N]
//
// Tiling (wave64, gfx950): BM x BN block tile with BM in {32,64,128}, BN in {64,128} (template parameters; the host picks
// the pair per shape so that the grid fills the 256 CUs without a K split wherever it can), BK = 128 (two MFMA K-slices),
// 256 threads = 2x2 waves, each wave owns (BM/2)x(BN/2) = TM x TN MFMA tiles: 4·TM·TN int32 accumulators + (per-K mode)
// as many fp32 accumulators (128x128: 64 + 64).
//
// Operand staging is LDS-DMA only (global_load_lds_dwordx4: no staging VGPRs, no ds_write — ds_write_b128 runs at
// ~79 B/clk/CU and was the bottleneck of the register-staged version): a 3-stage LDS ring, tile t+2 is issued
// before the MFMAs of tile t, a counted s_waitcnt vmcnt leaves it in flight across the (raw) barrier.
// LDS images are lane-linear per DMA instruction, so the bank swizzle is applied to the per-lane SOURCE address
// and to the ds_read address (cdna guide rule 21):
//   A / int8-W tile: 128-byte rows, 16-byte chunk c stored at c ^ ((row>>1)&7)   -> conflict-free ds_read_b128
//   int4-W tile    : 64-byte rows (packed), 8-byte slot s stored at s ^ (((row>>2)&3)<<1) -> conflict-free ds_read_b64
// int4 weights stay packed in LDS (half the LDS bytes) and are widened to int8 in registers right after the
// ds_read: (w & 0x0F0F0F0F), ((w>>4) & 0x0F0F0F0F) — the dgq_pack_w4 nibble order makes those 4 consecutive k each.
//
// Small grids (most SD1.4 layers give 10..192 tiles on 256 CUs) use deterministic split-K: grid.z slices of the
// K-tile range write fp32 partial slabs [S][M][N] to a caller-provided workspace; splitk_epilogue_kernel sums
// them in a fixed order and applies the dequantisation epilogue (no float atomics: results are bit-reproducible).
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include "dgq_common.h"
#include "gemm_device.h"

#define BK 128

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
    dgq_gemm_extra_t ex;  // optional epilogue extras (residual add, fused attention-side quantizer)
};

// Up to DGQ_GEMM_BATCH problems of one kernel instance (tile shape, weight bits, scale mode, output dtype; no K split) in
// ONE launch: blockIdx.z = problem.  Grid x / y cover the widest problem; blocks outside a problem's tile grid exit.
#define DGQ_GEMM_BATCH 8
struct GemmBatch {
    GemmParams p[DGQ_GEMM_BATCH];
    int n;                 // 1: p[0], blockIdx.z = K split;  > 1: blockIdx.z = problem, no split
};

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

// LDS ring depth: 3 stages for every tile shape.  Deeper rings for the small tiles (6 stages at 32x64, 4 at 64x64 — more
// bytes in flight per block) measured 5-25 % SLOWER on the SD layer shapes (tools/tile_sweep.py: 2048x640x1408 per-K
// 11.4 -> 12.4 us, 8192x320x1024 14.2 -> 17.9): the extra LDS costs a resident block per CU, which hides more latency than
// the longer ring does.
constexpr int gemm_stage_bytes(int wbits, int bm, int bn) { return bm * BK + bn * (wbits == 4 ? BK / 2 : BK); }
constexpr int gemm_stages(int wbits, int bm, int bn) { return 3; }
// ... except for grids of at most ~2 workgroups per CU (M <= 512 layers: 320 tiles of 32x64), where occupancy is not LDS-bound
// and the K loop waits on every tile: there 6 stages keep 5 tiles in flight per workgroup (template parameter NST).
// blocks per CU the LDS ring of a tile shape allows (ring + tables), capped at 5 (32x64: 1280 slots = the whole
// 8192 x 320 grid in one round): the register budget follows from it
constexpr int gemm_occupancy(int wbits, int bm, int bn, int nst) {
    const int per_block = nst * gemm_stage_bytes(wbits, bm, bn) + 6 * 1024;
    const int o = (160 * 1024) / per_block;
    return o > 5 ? 5 : (o < 1 ? 1 : o);
}

// WVN = waves along n (2: the four waves form a 2 x 2 grid; 4: a 1 x 4 row, each wave owning all BM rows of BN/4 columns).
// With int4 weights every B fragment costs 6 VALU to widen and is reused by the TM row tiles of its wave: at BM = 32 the
// 2 x 2 grid has TM = 1 (6 VALU per MFMA), the 1 x 4 row TM = 2 (3 per MFMA) at the same number of LDS reads.
template <int WBITS, bool PER_M, typename TOut, int BM, int BN, int NST = 3, int WVN = 2>
__global__ __launch_bounds__(256, gemm_occupancy(WBITS, BM, BN, NST)) void gemm_wxa8_kernel(GemmBatch bt) {
    const GemmParams& p = bt.p[bt.n > 1 ? blockIdx.z : 0];
    const int zsplit = bt.n > 1 ? 0 : blockIdx.z;
    // XCD-aware tile order.  Workgroups are dealt round-robin over the 8 XCDs (each with its own L2): in launch order the
    // n tiles of one m tile — which all read the same activation rows — would sit on 8 different XCDs, and every XCD would
    // fetch the whole activation matrix through the fabric (PMC: 4.4x the algorithmic bytes per launch).  Remapped so that
    // XCD k owns a contiguous range of (m-major) tiles: the activation rows of an m tile are fetched by one XCD and re-used
    // from its L2 by the other n tiles; only the (small) weight matrix is read by all eight.  Bijective for any grid size.
    int tile_n, tile_m;
    {
        const int gx = gridDim.x, T = gridDim.x * gridDim.y;
        const int bid = blockIdx.x + gx * blockIdx.y;
        const int q = T >> 3, r = T & 7, xcd = bid & 7;
        const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
        tile_m = logical / gx;
        tile_n = logical - tile_m * gx;
    }
    if (tile_n * BN >= p.N || tile_m * BM >= p.M) return;   // batch: a narrower problem than the grid (whole block)
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int STAGES = NST;
    constexpr int WVM = 4 / WVN;
    constexpr int WM = BM / WVM, WN = BN / WVN;            // per-wave output tile
    static_assert(WM % 16 == 0 && WN % 16 == 0, "wave tile");
    constexpr int TM = WM / 16, TN = WN / 16;              // MFMA tiles per wave
    static_assert(BM % 32 == 0 && BM <= 128 && BN % 64 == 0 && BN <= 128, "tile shape");
    constexpr int A_BYTES = BM * BK;                       // 16 KiB at BM = 128
    constexpr int W_ROW = (WBITS == 4) ? BK / 2 : BK;      // bytes per n-row per stage
    constexpr int W_BYTES = BN * W_ROW;
    constexpr int STAGE_BYTES = A_BYTES + W_BYTES;
    constexpr int A_DMA = A_BYTES / 1024 / 4;              // DMA instructions per wave per tile (1 KiB each)
    constexpr int W_DMA = W_BYTES / 1024 / 4;
    static_assert(A_DMA >= 1 && W_DMA >= 1, "every wave stages at least one piece of each operand");
    constexpr int DMA_PER_TILE = A_DMA + W_DMA;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wid / WVN, wave_n = wid % WVN;
    const int n0 = tile_n * BN;
    const int m0 = tile_m * BM;
    const int nk_total = p.Kp / BK;
    const int kt_begin = zsplit * p.tiles_per_split;
    const int kt_end = min(nk_total, kt_begin + p.tiles_per_split);
    const int nk = kt_end - kt_begin;

    // per-lane global source pointers of this wave's DMA pieces (k offset added per tile)
    const int8_t* a_src[A_DMA];
#pragma unroll
    for (int i = 0; i < A_DMA; ++i) {
        const int blk = wid * A_DMA + i;                   // 1 KiB = 8 rows of 128 B
        const int row = blk * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        const int m = min(m0 + row, p.M - 1);
        a_src[i] = p.codes + (int64_t)m * p.Kp + 16 * c;
    }
    const uint8_t* w_src[W_DMA];
#pragma unroll
    for (int i = 0; i < W_DMA; ++i) {
        const int blk = wid * W_DMA + i;
        if (WBITS == 4) {                                  // 1 KiB = 16 rows of 64 B
            const int row = blk * 16 + (lane >> 2);
            const int c = (lane & 3) ^ ((row >> 2) & 3);
            const int n = min(n0 + row, p.N - 1);
            w_src[i] = p.wpacked + (int64_t)n * (p.Kp / 2) + 16 * c;
        } else {
            const int row = blk * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((row >> 1) & 7);
            const int n = min(n0 + row, p.N - 1);
            w_src[i] = p.wpacked + (int64_t)n * p.Kp + 16 * c;
        }
    }

    const uint32_t lds_base = (uint32_t)(uintptr_t)(lptr_t)smem;
    auto issue_tile = [&](int kt, int stage) {
        const uint32_t sa = lds_base + stage * STAGE_BYTES;
        const uint32_t sw = sa + A_BYTES;
        const int64_t ka = (int64_t)kt * BK;
        const int64_t kw = (WBITS == 4) ? ka / 2 : ka;
#pragma unroll
        for (int i = 0; i < A_DMA; ++i)
            glds16(a_src[i] + ka, __builtin_amdgcn_readfirstlane(sa + (wid * A_DMA + i) * 1024));
#pragma unroll
        for (int i = 0; i < W_DMA; ++i)
            glds16(w_src[i] + kw, __builtin_amdgcn_readfirstlane(sw + (wid * W_DMA + i) * 1024));
    };

    v4i acc[TM][TN];
    v4f accf[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            acc[i][j] = (v4i){0, 0, 0, 0};
            accf[i][j] = (v4f){0.f, 0.f, 0.f, 0.f};
        }

    const int fr = lane & 15, fq = lane >> 4;
    // ds_read byte offsets inside a stage (h = K half adds its chunk index below)
    int a_off[TM][2], w_off[TN][2];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = wave_m * WM + i * 16 + fr;
#pragma unroll
        for (int h = 0; h < 2; ++h) a_off[i][h] = row * BK + (((4 * h + fq) ^ ((row >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int row = wave_n * WN + j * 16 + fr;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (WBITS == 4) w_off[j][h] = row * W_ROW + (((4 * h + fq) ^ (((row >> 2) & 3) << 1)) << 3);
            else w_off[j][h] = row * BK + (((4 * h + fq) ^ ((row >> 1) & 7)) << 4);
        }
    }

    // prologue: two tiles in flight first, then (while they fly) stage everything the epilogue needs into LDS with
    // ordinary loads: per-chunk coefficients of the running totals (see below), per-row and
    // per-column dequantisation vectors.  A VMEM load inside the main loop would make hipcc drain the DMA ring with
    // vmcnt(0), and dependent global loads in the epilogue cost ~1 us each on a lone wave — both avoided this way.
#pragma unroll
    for (int i = 0; i < STAGES - 1; ++i)
        if (i < nk) issue_tile(kt_begin + i, i);
    float* vtab = reinterpret_cast<float*>(smem + STAGES * STAGE_BYTES);   // [3][BM]: R0 R1 R2 | [4][BN]: alpha zw gamma vn
    float* vcol = vtab + 3 * BM;
    float* ctab = vcol + 4 * BN;
    {
        const bool final_ep = (p.splits == 1);
        if (final_ep) {
            if (tid < BM) {
                const int m = min(m0 + tid, p.M - 1);
                float rs = 0.0f;
                for (int j = 0; j < p.rowsum_parts; ++j) rs += p.rowsum[(int64_t)j * p.M + m];
                float r0 = 1.0f, r1 = rs, r2 = 0.0f;
                if (PER_M) {
                    const int li = m % p.L;
                    const float md = p.mdelta[li], mz = p.mzp[li];
                    r0 = md; r1 = md * rs; r2 = md * (p.offset - mz);
                }
                vtab[tid] = r0; vtab[BM + tid] = r1; vtab[2 * BM + tid] = r2;
            } else if (tid >= 128 && tid - 128 < BN) {
                const int c = tid - 128;
                const int n = min(n0 + c, p.N - 1);
                vcol[c] = p.alpha[n]; vcol[BN + c] = p.zw[n]; vcol[2 * BN + c] = p.gamma[n];
                vcol[3 * BN + c] = PER_M ? p.vn[n] : 0.0f;
            }
        }
        if (!PER_M) {
            // Summation by parts: with T_c the RUNNING int32 total after chunk c (never cleared) and δ_c the scale of the
            // chunk's group, Σ_groups δ_g·P_g = Σ_c (δ_c − δ_{c+1})·T_c, δ := 0 past this block's K range.  The coefficient
            // is non-zero exactly at group ends, so a flush is cvt + fma per accumulator register and the clear (a third
            // VALU per register per group: PMC, 9.0 non-MFMA VALU per MFMA on the g16 GEGLU shape) disappears.  |T| stays
            // below 2^31 for every Kp the ABI admits with int4 weights, and below 2^24 (exact in fp32) up to Kp = 8800.
            for (int c = tid; c < 2 * nk; c += 256) {
                const int chunk = kt_begin * 2 + c;
                const float d = p.cdelta[chunk];
                const float dn = (c == 2 * nk - 1) ? 0.0f : p.cdelta[chunk + 1];
                ctab[c] = d - dn;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // DMA tiles 0,1 + the staging loads/stores
    __builtin_amdgcn_s_barrier();

    // ring invariant at the top of iteration t: tiles t .. t+STAGES-2 are issued, tile t has landed.
    // The fragment loads are software-pipelined one K-half ahead of the MFMAs that consume them: while the 16 MFMAs of
    // half h run, the ds_reads of the next half (or of the next tile's first half, after the barrier) are in flight.
    // A/B on one box (8192^3): per-K g16 34.3 -> 35.3 % of peak, per-M 41.7 -> 41.9 %.  int4 stays packed in the fragment
    // registers and is widened right before its MFMAs, so the load itself has no consumer until then.
    typedef typename std::conditional<WBITS == 4, uint2, v4i>::type wfrag_t;
    auto load_frags = [&](const uint8_t* sa, const uint8_t* sw, int h, v4i (&af)[TM], wfrag_t (&wf)[TN]) {
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const v4i*>(sa + a_off[i][h]);
#pragma unroll
        for (int j = 0; j < TN; ++j) wf[j] = *reinterpret_cast<const wfrag_t*>(sw + w_off[j][h]);
    };
    auto mma_half = [&](const v4i (&af)[TM], const wfrag_t (&wf)[TN], int chunk) {
        v4i bf[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            if constexpr (WBITS == 4) {
                const uint2 v = wf[j];
                bf[j] = (v4i){(int)(v.x & 0x0F0F0F0Fu), (int)((v.x >> 4) & 0x0F0F0F0Fu),
                              (int)(v.y & 0x0F0F0F0Fu), (int)((v.y >> 4) & 0x0F0F0F0Fu)};
            } else {
                bf[j] = wf[j];
            }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[i], bf[j], acc[i][j], 0, 0, 0);
        if (!PER_M) {
            // wave-uniform coefficient of this chunk's running total (0 inside a group: nothing to add)
            const float sc = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(
                __builtin_bit_cast(int, ctab[chunk])));
            if (sc != 0.0f) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) accf[i][j][r] = __builtin_fmaf(sc, (float)acc[i][j][r], accf[i][j][r]);
                    }
            }
        }
    };
    v4i af0[TM], af1[TM];
    wfrag_t wf0[TN], wf1[TN];
    int stage = 0, istage = STAGES - 1;
    load_frags(smem, smem + A_BYTES, 0, af0, wf0);
    for (int t = 0; t < nk; ++t) {
        if (t + STAGES - 1 < nk) issue_tile(kt_begin + t + STAGES - 1, istage);
        const uint8_t* sa = smem + stage * STAGE_BYTES;
        load_frags(sa, sa + A_BYTES, 1, af1, wf1);
        mma_half(af0, wf0, 2 * t);
        // tile t+1 must have landed (this wave's pieces) before anyone reads it; the STAGES-2 younger tiles may stay
        // in flight (vmcnt counts the wave's DMA instructions in issue order).  lgkmcnt(0): this wave's reads of tile t
        // are complete, so after the barrier its stage may be overwritten.
        {
            const int younger = min(STAGES - 2, max(0, nk - 2 - t));
            switch (younger) {
                case 0: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); break;
                case 1: asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(1 * DMA_PER_TILE) : "memory"); break;
                case 2: asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * DMA_PER_TILE) : "memory"); break;
                case 3: asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(3 * DMA_PER_TILE) : "memory"); break;
                default: asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(4 * DMA_PER_TILE) : "memory"); break;
            }
        }
        __builtin_amdgcn_s_barrier();
        stage = (stage + 1 == STAGES) ? 0 : stage + 1;
        istage = (istage + 1 == STAGES) ? 0 : istage + 1;
        if (t + 1 < nk) {
            const uint8_t* sn = smem + stage * STAGE_BYTES;
            load_frags(sn, sn + A_BYTES, 0, af0, wf0);
        }
        mma_half(af1, wf1, 2 * t + 1);
    }

    // epilogue.  The MFMA C/D layout (col = lane&15, row = (lane>>4)*4 + reg) would give 4-byte stores in 64-byte
    // runs; measured, that store pattern (not the MFMAs) bounded every small-K layer (~1 TB/s).  Each wave
    // therefore transposes its WM x WN fp32 tile through the (now idle) LDS ring and writes 16 bytes per lane, WN·4
    // contiguous bytes per row.  Row stride WN + 4 floats: conflict-free ds_write_b32, near conflict-free ds_read_b128.
    constexpr int EP_LD = WN + 4;
    constexpr int LPR = WN / 4;                              // lanes per output row (4 consecutive n each)
    constexpr int RPP = 64 / LPR;                            // rows per pass of the wave
    constexpr int PASSES = WM / RPP;
    static_assert(4 * WM * EP_LD * 4 <= STAGES * STAGE_BYTES, "epilogue staging must fit the LDS ring");
    float* ep = reinterpret_cast<float*>(smem) + wid * WM * EP_LD;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                ep[(i * 16 + fq * 4 + r) * EP_LD + j * 16 + fr] = PER_M ? (float)acc[i][j][r] : accf[i][j][r];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // same-wave LDS round trip: no barrier needed
    const int c4 = (lane % LPR) * 4;                         // 4 consecutive n per lane
    const int lrow = lane / LPR;
    const int nb = n0 + wave_n * WN + c4;
    const bool vec_ok = (nb + 3 < p.N);
    if (p.splits > 1) {
        float* slab = p.slab + (int64_t)zsplit * p.M * p.N;
        const bool al16 = ((p.N & 3) == 0);
#pragma unroll 4
        for (int rr = 0; rr < PASSES; ++rr) {
            const int row = rr * RPP + lrow;
            const int m = m0 + wave_m * WM + row;
            if (m >= p.M || nb >= p.N) continue;
            const float4 v = *reinterpret_cast<const float4*>(ep + row * EP_LD + c4);
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
    const bool st_vec = vec_ok && ((p.ldy * (int)sizeof(TOut)) % 16 == 0) &&
                        ((reinterpret_cast<uintptr_t>(p.y) & 15) == 0) && (sizeof(TOut) == 4 || (p.ldy & 3) == 0);
    // residual tile: all rows of this lane fetched up front as 16-byte loads, so the epilogue pays one memory latency
    // (fetched row by row inside the store loop, the dependent loads made the fused add slower than a separate kernel)
    const int res_es = p.ex.res_dtype == DGQ_F32 ? 4 : 2;
    const bool res_vec = p.ex.residual != nullptr && vec_ok && (p.ex.ldr & 3) == 0 &&
                         (reinterpret_cast<uintptr_t>(p.ex.residual) & (4 * res_es - 1)) == 0;
    float4 res[PASSES];
    if (res_vec) {
#pragma unroll
        for (int rr = 0; rr < PASSES; ++rr) {
            const int m = min(m0 + wave_m * WM + rr * RPP + lrow, p.M - 1);
            const int64_t i = (int64_t)(m / p.ex.res_div) * p.ex.ldr + nb;
            if (p.ex.res_dtype == DGQ_F32) {
                res[rr] = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.ex.residual) + i);
            } else {
                const uint2 t = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(p.ex.residual) + i);
                if (p.ex.res_dtype == DGQ_F16) {
                    const __half* h = reinterpret_cast<const __half*>(&t);
                    res[rr] = make_float4(__half2float(h[0]), __half2float(h[1]), __half2float(h[2]), __half2float(h[3]));
                } else {
                    res[rr] = make_float4(__uint_as_float(t.x << 16), __uint_as_float(t.x & 0xFFFF0000u),
                                          __uint_as_float(t.y << 16), __uint_as_float(t.y & 0xFFFF0000u));
                }
            }
        }
    }
    dgq_gemm_extra_t exl = p.ex;
    if (res_vec) exl.residual = nullptr;                 // added below from the prefetched tile
    const bool has_extra = exl.fq_mode != 0 || exl.residual != nullptr;
#pragma unroll
    for (int rr = 0; rr < PASSES; ++rr) {
        const int row = rr * RPP + lrow;
        const int m = m0 + wave_m * WM + row;
        if (m >= p.M || nb >= p.N) continue;
        const float4 v = *reinterpret_cast<const float4*>(ep + row * EP_LD + c4);
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
            o[0] += res[rr].x; o[1] += res[rr].y; o[2] += res[rr].z; o[3] += res[rr].w;
        }
        TOut* dst = y + (int64_t)m * p.ldy + nb;
        if (st_vec) {
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
}

// Deterministic split-K combine + dequantisation epilogue: one thread per 4 consecutive n.
template <bool PER_M, typename TOut>
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(GemmParams p) {
    const int n4 = (p.N + 3) / 4;
    const int64_t total = (int64_t)p.M * n4;
    TOut* y = reinterpret_cast<TOut*>(p.y);
    const int64_t slab_stride = (int64_t)p.M * p.N;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(i / n4);
        const int nb = (int)(i - (int64_t)m * n4) * 4;
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        const bool full = (nb + 3 < p.N) && ((p.N & 3) == 0);
        for (int s = 0; s < p.splits; ++s) {
            const float* src = p.slab + s * slab_stride + (int64_t)m * p.N + nb;
            if (full) {
                const float4 v = *reinterpret_cast<const float4*>(src);
                a[0] += v.x; a[1] += v.y; a[2] += v.z; a[3] += v.w;
            } else {
                for (int e = 0; e < 4 && nb + e < p.N; ++e) a[e] += src[e];
            }
        }
        for (int e = 0; e < 4 && nb + e < p.N; ++e) {
            const int n = nb + e;
            float out = dgq_epilogue<PER_M>(p, a[e], m, n, p.alpha[n], p.zw[n], p.gamma[n], PER_M ? p.vn[n] : 0.0f);
            out = dgq_extra(p.ex, out, m, n);
            y[(int64_t)m * p.ldy + n] = dgq_from_float<TOut>(out);
        }
    }
}

template <int WBITS, bool PER_M, typename TOut, int BM, int BN, int NST = 3, int WVN = 2>
static void launch_tile(const GemmBatch& bt, hipStream_t st) {
    const GemmParams& p = bt.p[0];
    constexpr int lds_stages = NST * gemm_stage_bytes(WBITS, BM, BN);
    constexpr int lds_vec = (3 * BM + 4 * BN) * 4;
    constexpr int lds_max = lds_stages + lds_vec + 8192;        // + epilogue vectors + per-chunk scales (<= 2048 chunks)
    // the attribute is per device: one flag per device ordinal (set again by whichever thread gets there first — the
    // call is idempotent, so a benign race at worst repeats it)
    static std::atomic<bool> attr_set[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !attr_set[dev].load(std::memory_order_acquire)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_wxa8_kernel<WBITS, PER_M, TOut, BM, BN, NST, WVN>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
        if (dev >= 0 && dev < 64) attr_set[dev].store(true, std::memory_order_release);
    }
    int maxN = 0, maxM = 0, max_tps = 0;
    for (int i = 0; i < bt.n; ++i) {
        maxN = bt.p[i].N > maxN ? bt.p[i].N : maxN;
        maxM = bt.p[i].M > maxM ? bt.p[i].M : maxM;
        max_tps = bt.p[i].tiles_per_split > max_tps ? bt.p[i].tiles_per_split : max_tps;
    }
    const int lds = lds_stages + lds_vec + (PER_M ? 0 : ((2 * max_tps * 4 + 15) & ~15));
    dim3 grid((maxN + BN - 1) / BN, (maxM + BM - 1) / BM, bt.n > 1 ? bt.n : p.splits), block(256);
    hipLaunchKernelGGL((gemm_wxa8_kernel<WBITS, PER_M, TOut, BM, BN, NST, WVN>), grid, block, lds, st, bt);
}

// Tile shapes the host may pick (BM, BN): W4 {128x128, 128x64, 64x128, 64x64, 32x128, 32x64}; W8 (a secondary
// configuration) carries {128x128, 64x64, 32x64}.

template <int WBITS, bool PER_M, typename TOut>
static int launch_one(const GemmBatch& bt, int bm, int bn, hipStream_t st) {
    const GemmParams& p = bt.p[0];
    const int key = bm * 1000 + bn;
    switch (key) {
        case 128128: launch_tile<WBITS, PER_M, TOut, 128, 128>(bt, st); break;
        case 64064: launch_tile<WBITS, PER_M, TOut, 64, 64>(bt, st); break;   // (1 x 4 waves here: 3.567 -> 3.560 ms, not kept)
        case 32064: {
            // small grids (<= 2 workgroups per CU) with enough K tiles: the 6-stage ring — measured 2 % SLOWER over the SD step's
            // GEMMs (3.65 -> 3.73 ms: the longer prologue wait outweighs the extra tiles in flight); opt-in with DGQ_GEMM_DEEP=1
            static const bool deep_ok = [] { const char* e = getenv("DGQ_GEMM_DEEP"); return e && *e == '1'; }();
            long blocks = 0;
            for (int i = 0; i < bt.n; ++i) blocks += (long)((bt.p[i].M + 31) / 32) * ((bt.p[i].N + 63) / 64) * (bt.n > 1 ? 1 : p.splits);
            static const bool row4 = [] { const char* e = getenv("DGQ_GEMM_WAVES"); return !(e && *e == '2'); }();   // A/B hook
            if (deep_ok && blocks <= 512 && p.tiles_per_split >= 8) launch_tile<WBITS, PER_M, TOut, 32, 64, 6>(bt, st);
            else if (row4 && WBITS == 4) launch_tile<WBITS, PER_M, TOut, 32, 64, 3, 4>(bt, st);
            else launch_tile<WBITS, PER_M, TOut, 32, 64>(bt, st);
            break;
        }
        default:
            if constexpr (WBITS == 4) {
                switch (key) {
                    case 128064: launch_tile<WBITS, PER_M, TOut, 128, 64>(bt, st); break;
                    case 64128: launch_tile<WBITS, PER_M, TOut, 64, 128>(bt, st); break;
                    case 32128: {
                        static const bool row4 = [] { const char* e = getenv("DGQ_GEMM_WAVES"); return !(e && *e == '2'); }();
                        if (row4) launch_tile<WBITS, PER_M, TOut, 32, 128, 3, 4>(bt, st);
                        else launch_tile<WBITS, PER_M, TOut, 32, 128>(bt, st);
                        break;
                    }
                    default: dgq_set_error("dgq_gemm_wxa8: no %dx%d tile", bm, bn); return DGQ_EINVAL;
                }
            } else {
                dgq_set_error("dgq_gemm_wxa8: no %dx%d tile for W8", bm, bn);
                return DGQ_EINVAL;
            }
    }
    if (bt.n == 1 && p.splits > 1) {
        int64_t total = (int64_t)p.M * ((p.N + 3) / 4);
        int g = (int)((total + 255) / 256);
        if (g > 4096) g = 4096;
        hipLaunchKernelGGL((splitk_epilogue_kernel<PER_M, TOut>), dim3(g), dim3(256), 0, st, p);
    }
    return DGQ_OK;
}

template <int WBITS, bool PER_M>
static int launch_gemm(const GemmBatch& p, int bm, int bn, int y_dtype, hipStream_t st) {
    int rc;
    switch (y_dtype) {
        case DGQ_F32: rc = launch_one<WBITS, PER_M, float>(p, bm, bn, st); break;
        case DGQ_F16: rc = launch_one<WBITS, PER_M, __half>(p, bm, bn, st); break;
        case DGQ_BF16: rc = launch_one<WBITS, PER_M, __hip_bfloat16>(p, bm, bn, st); break;
        default: dgq_set_error("dgq_gemm_wxa8: unknown y dtype %d", y_dtype); return DGQ_EINVAL;
    }
    if (rc != DGQ_OK) return rc;
    return dgq_launch_status("dgq_gemm_wxa8");
}

// Launch plan (tile shape + K split).  Rules distilled from the measured sweep of every (tile, split) candidate over the
// SD1.4 / SDXL layer shapes (tools/tile_sweep.py, profiles/r02_gemm_tile_sweep_*.txt; within 2 % of the per-shape optimum
// on the SD step, 14 % better than the analytic model they replace).  What the measurements say:
//   * these GEMMs are latency-bound per K tile, not MFMA-bound: the smallest tile (32x64: 1280 blocks at 8192x320, 640 at
//     2048x640) wins whenever the output is small (M·N <= 3M), because many resident blocks hide each other's DMA latency;
//   * once K is long (>= 24 K tiles) and M large, operand re-reads through L2 dominate (blocks · tiles · (BM·128 + BN·64)
//     bytes at ~12 TB/s): wider tiles (32x128, 64x128) halve the activation re-reads;
//   * large outputs (M·N > 3M) are bound by their own stores: 64x128 (128x128 from 30M outputs on);
//   * a K split pays only when the unsplit grid cannot fill the chip (< 256 blocks) AND K is long (> 60 tiles): slabs
//     cost S·M·N·8 B of traffic plus a combine launch; then S brings the grid to ~480 blocks.
struct GemmPlan { int bm, bn, splits; double t; };
static GemmPlan plan_gemm(int M, int N, int Kp, int w_bits, size_t ws_bytes, bool per_m) {
    const int nk = Kp / BK;
    const double out = (double)M * N;
    GemmPlan pl = {32, 64, 1, 0.0};
    if (out > 3.0e7) {
        pl = {128, 128, 1, 0.0};
    } else if (out > 3.0e6) {
        pl = {64, 128, 1, 0.0};
    } else if (nk > 60) {
        if (M <= 160) pl = {32, 64, 1, 0.0};
        else if (N >= 1280) pl = (M >= 2048) ? GemmPlan{64, 64, 1, 0.0} : GemmPlan{128, 64, 1, 0.0};   // 2048x1280x11520: 57 -> 49 us
        else if (M >= 4096 && N % 128 != 0) pl = {64, 64, 1, 0.0};        // N = 320: 128-wide tiles compute 384 columns
        else pl = {64, 128, 1, 0.0};
        const long grid = (long)((M + pl.bm - 1) / pl.bm) * ((N + pl.bn - 1) / pl.bn);
        if (grid < 256) {
            int s = (int)((480 + grid - 1) / grid);
            if (s > nk / 8) s = nk / 8;
            if (s > 16) s = 16;
            while (s > 1 && (double)s * out * 4.0 > (double)ws_bytes) --s;
            pl.splits = s < 1 ? 1 : s;
        }
    } else if (M >= 4096 && nk >= 24) {
        pl = (N % 128 != 0) ? GemmPlan{64, 64, 1, 0.0} : GemmPlan{32, 128, 1, 0.0};   // 8192x320x2880: 28 -> 25 us at 64x64
    }
    if (w_bits != 4) {                                   // W8 carries three tile shapes: nearest one
        if (pl.bm == 128 || pl.bn == 128) { pl.bm = 128; pl.bn = 128; }
        else if (pl.bm == 64) { pl.bm = 64; pl.bn = 64; }
        else { pl.bm = 32; pl.bn = 64; }
    }
    return pl;
}

// Development hook: DGQ_GEMM_FORCE="BM,BN,S" overrides the plan (tile sweeps, tools/bench_gemm_sweep.py); read per call.
static bool forced_plan(GemmPlan& pl) {
    const char* e = getenv("DGQ_GEMM_FORCE");
    if (!e || !*e) return false;
    int bm = 0, bn = 0, s = 0;
    if (sscanf(e, "%d,%d,%d", &bm, &bn, &s) != 3) return false;
    pl.bm = bm; pl.bn = bn; pl.splits = s < 1 ? 1 : s;
    return true;
}

extern "C" size_t dgq_gemm_workspace_bytes(int M, int N, int Kp) {
    const GemmPlan a = plan_gemm(M, N, Kp, 4, (size_t)-1, false), b = plan_gemm(M, N, Kp, 4, (size_t)-1, true);
    const int s = a.splits > b.splits ? a.splits : b.splits;
    return s > 1 ? (size_t)s * M * N * sizeof(float) : 0;
}

static int fill_gemm(const dgq_gemm_args_t& a, GemmParams& p) {
    DGQ_CHECK_ARG(a.codes && a.rowsum && a.wpacked && a.alpha && a.zw && a.gamma && a.y, "dgq_gemm_wxa8: null pointer");
    DGQ_CHECK_ARG(a.M > 0 && a.N > 0 && a.Kp > 0 && a.Kp % DGQ_KTILE == 0, "dgq_gemm_wxa8: bad shape M=%d N=%d Kp=%d", a.M, a.N, a.Kp);
    DGQ_CHECK_ARG(a.w_bits == 4 || a.w_bits == 8, "dgq_gemm_wxa8: w_bits=%d unsupported", a.w_bits);
    DGQ_CHECK_ARG(a.rowsum_parts >= 1 && a.rowsum_parts <= 64, "dgq_gemm_wxa8: rowsum_parts=%d", a.rowsum_parts);
    DGQ_CHECK_ARG(a.ldy >= a.N, "dgq_gemm_wxa8: ldy < N");
    DGQ_CHECK_ARG(a.Kp / DGQ_KCHUNK <= 2048, "dgq_gemm_wxa8: Kp=%d too large (max %d)", a.Kp, 2048 * DGQ_KCHUNK);
    DGQ_CHECK_ARG((reinterpret_cast<uintptr_t>(a.codes) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.wpacked) & 15) == 0,
                  "dgq_gemm_wxa8: codes/wpacked must be 16-byte aligned");
    if (a.per_m) {
        DGQ_CHECK_ARG(a.mdelta && a.mzp && a.vn && a.L >= 1, "dgq_gemm_wxa8: per_m needs mdelta/mzp/vn/L");
    } else {
        DGQ_CHECK_ARG(a.cdelta && a.cflush, "dgq_gemm_wxa8: per-K mode needs cdelta/cflush");
    }
    p.codes = a.codes; p.rowsum = a.rowsum; p.rowsum_parts = a.rowsum_parts; p.M = a.M; p.Kp = a.Kp; p.N = a.N;
    p.wpacked = reinterpret_cast<const uint8_t*>(a.wpacked);
    p.cdelta = a.cdelta; p.cflush = a.cflush; p.mdelta = a.mdelta; p.mzp = a.mzp; p.L = a.per_m ? a.L : 1; p.offset = a.offset;
    p.alpha = a.alpha; p.zw = a.zw; p.gamma = a.gamma; p.vn = a.vn; p.y = a.y; p.ldy = a.ldy;
    if (a.extra) {
        p.ex = *a.extra;
        DGQ_CHECK_ARG(p.ex.fq_mode >= 0 && p.ex.fq_mode <= 3, "dgq_gemm_wxa8: bad fq_mode");
        DGQ_CHECK_ARG(p.ex.fq_mode == 0 || (p.ex.fq_delta && p.ex.fq_zp && p.ex.fq_T > 0 && p.ex.fq_D > 0), "dgq_gemm_wxa8: fused quantizer needs tables");
        DGQ_CHECK_ARG(!p.ex.residual || (p.ex.ldr >= a.N && p.ex.res_div >= 1 && p.ex.res_dtype >= DGQ_F32 && p.ex.res_dtype <= DGQ_BF16),
                      "dgq_gemm_wxa8: bad residual descriptor (ldr < N, res_div < 1 or unknown dtype)");
    } else {
        p.ex.residual = nullptr; p.ex.ldr = 0; p.ex.res_div = 1; p.ex.res_dtype = DGQ_F32; p.ex.fq_mode = 0; p.ex.fq_delta = nullptr; p.ex.fq_zp = nullptr;
        p.ex.fq_T = 1; p.ex.fq_D = 1; p.ex.fq_skip = 0; p.ex.fq_qmax = 255.0f;
    }
    p.splits = 1; p.slab = nullptr;
    p.tiles_per_split = a.Kp / BK;
    return DGQ_OK;
}

template <typename B>
static int dispatch_gemm(const B& bt, int w_bits, bool per_m, int bm, int bn, int y_dtype, hipStream_t st) {
    if (w_bits == 4) return per_m ? launch_gemm<4, true>(bt, bm, bn, y_dtype, st) : launch_gemm<4, false>(bt, bm, bn, y_dtype, st);
    return per_m ? launch_gemm<8, true>(bt, bm, bn, y_dtype, st) : launch_gemm<8, false>(bt, bm, bn, y_dtype, st);
}

// 2..8 problems in one launch: same weight bits, scale mode, output dtype and launch plan (tile shape) — the plan of the
// FIRST problem is used and must not split K (batches exist for small, launch-bound layers); DGQ_EINVAL otherwise.
extern "C" int dgq_gemm_wxa8_batch(int n, const dgq_gemm_args_t* args, void* stream) {
    DGQ_CHECK_ARG(args && n >= 1 && n <= DGQ_GEMM_BATCH, "dgq_gemm_wxa8_batch: n=%d (1..%d)", n, DGQ_GEMM_BATCH);
    GemmBatch bt;
    bt.n = n;
    for (int i = 0; i < n; ++i) {
        const int rc = fill_gemm(args[i], bt.p[i]);
        if (rc != DGQ_OK) return rc;
        DGQ_CHECK_ARG(args[i].w_bits == args[0].w_bits && (args[i].per_m != 0) == (args[0].per_m != 0) && args[i].y_dtype == args[0].y_dtype,
                      "dgq_gemm_wxa8_batch: problem %d differs from problem 0 in weight bits / scale mode / output dtype", i);
    }
    const dgq_gemm_args_t& a0 = args[0];
    GemmPlan pl = plan_gemm(a0.M, a0.N, a0.Kp, a0.w_bits, 0, a0.per_m != 0);
    forced_plan(pl);
    pl.splits = 1;
    return dispatch_gemm(bt, a0.w_bits, a0.per_m != 0, pl.bm, pl.bn, a0.y_dtype, (hipStream_t)stream);
}

extern "C" int dgq_gemm_wxa8(const int8_t* codes, const float* rowsum, int rowsum_parts, int M, int Kp,
                             const void* wpacked, int w_bits, int N,
                             int per_m, const float* cdelta, const uint8_t* cflush,
                             const float* mdelta, const float* mzp, int L, float offset,
                             const float* alpha, const float* zw, const float* gamma, const float* vn,
                             void* y, int y_dtype, int ldy, void* workspace, size_t workspace_bytes,
                             const dgq_gemm_extra_t* extra, void* stream) {
    DGQ_CHECK_ARG(!workspace || (reinterpret_cast<uintptr_t>(workspace) & 15) == 0, "dgq_gemm_wxa8: workspace alignment");
    dgq_gemm_args_t a;
    a.codes = codes; a.rowsum = rowsum; a.rowsum_parts = rowsum_parts; a.M = M; a.Kp = Kp; a.wpacked = wpacked; a.w_bits = w_bits; a.N = N;
    a.per_m = per_m; a.cdelta = cdelta; a.cflush = cflush; a.mdelta = mdelta; a.mzp = mzp; a.L = L; a.offset = offset;
    a.alpha = alpha; a.zw = zw; a.gamma = gamma; a.vn = vn; a.y = y; a.y_dtype = y_dtype; a.ldy = ldy; a.extra = extra;
    GemmBatch bt;
    bt.n = 1;
    GemmParams& p = bt.p[0];
    const int rc = fill_gemm(a, p);
    if (rc != DGQ_OK) return rc;
    GemmPlan pl = plan_gemm(M, N, Kp, w_bits, workspace ? workspace_bytes : 0, per_m != 0);
    if (forced_plan(pl)) {
        DGQ_CHECK_ARG(pl.splits == 1 || (workspace && (size_t)pl.splits * M * N * 4 <= workspace_bytes),
                      "dgq_gemm_wxa8: DGQ_GEMM_FORCE split does not fit the workspace");
    }
    p.splits = pl.splits;
    p.slab = p.splits > 1 ? reinterpret_cast<float*>(workspace) : nullptr;
    const int nk = Kp / BK;
    p.tiles_per_split = (nk + p.splits - 1) / p.splits;
    p.splits = (nk + p.tiles_per_split - 1) / p.tiles_per_split;      // no empty split
    if (p.splits == 1) p.slab = nullptr;
    return dispatch_gemm(bt, w_bits, per_m != 0, pl.bm, pl.bn, y_dtype, (hipStream_t)stream);
}
