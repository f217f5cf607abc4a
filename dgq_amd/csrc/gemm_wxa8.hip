// W4A8 / W8A8 GEMM on V_MFMA_I32_32X32X32_I8 with DGQ per-group dequantisation fused in.
//
//   A  = int8 activation codes [M][Kp]  (from dgq_quant_act; K already permuted so that each DGQ group is a run
//        of 32-wide chunks)
//   W  = int4 packed [N][Kp/2] (layout 1 of dgq_pack_w4)  or int8 [N][Kp]
//   Y  = fp [M][N]
//
// One 32-wide chunk = one MFMA K slice, scaled by a single δ.  Round 3 moved the kernel from MFMA_I32_16X16X64 (groups padded
// to 64: K = 320 in 16 groups ran Kp = 1024) to the 32x32x32 shape: same operand bytes per MFMA cycle, half the group
// padding, and a higher sustained rate on this chip (tools/mfma_rate.hip, profiles/r03_mfma_rate_int8_shapes.txt:
// 4.04 against 3.13 Pop/s back to back; the 16x16x32 / 32x32x16 forms of gfx942 run at half rate).
//
// Tiling (wave64, gfx950): BM x BN block tile, BK = 128 (four chunks), WVM x WVN x WVK waves: each wave owns a
// (BM/WVM) x (BN/WVN) output tile of TM x TN MFMA tiles and every WVK-th chunk of a K tile (WVK = 2 splits K INSIDE the
// workgroup for the 32-row tiles, whose four waves would otherwise not find four 32x32 tiles; the two partial tiles meet in
// LDS in the epilogue).  4·TM·TN·4 int32 accumulators + (per-K mode) as many fp32 accumulators per lane.
//
// Operand staging is LDS-DMA only (global_load_lds_dwordx4: no staging VGPRs, no ds_write — ds_write_b128 runs at
// ~79 B/clk/CU and was the bottleneck of the register-staged version): a 3-stage LDS ring, tile t+2 is issued
// before the MFMAs of tile t, a counted s_waitcnt vmcnt leaves it in flight across the (raw) barrier.
// LDS images are lane-linear per DMA instruction, so the bank swizzle is applied to the per-lane SOURCE address
// and to the ds_read address (cdna guide rule 21):
//   A / int8-W tile: 128-byte rows, 16-byte piece p stored at p ^ ((row>>1)&7)       -> conflict-free ds_read_b128
//   int4-W tile    : 64-byte rows (packed), 16-byte piece p (= one 32-chunk) stored at p ^ ((row>>2)&3); a lane reads the
//                    8-byte half of its K half — the DMA cannot move 8-byte units, so rows with bit 4 set carry their two
//                    halves exchanged in HBM already (dgq_pack_w4 layout 1): 32 lanes then cover all 64 banks  -> conflict-
//                    free ds_read_b64
// int4 weights stay packed in LDS (half the LDS bytes) and are widened to int8 in registers right after the
// ds_read: (w & 0x0F0F0F0F), ((w>>4) & 0x0F0F0F0F) — the dgq_pack_w4 nibble order makes those 4 consecutive k each.
//
// Small grids use deterministic split-K: grid.z slices of the K-tile range write fp32 partial slabs [S][M][N] to a
// caller-provided workspace; splitk_epilogue_kernel sums them in a fixed order and applies the dequantisation epilogue
// (no float atomics: results are bit-reproducible).
#include "gemm_tile.h"
#include "quant_common.h"

DGQ_DIAG_BUFFER(gemm)

// WVM x WVN x WVK waves; NST ring stages; ACCS int32 accumulator sets per wave (per-K mode only).  With ACCS = 2 consecutive
// chunks of a wave alternate between two accumulator sets, each with its own running total, and a chunk's flush is issued
// AFTER the MFMAs of the wave's next chunk: on the small tiles (one or two MFMAs per chunk) the MFMA latency and the flush
// VALU of one chunk then overlap the next chunk's MFMAs instead of sitting between them.
// CONV: the A operand is the implicit im2col of an int8 NHWC code tensor (dgq_gemm_conv_t: scalar-δ convolutions) — the same LDS
// image, filled from per-lane source addresses that walk (tap, channel) instead of a materialised [M][Kp] row.
template <int WBITS, bool PER_M, typename TOut, int BM, int BN, int WVM, int WVN, int WVK, int NST, int ACCS, bool CONV = false>
__global__ __launch_bounds__(64 * WVM * WVN * WVK, gemm_waves_per_simd(WBITS, BM, BN, WVM * WVN * WVK, NST))
void gemm_wxa8_kernel(GemmBatch bt) {
    const GemmParams& p = bt.p[bt.n > 1 ? blockIdx.z : 0];
    const int zsplit = bt.n > 1 ? 0 : blockIdx.z;
    gemm_prefetch_params(p);
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
    DGQ_DIAG_DECL
    DGQ_STAMP(0); DGQ_STAMP_REAL(1); DGQ_STAMP_WHERE(2);
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int STAGES = NST;
    constexpr int NW = WVM * WVN * WVK, NT = 64 * NW;
    constexpr int WM = BM / WVM, WN = BN / WVN;            // per-wave output tile
    static_assert(WM % 32 == 0 && WN % 32 == 0, "wave tile: multiples of the 32x32 MFMA tile");
    constexpr int TM = WM / 32, TN = WN / 32;              // MFMA tiles per wave
    constexpr int MYCH = NCH / WVK;                        // chunks per wave per K tile
    static_assert(WVK == 1 || WVK == 2, "K split inside the workgroup: 1 or 2");
    static_assert(ACCS == 1 || (ACCS == 2 && !PER_M), "two accumulator sets: per-K mode only");
    static_assert(NST >= 3 && NST <= 6, "ring depth");
    constexpr int A_BYTES = BM * BK;
    constexpr int W_ROW = (WBITS == 4) ? BK / 2 : BK;      // bytes per n-row per stage
    constexpr int W_BYTES = BN * W_ROW;
    constexpr int STAGE_BYTES = A_BYTES + W_BYTES;
    static_assert(A_BYTES % (1024 * NW) == 0 && W_BYTES % (1024 * NW) == 0, "every wave stages whole 1-KiB pieces of each operand");
    constexpr int A_DMA = A_BYTES / 1024 / NW;             // DMA instructions per wave per tile (1 KiB each)
    constexpr int W_DMA = W_BYTES / 1024 / NW;
    constexpr int DMA_PER_TILE = A_DMA + W_DMA;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_k = wid / (WVM * WVN), wave_m = (wid / WVN) % WVM, wave_n = wid % WVN;
    const int n0 = tile_n * BN;
    const int m0 = tile_m * BM;
    const int nk_total = p.Kp / BK;
    const int kt_begin = zsplit * p.tiles_per_split;
    const int kt_end = min(nk_total, kt_begin + p.tiles_per_split);
    const int nk = kt_end - kt_begin;

    // per-lane global source pointers of this wave's DMA pieces (k offset added per tile)
    const int8_t* a_src[A_DMA];
    // CONV: per piece the lane's 16-byte granule walks the K order kp = tap·C + c of its output position: (channel offset, tap
    // row, tap column, k) advance by one K tile per issue_tile call (tiles are issued in increasing order)
    int cv_cc[A_DMA], cv_dh[A_DMA], cv_dw[A_DMA], cv_k[A_DMA], cv_h[A_DMA], cv_w[A_DMA];
    const int8_t* cv_img[A_DMA];
    bool cv_in[A_DMA];
#pragma unroll
    for (int i = 0; i < A_DMA; ++i) {
        const int blk = wid * A_DMA + i;                   // 1 KiB = 8 rows of 128 B
        const int row = blk * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        const int m = min(m0 + row, p.M - 1);
        if constexpr (CONV) {
            const int L = p.cv.Ho * p.cv.Wo;
            const int b = m / L, l = m - b * L;
            const int ho = l / p.cv.Wo, wo = l - ho * p.cv.Wo;
            cv_h[i] = ho * p.cv.stride - p.cv.pad;
            cv_w[i] = wo * p.cv.stride - p.cv.pad;
            a_src[i] = p.cv.codes_in + (int64_t)b * p.cv.H * p.cv.W * p.cv.ldc;      // the image of this row
            const int k = kt_begin * BK + 16 * c;
            const int tap = k / p.cv.C;
            cv_k[i] = k;
            cv_cc[i] = k - tap * p.cv.C;
            cv_dh[i] = tap / p.cv.kw;
            cv_dw[i] = tap - cv_dh[i] * p.cv.kw;
            cv_img[i] = a_src[i];
            const int hi = cv_h[i] + cv_dh[i], wi = cv_w[i] + cv_dw[i];
            cv_in[i] = (unsigned)hi < (unsigned)p.cv.H && (unsigned)wi < (unsigned)p.cv.W;
            a_src[i] = cv_img[i] + ((int64_t)hi * p.cv.W + wi) * p.cv.ldc + cv_cc[i];          // the lane's granule in the current tap
        } else {
            a_src[i] = p.codes + (int64_t)m * p.Kp + 16 * c;
        }
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
        for (int i = 0; i < A_DMA; ++i) {
            if constexpr (CONV) {
                const int8_t* src = cv_in[i] ? a_src[i] : p.cv.fill;                 // outside the image: the code of 0.0
                if (cv_k[i] >= p.cv.C * p.cv.kh * p.cv.kw) src = p.cv.fill + 16;       // K padding: zero codes
                glds16(src, __builtin_amdgcn_readfirstlane(sa + (wid * A_DMA + i) * 1024));
                // one K tile further: inside a tap the granule just moves 128 bytes on; a tap boundary (every C/128 tiles; for
                // C % 128 == 0 the whole wave at once) re-derives the pixel — out of line, the common tile falls through
                cv_k[i] += BK;
                cv_cc[i] += BK;
                a_src[i] += BK;
                if (__builtin_expect(__any(cv_cc[i] >= p.cv.C), 0)) {
                    if (cv_cc[i] >= p.cv.C) {
                        const int tap = cv_k[i] / p.cv.C;
                        cv_cc[i] = cv_k[i] - tap * p.cv.C;
                        cv_dh[i] = tap / p.cv.kw;
                        cv_dw[i] = tap - cv_dh[i] * p.cv.kw;
                        const int hi = cv_h[i] + cv_dh[i], wi = cv_w[i] + cv_dw[i];
                        cv_in[i] = (unsigned)hi < (unsigned)p.cv.H && (unsigned)wi < (unsigned)p.cv.W;
                        a_src[i] = cv_img[i] + ((int64_t)hi * p.cv.W + wi) * p.cv.ldc + cv_cc[i];
                    }
                }
            } else {
                glds16(a_src[i] + ka, __builtin_amdgcn_readfirstlane(sa + (wid * A_DMA + i) * 1024));
            }
        }
#pragma unroll
        for (int i = 0; i < W_DMA; ++i)
            glds16(w_src[i] + kw, __builtin_amdgcn_readfirstlane(sw + (wid * W_DMA + i) * 1024));
    };

    v16i acc[ACCS][TM][TN];
    v16f accf[TM][TN];
    constexpr bool BIASED = !PER_M && WBITS == 4;           // per-K W4: totals carry DGQ_ACC_BIAS_I (gemm_device.h)
    constexpr int ACC0 = BIASED ? DGQ_ACC_BIAS_I : 0;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
#pragma unroll
                for (int a = 0; a < ACCS; ++a) acc[a][i][j][r] = ACC0;
                accf[i][j][r] = 0.0f;
            }
        }

    // MFMA_I32_32X32X32_I8 operands: lane l holds row (A) / column (B) l & 31 and the 16 k of half l >> 5 of the chunk
    const int lr = lane & 31, hh = lane >> 5;
    // ds_read byte offsets inside a stage, per chunk of this wave (chunk ci of the wave = chunk ci·WVK + wave_k of the tile)
    int a_off[TM][MYCH], w_off[TN][MYCH];
#pragma unroll
    for (int ci = 0; ci < MYCH; ++ci) {
        const int cg = ci * WVK + wave_k;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row = wave_m * WM + i * 32 + lr;
            a_off[i][ci] = row * BK + (((2 * cg + hh) ^ ((row >> 1) & 7)) << 4);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int row = wave_n * WN + j * 32 + lr;
            if (WBITS == 4) w_off[j][ci] = row * W_ROW + ((cg ^ ((row >> 2) & 3)) << 4) + ((hh ^ ((row >> 4) & 1)) << 3);
            else w_off[j][ci] = row * BK + (((2 * cg + hh) ^ ((row >> 1) & 7)) << 4);
        }
    }

    // ---- prologue.  Everything the epilogue and the flushes need is staged into LDS once: per-row and per-column
    // dequantisation vectors, per-chunk flush coefficients.  The global loads for it are issued FIRST (asm loads the compiler
    // does not count), then NST−1 tiles of LDS-DMA; one counted wait then retires the table loads and tile 0 only and leaves
    // the younger tiles in flight.  All loads are unconditional at clamped indices: a load whose address or predicate depended
    // on another load's result cost a full round trip each (four such chains made a per-K launch 3-4 us longer than its
    // per-M twin).
    float* vtab = reinterpret_cast<float*>(smem + STAGES * STAGE_BYTES);   // [3][BM]: R0 R1 R2 | [4][BN]: alpha zw gamma vn
    float* vcol = vtab + 3 * BM;
    float* ctab = vcol + 4 * BN;                                           // [WVK][nk·MYCH] flush coefficients | [nk] clear flags
    static_assert(BM <= NT && BN <= NT, "one row / column of the epilogue vectors per thread");
    // Summation by parts: with T the RUNNING int32 total of a chunk sequence and δ_c the scale of chunk c's group,
    // Σ_groups δ_g·P_g = Σ_c (δ_c − δ_next(c))·T_c, δ := 0 past the sequence.  The coefficient is non-zero exactly at group
    // ends, so a flush is cvt + fma per accumulator register.  A wave runs ACCS sequences (chunk ci of a tile belongs to
    // sequence ci mod ACCS), the workgroup WVK·ACCS.  A clear mark (cflush == 2) on the LAST chunk of a K tile asks for the
    // totals to be cleared behind that tile (|T| < 2^24: float(T) exact, dgq_amd/plan.py:mark_clears): the coefficient of each
    // sequence's last chunk in the tile is then the full δ_c.  Marks on other chunks are not honoured (the planner places none).
    // Table entries e < n_coef: coefficient of chunk i of wave kq;  e >= n_coef: clear flag of tile e − n_coef.
    const int per_wave = nk * MYCH, n_coef = PER_M ? 0 : WVK * per_wave, n_tab = PER_M ? 0 : n_coef + nk;
    struct CoefIdx { int g, gn, tl; bool is_coef, seq_last, tile_end, not_last_tile; };
    auto coef_idx = [&](int e) {
        CoefIdx x;
        x.is_coef = e < n_coef;
        const int ec = x.is_coef ? e : 0;
        const int kq = ec / per_wave, i = ec - kq * per_wave;
        const int tc = i / MYCH, ci = i - tc * MYCH;
        const int t = x.is_coef ? tc : e - n_coef;
        x.g = (kt_begin + tc) * NCH + ci * WVK + kq;
        x.tile_end = (ci + ACCS >= MYCH);                                    // last chunk of its sequence in the tile
        x.seq_last = x.tile_end && tc == nk - 1;
        x.gn = min(x.tile_end ? (kt_begin + tc + 1) * NCH + (ci + ACCS - MYCH) * WVK + kq : x.g + ACCS * WVK, nk_total * NCH - 1);
        x.tl = (kt_begin + t) * NCH + NCH - 1;
        x.not_last_tile = t + 1 < nk;
        return x;
    };
    auto coef_val = [&](const CoefIdx& x, float d, float dn, uint32_t cf) {
        const bool clr = (cf & 0xFF) == 2;
        const float coef = (x.seq_last || (x.tile_end && clr)) ? d : d - dn;
        const float flag = (x.not_last_tile && clr) ? 1.0f : 0.0f;
        return x.is_coef ? coef : flag;
    };
    {
        const bool final_ep = (p.splits == 1);
        const bool has_row = final_ep && tid < BM, has_col = final_ep && tid < BN;
        const int m = min(m0 + tid, p.M - 1), n = min(n0 + tid, p.N - 1);
        const int li = PER_M ? m % p.L : 0;
        float md = 1.0f, mz = 0.0f, c_vn = 0.0f, c_d = 0.0f, c_dn = 0.0f;
        uint32_t c_cf = 0;
        float rs = gload_f32(p.rowsum + m);
        if constexpr (PER_M) { md = gload_f32(p.mdelta + li); mz = gload_f32(p.mzp + li); }
        float c_al = gload_f32(p.alpha + n), c_zw = gload_f32(p.zw + n), c_ga = gload_f32(p.gamma + n);
        if constexpr (PER_M) c_vn = gload_f32(p.vn + n);
        CoefIdx cx = {};
        if constexpr (!PER_M) {
            cx = coef_idx(min(tid, n_tab - 1));
            c_d = gload_f32(p.cdelta + cx.g); c_dn = gload_f32(p.cdelta + cx.gn); c_cf = gload_u8(p.cflush + cx.tl);
        }
        int issued = 0;
#pragma unroll
        for (int i = 0; i < STAGES - 1; ++i)
            if (i < nk) { issue_tile(kt_begin + i, i); ++issued; }
        // table loads + tile 0 done; up to NST − 2 younger tiles stay in flight
        DGQ_STAMP(3);
        wait_ring<DMA_PER_TILE, STAGES - 2>(issued - 1);
        DGQ_STAMP(4);
        // the loaded registers become defined HERE for the compiler (volatile asm statements keep their order)
        if (PER_M) asm volatile("" : "+v"(rs), "+v"(md), "+v"(mz), "+v"(c_al), "+v"(c_zw), "+v"(c_ga), "+v"(c_vn));
        else asm volatile("" : "+v"(rs), "+v"(c_al), "+v"(c_zw), "+v"(c_ga), "+v"(c_d), "+v"(c_dn), "+v"(c_cf));
        __builtin_amdgcn_sched_barrier(0);
        if (has_row) {
            for (int j = 1; j < p.rowsum_parts; ++j) rs += p.rowsum[(int64_t)j * p.M + m];     // K-split quantise passes only
            float r0 = 1.0f, r1 = rs, r2 = 0.0f;
            if (PER_M) { r0 = md; r1 = md * rs; r2 = md * (p.offset - mz); }
            vtab[tid] = r0; vtab[BM + tid] = r1; vtab[2 * BM + tid] = r2;
        }
        if (has_col) {
            vcol[tid] = c_al; vcol[BN + tid] = c_zw; vcol[2 * BN + tid] = c_ga; vcol[3 * BN + tid] = c_vn;
        }
        if constexpr (!PER_M) {
            if (tid < n_tab) ctab[tid] = coef_val(cx, c_d, c_dn, c_cf);
            for (int e = tid + NT; e < n_tab; e += NT) {    // long K only (ordinary loads: they drain the ring once)
                const CoefIdx x = coef_idx(e);
                ctab[e] = coef_val(x, p.cdelta[x.g], p.cdelta[x.gn], p.cflush[x.tl]);
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the staging stores
    __builtin_amdgcn_s_barrier();

    // ring invariant at the top of iteration t: tiles t .. t+STAGES-2 are issued, tile t has landed.
    // The fragment loads are software-pipelined one chunk ahead of the MFMAs that consume them: while the MFMAs of a chunk
    // run, the ds_reads of the next one (or of the next tile's first chunk, after the barrier) are in flight.  int4 stays
    // packed in the fragment registers and is widened right before its MFMAs, so the load itself has no consumer until then.
    typedef typename std::conditional<WBITS == 4, uint2, v4i>::type wfrag_t;
    auto load_frags = [&](const uint8_t* sa, const uint8_t* sw, int ci, v4i (&af)[TM], wfrag_t (&wf)[TN]) {
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const v4i*>(sa + a_off[i][ci]);
#pragma unroll
        for (int j = 0; j < TN; ++j) wf[j] = *reinterpret_cast<const wfrag_t*>(sw + w_off[j][ci]);
    };
    const float* ctw = ctab + wave_k * (nk * MYCH);
    const float* tclr = ctab + WVK * (nk * MYCH);
    typedef float cvec_t __attribute__((ext_vector_type(MYCH)));
    auto mfma_chunk = [&](const v4i (&af)[TM], const wfrag_t (&wf)[TN], v16i (&ac)[TM][TN]) {
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
                ac[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[i], bf[j], ac[i][j], 0, 0, 0);
    };
    // accf += coef · float(running total); coef is wave-uniform and 0 inside a group (nothing to add)
    auto flush = [&](const v16i (&ac)[TM][TN], float coef) {
        const float sc = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, coef)));
        if (sc != 0.0f) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) accf[i][j][r] = __builtin_fmaf(sc, dgq_total_to_float<BIASED>(ac[i][j][r]), accf[i][j][r]);
                }
        }
    };
    // one chunk step: MFMAs of chunk ci into its accumulator set, then the flush that is due — the chunk's own (ACCS = 1) or,
    // with two sets, the PREVIOUS chunk's (other set), whose MFMAs have had a whole step to finish
    float pend = 0.0f;                                      // ACCS = 2: coefficient of the chunk whose flush is pending
    auto step = [&](const v4i (&af)[TM], const wfrag_t (&wf)[TN], int ci, float coef) {
        if constexpr (PER_M) {
            mfma_chunk(af, wf, acc[0]);
        } else if constexpr (ACCS == 1) {
            mfma_chunk(af, wf, acc[0]);
            flush(acc[0], coef);
        } else {
            if (ci & 1) { mfma_chunk(af, wf, acc[1]); flush(acc[0], pend); }
            else { mfma_chunk(af, wf, acc[0]); flush(acc[1], pend); }
            pend = coef;
        }
    };
    v4i af0[TM], af1[TM];
    wfrag_t wf0[TN], wf1[TN];
    int stage = 0, istage = STAGES - 1;
    DGQ_STAMP(5);
    load_frags(smem, smem + A_BYTES, 0, af0, wf0);
    for (int t = 0; t < nk; ++t) {
        if (t + STAGES - 1 < nk) issue_tile(kt_begin + t + STAGES - 1, istage);
        // this tile's flush coefficients and clear flag: two broadcast LDS reads issued FIRST, so that the flush decisions
        // below never wait for the fragment reads queued behind them (LDS returns in order)
        cvec_t cq;
        float tc = 0.0f;
        if (!PER_M) {
            cq = *reinterpret_cast<const cvec_t*>(ctw + t * MYCH);
            tc = tclr[t];
            __builtin_amdgcn_sched_barrier(0);
        }
        const uint8_t* sa = smem + stage * STAGE_BYTES;
#pragma unroll
        for (int ci = 0; ci + 1 < MYCH; ++ci) {
            if ((ci & 1) == 0) {
                load_frags(sa, sa + A_BYTES, ci + 1, af1, wf1);
                step(af0, wf0, ci, PER_M ? 0.0f : cq[ci]);
            } else {
                load_frags(sa, sa + A_BYTES, ci + 1, af0, wf0);
                step(af1, wf1, ci, PER_M ? 0.0f : cq[ci]);
            }
        }
        // tile t+1 must have landed (this wave's pieces) before anyone reads it; the STAGES-2 younger tiles may stay
        // in flight (vmcnt counts the wave's DMA instructions in issue order).  lgkmcnt(0): this wave's reads of tile t
        // are complete, so after the barrier its stage may be overwritten.
        DGQ_STAMP_NOW(dg_w0);
        wait_ring<DMA_PER_TILE, STAGES - 2>(min(STAGES - 2, nk - 2 - t));
        DGQ_STAMP_ACC(12, dg_w0);
        __builtin_amdgcn_s_barrier();
        DGQ_STAMP_ACC(7, dg_w0);
        stage = (stage + 1 == STAGES) ? 0 : stage + 1;
        istage = (istage + 1 == STAGES) ? 0 : istage + 1;
        if (t + 1 < nk) {
            const uint8_t* sn = smem + stage * STAGE_BYTES;
            load_frags(sn, sn + A_BYTES, 0, af0, wf0);
        }
        step(af1, wf1, MYCH - 1, PER_M ? 0.0f : cq[MYCH - 1]);   // MYCH is even: the last chunk of a tile sits in fragment set 1
        if (!PER_M) {
            if (__builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, tc)) != 0) {     // rare: a segment of running totals ends
                if constexpr (ACCS == 2) { flush(acc[1], pend); pend = 0.0f; }
#pragma unroll
                for (int a = 0; a < ACCS; ++a)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
#pragma unroll
                            for (int r = 0; r < 16; ++r) acc[a][i][j][r] = ACC0;
            }
        }
    }
    if constexpr (!PER_M && ACCS == 2) flush(acc[1], pend);
    DGQ_STAMP(6);

    // epilogue (gemm_tile.h): the wave tiles are transposed through the now idle ring and stored 16 bytes per lane
    gemm_store_tile<PER_M, TOut, BM, BN, WVM, WVN, WVK, STAGES * STAGE_BYTES, TM, TN>(p, zsplit, smem, vtab, vcol, wid, lane, wave_m, wave_n,
                                                                                      wave_k, m0, n0, acc[0], accf DGQ_DIAG_ARG);
    DGQ_STAMP(9);
    DGQ_DIAG_DRAIN();
    DGQ_STAMP(10); DGQ_STAMP_REAL(11);
#ifdef DGQ_DIAG
    dg.t[13] = (unsigned long long)nk | ((unsigned long long)tile_m << 16) | ((unsigned long long)tile_n << 32) | ((unsigned long long)wid << 48);
#endif
    DGQ_DIAG_FLUSH(gemm, NW, wid, lane);
}

// Sum of the split-K slabs of one (row, 4 columns) segment, in slab order.  R loads go out per round (a load per iteration of a
// runtime-bounded loop is waited for before the next is issued: S dependent round trips); the launch picks R >= S where it can (one
// round: the wide conv splits of the 8x8 / 16x16 levels have 8-16 slabs and spent 3-4 dependent rounds here).
template <int R>
__device__ __forceinline__ void splitk_sum_slabs(const float* base, int64_t slab_stride, int splits, float (&a)[4]) {
    for (int s0 = 0; s0 < splits; s0 += R) {
        float4 v[R];
#pragma unroll
        for (int j = 0; j < R; ++j) v[j] = *reinterpret_cast<const float4*>(base + (int64_t)min(s0 + j, splits - 1) * slab_stride);
#pragma unroll
        for (int j = 0; j < R; ++j)
            if (s0 + j < splits) { a[0] += v[j].x; a[1] += v[j].y; a[2] += v[j].z; a[3] += v[j].w; }
    }
}
__device__ __forceinline__ void splitk_sum(const float* base, int64_t slab_stride, int splits, float (&a)[4]) {
    if (splits <= 4) splitk_sum_slabs<4>(base, slab_stride, splits, a);          // (uniform branches: splits is a kernel argument)
    else if (splits <= 8) splitk_sum_slabs<8>(base, slab_stride, splits, a);
    else splitk_sum_slabs<16>(base, slab_stride, splits, a);
}

// Deterministic split-K combine + dequantisation epilogue: one thread per 4 consecutive n.
template <bool PER_M, typename TOut>
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(GemmParams p) {
    const int n4 = (p.N + 3) / 4;
    TOut* y = reinterpret_cast<TOut*>(p.y);
    const int64_t slab_stride = (int64_t)p.M * p.N;
    // one output row segment (4 consecutive n): slabs summed in a fixed order, dequantised, stored; returns the values AS STORED.
    // Full, 16-byte aligned segments (the layers' case) take every operand as ONE vector load — the column constants, the residual —
    // form the row constants once and leave as one 16-byte store; before, each of the four elements re-read the row sums and its
    // own column constants and left as a 4-byte store (8-14 us per combine launch for 3 us of slab traffic).
    const bool vec4 = ((p.N & 3) == 0) && ((p.ldy * (int)sizeof(TOut)) % 16 == 0) && ((reinterpret_cast<uintptr_t>(p.y) & 15) == 0) &&
                      (p.ex.y2 == nullptr || (((p.ex.ldy2 * (int)sizeof(TOut)) % 16 == 0) && (reinterpret_cast<uintptr_t>(p.ex.y2) & 15) == 0)) &&
                      p.ex.fq_mode == 0 && !p.ex.geglu &&
                      (((reinterpret_cast<uintptr_t>(p.alpha) | reinterpret_cast<uintptr_t>(p.zw) | reinterpret_cast<uintptr_t>(p.gamma) |
                         (PER_M ? reinterpret_cast<uintptr_t>(p.vn) : 0)) & 15) == 0) &&
                      (p.ex.residual == nullptr || ((p.ex.ldr & 3) == 0 && (reinterpret_cast<uintptr_t>(p.ex.residual) & (p.ex.res_dtype == DGQ_F32 ? 15 : 7)) == 0));
    auto row4 = [&](int m, int nb, float (&val)[4], bool live) {      // live == false: compute (the values as stored), store nothing
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        const bool full = (nb + 3 < p.N) && ((p.N & 3) == 0);
        if (full) {
            splitk_sum(p.slab + (int64_t)m * p.N + nb, slab_stride, p.splits, a);
        } else {
            for (int s = 0; s < p.splits; ++s) {
                const float* src = p.slab + s * slab_stride + (int64_t)m * p.N + nb;
                for (int e = 0; e < 4 && nb + e < p.N; ++e) a[e] += src[e];
            }
        }
        if (full && vec4) {
            const float4 al = *reinterpret_cast<const float4*>(p.alpha + nb), zw = *reinterpret_cast<const float4*>(p.zw + nb);
            const float4 ga = *reinterpret_cast<const float4*>(p.gamma + nb);
            const float4 vn = PER_M ? *reinterpret_cast<const float4*>(p.vn + nb) : make_float4(0.f, 0.f, 0.f, 0.f);
            float rs = 0.0f;
            for (int j = 0; j < p.rowsum_parts; ++j) rs += p.rowsum[(int64_t)j * p.M + m];
            float r0 = 1.0f, r1 = rs, r2 = 0.0f;
            if (PER_M) {
                const int li = m % p.L;
                const float md = p.mdelta[li], mz = p.mzp[li];
                r0 = md; r1 = md * rs; r2 = md * (p.offset - mz);
            }
            float o[4];
            o[0] = dgq_dequant<PER_M>(a[0], r0, r1, r2, al.x, zw.x, ga.x, vn.x);
            o[1] = dgq_dequant<PER_M>(a[1], r0, r1, r2, al.y, zw.y, ga.y, vn.y);
            o[2] = dgq_dequant<PER_M>(a[2], r0, r1, r2, al.z, zw.z, ga.z, vn.z);
            o[3] = dgq_dequant<PER_M>(a[3], r0, r1, r2, al.w, zw.w, ga.w, vn.w);
            if (p.ex.residual) {
                // (16-bit residuals — the bf16 / fp16 states — as one 8-byte load: they used to drop the whole segment onto the
                // element-by-element path below, 14 us per combine launch against 10)
                const int64_t ri = (int64_t)(m / p.ex.res_div) * p.ex.ldr + nb;
                float r[4];
                if (p.ex.res_dtype == DGQ_F32) load4<float>(reinterpret_cast<const float*>(p.ex.residual) + ri, r);      // (kernel-uniform)
                else if (p.ex.res_dtype == DGQ_BF16) load4<__hip_bfloat16>(reinterpret_cast<const __hip_bfloat16*>(p.ex.residual) + ri, r);
                else load4<__half>(reinterpret_cast<const __half*>(p.ex.residual) + ri, r);
                o[0] += r[0]; o[1] += r[1]; o[2] += r[2]; o[3] += r[3];
            }
            TOut* dst = y + (int64_t)m * p.ldy + nb;
            TOut* dst2 = p.ex.y2 ? reinterpret_cast<TOut*>(p.ex.y2) + (int64_t)m * p.ex.ldy2 + nb : nullptr;     // (ex.y2: a second copy of the rows)
            if (sizeof(TOut) == 4) {
                if (live) *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
                if (live && dst2) *reinterpret_cast<float4*>(dst2) = make_float4(o[0], o[1], o[2], o[3]);
#pragma unroll
                for (int e = 0; e < 4; ++e) val[e] = o[e];
            } else {
                TOut t[4] = {dgq_from_float<TOut>(o[0]), dgq_from_float<TOut>(o[1]), dgq_from_float<TOut>(o[2]), dgq_from_float<TOut>(o[3])};
                if (live) *reinterpret_cast<uint2*>(dst) = *reinterpret_cast<const uint2*>(t);
                if (live && dst2) *reinterpret_cast<uint2*>(dst2) = *reinterpret_cast<const uint2*>(t);
#pragma unroll
                for (int e = 0; e < 4; ++e) val[e] = dgq_to_float(t[e]);
            }
            return;
        }
        for (int e = 0; e < 4 && nb + e < p.N; ++e) {
            const int n = nb + e;
            float out = dgq_epilogue<PER_M>(p, a[e], m, n, p.alpha[n], p.zw[n], p.gamma[n], PER_M ? p.vn[n] : 0.0f);
            out = dgq_extra(p.ex, out, m, n);
            const TOut st = dgq_from_float<TOut>(out);
            if (live) y[(int64_t)m * p.ldy + n] = st;
            if (live && p.ex.y2) reinterpret_cast<TOut*>(p.ex.y2)[(int64_t)m * p.ex.ldy2 + n] = st;
            val[e] = dgq_to_float(st);
        }
    };
    if (p.ex.gn_partial) {
        // GroupNorm partials of the output (as the unsplit GEMM's epilogue writes them: per 16-row block and column the mean
        // and the sum of squared deviations).  Sixteen lanes share a (16-row block, 4 columns) unit, one row each — the
        // parallelism of the plain combine (four rows per lane made it 12 us slower per launch) — and merge their partials
        // with four equal-count xor steps (Chan's formula).  A wave holds 4 adjacent column groups x 16 rows (64 contiguous
        // bytes per row).  M % 16 == 0, N % 4 == 0: host-checked; the grid is a whole number of waves, idle lanes run a clamped
        // unit and do not store.
        const int64_t units = (int64_t)(p.M >> 4) * n4;
        const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
        const int lane = (int)(tid & 63);
        const int qd = lane >> 2;                            // row of the block
        const int64_t unit = (tid >> 6) * 4 + (lane & 3);    // units of a wave: 4 adjacent column groups of one row block
        // (n4 % 4 != 0: the last wave of a row block would straddle into the next block — units are linear over (rb, n4), and
        // the 16 lanes of a unit always agree on it, so that is harmless)
        const bool live = unit < units;
        const int64_t g = live ? unit : units - 1;
        const int rb = (int)(g / n4);
        const int nb = (int)(g - (int64_t)rb * n4) * 4;
        const int m = rb * 16 + qd;
        // the segment through row4 — the plain combine's path: column constants and residual as single vector loads, the row constants
        // formed once, one 16-byte store (this branch used to walk its four elements one by one: per-M launches 12.5-17.5 us against
        // 6.5-10 for the same slabs without partials)
        float mean[4], m2[4] = {0.f, 0.f, 0.f, 0.f};
        row4(m, nb, mean, live);                             // mean[e]: the value as stored
        float cnt = 1.0f;
#pragma unroll
        for (int off = 4; off < 64; off <<= 1) {             // lanes of a unit: lane & 3 fixed, lane >> 2 = row
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float om = __shfl_xor(mean[e], off, 64), o2 = __shfl_xor(m2[e], off, 64);
                const float dd = om - mean[e];
                m2[e] = m2[e] + o2 + dd * dd * (0.5f * cnt);
                mean[e] = 0.5f * (mean[e] + om);
            }
            cnt *= 2.0f;
        }
        if (live && qd == 0) {
            float* q = p.ex.gn_partial + ((int64_t)rb * p.N + nb) * 2;
            *reinterpret_cast<float4*>(q) = make_float4(mean[0], m2[0], mean[1], m2[1]);
            *reinterpret_cast<float4*>(q + 4) = make_float4(mean[2], m2[2], mean[3], m2[3]);
        }
        return;
    }
    const int64_t total = (int64_t)p.M * n4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(i / n4);
        float val[4];
        row4(m, (int)(i - (int64_t)m * n4) * 4, val, true);
    }
}

// gemm_wxa8_big.hip: the 256-row ping-pong kernel (W4, one problem, no K split); the plan names it by bm == 256
int dgq_launch_gemm_big(const GemmBatch& bt, bool per_m, int y_dtype, hipStream_t st);
size_t dgq_gemm_big_lds_bytes(bool per_m, int Kp);
// gemm_panel.hip: the quantise-on-load kernel (the workgroup's 32 rows x the whole K quantised into an LDS panel, weights streamed
// fragment-major into registers); the plan names it by bm = PANEL_BM0 + TM (wave tile rows / 32), bn = column waves per workgroup
#define PANEL_BM0 1000
int dgq_launch_gemm_panel(const GemmBatch& bt, bool per_m, int y_dtype, int tm, int nw, int kw, hipStream_t st);
size_t dgq_gemm_panel_lds_bytes(int tm, int nw, int kw, bool per_m, int tiles);
// gemm_convq.hip: a 3x3 convolution with its activation quantiser inside the launch (act.kh > 1): slab length / LDS bytes, 0 = not taken
int dgq_gemm_convq_plan(int B, int H, int W, int C, int kh, int kw, int stride, int pad, int N, int Kp, int w_bits, bool per_m, int* lds_bytes);
int dgq_launch_gemm_convq(const GemmBatch& bt, bool per_m, int y_dtype, int slab_tiles, int lds, hipStream_t st);

template <bool PER_M, typename TOut>
static void launch_combine(const GemmParams& p, hipStream_t st);
struct GemmPlan { int bm, bn, splits; double t; int kw = 1; bool fuse = false; };   // kw / fuse: the panel kernel's K waves / quantise-on-load

template <int WBITS, bool PER_M, typename TOut, int BM, int BN, int WVM, int WVN, int WVK, int NST>
static void launch_tile(const GemmBatch& bt, hipStream_t st) {
    const GemmParams& p = bt.p[0];
    constexpr int NW = WVM * WVN * WVK;
    // two accumulator sets (flush behind the next chunk's MFMAs) wherever a wave has at most two MFMA tiles; with one set
    // the same per-K launches take 0.4-1.9 µs longer (profiles/r03_gemm_perk_gap.txt)
    constexpr int ACCS = (!PER_M && (BM / WVM / 32) * (BN / WVN / 32) <= 2) ? 2 : 1;
    constexpr int lds_stages = NST * gemm_stage_bytes(WBITS, BM, BN);
    constexpr int lds_vec = (3 * BM + 4 * BN) * 4;
    constexpr int lds_max = lds_stages + lds_vec + 32768;       // + epilogue vectors + per-chunk coefficients (<= 4096 chunks)
    // the attribute is per device: one flag per device ordinal (set again by whichever thread gets there first — the
    // call is idempotent, so a benign race at worst repeats it)
    static std::atomic<bool> attr_set[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !attr_set[dev].load(std::memory_order_acquire)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_wxa8_kernel<WBITS, PER_M, TOut, BM, BN, WVM, WVN, WVK, NST, ACCS>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
        if (dev >= 0 && dev < 64) attr_set[dev].store(true, std::memory_order_release);
    }
    int maxN = 0, maxM = 0, max_tps = 0;
    for (int i = 0; i < bt.n; ++i) {
        maxN = bt.p[i].N > maxN ? bt.p[i].N : maxN;
        maxM = bt.p[i].M > maxM ? bt.p[i].M : maxM;
        max_tps = bt.p[i].tiles_per_split > max_tps ? bt.p[i].tiles_per_split : max_tps;
    }
    const int lds = lds_stages + lds_vec + (PER_M ? 0 : (((NCH + 1) * max_tps * 4 + 15) & ~15));
    dim3 grid((maxN + BN - 1) / BN, (maxM + BM - 1) / BM, bt.n > 1 ? bt.n : p.splits), block(64 * NW);
    if (p.cv.codes_in) {                                 // implicit im2col A operand: four tile shapes carry it (conv_tile)
        if constexpr (WBITS == 4 && PER_M && ((BM == 32 && BN == 64) || (BM == 64 && BN == 64) || (BM == 64 && BN == 128) || (BM == 128 && BN == 128))) {
            static std::atomic<bool> cattr[64];
            if (dev < 0 || dev >= 64 || !cattr[dev].load(std::memory_order_acquire)) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_wxa8_kernel<WBITS, PER_M, TOut, BM, BN, WVM, WVN, WVK, NST, ACCS, true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
                if (dev >= 0 && dev < 64) cattr[dev].store(true, std::memory_order_release);
            }
            hipLaunchKernelGGL((gemm_wxa8_kernel<WBITS, PER_M, TOut, BM, BN, WVM, WVN, WVK, NST, ACCS, true>), grid, block, lds, st, bt);
        }
        return;                                          // (other shapes: refused by dgq_gemm_wxa8 before it gets here)
    }
    hipLaunchKernelGGL((gemm_wxa8_kernel<WBITS, PER_M, TOut, BM, BN, WVM, WVN, WVK, NST, ACCS>), grid, block, lds, st, bt);
}

// Ring depth: 3 stages for every tile shape.  A 6-stage ring (five tiles in flight, counted prologue wait so that only tile 0
// is waited for) measured SLOWER on every SD layer shape in round 3 as in round 2 (tools/gemm_gap.py: 512x1280x1280 per-M
// 6.5 -> 7.2 us, 8192x320x320 7.5 -> 9.1): the K loop of these launches is not paced by the DMA latency.
template <int WBITS, bool PER_M, typename TOut, int BM, int BN, int WVM, int WVN, int WVK>
static void launch_ring(const GemmBatch& bt, hipStream_t st) {
    launch_tile<WBITS, PER_M, TOut, BM, BN, WVM, WVN, WVK, 3>(bt, st);
}

// Tile shapes the host may pick (BM x BN, waves m x n x k):
//   W4: 32x64 (1x2x2), 32x128 (1x4x1), 64x64 (2x2x1), 64x128 (1x4x1: a widened int4 fragment feeds two row tiles),
//       128x64 (2x2x1), 128x128 (2x2x1); bm = 256 names the 256-row ping-pong kernel of gemm_wxa8_big.hip (the 128x256 tile of
//       round 3 — equal to 128x128 on every shape, never planned — is gone)
//   W8 (a secondary configuration): 32x64, 64x64, 128x128
template <int WBITS, bool PER_M, typename TOut>
static int launch_one(const GemmBatch& bt, int bm, int bn, hipStream_t st) {
    const GemmParams& p = bt.p[0];
    const int key = bm * 1000 + bn;
    switch (key) {
        case 128128: launch_ring<WBITS, PER_M, TOut, 128, 128, 2, 2, 1>(bt, st); break;
        case 64064: launch_ring<WBITS, PER_M, TOut, 64, 64, 2, 2, 1>(bt, st); break;
        case 32064: launch_ring<WBITS, PER_M, TOut, 32, 64, 1, 2, 2>(bt, st); break;
        default:
            if constexpr (WBITS == 4) {
                switch (key) {
                    case 128064: launch_ring<WBITS, PER_M, TOut, 128, 64, 2, 2, 1>(bt, st); break;
                    case 64128: launch_ring<WBITS, PER_M, TOut, 64, 128, 1, 4, 1>(bt, st); break;
                    case 32128: launch_ring<WBITS, PER_M, TOut, 32, 128, 1, 4, 1>(bt, st); break;
                    default: dgq_set_error("dgq_gemm_wxa8: no %dx%d tile", bm, bn); return DGQ_EINVAL;
                }
            } else {
                dgq_set_error("dgq_gemm_wxa8: no %dx%d tile for W8", bm, bn);
                return DGQ_EINVAL;
            }
    }
    if (bt.n == 1 && p.splits > 1) launch_combine<PER_M, TOut>(p, st);
    return DGQ_OK;
}

template <bool PER_M, typename TOut>
static void launch_combine(const GemmParams& p, hipStream_t st) {
    int64_t total = (int64_t)p.M * ((p.N + 3) / 4);
    if (p.ex.gn_partial) total = ((int64_t)(p.M / 16) * ((p.N + 3) / 4) + 3) / 4 * 64;      // 4 units per wave
    int g = (int)((total + 255) / 256);
    if (g > 4096 && !p.ex.gn_partial) g = 4096;           // (the partials form has no grid-stride loop: one unit per 4 lanes)
    hipLaunchKernelGGL((splitk_epilogue_kernel<PER_M, TOut>), dim3(g), dim3(256), 0, st, p);
}

template <int WBITS, bool PER_M>
static int launch_gemm(const GemmBatch& p, const GemmPlan& pl, int y_dtype, hipStream_t st) {
    const int bm = pl.bm, bn = pl.bn;
    int rc;
    if (bm == 256) {
        DGQ_CHECK_ARG(WBITS == 4 && p.n == 1 && p.p[0].splits == 1 && dgq_gemm_big_lds_bytes(PER_M, p.p[0].Kp) <= 160 * 1024,
                      "dgq_gemm_wxa8: the 256-row kernel takes one unsplit W4 problem whose tables fit the LDS");
        rc = dgq_launch_gemm_big(p, PER_M, y_dtype, st);
        if (rc != DGQ_OK) return rc;
        return dgq_launch_status("dgq_gemm_wxa8");
    }
    if (bm > PANEL_BM0) {
        const int tm = bm - PANEL_BM0, nw = bn;
        DGQ_CHECK_ARG(WBITS == 4 && pl.fuse, "dgq_gemm_wxa8: the panel kernel is the quantise-on-load form of W4 layers");
        for (int i = 0; i < p.n; ++i) {
            DGQ_CHECK_ARG(p.p[i].wfrag && !p.p[i].cv.codes_in && p.p[i].act.x, "dgq_gemm_wxa8: quantise-on-load needs the fragment-major weights (extra.wfrag) and the activation descriptor (extra.act)");
            const size_t need = dgq_gemm_panel_lds_bytes(tm, nw, pl.kw, PER_M, p.p[i].tiles_per_split);
            DGQ_CHECK_ARG(need > 0 && need <= 160 * 1024, "dgq_gemm_wxa8: no panel configuration TM=%d NW=%d KW=%d for a K extent of %d tiles",
                          tm, nw, pl.kw, p.p[i].tiles_per_split);
            DGQ_CHECK_ARG(p.p[i].splits == 1, "dgq_gemm_wxa8: quantise-on-load takes the whole K extent in one workgroup");
        }
        rc = dgq_launch_gemm_panel(p, PER_M, y_dtype, tm, nw, pl.kw, st);
        if (rc != DGQ_OK) return rc;
        return dgq_launch_status("dgq_gemm_wxa8");
    }
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
// SD1.4 / SDXL layer shapes (tools/tile_sweep.py, profiles/r0*_gemm_tile_sweep_*.txt).  What the measurements say:
//   * these GEMMs are latency-bound per K tile, not MFMA-bound: the smallest tile (32x64: 1280 blocks at 8192x320, 640 at
//     2048x640) wins whenever the output is small (M·N <= 3M), because many resident blocks hide each other's DMA latency;
//   * once K is long (>= 24 K tiles) and M large, operand re-reads through L2 dominate (blocks · tiles · (BM·128 + BN·64)
//     bytes at ~12 TB/s): wider tiles (32x128, 64x128) halve the activation re-reads;
//   * large outputs (M·N > 3M) are bound by their own stores: 64x128 (128x128 from 30M outputs on);
//   * a K split pays only when the unsplit grid cannot fill the chip (< 256 blocks) AND K is long (> 60 tiles): slabs
//     cost S·M·N·8 B of traffic plus a combine launch; then S brings the grid to ~480 blocks.
static GemmPlan plan_gemm(int M, int N, int Kp, int w_bits, size_t ws_bytes, bool per_m, bool allow_big = true) {
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
    // The 256-row ping-pong kernel (gemm_wxa8_big.hip; bm = 256 names it): W4, one unsplit problem.  One workgroup per CU and no
    // overlap between a workgroup's store epilogue and the next one's prologue, so it needs (a) at least one full round of
    // 256x256 (per-M) / 256x128 (per-K) tiles over the 256 CUs with N a whole number of tiles, and (b) enough work per tile:
    // K >= 4096, or — per-M — a wide N (>= 4096).  Measured in the model (SDXL C5, fp32 outputs with residual / GroupNorm
    // epilogues, profiles/r04_c5_gemm_big_vs_tiles.txt): 8192x10240x1280 -12 %, 32768x1280x11520 -2 %; 8192x1280x{1280,5120}
    // (160 tiles) and 32768x640x2560 (2.5 column tiles) lose 35-100 % and stay on the tile family.  Per-K it pays on long K
    // only (8192^3 34 -> 39 %): at K = 1280 with 16 groups the flushes bound either kernel.
    {
        const int bn_big = per_m ? 256 : 128;
        const long tiles = (long)((M + 255) / 256) * ((N + bn_big - 1) / bn_big);
        constexpr long min_tiles = 256;
        if (allow_big && w_bits == 4 && pl.splits == 1 && M >= 2048 && N % bn_big == 0 && tiles >= min_tiles &&
            (nk >= 32 || (per_m && nk >= 8 && N >= 4096)) && dgq_gemm_big_lds_bytes(per_m, Kp) <= 160 * 1024)
            pl = {256, 256, 1, 0.0};
    }
    if (w_bits != 4) {                                   // W8 carries three tile shapes: nearest one
        if (pl.bm == 128 || pl.bn == 128) { pl.bm = 128; pl.bn = 128; }
        else if (pl.bm == 64) { pl.bm = 64; pl.bn = 64; }
        else { pl.bm = 32; pl.bn = 64; }
    }
    return pl;
}

// Quantise-on-load inside the GEMM (dgq_gemm_act_t): the panel kernel's configuration for a Linear / 1x1 layer, or false where the
// layer stays on the two-launch form.  One 32-row tile per workgroup, NW column waves x KW K waves: enough waves to give every SIMD
// two or three (the quantising prologue and the K loop are latency chains), the whole padded K in the LDS panel.
static bool plan_panel_fuse(int M, int N, int Kp, bool per_m, GemmPlan& pl) {
    static const bool on = [] { const char* e = getenv("DGQ_GEMM_FUSE"); return !(e && *e == '0'); }();      // A/B hook
    if (!on) return false;
    const int nk = Kp / BK;
    const long mb = (M + 31) / 32, nt = (N + 31) / 32;
    const int kw = (mb * nt < 2048 && nk >= 4) ? 2 : 1;
    int nw;
    if (kw == 1) nw = (N % 320 == 0) ? 10 : 5;
    else nw = (N % 160 == 0) ? 5 : 4;
    const size_t need = dgq_gemm_panel_lds_bytes(1, nw, kw, per_m, nk);
    if (need == 0 || need > 150 * 1024) return false;
    // every column block of a row block quantises the rows again: the fused form pays while that redundancy is small and the row
    // blocks alone fill the chip (tools/bench_fused.py, profiles/r05_fused_linear_shapes.txt); DGQ_GEMM_FUSE_ALL=1: wherever it fits
    const char* ea = getenv("DGQ_GEMM_FUSE_ALL");        // (read per call: the test suite switches it)
    const bool all = ea && *ea == '1';
    if (!all && ((N + 32 * nw - 1) / (32 * nw) > 2 || M < 2048)) return false;
    pl.bm = PANEL_BM0 + 1; pl.bn = nw; pl.kw = kw; pl.fuse = true; pl.splits = 1;
    return true;
}

extern "C" int dgq_gemm_act_fuses(int M, int N, int K, int Kp, int w_bits, int per_m, int n_problems, int x_dtype, int y_dtype) {
    if (w_bits != 4 || x_dtype != y_dtype || M < 1 || N < 1 || K < 4 || K % 4 != 0 || Kp < K || Kp % BK != 0 || n_problems < 1 || n_problems > DGQ_GEMM_BATCH)
        return 0;
    GemmPlan pl = {32, 64, 1, 0.0};
    return plan_panel_fuse(M, N, Kp, per_m != 0, pl) ? 1 : 0;
}

extern "C" int dgq_gemm_conv_act_fuses(int B, int H, int W, int C, int kh, int kw, int stride, int pad, int N, int Kp, int w_bits, int per_m,
                                       int x_dtype, int y_dtype) {
    static const bool on = [] { const char* e = getenv("DGQ_GEMM_FUSE"); return !(e && *e == '0'); }();      // A/B hook (as dgq_gemm_act_fuses)
    if (!on || x_dtype != y_dtype) return 0;
    return dgq_gemm_convq_plan(B, H, W, C, kh, kw, stride, pad, N, Kp, w_bits, per_m != 0, nullptr) > 0 ? 1 : 0;
}

// implicit-conv launches: four tile shapes carry the CONV addressing.  Measured on the C5 shapes (tools/bench_conv_implicit.py,
// profiles/r04_conv_implicit_shapes.txt): 64x128 is the best or within 3 % of it on every one of them — the address arithmetic is
// per DMA piece, and the 64x64 tile the materialised operand prefers at M = 8192 has half the MFMAs per piece.
static void conv_tile(GemmPlan& pl, int M) {
    const int key = pl.bm * 1000 + pl.bn;
    if (M >= 2048) { pl.bm = 64; pl.bn = 128; return; }
    if (key == 32064 || key == 64064 || key == 64128 || key == 128128) return;
    if (pl.bm >= 128) { pl.bm = 128; pl.bn = 128; }
    else if (pl.bn >= 128) { pl.bm = 64; pl.bn = 128; }
    else { pl.bm = 64; pl.bn = 64; }
}

// rowsum[m] = Σ_k s[m][k] of the unfolded operand = Σ over the taps of the per-pixel sums (C·zero_code for a tap outside the image)
__global__ __launch_bounds__(256) void conv_rowsum_kernel(dgq_gemm_conv_t c, int M, float* rowsum) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    const int L = c.Ho * c.Wo;
    const int b = m / L, l = m - b * L;
    const int ho = l / c.Wo, wo = l - ho * c.Wo;
    const int64_t npix = (int64_t)c.B * c.H * c.W;
    const float* ps = c.pixsum + (int64_t)b * c.H * c.W;
    const float outside = (float)c.C * c.zero_code;
    float s = 0.0f;                                          // exact: integers below 2^24
    for (int dh = 0; dh < c.kh; ++dh)
        for (int dw = 0; dw < c.kw; ++dw) {
            const int hi = ho * c.stride - c.pad + dh, wi = wo * c.stride - c.pad + dw;
            if ((unsigned)hi < (unsigned)c.H && (unsigned)wi < (unsigned)c.W) {
                for (int q = 0; q < c.pixsum_parts; ++q) s += ps[q * npix + hi * c.W + wi];
            } else {
                s += outside;
            }
        }
    rowsum[m] = s;
}

// Development hook: DGQ_GEMM_FORCE="BM,BN,S" overrides the plan (tile sweeps, tools/bench_gemm_sweep.py); read per call.
// "F<TM>,<NW>,1,<KW>" names a quantise-on-load configuration (gemm_panel.hip; honoured only for calls that carry extra.act).
static bool forced_plan(GemmPlan& pl) {
    const char* e = getenv("DGQ_GEMM_FORCE");
    if (!e || !*e) return false;
    int bm = 0, bn = 0, s = 0;
    if (*e == 'F') {
        int kw = 1;
        if (sscanf(e + 1, "%d,%d,%d,%d", &bm, &bn, &s, &kw) < 3) return false;
        bm += PANEL_BM0;
        pl.kw = kw < 1 ? 1 : kw;
        pl.fuse = true;
    } else if (sscanf(e, "%d,%d,%d", &bm, &bn, &s) != 3) {
        return false;
    } else {
        pl.fuse = false;
    }
    pl.bm = bm; pl.bn = bn; pl.splits = s < 1 ? 1 : s;
    return true;
}

// K splits the library would use for a shape (1: the whole epilogue runs in the GEMM kernel; callers ask before they request
// GroupNorm partials for a layer that is better off split — the request forces splits = 1)
extern "C" int dgq_gemm_plan_splits(int M, int N, int Kp, int w_bits, int per_m, size_t workspace_bytes) {
    GemmPlan pl = plan_gemm(M, N, Kp, w_bits, workspace_bytes, per_m != 0);
    forced_plan(pl);
    const int nk = Kp / BK;
    if (pl.splits < 1) pl.splits = 1;
    const int tps = (nk + pl.splits - 1) / pl.splits;
    return (nk + tps - 1) / tps;
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
    DGQ_CHECK_ARG(a.ldy >= (a.extra && a.extra->geglu ? a.N / 2 : a.N), "dgq_gemm_wxa8: ldy < output columns");
    // per-M keeps ONE running int32 total over the whole K range: |T| <= Kp·128·wmax must stay below 2^31
    // (per-K totals are cleared by the cflush == 2 marks of the planner and never pass 2^24)
    DGQ_CHECK_ARG(a.Kp / DGQ_KCHUNK <= 4096 && (!a.per_m || a.Kp <= (a.w_bits == 4 ? 1 << 20 : 1 << 17)),
                  "dgq_gemm_wxa8: Kp=%d too large for W%d", a.Kp, a.w_bits);
    DGQ_CHECK_ARG((reinterpret_cast<uintptr_t>(a.codes) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.wpacked) & 15) == 0,
                  "dgq_gemm_wxa8: codes/wpacked must be 16-byte aligned");
    if (a.per_m) {
        DGQ_CHECK_ARG(a.mdelta && a.mzp && a.vn && a.L >= 1, "dgq_gemm_wxa8: per_m needs mdelta/mzp/vn/L");
    } else {
        DGQ_CHECK_ARG(a.cdelta && a.cflush, "dgq_gemm_wxa8: per-K mode needs cdelta/cflush");
    }
    p.cv.codes_in = nullptr;
    p.ccoef = nullptr;
    p.wfrag = nullptr;
    p.act.x = nullptr;
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
        DGQ_CHECK_ARG(!p.ex.geglu || (a.N % 4 == 0 && !p.ex.residual && p.ex.fq_mode == 0),
                      "dgq_gemm_wxa8: the GEGLU epilogue needs N %% 4 == 0 and no other extra");
        if (p.ex.wfrag) {
            DGQ_CHECK_ARG(a.w_bits == 4 && (reinterpret_cast<uintptr_t>(p.ex.wfrag) & 15) == 0, "dgq_gemm_wxa8: wfrag is the W4 layout-2 image, 16-byte aligned");
            p.wfrag = reinterpret_cast<const uint8_t*>(p.ex.wfrag);
        }
        if (p.ex.act) {
            const dgq_gemm_act_t& q = *p.ex.act;
            DGQ_CHECK_ARG(q.x && p.ex.wfrag && q.K > 0 && q.K % 4 == 0 && q.K <= a.Kp && q.ldx >= q.K && q.bits >= 2 && q.bits <= 8 &&
                          q.x_dtype == a.y_dtype && (a.per_m || (q.czp && (q.kh > 1 || q.kdst))) && (q.pre_scale == nullptr) == (q.pre_shift == nullptr) &&
                          (q.kh <= 1 || (q.kpat && !q.ln_gamma && q.B >= 1 && q.H >= 1 && q.W >= 1 && q.kw >= 1 && q.stride >= 1 && q.pad >= 0)) &&
                          (!q.pre_scale || q.rows_per_image >= 1) && (q.pre_act == 0 || q.pre_act == 1) &&
                          (q.ln_gamma == nullptr) == (q.ln_beta == nullptr) && (!q.ln_gamma || (q.ln_eps > 0.0f && !q.pre_scale && q.pre_act == 0)) &&
                          (reinterpret_cast<uintptr_t>(q.x) & 15) == 0 && (q.ldx * (q.x_dtype == DGQ_F32 ? 4 : 2)) % 8 == 0,
                          "dgq_gemm_wxa8: bad quantise-on-load descriptor (needs wfrag, x of the output dtype, K %% 4 == 0, per-K: kdst + czp)");
            p.act = q;
        }
        if (p.ex.flush_coef && !a.per_m) {
            DGQ_CHECK_ARG((reinterpret_cast<uintptr_t>(p.ex.flush_coef) & 15) == 0, "dgq_gemm_wxa8: flush_coef must be 16-byte aligned");
            p.ccoef = p.ex.flush_coef;
        }
        if (p.ex.conv) {
            const dgq_gemm_conv_t& c = *p.ex.conv;
            DGQ_CHECK_ARG(c.codes_in && c.pixsum && c.pixsum_parts >= 1 && c.fill && a.per_m && a.L == 1 && a.w_bits == 4 && c.C > 0 && c.C % 16 == 0 && c.ldc >= c.C &&
                          c.ldc % 16 == 0 && c.kh >= 1 && c.kw >= 1 && c.stride >= 1 && c.Ho > 0 && c.Wo > 0 && c.B > 0 &&
                          a.M == c.B * c.Ho * c.Wo && a.Kp >= c.C * c.kh * c.kw && a.Kp - c.C * c.kh * c.kw < DGQ_KTILE + 16 &&
                          (reinterpret_cast<uintptr_t>(c.codes_in) & 15) == 0 && (reinterpret_cast<uintptr_t>(c.fill) & 15) == 0,
                          "dgq_gemm_wxa8: bad implicit-conv descriptor (needs per_m with L = 1, W4, C %% 16 == 0, M = B*Ho*Wo, 16-byte aligned codes_in / fill)");
            p.cv = c;
        }
        DGQ_CHECK_ARG(!p.ex.y2 || (p.ex.ldy2 >= a.N && !p.ex.geglu), "dgq_gemm_wxa8: y2 (second copy of the output rows) needs ldy2 >= N and no GEGLU epilogue");
        DGQ_CHECK_ARG(!p.ex.gn_partial || (a.M % 16 == 0 && a.N % 4 == 0 && !p.ex.geglu && p.ex.fq_mode == 0 &&
                                           (reinterpret_cast<uintptr_t>(p.ex.gn_partial) & 15) == 0),
                      "dgq_gemm_wxa8: GroupNorm partials need M %% 16 == 0, N %% 4 == 0, a 16-byte aligned buffer and no GEGLU / fused quantizer");
    } else {
        p.ex.residual = nullptr; p.ex.ldr = 0; p.ex.res_div = 1; p.ex.res_dtype = DGQ_F32; p.ex.fq_mode = 0; p.ex.fq_delta = nullptr; p.ex.fq_zp = nullptr;
        p.ex.fq_T = 1; p.ex.fq_D = 1; p.ex.fq_skip = 0; p.ex.fq_qmax = 255.0f; p.ex.geglu = 0; p.ex.gn_partial = nullptr; p.ex.conv = nullptr; p.ex.flush_coef = nullptr; p.ex.wfrag = nullptr; p.ex.act = nullptr;
        p.ex.y2 = nullptr; p.ex.ldy2 = 0;
    }
    p.splits = 1; p.slab = nullptr;
    p.tiles_per_split = a.Kp / BK;
    return DGQ_OK;
}

template <typename B>
static int dispatch_gemm(const B& bt, int w_bits, bool per_m, const GemmPlan& pl, int y_dtype, hipStream_t st) {
    if (w_bits == 4) return per_m ? launch_gemm<4, true>(bt, pl, y_dtype, st) : launch_gemm<4, false>(bt, pl, y_dtype, st);
    return per_m ? launch_gemm<8, true>(bt, pl, y_dtype, st) : launch_gemm<8, false>(bt, pl, y_dtype, st);
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
        DGQ_CHECK_ARG(!bt.p[i].cv.codes_in, "dgq_gemm_wxa8_batch: implicit-conv problems take dgq_gemm_wxa8");
        DGQ_CHECK_ARG(args[i].w_bits == args[0].w_bits && (args[i].per_m != 0) == (args[0].per_m != 0) && args[i].y_dtype == args[0].y_dtype,
                      "dgq_gemm_wxa8_batch: problem %d differs from problem 0 in weight bits / scale mode / output dtype", i);
    }
    const dgq_gemm_args_t& a0 = args[0];
    GemmPlan pl = plan_gemm(a0.M, a0.N, a0.Kp, a0.w_bits, 0, a0.per_m != 0, n == 1);
    const bool fuse = bt.p[0].act.x != nullptr;
    for (int i = 1; i < n; ++i) DGQ_CHECK_ARG((bt.p[i].act.x != nullptr) == fuse, "dgq_gemm_wxa8_batch: quantise-on-load for all problems of a launch or none");
    if (fuse) {
        int maxKp = 0;
        for (int i = 0; i < n; ++i) maxKp = args[i].Kp > maxKp ? args[i].Kp : maxKp;
        DGQ_CHECK_ARG(plan_panel_fuse(a0.M, a0.N, maxKp, a0.per_m != 0, pl), "dgq_gemm_wxa8_batch: this shape does not take quantise-on-load (dgq_gemm_act_fuses)");
        GemmPlan f = pl;
        if (forced_plan(f) && f.fuse) pl = f;
    } else {
        forced_plan(pl);
        DGQ_CHECK_ARG(!pl.fuse, "dgq_gemm_wxa8_batch: DGQ_GEMM_FORCE names quantise-on-load, the call carries codes");
    }
    pl.splits = 1;
    return dispatch_gemm(bt, a0.w_bits, a0.per_m != 0, pl, a0.y_dtype, (hipStream_t)stream);
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
    GemmPlan pl = plan_gemm(M, N, Kp, w_bits, workspace ? workspace_bytes : 0, per_m != 0, /*allow_big=*/!p.cv.codes_in);
    if (p.act.x && p.act.kh > 1) {                       // a k x k convolution with its quantiser inside the launch (gemm_convq.hip)
        const dgq_gemm_act_t& q = p.act;
        int lds = 0;
        const int slab = dgq_gemm_convq_plan(q.B, q.H, q.W, q.K, q.kh, q.kw, q.stride, q.pad, N, Kp, w_bits, per_m != 0, &lds);
        const int Ho = (q.H + 2 * q.pad - q.kh) / q.stride + 1, Wo = (q.W + 2 * q.pad - q.kw) / q.stride + 1;
        DGQ_CHECK_ARG(slab > 0 && !p.cv.codes_in && M == q.B * Ho * Wo && !p.ex.geglu && p.ex.fq_mode == 0 && q.rows_per_image == q.H * q.W,
                      "dgq_gemm_wxa8: this convolution does not take quantise-on-load (dgq_gemm_conv_act_fuses)");
        p.splits = 1; p.slab = nullptr; p.tiles_per_split = Kp / BK;
        const int rc2 = dgq_launch_gemm_convq(bt, per_m != 0, y_dtype, slab, lds, (hipStream_t)stream);
        if (rc2 != DGQ_OK) return rc2;
        return dgq_launch_status("dgq_gemm_wxa8");
    }
    if (p.act.x) {                                       // quantise-on-load: the panel kernel, whole K in one workgroup
        DGQ_CHECK_ARG(!p.cv.codes_in && plan_panel_fuse(M, N, Kp, per_m != 0, pl), "dgq_gemm_wxa8: this shape does not take quantise-on-load (dgq_gemm_act_fuses)");
        GemmPlan f = pl;
        if (forced_plan(f) && f.fuse) pl = f;
        p.splits = 1; p.slab = nullptr; p.tiles_per_split = Kp / BK;
        return dispatch_gemm(bt, w_bits, per_m != 0, pl, y_dtype, (hipStream_t)stream);
    }
    if (p.cv.codes_in) {
        // implicit im2col: the row sums of the unfolded operand from the per-pixel sums first (a tiny launch), then one of the three
        // tile shapes that carry the CONV addressing (the plan's nearest)
        conv_tile(pl, M);
        const int thr = 256;
        hipLaunchKernelGGL(conv_rowsum_kernel, dim3((M + thr - 1) / thr), dim3(thr), 0, (hipStream_t)stream, p.cv, M, const_cast<float*>(rowsum));
        p.rowsum_parts = 1;
    }
    if (forced_plan(pl)) {
        DGQ_CHECK_ARG(!pl.fuse, "dgq_gemm_wxa8: DGQ_GEMM_FORCE names quantise-on-load, the call carries codes");
        if (p.cv.codes_in) { const int fs = pl.splits; conv_tile(pl, 0); pl.splits = fs; }
        DGQ_CHECK_ARG(pl.splits == 1 || (workspace && (size_t)pl.splits * M * N * 4 <= workspace_bytes),
                      "dgq_gemm_wxa8: DGQ_GEMM_FORCE split does not fit the workspace");
    }
    if (p.ex.geglu) pl.splits = 1;                       // this epilogue lives in the GEMM kernel, not in the combine
    p.splits = pl.splits;
    p.slab = p.splits > 1 ? reinterpret_cast<float*>(workspace) : nullptr;
    const int nk = Kp / BK;
    p.tiles_per_split = (nk + p.splits - 1) / p.splits;
    p.splits = (nk + p.tiles_per_split - 1) / p.tiles_per_split;      // no empty split
    if (p.splits == 1) p.slab = nullptr;
    return dispatch_gemm(bt, w_bits, per_m != 0, pl, y_dtype, (hipStream_t)stream);
}
