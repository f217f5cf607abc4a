// W4A8 GEMM for the compute-bound shapes: 256-row / 256-column workgroup tiles, eight waves in two groups that run ONE
// BARRIER APART ("ping-pong"), a three-tile LDS-DMA ring with counted waits.  Same operands, same arithmetic and the same
// epilogue (gemm_tile.h) as gemm_wxa8.hip's tile family — results are bit-identical — but a different loop structure:
//
//   * what bounds the 128x128 kernel at 8192^3 is the SIMD's issue port, not the matrix pipe: per K tile a wave issues 6
//     LDS-DMA pieces (~100 issue cycles each) + 12 VALU per chunk of int4 widening around 16 MFMAs (512 pipe cycles), and the
//     waves of a workgroup do so in lockstep — all of them stage, then all of them multiply (profiles/r02_gemm_pmc_*:
//     MFMA busy 31-46 %).  Here a wave owns a 128x64 (per-M) output tile: half the DMA pieces and half the widening per MFMA;
//   * the two waves of a SIMD belong to different GROUPS (waves 0-3 / 4-7).  Every phase is
//         LOAD    ds_read the fragments of two chunks, issue this phase's share of tile t+2's DMA, [counted vmcnt], lgkmcnt(0)
//         s_barrier
//         COMPUTE the MFMAs of those two chunks (+ int4 widening, + per-K group flushes)
//         s_barrier
//     and group 1 runs one barrier behind group 0, so on every SIMD one wave's COMPUTE overlaps its partner's LOAD by
//     construction instead of by the luck of two unsynchronised workgroups;
//   * hazards are closed by construction, not by timing: a wave's ds_reads of a tile have RETURNED (lgkmcnt(0)) before the
//     barrier it signals next, the DMA that overwrites that buffer is issued at least one barrier later by anyone; a tile's
//     DMA is retired by every wave's counted vmcnt BEFORE a barrier that precedes its first read (cdna guide §5, "Read a
//     staged buffer one phase AFTER the wait that retires it").
//
// Per-M (scalar / per-token activation scales): 256x256 tile, wave tile 128x64, int32 accumulators only (128 registers).
// Per-K (DGQ channel groups): 256x128 tile, wave tile 128x32 (BIG_PERK_TILE = 0: 128x256, 64x64) — the fp32 group accumulators
// double the accumulator registers, so the wave tile is half the per-M one.
#include "gemm_tile.h"

// Build-time experiment switches (tools/build_variants.sh builds one library per setting for A/B runs on one box)
#ifndef BIG_GROUP_M
#define BIG_GROUP_M 4          // > 0: inside an XCD's tile range, walk GROUP_M row tiles per column tile (L2 reuse of both operands)
#endif
#ifndef BIG_NPH_M
#define BIG_NPH_M 2            // phases per K tile, per-M (2: two chunks = 16 MFMAs per phase and wave)
#endif
#ifndef BIG_NPH_K
#define BIG_NPH_K 1            // phases per K tile, per-K (1: four chunks = 16 MFMAs per phase on the 64x64 wave tile)
#endif
#ifndef BIG_DMA_LOAD_M
#define BIG_DMA_LOAD_M 0       // DMA pieces of a phase issued in its LOAD segment (the rest: between the MFMAs), per-M
#endif
#ifndef BIG_DMA_LOAD_K
#define BIG_DMA_LOAD_K 0       // ... per-K
#endif
#ifndef BIG_PRIO
#define BIG_PRIO 1             // s_setprio 1 around the COMPUTE segment (0: not; measured neutral)
#endif

#ifndef BIG_PERK_TILE
#define BIG_PERK_TILE 1        // per-K tile: 0 = 128x256 (wave tile 64x64), 1 = 256x128 (wave tile 128x32; measured 38.2 vs 37.3 % at 8192^3)
#endif
// ablations (WRONG results, timing only): what the loop costs without its DMA / fragment reads / int4 widening / flushes
#ifndef BIG_ABL
#define BIG_ABL 0              // bit 0: no DMA in the loop, bit 1: no fragment reads, bit 2: no widening, bit 3: no flush
#endif

#ifndef BIG_STAMP
#define BIG_STAMP 0            // 1 (diagnostic build, WRONG output): s_memtime stamps of K tile 8 of workgroup 0, written over y row 0
#endif
// the two switches above produce WRONG output: they exist in diagnostic builds (-DDGQ_DIAG, `make diag` / tools/build_variants.sh)
// only — the shipped library cannot be built with them
#if !defined(DGQ_DIAG) && (BIG_ABL != 0 || BIG_STAMP != 0)
#error "BIG_ABL / BIG_STAMP change the kernel's results: diagnostic builds only (add -DDGQ_DIAG)"
#endif

namespace {

__device__ __forceinline__ uint64_t big_stamp() {
    uint64_t v;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v)::"memory");
    return v;
}

constexpr int BIG_WVM = 2, BIG_WVN = 4, BIG_NW = 8, BIG_NT = 512;

template <bool PER_M, typename TOut, int BM, int BN, int NBUF>
__global__ __launch_bounds__(BIG_NT, 2) void gemm_big_kernel(GemmBatch bt) {
    const GemmParams& p = bt.p[0];
    const uint64_t st_entry = BIG_STAMP == 2 ? big_stamp() : 0;      // BIG_STAMP = 2: block timeline (entry / loop start / loop end / stores issued / done)
    int tile_n, tile_m;
    {   // XCD-aware tile order (as gemm_wxa8_kernel): XCD k owns a contiguous m-major tile range
        const int gx = gridDim.x, T = gridDim.x * gridDim.y;
        const int bid = blockIdx.x + gx * blockIdx.y;
        const int q = T >> 3, r = T & 7, xcd = bid & 7;
        const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
        if (BIG_GROUP_M > 0) {
            constexpr int GM = BIG_GROUP_M > 0 ? BIG_GROUP_M : 1;
            const int gy = gridDim.y, band = logical / (GM * gx), first_m = band * GM;
            const int gsz = min(GM, gy - first_m), in = logical - band * GM * gx;
            tile_n = in / gsz;
            tile_m = first_m + (in - tile_n * gsz);
        } else {
            tile_m = logical / gx;
            tile_n = logical - tile_m * gx;
        }
    }
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int WVM = BIG_WVM, WVN = BIG_WVN, NW = BIG_NW, NT = BIG_NT;
    constexpr int WM = BM / WVM, WN = BN / WVN;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int A_BYTES = BM * BK, W_ROW = BK / 2, W_BYTES = BN * W_ROW, STAGE_BYTES = A_BYTES + W_BYTES;
    constexpr int A_DMA = A_BYTES / 1024 / NW, W_DMA = W_BYTES / 1024 / NW, PER_TILE = A_DMA + W_DMA;
    static_assert(A_BYTES % (1024 * NW) == 0 && W_BYTES % (1024 * NW) == 0, "whole 1-KiB pieces per wave");
    static_assert(PER_TILE * 2 <= 63, "vmcnt is 6 bits");
    constexpr int NPH = PER_M ? BIG_NPH_M : BIG_NPH_K, CPP = NCH / NPH;   // phases per K tile, chunks per phase

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int group = wid >> 2;                              // waves 0-3 / 4-7: the two waves of a SIMD are in different groups
    const int wave_m = wid >> 2, wave_n = wid & 3;
    const int n0 = tile_n * BN, m0 = tile_m * BM;
    const int nk = p.Kp / BK;

    // per-lane global source pointers of this wave's DMA pieces (k offset added per tile); swizzles as gemm_wxa8_kernel
    const int8_t* a_src[A_DMA];
#pragma unroll
    for (int i = 0; i < A_DMA; ++i) {
        const int row = (wid * A_DMA + i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        a_src[i] = p.codes + (int64_t)min(m0 + row, p.M - 1) * p.Kp + 16 * c;
    }
    const uint8_t* w_src[W_DMA];
#pragma unroll
    for (int i = 0; i < W_DMA; ++i) {
        const int row = (wid * W_DMA + i) * 16 + (lane >> 2);
        const int c = (lane & 3) ^ ((row >> 2) & 3);
        w_src[i] = p.wpacked + (int64_t)min(n0 + row, p.N - 1) * (p.Kp / 2) + 16 * c;
    }
    const uint32_t lds_base = (uint32_t)(uintptr_t)(lptr_t)smem;
    // piece q (0 .. PER_TILE-1) of K tile kt into ring stage `stage`: the A pieces first, then the W pieces
    auto issue_piece = [&](int q, int kt, int stage) {
        const uint32_t sa = lds_base + stage * STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < A_DMA; ++i)
            if (q == i) glds16(a_src[i] + (int64_t)kt * BK, __builtin_amdgcn_readfirstlane(sa + (wid * A_DMA + i) * 1024));
#pragma unroll
        for (int i = 0; i < W_DMA; ++i)
            if (q == A_DMA + i) glds16(w_src[i] + (int64_t)kt * (BK / 2), __builtin_amdgcn_readfirstlane(sa + A_BYTES + (wid * W_DMA + i) * 1024));
    };

    v16i acc[TM][TN];
    v16f accf[TM][TN];
    constexpr bool BIASED = !PER_M;                         // per-K (W4): totals carry DGQ_ACC_BIAS_I (gemm_device.h)
    constexpr int ACC0 = BIASED ? DGQ_ACC_BIAS_I : 0;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][j][r] = ACC0; accf[i][j][r] = 0.0f; }

    // fragment addresses inside a stage: lane l holds row / column l & 31 and the 16 k of half l >> 5 of a chunk.  Row tiles are
    // 32 rows apart — 4096 (A) / 2048 (W) bytes — and the swizzle terms do not change with them, so one address per chunk serves
    // every row / column tile through a constant offset.
    const int lr = lane & 31, hh = lane >> 5;
    int a_base[NCH], w_base[NCH];
    {
        const int ra = wave_m * WM + lr, rw = wave_n * WN + lr;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            a_base[c] = ra * BK + (((2 * c + hh) ^ ((ra >> 1) & 7)) << 4);
            w_base[c] = rw * W_ROW + ((c ^ ((rw >> 2) & 3)) << 4) + ((hh ^ ((rw >> 4) & 1)) << 3);
        }
    }

    // ---- prologue: tiles 0 and 1 go out first; the epilogue vectors and the per-chunk flush coefficients are staged with ordinary
    // loads (hipcc waits vmcnt(0) for them, which also lands tile 0 and 1: nothing else is in flight yet).
    float* vtab = reinterpret_cast<float*>(smem + NBUF * STAGE_BYTES);     // [3][BM]: R0 R1 R2 | [4][BN]: alpha zw gamma vn
    float* vcol = vtab + 3 * BM;
    float* ctab = vcol + 4 * BN;                                           // per-K: [nk][NCH] flush coefficients | [nk] clear flags
    static_assert(BM <= NT && BN <= NT, "one row / column of the epilogue vectors per thread");
#pragma unroll
    for (int q = 0; q < PER_TILE; ++q) issue_piece(q, 0, 0);
    if (nk > 1) {
#pragma unroll
        for (int q = 0; q < PER_TILE; ++q) issue_piece(q, 1, 1);
    }
    {
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
        }
        if (tid < BN) {
            const int n = min(n0 + tid, p.N - 1);
            vcol[tid] = p.alpha[n]; vcol[BN + tid] = p.zw[n]; vcol[2 * BN + tid] = p.gamma[n];
            vcol[3 * BN + tid] = PER_M ? p.vn[n] : 0.0f;
        }
        if constexpr (!PER_M) {
            // Summation by parts (gemm_wxa8.hip): accf += (δ_c − δ_next)·float(running total) at every chunk; the coefficient is
            // non-zero at group ends only.  A clear mark on a K tile's last chunk makes that chunk's coefficient the full δ_c and
            // asks for the totals to be cleared behind the tile.
            const int nchunk = nk * NCH;
            for (int e = tid; e < nchunk + nk; e += NT) {
                if (e < nchunk) {
                    const int tcl = (e / NCH) * NCH + NCH - 1;                  // last chunk of this chunk's K tile
                    const bool clr = (p.cflush[tcl] & 0xFF) == 2;
                    const float d = p.cdelta[e], dn = p.cdelta[min(e + 1, nchunk - 1)];
                    const bool full = (e == nchunk - 1) || (e == tcl && clr);
                    ctab[e] = full ? d : d - dn;
                } else {
                    const int t = e - nchunk;
                    ctab[e] = (t + 1 < nk && (p.cflush[t * NCH + NCH - 1] & 0xFF) == 2) ? 1.0f : 0.0f;
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (group == 1) __builtin_amdgcn_s_barrier();            // group 1 runs one barrier behind group 0 from here on

    const bool use_ccoef = !PER_M && p.ccoef != nullptr;      // wave-uniform (a kernel argument)
    const uint64_t st_loop0 = BIG_STAMP == 2 ? big_stamp() : 0;
    uint64_t stamps[6 * NPH];
    auto stamp = [&](int k, int t) {
        if (BIG_STAMP && t == 8) stamps[k] = big_stamp();
    };
#pragma unroll
    for (int k = 0; k < 6 * NPH; ++k) stamps[k] = 0;
    int stage = 0, istage = 2 % NBUF;
    // One K tile.  DMA = true: the steady state — tile t+2 goes out, every wait is the counted one, and the instruction stream has
    // NO taken branch besides the loop's back edge (a taken branch costs a wave ~100 cycles of refetch: the in-line flush block of
    // the first version made 8192^3 g16 take 646 us against 387 us without any flush code; conditionals of the tail — "is there
    // a tile t+2?" — inside one loop body cost the LOAD segment two of them per tile).  DMA = false: the last two tiles.
    // pieces of a K tile are dealt to the phases in order: piece q belongs to phase (q·NPH)/PER_TILE
    auto pieces_in_phase = [](int ph) constexpr { int n = 0; for (int q = 0; q < PER_TILE; ++q) n += ((q * NPH) / PER_TILE == ph); return n; };
    auto pieces_before_phase = [](int ph) constexpr { int n = 0; for (int q = 0; q < PER_TILE; ++q) n += ((q * NPH) / PER_TILE < ph); return n; };
    auto piece_rank = [](int q, int ph) constexpr { int n = 0; for (int r = 0; r < q; ++r) n += ((r * NPH) / PER_TILE == ph); return n; };
    constexpr int DMA_LOAD = PER_M ? BIG_DMA_LOAD_M : BIG_DMA_LOAD_K;    // pieces of a phase issued in its LOAD segment
    auto tile = [&](int t, auto dma_c) {
        constexpr bool DMA = decltype(dma_c)::value && !(BIG_ABL & 1);
        const uint8_t* sa = smem + stage * STAGE_BYTES;
        const uint8_t* sw = sa + A_BYTES;
#pragma unroll
        for (int ph = 0; ph < NPH; ++ph) {
            // ---------------- LOAD
            stamp(6 * ph + 0, t);
            v4i af[CPP][TM];
            uint2 wf[CPP][TN];
#pragma unroll
            for (int cc = 0; cc < CPP; ++cc) {
                const int c = ph * CPP + cc;
                if (BIG_ABL & 2) {
#pragma unroll
                    for (int j = 0; j < TN; ++j) wf[cc][j] = make_uint2(lane + c, lane + j);
#pragma unroll
                    for (int i = 0; i < TM; ++i) af[cc][i] = (v4i){lane, c, i, t};
                } else {
#pragma unroll
                    for (int j = 0; j < TN; ++j) wf[cc][j] = *reinterpret_cast<const uint2*>(sw + w_base[c] + j * (32 * W_ROW));
#pragma unroll
                    for (int i = 0; i < TM; ++i) af[cc][i] = *reinterpret_cast<const v4i*>(sa + a_base[c] + i * (32 * BK));
                }
            }
            // this phase's flush coefficients and the tile's clear flag, as wave-uniform bit patterns.  From the caller's table
            // (p.ccoef) they arrive by SCALAR loads issued here and retired by the lgkmcnt(0) below — the tests between the MFMAs
            // are then s_cmp + s_cbranch (not taken); from the LDS table they take a v_readfirstlane each.
            int cqb[CPP];
            int tclr_b = 0;
            v4i c4 = {0, 0, 0, 0};
            float cq[CPP];
            float tclr = 0.0f;
            if constexpr (!PER_M) {
                if (__builtin_expect(use_ccoef, 1)) {
                    const float* cp = p.ccoef + (t * NCH + ph * CPP);
                    if constexpr (CPP == 4) {
                        asm volatile("s_load_dwordx4 %0, %1, 0x0" : "=s"(c4) : "s"(cp) : "memory");      // (read only behind the wait: see below)
                    } else {
#pragma unroll
                        for (int cc = 0; cc < CPP; ++cc) asm volatile("s_load_dword %0, %1, 0x0" : "=s"(cqb[cc]) : "s"(cp + cc) : "memory");
                    }
                    if (ph == NPH - 1) asm volatile("s_load_dword %0, %1, 0x0" : "=s"(tclr_b) : "s"(p.ccoef + nk * NCH + t) : "memory");
                } else {
#pragma unroll
                    for (int cc = 0; cc < CPP; ++cc) cq[cc] = ctab[t * NCH + ph * CPP + cc];
                    if (ph == NPH - 1) tclr = ctab[nk * NCH + t];
                }
            }
            // this phase's share of tile t+2: the first DMA_LOAD pieces here, behind the fragment reads (where a piece costs its
            // issuing wave 100+ cycles: the LDS queue is full of reads), the rest between the MFMAs of the COMPUTE segment
            if (DMA) {
#pragma unroll
                for (int q = 0; q < PER_TILE; ++q)
                    if ((q * NPH) / PER_TILE == ph && piece_rank(q, ph) < DMA_LOAD) issue_piece(q, t + 2, istage);
            }
            stamp(6 * ph + 1, t);
            if (ph == NPH - 1) {
                // tile t+1 (this wave's pieces) has landed; what has been issued of tile t+2 so far stays in flight
                constexpr int YOUNGER = pieces_before_phase(NPH - 1) + (pieces_in_phase(NPH - 1) < DMA_LOAD ? pieces_in_phase(NPH - 1) : DMA_LOAD);
                if (DMA) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(YOUNGER) : "memory");
                else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            if constexpr (!PER_M) {
                if (__builtin_expect(!use_ccoef, 0)) {
#pragma unroll
                    for (int cc = 0; cc < CPP; ++cc) cqb[cc] = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, cq[cc]));
                    tclr_b = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, tclr));
                } else {
                    // the scalar loads above were issued from inline asm, which the compiler's waitcnt insertion does not track: their
                    // outputs pass through this (empty, volatile: ordered behind the s_waitcnt lgkmcnt(0) asm above) statement as
                    // read-write operands, so that every use of them is scheduled behind the wait
                    if constexpr (CPP == 4) {
                        asm volatile("" : "+s"(c4), "+s"(tclr_b));
                        cqb[0] = c4.x; cqb[1] = c4.y; cqb[2] = c4.z; cqb[3] = c4.w;
                    } else {
#pragma unroll
                        for (int cc = 0; cc < CPP; ++cc) asm volatile("" : "+s"(cqb[cc]));
                        asm volatile("" : "+s"(tclr_b));
                    }
                }
            }
            // int4 -> int8 here, not in COMPUTE: the fragments have just arrived, and the MFMA segment then opens with an MFMA
            v4i bf[CPP][TN];
#pragma unroll
            for (int cc = 0; cc < CPP; ++cc)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const uint2 v = wf[cc][j];
                    if (BIG_ABL & 4) bf[cc][j] = (v4i){(int)v.x, (int)v.y, (int)v.x, (int)v.y};
                    else bf[cc][j] = (v4i){(int)(v.x & 0x0F0F0F0Fu), (int)((v.x >> 4) & 0x0F0F0F0Fu),
                                           (int)(v.y & 0x0F0F0F0Fu), (int)((v.y >> 4) & 0x0F0F0F0Fu)};
                }
            stamp(6 * ph + 2, t);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            stamp(6 * ph + 3, t);
            __builtin_amdgcn_sched_barrier(0);
            // ---------------- COMPUTE
            if (BIG_PRIO == 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int cc = 0; cc < CPP; ++cc) {
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
                        acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[cc][i], bf[cc][j], acc[i][j], 0, 0, 0);
                if (DMA) {                                           // the rest of this phase's DMA share, one piece behind each chunk
#pragma unroll
                    for (int q = 0; q < PER_TILE; ++q)
                        if ((q * NPH) / PER_TILE == ph && piece_rank(q, ph) >= DMA_LOAD && (piece_rank(q, ph) - DMA_LOAD) % CPP == cc)
                            issue_piece(q, t + 2, istage);
                }
                if constexpr (!PER_M && !(BIG_ABL & 8)) {
                    // a group ends behind this chunk (coefficient != 0): out of line, the common case falls through
                    if (__builtin_expect((cqb[cc] & 0x7FFFFFFF) != 0, 0)) {
                        const float sc = __builtin_bit_cast(float, cqb[cc]);
#pragma unroll
                        for (int j = 0; j < TN; ++j)
#pragma unroll
                            for (int i = 0; i < TM; ++i)
#pragma unroll
                                for (int r = 0; r < 16; ++r) accf[i][j][r] = __builtin_fmaf(sc, dgq_total_to_float<BIASED>(acc[i][j][r]), accf[i][j][r]);
                    }
                }
            }
            if constexpr (!PER_M) {
                if (ph == NPH - 1 && __builtin_expect(tclr_b != 0, 0)) {   // rare: a segment of totals ends
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int r = 0; r < 16; ++r) acc[i][j][r] = ACC0;
                }
            }
            if (BIG_PRIO == 1) __builtin_amdgcn_s_setprio(0);
            stamp(6 * ph + 4, t);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            stamp(6 * ph + 5, t);
            __builtin_amdgcn_sched_barrier(0);
        }
        stage = (stage + 1 == NBUF) ? 0 : stage + 1;
        istage = (istage + 1 == NBUF) ? 0 : istage + 1;
    };
    int t = 0;
    for (; t + 2 < nk; ++t) tile(t, std::true_type());
    for (; t < nk; ++t) tile(t, std::false_type());
    if (group == 0) __builtin_amdgcn_s_barrier();            // both groups have now passed the same number of barriers
    const uint64_t st_loop1 = BIG_STAMP == 2 ? big_stamp() : 0;

    DGQ_DIAG_DECL                                            // (this kernel has its own BIG_STAMP timeline; the shared epilogue's stamp goes nowhere)
    gemm_store_tile<PER_M, TOut, BM, BN, WVM, WVN, 1, NBUF * STAGE_BYTES, TM, TN>(p, 0, smem, vtab, vcol, wid, lane, wave_m, wave_n, 0,
                                                                                  m0, n0, acc, accf DGQ_DIAG_ARG);
    const uint64_t st_issued = BIG_STAMP == 2 ? big_stamp() : 0;
    if (BIG_STAMP) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    if (BIG_STAMP == 2 && lane == 0 && blockIdx.y == 0 && (blockIdx.x == 0 || blockIdx.x == 8)) {
        uint64_t* dbg = reinterpret_cast<uint64_t*>(p.y) + (blockIdx.x ? 128 : 0) + wid * 16;
        dbg[0] = st_entry; dbg[1] = st_loop0; dbg[2] = st_loop1; dbg[3] = st_issued; dbg[4] = big_stamp();
        return;
    }
    if (BIG_STAMP && blockIdx.x == 0 && blockIdx.y == 0 && lane == 0) {
        uint64_t* dbg = reinterpret_cast<uint64_t*>(p.y) + wid * 16;
#pragma unroll
        for (int k = 0; k < 6 * NPH; ++k) dbg[k] = stamps[k];
    }
}

template <bool PER_M, typename TOut, int BM, int BN, int NBUF>
void launch_big(const GemmBatch& bt, hipStream_t st) {
    const GemmParams& p = bt.p[0];
    constexpr int stage = BM * BK + BN * (BK / 2);
    const int nk = p.Kp / BK;
    const int lds = NBUF * stage + (3 * BM + 4 * BN) * 4 + (PER_M ? 0 : (((NCH + 1) * nk * 4 + 15) & ~15));
    static std::atomic<bool> attr_set[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !attr_set[dev].load(std::memory_order_acquire)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_big_kernel<PER_M, TOut, BM, BN, NBUF>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (dev >= 0 && dev < 64) attr_set[dev].store(true, std::memory_order_release);
    }
    dim3 grid((p.N + BN - 1) / BN, (p.M + BM - 1) / BM, 1), block(BIG_NT);
    hipLaunchKernelGGL((gemm_big_kernel<PER_M, TOut, BM, BN, NBUF>), grid, block, lds, st, bt);
}

template <typename TOut>
int launch_big_dtype(const GemmBatch& bt, bool per_m, hipStream_t st) {
    if (per_m) launch_big<true, TOut, 256, 256, 3>(bt, st);
    else if (BIG_PERK_TILE == 0) launch_big<false, TOut, 128, 256, 3>(bt, st);
    else launch_big<false, TOut, 256, 128, 3>(bt, st);
    return DGQ_OK;
}

}  // namespace

// LDS the big kernel needs for a problem (the host checks it against the 160 KiB of a CU before it plans this kernel)
size_t dgq_gemm_big_lds_bytes(bool per_m, int Kp) {
    const int nk = Kp / BK;
    const int bm = (per_m || BIG_PERK_TILE) ? 256 : 128, bn = (per_m || !BIG_PERK_TILE) ? 256 : 128;
    return (size_t)3 * (bm * BK + bn * (BK / 2)) + (3 * bm + 4 * bn) * 4 + (per_m ? 0 : (((NCH + 1) * nk * 4 + 15) & ~15));
}

// W4 only, one problem, no K split (bt.n == 1, p.splits == 1): the caller (gemm_wxa8.hip: dispatch_gemm) has checked
int dgq_launch_gemm_big(const GemmBatch& bt, bool per_m, int y_dtype, hipStream_t st) {
    switch (y_dtype) {
        case DGQ_F32: return launch_big_dtype<float>(bt, per_m, st);
        case DGQ_F16: return launch_big_dtype<__half>(bt, per_m, st);
        case DGQ_BF16: return launch_big_dtype<__hip_bfloat16>(bt, per_m, st);
        default: dgq_set_error("dgq_gemm_wxa8: unknown y dtype %d", y_dtype); return DGQ_EINVAL;
    }
}
