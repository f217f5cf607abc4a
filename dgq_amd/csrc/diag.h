// In-kernel clock stamps — DIAGNOSTIC BUILDS ONLY (-DDGQ_DIAG; `make diag` -> libdgq_hip_diag.so).  __graft_entry__.build() and the
// plain `make` never define DGQ_DIAG: in the shipped library every macro below expands to nothing, no stamp executes and no
// dgq_diag_* symbol exists (tests/test_host_cpu.py checks the shipped .so for that).  A diagnostic build's RESULTS are still the
// product's (stamps leave the kernel only through the buffer below, which no other code reads), its TIMES are not: the fences
// around a stamp forbid overlaps the real kernel has, so read the SHARES of a stamped run, never its length
// (cdna guide §7 "In-kernel stamps").
//
// Per wave DGQ_DIAG_SLOTS 64-bit words: the kernel keeps them in SGPRs (s_memtime / s_memrealtime write scalar pairs) and lane 0
// stores them behind the kernel's own last memory operation — never in between, where a store would shift the hand-counted
// vmcnt waits of the LDS-DMA rings.
#pragma once
#ifdef DGQ_DIAG
#define DGQ_DIAG_SLOTS 16
#define DGQ_DIAG_WAVES (1 << 16)
struct DiagStamps { unsigned long long t[DGQ_DIAG_SLOTS]; };
// shader clock (one tick = one shader cycle; per XCD — compare inside a workgroup only)
__device__ __forceinline__ unsigned long long dgq_diag_now() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}
// constant 100 MHz clock shared by the whole chip: workgroup entry / exit on ONE time axis
__device__ __forceinline__ unsigned long long dgq_diag_real() {
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}
// HW_ID (wave / SIMD / CU / SE) in the low word, XCC_ID in the high one
__device__ __forceinline__ unsigned long long dgq_diag_where() {
    const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));
    const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));
    return (unsigned long long)hw | ((unsigned long long)xcc << 32);
}
#define DGQ_DIAG_BUFFER(name)                                                                                           \
    __device__ unsigned long long dgq_diag_buf_##name[(size_t)DGQ_DIAG_WAVES * DGQ_DIAG_SLOTS];                          \
    extern "C" int dgq_diag_fetch_##name(void* dst, size_t bytes) {                                                      \
        const size_t cap = sizeof(unsigned long long) * (size_t)DGQ_DIAG_WAVES * DGQ_DIAG_SLOTS;                          \
        return hipMemcpyFromSymbol(dst, HIP_SYMBOL(dgq_diag_buf_##name), bytes < cap ? bytes : cap) == hipSuccess ? 0 : -1; \
    }                                                                                                                    \
    extern "C" int dgq_diag_clear_##name(void) {                                                                         \
        void* q = nullptr;                                                                                               \
        if (hipGetSymbolAddress(&q, HIP_SYMBOL(dgq_diag_buf_##name)) != hipSuccess) return -1;                            \
        return hipMemset(q, 0, sizeof(unsigned long long) * (size_t)DGQ_DIAG_WAVES * DGQ_DIAG_SLOTS) == hipSuccess ? 0 : -1; \
    }
#define DGQ_DIAG_DECL DiagStamps dg = {};
#define DGQ_DIAG_PARAM , DiagStamps& dg
#define DGQ_DIAG_ARG , dg
#define DGQ_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); dg.t[i] = dgq_diag_now(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define DGQ_STAMP_REAL(i) do { __builtin_amdgcn_sched_barrier(0); dg.t[i] = dgq_diag_real(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define DGQ_STAMP_WHERE(i) do { dg.t[i] = dgq_diag_where(); } while (0)
// dg.t[i] += now − since   (time spent in a repeated segment)
#define DGQ_STAMP_ACC(i, since) do { __builtin_amdgcn_sched_barrier(0); dg.t[i] += dgq_diag_now() - (since); __builtin_amdgcn_sched_barrier(0); } while (0)
#define DGQ_STAMP_NOW(var) unsigned long long var; do { __builtin_amdgcn_sched_barrier(0); var = dgq_diag_now(); __builtin_amdgcn_sched_barrier(0); } while (0)
// every memory operation of the wave has completed, then lane 0 writes the wave's record
#define DGQ_DIAG_FLUSH(name, nwaves, wave, lane)                                                                         \
    do {                                                                                                                 \
        const size_t wg_ = blockIdx.x + (size_t)gridDim.x * (blockIdx.y + (size_t)gridDim.y * blockIdx.z);                \
        const size_t rec_ = wg_ * (nwaves) + (wave);                                                                     \
        if ((lane) == 0 && rec_ < DGQ_DIAG_WAVES) {                                                                      \
            _Pragma("unroll") for (int i_ = 0; i_ < DGQ_DIAG_SLOTS; ++i_) dgq_diag_buf_##name[rec_ * DGQ_DIAG_SLOTS + i_] = dg.t[i_]; \
        }                                                                                                                \
    } while (0)
#define DGQ_DIAG_DRAIN() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")
#else
#define DGQ_DIAG_BUFFER(name)
#define DGQ_DIAG_DECL
#define DGQ_DIAG_PARAM
#define DGQ_DIAG_ARG
#define DGQ_STAMP(i) do {} while (0)
#define DGQ_STAMP_REAL(i) do {} while (0)
#define DGQ_STAMP_WHERE(i) do {} while (0)
#define DGQ_STAMP_ACC(i, since) do {} while (0)
#define DGQ_STAMP_NOW(var) do {} while (0)
#define DGQ_DIAG_FLUSH(name, nwaves, wave, lane) do {} while (0)
#define DGQ_DIAG_DRAIN() do {} while (0)
#endif
