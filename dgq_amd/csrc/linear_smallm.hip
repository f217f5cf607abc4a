// Batched small-M quantized Linear: y_l = W_l · aqtizer_l(act(x)) + b_l for up to 24 layers that share ONE input of a
// few rows — the 23 `time_emb_proj(SiLU(temb))` projections of a UNet forward (QuantResnetBlock2D.forward,
// quant_block.py:98-119; x = [2·prompts, 1280]) are independent of the latents, so they run as ONE launch instead of 23
// quantise-on-load + 23 GEMM launches whose 128-row MFMA tiles would hold two live rows.
// Same integer identity as dgq_gemm_wxa8 (per_m = 1, L = 1: scalar activation quantizer), evaluated with V_DOT4_I32_I8:
// block = 256 threads = 64 output columns; the block first quantises the M x K input into int8 codes in LDS (the SiLU
// prologue and the exact-division rounding of dgq_quant_act), then each wave walks 16 weight rows with its lanes spread
// over K (coalesced 16-byte loads of packed int4) and reduces across the wave.
#include <atomic>
#include "dgq_common.h"
#include "gemm_device.h"

#define SMALLM_MAX_PROBLEMS 24
#define SMALLM_MAX_M 16
#define SMALLM_MAX_K 2048

struct SmallMProblem {
    const uint8_t* wpacked;    // W4: [N][Kp/2] (dgq_pack_w4 layout 1, natural K order); W8: [N][Kp] int8
    const float* alpha;        // δw [N]
    const float* zw;           // zero point in the stored code domain [N]
    const float* gamma;        // bias [N]
    const float* vn;           // Σ_k qw' − K·zw [N]
    const float* mdelta;       // scalar activation δ (device, 1 element)
    const float* mzp;
    void* y;                   // [M][ldy]
    int ldy, N, Kp, w_bits, a_bits, block0;
};

struct SmallMBatch {
    SmallMProblem p[SMALLM_MAX_PROBLEMS];
    int n;
    const void* x;             // [M][ldx] shared input
    int x_dtype, y_dtype, M, K, ldx, pre_act;
};

template <typename TIn, typename TOut>
__global__ __launch_bounds__(256) void linear_smallm_kernel(SmallMBatch b) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    int pi = 0;
    while (pi + 1 < b.n && (int)blockIdx.x >= b.p[pi + 1].block0) ++pi;
    const SmallMProblem& P = b.p[pi];
    const int n0 = ((int)blockIdx.x - P.block0) * 64;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int Kq = (b.K + 15) & ~15;                     // code row stride in LDS (16-byte aligned, zero padded)
    int8_t* codes = reinterpret_cast<int8_t*>(smem);     // [M][Kq]
    const size_t codes_bytes = ((size_t)b.M * Kq + 15) & ~(size_t)15;
    float* rsum = reinterpret_cast<float*>(smem + codes_bytes);                          // [M] (64 bytes)
    const float md = P.mdelta[0], mz = P.mzp[0], inv = dgq_rcp(md);
    const float qmax = (float)((1 << P.a_bits) - 1), off = (float)(1 << (P.a_bits - 1));
    if (tid < b.M) rsum[tid] = 0.0f;
    __syncthreads();
    const TIn* x = reinterpret_cast<const TIn*>(b.x);
    for (int m = 0; m < b.M; ++m) {
        float part = 0.0f;
        for (int k = tid; k < Kq; k += 256) {
            float c = 0.0f;
            if (k < b.K) {
                float v = dgq_to_float(x[(int64_t)m * b.ldx + k]);
                if (b.pre_act == 1) v = dgq_silu(v);
                c = dgq_affine_code_fast(v, md, inv, mz, qmax) - off;
            }
            codes[m * Kq + k] = (int8_t)(int)c;
            part += c;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
        if (lane == 0) atomicAdd(&rsum[m], part);          // 4 exact small integers: order-independent
    }
    __syncthreads();
    TOut* y = reinterpret_cast<TOut*>(P.y);
    const int row_bytes = P.w_bits == 4 ? P.Kp / 2 : P.Kp;
    // W4 rows whose K fits one pass of the wave (K <= 2048: always, by SMALLM_MAX_K): the 16 weight rows of the wave and their
    // epilogue vectors are loaded up front — walked row by row, each row paid its own memory latency (41 us for the 23
    // time_emb_proj of an SD step, 12 MB of weights)
    if (P.w_bits == 4) {
        const int k0 = lane * 32;                           // 16 bytes = 32 k per lane
        const bool live = k0 < Kq;
        uint4 wq[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = min(n0 + wid * 16 + r, P.N - 1);
            wq[r] = live ? *reinterpret_cast<const uint4*>(P.wpacked + (int64_t)n * row_bytes + k0 / 2) : make_uint4(0u, 0u, 0u, 0u);
        }
        // Round 4: the cross-lane sums.  One 6-step shuffle chain per (row, m) — 16·M dependent ds_bpermute chains per wave — was the
        // kernel (47 us for M = 2, 82 us for M = 8 under the profiler, ~7 us per input row).  Now each lane keeps its partial dot
        // products of the 16 rows x 2 input rows, the wave transposes them through an LDS scratch ([pair][lane], stride 65: no bank
        // conflicts) and lane p < 32 adds the 64 partials of pair p = 2·r + mm and runs that output's epilogue.  Integer sums: same bits.
        int* red = reinterpret_cast<int*>(smem + codes_bytes + 64) + wid * (32 * 65);
        for (int m0 = 0; m0 < b.M; m0 += 2) {
            int4 c0[2], c1[2];
#pragma unroll
            for (int mm = 0; mm < 2; ++mm) {
                const int m = min(m0 + mm, b.M - 1);
                c0[mm] = live ? *reinterpret_cast<const int4*>(codes + m * Kq + k0) : make_int4(0, 0, 0, 0);
                c1[mm] = (live && k0 + 16 < Kq) ? *reinterpret_cast<const int4*>(codes + m * Kq + k0 + 16) : make_int4(0, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                // layout 1: rows with bit 4 set store the two 8-byte halves of a 32-chunk exchanged (wave-uniform)
                const bool sw = ((n0 + wid * 16 + r) & 16) != 0;
                const uint4 w = wq[r];
                const unsigned ww[4] = {sw ? w.z : w.x, sw ? w.w : w.y, sw ? w.x : w.z, sw ? w.y : w.w};
#pragma unroll
                for (int mm = 0; mm < 2; ++mm) {
                    const int cc[8] = {c0[mm].x, c0[mm].y, c0[mm].z, c0[mm].w, c1[mm].x, c1[mm].y, c1[mm].z, c1[mm].w};
                    int a = 0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {                            // dword j: k0+8j..+3 low nibbles, +4..+7 high nibbles
                        a = __builtin_amdgcn_sdot4(cc[2 * j], (int)(ww[j] & 0x0F0F0F0Fu), a, false);
                        a = __builtin_amdgcn_sdot4(cc[2 * j + 1], (int)((ww[j] >> 4) & 0x0F0F0F0Fu), a, false);
                    }
                    red[(2 * r + mm) * 65 + lane] = a;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // same-wave LDS round trip: no barrier
            if (lane < 32) {
                int tot = 0;
#pragma unroll 16
                for (int l = 0; l < 64; ++l) tot += red[lane * 65 + l];
                const int r = lane >> 1, m = m0 + (lane & 1), n = n0 + wid * 16 + r;
                if (m < b.M && n < P.N) {
                    // the per_m epilogue of dgq_gemm_wxa8, term for term
                    const float rs = rsum[m];
                    const float out = dgq_dequant<true>((float)tot, md, md * rs, md * (off - mz), P.alpha[n], P.zw[n], P.gamma[n], P.vn[n]);
                    y[(int64_t)m * P.ldy + n] = dgq_from_float<TOut>(out);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the scratch is rewritten by the next pair of input rows
        }
        return;
    }
    for (int r = 0; r < 16; ++r) {
        const int n = n0 + wid * 16 + r;
        if (n >= P.N) break;                                // wave-uniform
        const uint8_t* wrow = P.wpacked + (int64_t)n * row_bytes;
        int acc[SMALLM_MAX_M];
#pragma unroll
        for (int m = 0; m < SMALLM_MAX_M; ++m) acc[m] = 0;
        for (int k0 = lane * 16; k0 < Kq; k0 += 64 * 16) {
            const int4 w = *reinterpret_cast<const int4*>(wrow + k0);
#pragma unroll
            for (int m = 0; m < SMALLM_MAX_M; ++m) {
                if (m >= b.M) continue;
                const int4 c = *reinterpret_cast<const int4*>(codes + m * Kq + k0);
                int a = acc[m];
                a = __builtin_amdgcn_sdot4(c.x, w.x, a, false);
                a = __builtin_amdgcn_sdot4(c.y, w.y, a, false);
                a = __builtin_amdgcn_sdot4(c.z, w.z, a, false);
                a = __builtin_amdgcn_sdot4(c.w, w.w, a, false);
                acc[m] = a;
            }
        }
        const float al = P.alpha[n], zw = P.zw[n], ga = P.gamma[n], vn = P.vn[n];
#pragma unroll
        for (int m = 0; m < SMALLM_MAX_M; ++m) {
            if (m >= b.M) continue;
            int a = acc[m];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
            if (lane == 0) {
                // the per_m epilogue of dgq_gemm_wxa8, term for term
                const float rs = rsum[m];
                const float out = dgq_dequant<true>((float)a, md, md * rs, md * (off - mz), al, zw, ga, vn);
                y[(int64_t)m * P.ldy + n] = dgq_from_float<TOut>(out);
            }
        }
    }
}

extern "C" int dgq_linear_smallm_batch(const void* x, int x_dtype, int M, int K, int64_t ldx, int pre_act, int n_problems,
                                       const dgq_smallm_problem_t* probs, int y_dtype, void* stream) {
    DGQ_CHECK_ARG(x && probs, "dgq_linear_smallm_batch: null pointer");
    DGQ_CHECK_ARG(M >= 1 && M <= SMALLM_MAX_M && K >= 1 && K <= SMALLM_MAX_K && ldx >= K, "dgq_linear_smallm_batch: M=%d (<= %d) K=%d (<= %d)", M, SMALLM_MAX_M, K, SMALLM_MAX_K);
    DGQ_CHECK_ARG(n_problems >= 1 && n_problems <= SMALLM_MAX_PROBLEMS, "dgq_linear_smallm_batch: %d problems (max %d)", n_problems, SMALLM_MAX_PROBLEMS);
    DGQ_CHECK_ARG(pre_act == 0 || pre_act == 1, "dgq_linear_smallm_batch: pre_act");
    SmallMBatch b;
    b.n = n_problems; b.x = x; b.x_dtype = x_dtype; b.y_dtype = y_dtype; b.M = M; b.K = K; b.ldx = (int)ldx; b.pre_act = pre_act;
    int blocks = 0;
    for (int i = 0; i < n_problems; ++i) {
        const dgq_smallm_problem_t& q = probs[i];
        DGQ_CHECK_ARG(q.wpacked && q.alpha && q.zw && q.gamma && q.vn && q.mdelta && q.mzp && q.y, "dgq_linear_smallm_batch: null pointer in problem %d", i);
        DGQ_CHECK_ARG(q.N > 0 && q.ldy >= q.N && q.Kp >= K && q.Kp % DGQ_KTILE == 0 && (q.w_bits == 4 || q.w_bits == 8) && q.a_bits >= 2 && q.a_bits <= 8,
                      "dgq_linear_smallm_batch: bad problem %d", i);
        DGQ_CHECK_ARG((reinterpret_cast<uintptr_t>(q.wpacked) & 15) == 0, "dgq_linear_smallm_batch: wpacked alignment");
        SmallMProblem& P = b.p[i];
        P.wpacked = reinterpret_cast<const uint8_t*>(q.wpacked); P.alpha = q.alpha; P.zw = q.zw; P.gamma = q.gamma; P.vn = q.vn;
        P.mdelta = q.mdelta; P.mzp = q.mzp; P.y = q.y; P.ldy = q.ldy; P.N = q.N; P.Kp = q.Kp; P.w_bits = q.w_bits; P.a_bits = q.a_bits;
        P.block0 = blocks;
        blocks += (q.N + 63) / 64;
    }
    // codes [M][Kq] + row sums [16] + per wave a 32 x 65 int transposition scratch (66 KB at the M = 16, K = 2048 limits: opt in once)
    const size_t lds = ((((size_t)M * ((K + 15) & ~15)) + 15) & ~(size_t)15) + 64 + 4 * (32 * 65) * sizeof(int);
    static std::atomic<bool> attr_set[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !attr_set[dev].load(std::memory_order_acquire)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_smallm_kernel<float, float>), hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_smallm_kernel<__half, __half>), hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_smallm_kernel<__hip_bfloat16, __hip_bfloat16>), hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
        if (dev >= 0 && dev < 64) attr_set[dev].store(true, std::memory_order_release);
    }
    hipStream_t st = (hipStream_t)stream;
#define DGQ_SMALLM(TI, TO) hipLaunchKernelGGL((linear_smallm_kernel<TI, TO>), dim3(blocks), dim3(256), lds, st, b)
    if (x_dtype == DGQ_F32 && y_dtype == DGQ_F32) DGQ_SMALLM(float, float);
    else if (x_dtype == DGQ_F16 && y_dtype == DGQ_F16) DGQ_SMALLM(__half, __half);
    else if (x_dtype == DGQ_BF16 && y_dtype == DGQ_BF16) DGQ_SMALLM(__hip_bfloat16, __hip_bfloat16);
    else { dgq_set_error("dgq_linear_smallm_batch: x/y dtypes %d/%d", x_dtype, y_dtype); return DGQ_EINVAL; }
#undef DGQ_SMALLM
    return dgq_launch_status("dgq_linear_smallm_batch");
}
