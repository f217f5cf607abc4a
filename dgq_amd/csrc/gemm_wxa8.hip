// W4A8 / W8A8 GEMM on V_MFMA_I32_16X16X64_I8 with DGQ per-group dequantisation fused in.
//
//   A  = int8 activation codes [M][Kp]  (from dgq_quant_act; K already permuted so each DGQ group is a run of
//        64-wide chunks)
//   W  = int4 packed [N][Kp/2] (nibbles -> int8 on the way into LDS) or int8 [N][Kp]
//   Y  = fp [M][N]
//
// Tiling (wave64, gfx950): 128x128 block tile, BK = 128 (two MFMA K-slices), 256 threads = 2x2 waves,
// each wave owns 64x64 = 4x4 MFMA tiles: 64 int32 accumulators + (per-K mode) 64 fp32 accumulators.
// LDS: double-buffered A and W tiles of 128 rows x 128 B, 16-byte chunks XOR-swizzled with (row>>1)&7 so
// that every ds_read_b128 lane group covers all 64 banks once.  Global->register prefetch of tile t+1
// overlaps the MFMAs of tile t; one barrier per K step.
#include "dgq_common.h"

#define BM 128
#define BN 128
#define BK 128

struct GemmParams {
    const int8_t* codes;
    const float* rowsum;
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
};

__device__ __forceinline__ int swz(int row, int chunk) { return row * BK + ((chunk ^ ((row >> 1) & 7)) << 4); }

template <int WBITS, bool PER_M, typename TOut>
__global__ __launch_bounds__(256, 2) void gemm_wxa8_kernel(GemmParams p) {
    __shared__ __attribute__((aligned(16))) uint8_t smem[2 * 2 * BM * BK];   // [buf][A|W][128][128]
    uint8_t* sA = smem;
    uint8_t* sW = smem + 2 * BM * BK;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = tid >> 6;
    const int wave_m = wid >> 1, wave_n = wid & 1;
    const int n0 = blockIdx.x * BN;
    const int m0 = blockIdx.y * BM;
    const int nk = p.Kp / BK;

    uint4 ra[4];
    uint4 rw[WBITS == 4 ? 2 : 4];

    auto load_tile = [&](int kt) {
        const int k0 = kt * BK;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int id = tid + 256 * i;
            int row = id >> 3, c = id & 7;
            int m = m0 + row;
            ra[i] = (m < p.M) ? *reinterpret_cast<const uint4*>(p.codes + (int64_t)m * p.Kp + k0 + 16 * c)
                              : make_uint4(0, 0, 0, 0);
        }
        if (WBITS == 4) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                int id = tid + 256 * i;
                int row = id >> 2, pc = id & 3;
                int n = n0 + row;
                rw[i] = (n < p.N) ? *reinterpret_cast<const uint4*>(p.wpacked + (int64_t)n * (p.Kp / 2) + k0 / 2 + 16 * pc)
                                  : make_uint4(0, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int id = tid + 256 * i;
                int row = id >> 3, c = id & 7;
                int n = n0 + row;
                rw[i] = (n < p.N) ? *reinterpret_cast<const uint4*>(p.wpacked + (int64_t)n * p.Kp + k0 + 16 * c)
                                  : make_uint4(0, 0, 0, 0);
            }
        }
    };

    auto store_tile = [&](int buf) {
        uint8_t* a = sA + buf * BM * BK;
        uint8_t* w = sW + buf * BN * BK;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int id = tid + 256 * i;
            int row = id >> 3, c = id & 7;
            *reinterpret_cast<uint4*>(a + swz(row, c)) = ra[i];
        }
        if (WBITS == 4) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                int id = tid + 256 * i;
                int row = id >> 2, pc = id & 3;
                uint4 v = rw[i];
                // int4 -> int8: low nibbles = 4 consecutive k, high nibbles = the next 4 (dgq_pack_w4 layout)
                uint4 c0 = make_uint4(v.x & 0x0F0F0F0Fu, (v.x >> 4) & 0x0F0F0F0Fu, v.y & 0x0F0F0F0Fu, (v.y >> 4) & 0x0F0F0F0Fu);
                uint4 c1 = make_uint4(v.z & 0x0F0F0F0Fu, (v.z >> 4) & 0x0F0F0F0Fu, v.w & 0x0F0F0F0Fu, (v.w >> 4) & 0x0F0F0F0Fu);
                *reinterpret_cast<uint4*>(w + swz(row, 2 * pc)) = c0;
                *reinterpret_cast<uint4*>(w + swz(row, 2 * pc + 1)) = c1;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int id = tid + 256 * i;
                int row = id >> 3, c = id & 7;
                *reinterpret_cast<uint4*>(w + swz(row, c)) = rw[i];
            }
        }
    };

    v4i acc[4][4];
    v4f accf[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc[i][j] = (v4i){0, 0, 0, 0};
            accf[i][j] = (v4f){0.f, 0.f, 0.f, 0.f};
        }

    const int fr = lane & 15, fq = lane >> 4;

    load_tile(0);
    store_tile(0);
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) load_tile(kt + 1);
        const uint8_t* a = sA + cur * BM * BK;
        const uint8_t* w = sW + cur * BN * BK;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            v4i af[4], bf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int row = wave_m * 64 + i * 16 + fr;
                af[i] = *reinterpret_cast<const v4i*>(a + swz(row, 4 * h + fq));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int row = wave_n * 64 + j * 16 + fr;
                bf[j] = *reinterpret_cast<const v4i*>(w + swz(row, 4 * h + fq));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[i], bf[j], acc[i][j], 0, 0, 0);
            if (!PER_M) {
                const int chunk = kt * 2 + h;
                if (p.cflush[chunk]) {          // wave-uniform: last chunk of a DGQ group
                    const float sc = p.cdelta[chunk];
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) accf[i][j][r] += sc * (float)acc[i][j][r];
                            acc[i][j] = (v4i){0, 0, 0, 0};
                        }
                }
            }
        }
        if (kt + 1 < nk) store_tile(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // epilogue: C/D layout of the 16x16 MFMA: col = lane&15, row = (lane>>4)*4 + reg
    TOut* y = reinterpret_cast<TOut*>(p.y);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + wave_n * 64 + j * 16 + fr;
        if (n >= p.N) continue;
        const float al = p.alpha[n], zw = p.zw[n], ga = p.gamma[n];
        const float vn = PER_M ? p.vn[n] : 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wave_m * 64 + i * 16 + fq * 4 + r;
                if (m >= p.M) continue;
                const float rs = p.rowsum[m];
                float out;
                if (PER_M) {
                    const int li = m % p.L;
                    const float md = p.mdelta[li], mz = p.mzp[li];
                    out = al * md * ((float)acc[i][j][r] - zw * rs + (p.offset - mz) * vn) + ga;
                } else {
                    out = al * (accf[i][j][r] - zw * rs) + ga;
                }
                y[(int64_t)m * p.ldy + n] = dgq_from_float<TOut>(out);
            }
        }
    }
}

template <int WBITS, bool PER_M>
static int launch_gemm(const GemmParams& p, int y_dtype, hipStream_t st) {
    dim3 grid((p.N + BN - 1) / BN, (p.M + BM - 1) / BM), block(256);
    switch (y_dtype) {
        case DGQ_F32: hipLaunchKernelGGL((gemm_wxa8_kernel<WBITS, PER_M, float>), grid, block, 0, st, p); break;
        case DGQ_F16: hipLaunchKernelGGL((gemm_wxa8_kernel<WBITS, PER_M, __half>), grid, block, 0, st, p); break;
        case DGQ_BF16: hipLaunchKernelGGL((gemm_wxa8_kernel<WBITS, PER_M, __hip_bfloat16>), grid, block, 0, st, p); break;
        default: dgq_set_error("dgq_gemm_wxa8: unknown y dtype %d", y_dtype); return DGQ_EINVAL;
    }
    return dgq_launch_status("dgq_gemm_wxa8");
}

extern "C" int dgq_gemm_wxa8(const int8_t* codes, const float* rowsum, int M, int Kp,
                             const void* wpacked, int w_bits, int N,
                             int per_m, const float* cdelta, const uint8_t* cflush,
                             const float* mdelta, const float* mzp, int L, float offset,
                             const float* alpha, const float* zw, const float* gamma, const float* vn,
                             void* y, int y_dtype, int ldy, void* stream) {
    DGQ_CHECK_ARG(codes && rowsum && wpacked && alpha && zw && gamma && y, "dgq_gemm_wxa8: null pointer");
    DGQ_CHECK_ARG(M > 0 && N > 0 && Kp > 0 && Kp % DGQ_KTILE == 0, "dgq_gemm_wxa8: bad shape M=%d N=%d Kp=%d", M, N, Kp);
    DGQ_CHECK_ARG(w_bits == 4 || w_bits == 8, "dgq_gemm_wxa8: w_bits=%d unsupported", w_bits);
    DGQ_CHECK_ARG(ldy >= N, "dgq_gemm_wxa8: ldy < N");
    DGQ_CHECK_ARG((reinterpret_cast<uintptr_t>(codes) & 15) == 0 && (reinterpret_cast<uintptr_t>(wpacked) & 15) == 0,
                  "dgq_gemm_wxa8: codes/wpacked must be 16-byte aligned");
    if (per_m) {
        DGQ_CHECK_ARG(mdelta && mzp && vn && L >= 1, "dgq_gemm_wxa8: per_m needs mdelta/mzp/vn/L");
    } else {
        DGQ_CHECK_ARG(cdelta && cflush, "dgq_gemm_wxa8: per-K mode needs cdelta/cflush");
    }
    GemmParams p;
    p.codes = codes; p.rowsum = rowsum; p.M = M; p.Kp = Kp; p.N = N;
    p.wpacked = reinterpret_cast<const uint8_t*>(wpacked);
    p.cdelta = cdelta; p.cflush = cflush; p.mdelta = mdelta; p.mzp = mzp; p.L = per_m ? L : 1; p.offset = offset;
    p.alpha = alpha; p.zw = zw; p.gamma = gamma; p.vn = vn; p.y = y; p.ldy = ldy;
    hipStream_t st = (hipStream_t)stream;
    if (w_bits == 4) return per_m ? launch_gemm<4, true>(p, y_dtype, st) : launch_gemm<4, false>(p, y_dtype, st);
    return per_m ? launch_gemm<8, true>(p, y_dtype, st) : launch_gemm<8, false>(p, y_dtype, st);
}
