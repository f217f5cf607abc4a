// Issue rate of the int8 MFMA shapes on gfx950: which instruction a 32-wide DGQ chunk should be contracted on.
// build: hipcc -O3 --offload-arch=gfx950 tools/mfma_rate.hip -o tools/bin/mfma_rate ; run: tools/bin/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int KIND>
__global__ __launch_bounds__(256) void rate_kernel(int iters, int* out) {
    const int lane = threadIdx.x;
    v4i a = {lane, lane * 3, lane ^ 5, 7}, b = {lane + 1, 9, lane * 7, 3};
    long a8 = ((long)lane << 32) | (lane * 3), b8 = ((long)(lane + 1) << 32) | 11;
    int acc_sum = 0;
    if (KIND == 0) {            // 16x16x64
        v4i c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        for (int i = 0; i < iters; ++i) {
            c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c3, 0, 0, 0);
        }
        acc_sum = c0[0] + c1[1] + c2[2] + c3[3];
    } else if (KIND == 1) {     // 32x32x32
        v16i c0 = {}, c1 = {}, c2 = {}, c3 = {};
        for (int i = 0; i < iters; ++i) {
            c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c3, 0, 0, 0);
        }
        acc_sum = c0[0] + c1[1] + c2[2] + c3[3];
    } else if (KIND == 2) {     // 16x16x32 (gfx942 form)
        v4i c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        for (int i = 0; i < iters; ++i) {
            c0 = __builtin_amdgcn_mfma_i32_16x16x32_i8(a8, b8, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_i32_16x16x32_i8(a8, b8, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_i32_16x16x32_i8(a8, b8, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_i32_16x16x32_i8(a8, b8, c3, 0, 0, 0);
        }
        acc_sum = c0[0] + c1[1] + c2[2] + c3[3];
    } else {                    // 32x32x16 (gfx942 form)
        v16i c0 = {}, c1 = {}, c2 = {}, c3 = {};
        for (int i = 0; i < iters; ++i) {
            c0 = __builtin_amdgcn_mfma_i32_32x32x16_i8(a8, b8, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_i32_32x32x16_i8(a8, b8, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_i32_32x32x16_i8(a8, b8, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_i32_32x32x16_i8(a8, b8, c3, 0, 0, 0);
        }
        acc_sum = c0[0] + c1[1] + c2[2] + c3[3];
    }
    if (acc_sum == 0x7fffffff) out[0] = acc_sum;
}

template <int KIND>
static void run(const char* name, double ops_per_mfma, int* out) {
    const int iters = 20000, blocks = 256 * 2;   // 2 blocks of 4 waves per CU: two waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(rate_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, 100, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, iters, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = (double)blocks * 4 * iters * 4;
    const double per_simd = mfmas / (256.0 * 4);
    printf("%-14s %8.3f ms  %7.1f TOP/s  %6.2f ns per MFMA per SIMD (= %.1f cycles at 2.4 GHz)\n", name, ms,
           mfmas * ops_per_mfma / (ms * 1e-3) / 1e12, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);
}

int main() {
    int* out;
    hipMalloc(&out, 4);
    run<0>("i32_16x16x64", 2.0 * 16 * 16 * 64, out);
    run<1>("i32_32x32x32", 2.0 * 32 * 32 * 32, out);
    run<2>("i32_16x16x32", 2.0 * 16 * 16 * 32, out);
    run<3>("i32_32x32x16", 2.0 * 32 * 32 * 16, out);
    return 0;
}
