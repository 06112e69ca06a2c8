// mfma_f32_tiles: the K=64 Gramian of four gathered rows as 10 v_mfma_f32_16x16x4_f32 (upper 16x16 tiles) against 6
// v_mfma_f32_32x32x2_f32 (upper 32x32 tiles), matrix pipe only, at 1, 2 and 3 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_f32_tiles mfma_f32_tiles.hip && ./mfma_f32_tiles
#include <hip/hip_runtime.h>
#include <cstdio>
using f4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
template <int MODE>
__global__ __launch_bounds__(64, 3) void k(float *out, const float *in, int iters) {
    float a0 = in[threadIdx.x], a1 = in[threadIdx.x + 64], a2 = in[threadIdx.x + 128], a3 = in[threadIdx.x + 192];
    float s = 0;
    if (MODE == 0) {
        f4 acc[10];
        for (int t = 0; t < 10; ++t) acc[t] = f4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
            float v[4] = {a0, a1, a2, a3};
            asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
            int t = 0;
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int m = 0; m <= n; ++m) { acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[m], v[n], acc[t], 0, 0, 0); ++t; }
        }
        for (int t = 0; t < 10; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    } else {
        f32x16 acc[3];
        for (int t = 0; t < 3; ++t) acc[t] = (f32x16)(0.0f);
        for (int it = 0; it < iters; ++it) {
            float v[4] = {a0, a1, a2, a3};
            asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[0], v[0], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[0], v[1], acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[1], v[1], acc[2], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[2], v[2], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[2], v[3], acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[3], v[3], acc[2], 0, 0, 0);
        }
        for (int t = 0; t < 3; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    }
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int MODE>
static void run(const char *name, int grid, float *out, float *in) {
    const int iters = 4000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, out, in, 100);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, out, in, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s grid %5d: %8.3f ms = %6.1f ns per 4 gathered rows and SIMD-slot, %.1f cycles at 2.4 GHz per 4-row step and SIMD\n", name, grid, ms, ms * 1e6 / iters, ms * 1e-3 * 2.4e9 / ((double)grid * iters / 1024.0));
}
int main() {
    float *out, *in;
    (void)hipMalloc(&out, 8192 * 64 * 4); (void)hipMalloc(&in, 4096); (void)hipMemset(in, 0, 4096);
    for (int grid : {1024, 2048, 3072}) {
        run<0>("10 x mfma_f32_16x16x4 per 4 rows", grid, out, in);
        run<1>("6 x mfma_f32_32x32x2 per 4 rows", grid, out, in);
    }
    return 0;
}
