// mfma_valu_mix: do VALU instructions run UNDER the matrix pipe, or beside it?  Every wave issues, per step, ten independent
// v_mfma_f32_16x16x4_f32 and NV independent v_fma_f32 (interleaved one MFMA : NV/10 FMAs), at 1..4 waves per SIMD.
// If the two overlapped, the step would cost max(10 x 32, 4 NV) cycles; if the SIMD runs one or the other, 320 + 4 NV.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_valu_mix mfma_valu_mix.hip && ./mfma_valu_mix
#include <hip/hip_runtime.h>
#include <cstdio>
using f4 = __attribute__((ext_vector_type(4))) float;
template <int NV>
__global__ __launch_bounds__(64, 4) void k(float *out, const float *in, int iters) {
    float a0 = in[threadIdx.x], a1 = in[threadIdx.x + 64];
    f4 acc[10];
    for (int t = 0; t < 10; ++t) acc[t] = f4{0, 0, 0, 0};
    float f[8];
    for (int q = 0; q < 8; ++q) f[q] = a0 + q;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 10; ++t) {
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, a1, acc[t], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < NV / 10; ++u) {
                const int q = (t * (NV / 10) + u) & 7;
                f[q] = __builtin_fmaf(f[q], a1, a0);
                asm volatile("" : "+v"(f[q]));
            }
        }
    }
    float s = 0;
    for (int t = 0; t < 10; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    for (int q = 0; q < 8; ++q) s += f[q];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int NV>
static void run(int grid, float *out, float *in) {
    const int iters = 4000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NV>, dim3(grid), dim3(64), 0, 0, out, in, 100);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<NV>, dim3(grid), dim3(64), 0, 0, out, in, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double per_step = ms * 1e-3 * 2.4e9 / ((double)grid / 1024.0 * iters);
    printf("10 MFMA + %3d VALU per step, %d wave(s) per SIMD: %7.1f cycles (at 2.4 GHz) per step and SIMD   [max(320, %d) = overlap, %d = one or the other]\n",
           NV, grid / 1024, per_step, 4 * NV, 320 + 4 * NV);
}
int main() {
    float *out, *in;
    (void)hipMalloc(&out, 8192 * 64 * 4); (void)hipMalloc(&in, 4096); (void)hipMemset(in, 0, 4096);
    for (int grid : {1024, 2048, 4096}) {
        run<0>(grid, out, in);
        run<20>(grid, out, in);
        run<40>(grid, out, in);
        run<80>(grid, out, in);
    }
    return 0;
}
