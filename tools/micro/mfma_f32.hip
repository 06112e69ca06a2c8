// Microbenchmark: issue rate of v_mfma_f32_32x32x2_f32 on ten independent accumulators (the WMF Gramian loop at K=128),
// alone and with the loop's VALU companions, at one and two waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_f32 mfma_f32.hip && ./mfma_f32
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int MODE>
__global__ __launch_bounds__(64, 2) void k(float *out, const float *in, int iters) {
    f32x16 acc[10];
    for (int t = 0; t < 10; ++t) acc[t] = (f32x16)(0.0f);
    float a0 = in[threadIdx.x], a1 = in[threadIdx.x + 64], a2 = in[threadIdx.x + 128], a3 = in[threadIdx.x + 192];
    float s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    for (int it = 0; it < iters; ++it) {
        float v0 = a0, v1 = a1, v2 = a2, v3 = a3;
        if (MODE >= 1) {   // the selects and column sums of the real loop
            const bool ok = (it + (int)threadIdx.x) >= 0;
            v0 = ok ? a0 : 0.0f; v1 = ok ? a1 : 0.0f; v2 = ok ? a2 : 0.0f; v3 = ok ? a3 : 0.0f;
            asm volatile("" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
        }
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(v0, v0, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v0, v1, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(v1, v1, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(v0, v2, acc[3], 0, 0, 0);
        acc[4] = __builtin_amdgcn_mfma_f32_32x32x2f32(v1, v2, acc[4], 0, 0, 0);
        acc[5] = __builtin_amdgcn_mfma_f32_32x32x2f32(v2, v2, acc[5], 0, 0, 0);
        acc[6] = __builtin_amdgcn_mfma_f32_32x32x2f32(v0, v3, acc[6], 0, 0, 0);
        acc[7] = __builtin_amdgcn_mfma_f32_32x32x2f32(v1, v3, acc[7], 0, 0, 0);
        acc[8] = __builtin_amdgcn_mfma_f32_32x32x2f32(v2, v3, acc[8], 0, 0, 0);
        acc[9] = __builtin_amdgcn_mfma_f32_32x32x2f32(v3, v3, acc[9], 0, 0, 0);
        if (MODE >= 1) { s0 += v0; s1 += v1; s2 += v2; s3 += v3; }
        if (MODE >= 2) {   // a gather per step as in the real loop: 4 dword loads from a 14 MB table
            const unsigned r = (unsigned)(it * 2654435761u + blockIdx.x * 40503u) % 26744u;
            const float *y = in + (size_t)r * 128 + (threadIdx.x & 31);
            a0 = y[0]; a1 = y[32]; a2 = y[64]; a3 = y[96];
        }
    }
    float s = s0 + s1 + s2 + s3;
    for (int t = 0; t < 10; ++t)
        for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int MODE>
static void run(const char *name, int grid, float *out, float *in) {
    const int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, out, in, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, out, in, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma = (double)grid * iters * 10;
    printf("%-44s grid %5d: %8.3f ms, %6.1f TFLOP/s, %6.1f cycles per MFMA and SIMD at 2.4 GHz\n", name, grid, ms, mfma * 4096 / ms / 1e9,
           ms * 1e-3 * 2.4e9 / (mfma / 1024.0));
}

int main() {
    float *out, *in;
    hipMalloc(&out, 8192 * 64 * 4);
    hipMalloc(&in, (size_t)26744 * 128 * 4 + 4096);
    hipMemset(in, 0, (size_t)26744 * 128 * 4 + 4096);
    for (int grid : {1024, 2048, 4096}) {
        run<0>("MFMAs only", grid, out, in);
        run<1>("+ 4 v_cndmask, 4 v_add per step", grid, out, in);
        run<2>("+ gather of the next step's operands", grid, out, in);
    }
    return 0;
}
