// wave_sum: the production version (csrc/common.h: 4 DPP row steps + row_bcast:15 / row_bcast:31 + one v_readlane) against the
// four-v_readlane form it replaced -- the results must be identical bit for bit on any input.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../cymf_amd/csrc/common.h"
using namespace cymf;
__device__ float rl(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }
__global__ void k(float *o, int n) {
    int bad = 0;
    for (int it = 0; it < n; ++it) {
        const unsigned h = (threadIdx.x * 2654435761u + it * 40503u) ^ (it << 7);
        const float x = __uint_as_float(0x3f000000u | (h & 0x007fffffu)) * ((h >> 24) & 1 ? -1.0f : 1.0f) * (float)(1 + (h >> 28));
        float v = x;
        v += dpp_f32<DPP_QUAD_PERM_1032>(v); v += dpp_f32<DPP_QUAD_PERM_2301>(v); v += dpp_f32<DPP_ROW_HALF_MIRROR>(v); v += dpp_f32<DPP_ROW_MIRROR>(v);
        const float a = (rl(v, 0) + rl(v, 16)) + (rl(v, 32) + rl(v, 48));
        const float b = wave_sum(x);
        bad += a != b;
    }
    if (threadIdx.x == 0) o[0] = (float)bad;
}
// The sum consumed ONLY by lane 0 inside a divergent branch (as the loss terms of the step kernels are): if the compiler sank
// the cross-lane tail into the branch, row_bcast would read switched-off lanes and the result would differ.
__global__ void k_lane0(float *o, int n) {
    float bad = 0.0f;
    for (int it = 0; it < n; ++it) {
        const unsigned h = (threadIdx.x * 2246822519u + it * 40503u) ^ (it << 9);
        const float x = __uint_as_float(0x3f000000u | (h & 0x007fffffu)) * ((h >> 24) & 1 ? -1.0f : 1.0f) * (float)(1 + (h >> 28));
        float v = x;
        v += dpp_f32<DPP_QUAD_PERM_1032>(v); v += dpp_f32<DPP_QUAD_PERM_2301>(v); v += dpp_f32<DPP_ROW_HALF_MIRROR>(v); v += dpp_f32<DPP_ROW_MIRROR>(v);
        const float a = (rl(v, 0) + rl(v, 16)) + (rl(v, 32) + rl(v, 48));
        const float b = wave_sum(x);
        if (threadIdx.x == 0) {               // only lane 0 ever looks at either
            if ((h & 3u) != 3u) bad += (a != b) ? 1.0f : 0.0f;
        }
    }
    if (threadIdx.x == 0) o[1] = bad;
}
int main() {
    float *d, h[2];
    (void)hipMalloc(&d, 8);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 100000);
    hipLaunchKernelGGL(k_lane0, dim3(1), dim3(64), 0, 0, d, 100000);
    (void)hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    printf("wave_sum: %d of 100000 random inputs differ from the four-readlane form\n", (int)h[0]);
    printf("wave_sum consumed by lane 0 only: %d of 100000 differ\n", (int)h[1]);
    return h[0] != 0 || h[1] != 0;
}
