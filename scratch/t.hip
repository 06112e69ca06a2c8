#include <hip/hip_runtime.h>
__global__ void k(const float *in, float *out) {
    __shared__ float s[256];
    s[threadIdx.x] = in[threadIdx.x];
    __syncthreads();
    float v = 0.f;
    if ((threadIdx.x & 63) < 32) v = s[5];
    unsigned u = __builtin_bit_cast(unsigned, v);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    out[threadIdx.x] = __builtin_bit_cast(float, r[0]);
}
