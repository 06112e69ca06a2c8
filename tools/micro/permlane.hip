// v_permlane16_swap / v_permlane32_swap on gfx950: which wait states a swap needs after the VALU instruction that produced its
// operands (hipcc 7.2 inserts none and the swap then reads stale data: half-wave sums came out as 2 x row 0).
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ float dpp_rows(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));
    return v;
}
#define SWAP16(PRE, POST)                                                                                               \
    {                                                                                                                   \
        int ra, rb;                                                                                                     \
        asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %2\n\t" PRE "v_permlane16_swap_b32 %0, %1\n\t" POST             \
                     : "=&v"(ra), "=&v"(rb) : "v"(i));                                                                  \
        f[64 * (q++) + threadIdx.x] = __builtin_bit_cast(float, ra) + __builtin_bit_cast(float, rb);                    \
    }
#define SWAP32(PRE, POST)                                                                                               \
    {                                                                                                                   \
        int ra, rb;                                                                                                     \
        asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %2\n\t" PRE "v_permlane32_swap_b32 %0, %1\n\t" POST             \
                     : "=&v"(ra), "=&v"(rb) : "v"(i));                                                                  \
        f[64 * (q++) + threadIdx.x] = __builtin_bit_cast(float, ra) + __builtin_bit_cast(float, rb);                    \
    }
__global__ void k(float *f) {
    float x = (float)(threadIdx.x * threadIdx.x % 17);
    const float rows = dpp_rows(x);
    const int i = __builtin_bit_cast(int, rows);
    int q = 0;
    f[64 * (q++) + threadIdx.x] = rows + __shfl_xor(rows, 16, 64);
    SWAP16("", "")
    SWAP16("s_nop 0\n\t", "")
    SWAP16("s_nop 1\n\t", "")
    SWAP16("s_nop 3\n\t", "")
    SWAP16("", "s_nop 3\n\t")
    SWAP16("s_nop 0\n\t", "s_nop 0\n\t")
    f[64 * (q++) + threadIdx.x] = rows + __shfl_xor(rows, 32, 64);
    SWAP32("", "")
    SWAP32("s_nop 0\n\t", "")
    SWAP32("s_nop 1\n\t", "")
    SWAP32("", "s_nop 1\n\t")
}
int main() {
    float *df, hf[64 * 12];
    (void)hipMalloc(&df, sizeof(hf));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, df);
    (void)hipMemcpy(hf, df, sizeof(hf), hipMemcpyDeviceToHost);
    const char *names[12] = {"row pair sums, bpermute", "swap16, no nop", "swap16, s_nop 0 before", "swap16, s_nop 1 before", "swap16, s_nop 3 before",
                             "swap16, s_nop 3 after", "swap16, s_nop 0 before and after", "half pair sums, bpermute", "swap32, no nop",
                             "swap32, s_nop 0 before", "swap32, s_nop 1 before", "swap32, s_nop 1 after"};
    for (int q = 0; q < 12; ++q) {
        printf("%-34s:", names[q]);
        for (int l = 0; l < 64; l += 8) printf(" %g", hf[64 * q + l]);
        printf("\n");
    }
    return 0;
}
