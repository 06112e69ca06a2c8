// glds_gather: the gather of 256-byte factor rows (K = 64 floats) that bounds the WMF K=64 sweep (csrc/wmf.hip), two ways:
//   regs : as wmf_row_reg_kernel does it -- dword loads straight into the 32x32x2 MFMA operand layout (2 rows x 128 B per
//          instruction), 8 steps of loads in flight per wave, then their 24 MFMAs;
//   ring : global_load_lds_dwordx4 (4 rows x 256 B per instruction) into a per-wave LDS ring of R KiB, one ds_read_b128 per lane
//          hands every lane back its own 16 bytes = the four 16-column chunk operands of v_mfma_f32_16x16x4_f32 (column chunk
//          q = columns 4 i + q), 10 MFMAs per 4 rows; bytes in flight are LDS, not registers.
// First a mechanics check of the LDS-DMA path (every lane gets back the bytes it asked for), then rates on a 7 MB and a 35 MB table.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

using f4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
typedef __attribute__((address_space(3))) unsigned char lds_u8;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

__device__ __forceinline__ uint32_t lds_offset(const void *p) { return (uint32_t)(uintptr_t)(lds_u8 *)p; }

// one LDS-DMA wave instruction: lane l's 16 bytes at gsrc -> LDS byte lds_dst + 16 l  (M0 saved and restored: it is the compiler's)
__device__ __forceinline__ void glds16(const void *gsrc, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

__global__ __launch_bounds__(64) void check_kernel(const float *__restrict__ Y, const int32_t *__restrict__ idx, int n_steps, float *__restrict__ out) {
    __shared__ __attribute__((aligned(1024))) unsigned char ring[2048];
    const int lane = threadIdx.x;
    const char *Yb = reinterpret_cast<const char *>(Y);
    for (int s = 0; s < n_steps; ++s) {
        const int32_t i = idx[4 * s + (lane >> 4)];
        const uint32_t slot = lds_offset(ring) + 1024u * (s & 1);
        glds16(Yb + (size_t)i * 256 + (lane & 15) * 16, slot);
        wait_vm<0>();
        const f4 v = *reinterpret_cast<const f4 *>(ring + 1024 * (s & 1) + lane * 16);
        wait_lgkm0();
        reinterpret_cast<f4 *>(out)[s * 64 + lane] = v;
    }
}

template <int R, bool MFMA>
__global__ __launch_bounds__(64, 3) void ring_kernel(const float *__restrict__ Y, const int32_t *__restrict__ idx, int64_t n_idx, float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char ring[];
    const int lane = threadIdx.x;
    const char *Yb = reinterpret_cast<const char *>(Y);
    const int64_t per = (n_idx / 4 / gridDim.x) * 4;
    const int64_t b = (int64_t)blockIdx.x * per;
    const int64_t steps = per / 4;
    const uint32_t base = lds_offset(ring);
    f4 acc[10];
#pragma unroll
    for (int t = 0; t < 10; ++t) acc[t] = f4{0, 0, 0, 0};
    f4 sum = {0, 0, 0, 0};
    auto issue = [&](int64_t s, int slot) {
        const int64_t p = b + 4 * (s < steps ? s : 0);
        const int32_t i0 = idx[p], i1 = idx[p + 1], i2 = idx[p + 2], i3 = idx[p + 3];   // (uniform: scalar loads)
        const int g = lane >> 4;
        const int32_t i = g == 0 ? i0 : g == 1 ? i1 : g == 2 ? i2 : i3;
        glds16(Yb + (uint32_t)i * 256u + (uint32_t)(lane & 15) * 16u, base + 1024u * (uint32_t)slot);
    };
    for (int s = 0; s < R; ++s) issue(s, s);
    int slot = 0;
    for (int64_t s = 0; s < steps; ++s) {
        wait_vm<R - 1>();
        const f4 v = *reinterpret_cast<const f4 *>(ring + 1024 * slot + lane * 16);
        if (MFMA) {
            int t = 0;
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int m = 0; m <= n; ++m) { acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[m], v[n], acc[t], 0, 0, 0); ++t; }
        }
        sum += v;
        wait_lgkm0();
        issue(s + R, slot);
        slot = slot + 1 == R ? 0 : slot + 1;
    }
    wait_vm<0>();
#pragma unroll
    for (int t = 0; t < 10; ++t) sum += acc[t];
    out[(size_t)blockIdx.x * 64 + lane] = sum[0] + sum[1] + sum[2] + sum[3];
}

template <bool MFMA>
__global__ __launch_bounds__(64, 3) void regs_kernel(const float *__restrict__ Y, const int32_t *__restrict__ idx, int64_t n_idx, float *__restrict__ out) {
    const int lane = threadIdx.x, li = lane & 31, lh = lane >> 5;
    const char *Yb = reinterpret_cast<const char *>(Y);
    const int64_t per = (n_idx / 64 / gridDim.x) * 64;
    const int64_t b = (int64_t)blockIdx.x * per;
    f32x16 acc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) acc[t] = (f32x16)(0.0f);
    float sum = 0;
    for (int64_t pb = b; pb < b + per; pb += 64) {
        const int32_t myidx = idx[pb + lane];
        for (int s0 = 0; s0 < 32; s0 += 8) {
            float ch[8][2];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int32_t i = __shfl(myidx, 2 * (s0 + u) + lh, 64);
                const uint32_t off = (uint32_t)i * 256u + (uint32_t)(li * 4);
                ch[u][0] = *reinterpret_cast<const float *>(Yb + off);
                ch[u][1] = *reinterpret_cast<const float *>(Yb + off + 128);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (MFMA) {
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ch[u][0], ch[u][0], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ch[u][0], ch[u][1], acc[1], 0, 0, 0);
                    acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(ch[u][1], ch[u][1], acc[2], 0, 0, 0);
                }
                sum += ch[u][0] + ch[u][1];
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += acc[t][r];
    out[(size_t)blockIdx.x * 64 + lane] = sum;
}

template <typename F>
static float time_ms(F launch) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch();
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, 0));
        launch();
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    CK(hipGetLastError());
    return best;
}

template <int R, bool MFMA>
static void run_ring(const float *Y, const int32_t *idx, int64_t n, float *out, int grid, const char *what) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(ring_kernel<R, MFMA>), hipFuncAttributeMaxDynamicSharedMemorySize, R * 1024));
    const float ms = time_ms([&] { hipLaunchKernelGGL((ring_kernel<R, MFMA>), dim3(grid), dim3(64), R * 1024, 0, Y, idx, n, out); });
    printf("  ring R=%2d %s %s: %.3f ms  %.2f TB/s\n", R, MFMA ? "+10 mfma16" : "loads only", what, ms, (double)n * 256 / ms * 1e-9);
}

int main() {
    // ---- mechanics
    {
        const int rows = 1000, steps = 50;
        std::vector<float> Y((size_t)rows * 64);
        for (size_t i = 0; i < Y.size(); ++i) Y[i] = (float)i;
        std::vector<int32_t> idx(4 * steps);
        for (int i = 0; i < 4 * steps; ++i) idx[i] = (int32_t)((i * 7919u + 13u) % rows);
        float *dY, *dout; int32_t *didx;
        CK(hipMalloc(&dY, Y.size() * 4)); CK(hipMalloc(&didx, idx.size() * 4)); CK(hipMalloc(&dout, (size_t)steps * 256 * 4));
        CK(hipMemcpy(dY, Y.data(), Y.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(didx, idx.data(), idx.size() * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(check_kernel, dim3(1), dim3(64), 0, 0, dY, didx, steps, dout);
        CK(hipDeviceSynchronize());
        std::vector<float> out((size_t)steps * 256);
        CK(hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int s = 0; s < steps; ++s)
            for (int l = 0; l < 64; ++l)
                for (int e = 0; e < 4; ++e)
                    bad += out[((size_t)s * 64 + l) * 4 + e] != Y[(size_t)idx[4 * s + (l >> 4)] * 64 + (l & 15) * 4 + e];
        printf("glds mechanics: %d of %d values wrong\n", bad, steps * 256);
        if (bad) return 1;
        CK(hipFree(dY)); CK(hipFree(didx)); CK(hipFree(dout));
    }
    // ---- rates
    const int64_t n = 20'000'000 / 64 * 64;
    const int grid = 256 * 12;
    for (int rows : {27000, 138000}) {
        std::vector<int32_t> idx((size_t)n);
        uint64_t st = 88172645463325252ull;
        for (auto &v : idx) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; v = (int32_t)(st % (uint64_t)rows); }
        float *dY, *dout; int32_t *didx;
        CK(hipMalloc(&dY, (size_t)rows * 256)); CK(hipMemset(dY, 0, (size_t)rows * 256));
        CK(hipMalloc(&didx, (size_t)n * 4)); CK(hipMalloc(&dout, (size_t)grid * 64 * 4));
        CK(hipMemcpy(didx, idx.data(), (size_t)n * 4, hipMemcpyHostToDevice));
        char what[64];
        snprintf(what, sizeof what, "(%d rows, %.0f MB)", rows, rows * 256e-6);
        printf("table %s, %lld gathered rows, %d one-wave workgroups\n", what, (long long)n, grid);
        float ms = time_ms([&] { hipLaunchKernelGGL((regs_kernel<false>), dim3(grid), dim3(64), 0, 0, dY, didx, n, dout); });
        printf("  regs loads only: %.3f ms  %.2f TB/s\n", ms, (double)n * 256 / ms * 1e-9);
        ms = time_ms([&] { hipLaunchKernelGGL((regs_kernel<true>), dim3(grid), dim3(64), 0, 0, dY, didx, n, dout); });
        printf("  regs +3 mfma32 per 2 rows: %.3f ms  %.2f TB/s\n", ms, (double)n * 256 / ms * 1e-9);
        run_ring<4, false>(dY, didx, n, dout, grid, what);
        run_ring<8, false>(dY, didx, n, dout, grid, what);
        run_ring<12, false>(dY, didx, n, dout, grid, what);
        run_ring<4, true>(dY, didx, n, dout, grid, what);
        run_ring<8, true>(dY, didx, n, dout, grid, what);
        run_ring<12, true>(dY, didx, n, dout, grid, what);
        CK(hipFree(dY)); CK(hipFree(didx)); CK(hipFree(dout));
    }
    return 0;
}
