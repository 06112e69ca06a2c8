// rows.h -- one embedding row spread over the 64 lanes of a wavefront, and the fused
// optimizer update rules (cymf/optimizer.pyx:40-160).  Shared by the BPR / RelMF / GloVe kernels.
#pragma once
#include "common.h"

namespace cymf {

#if defined(__HIPCC__)

// A row of K factors held by one wavefront.
//   PACKED  (K == 64*R): lane l owns k = l*R .. l*R+R-1  -> ONE dwordx{R} access per lane,
//                        64*R*sizeof(T) contiguous bytes per wave instruction.
//   STRIDED (any K <= 64*R): lane l owns k = l, l+64, ... -> R dword accesses, each a
//                        contiguous 64*sizeof(T)-byte segment; lanes past K are masked and hold 0.
template <typename T, int R, bool PACKED>
struct Row {
    T v[R];
    static constexpr bool packed = PACKED;

    struct alignas(sizeof(T) * R) Vec { T x[R]; };

    // `fill` is what the lanes past K hold (strided layout).  Parameter rows use 0: they then drop out of every dot
    // product and gradient.  AdaGrad accumulator rows that stay in registers across several samples must use 1: a
    // masked lane has g = 0 and acc = 0, and lr * 0 * rsqrt(0) = NaN would enter the row's masked lanes and, through the
    // next sample's wave-wide dot product, everything (found by tests/test_gpu_bpr.py's boundary test at K = 8).
    __device__ __forceinline__ void load(const T *__restrict__ base, int K, int lane, T fill = T(0)) {
        if constexpr (PACKED) {
            Vec t = *reinterpret_cast<const Vec *>(base + lane * R);
#pragma unroll
            for (int r = 0; r < R; ++r) v[r] = t.x[r];
        } else {
            // R = ceil(K / 64): every element but the last of a lane exists whatever the lane, only r = R - 1 is masked; its
            // load is unconditional from a clamped index and selected afterwards (a branch per element otherwise: the GloVe
            // step kernel at K = 100 spent a tenth of its instructions on exec-mask bookkeeping)
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int k = lane + 64 * r;
                if (r + 1 < R) {
                    v[r] = base[k];
                } else {
                    const T x = base[k < K ? k : 0];
                    v[r] = k < K ? x : fill;
                }
            }
        }
    }
    // Agent-scope atomic loads, element by element: a row that other wavefronts update with float atomics is read
    // from where those atomics are performed, not from a cache level they bypass.
    __device__ __forceinline__ void load_coherent(const T *base, int K, int lane) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int k = kof(lane, r);
            v[r] = (PACKED || k < K) ? __hip_atomic_load(base + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : T(0);
        }
    }
    __device__ __forceinline__ void store(T *__restrict__ base, int K, int lane) const {
        if constexpr (PACKED) {
            Vec t;
#pragma unroll
            for (int r = 0; r < R; ++r) t.x[r] = v[r];
            *reinterpret_cast<Vec *>(base + lane * R) = t;
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int k = lane + 64 * r;
                if (r + 1 < R || k < K) base[k] = v[r];
            }
        }
    }
    // does slot r of this lane hold an element of the row?  (strided layout: only the last slot can be past K)
    __device__ __forceinline__ static bool live(int lane, int r, int K) { return PACKED || r + 1 < R || lane + 64 * r < K; }
    // element index of slot r (for atomics on single elements)
    __device__ __forceinline__ static int kof(int lane, int r) { return PACKED ? lane * R + r : lane + 64 * r; }
    __device__ __forceinline__ void fill(T x) {
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = x;
    }
};

// Update rules.  s0/s1 are the per-element optimizer state (unused ones are ignored):
//   SGD      cymf/optimizer.pyx:52-58    p -= lr*g
//   AdaGrad  cymf/optimizer.pyx:74-82    acc += g^2 (acc starts at 1, :69-70); p -= lr*g/sqrt(acc)
//   Adam     cymf/optimizer.pyx:150-160  constant "bias correction" 1/(1-beta), no step counter
template <typename T>
struct OptParams {
    T lr;
    T b1, b2, omb1, omb2, eps;   // Adam: beta1, beta2, (1-beta1), (1-beta2), epsilon
    T inv_omb1, inv_omb2;        // reciprocals, for the lock-free kernels
};

template <typename T>
inline OptParams<T> make_opt_params(double lr) {
    OptParams<T> p;
    p.lr = (T)lr;
    p.b1 = (T)0.9;
    p.b2 = (T)0.999;
    p.omb1 = (T)(1 - 0.9);
    p.omb2 = (T)(1 - 0.999);
    p.eps = (T)1e-8;
    p.inv_omb1 = (T)(1.0 / (1 - 0.9));
    p.inv_omb2 = (T)(1.0 / (1 - 0.999));
    return p;
}

__device__ __forceinline__ float fsqrt(float x) { return __fsqrt_rn(x); }
__device__ __forceinline__ double fsqrt(double x) { return sqrt(x); }

// HOG = true (throughput kernels): several wavefronts may add their deltas of a shared row's Adam
// moments, which can pair a (nearly) zeroed v with a live m; in sequential execution the two EMAs
// always satisfy m^2 <= v (1-b1)^2 / ((1-b2)(1-b1^2/b2)) (Cauchy-Schwarz), so clamping v from below
// with that bound is a no-op there and keeps the HOGWILD step bounded (|step| <= 2.3 lr).
template <typename T, int OPT, bool HOG = false>
__device__ __forceinline__ void opt_update(const OptParams<T> &o, T &p, T &s0, T &s1, T g) {
    if constexpr (OPT == CYMF_OPT_SGD) {
        p -= o.lr * g;
    } else if constexpr (OPT == CYMF_OPT_ADAGRAD) {
        s0 += g * g;
        if constexpr (HOG && sizeof(T) == 4) p -= o.lr * g * __frsqrt_rn(s0);   // lock-free kernels: v_rsq_f32 instead of sqrt + IEEE division
        else p -= o.lr * g / fsqrt(s0);
    } else {
        if constexpr (HOG) s1 = s1 < T(0) ? T(0) : s1;
        s0 = o.b1 * s0 + o.omb1 * g;
        s1 = o.b2 * s1 + o.omb2 * (g * g);
        T v = s1;
        if constexpr (HOG) {
            const T bound = s0 * s0 * (o.omb2 * (T(1) - o.b1 * o.b1 / o.b2) / (o.omb1 * o.omb1));
            v = v < bound ? bound : v;
        }
        if constexpr (HOG && sizeof(T) == 4) p -= o.lr * (s0 * o.inv_omb1) * __builtin_amdgcn_rcpf(fsqrt(v * o.inv_omb2) + o.eps);
        else p -= o.lr * (s0 / o.omb1) / (fsqrt(v / o.omb2) + o.eps);
    }
}

constexpr int opt_num_states(int opt) { return opt == CYMF_OPT_SGD ? 0 : (opt == CYMF_OPT_ADAGRAD ? 1 : 2); }

#endif  // __HIPCC__

}  // namespace cymf
