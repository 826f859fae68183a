// common.h -- shared device/host helpers for libnsgp_hip (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <mutex>
#include <unordered_set>
#include "../../include/nsgp.h"

#define NSGP_WAVE 64

static inline int nsgp_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

// Opt a kernel in to more than 64 KB of dynamic LDS.  The attribute is per (kernel, device): keyed by hipGetDevice()
// so that a second GPU in the same process gets it too, and guarded by a mutex because launches also come from
// autograd's backward thread.
static inline void nsgp_opt_in_lds(const void* kern, size_t lds) {
    if (lds <= 65536) return;
    static std::mutex mu;
    static std::unordered_set<uint64_t> done;
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t key = (uint64_t)(uintptr_t)kern * 64u + (uint64_t)(dev & 63);
    std::lock_guard<std::mutex> lk(mu);
    if (done.insert(key).second)
        (void)hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

template <typename T> __device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// sum over a 256-thread block; result valid in thread 0. `lds` holds >= 4 elements.
template <typename T> __device__ __forceinline__ T block_sum_256(T v, T* lds) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) lds[w] = v;
    __syncthreads();
    return lds[0] + lds[1] + lds[2] + lds[3];
}

template <typename T> __device__ __forceinline__ T t_exp(T x);
template <> __device__ __forceinline__ float t_exp<float>(float x) { return expf(x); }
template <> __device__ __forceinline__ double t_exp<double>(double x) { return exp(x); }
template <typename T> __device__ __forceinline__ T t_sqrt(T x);
template <> __device__ __forceinline__ float t_sqrt<float>(float x) { return sqrtf(x); }
template <> __device__ __forceinline__ double t_sqrt<double>(double x) { return sqrt(x); }
// fp32: single hardware instructions (v_rcp_f32 / v_sqrt_f32 / v_exp_f32, ~1 ulp) keep the build kernels
// HBM-bound instead of VALU-bound; fp64 uses the full-precision library sequences.
template <typename T> __device__ __forceinline__ T t_rcp(T x);
template <> __device__ __forceinline__ float t_rcp<float>(float x) { return __builtin_amdgcn_rcpf(x); }
template <> __device__ __forceinline__ double t_rcp<double>(double x) { return 1.0 / x; }
template <typename T> __device__ __forceinline__ T t_rsqrt(T x);     // 1/sqrt(x), full precision
template <> __device__ __forceinline__ float t_rsqrt<float>(float x) { return __builtin_amdgcn_rsqf(x); }
template <> __device__ __forceinline__ double t_rsqrt<double>(double x) {
    // v_rsq_f64 is good to 5.2e-8 relative (tools/probes/rsq_f64_accuracy.hip); one cubic step
    // r (1 - e)^(-1/2) = r (1 + e/2 + 3 e^2 / 8 + O(e^3)), e = 1 - x r^2, lands within 1 ulp in 5 instructions
    // (two quadratic Newton steps: 7 instructions, 2.2 ulp).
    const double r = __builtin_amdgcn_rsq(x);
    const double e = __builtin_fma(-(x * r), r, 1.0);
    const double p = __builtin_fma(e, 0.375, 0.5);
    return __builtin_fma(r * e, p, r);
}
template <typename T> __device__ __forceinline__ T t_fsqrt(T x);
template <> __device__ __forceinline__ float t_fsqrt<float>(float x) { return __builtin_amdgcn_sqrtf(x); }
template <> __device__ __forceinline__ double t_fsqrt<double>(double x) { return sqrt(x); }
__device__ __forceinline__ float t_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double t_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
template <typename T> __device__ __forceinline__ T t_fexp(T x);      // exp(x), x <= 0 on this path
template <> __device__ __forceinline__ float t_fexp<float>(float x) {
    return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f);
}
template <> __device__ __forceinline__ double t_fexp<double>(double x) {
    // exp(x) = 2^k e^r with k = rint(x log2 e), r = x - k ln2 (two-part ln2), |r| <= ln2/2, e^r by its degree-13
    // Taylor polynomial (truncation 4e-18); arguments below the denormal range clamp to it (x <= 0 here, so no
    // overflow branch).  ~1 ulp, about 3/4 of the instructions of the library exp() in the build kernels.
    // clamp on the high dword only (2 instructions): anything below -745 becomes a value in (-746, -745]
    {
        const int hi = __double2hiint(x), lo = __double2loint(x);
        x = __hiloint2double(x < -745.0 ? (int)0xC0874800 : hi, lo);
    }
    const double k = __builtin_rint(x * 1.44269504088896340736);
    double r = __builtin_fma(k, -6.93147180369123816490e-01, x);
    r = __builtin_fma(k, -1.90821492927058770002e-10, r);
    // v_fma_f64 spelled out: the compiler otherwise emits a v_mov_b64 + v_fmac_f64 pair per coefficient
    double p = 1.6059043836821613e-10;                 // 1/13!
#define NSGP_EXP_FMA(c) asm("v_fma_f64 %0, %1, %2, %3" : "=v"(p) : "v"(p), "v"(r), "v"((double)(c)))
    NSGP_EXP_FMA(2.08767569878681e-09);                // 1/12!
    NSGP_EXP_FMA(2.505210838544172e-08);               // 1/11!
    NSGP_EXP_FMA(2.755731922398589e-07);               // 1/10!
    NSGP_EXP_FMA(2.7557319223985893e-06);              // 1/9!
    NSGP_EXP_FMA(2.48015873015873e-05);                // 1/8!
    NSGP_EXP_FMA(1.984126984126984e-04);               // 1/7!
    NSGP_EXP_FMA(1.3888888888888889e-03);              // 1/6!
    NSGP_EXP_FMA(8.333333333333333e-03);               // 1/5!
    NSGP_EXP_FMA(4.1666666666666664e-02);              // 1/4!
    NSGP_EXP_FMA(1.6666666666666666e-01);              // 1/3!
#undef NSGP_EXP_FMA
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    return __builtin_ldexp(p, (int)k);
}
template <typename T> __device__ __forceinline__ T t_log(T x);
template <> __device__ __forceinline__ float t_log<float>(float x) { return logf(x); }
template <> __device__ __forceinline__ double t_log<double>(double x) { return log(x); }
