// common.h -- shared device/host helpers for libnsgp_hip (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/nsgp.h"

#define NSGP_WAVE 64

static inline int nsgp_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
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
    double r = __builtin_amdgcn_rsq(x);                      // ~2^-26 relative; two Newton steps -> ~1 ulp
    r = r * (1.5 - 0.5 * x * r * r);
    r = r * (1.5 - 0.5 * x * r * r);
    return r;
}
template <typename T> __device__ __forceinline__ T t_fsqrt(T x);
template <> __device__ __forceinline__ float t_fsqrt<float>(float x) { return __builtin_amdgcn_sqrtf(x); }
template <> __device__ __forceinline__ double t_fsqrt<double>(double x) { return sqrt(x); }
template <typename T> __device__ __forceinline__ T t_fexp(T x);      // exp(x), x <= 0 on this path
template <> __device__ __forceinline__ float t_fexp<float>(float x) {
    return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f);
}
template <> __device__ __forceinline__ double t_fexp<double>(double x) { return exp(x); }
template <typename T> __device__ __forceinline__ T t_log(T x);
template <> __device__ __forceinline__ float t_log<float>(float x) { return logf(x); }
template <> __device__ __forceinline__ double t_log<double>(double x) { return log(x); }
