// misc.hip -- small stream-ordered helpers: Cholesky-backward symmetrisation, precision casts,
// counter-based normals (Philox4x32-10), fused Adam, library identification.
#include <cmath>
#include "common.h"

namespace {

template <typename T>
__global__ void phi_sym_kernel(const T* __restrict__ P, T* __restrict__ S, int64_t n, int64_t ld, int64_t sP,
                               int64_t batch) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= batch * n * n) return;
    const int64_t b = idx / (n * n), e = idx % (n * n);
    const int64_t i = e / n, j = e % n;
    const T* Pb = P + b * sP;
    // Phi keeps the lower triangle and halves the diagonal; S = Phi + Phi^T
    S[b * sP + i * ld + j] = (i >= j) ? Pb[i * ld + j] : Pb[j * ld + i];
}

template <typename T>
__global__ void scale_diag_kernel(T* __restrict__ P, int64_t n, int64_t ld, int64_t sP, int64_t batch, T factor) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= batch * n) return;
    const int64_t b = idx / n, i = idx % n;
    P[b * sP + i * ld + i] *= factor;
}

template <typename TS, typename TD>
__global__ void cast_kernel(const TS* __restrict__ src, int64_t lds, TD* __restrict__ dst, int64_t ldd, int64_t rows,
                            int64_t cols) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * cols) return;
    const int64_t r = idx / cols, c = idx % cols;
    dst[r * ldd + c] = (TD)src[r * lds + c];
}

// ---- Philox4x32-10 (Salmon et al. 2011) ------------------------------------------------------------
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    const uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
    c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
}

__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}

// eps[s, i, c] for global row row0 + i: counter = (row_lo, row_hi, s | (c/4) << 20, stream_id_lo),
// key = (seed_lo, seed_hi ^ stream_id_hi); words (2q, 2q+1) -> Box-Muller pair q; column c uses
// normal number (c % 4): pair (c%4)/2, cos for even, sin for odd.
template <typename T>
__global__ void philox_normal_kernel(uint64_t seed, uint64_t stream_id, const int64_t* __restrict__ step_dev,
                                     int64_t row0, int64_t S, int64_t n, int64_t b, T* __restrict__ eps) {
    // a device-side step counter (hipGraph replays freeze host scalars) replaces the high stream word
    if (step_dev) stream_id = ((uint64_t)step_dev[0] << 32) | (stream_id & 0xFFFFFFFFull);
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t bq = (b + 3) / 4;
    if (idx >= S * n * bq) return;
    const int64_t cq = idx % bq, i = (idx / bq) % n, s = idx / (bq * n);
    const uint64_t row = (uint64_t)(row0 + i);
    uint32_t c[4] = {(uint32_t)row, (uint32_t)(row >> 32), (uint32_t)s | ((uint32_t)cq << 20), (uint32_t)stream_id};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32) ^ (uint32_t)(stream_id >> 32));
    const double two_m32 = 2.3283064365386963e-10;       // 2^-32
    double z[4];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const double u1 = ((double)c[2 * q] + 0.5) * two_m32;
        const double u2 = ((double)c[2 * q + 1] + 0.5) * two_m32;
        const double r = sqrt(-2.0 * log(u1));
        const double th = 6.283185307179586476925286766559 * u2;
        z[2 * q] = r * cos(th);
        z[2 * q + 1] = r * sin(th);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t col = cq * 4 + k;
        if (col < b) eps[(s * n + i) * b + col] = (T)z[k];
    }
}

// 4 parameters per thread (16-byte accesses); the bias corrections of a device-side step count are computed by one
// thread per workgroup (two double-precision pow calls per THREAD made the 3.2 M-parameter update take 41 us).
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t n, float lr,
                                                   float b1, float b2, float eps, float bc1, float bc2_sqrt,
                                                   float gscale, const int64_t* __restrict__ step_dev) {
    __shared__ float sbc[2];
    if (step_dev) {                       // bias corrections from the device-side step count (graph replay)
        if (threadIdx.x == 0) {
            const double st = (double)step_dev[0];
            sbc[0] = (float)(1.0 - pow((double)b1, st));
            sbc[1] = (float)sqrt(1.0 - pow((double)b2, st));
        }
        __syncthreads();
        bc1 = sbc[0];
        bc2_sqrt = sbc[1];
    }
    const float step_size = lr / bc1, ibc2 = 1.0f / bc2_sqrt;
    const int64_t i0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i0 >= n) return;
    auto upd = [&](float& pi, float gi, float& mi, float& vi) {
        const float gr = gi * gscale;
        mi = b1 * mi + (1.0f - b1) * gr;
        vi = b2 * vi + (1.0f - b2) * gr * gr;
        pi -= step_size * mi / (sqrtf(vi) * ibc2 + eps);
    };
    const bool vec = (i0 + 3 < n) && (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) % 16 == 0);
    if (vec) {
        float4 pp = *reinterpret_cast<float4*>(p + i0), mm = *reinterpret_cast<float4*>(m + i0);
        float4 vv = *reinterpret_cast<float4*>(v + i0);
        const float4 gg = *reinterpret_cast<const float4*>(g + i0);
        upd(pp.x, gg.x, mm.x, vv.x); upd(pp.y, gg.y, mm.y, vv.y); upd(pp.z, gg.z, mm.z, vv.z); upd(pp.w, gg.w, mm.w, vv.w);
        *reinterpret_cast<float4*>(p + i0) = pp;
        *reinterpret_cast<float4*>(m + i0) = mm;
        *reinterpret_cast<float4*>(v + i0) = vv;
    } else {
        for (int64_t i = i0; i < n && i < i0 + 4; ++i) upd(p[i], g[i], m[i], v[i]);
    }
}

template <typename T>
int phi_impl(const T* P, T* S, int64_t n, int64_t ld, int64_t sP, int64_t batch, void* stream) {
    if (!P) return -1; if (!S) return -2; if (n < 0) return -3; if (ld < n) return -4; if (batch < 0) return -6;
    const int64_t tot = batch * n * n;
    if (tot == 0) return 0;
    hipLaunchKernelGGL((phi_sym_kernel<T>), dim3((unsigned)cdiv64(tot, 256)), dim3(256), 0, (hipStream_t)stream, P, S,
                       n, ld, sP, batch);
    return nsgp_launch_status();
}

template <typename T>
int scale_diag_impl(T* P, int64_t n, int64_t ld, int64_t sP, int64_t batch, T factor, void* stream) {
    if (!P) return -1; if (n < 0) return -2; if (ld < n) return -3; if (batch < 0) return -5;
    if (batch * n == 0) return 0;
    hipLaunchKernelGGL((scale_diag_kernel<T>), dim3((unsigned)cdiv64(batch * n, 256)), dim3(256), 0, (hipStream_t)stream,
                       P, n, ld, sP, batch, factor);
    return nsgp_launch_status();
}

template <typename TS, typename TD>
int cast_impl(const TS* src, int64_t lds, TD* dst, int64_t ldd, int64_t rows, int64_t cols, void* stream) {
    if (!src) return -1; if (lds < cols) return -2; if (!dst) return -3; if (ldd < cols) return -4;
    if (rows < 0) return -5; if (cols < 0) return -6;
    const int64_t tot = rows * cols;
    if (tot == 0) return 0;
    hipLaunchKernelGGL((cast_kernel<TS, TD>), dim3((unsigned)cdiv64(tot, 256)), dim3(256), 0, (hipStream_t)stream, src,
                       lds, dst, ldd, rows, cols);
    return nsgp_launch_status();
}

template <typename T>
int philox_impl(uint64_t seed, uint64_t stream_id, const int64_t* step_dev, int64_t row0, int64_t S, int64_t n,
                int64_t b, T* eps, void* stream) {
    if (row0 < 0) return -3; if (S < 0 || S >= (1 << 20)) return -4; if (n < 0) return -5;
    if (b < 0 || b > 4 * 4096) return -6; if (!eps) return -7;
    const int64_t tot = S * n * ((b + 3) / 4);
    if (tot == 0) return 0;
    hipLaunchKernelGGL((philox_normal_kernel<T>), dim3((unsigned)cdiv64(tot, 256)), dim3(256), 0, (hipStream_t)stream,
                       seed, stream_id, step_dev, row0, S, n, b, eps);
    return nsgp_launch_status();
}

// ---- sustained matrix-core rate of THIS chip under load (bench.py's roofline context) ----------------------------------
// Every wave of a chip-filling grid issues `iters` x 8 independent register-only MFMAs (no memory, no LDS) and stamps the
// shader clock (s_memtime) and the 100 MHz wall clock (s_memrealtime) around them: the ratio is the clock the chip holds
// while all matrix cores are busy (MI355X: ~2.1 GHz, not the 2.4 GHz the data-sheet peak is quoted at), and flops / time of
// the launch is the rate no GEMM can exceed.  KIND 0: v_mfma_f32_32x32x2_f32, 1: v_mfma_f64_16x16x4_f64, 2: v_mfma_i32_32x32x32_i8.
template <int KIND>
__global__ __launch_bounds__(256) void mfma_rate_probe_kernel(int iters, unsigned long long* __restrict__ out, float* __restrict__ sink) {
    typedef float v16f __attribute__((ext_vector_type(16)));
    typedef double v4d __attribute__((ext_vector_type(4)));
    typedef int v16i __attribute__((ext_vector_type(16)));
    typedef int v4i __attribute__((ext_vector_type(4)));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float keep = 0.f;
    if constexpr (KIND == 0) {
        v16f acc[8];
        for (int q = 0; q < 8; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
        const float a = 1.0f + threadIdx.x * 1e-3f, b = 1.0f - threadIdx.x * 1e-3f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int q = 0; q < 8; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[q], 0, 0, 0);
        }
        for (int q = 0; q < 8; ++q) keep += acc[q][0];
    } else if constexpr (KIND == 1) {
        v4d acc[8];
        for (int q = 0; q < 8; ++q) for (int r = 0; r < 4; ++r) acc[q][r] = 0.0;
        const double a = 1.0 + threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-3;
        for (int it = 0; it < iters; it += 16) {         // 16 rounds per trip: hipcc moves the float64 accumulators between
#pragma unroll                                            // AGPRs and VGPRs at the loop's back edge (128 copies)
            for (int u = 0; u < 16; ++u)
#pragma unroll
                for (int q = 0; q < 8; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[q], 0, 0, 0);
        }
        for (int q = 0; q < 8; ++q) keep += (float)acc[q][0];
    } else {
        v16i acc[8];
        for (int q = 0; q < 8; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0;
        const v4i a = {(int)threadIdx.x, 0x01010101, 0x01020304, 0x7f7f7f7f}, b = {0x01010101, (int)threadIdx.x, 0x04030201, 0x01010101};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int q = 0; q < 8; ++q) acc[q] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[q], 0, 0, 0);
        }
        for (int q = 0; q < 8; ++q) keep += (float)acc[q][0];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) {
        const size_t wv = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
        out[2 * wv] = t1 - t0;
        out[2 * wv + 1] = r1 - r0;
    }
    if (keep == 123.456f) sink[0] = keep;                 // keeps the accumulators alive
}

}  // namespace

extern "C" {

int nsgp_abi_version(void) { return 1; }
const char* nsgp_build_arch(void) { return "gfx950"; }

int nsgp_chol_bwd_phi_sym_f32(const float* P, float* S, int64_t n, int64_t ld, int64_t sP, int64_t batch,
                              void* stream) {
    return phi_impl<float>(P, S, n, ld, sP, batch, stream);
}
int nsgp_chol_bwd_phi_sym_f64(const double* P, double* S, int64_t n, int64_t ld, int64_t sP, int64_t batch,
                              void* stream) {
    return phi_impl<double>(P, S, n, ld, sP, batch, stream);
}
int nsgp_scale_diag_f32(float* P, int64_t n, int64_t ld, int64_t sP, int64_t batch, float factor, void* stream) {
    return scale_diag_impl<float>(P, n, ld, sP, batch, factor, stream);
}
int nsgp_scale_diag_f64(double* P, int64_t n, int64_t ld, int64_t sP, int64_t batch, double factor, void* stream) {
    return scale_diag_impl<double>(P, n, ld, sP, batch, factor, stream);
}
int nsgp_cast_f64_to_f32(const double* src, int64_t lds, float* dst, int64_t ldd, int64_t rows, int64_t cols,
                         void* stream) {
    return cast_impl<double, float>(src, lds, dst, ldd, rows, cols, stream);
}
int nsgp_cast_f32_to_f64(const float* src, int64_t lds, double* dst, int64_t ldd, int64_t rows, int64_t cols,
                         void* stream) {
    return cast_impl<float, double>(src, lds, dst, ldd, rows, cols, stream);
}
int nsgp_philox_normal_f32(uint64_t seed, uint64_t stream_id, const int64_t* step_dev, int64_t row0, int64_t S,
                           int64_t n, int64_t b, float* eps, void* stream) {
    return philox_impl<float>(seed, stream_id, step_dev, row0, S, n, b, eps, stream);
}
int nsgp_philox_normal_f64(uint64_t seed, uint64_t stream_id, const int64_t* step_dev, int64_t row0, int64_t S,
                           int64_t n, int64_t b, double* eps, void* stream) {
    return philox_impl<double>(seed, stream_id, step_dev, row0, S, n, b, eps, stream);
}
int nsgp_adam_step_f32(float* p, const float* g, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                       float beta2, float eps, int64_t step, const int64_t* step_dev, float grad_scale,
                       void* stream) {
    if (!p) return -1; if (!g) return -2; if (!exp_avg) return -3; if (!exp_avg_sq) return -4; if (n < 0) return -5;
    if (step < 1 && !step_dev) return -10;
    if (n == 0) return 0;
    const double st = step < 1 ? 1.0 : (double)step;
    const double bc1 = 1.0 - pow((double)beta1, st);
    const double bc2 = 1.0 - pow((double)beta2, st);
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)cdiv64(n, 1024)), dim3(256), 0, (hipStream_t)stream, p, g, exp_avg,
                       exp_avg_sq, n, lr, beta1, beta2, eps, (float)bc1, (float)sqrt(bc2), grad_scale, step_dev);
    return nsgp_launch_status();
}

int nsgp_mfma_rate_probe(int kind, int64_t workgroups, int iters, uint64_t* out, float* sink, void* stream) {
    // out: 2 x 4 x workgroups uint64 (per wave: shader-clock cycles, 100 MHz ticks); one wave issues iters x 8 MFMAs
    if (kind < 0 || kind > 2) return -1; if (workgroups <= 0 || workgroups > 1048576) return -2; if (iters <= 0 || iters % 16) return -3;
    if (!out) return -4; if (!sink) return -5;
    hipStream_t st = (hipStream_t)stream;
    unsigned long long* o = (unsigned long long*)out;
    if (kind == 0) hipLaunchKernelGGL((mfma_rate_probe_kernel<0>), dim3((unsigned)workgroups), dim3(256), 0, st, iters, o, sink);
    else if (kind == 1) hipLaunchKernelGGL((mfma_rate_probe_kernel<1>), dim3((unsigned)workgroups), dim3(256), 0, st, iters, o, sink);
    else hipLaunchKernelGGL((mfma_rate_probe_kernel<2>), dim3((unsigned)workgroups), dim3(256), 0, st, iters, o, sink);
    return nsgp_launch_status();
}
}  // extern "C"
