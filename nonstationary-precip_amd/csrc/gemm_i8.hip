// gemm_i8.hip -- the whitened projection A = W Kzx of a float32 SVGP layer on the int8 matrix cores, to better than
// float32 accuracy (an error-free-transformation product in the manner of Ozaki's scheme).
//
// Why: A = L^-1 Kzx (gpytorch VariationalStrategy.forward behind /root/reference/models/dgps.py:44-51; SURVEY A.3: a float64
// triangular solve, cast back) cannot be accumulated in float32 -- at kappa(Kzz) ~ 1e6 the terms |W||Kzx| ~ 1e2 cancel to
// O(1) and float32 accumulation costs 2e-4 of the posterior mean -- so round 2 ran it on the float64 MFMA
// (v_mfma_f64_16x16x4_f64, 78.6 TFLOP/s peak, 47 achieved: the single largest kernel of a DSVI step).  gfx950's int8 MFMA
// (v_mfma_i32_32x32x32_i8) is 64x the float64 rate per multiply-add, and integer accumulation is EXACT.  So:
//
//   W[m][k]  = wsc[m] * sum_{a<5}  dW_a[m][k] 128^-a    dW_a in [-64, 64]  (wsc[m] = 2^e / 64 >= max_k |W[m][k]| / 64)
//   K[k][j]  = ksc    * sum_{b<SK} dK_b[k][j] 128^-b    dK_b in [-64, 64]  (ksc    = 2^e / 64 >= os / 64: 0 < K <= os)
//   A[m][j]  = wsc[m] ksc sum_{l<LEV} 128^-l  sum_{a+b=l} sum_k dW_a[m][k] dK_b[k][j]
//
// in two instantiations: SK = 4 planes of Kzx, LEV = 5 levels (14 digit-plane products) for a layer whose output feeds the
// likelihood, and SK = 5, LEV = 6 (19 products) for a layer whose output is the next layer's input (nsgp/svgp.py chooses).
// Each inner sum is an int32 (|.| <= 5 x 1024 x 64^2 < 2^25 for M <= 1024, < 2^27 at the M = 4096 limit nsgp_i8_supported
// admits), the level sums are combined by Horner's rule in float64 and rounded ONCE to float32.  What is dropped: W below
// 2^-35 of its row maximum, K below 2^-28 (2^-35 with five planes) of os -- K is evaluated in float64 from the float32
// inputs, the digits carry 4 (11) more bits than a float32 Kzx, so this also replaces settings.hidden_kzx_f64's float64 Kzx
// -- and the digit pairs with a + b >= LEV.
// Measured against the float64 product at the headline shape (kappa 8.5e5): 9.5e-7 of max|A| with 14 products, < 8e-7 (the
// float32 rounding of A itself) with 19 -- float64 accumulation of the float32 Kzx: 5.6e-6; float32 accumulation: 6.6e-5
// (tests/test_gpu_i8.py; the numpy emulation the plane counts were chosen with: tools/probes/ozaki_emulation.py).
//
// Layout.  Digit planes are stored the way the kernel stages them, so a K-step's operand block is one contiguous run:
//   Wd[b][a][kb][h][Mp][16]   int8, kb = k / 32, h = (k % 32) / 16, Mp = M rounded up to 128 (zero rows), zeros for k > m
//   Kd[b][s][kb][h][np][16]   int8, np = n rounded up to 64 (zero columns)
// One 256-thread workgroup = a 128 x 64 output tile, 4 waves of 64 x 32 (two 32 x 32 MFMA tiles), LEV int32 level
// accumulators per MFMA tile (the five-plane form sits at 255 VGPRs, no spills).  LDS holds two stages of
// (5 x 128 + SK x 64) x 32 bytes (28 / 30 KB), images [plane][h][row][16] so that an operand fragment is one
// conflict-free ds_read_b128 per lane; the stages are filled by LDS-DMA (global_load_lds_dwordx4: a K-step's block is one
// contiguous run in memory, so no registers are spent on staging; NSGP_I8_DMA=0 builds the register-staged form); the K
// planes of a K-step stay in registers while the W planes stream through; two workgroups per CU.
#include <cstdlib>
#include "common.h"

namespace {

#ifndef NSGP_I8_DMA
#define NSGP_I8_DMA 1
#endif
constexpr int I8_BM = 128, I8_BN = 64, I8_BK = 32;
constexpr int I8_SW = 5;                                    // digit planes of W (K: 4 or 5, template SK; levels a + b < LEV)

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// ---- digit planes of W (one workgroup per row) ---------------------------------------------------------------------
__global__ __launch_bounds__(256) void i8_slice_w_kernel(const double* __restrict__ W, int64_t M, int64_t Mp, int64_t KB,
                                                         signed char* __restrict__ Wd, double* __restrict__ wsc) {
    __shared__ double lds[4];
    const int64_t b = blockIdx.y, m = blockIdx.x;
    signed char* out = Wd + b * (int64_t)I8_SW * KB * 2 * Mp * 16;
    const int64_t plane = KB * 2 * Mp * 16;
    if (m >= M) {                                           // padding rows: zeros
        for (int64_t k = threadIdx.x; k < KB * 32; k += 256)
            for (int a = 0; a < I8_SW; ++a) out[a * plane + ((k >> 5) * 2 + ((k >> 4) & 1)) * Mp * 16 + m * 16 + (k & 15)] = 0;
        return;
    }
    const double* row = W + (b * M + m) * M;
    double mx = 0.0;
    for (int64_t k = threadIdx.x; k <= m; k += 256) { const double v = fabs(row[k]); mx = v > mx ? v : mx; }
    // block max through LDS
    for (int off = 32; off > 0; off >>= 1) { const double o = __shfl_xor(mx, off, 64); mx = o > mx ? o : mx; }
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmax(fmax(lds[0], lds[1]), fmax(lds[2], lds[3]));
    int e = 0;
    (void)frexp(mx, &e);                                    // mx = f 2^e, f in [0.5, 1)  ->  2^e >= mx
    const double rs = mx > 0.0 ? ldexp(1.0, e) : 1.0;
    if (threadIdx.x == 0) wsc[b * M + m] = rs / 64.0;
    const double inv = 64.0 / rs;
    for (int64_t k = threadIdx.x; k < KB * 32; k += 256) {
        double t = (k <= m && k < M) ? row[k] * inv : 0.0;
        const int64_t o = ((k >> 5) * 2 + ((k >> 4) & 1)) * Mp * 16 + m * 16 + (k & 15);
#pragma unroll
        for (int a = 0; a < I8_SW; ++a) {
            const double d = rint(t);
            out[a * plane + o] = (signed char)(int)d;
            t = (t - d) * 128.0;
        }
    }
}

// ---- digit planes of Kzx, evaluated in float64 from the float32 kernel inputs ------------------------------------
// One thread = one column j and one k-block of 32: K[k][j] = os exp(-1/2 sum_d ((z[k][d] - x[j][d]) / ls[d])^2).
template <int D, int SK>
__global__ __launch_bounds__(256) void i8_rbf_build_kernel(const float* __restrict__ Z, const float* __restrict__ x, int64_t sx,
                                                           const float* __restrict__ ls, const float* __restrict__ os,
                                                           int64_t M, int64_t n, int64_t np, int64_t KB,
                                                           signed char* __restrict__ Kd, double* __restrict__ ksc,
                                                           float* __restrict__ Kf) {
    const int64_t b = blockIdx.z, kb = blockIdx.y;
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    __shared__ double zs[32][D];
    if (threadIdx.x < 32 * D) {
        const int kk = threadIdx.x / D, d = threadIdx.x % D;
        const int64_t k = kb * 32 + kk;
        zs[kk][d] = k < M ? (double)Z[(b * M + k) * D + d] / (double)ls[b * D + d] : 0.0;
    }
    __syncthreads();
    if (j >= np) return;
    const double osd = (double)os[b];
    int e = 0;
    (void)frexp(osd, &e);
    const double cs = ldexp(1.0, e);                        // >= os
    if (j == 0 && kb == 0) ksc[b] = cs / 64.0;
    const double sc = osd * 64.0 / cs;
    double xs[D];
#pragma unroll
    for (int d = 0; d < D; ++d) xs[d] = j < n ? (double)x[b * sx + j * D + d] / (double)ls[b * D + d] : 0.0;
    signed char* out = Kd + b * (int64_t)SK * KB * 2 * np * 16;
    const int64_t plane = KB * 2 * np * 16;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        int pk[SK][4] = {};
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int kk = h * 16 + q;
            double r2 = 0.0;
#pragma unroll
            for (int d = 0; d < D; ++d) { const double df = zs[kk][d] - xs[d]; r2 = __builtin_fma(df, df, r2); }
            const bool in = j < n && kb * 32 + kk < M;
            const double ev = in ? t_fexp<double>(-0.5 * r2) : 0.0;
            // the float32 Kzx the backward pass needs for Wbar = tril(Abar Kzx^T), rounded from the float64 value (the forward
            // pass itself never reads it): saves the backward's own build launch
            if (Kf && in) Kf[(b * M + kb * 32 + kk) * n + j] = (float)(osd * ev);
            double t = sc * ev;
#pragma unroll
            for (int s = 0; s < SK; ++s) {
                const double dg = rint(t);
                pk[s][q >> 2] |= ((int)dg & 0xFF) << (8 * (q & 3));
                t = (t - dg) * 128.0;
            }
        }
#pragma unroll
        for (int s = 0; s < SK; ++s) {
            v4i v = {pk[s][0], pk[s][1], pk[s][2], pk[s][3]};
            *reinterpret_cast<v4i*>(out + s * plane + (kb * 2 + h) * np * 16 + j * 16) = v;
        }
    }
}

// ---- the product --------------------------------------------------------------------------------------------------
// P64: float64 column-statistic partials (layers whose variance also accumulates in float64, see nsgp.h)
template <int P64, int SK, int LEV>
__global__ __launch_bounds__(256, 2) void i8_proj_kernel(const signed char* __restrict__ Wd, const double* __restrict__ wsc,
                                                         const signed char* __restrict__ Kd, const double* __restrict__ ksc,
                                                         const float* __restrict__ rv, int64_t M, int64_t n, int64_t Mp,
                                                         int64_t np, int64_t KB, int tiles_n, float* __restrict__ Y,
                                                         void* __restrict__ p0, void* __restrict__ p1, int part_rows) {
    extern __shared__ __attribute__((aligned(16))) unsigned char i8_smem[];
    constexpr int A_STAGE = I8_SW * 2 * I8_BM * 16, B_STAGE = SK * 2 * I8_BN * 16;       // 20480 + 8192 bytes
    unsigned char* As = i8_smem;                            // [2][A_STAGE]
    unsigned char* Bs = i8_smem + 2 * A_STAGE;              // [2][B_STAGE]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 32;
    const int64_t bb = blockIdx.y;
    const int tiles_m = (int)(Mp / I8_BM);
    // Anatomy of the two launches of a headline step (0.441 ms; switches that skipped the MFMAs / the staging / both):
    // staging alone 0.345, MFMAs + fragment reads alone 0.363, neither -- first-stage latency, barriers, the float64 Horner
    // epilogue and the stores of 5120 + 1024 short workgroups -- 0.187 ms.  The K loop moves 28 KB into LDS per 112 MFMAs
    // (2.6 GB per launch, 7.4 TB/s): larger tiles (128 x 128 with eight waves) are the lever, not the tile order.
    // (Tried and dropped: an XCD-grouped order -- workgroup 8 q + x, which the hardware places on XCD x, walking all row tiles of
    // column tiles x, x + 8, ... back to back so that a column tile's Kzx planes enter that L2 once: 0.441 -> 0.534 ms per
    // headline step, like the same experiment on the float32 GEMM.)
    const int bm = tiles_m - 1 - (int)blockIdx.x / tiles_n;          // long K ranges first
    const int bn = (int)blockIdx.x % tiles_n;
    const int64_t m0 = (int64_t)bm * I8_BM, n0 = (int64_t)bn * I8_BN;
    int nkb = (int)((m0 + I8_BM) / I8_BK);                  // lower triangular W: k < m0 + 128
    if (nkb > KB) nkb = (int)KB;
    const signed char* Wb = Wd + bb * (int64_t)I8_SW * KB * 2 * Mp * 16;
    const signed char* Kb = Kd + bb * (int64_t)SK * KB * 2 * np * 16;
    const int64_t wplane = KB * 2 * Mp * 16, kplane = KB * 2 * np * 16;

    // staging: piece q of A = (plane a, half h, row) = q / 256, (q % 256) / 128, q % 128; LDS offset 16 q.  5 per thread.
    const signed char* ga[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int q = tid + 256 * j;
        const int a = q >> 8, h = (q >> 7) & 1, row = q & 127;
        ga[j] = Wb + a * wplane + (int64_t)h * Mp * 16 + (m0 + row) * 16;
    }
    // piece q of B = (plane s, half h, col) = q / 128, (q % 128) / 64, q % 64: SK x 128 pieces, NPB per thread (the last
    // one of 5 planes only for the first 128 threads)
    constexpr int NPB = (SK * 128 + 255) / 256;
    const signed char* gb[NPB];
#pragma unroll
    for (int j = 0; j < NPB; ++j) {
        int q = tid + 256 * j;
        if (q >= SK * 128) q = SK * 128 - 1;                       // (clamped: loaded twice, stored once)
        const int s = q >> 7, h = (q >> 6) & 1, col = q & 63;
        gb[j] = Kb + s * kplane + (int64_t)h * np * 16 + (n0 + col) * 16;
    }
    const int64_t astep = 2 * Mp * 16, bstep = 2 * np * 16;          // bytes per k-block
#if NSGP_I8_DMA
    // Staging by LDS-DMA (global_load_lds_dwordx4): the digit planes are stored in the order the LDS image wants them, so a
    // wave's 64 consecutive pieces are 1 KB of global memory -> 1 KB of LDS, no staging registers (the 5 level accumulators
    // and the operand fragments need all 256), no ds_write.  The LDS base of a wave's run is wave-uniform (M0).
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    auto dma = [&](int kb, int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 5; ++j)
            __builtin_amdgcn_global_load_lds((gptr_t)(ga[j] + (int64_t)kb * astep),
                                             (lptr_t)(As + buf * A_STAGE + (wave * 64 + 256 * j) * 16), 16, 0, 0);
#pragma unroll
        for (int j = 0; j < NPB; ++j)
            if (wave * 64 + 256 * j < SK * 128)
                __builtin_amdgcn_global_load_lds((gptr_t)(gb[j] + (int64_t)kb * bstep),
                                                 (lptr_t)(Bs + buf * B_STAGE + (wave * 64 + 256 * j) * 16), 16, 0, 0);
    };
#endif
    v4i ra[5], rb[NPB];
    auto gload = [&](int kb) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 5; ++j) ra[j] = *reinterpret_cast<const v4i*>(ga[j] + (int64_t)kb * astep);
#pragma unroll
        for (int j = 0; j < NPB; ++j) rb[j] = *reinterpret_cast<const v4i*>(gb[j] + (int64_t)kb * bstep);
    };
    auto sstore = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 5; ++j) *reinterpret_cast<v4i*>(As + buf * A_STAGE + (tid + 256 * j) * 16) = ra[j];
#pragma unroll
        for (int j = 0; j < NPB; ++j)
            if (tid + 256 * j < SK * 128) *reinterpret_cast<v4i*>(Bs + buf * B_STAGE + (tid + 256 * j) * 16) = rb[j];
    };

    v16i acc[LEV][2];
#pragma unroll
    for (int l = 0; l < LEV; ++l)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[l][i][r] = 0;

    // this wave's rows reach k < m0 + wm0 + 64: the upper wave row has nothing to do in the tile's last two k-blocks
    const int wave_nkb = __builtin_amdgcn_readfirstlane((int)((m0 + wm0 + 64 + I8_BK - 1) / I8_BK));
    const int half = lane >> 5, lr = lane & 31;
#if NSGP_I8_DMA
    dma(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
    gload(0);
    sstore(0);
#endif
    __syncthreads();
    for (int kb = 0; kb < nkb; ++kb) {
        const int buf = kb & 1;
#if NSGP_I8_DMA
        if (kb + 1 < nkb) dma(kb + 1, buf ^ 1);           // (every wave left buffer buf ^ 1 behind the last barrier)
#else
        if (kb + 1 < nkb) gload(kb + 1);
#endif
        if (kb < wave_nkb) {
            // K's planes stay in registers for the whole k-block; W's planes pass through one at a time, the next one's
            // fragments read while the current one's MFMAs run
            v4i af[2][2], bf[SK];
#pragma unroll
            for (int s = 0; s < SK; ++s)
                bf[s] = *reinterpret_cast<const v4i*>(Bs + buf * B_STAGE + ((s * 2 + half) * I8_BN + wn0 + lr) * 16);
#pragma unroll
            for (int i = 0; i < 2; ++i)
                af[0][i] = *reinterpret_cast<const v4i*>(As + buf * A_STAGE + ((0 * 2 + half) * I8_BM + wm0 + 32 * i + lr) * 16);
#pragma unroll
            for (int a = 0; a < I8_SW; ++a) {
                if (a + 1 < I8_SW) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        af[(a + 1) & 1][i] = *reinterpret_cast<const v4i*>(
                            As + buf * A_STAGE + (((a + 1) * 2 + half) * I8_BM + wm0 + 32 * i + lr) * 16);
                }
#pragma unroll
                for (int s = 0; s < SK; ++s)
                    if (a + s < LEV) {
#pragma unroll
                        for (int i = 0; i < 2; ++i)
                            acc[a + s][i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[a & 1][i], bf[s], acc[a + s][i], 0, 0, 0);
                    }
            }
        }
#if NSGP_I8_DMA
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of the next stage have landed ...
#else
        if (kb + 1 < nkb) sstore(buf ^ 1);
#endif
        __syncthreads();                                   // ... and so have everybody else's
    }

    // epilogue: levels -> float64 -> one rounding to float32; column statistics from the float64 values
    const double kscale = ksc[bb];
    double sdot = 0.0, ssq = 0.0;
    const int64_t col = n0 + wn0 + lr;
    const float* rvb = rv ? rv + bb * M : nullptr;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t row = m0 + wm0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * half;
            double v = (double)acc[LEV - 1][i][r];
#pragma unroll
            for (int l = LEV - 2; l >= 0; --l) v = __builtin_fma(v, 0.0078125, (double)acc[l][i][r]);
            const bool in = row < M && col < n;
            v *= (in ? wsc[bb * M + row] : 0.0) * kscale;
            if (in) Y[(bb * M + row) * n + col] = (float)v;
            sdot += v * ((rvb && in) ? (double)rvb[row] : 0.0);
            ssq += v * v;
        }
    sdot += __shfl_xor(sdot, 32, 64);
    ssq += __shfl_xor(ssq, 32, 64);
    double* red = reinterpret_cast<double*>(i8_smem);       // [2 quantities][2 wave rows][64 columns]
    __syncthreads();
    if (lane < 32) {
        red[(0 + (wave >> 1)) * I8_BN + wn0 + lane] = sdot;
        red[(2 + (wave >> 1)) * I8_BN + wn0 + lane] = ssq;
    }
    __syncthreads();
    if (tid < I8_BN && n0 + tid < n) {
        const int64_t o = (bb * part_rows + bm) * n + n0 + tid;
        const double d = red[tid] + red[I8_BN + tid], q = red[2 * I8_BN + tid] + red[3 * I8_BN + tid];
        if (P64) {
            if (p0) reinterpret_cast<double*>(p0)[o] = d;
            reinterpret_cast<double*>(p1)[o] = q;
        } else {
            if (p0) reinterpret_cast<float*>(p0)[o] = (float)d;
            reinterpret_cast<float*>(p1)[o] = (float)q;
        }
    }
}

static inline int64_t i8_mp(int64_t M) { return cdiv64(M, I8_BM) * I8_BM; }
static inline int64_t i8_np(int64_t n) { return cdiv64(n, I8_BN) * I8_BN; }
static inline int64_t i8_kb(int64_t M) { return cdiv64(M, I8_BK); }

}  // namespace

extern "C" {

int nsgp_i8_supported(int64_t M) { return (M >= 1 && M <= 4096) ? 1 : 0; }       // int32 level sums: 5 M 64^2 < 2^31
size_t nsgp_i8_w_planes_bytes(int64_t batch, int64_t M) {
    return (batch > 0 && M > 0) ? (size_t)(batch * I8_SW * i8_kb(M) * 2 * i8_mp(M) * 16) : 0;
}
size_t nsgp_i8_k_planes_bytes(int64_t batch, int64_t M, int64_t n, int planes) {
    return (batch > 0 && M > 0 && n > 0 && (planes == 4 || planes == 5)) ? (size_t)(batch * planes * i8_kb(M) * 2 * i8_np(n) * 16) : 0;
}
size_t nsgp_i8_tiles(int64_t M) { return M > 0 ? (size_t)cdiv64(M, I8_BM) : 0; }

int nsgp_i8_slice_w_f64(const double* W, int64_t batch, int64_t M, void* Wd, double* wscale, void* stream) {
    if (!W) return -1; if (batch < 0) return -2; if (M < 0 || !nsgp_i8_supported(M > 0 ? M : 1)) return -3;
    if (!Wd) return -4; if (!wscale) return -5;
    if (batch == 0 || M == 0) return 0;
    if (batch > 65535) return -2;
    hipLaunchKernelGGL(i8_slice_w_kernel, dim3((unsigned)i8_mp(M), (unsigned)batch), dim3(256), 0, (hipStream_t)stream, W, M,
                       i8_mp(M), i8_kb(M), (signed char*)Wd, wscale);
    return nsgp_launch_status();
}

int nsgp_i8_rbf_build_f32(const float* Z, const float* x, int64_t x_batch_stride, const float* ls, const float* os,
                          int64_t batch, int64_t M, int64_t n, int D, int planes, void* Kd, double* kscale, float* Kzx_f32,
                          void* stream) {
    if (planes != 4 && planes != 5) return -10;
    if (!Z) return -1; if (!x) return -2; if (x_batch_stride < 0) return -3; if (!ls) return -4; if (!os) return -5;
    if (batch < 0) return -6; if (M < 0 || !nsgp_i8_supported(M > 0 ? M : 1)) return -7; if (n < 0) return -8;
    if (D < 1 || D > 4) return -9; if (!Kd) return -10; if (!kscale) return -11;
    if (batch == 0 || M == 0 || n == 0) return 0;
    if (batch > 65535 || i8_kb(M) > 65535) return -6;
    const int64_t np = i8_np(n), KB = i8_kb(M);
    dim3 grid((unsigned)cdiv64(np, 256), (unsigned)KB, (unsigned)batch);
    hipStream_t st = (hipStream_t)stream;
    signed char* kd = (signed char*)Kd;
#define NSGP_I8_BUILD(DD, SS) hipLaunchKernelGGL((i8_rbf_build_kernel<DD, SS>), grid, dim3(256), 0, st, Z, x, x_batch_stride, ls, os, M, n, np, KB, kd, kscale, Kzx_f32)
    if (planes == 4) {
        switch (D) { case 1: NSGP_I8_BUILD(1, 4); break; case 2: NSGP_I8_BUILD(2, 4); break; case 3: NSGP_I8_BUILD(3, 4); break;
                     default: NSGP_I8_BUILD(4, 4); break; }
    } else {
        switch (D) { case 1: NSGP_I8_BUILD(1, 5); break; case 2: NSGP_I8_BUILD(2, 5); break; case 3: NSGP_I8_BUILD(3, 5); break;
                     default: NSGP_I8_BUILD(4, 5); break; }
    }
#undef NSGP_I8_BUILD
    return nsgp_launch_status();
}

int nsgp_svgp_tri_gemm_colstats_i8(const void* Wd, const double* wscale, const void* Kd, const double* kscale,
                                   int planes, const float* rowvec, int64_t batch, int64_t M, int64_t n, float* Y,
                                   void* part_dot, void* part_sq, int64_t part_rows, int partials_f64, void* stream) {
    if (planes != 4 && planes != 5) return -5;
    if (!Wd) return -1; if (!wscale) return -2; if (!Kd) return -3; if (!kscale) return -4;
    if (batch < 0) return -6; if (M < 0 || !nsgp_i8_supported(M > 0 ? M : 1)) return -7; if (n < 0) return -8;
    if (!Y) return -9; if (!part_sq) return -11; if (part_rows < cdiv64(M > 0 ? M : 1, I8_BM)) return -12;
    if (batch == 0 || M == 0 || n == 0) return 0;
    const int64_t Mp = i8_mp(M), np = i8_np(n), KB = i8_kb(M);
    const int64_t tiles_n = np / I8_BN, tiles = (Mp / I8_BM) * tiles_n;
    if (tiles > 2147483647LL || batch > 65535) return -8;
    const size_t lds = 2 * (size_t)(I8_SW * 2 * I8_BM * 16 + planes * 2 * I8_BN * 16);
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)tiles, (unsigned)batch);
#define NSGP_I8_PROJ(PP, SS) hipLaunchKernelGGL((i8_proj_kernel<PP, SS, (SS == 5 ? 6 : 5)>), grid, dim3(256), lds, st, (const signed char*)Wd, wscale, \
        (const signed char*)Kd, kscale, rowvec, M, n, Mp, np, KB, (int)tiles_n, Y, part_dot, part_sq, (int)part_rows)
    if (planes == 4) { if (partials_f64) NSGP_I8_PROJ(1, 4); else NSGP_I8_PROJ(0, 4); }
    else { if (partials_f64) NSGP_I8_PROJ(1, 5); else NSGP_I8_PROJ(0, 5); }
#undef NSGP_I8_PROJ
    return nsgp_launch_status();
}

}  // extern "C"
