// gemm_bf16.hip -- the "bf16 forward" of BASELINE configs[4] (3-layer DSVI DeepGP, M = 2048: "bf16 forward / fp32
// Cholesky panels"): the two forward projections of a whitened SVGP layer,
//
//     A = W Kzx            W = chol(Kzz)^-1 (lower triangular)
//     C = Lq^T A           Lq = chol of the variational covariance (lower triangular)
//
// (gpytorch VariationalStrategy.forward behind /root/reference/models/dgps.py:44-51, driven by :92-98) on
// v_mfma_f32_32x32x16_bf16 -- bf16 operands, float32 accumulation, float32 outputs -- with the column statistics
// (sum A m, sum A^2, sum C^2) reduced in the epilogue exactly like the float32 path (gemm.hip, EPI 1).  The backward
// pass, the Cholesky and everything else stay float32 / float64.
//
// Both products are "NT" products of k-contiguous bf16 operands:
//     Y[m][n] = sum_k  P[m][k] * Qt[n][k]
//   product 1: P = bf16(W) (M x M; k <= m),  Qt = Kxz = Kzx^T (n x M), written in bf16 by nsgp_rbf_build_t_bf16
//   product 2: P = bf16(tril(Lq)^T) (k >= m), Qt = A^T (n x M), the bf16 transposed copy product 1 writes beside A
// so a lane's MFMA fragment (8 consecutive k of one row / column) is ONE 16-byte LDS read for either operand, and
// the triangular operands need no masking in the kernel (their zeros are in memory), only K-range skipping.
//
// Tile 128 x 128 x 64, 256 threads = 2 x 2 waves of 64 x 64 (2 x 2 MFMA tiles), LDS rows padded to 144 B
// (conflict-free ds_read_b128 / ds_write_b128), register-staged with TWO K-tiles of global loads in flight, one
// barrier per K-tile.  This is the simple two-phase structure (about a third of the 2.5 PFLOP/s bf16 peak in the
// container guide's ladder): a K-tile is only 16 MFMAs per wave, so the kernel is bound by the global -> LDS stream,
// not by the matrix pipe -- already ~5x the float32 MFMA rate on these products.
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int LDT = BK + 8;                 // LDS row stride in bf16 elements (144 B)

struct Bf16Args {
    int64_t M, N;                           // output rows (= K, the operands are M x M) and columns
    int64_t sP, sQ, sY, sYT;                // batch strides (elements)
    int tri;                                // 1: P lower (k <= m), 2: P upper (k >= m)
    int tiles_m, tiles_n;
};

__device__ __forceinline__ int crow(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// Y = P Qt^T (+ transposed bf16 copy YT, + column-statistic partials), see the file header.
__global__ __launch_bounds__(256, 2) void bf16_proj_kernel(Bf16Args g, const __bf16* __restrict__ P,
                                                           const __bf16* __restrict__ Qt, const float* __restrict__ rowvec,
                                                           float* __restrict__ Y, __bf16* __restrict__ YT,
                                                           float* __restrict__ part_dot, float* __restrict__ part_sq) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __bf16 (*Ps)[BM * LDT] = reinterpret_cast<__bf16 (*)[BM * LDT]>(smem);
    __bf16 (*Qs)[BN * LDT] = reinterpret_cast<__bf16 (*)[BN * LDT]>(smem + 2 * BM * LDT * sizeof(__bf16));
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 64;
    const int bb = blockIdx.y;
    // longest K range first (lower: large m; upper: small m)
    int bm = (int)blockIdx.x / g.tiles_n, bn = (int)blockIdx.x % g.tiles_n;
    if (g.tri == 1) bm = g.tiles_m - 1 - bm;
    const int64_t m0 = (int64_t)bm * BM, n0 = (int64_t)bn * BN;
    const int64_t K = g.M;
    int64_t kbeg = 0, kend = K;
    if (g.tri == 1) kend = m0 + BM < K ? m0 + BM : K;
    if (g.tri == 2) kbeg = m0 / BK * BK;
    const int nt = (int)((kend - kbeg + BK - 1) / BK);
    const __bf16* Pb = P + bb * g.sP;
    const __bf16* Qb = Qt + bb * g.sQ;

    // staging: a 128 x 64 bf16 tile = 1024 16-byte chunks, 4 per thread; chunk c -> (row c / 8, k chunk c % 8):
    // 8 consecutive lanes read one row's 128 contiguous bytes
    const int crow0 = tid >> 3, cch = tid & 7;
    uint4 ra[2][4], rb[2][4];
    auto gload = [&](int t, uint4* a, uint4* b) __attribute__((always_inline)) {
        const int64_t k0 = kbeg + (int64_t)t * BK + cch * 8;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int64_t r = m0 + crow0 + 32 * p, c = n0 + crow0 + 32 * p;
            // rows / columns past the matrix edge read row 0 (never stored), k past K reads zeros via the clamp below
            const bool okr = r < g.M && k0 + 7 < K, okc = c < g.N && k0 + 7 < K;
            a[p] = okr ? *reinterpret_cast<const uint4*>(Pb + r * K + k0) : make_uint4(0, 0, 0, 0);
            b[p] = okc ? *reinterpret_cast<const uint4*>(Qb + c * K + k0) : make_uint4(0, 0, 0, 0);
        }
    };
    auto sstore = [&](int buf, const uint4* a, const uint4* b) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            *reinterpret_cast<uint4*>(&Ps[buf][(crow0 + 32 * p) * LDT + cch * 8]) = a[p];
            *reinterpret_cast<uint4*>(&Qs[buf][(crow0 + 32 * p) * LDT + cch * 8]) = b[p];
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int fr = lane & 31, fh = lane >> 5;
    auto compute = [&](int buf) __attribute__((always_inline)) {
        const __bf16* ps = &Ps[buf][(wm0 + fr) * LDT + fh * 8];
        const __bf16* qs = &Qs[buf][(wn0 + fr) * LDT + fh * 8];
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            bf16x8 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const bf16x8*>(ps + i * 32 * LDT + ks * 16);
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const bf16x8*>(qs + j * 32 * LDT + ks * 16);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    };
    if (nt > 0) {
        gload(0, ra[0], rb[0]);
        if (nt > 1) gload(1, ra[1], rb[1]);
        sstore(0, ra[0], rb[0]);
        __syncthreads();
        // tile t: LDS buffer t & 1 holds it; register set (t + 1) & 1 holds tile t + 1; set t & 1 is free
        for (int t = 0; t < nt; t += 2) {
            if (t + 2 < nt) gload(t + 2, ra[0], rb[0]);
            compute(0);
            if (t + 1 < nt) sstore(1, ra[1], rb[1]);
            __syncthreads();
            if (t + 1 >= nt) break;
            if (t + 3 < nt) gload(t + 3, ra[1], rb[1]);
            compute(1);
            if (t + 2 < nt) sstore(0, ra[0], rb[0]);
            __syncthreads();
        }
    }

    // epilogue: float32 Y, bf16 transposed copy (4 consecutive rows of one column = 8 bytes), column statistics
    const float* rv = rowvec ? rowvec + (int64_t)bb * g.M : nullptr;
    float* Yb = Y + bb * g.sY;
    __bf16* YTb = YT ? YT + bb * g.sYT : nullptr;
    float sdot[2] = {0.f, 0.f}, ssq[2] = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int64_t col = n0 + wn0 + j * 32 + fr;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int64_t rbase = m0 + wm0 + i * 32 + 8 * q + 4 * fh;     // rows rbase .. rbase + 3 = regs 4q .. 4q + 3
                __bf16 pk[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int64_t row = rbase + e;
                    const float v = acc[i][j][4 * q + e];
                    pk[e] = (__bf16)v;
                    if (row < g.M && col < g.N) {
                        Yb[row * g.N + col] = v;
                        sdot[j] += rv ? v * rv[row] : 0.f;
                        ssq[j] += v * v;
                    }
                }
                if (YTb && col < g.N) {
                    if (rbase + 3 < g.M) {
                        *reinterpret_cast<uint2*>(YTb + col * g.M + rbase) = *reinterpret_cast<const uint2*>(pk);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (rbase + e < g.M) YTb[col * g.M + rbase + e] = pk[e];
                    }
                }
            }
        }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        sdot[j] += __shfl_xor(sdot[j], 32);
        ssq[j] += __shfl_xor(ssq[j], 32);
    }
    float* red = reinterpret_cast<float*>(smem);                    // [2 quantities][2 waves along m][BN]
    __syncthreads();
    if (lane < 32) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = wn0 + j * 32 + lane;
            red[(0 + (wave >> 1)) * BN + c] = sdot[j];
            red[(2 + (wave >> 1)) * BN + c] = ssq[j];
        }
    }
    __syncthreads();
    if (tid < BN && n0 + tid < g.N) {
        const int64_t o = ((int64_t)bb * g.tiles_m + bm) * g.N + n0 + tid;
        if (part_dot) part_dot[o] = red[tid] + red[BN + tid];
        part_sq[o] = red[2 * BN + tid] + red[3 * BN + tid];
    }
}

// dst (b, n, n) bf16 <- src (b, n, n) float32 / float64: plain, transposed, and / or lower-triangle-only (tril)
template <typename TS>
__global__ void cast_sq_bf16_kernel(const TS* __restrict__ src, __bf16* __restrict__ dst, int64_t n, int64_t batch,
                                    int transpose, int tril) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= batch * n * n) return;
    const int64_t b = idx / (n * n), e = idx % (n * n);
    const int64_t i = e / n, j = e % n;                             // destination (i, j)
    const int64_t si = transpose ? j : i, sj = transpose ? i : j;   // source element
    const TS v = (tril && sj > si) ? TS(0) : src[b * n * n + si * n + sj];
    dst[idx] = (__bf16)(float)v;
}

// Kxz[b][i][k] = bf16( os[b] exp(-1/2 sum_d ((x[i][d] - z[b][k][d]) / ls[b][d])^2) ): the TRANSPOSE of the Kzx the float32
// build writes, k contiguous -- the Qt operand of product 1.  One thread per (i, 8 consecutive k): 16-byte stores.
__global__ void rbf_build_t_bf16_kernel(const float* __restrict__ z, const float* __restrict__ x, const float* __restrict__ ls,
                                        const float* __restrict__ os, int64_t batch, int64_t M, int64_t n, int D,
                                        int64_t sxb, __bf16* __restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t kc = (M + 7) / 8;
    if (idx >= batch * n * kc) return;
    const int64_t b = idx / (n * kc), r = idx % (n * kc);
    const int64_t i = r / kc, k0 = (r % kc) * 8;
    const float* xb = x + b * sxb + i * D;
    const float* zb = z + b * M * D;
    float xi[8], il[8];
    for (int d = 0; d < D && d < 8; ++d) { xi[d] = xb[d]; il[d] = 1.0f / ls[b * D + d]; }
    const float osb = os[b];
    __bf16 v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int64_t k = k0 + e;
        float s = 0.f;
        if (k < M)
            for (int d = 0; d < D && d < 8; ++d) {
                const float t = (xi[d] - zb[k * D + d]) * il[d];
                s += t * t;
            }
        v[e] = (__bf16)(k < M ? osb * __expf(-0.5f * s) : 0.f);
    }
    __bf16* o = out + (b * n + i) * M + k0;
    if (k0 + 7 < M && (M % 8) == 0) {
        *reinterpret_cast<uint4*>(o) = *reinterpret_cast<const uint4*>(v);
    } else {
        for (int e = 0; e < 8; ++e)
            if (k0 + e < M) o[e] = v[e];
    }
}

// dst (b, n, M) bf16 <- transpose of src (b, M, n) float32: 32 x 32 tiles through LDS, coalesced on both sides
__global__ void transpose_cast_bf16_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, int64_t M, int64_t n) {
    __shared__ float tile[32][33];
    const int64_t b = blockIdx.z, m0 = (int64_t)blockIdx.y * 32, n0 = (int64_t)blockIdx.x * 32;
    const int tx = threadIdx.x, ty = threadIdx.y;
    const float* s = src + b * M * n;
    __bf16* d = dst + b * M * n;
#pragma unroll
    for (int i = 0; i < 32; i += 8) {
        const int64_t r = m0 + ty + i, c = n0 + tx;
        tile[ty + i][tx] = (r < M && c < n) ? s[r * n + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 32; i += 8) {
        const int64_t r = n0 + ty + i, c = m0 + tx;
        if (r < n && c < M) d[r * M + c] = (__bf16)tile[tx][ty + i];
    }
}

}  // namespace

extern "C" {

int nsgp_transpose_cast_bf16(const float* src, void* dst, int64_t batch, int64_t M, int64_t n, void* stream) {
    if (!src) return -1; if (!dst) return -2; if (batch < 0) return -3; if (M < 0) return -4; if (n < 0) return -5;
    if (batch == 0 || M == 0 || n == 0) return 0;
    if (batch > 65535 || cdiv64(M, 32) > 65535) return -24;
    hipLaunchKernelGGL(transpose_cast_bf16_kernel, dim3((unsigned)cdiv64(n, 32), (unsigned)cdiv64(M, 32), (unsigned)batch),
                       dim3(32, 8), 0, (hipStream_t)stream, src, (__bf16*)dst, M, n);
    return nsgp_launch_status();
}

size_t nsgp_svgp_bf16_tiles(int64_t M) { return M > 0 ? (size_t)cdiv64(M, BM) : 0; }

int nsgp_svgp_tri_gemm_colstats_bf16(const void* P, int tri, const void* Qt, const float* rowvec, int64_t batch, int64_t M,
                                     int64_t n, float* Y, void* YT, float* part_dot, float* part_sq, void* stream) {
    if (!P) return -1; if (tri != 1 && tri != 2) return -2; if (!Qt) return -3;
    if (batch < 0) return -5; if (M < 0) return -6; if (n < 0) return -7; if (!Y) return -8; if (!part_sq) return -11;
    if (batch == 0 || M == 0 || n == 0) return 0;
    if (M % 8 != 0) return -6;                              // 16-byte k chunks
    if (((uintptr_t)P | (uintptr_t)Qt) % 16 != 0) return -1;
    if (YT && (M % 4 != 0 || (uintptr_t)YT % 8 != 0)) return -9;
    Bf16Args g;
    g.M = M; g.N = n; g.sP = M * M; g.sQ = n * M; g.sY = M * n; g.sYT = n * M; g.tri = tri;
    g.tiles_m = (int)cdiv64(M, BM); g.tiles_n = (int)cdiv64(n, BN);
    if ((int64_t)g.tiles_m * g.tiles_n > 2147483647LL || batch > 65535) return -24;
    constexpr size_t lds = 2 * (BM + BN) * LDT * sizeof(__bf16);
    nsgp_opt_in_lds((const void*)bf16_proj_kernel, lds);
    hipLaunchKernelGGL(bf16_proj_kernel, dim3((unsigned)(g.tiles_m * g.tiles_n), (unsigned)batch), dim3(256), lds,
                       (hipStream_t)stream, g, (const __bf16*)P, (const __bf16*)Qt, rowvec, Y, (__bf16*)YT, part_dot, part_sq);
    return nsgp_launch_status();
}

int nsgp_cast_sq_bf16_f32(const float* src, void* dst, int64_t n, int64_t batch, int transpose, int tril, void* stream) {
    if (!src) return -1; if (!dst) return -2; if (n < 0) return -3; if (batch < 0) return -4;
    if (n == 0 || batch == 0) return 0;
    const int64_t tot = batch * n * n;
    hipLaunchKernelGGL((cast_sq_bf16_kernel<float>), dim3((unsigned)cdiv64(tot, 256)), dim3(256), 0, (hipStream_t)stream,
                       src, (__bf16*)dst, n, batch, transpose, tril);
    return nsgp_launch_status();
}
int nsgp_cast_sq_bf16_f64(const double* src, void* dst, int64_t n, int64_t batch, int transpose, int tril, void* stream) {
    if (!src) return -1; if (!dst) return -2; if (n < 0) return -3; if (batch < 0) return -4;
    if (n == 0 || batch == 0) return 0;
    const int64_t tot = batch * n * n;
    hipLaunchKernelGGL((cast_sq_bf16_kernel<double>), dim3((unsigned)cdiv64(tot, 256)), dim3(256), 0, (hipStream_t)stream,
                       src, (__bf16*)dst, n, batch, transpose, tril);
    return nsgp_launch_status();
}

int nsgp_rbf_build_t_bf16(const float* z, const float* x, const float* ls, const float* os, int64_t batch, int64_t M,
                          int64_t n, int64_t D, int64_t sxb, void* out, void* stream) {
    if (!z) return -1; if (!x) return -2; if (!ls) return -3; if (!os) return -4;
    if (batch < 0) return -5; if (M < 0) return -6; if (n < 0) return -7; if (D < 1 || D > 8) return -8; if (!out) return -10;
    if (batch == 0 || M == 0 || n == 0) return 0;
    const int64_t tot = batch * n * cdiv64(M, 8);
    hipLaunchKernelGGL(rbf_build_t_bf16_kernel, dim3((unsigned)cdiv64(tot, 256)), dim3(256), 0, (hipStream_t)stream, z, x, ls,
                       os, batch, M, n, (int)D, sxb, (__bf16*)out);
    return nsgp_launch_status();
}

}  // extern "C"
