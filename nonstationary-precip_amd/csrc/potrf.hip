// potrf.hip -- K4 blocked right-looking Cholesky (lower) and K5 triangular inverse, batched.
//
// Replaces psd_safe_cholesky (models/gibbs_kernels.py:201,298), gpytorch
// VariationalStrategy._cholesky_factor / L.inv_matmul, the Cholesky under
// ExactMarginalLogLikelihood / MultivariateNormal.log_prob and torch.triangular_solve(eye, chol)
// (models/gibbs_kernels.py:203,300).
//
// potrf, per 64-column panel j:
//   panel kernel   every workgroup re-factors the 64x64 diagonal block in LDS (5 us, saves a launch
//                  and a grid-wide dependency), inverts it in LDS and applies it to its own 64-row slab of
//                  the panel:  L21 = A21 * L11^-T.  The factor goes to a side buffer (other
//                  workgroups still read A11); one extra workgroup copies the PREVIOUS panel's
//                  factor into place and zeroes the strict upper triangle of those rows.
//   trailing       A22 -= L21 L21^T (lower tiles only) on the MFMA GEMM of gemm.hip.
// trtri: invert the 64x64 diagonal blocks in LDS, then merge pairs of blocks bottom-up,
//   X21 = -B^-1 (C A^-1), every level two batched MFMA GEMMs (log2(n/64) levels).
#include "common.h"

// from gemm.hip
extern "C" int nsgp_gemm_f32(int64_t, int64_t, int64_t, float, const float*, int64_t, int64_t, int64_t, int64_t,
                             const float*, int64_t, int64_t, int64_t, int64_t, float, float*, int64_t, int64_t,
                             int64_t, int64_t, int64_t, int, void*, size_t, void*);
extern "C" int nsgp_gemm_f64(int64_t, int64_t, int64_t, double, const double*, int64_t, int64_t, int64_t, int64_t,
                             const double*, int64_t, int64_t, int64_t, int64_t, double, double*, int64_t, int64_t,
                             int64_t, int64_t, int64_t, int, void*, size_t, void*);

namespace {

constexpr int NB = 64;          // panel width
constexpr int LDD = NB + 1;     // LDS leading dimension (odd: conflict-free column walks)

template <typename T> int gemm_t(int64_t M, int64_t N, int64_t K, T alpha, const T* A, int64_t sam, int64_t sak,
                                 int64_t sa1, int64_t sa2, const T* B, int64_t sbk, int64_t sbn, int64_t sb1,
                                 int64_t sb2, T beta, T* C, int64_t ldc, int64_t sc1, int64_t sc2, int64_t nb1,
                                 int64_t nb2, int flags, void* stream);
template <> int gemm_t<float>(int64_t M, int64_t N, int64_t K, float alpha, const float* A, int64_t sam, int64_t sak,
                              int64_t sa1, int64_t sa2, const float* B, int64_t sbk, int64_t sbn, int64_t sb1,
                              int64_t sb2, float beta, float* C, int64_t ldc, int64_t sc1, int64_t sc2, int64_t nb1,
                              int64_t nb2, int flags, void* stream) {
    return nsgp_gemm_f32(M, N, K, alpha, A, sam, sak, sa1, sa2, B, sbk, sbn, sb1, sb2, beta, C, ldc, sc1, sc2, nb1,
                         nb2, flags | NSGP_GEMM_NO_SPLITK, nullptr, 0, stream);
}
template <> int gemm_t<double>(int64_t M, int64_t N, int64_t K, double alpha, const double* A, int64_t sam,
                               int64_t sak, int64_t sa1, int64_t sa2, const double* B, int64_t sbk, int64_t sbn,
                               int64_t sb1, int64_t sb2, double beta, double* C, int64_t ldc, int64_t sc1,
                               int64_t sc2, int64_t nb1, int64_t nb2, int flags, void* stream) {
    return nsgp_gemm_f64(M, N, K, alpha, A, sam, sak, sa1, sa2, B, sbk, sbn, sb1, sb2, beta, C, ldc, sc1, sc2, nb1,
                         nb2, flags | NSGP_GEMM_NO_SPLITK, nullptr, 0, stream);
}

// Factor the nb x nb lower block held in D (LDS, leading dim LDD) in place.  All 256 threads.
// Returns (to every thread) 0 or the 1-based index of the first non-positive pivot.
template <typename T> __device__ int factor_block(T* D, int nb, int* s_info) {
    const int tid = threadIdx.x;
    if (tid == 0) *s_info = 0;
    for (int k = 0; k < nb; ++k) {
        __syncthreads();
        const T dkk = D[k * LDD + k];
        if (tid == 0 && !(dkk > T(0)) && *s_info == 0) *s_info = k + 1;
        const T piv = t_sqrt(dkk);
        const T inv = T(1) / piv;
        __syncthreads();
        if (tid == k) D[k * LDD + k] = piv;
        else if (tid > k && tid < nb) D[tid * LDD + k] *= inv;
        __syncthreads();
        const int rem = nb - k - 1;
        for (int e = tid; e < rem * rem; e += 256) {
            const int i = k + 1 + e / rem, j = k + 1 + e % rem;
            if (j <= i) D[i * LDD + j] -= D[i * LDD + k] * D[j * LDD + k];
        }
    }
    __syncthreads();
    return *s_info;
}

// X = D^-1 for the nb x nb lower-triangular block D (LDS) into Li (LDS); strict upper of Li zeroed.
template <typename T> __device__ void invert_block(const T* D, T* Li, int nb) {
    const int c = threadIdx.x;
    if (c < nb) {
        for (int i = 0; i < c; ++i) Li[i * LDD + c] = T(0);
        Li[c * LDD + c] = T(1) / D[c * LDD + c];
        for (int i = c + 1; i < nb; ++i) {
            T s = T(0);
            for (int k = c; k < i; ++k) s += D[i * LDD + k] * Li[k * LDD + c];
            Li[i * LDD + c] = -s / D[i * LDD + i];
        }
    }
    __syncthreads();
}

// grid.x = nslab + 1: blocks [0, nslab) each own a 64-row slab of the panel below the diagonal block
// (block 0 also publishes the factor), the last block writes panel j-1's factor back into A.
template <typename T>
__global__ __launch_bounds__(256) void potrf_panel_kernel(T* __restrict__ A, int64_t n, int64_t lda, int64_t sA,
                                                          int64_t j0, T* __restrict__ wsL, int64_t npanels,
                                                          int32_t* __restrict__ info) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* D = reinterpret_cast<T*>(smem_raw);
    T* Li = D + NB * LDD;
    T* S = Li + NB * LDD;
    int* s_info = reinterpret_cast<int*>(S + NB * LDD);      // all LDS in the one dynamic region
    const int tid = threadIdx.x;
    const int64_t b = blockIdx.y;
    T* Ab = A + b * sA;
    const int64_t pj = j0 / NB;
    const int nslab = (int)gridDim.x - 1;

    if ((int)blockIdx.x == nslab) {
        // write-back of the previous panel (nobody reads its A11 any more)
        const int64_t p = pj - 1;
        if (p < 0) return;
        const int64_t r0 = p * NB;
        const int pnb = (int)((n - r0) < NB ? (n - r0) : NB);
        const T* src = wsL + (b * npanels + p) * NB * NB;
        for (int e = tid; e < pnb * pnb; e += 256) {
            const int i = e / pnb, j = e % pnb;
            if (j <= i) Ab[(r0 + i) * lda + r0 + j] = src[i * NB + j];
        }
        for (int i = 0; i < pnb; ++i)
            for (int64_t c = r0 + i + 1 + tid; c < n; c += 256) Ab[(r0 + i) * lda + c] = T(0);
        return;
    }

    const int nb = (int)((n - j0) < NB ? (n - j0) : NB);
    for (int e = tid; e < nb * nb; e += 256) {
        const int i = e / nb, j = e % nb;
        D[i * LDD + j] = (j <= i) ? Ab[(j0 + i) * lda + j0 + j] : T(0);
    }
    const int bad = factor_block(D, nb, s_info);
    if (blockIdx.x == 0) {
        T* dst = wsL + (b * npanels + pj) * NB * NB;
        for (int e = tid; e < nb * nb; e += 256) dst[(e / nb) * NB + e % nb] = D[(e / nb) * LDD + e % nb];
        if (tid == 0 && bad && info[b] == 0) info[b] = (int32_t)(j0 + bad);
    }
    const int64_t r0 = j0 + nb + (int64_t)blockIdx.x * NB;
    if (r0 >= n) return;                                   // block 0 of the last panel has no slab
    invert_block(D, Li, nb);
    const int rows = (int)((n - r0) < NB ? (n - r0) : NB);
    for (int e = tid; e < rows * nb; e += 256) {
        const int i = e / nb, j = e % nb;
        S[i * LDD + j] = Ab[(r0 + i) * lda + j0 + j];
    }
    __syncthreads();
    // X[r][c] = sum_{k<=c} S[r][k] * Li[c][k];  thread -> 4 rows x 4 cols
    const int tr = (tid >> 4) * 4, tc = (tid & 15) * 4;
    T acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = T(0);
    const int kmax = tc + 4 < nb ? tc + 4 : nb;
    for (int k = 0; k < kmax; ++k) {
        T sv[4], lv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) sv[i] = S[(tr + i) * LDD + k];
#pragma unroll
        for (int j = 0; j < 4; ++j) lv[j] = Li[(tc + j) * LDD + k];      // zero for k > c
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] += sv[i] * lv[j];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (tr + i < rows && tc + j < nb) Ab[(r0 + tr + i) * lda + j0 + tc + j] = acc[i][j];
}

template <typename T> size_t panel_lds_bytes() { return 3 * (size_t)NB * LDD * sizeof(T) + 16; }
template <typename T> size_t diag_lds_bytes() { return 2 * (size_t)NB * LDD * sizeof(T); }

template <typename T>
int potrf_impl(T* A, int64_t n, int64_t lda, int64_t sA, int64_t batch, int32_t* info, void* ws, size_t wsb,
               void* stream) {
    if (n < 0) return -2; if (lda < n) return -3; if (batch < 0) return -5;
    if (n == 0 || batch == 0) return 0;
    if (!A) return -1; if (!info) return -6;
    const int64_t npanels = cdiv64(n, NB);
    const size_t need = (size_t)batch * npanels * NB * NB * sizeof(T);
    if (!ws || wsb < need) return -7;
    if (batch > 65535) return -5;
    T* wsL = (T*)ws;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(info, 0, sizeof(int32_t) * batch, st);
    if (e != hipSuccess) return (int)e;
    static bool attr_set = false;       // idempotent attribute, set once per process
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)potrf_panel_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)panel_lds_bytes<T>());
        attr_set = true;
    }
    for (int64_t j0 = 0; j0 < n; j0 += NB) {
        const int64_t nb = (n - j0) < NB ? (n - j0) : NB;
        const int64_t below = n - j0 - nb;
        const int64_t nslab = below > 0 ? cdiv64(below, NB) : 1;
        hipLaunchKernelGGL((potrf_panel_kernel<T>), dim3((unsigned)(nslab + 1), (unsigned)batch), dim3(256),
                           panel_lds_bytes<T>(), st, A, n, lda, sA, j0, wsL, npanels, info);
        if (below > 0) {
            T* L21 = A + (j0 + nb) * lda + j0;
            T* A22 = A + (j0 + nb) * lda + (j0 + nb);
            int rc = gemm_t<T>(below, below, nb, T(-1), L21, lda, 1, sA, 0, L21, 1, lda, sA, 0, T(1), A22, lda, sA, 0,
                               batch, 1, NSGP_GEMM_C_LOWER, stream);
            if (rc) return rc;
        }
    }
    // final write-back of the last panel: launch with j0 = npanels*NB so that "previous" is the last
    hipLaunchKernelGGL((potrf_panel_kernel<T>), dim3(1, (unsigned)batch), dim3(256), panel_lds_bytes<T>(), st, A, n,
                       lda, sA, npanels * NB, wsL, npanels, info);
    return nsgp_launch_status();
}

// ---- trtri ---------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void trtri_diag_kernel(const T* __restrict__ L, int64_t n, int64_t ldl, int64_t sL,
                                                         T* __restrict__ X, int64_t ldx, int64_t sX) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* D = reinterpret_cast<T*>(smem_raw);
    T* Li = D + NB * LDD;
    const int tid = threadIdx.x;
    const int64_t b = blockIdx.y, r0 = (int64_t)blockIdx.x * NB;
    const T* Lb = L + b * sL;
    T* Xb = X + b * sX;
    const int nb = (int)((n - r0) < NB ? (n - r0) : NB);
    for (int e = tid; e < nb * nb; e += 256) {
        const int i = e / nb, j = e % nb;
        D[i * LDD + j] = (j <= i) ? Lb[(r0 + i) * ldl + r0 + j] : T(0);
    }
    __syncthreads();
    invert_block(D, Li, nb);
    for (int e = tid; e < nb * nb; e += 256) {
        const int i = e / nb, j = e % nb;
        Xb[(r0 + i) * ldx + r0 + j] = Li[i * LDD + j];
    }
    // zero everything right of the diagonal block in these rows
    for (int i = 0; i < nb; ++i)
        for (int64_t c = r0 + nb + tid; c < n; c += 256) Xb[(r0 + i) * ldx + c] = T(0);
}

template <typename T>
int trtri_impl(const T* L, int64_t n, int64_t ldl, int64_t sL, T* X, int64_t ldx, int64_t sX, int64_t batch,
               void* ws, size_t wsb, void* stream) {
    if (n < 0) return -2; if (ldl < n) return -3; if (ldx < n) return -6; if (batch < 0) return -8;
    if (n == 0 || batch == 0) return 0;
    if (!L) return -1; if (!X) return -5;
    const size_t need = (size_t)batch * n * n * sizeof(T);
    if (n > NB && (!ws || wsb < need)) return -9;
    if (batch > 65535) return -8;
    hipStream_t st = (hipStream_t)stream;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)trtri_diag_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)diag_lds_bytes<T>());
        attr_set = true;
    }
    hipLaunchKernelGGL((trtri_diag_kernel<T>), dim3((unsigned)cdiv64(n, NB), (unsigned)batch), dim3(256),
                       diag_lds_bytes<T>(), st, L, n, ldl, sL, X, ldx, sX);
    T* Tm = (T*)ws;                      // (batch, n, n) scratch, same indexing as X with ld = n
    for (int64_t s = NB; s < n; s *= 2) {
        // pairs (A = X[i0:i0+s, i0:i0+s], B = X[i0+s:i0+s+h, ...], C = L[i0+s:i0+s+h, i0:i0+s])
        const int64_t npairs_full = n / (2 * s);                  // pairs whose lower block is full (h == s)
        const int64_t rem = n - npairs_full * 2 * s;              // trailing rows
        if (npairs_full > 0) {
            // T = C * A^-1        (s x s), A^-1 lower  -> B-operand lower
            int rc = gemm_t<T>(s, s, s, T(1), L + s * ldl, ldl, 1, sL, 2 * s * (ldl + 1),
                               X, ldx, 1, sX, 2 * s * (ldx + 1), T(0), Tm + s * n, n, n * n, 2 * s * (n + 1), batch,
                               npairs_full, NSGP_GEMM_B_LOWER, stream);
            if (rc) return rc;
            // X21 = -B^-1 * T     B^-1 lower -> A-operand lower
            rc = gemm_t<T>(s, s, s, T(-1), X + s * (ldx + 1), ldx, 1, sX, 2 * s * (ldx + 1),
                           Tm + s * n, n, 1, n * n, 2 * s * (n + 1), T(0), X + s * ldx, ldx, sX, 2 * s * (ldx + 1),
                           batch, npairs_full, NSGP_GEMM_A_LOWER, stream);
            if (rc) return rc;
        }
        if (rem > s) {
            const int64_t i0 = npairs_full * 2 * s, h = rem - s;
            int rc = gemm_t<T>(h, s, s, T(1), L + (i0 + s) * ldl + i0, ldl, 1, sL, 0,
                               X + i0 * (ldx + 1), ldx, 1, sX, 0, T(0), Tm + (i0 + s) * n + i0, n, n * n, 0, batch, 1,
                               NSGP_GEMM_B_LOWER, stream);
            if (rc) return rc;
            rc = gemm_t<T>(h, s, h, T(-1), X + (i0 + s) * (ldx + 1), ldx, 1, sX, 0,
                           Tm + (i0 + s) * n + i0, n, 1, n * n, 0, T(0), X + (i0 + s) * ldx + i0, ldx, sX, 0, batch, 1,
                           NSGP_GEMM_A_LOWER, stream);
            if (rc) return rc;
        }
    }
    return nsgp_launch_status();
}

}  // namespace

extern "C" {

size_t nsgp_potrf_workspace(int64_t n, int64_t batch, int elem_size) {
    if (n <= 0 || batch <= 0) return 0;
    return (size_t)batch * cdiv64(n, NB) * NB * NB * elem_size;
}
int nsgp_potrf_f32(float* A, int64_t n, int64_t lda, int64_t sA, int64_t batch, int32_t* info, void* ws,
                   size_t wsb, void* stream) {
    return potrf_impl<float>(A, n, lda, sA, batch, info, ws, wsb, stream);
}
int nsgp_potrf_f64(double* A, int64_t n, int64_t lda, int64_t sA, int64_t batch, int32_t* info, void* ws,
                   size_t wsb, void* stream) {
    return potrf_impl<double>(A, n, lda, sA, batch, info, ws, wsb, stream);
}
size_t nsgp_trtri_workspace(int64_t n, int64_t batch, int elem_size) {
    if (n <= NB || batch <= 0) return 0;
    return (size_t)batch * n * n * elem_size;
}
int nsgp_trtri_f32(const float* L, int64_t n, int64_t ldl, int64_t sL, float* X, int64_t ldx, int64_t sX,
                   int64_t batch, void* ws, size_t wsb, void* stream) {
    return trtri_impl<float>(L, n, ldl, sL, X, ldx, sX, batch, ws, wsb, stream);
}
int nsgp_trtri_f64(const double* L, int64_t n, int64_t ldl, int64_t sL, double* X, int64_t ldx, int64_t sX,
                   int64_t batch, void* ws, size_t wsb, void* stream) {
    return trtri_impl<double>(L, n, ldl, sL, X, ldx, sX, batch, ws, wsb, stream);
}

}  // extern "C"
