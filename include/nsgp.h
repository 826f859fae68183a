/* nsgp.h -- C ABI of libnsgp_hip.so: the MI355X (gfx950) device kernels behind the
 * non-stationary GP / DSVI deep-GP hot path of Stansfash/nonstationary-precip.
 *
 * The reference is pure Python over gpytorch/torch and has no FFI of its own; each entry point
 * below names the reference arithmetic it replaces (paths relative to /root/reference) and is
 * what a maintainer of the reference would bind with ctypes (INTEGRATION.md shows the stub).
 *
 * Conventions
 *  - every pointer is a DEVICE pointer (hipMalloc / torch caching allocator) unless it says host;
 *  - matrices are row-major with an explicit leading dimension (elements, not bytes);
 *  - `stream` is a hipStream_t passed as void*; all functions are asynchronous and stream-ordered,
 *    never allocate, never synchronise; the caller owns every buffer, including workspaces whose
 *    size the matching *_workspace() function returns (bytes);
 *  - return 0 on success, -k when argument k (1-based) is invalid, >0 = hipError_t of the launch;
 *  - functions are re-entrant (no global mutable state);
 *  - suffix _f32 / _f64 is the arithmetic type the kernel computes in.
 */
#ifndef NSGP_H
#define NSGP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NSGP_MAX_DIM 8          /* input dimension D supported by the pairwise kernels */

/* library / device identification (host) */
int nsgp_abi_version(void);                 /* bumps when a signature changes */
const char* nsgp_build_arch(void);          /* "gfx950" */

/* Measurement aid (bench.py's roofline context; no reference counterpart): the matrix-core rate THIS chip sustains.  A
 * grid of `workgroups` x 4 waves issues iters x 8 independent register-only MFMAs per wave -- kind 0: v_mfma_f32_32x32x2_f32,
 * 1: v_mfma_f64_16x16x4_f64, 2: v_mfma_i32_32x32x32_i8 -- and every wave writes (shader-clock cycles, 100 MHz wall ticks) it
 * spent to out[2 (4 wg + wave) + {0, 1}].  cycles / ticks x 100 MHz = the clock held under full matrix-core load (MI355X:
 * ~2.1 GHz against the 2.4 GHz of the data-sheet peaks); operations / launch time = the rate no GEMM on this chip exceeds.
 * out: device, 8 x workgroups uint64; sink: device float (never written); iters: a positive multiple of 16. */
int nsgp_mfma_rate_probe(int kind, int64_t workgroups, int iters, uint64_t* out, float* sink, void* stream);

/* ------------------------------------------------------------------------------------------
 * K1  Diagonal Gibbs kernel (Rasmussen & Williams eq. 4.32)
 *     K[i,j] = os * prod_d sqrt(2 l1[d,i] l2[d,j] / (l1[d,i]^2 + l2[d,j]^2))
 *                 * exp(-sum_d (x1[i,d]-x2[j,d])^2 / (l1[d,i]^2 + l2[d,j]^2))  (+ diag_add on i==j)
 *     replaces models/gibbs_kernels.py:154-162 (GibbsKernel.forward; 8 full-size temporaries)
 *     and the ScaleKernel multiply of GibbsSafeScaleKernel (:164-168).
 *     x1:(n1,D) x2:(n2,D) row-major contiguous; ell1:(D,n1) ell2:(D,n2) dim-major contiguous.
 *     outputscale, diag_add: device scalars (1 element; NULL = 1 and 0) so trainable
 *     hyper-parameters (raw_outputscale, likelihood noise) never round-trip through the host.
 * ------------------------------------------------------------------------------------------ */
int nsgp_gibbs_build_fwd_f32(const float* x1, const float* x2, const float* ell1, const float* ell2,
                             int64_t n1, int64_t n2, int D, const float* outputscale, const float* diag_add,
                             float* K, int64_t ldk, void* stream);
int nsgp_gibbs_build_fwd_f64(const double* x1, const double* x2, const double* ell1, const double* ell2,
                             int64_t n1, int64_t n2, int D, const double* outputscale, const double* diag_add,
                             double* K, int64_t ldk, void* stream);
/* backward: G = dLoss/dK (n1,n2).  Outputs (any may be NULL): g_ell1:(D,n1) g_ell2:(D,n2)
 * g_x1:(n1,D) g_x2:(n2,D) g_os:(1).  Deterministic two-pass reduction (no atomics). */
size_t nsgp_gibbs_build_bwd_workspace(int64_t n1, int64_t n2, int D, int elem_size);
int nsgp_gibbs_build_bwd_f32(const float* x1, const float* x2, const float* ell1, const float* ell2,
                             int64_t n1, int64_t n2, int D, const float* outputscale,
                             const float* G, int64_t ldg,
                             float* g_ell1, float* g_ell2, float* g_x1, float* g_x2, float* g_os,
                             void* ws, size_t ws_bytes, void* stream);
int nsgp_gibbs_build_bwd_f64(const double* x1, const double* x2, const double* ell1, const double* ell2,
                             int64_t n1, int64_t n2, int D, const double* outputscale,
                             const double* G, int64_t ldg,
                             double* g_ell1, double* g_ell2, double* g_x1, double* g_x2, double* g_os,
                             void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * K2  Batched RBF-ARD kernel  K[b,i,j] = os[b] * exp(-0.5 sum_d ((x1[b,i,d]-x2[b,j,d])/ls[b,d])^2)
 *     replaces gpytorch ScaleKernel(RBFKernel(ard)) as built at models/dgps.py:44-46 and
 *     models/gibbs_kernels.py:67-69.  x1:(batch,n1,D) with batch stride sx1 (0 = shared),
 *     ls:(batch,D), os:(batch) device arrays (parameters stay on the device).
 * ------------------------------------------------------------------------------------------ */
int nsgp_rbf_build_fwd_f32(const float* x1, const float* x2, const float* ls, const float* os,
                           int64_t batch, int64_t n1, int64_t n2, int D, int64_t sx1, int64_t sx2,
                           float diag_add, float* K, int64_t ldk, int64_t sK, void* stream);
int nsgp_rbf_build_fwd_f64(const double* x1, const double* x2, const double* ls, const double* os,
                           int64_t batch, int64_t n1, int64_t n2, int D, int64_t sx1, int64_t sx2,
                           double diag_add, double* K, int64_t ldk, int64_t sK, void* stream);
/* backward: outputs (any may be NULL) g_x1:(batch,n1,D) g_x2:(batch,n2,D) g_ls:(batch,D) g_os:(batch).
 * g_x1 == g_x2 (one buffer, n1 == n2): the row- and column-side gradients are SUMMED into it -- the x1 = x2 (Kzz) case. */
size_t nsgp_rbf_build_bwd_workspace(int64_t batch, int64_t n1, int64_t n2, int D, int elem_size);
int nsgp_rbf_build_bwd_f32(const float* x1, const float* x2, const float* ls, const float* os,
                           int64_t batch, int64_t n1, int64_t n2, int D, int64_t sx1, int64_t sx2,
                           const float* G, int64_t ldg, int64_t sG,
                           float* g_x1, float* g_x2, float* g_ls, float* g_os,
                           void* ws, size_t ws_bytes, void* stream);
int nsgp_rbf_build_bwd_f64(const double* x1, const double* x2, const double* ls, const double* os,
                           int64_t batch, int64_t n1, int64_t n2, int D, int64_t sx1, int64_t sx2,
                           const double* G, int64_t ldg, int64_t sG,
                           double* g_x1, double* g_x2, double* g_ls, double* g_os,
                           void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * K3  Paciorek-Schervish kernel, D = 2, closed-form 2x2 determinants / inverse
 *     K[i,j] = |S1_i|^(1/4) |S2_j|^(1/4) |(S1_i+S2_j)/2|^(-1/2)
 *              * exp(-d^T ((S1_i+S2_j)/2 + jitter I)^-1 d),   d = x1_i - x2_j
 *     replaces models/multivariate_gibbs_kernel.py:98-150 and
 *     models/sparse_multivariate_gibbs_kernel.py:103-154 ((N,N,2,2) temporaries, det/inverse).
 *     sig1:(n1,4) sig2:(n2,4) row-major 2x2 per point.
 * ------------------------------------------------------------------------------------------ */
int nsgp_ps2d_build_fwd_f32(const float* x1, const float* x2, const float* sig1, const float* sig2,
                            int64_t n1, int64_t n2, float jitter, float* K, int64_t ldk, void* stream);
int nsgp_ps2d_build_fwd_f64(const double* x1, const double* x2, const double* sig1, const double* sig2,
                            int64_t n1, int64_t n2, double jitter, double* K, int64_t ldk, void* stream);
/* backward wrt the per-point matrices: g_sig1:(n1,4) g_sig2:(n2,4) (either may be NULL) */
size_t nsgp_ps2d_build_bwd_workspace(int64_t n1, int64_t n2, int elem_size);
int nsgp_ps2d_build_bwd_f32(const float* x1, const float* x2, const float* sig1, const float* sig2,
                            int64_t n1, int64_t n2, float jitter, const float* G, int64_t ldg,
                            float* g_sig1, float* g_sig2, void* ws, size_t ws_bytes, void* stream);
int nsgp_ps2d_build_bwd_f64(const double* x1, const double* x2, const double* sig1, const double* sig2,
                            int64_t n1, int64_t n2, double jitter, const double* G, int64_t ldg,
                            double* g_sig1, double* g_sig2, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Dense contraction on the matrix cores (v_mfma_f32_32x32x2_f32 / v_mfma_f64_16x16x4_f64)
 *     C[b] = alpha * op(A[b]) * op(B[b]) + beta * C[b]        (M x N, inner K)
 *     replaces the torch matmul / triangular_solve / inverse chains of
 *     models/gibbs_kernels.py:203,225 (K_xz Kzz^-1/2), models/nonstationary_models.py:55-58,142-150
 *     and gpytorch VariationalStrategy's  L^-1 Kzx,  (S - I) A  (SURVEY A.3).
 *     A element (m,k) at A[m*sam + k*sak], B element (k,n) at B[k*sbk + n*sbn] (one stride of each
 *     pair must be 1), C row-major with ldc.  Two-level batch: b = b1*nb2 + b2, offsets
 *     b1*s?1 + b2*s?2 (elements).
 *     flags: triangular structure that lets whole K-tiles / output tiles be skipped and masks the
 *     other triangle of the operand (so an un-zeroed triangle is never read as data).
 * ------------------------------------------------------------------------------------------ */
#define NSGP_GEMM_A_LOWER   1   /* op(A)(m,k) == 0 for k > m  */
#define NSGP_GEMM_A_UPPER   2   /* op(A)(m,k) == 0 for k < m  */
#define NSGP_GEMM_B_LOWER   4   /* op(B)(k,n) == 0 for n > k  */
#define NSGP_GEMM_B_UPPER   8   /* op(B)(k,n) == 0 for n < k  */
#define NSGP_GEMM_C_LOWER  16   /* only the lower triangle (n <= m) of C is computed/stored; the
                                   strict upper triangle is written as zero when beta == 0 */
#define NSGP_GEMM_NO_SPLITK 32  /* never split the inner dimension (no workspace needed) */
#define NSGP_GEMM_C_HALFDIAG 128 /* the diagonal of alpha*op(A)*op(B) is halved before beta*C is added (Phi of the
                                   Cholesky backward: tril with halved diagonal, without a pass of its own) */
#define NSGP_GEMM_C_NOFILL  64  /* with C_LOWER: leave the strict upper triangle of C untouched (the consumer reads C
                                   through a LOWER operand flag, which never loads it) -- no memset nodes */
size_t nsgp_gemm_workspace(int64_t M, int64_t N, int64_t K, int64_t nb1, int64_t nb2, int elem_size, int flags);
int nsgp_gemm_f32(int64_t M, int64_t N, int64_t K, float alpha,
                  const float* A, int64_t sam, int64_t sak, int64_t sa1, int64_t sa2,
                  const float* B, int64_t sbk, int64_t sbn, int64_t sb1, int64_t sb2,
                  float beta, float* C, int64_t ldc, int64_t sc1, int64_t sc2,
                  int64_t nb1, int64_t nb2, int flags, void* ws, size_t ws_bytes, void* stream);
int nsgp_gemm_f64(int64_t M, int64_t N, int64_t K, double alpha,
                  const double* A, int64_t sam, int64_t sak, int64_t sa1, int64_t sa2,
                  const double* B, int64_t sbk, int64_t sbn, int64_t sb1, int64_t sb2,
                  double beta, double* C, int64_t ldc, int64_t sc1, int64_t sc2,
                  int64_t nb1, int64_t nb2, int flags, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * K2' Batched  os * RBF-ARD(x; ls_rbf) * Periodic(x; ls_per, period)  in one launch:
 *       k = os_b exp(-1/2 sum_d ((x_d - x'_d)/ls_rbf[b,d])^2) exp(-2 sin^2(pi |x - x'| / period_b) / ls_per_b)
 *     replaces ScaleKernel(RBFKernel(active_dims=0) * PeriodicKernel(active_dims=0), GreaterThan(7))
 *     (models/spatio_temporal_models.py:22,42; experiments/temporal_exp.py:39) -- gpytorch < 1.9
 *     PeriodicKernel semantics as recalled in SURVEY A.2/A.7 (distance of x/period, division by ls_per).
 *     ls_rbf == NULL drops the RBF factor (plain PeriodicKernel); os == NULL means 1.  x as in K2.
 *     backward: g_x1/g_x2 (batch,n,D) per batch (NULL to skip), g_ls_rbf (batch,D), g_ls_per, g_period,
 *     g_os (batch); workspace from nsgp_rbf_periodic_build_bwd_workspace.
 * ------------------------------------------------------------------------------------------ */
int nsgp_rbf_periodic_build_fwd_f32(const float* x1, const float* x2, const float* ls_rbf, const float* ls_per,
                                    const float* period, const float* os, int64_t batch, int64_t n1, int64_t n2, int D,
                                    int64_t sx1, int64_t sx2, float diag_add, float* K, int64_t ldk, int64_t sK,
                                    void* stream);
int nsgp_rbf_periodic_build_fwd_f64(const double* x1, const double* x2, const double* ls_rbf, const double* ls_per,
                                    const double* period, const double* os, int64_t batch, int64_t n1, int64_t n2,
                                    int D, int64_t sx1, int64_t sx2, double diag_add, double* K, int64_t ldk,
                                    int64_t sK, void* stream);
size_t nsgp_rbf_periodic_build_bwd_workspace(int64_t batch, int64_t n1, int64_t n2, int D, int elem_size);
int nsgp_rbf_periodic_build_bwd_f32(const float* x1, const float* x2, const float* ls_rbf, const float* ls_per,
                                    const float* period, const float* os, int64_t batch, int64_t n1, int64_t n2, int D,
                                    int64_t sx1, int64_t sx2, const float* G, int64_t ldg, int64_t sG, float* g_x1,
                                    float* g_x2, float* g_ls_rbf, float* g_ls_per, float* g_period, float* g_os,
                                    void* ws, size_t ws_bytes, void* stream);
int nsgp_rbf_periodic_build_bwd_f64(const double* x1, const double* x2, const double* ls_rbf, const double* ls_per,
                                    const double* period, const double* os, int64_t batch, int64_t n1, int64_t n2,
                                    int D, int64_t sx1, int64_t sx2, const double* G, int64_t ldg, int64_t sG,
                                    double* g_x1, double* g_x2, double* g_ls_rbf, double* g_ls_per, double* g_period,
                                    double* g_os, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * K4  Blocked right-looking Cholesky A = L L^T (lower, in place, strict upper zeroed), batched.
 *     Panel: LDS-resident 64x64 diagonal factor + explicit inverse; trailing update on MFMA.
 *     replaces psd_safe_cholesky (models/gibbs_kernels.py:201,298), gpytorch
 *     VariationalStrategy._cholesky_factor and the Cholesky inside ExactMarginalLogLikelihood /
 *     MultivariateNormal.log_prob, and torch.linalg.solve at utils/functional.py:33.
 *     info:(batch) int32 device: 0 ok, k>0 = leading minor k not positive definite (LAPACK).
 * K5  trtri: X = L^-1 (lower, out of place, strict upper zeroed), recursive on MFMA GEMMs;
 *     replaces torch.triangular_solve(eye, chol) (models/gibbs_kernels.py:203,300) and
 *     L.inv_matmul(Kzx) of gpytorch (as the dense product L^-1 * Kzx).
 * ------------------------------------------------------------------------------------------ */
size_t nsgp_potrf_workspace(int64_t n, int64_t batch, int elem_size);
int nsgp_potrf_f32(float* A, int64_t n, int64_t lda, int64_t sA, int64_t batch, int32_t* info,
                   void* ws, size_t ws_bytes, void* stream);
int nsgp_potrf_f64(double* A, int64_t n, int64_t lda, int64_t sA, int64_t batch, int32_t* info,
                   void* ws, size_t ws_bytes, void* stream);
size_t nsgp_trtri_workspace(int64_t n, int64_t batch, int elem_size);
/* X = chol(A)^-1 in one call (the DSVI whitening chain needs only the inverse): the factor's write-back pass is skipped, so
 * on return A holds intermediate data (strictly lower part = L's, diagonal blocks and upper triangle undefined) and X the
 * lower-triangular inverse.  Replaces psd_safe_cholesky + triangular_solve(eye) of gpytorch's VariationalStrategy (SURVEY
 * A.3) behind /root/reference/models/dgps.py:29-33.  ws: nsgp_potrf_workspace(n, batch) + nsgp_trtri_workspace(n, batch). */
int nsgp_potrf_trtri_f32(float* A, int64_t n, int64_t lda, int64_t sA, int64_t batch, int32_t* info, float* X,
                         int64_t ldx, int64_t sX, void* ws, size_t ws_bytes, void* stream);
int nsgp_potrf_trtri_f64(double* A, int64_t n, int64_t lda, int64_t sA, int64_t batch, int32_t* info, double* X,
                         int64_t ldx, int64_t sX, void* ws, size_t ws_bytes, void* stream);
/* the float64 chain of a float32 model: X32 (same leading dimension and batch stride, in elements) receives a float32 copy
 * of X from the same launches when the in-launch inverse runs (n a multiple of 64, n <= 2048); *wrote32 (host int) says
 * whether it did -- otherwise the caller casts X itself */
int nsgp_potrf_trtri_f64_w32(double* A, int64_t n, int64_t lda, int64_t sA, int64_t batch, int32_t* info, double* X,
                             int64_t ldx, int64_t sX, float* X32, int* wrote32, void* ws, size_t ws_bytes, void* stream);
int nsgp_trtri_f32(const float* L, int64_t n, int64_t ldl, int64_t sL, float* X, int64_t ldx,
                   int64_t sX, int64_t batch, void* ws, size_t ws_bytes, void* stream);
int nsgp_trtri_f64(const double* L, int64_t n, int64_t ldl, int64_t sL, double* X, int64_t ldx,
                   int64_t sX, int64_t batch, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * K6  SVGP epilogue (whitened VariationalStrategy.forward), per output GP b:
 *       mean[b,j] = sum_k A[b,k,j] m[b,k]          (+ mu added by the caller)
 *       var [b,j] = base[b] + sum_k (C[b,k,j]^2 - A[b,k,j]^2)
 *     with A = L^-1 Kzx and C = Lq^T A, both (batch,M,n) row-major.
 *     backward (given gmean, gvar (batch,n)); the Lq C2 term of Abar is added by a beta=1 GEMM:
 *       Abar[b,k,j] = m[b,k] gmean[b,j] - 2 gvar[b,j] A[b,k,j]
 *       C2  [b,k,j] = 2 gvar[b,j] C[b,k,j]          (= dLoss/dC; Lq_bar = tril(A C2^T))
 *       mbar[b,k]   = sum_j A[b,k,j] gmean[b,j]
 * ------------------------------------------------------------------------------------------ */
int nsgp_svgp_colstats_f32(const float* A, const float* C, const float* m, const float* base,
                           int64_t batch, int64_t M, int64_t n, float* mean, float* var, void* stream);
int nsgp_svgp_colstats_f64(const double* A, const double* C, const double* m, const double* base,
                           int64_t batch, int64_t M, int64_t n, double* mean, double* var, void* stream);
int nsgp_svgp_colstats_bwd_f32(const float* A, const float* C, const float* m,
                               const float* gmean, const float* gvar, int64_t batch, int64_t M, int64_t n,
                               float* Abar, float* C2, float* mbar, void* stream);
int nsgp_svgp_colstats_bwd_f64(const double* A, const double* C, const double* m,
                               const double* gmean, const double* gvar, int64_t batch, int64_t M, int64_t n,
                               double* Abar, double* C2, double* mbar, void* stream);

/* K6, fused form used by the DSVI step (the stand-alone colstats kernels above remain for callers that
 * already hold A and C).  The two projections of VariationalStrategy.forward, A = L^-1 Kzx (as W Kzx with
 * W = L^-1 lower) and C = Lq^T A, run as triangular MFMA GEMMs whose EPILOGUE also reduces the column
 * statistics of the tile, so A and C are never re-read:
 *   tri_gemm_colstats: Y[b] = L[b] X[b] (trans = 0, L lower) or L[b]^T X[b] (trans = 1); X, Y (batch,M,n)
 *       part_dot[b,t,j] = sum_{k in row tile t} Y[b,k,j] rowvec[b,k]     (skipped when part_dot == NULL)
 *       part_sq [b,t,j] = sum_{k in row tile t} Y[b,k,j]^2
 *       with t < T = nsgp_svgp_colstats_tiles(M, n, batch, elem_size); partial buffers are (batch,T,n).
 *   colstats_finalize: mean = sum_t part_dot;  var = base + sum_t (part_sq_c - part_sq_a).
 * Backward (hand-derived adjoints, nsgp/svgp.py):
 *   abar : Abar = 2 (Lq C) diag(gvar) + m gmean^T - 2 A diag(gvar)        one GEMM, fused epilogue
 *   lqbar: Lqbar = tril(A diag(2 gvar) C^T)          one GEMM, operand scaled on its way into LDS
 *   rowdot: out[b,k] = sum_j A[b,k,j] g[b,j]         (mbar = A gmean) */
size_t nsgp_svgp_colstats_tiles(int64_t M, int64_t n, int64_t batch, int elem_size);
int nsgp_svgp_tri_gemm_colstats_f32(const float* L, int trans, const float* X, const float* rowvec, int64_t batch,
                                    int64_t M, int64_t n, float* Y, float* part_dot, float* part_sq, void* stream);
int nsgp_svgp_tri_gemm_colstats_f64(const double* L, int trans, const double* X, const double* rowvec, int64_t batch,
                                    int64_t M, int64_t n, double* Y, double* part_dot, double* part_sq,
                                    void* stream);
/* the same launches with the partial buffers laid out with `part_rows` tile rows per batch element (>= the launch's own
 * tile rows; rows it does not fill are the caller's to zero): lets two launches with different tile heights -- the
 * float64-accumulating product below and the float32 product that follows it -- share one set of buffers */
int nsgp_svgp_tri_gemm_colstats_rows_f32(const float* L, int trans, const float* X, const float* rowvec, int64_t batch,
                                         int64_t M, int64_t n, float* Y, float* part_dot, float* part_sq,
                                         int64_t part_rows, void* stream);
int nsgp_svgp_tri_gemm_colstats_rows_f64(const double* L, int trans, const double* X, const double* rowvec, int64_t batch,
                                         int64_t M, int64_t n, double* Y, double* part_dot, double* part_sq,
                                         int64_t part_rows, void* stream);
/* The whitened projection A = L^-1 Kzx with float64 arithmetic on float32 data: Y[b] = W[b] X[b], W float64 lower
 * triangular (= chol(Kzz)^-1), X / Y / rowvec / partials float32; float64 MFMA accumulation.  This is what the
 * reference computes -- gpytorch VariationalStrategy.forward solves L A = Kzx.double() and casts A back (SURVEY A.3;
 * driven by /root/reference/models/dgps.py:48-51) -- where a float32 product W Kzx loses 2e-4 at kappa(Kzz) ~ 1e6.
 * The kernel fills ceil(M / 128) tile rows (nsgp_svgp_f64acc_tiles) of partial buffers laid out with `part_rows`
 * (>= that number) tile rows per batch element; the caller zeroes the rows it does not fill. */
size_t nsgp_svgp_f64acc_tiles(int64_t M);
size_t nsgp_svgp_f64acc_tiles_for(int64_t M, int64_t n, int64_t batch);
/* The same product with the Kzx operand NEVER MATERIALISED: the loader generates each tile
 *   Kzx[b](k, j) = os[b] exp(-1/2 sum_d ((z[b][k][d] - x[j][d]) / ls[b][d])^2)
 * from the inducing points z (batch, M, D), the inputs x (n, D) shared (sx = 0) or (batch, n, D) (sx = n D), ls (batch, D)
 * and os (batch,) -- the arithmetic of nsgp_rbf_build_fwd_f32, operation for operation, so A equals the materialised
 * product bit for bit.  Replaces the (M x n) round trip of Kzx through HBM behind
 * /root/reference/models/dgps.py:44-51 ("fused with the RBF build of Kxz tiles").  Whole tiles only:
 * nsgp_svgp_kzx_gemm_supported() says whether a shape qualifies (D <= 4, M a multiple of the tile rows, n of 64,
 * 32-byte aligned W); -11 otherwise. */
int nsgp_svgp_kzx_gemm_supported(const double* W, int64_t M, int64_t n, int64_t batch, int D);
int nsgp_svgp_kzx_gemm_colstats_f64acc(const double* W, const float* z, const float* x, int64_t sx, const float* ls,
                                       const float* os, int D, const float* rowvec, int64_t batch, int64_t M, int64_t n,
                                       float* Y, float* part_dot, float* part_sq, int64_t part_rows, void* stream);   /* tile rows the launch uses for this shape (64- or 128-row tiles) */
int nsgp_svgp_tri_gemm_colstats_f64acc(const double* W, const float* X, const float* rowvec, int64_t batch, int64_t M,
                                       int64_t n, float* Y, float* part_dot, float* part_sq, int64_t part_rows,
                                       void* stream);
/* The same product with a float64 X (Kzx BUILT in float64, nsgp_rbf_build_fwd_f64): for the layers of a deep GP whose output
 * is the next layer's input.  There the float32 rounding of Kzx, amplified by |W||Kzx| ~ 1e2, costs 3e-5 of the layer's mean
 * and grows ten-fold through a trained next layer (tools/probes/precision_after_training.py: 3.8e-4 on the output mean after
* 1000 Adam steps, the reference's own float32 arithmetic 4.5e-4, with this entry 2.6e-5).  Y and rowvec stay float32; the
 * partials are FLOAT64 (the accumulators' sums unrounded: var = os + colsum(C^2) - colsum(A^2) cancels to << os once q(u) has
 * trained, and float32 partials cost 4e-8 absolute = 4e-5 of a small variance) -- summed by
 * nsgp_svgp_colstats_finalize_affine_p64_f32.  /root/reference/models/dgps.py:92-98 (hidden layers of DeepGP.forward). */
int nsgp_svgp_tri_gemm_colstats_f64acc_b64(const double* W, const double* X64, const float* rowvec, int64_t batch, int64_t M,
                                           int64_t n, float* Y, double* part_dot, double* part_sq, int64_t part_rows,
                                           void* stream);
/* The same with float32 partials (layout and meaning of nsgp_svgp_tri_gemm_colstats_f64acc): for layers whose second
 * projection stays on the float32 kernel (settings.hidden_var_f64 off). */
int nsgp_svgp_tri_gemm_colstats_f64acc_b64p32(const double* W, const double* X64, const float* rowvec, int64_t batch, int64_t M,
                                              int64_t n, float* Y, float* part_dot, float* part_sq, int64_t part_rows,
                                              void* stream);
/* The second projection of such a layer, Y[b] = L[b]^T X[b] (L: float64 copy of the stored lower-triangular Lq; X = the float32
 * A), accumulated in float64: its variance is os + colsum(C^2 - A^2), and float32 accumulation of the 1024-term dot products
 * of C is not consistent with the float64 colsum(A^2); both reach the next layer through sqrt(var) eps.  part_sq: the
 * colsum(C^2) partials from the float64 accumulators, float64 (same layout as above). */
int nsgp_svgp_tri_gemm_colstats_f64acc_t(const double* L, const float* X, int64_t batch, int64_t M, int64_t n, float* Y,
                                         double* part_sq, int64_t part_rows, void* stream);
/* nsgp_svgp_colstats_finalize_affine_f32 over float64 partials (the two entries above); mean / var come out float32. */
int nsgp_svgp_colstats_finalize_affine_p64_f32(const double* part_dot, const double* part_sq_a, const double* part_sq_c,
                                               const float* base, float base_add, int64_t batch, int64_t tiles, int64_t n,
                                               const float* x, int64_t x_batch_stride, int64_t D, const float* w,
                                               int64_t w_batch_stride, const float* c, int64_t c_batch_stride, float* mean,
                                               float* var, void* stream);
/* The same projection A = W Kzx on the INT8 matrix cores, to better than float32 accuracy (csrc/gemm_i8.hip): W and Kzx are
 * cut into signed 7-bit digit planes against a per-row / per-GP power-of-two scale (5 planes of W, 4 of Kzx, the latter
 * evaluated in float64 from the float32 kernel inputs), the 14 plane products with a + b <= 4 are accumulated EXACTLY in
 * int32 on v_mfma_i32_32x32x32_i8 and combined in float64, rounded once to float32: the reference's float64 triangular solve
 * (behind /root/reference/models/dgps.py:44-51) at 64x the float64 MFMA rate per multiply-add.
 *   nsgp_i8_slice_w_f64:  W:(batch,M,M) float64, lower triangle used -> Wd (nsgp_i8_w_planes_bytes), wscale:(batch,M)
 *   nsgp_i8_rbf_build_f32: Z:(batch,M,D), x:(n,D) (x_batch_stride 0) or (batch,n,D), ls:(batch,D), os:(batch,), D <= 4
 *                          -> Kd (nsgp_i8_k_planes_bytes), kscale:(batch,); Kzx_f32: optional (batch,M,n) float32 Kzx rounded
 *                          from the float64 values (what the backward's Wbar = tril(Abar Kzx^T) reads), NULL to skip
 *   nsgp_svgp_tri_gemm_colstats_i8: Y:(batch,M,n) float32 and the column-statistic partials of
 *       nsgp_svgp_tri_gemm_colstats (nsgp_i8_tiles(M) = ceil(M/128) tile rows; float32, or float64 with partials_f64 != 0),
 *       taken from the float64 values.  rowvec may be NULL (then part_dot is not written).
 * `planes` = digit planes of Kzx: 4 (28 bits below os: 1e-6 of max|A| at kappa ~ 1e6) or 5 (35 bits, 15 plane products: layers
 * whose output is the next layer's input).  nsgp_i8_supported(M): 1 <= M <= 4096. */
int nsgp_i8_supported(int64_t M);
size_t nsgp_i8_w_planes_bytes(int64_t batch, int64_t M);
size_t nsgp_i8_k_planes_bytes(int64_t batch, int64_t M, int64_t n, int planes);
size_t nsgp_i8_tiles(int64_t M);
int nsgp_i8_slice_w_f64(const double* W, int64_t batch, int64_t M, void* Wd, double* wscale, void* stream);
int nsgp_i8_rbf_build_f32(const float* Z, const float* x, int64_t x_batch_stride, const float* ls, const float* os,
                          int64_t batch, int64_t M, int64_t n, int D, int planes, void* Kd, double* kscale, float* Kzx_f32,
                          void* stream);
int nsgp_svgp_tri_gemm_colstats_i8(const void* Wd, const double* wscale, const void* Kd, const double* kscale,
                                   int planes, const float* rowvec, int64_t batch, int64_t M, int64_t n, float* Y,
                                   void* part_dot, void* part_sq, int64_t part_rows, int partials_f64, void* stream);
/* "bf16 forward" of BASELINE configs[4] (3-layer DSVI DeepGP, M = 2048: bf16 forward / fp32 Cholesky panels): the two
 * forward projections A = W Kzx, C = Lq^T A of a whitened SVGP layer (gpytorch VariationalStrategy.forward behind
 * /root/reference/models/dgps.py:44-51, driven by :92-98) on v_mfma_f32_32x32x16_bf16 -- bf16 operands, float32
 * accumulation and outputs -- csrc/gemm_bf16.hip.  All bf16 buffers are passed as void* (2-byte elements).
 *   nsgp_svgp_tri_gemm_colstats_bf16: Y[b] = P[b] Qt[b]^T with P (M x M) bf16 triangular (tri = 1: lower, k <= m;
 *       tri = 2: upper, k >= m; the zeros are in memory), Qt (n x M) bf16 (k contiguous); Y float32 (M x n); YT:
 *       optional bf16 transposed copy of Y (n x M), the Qt operand of the next product; part_dot / part_sq: the
 *       column-statistic partials of nsgp_svgp_tri_gemm_colstats (float32, ceil(M / 128) tile rows = nsgp_svgp_bf16_tiles).
 *   nsgp_cast_sq_bf16_{f32,f64}: dst (b, n, n) bf16 <- src, optionally transposed and / or lower triangle only.
 *   nsgp_rbf_build_t_bf16: Kxz[b][i][k] = bf16(os[b] RBF-ARD(x_i, z[b][k]; ls[b])), the transpose of Kzx, k contiguous
 *       (z:(b,M,D) x:(n,D) shared [sxb = 0] or (b,n,D) [sxb = n D]).
 *   nsgp_transpose_cast_bf16: dst (b, n, M) bf16 <- transpose of src (b, M, n) float32 (A -> the Qt operand of product 2
 *       when product 1 ran in float32 / float64). */
size_t nsgp_svgp_bf16_tiles(int64_t M);
int nsgp_transpose_cast_bf16(const float* src, void* dst, int64_t batch, int64_t M, int64_t n, void* stream);
int nsgp_svgp_tri_gemm_colstats_bf16(const void* P, int tri, const void* Qt, const float* rowvec, int64_t batch, int64_t M,
                                     int64_t n, float* Y, void* YT, float* part_dot, float* part_sq, void* stream);
int nsgp_cast_sq_bf16_f32(const float* src, void* dst, int64_t n, int64_t batch, int transpose, int tril, void* stream);
int nsgp_cast_sq_bf16_f64(const double* src, void* dst, int64_t n, int64_t batch, int transpose, int tril, void* stream);
int nsgp_rbf_build_t_bf16(const float* z, const float* x, const float* ls, const float* os, int64_t batch, int64_t M,
                          int64_t n, int64_t D, int64_t sxb, void* out, void* stream);
int nsgp_svgp_colstats_finalize_f32(const float* part_dot, const float* part_sq_a, const float* part_sq_c,
                                    const float* base, int64_t batch, int64_t tiles, int64_t n, float* mean,
                                    float* var, void* stream);
int nsgp_svgp_colstats_finalize_f64(const double* part_dot, const double* part_sq_a, const double* part_sq_c,
                                    const double* base, int64_t batch, int64_t tiles, int64_t n, double* mean,
                                    double* var, void* stream);
int nsgp_svgp_abar_f32(const float* Lq, const float* C, const float* A, const float* m, const float* gmean,
                       const float* gvar, int64_t batch, int64_t M, int64_t n, float* Abar, void* stream);
int nsgp_svgp_abar_f64(const double* Lq, const double* C, const double* A, const double* m, const double* gmean,
                       const double* gvar, int64_t batch, int64_t M, int64_t n, double* Abar, void* stream);
size_t nsgp_svgp_lqbar_workspace(int64_t batch, int64_t M, int64_t n, int elem_size);
int nsgp_svgp_lqbar_f32(const float* A, const float* C, const float* gvar, int64_t batch, int64_t M, int64_t n,
                        float* Lqbar, void* ws, size_t ws_bytes, void* stream);
int nsgp_svgp_lqbar_f64(const double* A, const double* C, const double* gvar, int64_t batch, int64_t M, int64_t n,
                        double* Lqbar, void* ws, size_t ws_bytes, void* stream);
/* Lqbar = tril(...) + beta * Lqbar: accumulates onto a gradient that is already in place (the KL term's, or an earlier
 * application of a tied layer) -- the product writes straight into the optimiser's gradient bucket */
int nsgp_svgp_lqbar_acc_f32(const float* A, const float* C, const float* gvar, int64_t batch, int64_t M, int64_t n,
                            float beta, float* Lqbar, void* ws, size_t ws_bytes, void* stream);
int nsgp_svgp_lqbar_acc_f64(const double* A, const double* C, const double* gvar, int64_t batch, int64_t M, int64_t n,
                            double beta, double* Lqbar, void* ws, size_t ws_bytes, void* stream);
/* The same two steps with the layer's affine prior mean and the cheap column reductions folded in, so that the mean
 * module of models/dgps.py:40-43 (gpytorch ConstantMean / LinearMean), the `+ 1e-4` of the predictive variance and the
 * output-scale gradient cost no launches of their own:
 *   colstats_finalize_affine: mean = sum_t part_dot + sum_d x[b,j,d] w[b,d] + c[b];
 *                             var  = (base + base_add) + sum_t (part_sq_c - part_sq_a).
 *       x:(.., n, D) row-major, w:(.., D), c:(..,) with BATCH strides that may be 0 (shared by the batch);
 *       w and/or c may be NULL (no linear / constant part).
 *   rowdot_affine: out[b,k] = sum_j A[b,k,j] g[b,j];  out_gv[b] = sum_j gv[b,j] (gv may be NULL);
 *                  out_1 = sum_j g[b,j];  out_x[d] = sum_j x[b,j,d] g[b,j]  -- per batch ((batch,), (batch,D)) or, with
 *                  shared != 0, summed over the batch ((1,), (D,)); out_1 / out_x may be NULL. */
int nsgp_svgp_colstats_finalize_affine_f32(const float* part_dot, const float* part_sq_a, const float* part_sq_c,
                                           const float* base, float base_add, int64_t batch, int64_t tiles, int64_t n,
                                           const float* x, int64_t x_batch_stride, int64_t D, const float* w,
                                           int64_t w_batch_stride, const float* c, int64_t c_batch_stride, float* mean,
                                           float* var, void* stream);
int nsgp_svgp_colstats_finalize_affine_f64(const double* part_dot, const double* part_sq_a, const double* part_sq_c,
                                           const double* base, double base_add, int64_t batch, int64_t tiles,
                                           int64_t n, const double* x, int64_t x_batch_stride, int64_t D,
                                           const double* w, int64_t w_batch_stride, const double* c,
                                           int64_t c_batch_stride, double* mean, double* var, void* stream);
int nsgp_rowdot_affine_f32(const float* A, const float* g, const float* gv, const float* x, int64_t x_batch_stride,
                           int64_t D, int shared, int64_t batch, int64_t M, int64_t n, float* out, float* out_gv,
                           float* out_x, float* out_1, void* stream);
int nsgp_rowdot_affine_f64(const double* A, const double* g, const double* gv, const double* x, int64_t x_batch_stride,
                           int64_t D, int shared, int64_t batch, int64_t M, int64_t n, double* out, double* out_gv,
                           double* out_x, double* out_1, void* stream);
int nsgp_rowdot_f32(const float* A, const float* g, int64_t batch, int64_t M, int64_t n, float* out, void* stream);
int nsgp_rowdot_f64(const double* A, const double* g, int64_t batch, int64_t M, int64_t n, double* out,
                    void* stream);

/* DeepGPLayer sampling:  h[s,i,c] = mean[c,s?,i] + sqrt(var[c,s?,i]) * eps[s,i,c]
 *   mean/var are (b, ns, n) with ns == 1 (deterministic first layer, broadcast over S) or ns == S.
 *   backward accumulates gmean/gvar (b, ns, n) from gh (S,n,b). */
int nsgp_dgp_sample_fwd_f32(const float* mean, const float* var, const float* eps, int64_t S, int64_t ns,
                            int64_t n, int64_t b, float* h, void* stream);
int nsgp_dgp_sample_bwd_f32(const float* var, const float* eps, const float* gh, int64_t S, int64_t ns,
                            int64_t n, int64_t b, float* gmean, float* gvar, void* stream);
int nsgp_dgp_sample_fwd_f64(const double* mean, const double* var, const double* eps, int64_t S, int64_t ns,
                            int64_t n, int64_t b, double* h, void* stream);
int nsgp_dgp_sample_bwd_f64(const double* var, const double* eps, const double* gh, int64_t S, int64_t ns,
                            int64_t n, int64_t b, double* gmean, double* gvar, void* stream);

/* ------------------------------------------------------------------------------------------
 * K7  ELBO reductions.
 *   gauss_ell: out[s] = scale * sum_i -0.5*(((y_i-mu_si)^2 + v_si)/noise + log noise + log 2pi)
 *              (GaussianLikelihood.expected_log_prob summed over the minibatch, one value per sample
 *               s as VariationalELBO returns it; DeepApproximateMLL then averages over s --
 *               experiments/deepgp_spatial_bench.py:61,84-88); noise:(1) device.
 *              bwd: upstream gout:(S) on the DEVICE -> gmu, gv (S,n) and gnoise (1).
 *   kl_whitened: out[b] = 0.5*(||Lq_b||_F^2(lower) + m_b.m_b - M - 2 sum log|diag Lq_b|)
 *              (CholeskyVariationalDistribution vs N(0,I)); bwd writes gm, gLq (lower).
 * ------------------------------------------------------------------------------------------ */
size_t nsgp_reduce_workspace(int64_t n_elems, int elem_size);
int nsgp_gauss_ell_fwd_f32(const float* y, const float* mu, const float* v, const float* noise,
                           int64_t S, int64_t n, float scale, float* out, void* ws, size_t ws_bytes, void* stream);
int nsgp_gauss_ell_bwd_f32(const float* y, const float* mu, const float* v, const float* noise,
                           int64_t S, int64_t n, float scale, const float* gout, float* gmu, float* gv, float* gnoise,
                           void* ws, size_t ws_bytes, void* stream);
int nsgp_gauss_ell_fwd_f64(const double* y, const double* mu, const double* v, const double* noise,
                           int64_t S, int64_t n, double scale, double* out, void* ws, size_t ws_bytes, void* stream);
int nsgp_gauss_ell_bwd_f64(const double* y, const double* mu, const double* v, const double* noise,
                           int64_t S, int64_t n, double scale, const double* gout, double* gmu, double* gv, double* gnoise,
                           void* ws, size_t ws_bytes, void* stream);
/* Scalar ("total") forms for the fused ELBO tail: one output, one device-resident upstream gradient.
 *   gauss_ell_total: out[0] = scale * sum_s sum_i E_q log N(y_i | f_si, noise)   (fold 1/S, 1/B, sign into scale);
 *       backward with gout a 1-element DEVICE scalar (no host round trip, no per-sample gradient vector)
 *   kl_whitened_total: out[0] = scale * sum_b KL(N(m_b, Lq_b Lq_b^T) || N(0, I));  backward: scale * gout[0] * dKL */
/* The whole DSVI objective as one scalar -- DeepApproximateMLL(VariationalELBO(...)) of /root/reference/models/dgps.py's model,
 * driven at /root/reference/experiments/deepgp_spatial_bench.py:61,84-88 -- in TWO launches forward and TWO backward:
 *   out[0] = ell_scale * sum_s sum_i E_q log N(y_i | f_si, noise)
 *          + kl_scale  * sum_g sum_b KL(N(m_gb, Lq_gb Lq_gb^T) || N(0, I))           (fold 1/(B S), beta/num_data, signs into the scales)
 * mu, v:(S,n); `ngroups` <= 8 variational groups (one per layer: tied layers counted once) given as HOST arrays of device
 * pointers m[g]:(batch[g], M), Lq[g]:(batch[g], M, M) (lower triangle used) and host batch[g]; all groups share M.
 * Backward: gout[0] (device) is the upstream gradient; writes gmu, gv:(S,n), gnoise[0] (optional), gm[g], gLq[g] (shapes of
 * m / Lq; the strict upper triangle of gLq is zero).  ws: nsgp_dsvi_objective_workspace bytes.  The per-term entry points
 * below stay for single-layer / non-whitened models. */
size_t nsgp_dsvi_objective_workspace(int64_t S, int64_t n, int64_t M, int64_t total_batch, int elem_size);
int nsgp_dsvi_objective_fwd_f32(const float* y, const float* mu, const float* v, const float* noise, int64_t S, int64_t n,
                                float ell_scale, int ngroups, const void* const* m, const void* const* Lq,
                                const int64_t* batch, int64_t M, float kl_scale, float* out, void* ws, size_t wsb,
                                void* stream);
int nsgp_dsvi_objective_fwd_f64(const double* y, const double* mu, const double* v, const double* noise, int64_t S, int64_t n,
                                double ell_scale, int ngroups, const void* const* m, const void* const* Lq,
                                const int64_t* batch, int64_t M, double kl_scale, double* out, void* ws, size_t wsb,
                                void* stream);
int nsgp_dsvi_objective_bwd_f32(const float* y, const float* mu, const float* v, const float* noise, int64_t S, int64_t n,
                                float ell_scale, int ngroups, const void* const* m, const void* const* Lq,
                                const int64_t* batch, int64_t M, float kl_scale, const float* gout, float* gmu, float* gv,
                                float* gnoise, void* const* gm, void* const* gLq, void* ws, size_t wsb, void* stream);
int nsgp_dsvi_objective_bwd_f64(const double* y, const double* mu, const double* v, const double* noise, int64_t S, int64_t n,
                                double ell_scale, int ngroups, const void* const* m, const void* const* Lq,
                                const int64_t* batch, int64_t M, double kl_scale, const double* gout, double* gmu, double* gv,
                                double* gnoise, void* const* gm, void* const* gLq, void* ws, size_t wsb, void* stream);
int nsgp_gauss_ell_total_fwd_f32(const float* y, const float* mu, const float* v, const float* noise, int64_t S,
                                 int64_t n, float scale, float* out, void* ws, size_t ws_bytes, void* stream);
int nsgp_gauss_ell_total_fwd_f64(const double* y, const double* mu, const double* v, const double* noise, int64_t S,
                                 int64_t n, double scale, double* out, void* ws, size_t ws_bytes, void* stream);
int nsgp_gauss_ell_total_bwd_f32(const float* y, const float* mu, const float* v, const float* noise, int64_t S,
                                 int64_t n, float scale, const float* gout, float* gmu, float* gv, float* gnoise,
                                 void* ws, size_t ws_bytes, void* stream);
int nsgp_gauss_ell_total_bwd_f64(const double* y, const double* mu, const double* v, const double* noise, int64_t S,
                                 int64_t n, double scale, const double* gout, double* gmu, double* gv, double* gnoise,
                                 void* ws, size_t ws_bytes, void* stream);
int nsgp_kl_whitened_total_fwd_f32(const float* m, const float* Lq, int64_t batch, int64_t M, float scale, float* out,
                                   void* ws, size_t ws_bytes, void* stream);
int nsgp_kl_whitened_total_fwd_f64(const double* m, const double* Lq, int64_t batch, int64_t M, double scale,
                                   double* out, void* ws, size_t ws_bytes, void* stream);
/* out[0] = addin[0] + scale * sum_b KL_b: the KL terms of several layers (and the likelihood term) chain into ONE
 * scalar without separate additions (addin may be NULL; with addin given, batch must be > 0). */
int nsgp_kl_whitened_total_acc_fwd_f32(const float* m, const float* Lq, int64_t batch, int64_t M, float scale,
                                       const float* addin, float* out, void* ws, size_t wsb, void* stream);
int nsgp_kl_whitened_total_acc_fwd_f64(const double* m, const double* Lq, int64_t batch, int64_t M, double scale,
                                       const double* addin, double* out, void* ws, size_t wsb, void* stream);
int nsgp_kl_whitened_total_bwd_f32(const float* m, const float* Lq, int64_t batch, int64_t M, float scale,
                                   const float* gout, float* gm, float* gLq, void* stream);
int nsgp_kl_whitened_total_bwd_f64(const double* m, const double* Lq, int64_t batch, int64_t M, double scale,
                                   const double* gout, double* gm, double* gLq, void* stream);
int nsgp_kl_whitened_fwd_f32(const float* m, const float* Lq, int64_t batch, int64_t M, float* out,
                             void* ws, size_t ws_bytes, void* stream);
int nsgp_kl_whitened_bwd_f32(const float* m, const float* Lq, int64_t batch, int64_t M, float gout,
                             float* gm, float* gLq, void* stream);
int nsgp_kl_whitened_fwd_f64(const double* m, const double* Lq, int64_t batch, int64_t M, double* out,
                             void* ws, size_t ws_bytes, void* stream);
int nsgp_kl_whitened_bwd_f64(const double* m, const double* Lq, int64_t batch, int64_t M, double gout,
                             double* gm, double* gLq, void* stream);

/* ------------------------------------------------------------------------------------------
 * Small helpers on the same stream
 * ------------------------------------------------------------------------------------------ */
/* P(i,i) *= factor for a batch of n x n matrices, in place (Phi of the Cholesky backward: tril with halved diagonal,
 * the strict upper triangle being masked by the GEMM that consumes P as a LOWER operand) */
int nsgp_scale_diag_f32(float* P, int64_t n, int64_t ld, int64_t sP, int64_t batch, float factor, void* stream);
int nsgp_scale_diag_f64(double* P, int64_t n, int64_t ld, int64_t sP, int64_t batch, double factor, void* stream);
/* out = tril(P) with halved diagonal, symmetrised: S = Phi(P) + Phi(P)^T   (Cholesky backward) */
int nsgp_chol_bwd_phi_sym_f32(const float* P, float* S, int64_t n, int64_t ld, int64_t sP, int64_t batch, void* stream);
int nsgp_chol_bwd_phi_sym_f64(const double* P, double* S, int64_t n, int64_t ld, int64_t sP, int64_t batch, void* stream);
/* precision changes of (batch, rows, cols) row-major blocks */
int nsgp_cast_f64_to_f32(const double* src, int64_t lds, float* dst, int64_t ldd, int64_t rows, int64_t cols, void* stream);
int nsgp_cast_f32_to_f64(const float* src, int64_t lds, double* dst, int64_t ldd, int64_t rows, int64_t cols, void* stream);
/* counter-based standard normals (Philox4x32-10 + Box-Muller), keyed by
 * (seed, stream_id, global_row, sample, column) so the union over data-parallel ranks equals the
 * single-GPU draw: eps[s, i, c] for rows row0 .. row0+n-1.
 * step_dev (device int64, may be NULL): when given, its value replaces the high 32 bits of stream_id,
 * so a captured hipGraph draws fresh noise on every replay. */
int nsgp_philox_normal_f32(uint64_t seed, uint64_t stream_id, const int64_t* step_dev, int64_t row0, int64_t S,
                           int64_t n, int64_t b, float* eps, void* stream);
int nsgp_philox_normal_f64(uint64_t seed, uint64_t stream_id, const int64_t* step_dev, int64_t row0, int64_t S,
                           int64_t n, int64_t b, double* eps, void* stream);
/* fused Adam over one flat parameter buffer (torch.optim.Adam semantics, lr 0.01 in
 * experiments/deepgp_spatial_bench.py:74-76); step is the 1-based step count, or step_dev a device
 * int64 holding it (used instead of `step` when non-NULL: hipGraph replays); grad_scale multiplies the
 * gradient first. */
int nsgp_adam_step_f32(float* p, const float* g, float* exp_avg, float* exp_avg_sq, int64_t n,
                       float lr, float beta1, float beta2, float eps, int64_t step, const int64_t* step_dev,
                       float grad_scale, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NSGP_H */
