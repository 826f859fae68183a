// potrf.hip -- K4 blocked right-looking Cholesky (lower) and K5 triangular inverse, batched.
//
// Replaces psd_safe_cholesky (models/gibbs_kernels.py:201,298), gpytorch
// VariationalStrategy._cholesky_factor / L.inv_matmul, the Cholesky under
// ExactMarginalLogLikelihood / MultivariateNormal.log_prob and torch.triangular_solve(eye, chol)
// (models/gibbs_kernels.py:203,300).
//
// potrf, per 64-column panel j (right-looking), ONE launch per panel (potrf_step_kernel):
//   panel workgroups   (4 waves each) apply the PREVIOUS panel's rank-64 update to their own block column, then
//                  re-factor the 64x64 diagonal block in LDS -- four 16-column sub-panels, each factored by one
//                  wave in registers, the rest updated with 16x16x4 MFMAs -- invert the 16x16 diagonal sub-blocks
//                  and solve their own 64-row slab of the panel L21 = A21 L11^-T by blocked substitution on the
//                  MFMA.  The factor goes to a side buffer (other workgroups still read A11); a final kernel
//                  copies the factors into place and zeroes the strict upper triangle.
//   update workgroups  the rest of the previous panel's rank-64 update (columns beyond the current panel): one
//                  64x64 lower tile each, ONE K step, all loads issued up front.  They touch no data of the panel
//                  workgroups, so the update overlaps the next factorisation (look-ahead of depth 1).
//   n >= 2048      rank-64 updates stay inside a 256-column outer panel; the rest of the trailing matrix is
//                  updated once per outer panel (K = 256) on the MFMA GEMM.
// trtri: invert the 64x64 diagonal blocks (16x16 substitution + MFMA merges), then merge pairs of blocks
//   bottom-up, X21 = -B^-1 (C A^-1), every level two batched MFMA GEMMs (log2(n/64) levels).
#include "common.h"

// from gemm.hip
extern "C" int nsgp_gemm_f32(int64_t, int64_t, int64_t, float, const float*, int64_t, int64_t, int64_t, int64_t,
                             const float*, int64_t, int64_t, int64_t, int64_t, float, float*, int64_t, int64_t,
                             int64_t, int64_t, int64_t, int, void*, size_t, void*);
extern "C" int nsgp_gemm_f64(int64_t, int64_t, int64_t, double, const double*, int64_t, int64_t, int64_t, int64_t,
                             const double*, int64_t, int64_t, int64_t, int64_t, double, double*, int64_t, int64_t,
                             int64_t, int64_t, int64_t, int, void*, size_t, void*);

namespace {

#ifdef NSGP_POTRF_STAMPS
// Diagnostic build only (tools/probes/potrf_stamps.py): workgroup 0 of matrix 0 records shader-clock stamps at the phase
// boundaries of panel_body2, one 16-word record per panel launch.  Never in the shipped library.
__device__ unsigned long long* nsgp_pstamp_buf = nullptr;
__device__ unsigned long long nsgp_pstamp_cap = 0;
#define NSGP_PSTAMP(i) do { if (pst_on) pst[i] = __builtin_amdgcn_s_memtime(); } while (0)
// arrival of every wave at the barrier that ends a phase (record 16 + 8 w + i of the panel's 64 words)
#define NSGP_WSTAMP(i) do { if (wst_on) wst[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define NSGP_PSTAMP(i) do { } while (0)
#define NSGP_WSTAMP(i) do { } while (0)
#endif

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

// ---- wave-level building blocks --------------------------------------------------------------------
// A wave holds ONE ROW PER LANE of a sub-panel in registers; values of another row are broadcast with
// v_readlane (the lane index is a compile-time constant after full unrolling), so a 16-column sub-panel
// factorisation is straight-line VALU code: no LDS traffic, no barriers.
__device__ __forceinline__ float bcast(float v, int l) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}
__device__ __forceinline__ double bcast(double v, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

// 1/sqrt(a) to full precision without the long v_sqrt / v_div sequences that would sit on the serial
// pivot chain: hardware estimate + Newton steps  r <- r (1.5 - 0.5 a r^2)  (each step squares the error).
__device__ __forceinline__ float fast_rsqrt(float a) {
    float r = __frsqrt_rn(a);
    r = r * (1.5f - 0.5f * a * r * r);
    return r;
}
__device__ __forceinline__ double fast_rsqrt(double a) {
    // v_rsq_f64 is good to 5.2e-8 (tools/probes/rsq_f64_accuracy.hip); ONE cubic step r (1 + e/2 + 3 e^2/8), e = 1 - a r^2,
    // lands within 0.62 ulp with a dependent chain of 4 operations (two quadratic Newton steps: 8, 2.2 ulp) -- this sits
    // on the serial pivot chain of every column
    const double r = __builtin_amdgcn_rsq(a);
    const double e = __builtin_fma(-(a * r), r, 1.0);
    return __builtin_fma(r * e, __builtin_fma(e, 0.375, 0.5), r);
}

// Ragged tail (nb < 64): same algorithm with the block in LDS (lane i owns row i) and run-time loop
// bounds; the lane index of v_readlane is a wave-uniform loop counter.  Only the last panel of a matrix
// whose order is not a multiple of 64 takes this path.
template <typename T> __device__ __forceinline__ T bcast_dyn(T v, int l);
template <> __device__ __forceinline__ float bcast_dyn<float>(float v, int l) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), __builtin_amdgcn_readfirstlane(l)));
}
template <> __device__ __forceinline__ double bcast_dyn<double>(double v, int l) {
    const int ul = __builtin_amdgcn_readfirstlane(l);
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), ul);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), ul);
    return __hiloint2double(hi, lo);
}

template <typename T> __device__ int factor_lds(T* Ls, int nb, int lane) {
    int bad = 0;
    const bool live = lane < nb;
    for (int k = 0; k < nb; ++k) {
        const T aik = live ? Ls[lane * LDD + k] : T(0);
        const T akk = bcast_dyn<T>(aik, k);
        if (!(akk > T(0)) && bad == 0) bad = k + 1;
        const T piv = t_sqrt(akk);
        const T inv = T(1) / piv;
        const T lik = lane == k ? piv : (lane > k ? aik * inv : T(0));
        if (live) Ls[lane * LDD + k] = lik;
        for (int j = k + 1; j < nb; ++j) {
            const T ljk = bcast_dyn<T>(lik, j);
            if (live) Ls[lane * LDD + j] -= lik * ljk;
        }
    }
    return bad;
}

// 16x16x4 MFMA (both precisions have one) for the in-panel block operations.
template <typename T> struct Mma16;
template <> struct Mma16<double> {
    typedef double acc_t __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ acc_t mma(double a, double b, acc_t c) {
        return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ int crow(int r, int lane) { return (lane >> 4) + 4 * r; }
};
template <> struct Mma16<float> {
    typedef float acc_t __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ int crow(int r, int lane) { return 4 * (lane >> 4) + r; }
};
// Pins a loaded value in a register: stops the compiler from sinking a clamped (always valid) load under the
// bounds condition of its later select, which would turn N independent loads into N serialized
// load -> s_waitcnt -> use round trips (measured: 13 us instead of 5 for the rank-64 update).
template <typename T> __device__ __forceinline__ void keep(T& v) { asm volatile("" : "+v"(v)); }
constexpr int SB = 16;          // sub-block of the panel
constexpr int LDI = SB + 1;

// Sub-panel step of the 64x64 diagonal block held in LDS (S, leading dimension LDD): ONE wave takes
// columns [c0, c0+16) with row `lane` in registers, factors them (16 pivots, v_readlane broadcasts),
// and writes them back.  Lanes < c0 write zeros (upper part of L).
#ifndef NSGP_POTRF_LDSCOL
#define NSGP_POTRF_LDSCOL 1
#endif
template <typename T, int C0> __device__ __forceinline__ int factor_subpanel(T* S, T* rd, T* cb, int lane) {
    T a[SB];
#pragma unroll
    for (int j = 0; j < SB; ++j) a[j] = S[lane * LDD + C0 + j];
    int bad = 0;
#if NSGP_POTRF_LDSCOL
    // (Tried in round 3 and dropped: FOUR pivots at a time -- the pivot rows publish the 4 x 4 diagonal block through LDS, every
    // lane factors it redundantly in registers, solves its own row against it and takes the rank-4 update's multipliers from
    // LDS as broadcast 128-bit reads.  Fewer instructions, but two LDS round trips and a 4-deep rsqrt chain per block: 1630
    // cycles per four pivots against 940 here, tools/probes/potrf_stamps.py -- profiles/r03/potrf_stamps_blk4_experiment.log.)
    // The wave is bound by instruction issue, and 2/5 of its instructions were v_readlane pairs (one pair per rank-1 update
    // a[j] -= l_k[row] * l_k[C0 + j], the multiplier broadcast from lane C0 + j through an SGPR pair -- plus v_writelane spills
    // of those SGPRs).  Here only the update the NEXT pivot waits for (j = k + 1) takes that route; the pivot column's
    // other 15 - k multipliers go through LDS: lanes C0 .. C0 + 15 publish l_k[lane], every lane reads l_k[C0 + j] back
    // with ONE broadcast ds_read_b64 per j (the LDS serves a wave's requests in order, so the reads see the write), and
    // the updates are applied one pivot later, when the data has long arrived (cb: two buffers of 16).
    T pend[SB];                                          // multipliers of the previous pivot, read one iteration ago
    T lprev = T(0);
    // addresses once per sub-panel, in VGPRs: the publishing lanes' slot, and a (uniform) read base that the compiler keeps
    // in a VGPR so that every read is base + immediate (it re-materialised an SGPR address per read: 10 of 53
    // instructions per pivot)
    const int pub = lane - C0;
    const bool is_pub = (unsigned)pub < (unsigned)SB;
    int wo = is_pub ? pub : 0, ro = lane & 0;            // element offsets pinned in VGPRs (the pointers stay LDS pointers)
    asm volatile("" : "+v"(wo), "+v"(ro));
    T rdv = T(0);                                        // lane k of the publishing lanes keeps 1 / L[p][p] of pivot k
#pragma unroll
    for (int k = 0; k < SB; ++k) {
        const int p = C0 + k;
        const T akk = bcast(a[k], p);
        if (!(akk > T(0)) && bad == 0) bad = p + 1;
        const T inv = fast_rsqrt(akk);
        const T piv = akk * inv;
        rdv = lane == p ? inv : rdv;
        const T lik = lane == p ? piv : (lane > p ? a[k] * inv : T(0));
        a[k] = lik;
        if (k + 1 < SB) {
            if (is_pub) cb[wo + (k & 1) * SB] = lik;
            a[k + 1] -= lik * bcast(lik, C0 + k + 1);                        // the next pivot's column: no LDS round trip
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");          // compiler ordering: the write is issued before the reads
        }
        if (k > 0) {                                                         // pivot k - 1's other updates (data is here)
#pragma unroll
            for (int j = k + 1; j < SB; ++j) a[j] -= lprev * pend[j];
        }
        if (k + 2 < SB) {
#pragma unroll
            for (int j = k + 2; j < SB; ++j) pend[j] = cb[ro + (k & 1) * SB + j];
        }
        lprev = lik;
    }
    if (is_pub) rd[lane] = rdv;                                              // 1 / L[p][p], one store for the sub-panel
#else
#pragma unroll
    for (int k = 0; k < SB; ++k) {
        const int p = C0 + k;
        const T akk = bcast(a[k], p);
        if (!(akk > T(0)) && bad == 0) bad = p + 1;
        const T inv = fast_rsqrt(akk);
        const T piv = akk * inv;
        if (lane == 0) rd[p] = inv;                                          // 1 / L[p][p]
        const T lik = lane == p ? piv : (lane > p ? a[k] * inv : T(0));
        a[k] = lik;
#pragma unroll
        for (int j = k + 1; j < SB; ++j) a[j] -= lik * bcast(lik, C0 + j);
    }
#endif
#pragma unroll
    for (int j = 0; j < SB; ++j) S[lane * LDD + C0 + j] = a[j];
    return bad;
}

// Inverse of the 16x16 diagonal sub-block B0 of the factored block: lane c < 16 builds column c by forward
// substitution (L entries are broadcast reads from LDS, reciprocal pivots from rd).  All 136 reads are issued first
// (independent, pipelined) and the substitution runs column by column -- x_k = acc_k / L_kk, then acc_i -= L_ik x_k for
// every i > k, independent FMAs -- instead of row by row with a dependent FMA chain behind each LDS read: 3700 -> 1400 cycles
// (float32; tools/probes/potrf_stamps.py), which took this off the critical path of the panel phases.  Same operations in the
// same order per element as the row-by-row form: bit-identical results.
template <typename T> __device__ __forceinline__ void invert_subblock(const T* S, const T* rd, T* Dinv, int B0,
                                                                      int lane) {
    if (lane >= SB) return;
    const T* Lw = S + (B0 * SB) * LDD + B0 * SB;
    T Lr[SB * (SB - 1) / 2], rv[SB], acc[SB];
#pragma unroll
    for (int i = 1; i < SB; ++i)
#pragma unroll
        for (int k = 0; k < i; ++k) Lr[i * (i - 1) / 2 + k] = Lw[i * LDD + k];
#pragma unroll
    for (int i = 0; i < SB; ++i) { rv[i] = rd[B0 * SB + i]; acc[i] = lane == i ? T(1) : T(0); }
#pragma unroll
    for (int k = 0; k < SB; ++k) {
        const T xk = k < lane ? T(0) : acc[k] * rv[k];
        acc[k] = xk;
#pragma unroll
        for (int i = k + 1; i < SB; ++i) acc[i] -= Lr[i * (i - 1) / 2 + k] * xk;
    }
#pragma unroll
    for (int i = 0; i < SB; ++i) Dinv[(B0 * SB + i) * LDI + lane] = acc[i];
}

// FULL 64-wide panel.  grid.x = nslab workgroups of 4 waves, each factors the
// diagonal block in LDS (redundantly: no grid-wide dependency) -- 4 sub-panels of 16 columns, each factored
// by wave 0 in registers, the rest of the block updated by all waves with 16x16x4 MFMAs -- invert the four
// 16x16 diagonal sub-blocks, and solve their own 64-row slab  L21 = A21 L11^-T  by blocked substitution
// (MFMA updates, multiplication by the 16x16 inverses).  The slab's global loads are issued before the
// factorisation so their latency is hidden.  Block 0 publishes L11 to the side buffer (other workgroups
// still read A11); potrf_finalize_kernel copies the factors into place at the end.
// `pre`: the rank-64 update of the PREVIOUS panel (columns [j0-64, j0)) has not been applied to this block column
// yet; the workgroup applies it to its own diagonal block and slab first (two 64x64x64 MFMA products on blocks
// that are loaded together with everything else), so that the previous panel's update of the REST of the
// trailing matrix can run concurrently in the same launch (potrf_step_kernel).
template <typename T>
__device__ __forceinline__ void panel_body(unsigned char* panel_smem, T* __restrict__ A, int64_t n, int64_t lda,
                                           int64_t sA, int64_t j0, T* __restrict__ wsL, int64_t npanels,
                                           int32_t* __restrict__ info, int64_t blk, int64_t b, bool pre) {
    typedef Mma16<T> MM;
    typedef typename MM::acc_t acc_t;
    T* S = reinterpret_cast<T*>(panel_smem);            // [64][LDD]   diagonal block -> L11
    T* Xs = S + NB * LDD;                               // [64][LDD]   this workgroup's slab
    T* Dinv = Xs + NB * LDD;                            // [4][16][LDI] inverses of the 16x16 diagonal sub-blocks
    T* rd = Dinv + 4 * SB * LDI;                        // [64] reciprocal pivots
    T* Ps = rd + NB;                                    // [64][LDD]   previous panel's L, diagonal-block rows
    T* Qs = Ps + NB * LDD;                              // [64][LDD]   previous panel's L, slab rows
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    T* Ab = A + b * sA;
    const int64_t pj = j0 / NB;
    const int64_t r0 = j0 + NB + blk * NB;
    const int rows = r0 >= n ? 0 : (int)((n - r0) < NB ? (n - r0) : NB);
    const int fm = lane & 15, fk = lane >> 4;            // MFMA operand lane -> (m|n, k)
    // global -> LDS: wave w takes rows [16w, 16w+16) of every block, one coalesced 64-element row per
    // instruction; the slab rows stay in registers until the factorisation is done.
    T xr[SB];
    {
        T dr[SB], pr[SB], qr[SB];                        // loads first, LDS stores after (one exposed latency)
#pragma unroll
        for (int i = 0; i < SB; ++i) dr[i] = Ab[(j0 + w * SB + i) * lda + j0 + lane];
#pragma unroll
        for (int i = 0; i < SB; ++i) {                   // clamped row: unconditional loads (no branch per load)
            const int64_t rr = r0 + w * SB + i;
            xr[i] = Ab[(rr < n ? rr : n - 1) * lda + j0 + lane];
        }
        if (pre) {
#pragma unroll
            for (int i = 0; i < SB; ++i) {
                const int64_t rr = r0 + w * SB + i;
                pr[i] = Ab[(j0 + w * SB + i) * lda + j0 - NB + lane];
                qr[i] = Ab[(rr < n ? rr : n - 1) * lda + j0 - NB + lane];
            }
        }
#pragma unroll
        for (int i = 0; i < SB; ++i) { keep(dr[i]); keep(xr[i]); }
        if (pre) {
#pragma unroll
            for (int i = 0; i < SB; ++i) { keep(pr[i]); keep(qr[i]); }
        }
#pragma unroll
        for (int i = 0; i < SB; ++i) {
            S[(w * SB + i) * LDD + lane] = dr[i];
            xr[i] = (w * SB + i) < rows ? xr[i] : T(0);
            if (pre) {
                Ps[(w * SB + i) * LDD + lane] = pr[i];
                Qs[(w * SB + i) * LDD + lane] = (w * SB + i) < rows ? qr[i] : T(0);
            }
        }
    }
    __syncthreads();
    if (pre) {                                           // S -= P P^T : wave w owns rows [16w, 16w+16)
        acc_t acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[t][r] = S[(w * SB + MM::crow(r, lane)) * LDD + t * SB + fm];
#pragma unroll
        for (int kk = 0; kk < NB / 4; ++kk) {
            const T av = -Ps[(w * SB + fm) * LDD + 4 * kk + fk];
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = MM::mma(av, Ps[(t * SB + fm) * LDD + 4 * kk + fk], acc[t]);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) S[(w * SB + MM::crow(r, lane)) * LDD + t * SB + fm] = acc[t][r];
        __syncthreads();
    }

    int bad = 0;
#define NSGP_SUBPANEL(C0)                                                                             \
    {                                                                                                 \
        constexpr int B0 = C0 / SB;                                                                   \
        if (w == 0) { const int bd = factor_subpanel<T, C0>(S, rd, Qs + NB * LDD, lane); if (bad == 0) bad = bd; }   \
        else if (w == 1 && B0 > 0) invert_subblock<T>(S, rd, Dinv, B0 - 1, lane);  /* overlaps the factor */ \
        __syncthreads();                                                                              \
        constexpr int NT = (3 - B0) * (4 - B0) / 2;      /* lower tiles of the trailing block */      \
        for (int q = w; q < NT; q += 4) {                                                             \
            int ti = B0 + 1, tj = B0 + 1, c = q;                                                      \
            while (c > ti - (B0 + 1)) { c -= ti - B0; ++ti; }                                         \
            tj = B0 + 1 + c;                                                                          \
            acc_t acc;                                                                                \
            _Pragma("unroll") for (int r = 0; r < 4; ++r)                                             \
                acc[r] = S[(ti * SB + MM::crow(r, lane)) * LDD + tj * SB + fm];                       \
            _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) {                                        \
                const T av = -S[(ti * SB + fm) * LDD + C0 + 4 * kk + fk];                             \
                const T bv = S[(tj * SB + fm) * LDD + C0 + 4 * kk + fk];                              \
                acc = MM::mma(av, bv, acc);                                                           \
            }                                                                                         \
            _Pragma("unroll") for (int r = 0; r < 4; ++r)                                             \
                S[(ti * SB + MM::crow(r, lane)) * LDD + tj * SB + fm] = acc[r];                       \
        }                                                                                             \
        if (NT > 0) __syncthreads();                                                                  \
    }
    NSGP_SUBPANEL(0)
    NSGP_SUBPANEL(16)
    NSGP_SUBPANEL(32)
    NSGP_SUBPANEL(48)
#undef NSGP_SUBPANEL

    if (w == 1) invert_subblock<T>(S, rd, Dinv, 3, lane);
#pragma unroll
    for (int i = 0; i < SB; ++i) Xs[(w * SB + i) * LDD + lane] = xr[i];
    __syncthreads();
    if (pre && rows > 0) {                               // X -= Q P^T on this wave's strip (rows stay wave-private)
        acc_t acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[t][r] = Xs[(w * SB + MM::crow(r, lane)) * LDD + t * SB + fm];
#pragma unroll
        for (int kk = 0; kk < NB / 4; ++kk) {
            const T av = -Qs[(w * SB + fm) * LDD + 4 * kk + fk];
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = MM::mma(av, Ps[(t * SB + fm) * LDD + 4 * kk + fk], acc[t]);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) Xs[(w * SB + MM::crow(r, lane)) * LDD + t * SB + fm] = acc[t][r];
    }
    __syncthreads();

    if (blk == 0) {
        T* dst = wsL + (b * npanels + pj) * NB * NB;
#pragma unroll
        for (int i = 0; i < SB; ++i) dst[(w * SB + i) * NB + lane] = S[(w * SB + i) * LDD + lane];
        // the first panel initialises info (no memset launch); later panels record only the first failure
        if (tid == 0) {
            if (j0 == 0) info[b] = bad ? (int32_t)bad : 0;
            else if (bad && info[b] == 0) info[b] = (int32_t)(j0 + bad);
        }
    }
    if (rows == 0) return;                               // workgroup-uniform

    // slab strip of wave w: rows [16w, 16w+16).  X_cb = (A_cb - sum_{kb<cb} X_kb L[cb][kb]^T) Dinv_cb^T
    T* Xw = Xs + (w * SB) * LDD;
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
        acc_t acc;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = Xw[MM::crow(r, lane) * LDD + cb * SB + fm];
#pragma unroll
        for (int kb = 0; kb < cb; ++kb)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const T av = -Xw[fm * LDD + kb * SB + 4 * kk + fk];
                const T bv = S[(cb * SB + fm) * LDD + kb * SB + 4 * kk + fk];
                acc = MM::mma(av, bv, acc);
            }
        __syncthreads();                                 // (uniform) all reads of the strip issued before it is overwritten
#pragma unroll
        for (int r = 0; r < 4; ++r) Xw[MM::crow(r, lane) * LDD + cb * SB + fm] = acc[r];
        __syncthreads();
        acc_t y = {T(0), T(0), T(0), T(0)};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const T av = Xw[fm * LDD + cb * SB + 4 * kk + fk];
            const T bv = Dinv[(cb * SB + fm) * LDI + 4 * kk + fk];
            y = MM::mma(av, bv, y);
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) Xw[MM::crow(r, lane) * LDD + cb * SB + fm] = y[r];
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < SB; ++i)
        if (w * SB + i < rows) Ab[(r0 + w * SB + i) * lda + j0 + lane] = Xw[i * LDD + lane];
}

// LDS traffic written and re-read by ONE wave (other lanes): the LDS executes a wave's instructions in order, the fence
// only stops the compiler from moving them across (and drains lgkmcnt)
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

// C(16 x 16 tile at Ct) -= A(16 rows at Ar) B(16 rows at Br)^T over K = 64, everything in LDS with leading dimension LDD
template <typename T>
__device__ __forceinline__ void rank64_tile(T* Ct, const T* Ar, const T* Br, int lane) {
    typedef Mma16<T> MM;
    const int fm = lane & 15, fk = lane >> 4;
    typename MM::acc_t acc;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = Ct[MM::crow(r, lane) * LDD + fm];
#pragma unroll
    for (int kk = 0; kk < NB / 4; ++kk) acc = MM::mma(-Ar[fm * LDD + 4 * kk + fk], Br[fm * LDD + 4 * kk + fk], acc);
#pragma unroll
    for (int r = 0; r < 4; ++r) Ct[MM::crow(r, lane) * LDD + fm] = acc[r];
}

// NT such tiles, tile t at Ct + t * cstride with A rows at Ar + t * astride and B rows at Br + t * bstride (a stride of 0 shares
// the operand; the repeated reads of a shared operand are to the same LDS words).  A tile is a chain of 16 DEPENDENT
// MFMAs (one accumulator): alone it runs at the MFMA's latency, 1150 cycles (float32) for 512 cycles of matrix-core work, and
// these tiles were the longest item of every panel phase (tools/probes/potrf_stamps.py).  Here the NT chains advance together,
// so their MFMAs overlap; each tile's own arithmetic and its order are those of rank64_tile (bit-identical).
template <typename T, int NT>
__device__ __forceinline__ void rank64_tiles(T* Ct, int cstride, const T* Ar, int astride, const T* Br, int bstride, int lane) {
    typedef Mma16<T> MM;
    const int fm = lane & 15, fk = lane >> 4;
    // Every operand is requested before the first MFMA (left to itself the compiler fetches two k-steps at a time and waits
    // for them: an exposed LDS latency per 8 MFMAs, 4300 cycles for 2048 of matrix-core work), and the sign lives in the
    // accumulator -- (-C) + A B^T, negated back on the way out -- instead of a v_xor per A element: exact either way.
    typename MM::acc_t acc[NT];
    T av[NT][NB / 4], bv[NT][NB / 4];
    // four groups of four k-steps: group g + 1 is requested before group g's MFMAs are issued (the scheduling barriers keep
    // the compiler from sinking the reads back to their uses), so LDS latency hides behind 4 NT MFMAs
    auto fetch = [&](int g) __attribute__((always_inline)) {
#pragma unroll
        for (int kk = 4 * g; kk < 4 * g + 4; ++kk) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                av[t][kk] = Ar[t * astride + fm * LDD + 4 * kk + fk];
                bv[t][kk] = Br[t * bstride + fm * LDD + 4 * kk + fk];
            }
        }
    };
    fetch(0);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][r] = -Ct[t * cstride + MM::crow(r, lane) * LDD + fm];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        if (g < 3) fetch(g + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 4 * g; kk < 4 * g + 4; ++kk)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = MM::mma(av[t][kk], bv[t][kk], acc[t]);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) Ct[t * cstride + MM::crow(r, lane) * LDD + fm] = -acc[t][r];
}

// ONE tile with its K range cut into four accumulator chains (k = 4 kk + fk, chain kk mod 4), summed at the end: for the tile
// every wave updates alone before the first sub-panel can start (U0).  Not the summation order of rank64_tile -- the panel and
// the inverse's row-block workgroups both take this route for the same data, so they still agree bit for bit.
template <typename T>
__device__ __forceinline__ void rank64_tile_split(T* Ct, const T* Ar, const T* Br, int lane) {
    typedef Mma16<T> MM;
    const int fm = lane & 15, fk = lane >> 4;
    typename MM::acc_t acc[4];
    T av[NB / 4], bv[NB / 4];
#pragma unroll
    for (int kk = 0; kk < NB / 4; ++kk) { av[kk] = Ar[fm * LDD + 4 * kk + fk]; bv[kk] = Br[fm * LDD + 4 * kk + fk]; }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        acc[0][r] = -Ct[MM::crow(r, lane) * LDD + fm];
        acc[1][r] = T(0); acc[2][r] = T(0); acc[3][r] = T(0);
    }
    if constexpr (sizeof(T) == 4) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kk = 0; kk < NB / 4; ++kk) acc[kk & 3] = MM::mma(av[kk], bv[kk], acc[kk & 3]);
#pragma unroll
    for (int r = 0; r < 4; ++r) Ct[MM::crow(r, lane) * LDD + fm] = -((acc[0][r] + acc[1][r]) + (acc[2][r] + acc[3][r]));
}

// One column block CB of the blocked substitution  X L11^T = A21  for NS 16-row strips of the slab (strip s at Xw + s * sstride),
// by ONE wave:  X_cb = (A_cb - sum_{kb < cb} X_kb L[cb][kb]^T) Dinv_cb^T.  The strips' accumulator chains advance together
// (see rank64_tiles); per strip the arithmetic and its order do not depend on NS.
template <typename T, int CB, int NS = 1>
__device__ __forceinline__ void slab_subst_step(T* Xw, const T* S, const T* Dinv, int lane, int sstride = 0) {
    typedef Mma16<T> MM;
    const int fm = lane & 15, fk = lane >> 4;
    typename MM::acc_t acc[NS];
    constexpr int KS = CB * 4 > 0 ? CB * 4 : 1;
    T lv[KS], xv[NS][KS], dv[4];
#pragma unroll
    for (int q = 0; q < CB * 4; ++q) lv[q] = S[(CB * SB + fm) * LDD + 4 * q + fk];       // k = 16 kb + 4 kk + fk = 4 q + fk
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int q = 0; q < CB * 4; ++q) xv[s][q] = Xw[s * sstride + fm * LDD + 4 * q + fk];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) dv[kk] = Dinv[(CB * SB + fm) * LDI + 4 * kk + fk];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[s][r] = -Xw[s * sstride + MM::crow(r, lane) * LDD + CB * SB + fm];
    if constexpr (sizeof(T) == 4) __builtin_amdgcn_sched_barrier(0);   // float64: MFMA-bound, the compiler's own interleaving is better
#pragma unroll
    for (int q = 0; q < CB * 4; ++q)
#pragma unroll
        for (int s = 0; s < NS; ++s) acc[s] = MM::mma(xv[s][q], lv[q], acc[s]);
    wave_sync();
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int r = 0; r < 4; ++r) Xw[s * sstride + MM::crow(r, lane) * LDD + CB * SB + fm] = -acc[s][r];
    wave_sync();
    typename MM::acc_t y[NS];
    T xr[NS][4];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        y[s] = typename MM::acc_t{T(0), T(0), T(0), T(0)};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) xr[s][kk] = Xw[s * sstride + fm * LDD + CB * SB + 4 * kk + fk];
    }
    if constexpr (sizeof(T) == 4) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int s = 0; s < NS; ++s) y[s] = MM::mma(xr[s][kk], dv[kk], y[s]);
    wave_sync();
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int r = 0; r < 4; ++r) Xw[s * sstride + MM::crow(r, lane) * LDD + CB * SB + fm] = y[s][r];
    wave_sync();
}

// F0 work of waves 1..3 on the diagonal block: S -= P P^T on the tiles ON OR BELOW the diagonal of column tiles 1..3 (the
// factorisation never reads above it; six of the twelve tiles used to be updated for nothing) -- wave 1: (1..3, 1), wave 2:
// (2..3, 2), wave 3: (3, 3).  panel_body2 and prow_body both come through here: same arithmetic on the same data.
template <typename T>
__device__ __forceinline__ void diag_prev_update(T* S, const T* Ps, int w, int lane) {
    constexpr int RS = SB * LDD;
    if (w == 1) rank64_tiles<T, 3>(S + RS + SB, RS, Ps + RS, RS, Ps + RS, 0, lane);
    else if (w == 2) rank64_tiles<T, 2>(S + 2 * RS + 2 * SB, RS, Ps + 2 * RS, RS, Ps + 2 * RS, 0, lane);
    else if (w == 3) rank64_tile_split<T>(S + 3 * RS + 3 * SB, Ps + 3 * RS, Ps + 3 * RS, lane);
}

// Panel with the idle waves put to work.  While wave 0 factors a 16-column sub-panel in registers (2.7 us, 4 times per
// panel) the other three waves used to wait at the barrier; here they run everything that does not depend on the
// sub-panel being factored:
//   F0: the previous panel's rank-64 update of columns 16..63 of the diagonal block (columns 0..15, which sub-panel 0
//       needs, are updated by all waves first);
//   F1: that update on the slab (X -= Q P^T), wave 1 also inverts diagonal sub-block 0;
//   F2, F3: column blocks 0 and 1 of the slab substitution (wave 1 inverts sub-blocks 1, 2);
// so that after the last sub-panel only sub-block 3's inverse, column blocks 2 and 3 of the substitution and the store
// remain (5.5 -> ~2 us) -- same arithmetic in the same order per element as panel_body, bit-identical results.
// Work on a slab strip is wave-local (wave_sync); every phase ends in ONE barrier that all four waves reach.
template <typename T>
__device__ __forceinline__ void panel_body2(unsigned char* panel_smem, T* __restrict__ A, int64_t n, int64_t lda,
                                            int64_t sA, int64_t j0, T* __restrict__ wsL, int64_t npanels,
                                            int32_t* __restrict__ info, int64_t blk, int64_t b, bool pre) {
    typedef Mma16<T> MM;
    typedef typename MM::acc_t acc_t;
    T* S = reinterpret_cast<T*>(panel_smem);            // [64][LDD]   diagonal block -> L11
    T* Xs = S + NB * LDD;                               // [64][LDD]   this workgroup's slab
    T* Dinv = Xs + NB * LDD;                            // [4][16][LDI] inverses of the 16x16 diagonal sub-blocks
    T* rd = Dinv + 4 * SB * LDI;                        // [64] reciprocal pivots
    T* Ps = rd + NB;                                    // [64][LDD]   previous panel's L, diagonal-block rows
    T* Qs = Ps + NB * LDD;                              // [64][LDD]   previous panel's L, slab rows
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    T* Ab = A + b * sA;
    const int64_t pj = j0 / NB;
    const int64_t r0 = j0 + NB + blk * NB;
    const int rows = r0 >= n ? 0 : (int)((n - r0) < NB ? (n - r0) : NB);
    const int fm = lane & 15, fk = lane >> 4;
#ifdef NSGP_POTRF_STAMPS
    unsigned long long pst[16] = {}, wst[8] = {};
    const bool pst_on = tid == 0 && blk == 0 && b == 0;
    const bool wst_on = lane == 0 && blk == 0 && b == 0;
#endif
    NSGP_PSTAMP(0);
    {
        T dr[SB], xr[SB], pr[SB], qr[SB];                // loads first, LDS stores after (one exposed latency)
#pragma unroll
        for (int i = 0; i < SB; ++i) dr[i] = Ab[(j0 + w * SB + i) * lda + j0 + lane];
#pragma unroll
        for (int i = 0; i < SB; ++i) {                   // clamped row: unconditional loads (no branch per load)
            const int64_t rr = r0 + w * SB + i;
            xr[i] = Ab[(rr < n ? rr : n - 1) * lda + j0 + lane];
        }
        if (pre) {
#pragma unroll
            for (int i = 0; i < SB; ++i) {
                const int64_t rr = r0 + w * SB + i;
                pr[i] = Ab[(j0 + w * SB + i) * lda + j0 - NB + lane];
                qr[i] = Ab[(rr < n ? rr : n - 1) * lda + j0 - NB + lane];
            }
        }
#pragma unroll
        for (int i = 0; i < SB; ++i) { keep(dr[i]); keep(xr[i]); }
        if (pre) {
#pragma unroll
            for (int i = 0; i < SB; ++i) { keep(pr[i]); keep(qr[i]); }
        }
#pragma unroll
        for (int i = 0; i < SB; ++i) {
            S[(w * SB + i) * LDD + lane] = dr[i];
            Xs[(w * SB + i) * LDD + lane] = (w * SB + i) < rows ? xr[i] : T(0);
            if (pre) {
                Ps[(w * SB + i) * LDD + lane] = pr[i];
                Qs[(w * SB + i) * LDD + lane] = (w * SB + i) < rows ? qr[i] : T(0);
            }
        }
    }
    __syncthreads();
    NSGP_PSTAMP(1);
    if (pre) {                                           // U0: columns 0..15 of S -= P P^T (what sub-panel 0 needs)
        rank64_tile_split<T>(S + (w * SB) * LDD, Ps + (w * SB) * LDD, Ps, lane);
        __syncthreads();
    }
    NSGP_PSTAMP(2);
    const bool slab = rows > 0;
    int bad = 0;
    // ---- F0 ----
    if (w == 0) { const int bd = factor_subpanel<T, 0>(S, rd, Qs + NB * LDD, lane); if (bad == 0) bad = bd; NSGP_PSTAMP(3); }
    else if (pre) {
        diag_prev_update<T>(S, Ps, w, lane);
        // ... and, behind the short lists, slab tiles that wait for nothing: X -= Q P^T, (strip, column tile) = (0, 2), (1, 2)
        if (w == 3 && slab) rank64_tiles<T, 2>(Xs + 2 * SB, SB * LDD, Qs, SB * LDD, Ps + (2 * SB) * LDD, 0, lane);
    }
    NSGP_WSTAMP(0);
    __syncthreads();
#define NSGP_TRAIL(C0)                                                                                \
    {                                                                                                 \
        constexpr int B0 = C0 / SB;                                                                   \
        constexpr int NT = (3 - B0) * (4 - B0) / 2;      /* lower tiles of the trailing block */      \
        for (int q = w; q < NT; q += 4) {                                                             \
            int ti = B0 + 1, tj = B0 + 1, c = q;                                                      \
            while (c > ti - (B0 + 1)) { c -= ti - B0; ++ti; }                                         \
            tj = B0 + 1 + c;                                                                          \
            acc_t acc;                                                                                \
            _Pragma("unroll") for (int r = 0; r < 4; ++r)                                             \
                acc[r] = S[(ti * SB + MM::crow(r, lane)) * LDD + tj * SB + fm];                       \
            _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) {                                        \
                const T av = -S[(ti * SB + fm) * LDD + C0 + 4 * kk + fk];                             \
                const T bv = S[(tj * SB + fm) * LDD + C0 + 4 * kk + fk];                              \
                acc = MM::mma(av, bv, acc);                                                           \
            }                                                                                         \
            _Pragma("unroll") for (int r = 0; r < 4; ++r)                                             \
                S[(ti * SB + MM::crow(r, lane)) * LDD + tj * SB + fm] = acc[r];                       \
        }                                                                                             \
        NSGP_WSTAMP(1 + 2 * B0);                                                                      \
        if (NT > 0) __syncthreads();                                                                  \
    }
    NSGP_PSTAMP(4);
    NSGP_TRAIL(0)
    NSGP_PSTAMP(5);
    // ---- F1 ----
    if (w == 0) { const int bd = factor_subpanel<T, 16>(S, rd, Qs + NB * LDD, lane); if (bad == 0) bad = bd; NSGP_PSTAMP(6); }
    else {
        // X -= Q P^T, 16 (strip, column tile) tasks over F0 .. F2, at most three to a wave and phase (float64: a tile is 1024
        // cycles of matrix-core time, a sub-panel factorisation 4900): column tiles 0 and 1 now (the substitution needs them
        // first), tile 2 in F0 / F2 (wave 3 / wave 1), tile 3 in F2
        if (w == 1) {                                    // strip 3 of column tiles 0 and 1 (shares Q's rows)
            invert_subblock<T>(S, rd, Dinv, 0, lane);
            if (pre && slab) rank64_tiles<T, 2>(Xs + (3 * SB) * LDD, SB, Qs + (3 * SB) * LDD, 0, Ps, SB * LDD, lane);
        } else if (pre && slab) {                        // strips 0..2 of column tile w - 2
            rank64_tiles<T, 3>(Xs + (w - 2) * SB, SB * LDD, Qs, SB * LDD, Ps + ((w - 2) * SB) * LDD, 0, lane);
        }
    }
    NSGP_WSTAMP(2);
    __syncthreads();
    NSGP_PSTAMP(7);
    NSGP_TRAIL(16)
    NSGP_PSTAMP(8);
    // ---- F2 ----
    if (w == 0) { const int bd = factor_subpanel<T, 32>(S, rd, Qs + NB * LDD, lane); if (bad == 0) bad = bd; NSGP_PSTAMP(9); }
    else {
        if (w == 1) {
            invert_subblock<T>(S, rd, Dinv, 1, lane);
            if (pre && slab) rank64_tiles<T, 2>(Xs + (2 * SB) * LDD + 2 * SB, SB * LDD, Qs + (2 * SB) * LDD, SB * LDD,
                                                Ps + (2 * SB) * LDD, 0, lane);
        } else if (slab) {                               // waves 2, 3: strips {0, 2} and {1, 3}
            if (pre) rank64_tiles<T, 2>(Xs + ((w - 2) * SB) * LDD + 3 * SB, 2 * SB * LDD, Qs + ((w - 2) * SB) * LDD, 2 * SB * LDD,
                                        Ps + (3 * SB) * LDD, 0, lane);
            slab_subst_step<T, 0, 2>(Xs + ((w - 2) * SB) * LDD, S, Dinv, lane, 2 * SB * LDD);
        }
    }
    NSGP_WSTAMP(4);
    __syncthreads();
    NSGP_PSTAMP(10);
    NSGP_TRAIL(32)
    NSGP_PSTAMP(11);
    // ---- F3 ----
    if (w == 0) { const int bd = factor_subpanel<T, 48>(S, rd, Qs + NB * LDD, lane); if (bad == 0) bad = bd; NSGP_PSTAMP(12); }
    else if (w == 1) invert_subblock<T>(S, rd, Dinv, 2, lane);
    else if (slab) {
        slab_subst_step<T, 1, 2>(Xs + ((w - 2) * SB) * LDD, S, Dinv, lane, 2 * SB * LDD);
    }
    NSGP_WSTAMP(6);
    __syncthreads();
    NSGP_PSTAMP(13);
#undef NSGP_TRAIL
    // ---- after the last sub-panel ----
    if (w == 1) invert_subblock<T>(S, rd, Dinv, 3, lane);
    else if (w >= 2 && slab) {
        slab_subst_step<T, 2, 2>(Xs + ((w - 2) * SB) * LDD, S, Dinv, lane, 2 * SB * LDD);
    }
    if (blk == 0) {
        T* dst = wsL + (b * npanels + pj) * NB * NB;
#pragma unroll
        for (int i = 0; i < SB; ++i) dst[(w * SB + i) * NB + lane] = S[(w * SB + i) * LDD + lane];
        // the first panel initialises info (no memset launch); later panels record only the first failure
        if (tid == 0) {
            if (j0 == 0) info[b] = bad ? (int32_t)bad : 0;
            else if (bad && info[b] == 0) info[b] = (int32_t)(j0 + bad);
        }
    }
    NSGP_WSTAMP(7);
    __syncthreads();
    NSGP_PSTAMP(14);
    if (slab) {                                          // workgroup-uniform
        T* Xw = Xs + (w * SB) * LDD;                     // last column block: every wave its own strip
        slab_subst_step<T, 3>(Xw, S, Dinv, lane);
#pragma unroll
        for (int i = 0; i < SB; ++i)
            if (w * SB + i < rows) Ab[(r0 + w * SB + i) * lda + j0 + lane] = Xw[i * LDD + lane];
    }
#ifdef NSGP_POTRF_STAMPS
    if (pst_on && nsgp_pstamp_buf && (unsigned long long)pj < nsgp_pstamp_cap) {
        pst[15] = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < 16; ++i) nsgp_pstamp_buf[pj * 64 + i] = pst[i];
    }
    if (wst_on && nsgp_pstamp_buf && (unsigned long long)pj < nsgp_pstamp_cap)
        for (int i = 0; i < 8; ++i) nsgp_pstamp_buf[pj * 64 + 16 + 8 * w + i] = wst[i];
#endif
}

// Rank-64 trailing update  C -= L21 L21^T  on the lower 64x64 tiles of a (rows x wcols) region, one tile per
// workgroup, ONE K step: both operand blocks and the C tile are requested up front (a single memory
// latency), then 16 MFMA k-steps per 16x16 tile.  The generic GEMM pays a load latency per BK=16 K-tile
// (13 us for this shape); this kernel takes ~5 us, which matters because it sits on the serial panel chain.
template <typename T>
__device__ __forceinline__ void syrk_body(unsigned char* panel_smem, T* __restrict__ A, int64_t n, int64_t lda,
                                          int64_t sA, int64_t j0, int64_t base, int64_t wcols, int64_t lin, int64_t b) {
    // j0: first of the 64 K columns (the factored panel); base: first row / column of the updated block
    typedef Mma16<T> MM;
    typedef typename MM::acc_t acc_t;
    T* As = reinterpret_cast<T*>(panel_smem);           // [64][LDD]  L21 rows of the tile's row block
    T* Bs = As + NB * LDD;                              // [64][LDD]  L21 rows of the tile's column block
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int fm = lane & 15, fk = lane >> 4;
    T* Ab = A + b * sA;
    const int64_t tn = (wcols + NB - 1) / NB;           // tile columns
    // compact enumeration of the lower tiles: rows 0..tn-1 form a triangle, the rest are full
    const int64_t tri = tn * (tn + 1) / 2;
    int64_t ti, tj;
    if (lin < tri) {
        int64_t r = (int64_t)((sqrtf(8.0f * (float)lin + 1.0f) - 1.0f) * 0.5f);
        while (r * (r + 1) / 2 > lin) --r;
        while ((r + 1) * (r + 2) / 2 <= lin) ++r;
        ti = r; tj = lin - r * (r + 1) / 2;
    } else {
        const int64_t q = lin - tri;
        ti = tn + q / tn; tj = q % tn;
    }
    const int64_t r0 = base + ti * NB, c0 = base + tj * NB;
    const int64_t clim = base + wcols < n ? base + wcols : n;
    // C tile straight into the accumulator layout: wave w owns rows [16w, 16w+16), 4 column tiles.
    // Every global load of the kernel is issued before anything waits (clamped indices, masks applied later).
    acc_t acc[4];
    T ar[SB], br[SB];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t row = r0 + w * SB + MM::crow(r, lane), col = c0 + t * SB + fm;
            acc[t][r] = Ab[(row < n ? row : n - 1) * lda + (col < clim ? col : clim - 1)];
        }
#pragma unroll
    for (int i = 0; i < SB; ++i) {
        const int64_t ra = r0 + w * SB + i, rb = c0 + w * SB + i;
        ar[i] = Ab[(ra < n ? ra : n - 1) * lda + j0 + lane];
        br[i] = Ab[(rb < n ? rb : n - 1) * lda + j0 + lane];          // == ar[i] on diagonal tiles (L2 hit)
    }
#pragma unroll
    for (int i = 0; i < SB; ++i) { keep(ar[i]); keep(br[i]); }
#pragma unroll
    for (int i = 0; i < SB; ++i) {
        const int64_t ra = r0 + w * SB + i, rb = c0 + w * SB + i;
        As[(w * SB + i) * LDD + lane] = ra < n ? ar[i] : T(0);
        Bs[(w * SB + i) * LDD + lane] = rb < n ? br[i] : T(0);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            T v = acc[t][r];
            keep(v);
            const int64_t row = r0 + w * SB + MM::crow(r, lane), col = c0 + t * SB + fm;
            acc[t][r] = (row < n && col < clim) ? v : T(0);
        }
    __syncthreads();
    const T* Bq = Bs;
#pragma unroll
    for (int kk = 0; kk < NB / 4; ++kk) {
        const T av = -As[(w * SB + fm) * LDD + 4 * kk + fk];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = MM::mma(av, Bq[(t * SB + fm) * LDD + 4 * kk + fk], acc[t]);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t row = r0 + w * SB + MM::crow(r, lane), col = c0 + t * SB + fm;
            if (row < n && col < clim) Ab[row * lda + col] = acc[t][r];
        }
}

// One launch per panel.  Workgroups [0, nslab): panel j0 (factor + slabs), applying the previous panel's rank-64
// update to their own block column first when `pre`.  Workgroups [nslab, ...): the previous panel's rank-64 update
// of the rest of the trailing block (columns [j0+64, j0+64+wcols)).  The two groups touch disjoint data, so the
// serial chain per panel is ONE kernel (the bulk update overlaps the next factorisation: look-ahead of depth 1).
template <typename T>
__global__ __launch_bounds__(256) void potrf_step_kernel(T* __restrict__ A, int64_t n, int64_t lda, int64_t sA,
                                                         int64_t j0, T* __restrict__ wsL, int64_t npanels,
                                                         int32_t* __restrict__ info, int64_t nslab, int pre,
                                                         int64_t wcols) {
    extern __shared__ __attribute__((aligned(16))) unsigned char panel_smem[];
    const int64_t lin = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;     // matrix fastest: see potrf_inv_step_kernel
    const int64_t blk = lin / gridDim.y, b = lin % gridDim.y;
    if (blk < nslab) {
        // bit 1 of `pre`: the round-1 panel (NSGP_POTRF_PANEL=1), kept for A/B timing
        if (pre & 2) panel_body<T>(panel_smem, A, n, lda, sA, j0, wsL, npanels, info, blk, b, (pre & 1) != 0);
        else panel_body2<T>(panel_smem, A, n, lda, sA, j0, wsL, npanels, info, blk, b, (pre & 1) != 0);
    }
    else syrk_body<T>(panel_smem, A, n, lda, sA, j0 - NB, j0 + NB, wcols, blk - nslab, b);
}

// ---- the inverse accumulated inside the factorisation's launches (nsgp_potrf_trtri, n a multiple of 64, one level) -----------
// W = L^-1 is the identity carried through the same block eliminations:  after panel j,
//     W[j][0..j]  <-  L_jj^-1 W[j][0..j]                (row block j: a 64-row triangular solve per 64-column chunk)
//     W[i][0..j]  -=  L[i][j] W[j][0..j]   for i > j    (rank-64 update: tiles with one K step, like the trailing update)
// The row-block solves run as extra workgroups of panel j's launch (they factor the diagonal block redundantly like the slab
// workgroups, so they add no latency), the updates as extra tiles of panel j + 1's launch, next to the trailing update of
// the factor -- on CUs those launches leave idle.  The eight latency-bound launches of the recursive inverse (trtri: 162 us
// for 3 x 1024^2) disappear from the DSVI whitening chain.

// C(16 x 16 tile at Ct) -= A(16 rows at Ar, K contiguous) B(64 x 16 at Bk, K-major: B(k, n) = Bk[k * LDD + n])
template <typename T>
__device__ __forceinline__ void rank64_tile_kn(T* Ct, const T* Ar, const T* Bk, int lane) {
    typedef Mma16<T> MM;
    const int fm = lane & 15, fk = lane >> 4;
    typename MM::acc_t acc;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = Ct[MM::crow(r, lane) * LDD + fm];
#pragma unroll
    for (int kk = 0; kk < NB / 4; ++kk) acc = MM::mma(-Ar[fm * LDD + 4 * kk + fk], Bk[(4 * kk + fk) * LDD + fm], acc);
#pragma unroll
    for (int r = 0; r < 4; ++r) Ct[MM::crow(r, lane) * LDD + fm] = acc[r];
}

// NT such tiles that share A: tile t at Ct + 16 t against B columns Bk + 16 t (accumulator chains interleaved, see rank64_tiles)
template <typename T, int NT>
__device__ __forceinline__ void rank64_tiles_kn(T* Ct, const T* Ar, const T* Bk, int lane) {
    typedef Mma16<T> MM;
    const int fm = lane & 15, fk = lane >> 4;
    typename MM::acc_t acc[NT];
    T av[NB / 4], bv[NT][NB / 4];
    auto fetch = [&](int g) __attribute__((always_inline)) {       // as rank64_tiles: one group of k-steps ahead
#pragma unroll
        for (int kk = 4 * g; kk < 4 * g + 4; ++kk) {
            av[kk] = Ar[fm * LDD + 4 * kk + fk];
#pragma unroll
            for (int t = 0; t < NT; ++t) bv[t][kk] = Bk[t * SB + (4 * kk + fk) * LDD + fm];
        }
    };
    fetch(0);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][r] = -Ct[t * SB + MM::crow(r, lane) * LDD + fm];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        if (g < 3) fetch(g + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 4 * g; kk < 4 * g + 4; ++kk)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = MM::mma(av[kk], bv[t][kk], acc[t]);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) Ct[t * SB + MM::crow(r, lane) * LDD + fm] = -acc[t][r];
}

// Row block RB of the blocked substitution  L11 Y = R  for NS 16-COLUMN strips of a 64 x 64 block (strip s at Rc + s * cstride),
// by ONE wave:  Y_rb = Dinv_rb (R_rb - sum_{kb < rb} L[rb][kb] Y_kb)      (Y overwrites R; Rc points at the first strip's first
// column).  The strips' accumulator chains advance together; per strip the arithmetic and its order do not depend on NS.
template <typename T, int RB, int NS = 1>
__device__ __forceinline__ void prow_subst_step(T* Rc, const T* S, const T* Dinv, int lane, int cstride = 0) {
    typedef Mma16<T> MM;
    const int fm = lane & 15, fk = lane >> 4;
    typename MM::acc_t acc[NS];
    constexpr int KS = RB * 4 > 0 ? RB * 4 : 1;
    T lv[KS], rv[NS][KS], dv[4];
#pragma unroll
    for (int q = 0; q < RB * 4; ++q) lv[q] = S[(RB * SB + fm) * LDD + 4 * q + fk];       // k = 16 kb + 4 kk + fk = 4 q + fk
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int q = 0; q < RB * 4; ++q) rv[s][q] = Rc[s * cstride + (4 * q + fk) * LDD + fm];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) dv[kk] = Dinv[(RB * SB + fm) * LDI + 4 * kk + fk];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[s][r] = -Rc[s * cstride + (RB * SB + MM::crow(r, lane)) * LDD + fm];
    if constexpr (sizeof(T) == 4) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < RB * 4; ++q)
#pragma unroll
        for (int s = 0; s < NS; ++s) acc[s] = MM::mma(lv[q], rv[s][q], acc[s]);
    wave_sync();
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int r = 0; r < 4; ++r) Rc[s * cstride + (RB * SB + MM::crow(r, lane)) * LDD + fm] = -acc[s][r];
    wave_sync();
    typename MM::acc_t y[NS];
    T rr[NS][4];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        y[s] = typename MM::acc_t{T(0), T(0), T(0), T(0)};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) rr[s][kk] = Rc[s * cstride + (RB * SB + 4 * kk + fk) * LDD + fm];
    }
    if constexpr (sizeof(T) == 4) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int s = 0; s < NS; ++s) y[s] = MM::mma(dv[kk], rr[s][kk], y[s]);
    wave_sync();
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int r = 0; r < 4; ++r) Rc[s * cstride + (RB * SB + MM::crow(r, lane)) * LDD + fm] = y[s][r];
    wave_sync();
}

// Row block j of the inverse, 64-column chunk c (c0 = 64 c <= j0):  W[j][c] <- L_jj^-1 (W[j][c] - L[j][j-1] W[j-1][c]).
// Same phases as panel_body2 (the diagonal block is factored redundantly; the idle waves do the chunk's work).
template <typename T>
__device__ __forceinline__ void prow_body(unsigned char* panel_smem, const T* __restrict__ A, int64_t n, int64_t lda,
                                          int64_t sA, int64_t j0, T* __restrict__ X, int64_t ldx, int64_t sX, int64_t c,
                                          int64_t b, bool pre, float* __restrict__ X32) {
    // X32 != null (float64 chain of a float32 model): a float32 copy of W (same leading dimension / batch stride in
    // elements) is written along with it -- the cast pass the layers would otherwise launch
    typedef Mma16<T> MM;
    typedef typename MM::acc_t acc_t;
    T* S = reinterpret_cast<T*>(panel_smem);            // [64][LDD]   diagonal block -> L11
    T* Rs = S + NB * LDD;                               // [64][LDD]   the chunk of W's row block
    T* Dinv = Rs + NB * LDD;                            // [4][16][LDI]
    T* rd = Dinv + 4 * SB * LDI;                        // [64]
    T* Ps = rd + NB;                                    // [64][LDD]   L[j][j-1]
    T* Qs = Ps + NB * LDD;                              // [64][LDD]   W[j-1][c]   (K x N)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const T* Ab = A + b * sA;
    T* Xb = X + b * sX;
    const int64_t c0 = c * NB;
    const bool diag = c0 == j0;                          // the chunk that starts as the identity
    const bool upd = pre && !diag;                       // chunks left of the diagonal carry the previous panel's update
    const bool first = c0 + NB == j0;                    // chunk j - 1: no earlier update has written it
    const int fm = lane & 15, fk = lane >> 4;
    if (diag) {                                          // zero the rest of these rows (W is lower triangular); off the
        for (int i = w; i < NB; i += 4)                  // critical path: this workgroup has no chunk to load
            for (int64_t cc = j0 + NB + lane; cc < n; cc += 64) {
                Xb[(j0 + i) * ldx + cc] = T(0);
                if (X32) X32[b * sX + (j0 + i) * ldx + cc] = 0.f;
            }
    }
    {
        T dr[SB], xr[SB], pr[SB], qr[SB];
#pragma unroll
        for (int i = 0; i < SB; ++i) dr[i] = Ab[(j0 + w * SB + i) * lda + j0 + lane];
#pragma unroll
        for (int i = 0; i < SB; ++i)            // identity on the diagonal; zero where no update has reached yet (W is never
            xr[i] = diag ? ((w * SB + i) == lane ? T(1) : T(0))          // initialised: chunk j - 1 is touched for the first
                         : (first ? T(0) : Xb[(j0 + w * SB + i) * ldx + c0 + lane]);   // time by panel j - 1's update)
        if (pre) {
#pragma unroll
            for (int i = 0; i < SB; ++i) pr[i] = Ab[(j0 + w * SB + i) * lda + j0 - NB + lane];
        }
        if (upd) {
#pragma unroll
            for (int i = 0; i < SB; ++i) qr[i] = Xb[(j0 - NB + w * SB + i) * ldx + c0 + lane];
        }
#pragma unroll
        for (int i = 0; i < SB; ++i) { keep(dr[i]); keep(xr[i]); }
        if (pre) {
#pragma unroll
            for (int i = 0; i < SB; ++i) keep(pr[i]);
        }
        if (upd) {
#pragma unroll
            for (int i = 0; i < SB; ++i) keep(qr[i]);
        }
#pragma unroll
        for (int i = 0; i < SB; ++i) {
            S[(w * SB + i) * LDD + lane] = dr[i];
            Rs[(w * SB + i) * LDD + lane] = xr[i];
            if (pre) Ps[(w * SB + i) * LDD + lane] = pr[i];
            if (upd) Qs[(w * SB + i) * LDD + lane] = qr[i];
        }
    }
    __syncthreads();
    if (pre) {                                           // U0: columns 0..15 of S -= P P^T
        rank64_tile_split<T>(S + (w * SB) * LDD, Ps + (w * SB) * LDD, Ps, lane);
        __syncthreads();
    }
    // ---- F0 ----
    if (w == 0) (void)factor_subpanel<T, 0>(S, rd, Qs + NB * LDD, lane);
    else if (pre) {
        diag_prev_update<T>(S, Ps, w, lane);
    }
    __syncthreads();
#define NSGP_TRAIL(C0)                                                                                \
    {                                                                                                 \
        constexpr int B0 = C0 / SB;                                                                   \
        constexpr int NT = (3 - B0) * (4 - B0) / 2;                                                   \
        for (int q = w; q < NT; q += 4) {                                                             \
            int ti = B0 + 1, tj = B0 + 1, cq = q;                                                     \
            while (cq > ti - (B0 + 1)) { cq -= ti - B0; ++ti; }                                       \
            tj = B0 + 1 + cq;                                                                         \
            acc_t acc;                                                                                \
            _Pragma("unroll") for (int r = 0; r < 4; ++r)                                             \
                acc[r] = S[(ti * SB + MM::crow(r, lane)) * LDD + tj * SB + fm];                       \
            _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) {                                        \
                const T av = -S[(ti * SB + fm) * LDD + C0 + 4 * kk + fk];                             \
                const T bv = S[(tj * SB + fm) * LDD + C0 + 4 * kk + fk];                              \
                acc = MM::mma(av, bv, acc);                                                           \
            }                                                                                         \
            _Pragma("unroll") for (int r = 0; r < 4; ++r)                                             \
                S[(ti * SB + MM::crow(r, lane)) * LDD + tj * SB + fm] = acc[r];                       \
        }                                                                                             \
        if (NT > 0) __syncthreads();                                                                  \
    }
    NSGP_TRAIL(0)
    // R -= L[j][j-1] W[j-1][c]: tiles (row block rb, column tiles t, t + 1, ...) share the rows of L
    auto rtiles2 = [&](int rb, int t) __attribute__((always_inline)) {
        rank64_tiles_kn<T, 2>(Rs + (rb * SB) * LDD + t * SB, Ps + (rb * SB) * LDD, Qs + t * SB, lane);
    };
    // ---- F1 ----  (the substitution walks row blocks 0, 1, 2, 3 of EVERY column strip: all column tiles of row blocks 0
    // and 1 first)
    if (w == 0) (void)factor_subpanel<T, 16>(S, rd, Qs + NB * LDD, lane);
    else if (w == 1) {
        invert_subblock<T>(S, rd, Dinv, 0, lane);
        if (upd) rtiles2(2, 0);
    } else if (upd) {
        rank64_tiles_kn<T, 4>(Rs + ((w - 2) * SB) * LDD, Ps + ((w - 2) * SB) * LDD, Qs, lane);
    }
    __syncthreads();
    NSGP_TRAIL(16)
    // ---- F2 ----
    if (w == 0) (void)factor_subpanel<T, 32>(S, rd, Qs + NB * LDD, lane);
    else if (w == 1) {
        invert_subblock<T>(S, rd, Dinv, 1, lane);
        if (upd) rtiles2(2, 2);
    } else {
        if (upd) rtiles2(3, 2 * (w - 2));
        prow_subst_step<T, 0, 2>(Rs + (w - 2) * SB, S, Dinv, lane, 2 * SB);          // column strips {0, 2} and {1, 3}
    }
    __syncthreads();
    NSGP_TRAIL(32)
    // ---- F3 ----
    if (w == 0) (void)factor_subpanel<T, 48>(S, rd, Qs + NB * LDD, lane);
    else if (w == 1) invert_subblock<T>(S, rd, Dinv, 2, lane);
    else {
        prow_subst_step<T, 1, 2>(Rs + (w - 2) * SB, S, Dinv, lane, 2 * SB);
    }
    __syncthreads();
#undef NSGP_TRAIL
    if (w == 1) invert_subblock<T>(S, rd, Dinv, 3, lane);
    else if (w >= 2) {
        prow_subst_step<T, 2, 2>(Rs + (w - 2) * SB, S, Dinv, lane, 2 * SB);
    }
    __syncthreads();
    prow_subst_step<T, 3>(Rs + w * SB, S, Dinv, lane);                    // last row block: every wave its column strip
    __syncthreads();
#pragma unroll
    for (int i = 0; i < SB; ++i) {
        const T v = Rs[(w * SB + i) * LDD + lane];
        Xb[(j0 + w * SB + i) * ldx + c0 + lane] = v;
        if (X32) X32[b * sX + (j0 + w * SB + i) * ldx + c0 + lane] = (float)v;
    }
}

// W[i][c] -= L[i][j-1] W[j-1][c]  for one 64 x 64 tile (row block i >= j + 1 given by r0, chunk c0 <= j0 - 64); kp = j0 - 64.
template <typename T>
__device__ __forceinline__ void pupd_body(unsigned char* panel_smem, const T* __restrict__ A, int64_t lda, int64_t sA,
                                          T* __restrict__ X, int64_t ldx, int64_t sX, int64_t kp, int64_t r0, int64_t c0,
                                          int64_t b) {
    typedef Mma16<T> MM;
    typedef typename MM::acc_t acc_t;
    T* As = reinterpret_cast<T*>(panel_smem);           // [64][LDD]  L[i][j-1]
    T* Bs = As + NB * LDD;                              // [64][LDD]  W[j-1][c]  (K x N)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int fm = lane & 15, fk = lane >> 4;
    const T* Ab = A + b * sA;
    T* Xb = X + b * sX;
    acc_t acc[4];
    T ar[SB], br[SB];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)              // chunk j - 1 is written here for the first time (W starts as the identity)
            acc[t][r] = c0 == kp ? T(0) : Xb[(r0 + w * SB + MM::crow(r, lane)) * ldx + c0 + t * SB + fm];
#pragma unroll
    for (int i = 0; i < SB; ++i) {
        ar[i] = Ab[(r0 + w * SB + i) * lda + kp + lane];
        br[i] = Xb[(kp + w * SB + i) * ldx + c0 + lane];
    }
#pragma unroll
    for (int i = 0; i < SB; ++i) { keep(ar[i]); keep(br[i]); }
#pragma unroll
    for (int i = 0; i < SB; ++i) {
        As[(w * SB + i) * LDD + lane] = ar[i];
        Bs[(w * SB + i) * LDD + lane] = br[i];
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) { T v = acc[t][r]; keep(v); acc[t][r] = v; }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < NB / 4; ++kk) {
        const T av = -As[(w * SB + fm) * LDD + 4 * kk + fk];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = MM::mma(av, Bs[(4 * kk + fk) * LDD + t * SB + fm], acc[t]);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) Xb[(r0 + w * SB + MM::crow(r, lane)) * ldx + c0 + t * SB + fm] = acc[t][r];
}

// One launch per panel of the factor-and-invert chain.  blockIdx.x ranges: [0, nslab) slab workgroups (factor + L21),
// [.., + nprow) chunks of W's row block j, [.., + nsyrk) the previous panel's trailing update of the factor, the rest: the
// previous panel's update of W's rows below j (tiles (row block, chunk), chunk fastest).
template <typename T>
__global__ __launch_bounds__(256) void potrf_inv_step_kernel(T* __restrict__ A, int64_t n, int64_t lda, int64_t sA,
                                                             int64_t j0, T* __restrict__ wsL, int64_t npanels,
                                                             int32_t* __restrict__ info, T* __restrict__ X, int64_t ldx,
                                                             int64_t sX, int64_t nslab, int64_t nprow, int64_t nsyrk,
                                                             int pre, int64_t wcols, float* __restrict__ X32) {
    extern __shared__ __attribute__((aligned(16))) unsigned char panel_smem[];
    // Workgroups are dispatched x fastest, then y.  With the matrix on y, a batched chain whose launch needs more than one
    // round (3 x 1024^2 in the DSVI step: up to 405 workgroups at ONE per CU, 142 KB of LDS in float64) started the panel
    // workgroups of matrices 1, 2 -- the serial chain -- behind all of matrix 0's update tiles: 34 us launches against 20.
    // Linear id -> (work item, matrix) with the matrix fastest: every matrix's panel / slab / row-block workgroups go first.
    const int64_t lin = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
    int64_t blk = lin / gridDim.y;
    const int64_t b = lin % gridDim.y;
    if (blk < nslab) { panel_body2<T>(panel_smem, A, n, lda, sA, j0, wsL, npanels, info, blk, b, pre != 0); return; }
    blk -= nslab;
    if (blk < nprow) { prow_body<T>(panel_smem, A, n, lda, sA, j0, X, ldx, sX, blk, b, pre != 0, X32); return; }
    blk -= nprow;
    if (blk < nsyrk) { syrk_body<T>(panel_smem, A, n, lda, sA, j0 - NB, j0 + NB, wcols, blk, b); return; }
    blk -= nsyrk;
    const int64_t nchunk = j0 / NB;                      // chunks 0 .. j - 1 carry the update of panel j - 1
    pupd_body<T>(panel_smem, A, lda, sA, X, ldx, sX, j0 - NB, j0 + NB + (blk / nchunk) * NB, (blk % nchunk) * NB, b);
}

// LAST, ragged panel (nb < 64, nothing below it), factored in LDS by one wave.
template <typename T>
__global__ __launch_bounds__(64) void potrf_tail_kernel(T* __restrict__ A, int64_t n, int64_t lda, int64_t sA,
                                                        int64_t j0, T* __restrict__ wsL, int64_t npanels,
                                                        int32_t* __restrict__ info, int pre) {
    __shared__ T Ls[NB * LDD];
    const int lane = threadIdx.x;
    const int64_t b = blockIdx.y;
    T* Ab = A + b * sA;
    const int64_t pj = j0 / NB;
    const int nb = (int)(n - j0);
    if (lane < nb)
        for (int j = 0; j <= lane; ++j) {
            T v = Ab[(j0 + lane) * lda + j0 + j];
            if (pre)                                     // pending rank-64 update of the previous panel
                for (int k = 0; k < NB; ++k) v -= Ab[(j0 + lane) * lda + j0 - NB + k] * Ab[(j0 + j) * lda + j0 - NB + k];
            Ls[lane * LDD + j] = v;
        }
    const int bad = factor_lds(Ls, nb, lane);
    T* dst = wsL + (b * npanels + pj) * NB * NB;
    if (lane < nb)
        for (int j = 0; j <= lane; ++j) dst[lane * NB + j] = Ls[lane * LDD + j];
    if (lane == 0) {
        if (j0 == 0) info[b] = bad ? (int32_t)bad : 0;
        else if (bad && info[b] == 0) info[b] = (int32_t)(j0 + bad);
    }
}

// Finalisation: one workgroup per 64x64 tile on or above the diagonal.  Diagonal tiles receive their factor
// from the side buffer (strict upper part zeroed), tiles above the diagonal are zeroed (L is lower).
template <typename T>
__global__ __launch_bounds__(256) void potrf_finalize_kernel(T* __restrict__ A, int64_t n, int64_t lda, int64_t sA,
                                                             const T* __restrict__ wsL, int64_t npanels) {
    const int64_t ti = blockIdx.y, tj = blockIdx.x, b = blockIdx.z;
    if (tj < ti) return;
    T* Ab = A + b * sA;
    const int64_t r0 = ti * NB, c0 = tj * NB;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t c = c0 + lane;
    if (c >= n) return;
    const T* src = wsL + (b * npanels + ti) * NB * NB;
    for (int i = w; i < NB; i += 4) {
        const int64_t r = r0 + i;
        if (r >= n) break;
        Ab[r * lda + c] = (tj == ti && lane <= i) ? src[i * NB + lane] : T(0);
    }
}

template <typename T>
int potrf_impl(T* A, int64_t n, int64_t lda, int64_t sA, int64_t batch, int32_t* info, void* ws, size_t wsb,
               void* stream, bool finalize = true) {
    if (n < 0) return -2; if (lda < n) return -3; if (batch < 0) return -5;
    if (n == 0 || batch == 0) return 0;
    if (!A) return -1; if (!info) return -6;
    const int64_t npanels = cdiv64(n, NB);
    const size_t need = (size_t)batch * npanels * NB * NB * sizeof(T);
    if (!ws || wsb < need) return -7;
    if (batch > 65535) return -5;
    T* wsL = (T*)ws;
    hipStream_t st = (hipStream_t)stream;
    const size_t step_lds = (4 * (size_t)NB * LDD + 4 * SB * LDI + NB + 2 * SB) * sizeof(T);   // + the pivot-column exchange buffers
    nsgp_opt_in_lds((const void*)potrf_step_kernel<T>, step_lds);
    // Two-level blocking for large matrices: rank-64 updates stay inside an outer panel of NB2 columns (they are
    // HBM-bound: 8 flop/B in float64), the rest of the trailing matrix is updated once per outer panel with
    // K = NB2.  Matrices up to n = 2048 (the DSVI Kzz, n ~ 1024) are latency-bound and use one level.
    // Outer panel width (measured, float64, MI355X): n = 2048: 985 / 954 / 921 / 878 us at 256 / 512 / 1024 / one level;
    // n = 4096: 2316 / 2245 / 2166 / 2147 us at 256 / 512 / 1024 / 2048; n = 8192: 7.53 / 6.99 / 7.02 / 7.53 ms;
    // n = 16384: 37.7 / 35.8 / 35.9 ms at 256 / 512 / 1024.  NSGP_POTRF_NB2 (units of 64 columns) overrides for A/B runs.
    const char* pve = getenv("NSGP_POTRF_PANEL");
    const int old_panel = (pve && pve[0] == '1') ? 2 : 0;
    const char* nb2e = getenv("NSGP_POTRF_NB2");
    const int64_t nb2m = (nb2e && atoi(nb2e) > 0) ? atoi(nb2e) : (n <= 4096 ? 32 : 16);
    const int64_t NB2 = n > 2048 ? nb2m * NB : n;
    for (int64_t J0 = 0; J0 < n; J0 += NB2) {
        const int64_t Jend = (J0 + NB2) < n ? (J0 + NB2) : n;
        for (int64_t j0 = J0; j0 < Jend; j0 += NB) {
            const int64_t nb = (n - j0) < NB ? (n - j0) : NB;
            const int pre = j0 > J0;                              // previous panel's rank-64 update still pending
            if (nb < NB) {
                hipLaunchKernelGGL((potrf_tail_kernel<T>), dim3(1, (unsigned)batch), dim3(64), 0, st, A, n, lda, sA,
                                   j0, wsL, npanels, info, pre);
                break;
            }
            const int64_t below = n - j0 - nb;
            const int64_t nslab = below > 0 ? cdiv64(below, NB) : 1;
            // rest of the previous panel's update: block starting at j0 + 64, columns up to the outer panel's end
            const int64_t wcols = pre ? Jend - (j0 + nb) : 0;
            int64_t ntile = 0;
            if (below > 0 && wcols > 0) {
                const int64_t tm = cdiv64(below, NB), tn = cdiv64(wcols, NB);
                ntile = tn * (tn + 1) / 2 + (tm - tn) * tn;
            }
            hipLaunchKernelGGL((potrf_step_kernel<T>), dim3((unsigned)(nslab + ntile), (unsigned)batch), dim3(256),
                               step_lds, st, A, n, lda, sA, j0, wsL, npanels, info, nslab, pre | old_panel, wcols);
        }
        if (Jend < n) {
            const int64_t rest = n - Jend, kw = Jend - J0;
            T* Lp = A + Jend * lda + J0;
            T* A22 = A + Jend * lda + Jend;
            int rc = gemm_t<T>(rest, rest, kw, T(-1), Lp, lda, 1, sA, 0, Lp, 1, lda, sA, 0, T(1), A22, lda, sA, 0,
                               batch, 1, NSGP_GEMM_C_LOWER, stream);
            if (rc) return rc;
        }
    }
    // finalize = false (nsgp_potrf_trtri): the diagonal blocks stay in the side buffer, where the inverse reads them
    if (finalize)
        hipLaunchKernelGGL((potrf_finalize_kernel<T>), dim3((unsigned)npanels, (unsigned)npanels, (unsigned)batch),
                           dim3(256), 0, st, A, n, lda, sA, (const T*)wsL, npanels);
    return nsgp_launch_status();
}

// ---- trtri ---------------------------------------------------------------------------------
// acc(16x16) += A(16xK) B(Kx16), operands in LDS: A[m * lda + k], B[k * ldb + n]  (K a multiple of 4)
template <typename T>
__device__ __forceinline__ typename Mma16<T>::acc_t mm16(const T* A, int lda, const T* B, int ldb, int K,
                                                         typename Mma16<T>::acc_t acc, int lane) {
    const int fm = lane & 15, fk = lane >> 4;
    for (int k = 0; k < K; k += 4) acc = Mma16<T>::mma(A[fm * lda + k + fk], B[(k + fk) * ldb + fm], acc);
    return acc;
}

// Inverse of one 64x64 lower-triangular diagonal block per workgroup (4 waves): the four 16x16 diagonal
// sub-blocks are inverted by forward substitution (one wave each), then merged bottom-up with 16x16x4 MFMAs,
// X21 = -X22 (L21 X11), first into two 32x32 inverses, then into the 64x64 one.  A ragged last block is padded
// with the identity.  Everything right of the block in its rows is zeroed (X is lower triangular).
template <typename T>
__global__ __launch_bounds__(256) void trtri_diag_kernel(const T* __restrict__ L, int64_t n, int64_t ldl, int64_t sL,
                                                         T* __restrict__ X, int64_t ldx, int64_t sX,
                                                         const T* __restrict__ Dsrc) {
    // Dsrc != null: the diagonal blocks come from potrf's side buffer (batch, panels, 64, 64) -- L's own diagonal blocks
    // have not been written back (nsgp_potrf_trtri skips that pass)
    typedef Mma16<T> MM;
    typedef typename MM::acc_t acc_t;
    extern __shared__ __attribute__((aligned(16))) unsigned char panel_smem[];
    T* S = reinterpret_cast<T*>(panel_smem);            // [64][LDD] L block
    T* Xs = S + NB * LDD;                               // [64][LDD] inverse
    T* Tm = Xs + NB * LDD;                              // [64][LDD] products L21 X11
    T* Dinv = Tm + NB * LDD;                            // [4][16][LDI]
    T* rd = Dinv + 4 * SB * LDI;                        // [64]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t b = blockIdx.y, r0 = (int64_t)blockIdx.x * NB;
    const T* Lb = L + b * sL;
    T* Xb = X + b * sX;
    const int nb = (int)((n - r0) < NB ? (n - r0) : NB);
    {
        T lr[SB];
#pragma unroll
        for (int i = 0; i < SB; ++i) {
            const int64_t rr = r0 + w * SB + i, cc = r0 + lane;
            lr[i] = Dsrc ? Dsrc[((b * gridDim.x + blockIdx.x) * NB + (w * SB + i)) * NB + lane]
                         : Lb[(rr < n ? rr : n - 1) * ldl + (cc < n ? cc : n - 1)];
        }
#pragma unroll
        for (int i = 0; i < SB; ++i) keep(lr[i]);
#pragma unroll
        for (int i = 0; i < SB; ++i) {
            const int row = w * SB + i;
            const bool in = row < nb && lane < nb;
            S[row * LDD + lane] = in ? (lane <= row ? lr[i] : T(0)) : (row == lane ? T(1) : T(0));
            Xs[row * LDD + lane] = T(0);
        }
    }
    __syncthreads();
    if (tid < NB) rd[tid] = T(1) / S[tid * LDD + tid];
    __syncthreads();
    invert_subblock<T>(S, rd, Dinv, w, lane);
    __syncthreads();
    {   // diagonal sub-blocks of X
        const int i = lane >> 2, c0 = (lane & 3) * 4;
#pragma unroll
        for (int c = 0; c < 4; ++c) Xs[(w * SB + i) * LDD + w * SB + c0 + c] = Dinv[(w * SB + i) * LDI + c0 + c];
    }
    __syncthreads();
    const int fm = lane & 15;
    // level 1: 16 -> 32 (pairs (0,1) and (2,3); waves 0 and 1)
    {
        const int p = w, lo = 2 * p * SB, hi = lo + SB;
        acc_t t = {T(0), T(0), T(0), T(0)};
        if (w < 2) t = mm16<T>(S + hi * LDD + lo, LDD, Xs + lo * LDD + lo, LDD, SB, t, lane);
        if (w < 2) {
#pragma unroll
            for (int r = 0; r < 4; ++r) Tm[(hi + MM::crow(r, lane)) * LDD + lo + fm] = t[r];
        }
        __syncthreads();
        acc_t x = {T(0), T(0), T(0), T(0)};
        if (w < 2) x = mm16<T>(Xs + hi * LDD + hi, LDD, Tm + hi * LDD + lo, LDD, SB, x, lane);
        __syncthreads();
        if (w < 2) {
#pragma unroll
            for (int r = 0; r < 4; ++r) Xs[(hi + MM::crow(r, lane)) * LDD + lo + fm] = -x[r];
        }
        __syncthreads();
    }
    // level 2: 32 -> 64; wave w owns tile (ti, tj) of the 32x32 block X[32:64, 0:32]
    {
        const int ti = w >> 1, tj = w & 1;
        acc_t t = {T(0), T(0), T(0), T(0)};
        t = mm16<T>(S + (32 + ti * SB) * LDD, LDD, Xs + tj * SB, LDD, 32, t, lane);            // L21 X11
#pragma unroll
        for (int r = 0; r < 4; ++r) Tm[(32 + ti * SB + MM::crow(r, lane)) * LDD + tj * SB + fm] = t[r];
        __syncthreads();
        acc_t x = {T(0), T(0), T(0), T(0)};
        x = mm16<T>(Xs + (32 + ti * SB) * LDD + 32, LDD, Tm + 32 * LDD + tj * SB, LDD, 32, x, lane);   // X22 T
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) Xs[(32 + ti * SB + MM::crow(r, lane)) * LDD + tj * SB + fm] = -x[r];
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < SB; ++i) {
        const int row = w * SB + i;
        if (row < nb && lane < nb) Xb[(r0 + row) * ldx + r0 + lane] = Xs[row * LDD + lane];
    }
    // zero everything right of the diagonal block in these rows
    for (int i = w; i < nb; i += 4)
        for (int64_t c = r0 + nb + lane; c < n; c += 64) Xb[(r0 + i) * ldx + c] = T(0);
}

template <typename T>
int trtri_impl(const T* L, int64_t n, int64_t ldl, int64_t sL, T* X, int64_t ldx, int64_t sX, int64_t batch,
               void* ws, size_t wsb, void* stream, const T* diag_src = nullptr) {
    if (n < 0) return -2; if (ldl < n) return -3; if (ldx < n) return -6; if (batch < 0) return -8;
    if (n == 0 || batch == 0) return 0;
    if (!L) return -1; if (!X) return -5;
    const size_t need = (size_t)batch * n * n * sizeof(T);
    if (n > NB && (!ws || wsb < need)) return -9;
    if (batch > 65535) return -8;
    hipStream_t st = (hipStream_t)stream;
    const size_t diag_lds = (3 * (size_t)NB * LDD + 4 * SB * LDI + NB) * sizeof(T);
    nsgp_opt_in_lds((const void*)trtri_diag_kernel<T>, diag_lds);
    hipLaunchKernelGGL((trtri_diag_kernel<T>), dim3((unsigned)cdiv64(n, NB), (unsigned)batch), dim3(256), diag_lds, st,
                       L, n, ldl, sL, X, ldx, sX, diag_src);
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

// Factor + inverse with the inverse accumulated inside the panel launches (n a multiple of 64, at most 2048: one level).
template <typename T>
int potrf_inv_impl(T* A, int64_t n, int64_t lda, int64_t sA, int64_t batch, int32_t* info, T* X, int64_t ldx, int64_t sX,
                   void* ws, size_t wsb, void* stream, float* X32 = nullptr) {
    const int64_t npanels = n / NB;
    const size_t need = (size_t)batch * npanels * NB * NB * sizeof(T);
    if (!ws || wsb < need) return -11;
    if (batch > 65535) return -5;
    T* wsL = (T*)ws;
    hipStream_t st = (hipStream_t)stream;
    const size_t step_lds = (4 * (size_t)NB * LDD + 4 * SB * LDI + NB + 2 * SB) * sizeof(T);   // + the pivot-column exchange buffers
    nsgp_opt_in_lds((const void*)potrf_inv_step_kernel<T>, step_lds);
    for (int64_t j0 = 0; j0 < n; j0 += NB) {
        const int pre = j0 > 0;
        const int64_t below = n - j0 - NB;
        const int64_t nslab = below > 0 ? below / NB : 1;
        const int64_t nprow = j0 / NB + 1;
        const int64_t wcols = pre ? below : 0;
        int64_t nsyrk = 0, npupd = 0;
        if (pre && below > 0) {
            const int64_t tn = below / NB;
            nsyrk = tn * (tn + 1) / 2;
            npupd = tn * (j0 / NB);
        }
        hipLaunchKernelGGL((potrf_inv_step_kernel<T>), dim3((unsigned)(nslab + nprow + nsyrk + npupd), (unsigned)batch),
                           dim3(256), step_lds, st, A, n, lda, sA, j0, wsL, npanels, info, X, ldx, sX, nslab, nprow, nsyrk,
                           pre, wcols, X32);
    }
    return nsgp_launch_status();
}

// Cholesky factor and its inverse in one call: X = chol(A)^-1 (lower), A is overwritten with intermediate data (its strictly
// lower part holds L21; the diagonal blocks and the upper triangle are NOT a valid factor).  Saves the write-back pass of
// the factor (potrf_finalize_kernel: one launch, 2 n^2 elements of traffic) on the DSVI whitening chain, which only needs
// the inverse.  ws: nsgp_potrf_workspace + nsgp_trtri_workspace bytes.
template <typename T>
int potrf_trtri_impl(T* A, int64_t n, int64_t lda, int64_t sA, int64_t batch, int32_t* info, T* X, int64_t ldx, int64_t sX,
                     void* ws, size_t wsb, void* stream, float* X32 = nullptr, int* wrote32 = nullptr) {
    // argument order of the C entry points: A 1, n 2, lda 3, sA 4, batch 5, info 6, X 7, ldx 8, sX 9 (as potrf_impl / trtri_impl)
    if (n < 0) return -2; if (batch < 0) return -5;
    if (n == 0 || batch == 0) return 0;
    if (!A) return -1; if (lda < n) return -3; if (batch > 65535) return -5; if (!info) return -6;
    if (!X) return -7; if (ldx < n) return -8;
    const size_t w1 = (size_t)batch * cdiv64(n, NB) * NB * NB * sizeof(T);
    const size_t w2 = n > NB ? (size_t)batch * n * n * sizeof(T) : 0;
    if (!ws || wsb < w1 + w2) return -11;
    const char* inve = getenv("NSGP_POTRF_INV");                     // A/B switch: 0 = factor, then the recursive inverse
    if (wrote32) *wrote32 = 0;
    if (n % NB == 0 && n <= 2048 && !(inve && inve[0] == '0')) {
        if (wrote32 && X32) *wrote32 = 1;
        return potrf_inv_impl<T>(A, n, lda, sA, batch, info, X, ldx, sX, ws, w1, stream, X32);
    }
    int rc = potrf_impl<T>(A, n, lda, sA, batch, info, ws, w1, stream, false);
    if (rc) return rc;
    return trtri_impl<T>(A, n, lda, sA, X, ldx, sX, batch, (char*)ws + w1, w2, stream, (const T*)ws);
}

}  // namespace

extern "C" {
#ifdef NSGP_POTRF_STAMPS
int nsgp_debug_potrf_stamps(void* buf, uint64_t cap_records) {
    unsigned long long* b = (unsigned long long*)buf;
    unsigned long long c = cap_records;
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(nsgp_pstamp_buf), &b, sizeof(b));
    if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(nsgp_pstamp_cap), &c, sizeof(c));
    return (int)e;
}
#endif

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
int nsgp_potrf_trtri_f32(float* A, int64_t n, int64_t lda, int64_t sA, int64_t batch, int32_t* info, float* X,
                         int64_t ldx, int64_t sX, void* ws, size_t wsb, void* stream) {
    return potrf_trtri_impl<float>(A, n, lda, sA, batch, info, X, ldx, sX, ws, wsb, stream);
}
int nsgp_potrf_trtri_f64(double* A, int64_t n, int64_t lda, int64_t sA, int64_t batch, int32_t* info, double* X,
                         int64_t ldx, int64_t sX, void* ws, size_t wsb, void* stream) {
    return potrf_trtri_impl<double>(A, n, lda, sA, batch, info, X, ldx, sX, ws, wsb, stream);
}
int nsgp_potrf_trtri_f64_w32(double* A, int64_t n, int64_t lda, int64_t sA, int64_t batch, int32_t* info, double* X,
                             int64_t ldx, int64_t sX, float* X32, int* wrote32, void* ws, size_t wsb, void* stream) {
    return potrf_trtri_impl<double>(A, n, lda, sA, batch, info, X, ldx, sX, ws, wsb, stream, X32, wrote32);
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
