// gemm.hip -- dense contraction on the gfx950 matrix cores, exact f32 / f64 arithmetic.
//
//   C[b] = alpha * op(A[b]) * op(B[b]) + beta * C[b]
//
// f32: v_mfma_f32_32x32x2_f32 (64 FLOP/clk/SIMD, 155 TF chip peak; no xf32/TF32 on gfx950)
// f64: v_mfma_f64_16x16x4_f64
// This is the one genuine dense contraction of the path: L^-1 Kzx, Lq^T A, (S-I)A and their
// adjoints in the whitened SVGP layer (gpytorch VariationalStrategy.forward), the
// trailing updates of the blocked Cholesky, trtri, and K_xz Kzz^-1/2 of the SGPR path
// (models/gibbs_kernels.py:225).
//
// Structure: 256-thread workgroup = 2x2 waves, LDS tiles stored k-major (As[k][m], Bs[k][n]) so an
// MFMA operand fragment is one conflict-free ds_read per lane; global loads are 4 elements per lane
// along whichever of (m|k) is contiguous, prefetched into registers one K-tile ahead, two LDS
// buffers, one barrier per K-tile.  Triangular operands skip whole K-tiles and are masked in the
// diagonal tiles; small outputs with a long inner dimension are split along K into slabs that a
// second kernel sums in a fixed order (deterministic).  Tiles: 128x128x32 (f32, two workgroups per CU),
// 128x64x32 (f32, three per CU; single-round launches with a triangular A operand), 64x64x16 (f32 small, f64).
// Fused variants for the SVGP layer (template parameters EPI / KSC, `struct Epi`): column statistics or the
// A-adjoint assembled in the epilogue, an operand scaled along k in the loader -- the elementwise passes of
// VariationalStrategy.forward and of its backward never run as separate kernels.
#include <cstdlib>
#include "common.h"

namespace {

template <typename T> struct Mfma;
template <> struct Mfma<float> {
    static constexpr int MT = 32, KS = 2, NREG = 16, PAD = 4;
    typedef float acc_t __attribute__((ext_vector_type(16)));
    static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ int krow(int lane) { return lane >> 5; }      // k index of the operand lane
    static __device__ __forceinline__ int mcol(int lane) { return lane & 31; }      // m / n index of the operand lane
    static __device__ __forceinline__ int crow(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }
    static __device__ __forceinline__ int ccol(int lane) { return lane & 31; }
};
template <> struct Mfma<double> {
    static constexpr int MT = 16, KS = 4, NREG = 4, PAD = 16;
    typedef double acc_t __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ acc_t mma(double a, double b, acc_t c) {
        return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ int krow(int lane) { return lane >> 4; }
    static __device__ __forceinline__ int mcol(int lane) { return lane & 15; }
    static __device__ __forceinline__ int crow(int r, int lane) { return (lane >> 4) + 4 * r; }
    static __device__ __forceinline__ int ccol(int lane) { return lane & 15; }
};

struct GemmArgs {
    int64_t M, N, K;
    int64_t sam, sak, sa1, sa2;
    int64_t sbk, sbn, sb1, sb2;
    int64_t ldc, sc1, sc2;
    int64_t nb2;
    int64_t tiles_m, tiles_n;
    int64_t ksplit, kper;         // K-slices (kper is a multiple of BK)
    int64_t slab;                 // elements per (slice) slab = nb1*nb2*M*N when ksplit > 1
    int flags;
    int vecA, vecB;               // 16-byte vector global loads allowed for A / B
    int modeA, modeB;             // 0: contiguous along k, 1: contiguous along m (n)
};

// Fused epilogues / loader hooks of the SVGP projection GEMMs (kind 0 = plain GEMM).
//   kind 1 (column statistics): C = alpha*acc as usual, plus per-tile-row partial column sums
//          p0[bb][bm][col] = sum_{rows of tile bm} C[row][col] * rv[bb][row]   (skipped when p0 == null)
//          p1[bb][bm][col] = sum_{rows of tile bm} C[row][col]^2
//   kind 2 (SVGP A-adjoint): C[row][col] = alpha*acc * 2 cs[col] + rv[row] * gc[col] - 2 cs[col] * mat[row][col]
//          (mat has C's layout; cs, gc: (batch, N); rv: (batch, M))
//   ks != null: the B operand is scaled along k on its way into LDS, B'(k, n) = ks[bb][k] * B(k, n).
struct Epi {
    int kind;
    const void *rv, *cs, *gc, *mat, *ks;
    void *p0, *p1;
    int modeA, modeB;       // operand modes of the instantiated variant (any strides are valid in either mode;
                            // the mode only decides the vector-load direction)
};

// 4-element register fragment loaded from global
template <typename T> struct Frag4 { T v[4]; };

template <typename T> __device__ __forceinline__ Frag4<T> ldg4(const T* p) {
    Frag4<T> f;
    if constexpr (sizeof(T) == 4) {
        const float4 q = *reinterpret_cast<const float4*>(p);
        f.v[0] = q.x; f.v[1] = q.y; f.v[2] = q.z; f.v[3] = q.w;
    } else {
        const double2 q0 = *reinterpret_cast<const double2*>(p);
        const double2 q1 = *reinterpret_cast<const double2*>(p + 2);
        f.v[0] = q0.x; f.v[1] = q0.y; f.v[2] = q1.x; f.v[3] = q1.y;
    }
    return f;
}

// Load 4 elements of an operand tile.  (r, k) is the element's (m|n, k) position; the 4 elements run
// along k (mode 0) or along r (mode 1).  `lo`/`up`: zero where k > r (lower) / k < r (upper) for A,
// and for B (r = n): "B lower" zero where n > k, "B upper" zero where n < k.
template <typename T>
__device__ __forceinline__ Frag4<T> load_operand4(const T* base, int64_t sr, int64_t sk, int64_t r, int64_t k,
                                                  int64_t R, int64_t kend, int mode, int vec, bool zero_k_gt_r,
                                                  bool zero_k_lt_r) {
    Frag4<T> f;
    if (mode == 0) {
        if (vec && r < R && k + 3 < kend) {
            f = ldg4(base + r * sr + k);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) f.v[e] = (r < R && k + e < kend) ? base[r * sr + (k + e) * sk] : T(0);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (zero_k_gt_r && k + e > r) f.v[e] = T(0);
            if (zero_k_lt_r && k + e < r) f.v[e] = T(0);
        }
    } else {
        if (vec && k < kend && r + 3 < R) {
            f = ldg4(base + k * sk + r);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) f.v[e] = (k < kend && r + e < R) ? base[(r + e) * sr + k * sk] : T(0);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (zero_k_gt_r && k > r + e) f.v[e] = T(0);
            if (zero_k_lt_r && k < r + e) f.v[e] = T(0);
        }
    }
    return f;
}

// MODE_A / MODE_B: 0 = operand contiguous along k, 1 = contiguous along m (n).
template <typename T, int BM, int BN, int BK, int MODE_A, int MODE_B, int EPI = 0, int KSC = 0, int PF = 0>
__global__ __launch_bounds__(256, (BM == 128 && BN == 64) ? 3 : 2) void gemm_kernel(GemmArgs g, T alpha, const T* __restrict__ A,
                                                   const T* __restrict__ B, T beta, T* __restrict__ C,
                                                   T* __restrict__ slabs, Epi ep) {
    using MF = Mfma<T>;
    constexpr int MT = MF::MT, KS = MF::KS, NKK = BK / KS;
    constexpr int KCH = NKK < 8 ? NKK : 8;                      // k-steps whose fragments are prefetched together
    constexpr int WM = BM / 2, WN = BN / 2;
    constexpr int TM = WM / MT, TN = WN / MT;
    constexpr int PADA = (MODE_A == 0 && sizeof(T) == 4) ? 1 : MF::PAD;
    constexpr int PADB = (MODE_B == 0 && sizeof(T) == 4) ? 1 : MF::PAD;
    constexpr int LDA = BM + PADA, LDB = BN + PADB;
    constexpr int PA = BM * BK / 1024, PB = BN * BK / 1024;     // 4-element fragments per thread
    constexpr int TPRA = BM / 4, TPRB = BN / 4;                 // threads per k-row in mode 1
    constexpr int TPK = BK / 4;                                 // threads per row in mode 0 (full BK bytes)
    // all LDS in ONE dynamic region (the 128x128x32 f32 tile needs 66 KB > the 64 KB static limit)
    extern __shared__ __attribute__((aligned(32))) unsigned char gemm_smem[];
    T (*As)[BK * LDA] = reinterpret_cast<T (*)[BK * LDA]>(gemm_smem);
    T (*Bs)[BK * LDB] = reinterpret_cast<T (*)[BK * LDB]>(gemm_smem + 2 * BK * LDA * sizeof(T));

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;
    // tile id -> (bm, bn): bn fastest, so the workgroups in flight share one A row panel (L2 resident) and
    // stream disjoint B column panels.  Triangular operands / outputs make the work per tile uneven:
    //  * rows with the longest K range are issued first (longest-processing-time order);
    //  * the column index is rotated by the row index.  Workgroups are dealt round-robin to the 8 XCDs
    //    (id % 8), so without the rotation an XCD owns fixed tile COLUMNS -- with a lower-triangular
    //    output and 8 tile columns XCD 0 gets 8 active tiles per slice and XCD 7 one (measured: the
    //    lower-only M=N=1024, K=40960 product took 879 us, the FULL product 780 us).
    const int64_t tid_lin = blockIdx.x;
    int64_t bm, bn;
    if (g.flags & NSGP_GEMM_C_LOWER) {
        // compact enumeration of the ACTIVE tiles only (n0 <= m0 + BM - 1): launching the strictly-upper
        // tiles as no-op workgroups perturbs the dispatcher's CU placement (measured: lower-only product
        // no faster than the full one, CUs ~58 % busy).  Rows 0..T-1 form a triangle, the rest are full.
        const int64_t TT = g.tiles_m < g.tiles_n ? g.tiles_m : g.tiles_n;
        const int64_t tri = TT * (TT + 1) / 2;
        if (tid_lin < tri) {
            int64_t r = (int64_t)((sqrtf(8.0f * (float)tid_lin + 1.0f) - 1.0f) * 0.5f);
            while (r * (r + 1) / 2 > tid_lin) --r;
            while ((r + 1) * (r + 2) / 2 <= tid_lin) ++r;
            bm = r; bn = tid_lin - r * (r + 1) / 2;
        } else {
            const int64_t q = tid_lin - tri;
            bm = TT + q / g.tiles_n; bn = q % g.tiles_n;
        }
    } else {
        const int64_t brow = tid_lin / g.tiles_n;
        // (j - row) mod tiles_n: spreads a triangular B operand's heavy columns over the XCDs (id % 8)
        bm = brow; bn = (tid_lin % g.tiles_n + g.tiles_n - brow % g.tiles_n) % g.tiles_n;
    }
    if (g.flags & NSGP_GEMM_A_LOWER) bm = g.tiles_m - 1 - bm;          // large m = long K range
    if (g.flags & NSGP_GEMM_B_UPPER) bn = g.tiles_n - 1 - bn;          // large n = long K range
    const int64_t z = blockIdx.y;
    const int64_t slice = z % g.ksplit, bb = z / g.ksplit;
    const int64_t b1 = bb / g.nb2, b2 = bb % g.nb2;
    const int64_t m0 = bm * BM, n0 = bn * BN;

    const T* Ab = A + b1 * g.sa1 + b2 * g.sa2;
    const T* Bb = B + b1 * g.sb1 + b2 * g.sb2;
    T* Cb = C + b1 * g.sc1 + b2 * g.sc2;

    const bool aL = g.flags & NSGP_GEMM_A_LOWER, aU = g.flags & NSGP_GEMM_A_UPPER;
    const bool bL = g.flags & NSGP_GEMM_B_LOWER, bU = g.flags & NSGP_GEMM_B_UPPER;
    const bool cL = g.flags & NSGP_GEMM_C_LOWER;

    // K range of this block: slice intersected with the triangular support
    int64_t kbeg = slice * g.kper, kend = kbeg + g.kper < g.K ? kbeg + g.kper : g.K;
    if (aL) { const int64_t e = m0 + BM; if (e < kend) kend = e; }                 // k <= m
    if (aU) { const int64_t s = m0 / BK * BK; if (s > kbeg) kbeg = s; }            // k >= m
    if (bL) { const int64_t s = n0 / BK * BK; if (s > kbeg) kbeg = s; }            // k >= n
    if (bU) { const int64_t e = n0 + BN; if (e < kend) kend = e; }                 // k <= n
    const bool skip_block = cL && (n0 > m0 + BM - 1);
    if (skip_block) kend = kbeg;
    const int64_t nt = kend > kbeg ? (kend - kbeg + BK - 1) / BK : 0;

    typename MF::acc_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < MF::NREG; ++r) acc[i][j][r] = T(0);

    // per-thread tile coordinates of its 4-element fragments
    int ar[PA], ak[PA], br[PB], bk[PB];
#pragma unroll
    for (int p = 0; p < PA; ++p) {
        if (MODE_A == 0) { ar[p] = p * (256 / TPK) + tid / TPK; ak[p] = (tid % TPK) * 4; }
        else { ar[p] = (tid % TPRA) * 4; ak[p] = p * (256 / TPRA) + tid / TPRA; }
    }
#pragma unroll
    for (int p = 0; p < PB; ++p) {
        if (MODE_B == 0) { br[p] = p * (256 / TPK) + tid / TPK; bk[p] = (tid % TPK) * 4; }
        else { br[p] = (tid % TPRB) * 4; bk[p] = p * (256 / TPRB) + tid / TPRB; }
    }
    // fast path: every 16-byte fragment of the tile is in bounds, aligned and unmasked
    const bool fastA = g.vecA && (m0 + BM <= g.M);
    const bool fastB = g.vecB && (n0 + BN <= g.N);
    const T* pa[PA];
    const T* pb[PB];
#pragma unroll
    for (int p = 0; p < PA; ++p)
        pa[p] = MODE_A == 0 ? Ab + (m0 + ar[p]) * g.sam + ak[p] : Ab + (m0 + ar[p]) + (int64_t)ak[p] * g.sak;
#pragma unroll
    for (int p = 0; p < PB; ++p)
        pb[p] = MODE_B == 0 ? Bb + (n0 + br[p]) * g.sbn + bk[p] : Bb + (n0 + br[p]) + (int64_t)bk[p] * g.sbk;
    const int64_t stepA = MODE_A == 0 ? 1 : g.sak, stepB = MODE_B == 0 ? 1 : g.sbk;

    // PF = 1 (float64, grids of at most one round): the tile is small (16 MFMAs per wave and K-tile, 0.25 us),
    // one K-tile of prefetch does not cover a global-load latency, so TWO K-tiles are kept in flight in two
    // register sets.  It costs a wave of occupancy (3 -> 2 per SIMD), which multi-round grids need more
    // (N=16384 potrf: 39 ms without, 51 ms with), so those keep PF = 0.
    constexpr bool PF2 = PF != 0 && sizeof(T) == 8 && KSC == 0;
    Frag4<T> ra0[PA], rb0[PB], ra1[PF2 ? PA : 1], rb1[PF2 ? PB : 1];
    Frag4<T> rks[KSC ? PB : 1];
    const T* ksb = KSC ? reinterpret_cast<const T*>(ep.ks) + bb * g.K : nullptr;
    const bool ks_vec = KSC && ((uintptr_t)ksb % (4 * sizeof(T)) == 0);

    auto gload = [&](int64_t k0, Frag4<T>* ra, Frag4<T>* rb) {
        const bool kfull = k0 + BK <= kend;
        const bool a_diag = (aL || aU) && (k0 < m0 + BM) && (k0 + BK > m0);
        const bool b_diag = (bL || bU) && (k0 < n0 + BN) && (k0 + BK > n0);
        if (fastA && kfull && !a_diag) {
#pragma unroll
            for (int p = 0; p < PA; ++p) ra[p] = ldg4(pa[p] + k0 * stepA);
        } else {
#pragma unroll
            for (int p = 0; p < PA; ++p)
                ra[p] = load_operand4<T>(Ab, g.sam, g.sak, m0 + ar[p], k0 + ak[p], g.M, kend, MODE_A, g.vecA, aL, aU);
        }
        if (fastB && kfull && !b_diag) {
#pragma unroll
            for (int p = 0; p < PB; ++p) rb[p] = ldg4(pb[p] + k0 * stepB);
        } else {
#pragma unroll
            for (int p = 0; p < PB; ++p)   // B(k,n): "lower" zero where n > k <=> k < r ; "upper" zero where k > r
                rb[p] = load_operand4<T>(Bb, g.sbn, g.sbk, n0 + br[p], k0 + bk[p], g.N, kend, MODE_B, g.vecB, bU, bL);
        }
        if constexpr (KSC != 0) {
            if (MODE_B == 0) {
                // every fragment of this thread covers the same 4 k's (bk[p] does not depend on p): ONE load
                const int64_t k = k0 + bk[0];
                if (ks_vec && k + 3 < kend) {
                    rks[0] = ldg4(ksb + k);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) rks[0].v[e] = k + e < kend ? ksb[k + e] : T(0);
                }
            } else {
#pragma unroll
                for (int p = 0; p < PB; ++p) {
                    const int64_t k = k0 + bk[p];
                    rks[p].v[0] = k < kend ? ksb[k] : T(0);
                }
            }
        }
    };
    auto sstore = [&](int buf, Frag4<T>* ra, Frag4<T>* rb) {
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            if (MODE_A == 0) {
#pragma unroll
                for (int e = 0; e < 4; ++e) As[buf][(ak[p] + e) * LDA + ar[p]] = ra[p].v[e];
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) As[buf][ak[p] * LDA + ar[p] + e] = ra[p].v[e];
            }
        }
        if constexpr (KSC != 0) {
#pragma unroll
            for (int p = 0; p < PB; ++p)
#pragma unroll
                for (int e = 0; e < 4; ++e) rb[p].v[e] *= (MODE_B == 0 ? rks[0].v[e] : rks[p].v[0]);
        }
#pragma unroll
        for (int p = 0; p < PB; ++p) {
            if (MODE_B == 0) {
#pragma unroll
                for (int e = 0; e < 4; ++e) Bs[buf][(bk[p] + e) * LDB + br[p]] = rb[p].v[e];
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) Bs[buf][bk[p] * LDB + br[p] + e] = rb[p].v[e];
            }
        }
    };

    const int kr = MF::krow(lane), mc = MF::mcol(lane);
    auto compute = [&](int buf) {
        const T* as = &As[buf][kr * LDA + wm0 + mc];
        const T* bs = &Bs[buf][kr * LDB + wn0 + mc];
        // operand fragments of KCH k-steps first (LDS latency overlaps the MFMA stream), then the MFMAs
#pragma unroll
        for (int kc = 0; kc < NKK; kc += KCH) {
            T af[KCH][TM], bf[KCH][TN];
#pragma unroll
            for (int kk = 0; kk < KCH; ++kk) {
#pragma unroll
                for (int i = 0; i < TM; ++i) af[kk][i] = as[(kc + kk) * KS * LDA + i * MT];
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[kk][j] = bs[(kc + kk) * KS * LDB + j * MT];
            }
#pragma unroll
            for (int kk = 0; kk < KCH; ++kk)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = MF::mma(af[kk][i], bf[kk][j], acc[i][j]);
        }
    };
    if constexpr (PF2) {
        if (nt > 0) {
            gload(kbeg, ra0, rb0);
            if (nt > 1) gload(kbeg + BK, ra1, rb1);
            sstore(0, ra0, rb0);
            __syncthreads();
            for (int64_t t = 0; t < nt; t += 2) {
                // even step: set 0 is free (stored), set 1 holds tile t+1
                if (t + 2 < nt) gload(kbeg + (t + 2) * BK, ra0, rb0);
                compute(0);
                if (t + 1 < nt) sstore(1, ra1, rb1);
                __syncthreads();
                if (t + 1 >= nt) break;
                // odd step: set 1 is free, set 0 holds tile t+2
                if (t + 3 < nt) gload(kbeg + (t + 3) * BK, ra1, rb1);
                compute(1);
                if (t + 2 < nt) sstore(0, ra0, rb0);
                __syncthreads();
            }
        }
    } else {
        if (nt > 0) {
            gload(kbeg, ra0, rb0);
            sstore(0, ra0, rb0);
            __syncthreads();
            for (int64_t t = 0; t < nt; ++t) {
                const int buf = (int)(t & 1);
                if (t + 1 < nt) gload(kbeg + (t + 1) * BK, ra0, rb0);      // next tile -> registers (in flight)
                compute(buf);
                if (t + 1 < nt) sstore(buf ^ 1, ra0, rb0);
                __syncthreads();
            }
        }
    }

    // epilogue
    if constexpr (EPI == 1) {
        // plain store + per-column partial sums over this tile's rows (rows >= M carry acc == 0)
        const T* rv = ep.rv ? reinterpret_cast<const T*>(ep.rv) + bb * g.M : nullptr;
        T sdot[TN], ssq[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) { sdot[j] = T(0); ssq[j] = T(0); }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < MF::NREG; ++r) {
                const int64_t row = m0 + wm0 + i * MT + MF::crow(r, lane);
                const T rvv = (rv && row < g.M) ? rv[row] : T(0);
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int64_t col = n0 + wn0 + j * MT + MF::ccol(lane);
                    const T v = alpha * acc[i][j][r];
                    if (row < g.M && col < g.N) Cb[row * g.ldc + col] = v;
                    sdot[j] += v * rvv;
                    ssq[j] += v * v;
                }
            }
        // lanes that share a column (f32: lane ^ 32; f64: lane ^ 16, lane ^ 32), then the two waves along m
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int off = MT; off < 64; off <<= 1) {
                sdot[j] += __shfl_xor(sdot[j], off);
                ssq[j] += __shfl_xor(ssq[j], off);
            }
        }
        T* red = reinterpret_cast<T*>(gemm_smem);               // [2 quantities][2 waves along m][BN]
        __syncthreads();
        if (lane < MT) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int c = wn0 + j * MT + lane;
                red[(0 + (wave >> 1)) * BN + c] = sdot[j];
                red[(2 + (wave >> 1)) * BN + c] = ssq[j];
            }
        }
        __syncthreads();
        if (tid < BN && n0 + tid < g.N) {
            const int64_t o = (bb * g.tiles_m + bm) * g.N + n0 + tid;
            if (ep.p0) reinterpret_cast<T*>(ep.p0)[o] = red[tid] + red[BN + tid];
            reinterpret_cast<T*>(ep.p1)[o] = red[2 * BN + tid] + red[3 * BN + tid];
        }
        return;
    } else if constexpr (EPI == 2) {
        const T* mat = reinterpret_cast<const T*>(ep.mat) + b1 * g.sc1 + b2 * g.sc2;
        const T* rv = reinterpret_cast<const T*>(ep.rv) + bb * g.M;
        const T* gc = reinterpret_cast<const T*>(ep.gc) + bb * g.N;
        const T* cs = reinterpret_cast<const T*>(ep.cs) + bb * g.N;
        const bool inside = (m0 + BM <= g.M) && (n0 + BN <= g.N);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int64_t col = n0 + wn0 + j * MT + MF::ccol(lane);
            const int64_t colc = col < g.N ? col : g.N - 1;
            const T gcol = gc[colc], c2 = T(2) * cs[colc];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                T av[MF::NREG], rvv[MF::NREG];
                if (inside) {
#pragma unroll
                    for (int r = 0; r < MF::NREG; ++r) {
                        const int64_t row = m0 + wm0 + i * MT + MF::crow(r, lane);
                        av[r] = mat[row * g.ldc + col];
                        rvv[r] = rv[row];
                    }
#pragma unroll
                    for (int r = 0; r < MF::NREG; ++r) {
                        const int64_t row = m0 + wm0 + i * MT + MF::crow(r, lane);
                        Cb[row * g.ldc + col] = (alpha * acc[i][j][r] - av[r]) * c2 + rvv[r] * gcol;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < MF::NREG; ++r) {
                        const int64_t row = m0 + wm0 + i * MT + MF::crow(r, lane);
                        if (row < g.M && col < g.N)
                            Cb[row * g.ldc + col] = (alpha * acc[i][j][r] - mat[row * g.ldc + col]) * c2 + rv[row] * gcol;
                    }
                }
            }
        }
        return;
    }
    const bool to_slab = g.ksplit > 1;
    T* out = to_slab ? slabs + slice * g.slab + bb * g.M * g.N : Cb;
    const int64_t ldo = to_slab ? g.N : g.ldc;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < MF::NREG; ++r) {
                const int64_t row = m0 + wm0 + i * MT + MF::crow(r, lane);
                const int64_t col = n0 + wn0 + j * MT + MF::ccol(lane);
                if (row < g.M && col < g.N) {
                    T v = acc[i][j][r];
                    if (to_slab) {
                        out[row * ldo + col] = v;
                    } else {
                        if (cL && col > row) {
                            if (beta == T(0)) out[row * ldo + col] = T(0);
                        } else {
                            v *= alpha;
                            if (beta != T(0)) v += beta * out[row * ldo + col];
                            out[row * ldo + col] = v;
                        }
                    }
                }
            }
}

// C = alpha * sum_s slab[s] + beta * C   (fixed summation order)
template <typename T>
__global__ void splitk_reduce_kernel(GemmArgs g, T alpha, const T* __restrict__ slabs, T beta, T* __restrict__ C,
                                     int64_t nb) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t per = g.M * g.N;
    if (idx >= nb * per) return;
    const int64_t bb = idx / per, e = idx % per;
    const int64_t row = e / g.N, col = e % g.N;
    const int64_t b1 = bb / g.nb2, b2 = bb % g.nb2;
    T* c = C + b1 * g.sc1 + b2 * g.sc2 + row * g.ldc + col;
    if ((g.flags & NSGP_GEMM_C_LOWER) && col > row) {
        if (beta == T(0)) *c = T(0);
        return;
    }
    T s = T(0);
    for (int64_t k = 0; k < g.ksplit; ++k) s += slabs[k * g.slab + idx];
    s *= alpha;
    if (beta != T(0)) s += beta * *c;
    *c = s;
}

struct Plan { int big, narrow; int64_t ksplit, kper; };

static inline int64_t active_tiles(int64_t M, int64_t N, int64_t bm, int flags) {
    const int64_t tm = cdiv64(M, bm), tn = cdiv64(N, bm);
    if (!(flags & NSGP_GEMM_C_LOWER)) return tm * tn;
    int64_t cnt = 0;                                  // tiles with n0 <= m0 + bm - 1
    for (int64_t i = 0; i < tm; ++i) cnt += (i + 1 < tn ? i + 1 : tn);
    return cnt;
}

// Tile size and K-split.  Resident workgroups ("slots") = 256 CUs x blocks/CU for the kernel variant
// (2 for the 128x128 f32 tile, 6 / 3 for the 64x64 f32 / f64 tiles); the number of rounds the grid needs
// is ceil(active_blocks / slots), so the K-split is chosen to minimise rounds x K-tiles per block plus
// the slab traffic of the split (measured: 792 active blocks on 512 slots ran 905 us, CUs 55 % busy).
template <typename T> Plan make_plan(int64_t M, int64_t N, int64_t K, int64_t nb, int flags, bool allow_big = true) {
    Plan p;
    const bool can_split = !(flags & NSGP_GEMM_NO_SPLITK) && K >= 512;
    int64_t maxks = can_split ? K / 256 : 1;
    if (maxks > 64) maxks = 64;
    if (maxks < 1) maxks = 1;
    const int64_t tiles_big = active_tiles(M, N, 128, flags) * nb;
    // 128x128 tiles (f32 only) when, with K-splitting, they still fill the 256 CUs.  (A 128x128x16 float64
    // tile was tried: 256 VGPRs + 132 AGPRs, one wave per SIMD -- 3 x 1024^3 went from 0.40 to 0.59 ms.)
    p.big = allow_big && sizeof(T) == 4 && tiles_big * maxks >= 256;
    const int64_t bm = p.big ? 128 : 64;
    const int64_t tiles = active_tiles(M, N, bm, flags) * nb;
    const int64_t slots = 256 * (p.big ? 2 : (sizeof(T) == 4 ? 6 : 3));
    const int64_t ktiles = cdiv64(K, 16);
    int64_t best_ks = 1;
    double best_cost = 1e300;
    for (int64_t ks = 1; ks <= maxks; ++ks) {
        const int64_t rounds = cdiv64(tiles * ks, slots);
        // K-tiles of MFMA work per round + per-block fixed cost (prologue/epilogue ~ 6 K-tiles) + the
        // split's slab write/read (in K-tile units: one 128x128 fp32 slab ~ 2 K-tiles of time)
        const double cost = (double)rounds * ((double)cdiv64(ktiles, ks) + 6.0) + (ks > 1 ? 2.0 * ks / 4.0 : 0.0);
        if (cost < best_cost - 1e-9) { best_cost = cost; best_ks = ks; }
    }
    p.kper = cdiv64(cdiv64(K, best_ks), 32) * 32;
    if (p.kper < 32) p.kper = 32;
    p.ksplit = cdiv64(K, p.kper);
    if (p.ksplit < 1) p.ksplit = 1;
    // A triangular A operand makes the work per tile ROW uneven (K range 1/8 .. 8/8 of K at M = 1024).  When the
    // grid fits one round of the 512 slots, list scheduling cannot balance it: the launch takes as long as its
    // longest tile (measured: hidden layer, n = 4096 x 2 GPs, 56 % of the balanced time).  128 x 64 tiles double
    // the tile count at half the work each, so short tiles back-fill behind the long ones.
    p.narrow = p.big && p.ksplit == 1 && !(flags & NSGP_GEMM_C_LOWER) &&
               (flags & (NSGP_GEMM_A_LOWER | NSGP_GEMM_A_UPPER)) && tiles_big <= slots;
    return p;
}

template <typename T>
int gemm_impl(int64_t M, int64_t N, int64_t K, T alpha, const T* A, int64_t sam, int64_t sak, int64_t sa1,
              int64_t sa2, const T* B, int64_t sbk, int64_t sbn, int64_t sb1, int64_t sb2, T beta, T* C,
              int64_t ldc, int64_t sc1, int64_t sc2, int64_t nb1, int64_t nb2, int flags, void* ws, size_t wsb,
              void* stream, const Epi* epi = nullptr, int64_t* tiles_m_out = nullptr) {
    if (M < 0) return -1; if (N < 0) return -2; if (K < 0) return -3;
    if (nb1 < 1 || nb2 < 1) return -20;
    if (M == 0 || N == 0) return 0;
    if (!A && K > 0) return -5; if (!B && K > 0) return -10; if (!C) return -16; if (ldc < N) return -17;
    if ((flags & NSGP_GEMM_A_LOWER) && (flags & NSGP_GEMM_A_UPPER)) return -22;
    if ((flags & NSGP_GEMM_B_LOWER) && (flags & NSGP_GEMM_B_UPPER)) return -22;
    const int64_t nb = nb1 * nb2;
    const bool has_epi = epi && (epi->kind != 0 || epi->ks);
    Plan p = make_plan<T>(M, N, K, nb, flags, !(has_epi && sizeof(T) == 8));   // fused variants: f64 on 64-tiles only
    GemmArgs g;
    g.M = M; g.N = N; g.K = K;
    g.sam = sam; g.sak = sak; g.sa1 = sa1; g.sa2 = sa2;
    g.sbk = sbk; g.sbn = sbn; g.sb1 = sb1; g.sb2 = sb2;
    g.ldc = ldc; g.sc1 = sc1; g.sc2 = sc2; g.nb2 = nb2;
    g.ksplit = p.ksplit; g.kper = p.kper; g.slab = nb * M * N; g.flags = flags;
    g.modeA = (sak == 1) ? 0 : (sam == 1 ? 1 : 0);
    g.modeB = (sbk == 1) ? 0 : (sbn == 1 ? 1 : 0);
    if (epi && (epi->kind != 0 || epi->ks)) { g.modeA = epi->modeA; g.modeB = epi->modeB; }
    const size_t al = 4 * sizeof(T);
    auto vec_ok = [&](const void* ptr, int64_t unit, int64_t other, int64_t s1, int64_t s2) {
        return unit == 1 && other % 4 == 0 && s1 % 4 == 0 && s2 % 4 == 0 && ((uintptr_t)ptr % al) == 0;
    };
    g.vecA = g.modeA == 0 ? vec_ok(A, sak, sam, sa1, sa2) : vec_ok(A, sam, sak, sa1, sa2);
    g.vecB = g.modeB == 0 ? vec_ok(B, sbk, sbn, sb1, sb2) : vec_ok(B, sbn, sbk, sb1, sb2);
    T* slabs = nullptr;
    if (p.ksplit > 1) {
        const size_t need = (size_t)p.ksplit * g.slab * sizeof(T);
        if (!ws || wsb < need) return -23;
        slabs = (T*)ws;
    }
    hipStream_t st = (hipStream_t)stream;
    const int64_t bmn = p.big ? 128 : 64;
    g.tiles_m = cdiv64(M, bmn);
    g.tiles_n = cdiv64(N, p.narrow ? 64 : bmn);
    if (g.tiles_m * g.tiles_n > 2147483647LL || nb * g.ksplit > 65535) return -24;
    const int64_t ngrid = (flags & NSGP_GEMM_C_LOWER) ? active_tiles(M, N, bmn, flags) : g.tiles_m * g.tiles_n;
    if ((flags & NSGP_GEMM_C_LOWER) && g.ksplit == 1 && beta == T(0)) {
        // the strictly-upper tiles are not launched: keep the "strict upper triangle is zero" contract
        // (the split-K reduce kernel writes those zeros itself)
        for (int64_t i1 = 0; i1 < nb1; ++i1)
            for (int64_t i2 = 0; i2 < nb2; ++i2) {
                hipError_t e = hipMemset2DAsync(C + i1 * sc1 + i2 * sc2, (size_t)ldc * sizeof(T), 0,
                                                (size_t)N * sizeof(T), (size_t)M, st);
                if (e != hipSuccess) return (int)e;
            }
    }
    dim3 grid((unsigned)ngrid, (unsigned)(nb * g.ksplit), 1);
    Epi ep{};
    if (epi) ep = *epi;
    if (tiles_m_out) *tiles_m_out = g.tiles_m;
    const int ekind = ep.kind, eks = ep.ks != nullptr;
    const bool one_round = ngrid * nb * g.ksplit <= 256 * 3;      // f64: latency-bound single-round grids take PF = 1
    if (ekind != 0 && (g.ksplit != 1 || beta != T(0) || (flags & NSGP_GEMM_C_LOWER))) return -30;
#define NSGP_LAUNCH_X(BMN, MA, MB, EP, KS) NSGP_LAUNCH_XY(BMN, BMN, MA, MB, EP, KS)
#define NSGP_LAUNCH_XY(BM_, BN_, MA, MB, EP, KS)                                                          \
    do {                                                                                                  \
        constexpr int BKc = (BM_ == 128 ? 32 : 16);                                                       \
        constexpr int pa = (MA == 0 && sizeof(T) == 4) ? 1 : Mfma<T>::PAD;                                \
        constexpr int pb = (MB == 0 && sizeof(T) == 4) ? 1 : Mfma<T>::PAD;                                \
        constexpr size_t lds = 2 * BKc * ((BM_ + pa) + (BN_ + pb)) * sizeof(T);                           \
        auto kern = gemm_kernel<T, BM_, BN_, BKc, MA, MB, EP, KS>;                                        \
        nsgp_opt_in_lds((const void*)kern, lds);                                                          \
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, g, alpha, A, B, beta, C, slabs, ep);           \
    } while (0)
#define NSGP_LAUNCH_EPI(BMN)                                                                              \
    do {                                                                                                  \
        if (ekind == 1 && !eks && g.modeA == 0 && g.modeB == 1) NSGP_LAUNCH_X(BMN, 0, 1, 1, 0);           \
        else if (ekind == 1 && !eks && g.modeA == 1 && g.modeB == 1) NSGP_LAUNCH_X(BMN, 1, 1, 1, 0);      \
        else if (ekind == 2 && !eks && g.modeA == 0 && g.modeB == 1) NSGP_LAUNCH_X(BMN, 0, 1, 2, 0);      \
        else if (ekind == 0 && eks && g.modeA == 0 && g.modeB == 0) NSGP_LAUNCH_X(BMN, 0, 0, 0, 1);       \
        else return -31;                                                                                  \
    } while (0)
    if (p.narrow) {
        if constexpr (sizeof(T) == 4) {
            if (eks) return -31;
            if (ekind == 1 && g.modeA == 0 && g.modeB == 1) NSGP_LAUNCH_XY(128, 64, 0, 1, 1, 0);
            else if (ekind == 1 && g.modeA == 1 && g.modeB == 1) NSGP_LAUNCH_XY(128, 64, 1, 1, 1, 0);
            else if (ekind == 2 && g.modeA == 0 && g.modeB == 1) NSGP_LAUNCH_XY(128, 64, 0, 1, 2, 0);
            else if (ekind != 0) return -31;
            else if (g.modeA == 0 && g.modeB == 0) NSGP_LAUNCH_XY(128, 64, 0, 0, 0, 0);
            else if (g.modeA == 0) NSGP_LAUNCH_XY(128, 64, 0, 1, 0, 0);
            else if (g.modeB == 0) NSGP_LAUNCH_XY(128, 64, 1, 0, 0, 0);
            else NSGP_LAUNCH_XY(128, 64, 1, 1, 0, 0);
        }
    } else if (ekind != 0 || eks) {
        if (p.big) {
            if constexpr (sizeof(T) == 4) NSGP_LAUNCH_EPI(128);
        } else {
            NSGP_LAUNCH_EPI(64);
        }
    } else {
#define NSGP_LAUNCH(BMN, MA, MB)                                                                          \
    do {                                                                                                  \
        constexpr int BKc = (BMN == 128 && sizeof(T) == 4 ? 32 : 16);                                                     \
        constexpr int pa = (MA == 0 && sizeof(T) == 4) ? 1 : Mfma<T>::PAD;                                \
        constexpr int pb = (MB == 0 && sizeof(T) == 4) ? 1 : Mfma<T>::PAD;                                \
        constexpr size_t lds = 2 * BKc * ((BMN + pa) + (BMN + pb)) * sizeof(T);                           \
        if (sizeof(T) == 8 && one_round) {                                                                \
            hipLaunchKernelGGL((gemm_kernel<T, BMN, BMN, BKc, MA, MB, 0, 0, (sizeof(T) == 8 ? 1 : 0)>), grid,   \
                               dim3(256), lds, st, g, alpha, A, B, beta, C, slabs, ep);                   \
            break;                                                                                        \
        }                                                                                                 \
        auto kern = gemm_kernel<T, BMN, BMN, BKc, MA, MB>;                                                \
        nsgp_opt_in_lds((const void*)kern, lds);                                                          \
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, g, alpha, A, B, beta, C, slabs, ep);           \
    } while (0)
#define NSGP_LAUNCH_MODES(BMN)                                                 \
    do {                                                                       \
        if (g.modeA == 0 && g.modeB == 0) NSGP_LAUNCH(BMN, 0, 0);              \
        else if (g.modeA == 0) NSGP_LAUNCH(BMN, 0, 1);                         \
        else if (g.modeB == 0) NSGP_LAUNCH(BMN, 1, 0);                         \
        else NSGP_LAUNCH(BMN, 1, 1);                                           \
    } while (0)
    if (p.big) {
        if constexpr (sizeof(T) == 4) NSGP_LAUNCH_MODES(128);
    } else {
        NSGP_LAUNCH_MODES(64);
    }
#undef NSGP_LAUNCH_MODES
#undef NSGP_LAUNCH
    }
#undef NSGP_LAUNCH_EPI
#undef NSGP_LAUNCH_X
#undef NSGP_LAUNCH_XY
    if (g.ksplit > 1) {
        const int64_t tot = nb * M * N;
        hipLaunchKernelGGL((splitk_reduce_kernel<T>), dim3((unsigned)cdiv64(tot, 256)), dim3(256), 0, st, g, alpha,
                           (const T*)slabs, beta, C, nb);
    }
    return nsgp_launch_status();
}

}  // namespace

// ---- SVGP projection GEMMs with fused epilogues (include/nsgp.h, section K6) ---------------------------
namespace {
template <typename T>
int tri_gemm_colstats_impl(const T* L, int trans, const T* X, const T* rowvec, int64_t batch, int64_t M, int64_t n,
                           T* Y, T* part_dot, T* part_sq, void* stream) {
    if (!L) return -1; if (trans != 0 && trans != 1) return -2; if (!X) return -3;
    if (batch < 0) return -5; if (M < 0) return -6; if (n < 0) return -7; if (!Y) return -8; if (!part_sq) return -10;
    if (batch == 0 || M == 0 || n == 0) return 0;
    Epi ep{};
    ep.kind = 1; ep.rv = rowvec; ep.p0 = part_dot; ep.p1 = part_sq; ep.modeA = trans ? 1 : 0; ep.modeB = 1;
    const int flags = (trans ? NSGP_GEMM_A_UPPER : NSGP_GEMM_A_LOWER) | NSGP_GEMM_NO_SPLITK;
    return gemm_impl<T>(M, n, M, T(1), L, trans ? 1 : M, trans ? M : 1, M * M, 0, X, n, 1, M * n, 0, T(0), Y, n, M * n, 0,
                        batch, 1, flags, nullptr, 0, stream, &ep);
}
template <typename T>
int abar_impl(const T* Lq, const T* C, const T* A, const T* m, const T* gmean, const T* gvar, int64_t batch, int64_t M,
              int64_t n, T* Abar, void* stream) {
    if (!Lq) return -1; if (!C) return -2; if (!A) return -3; if (!m) return -4; if (!gmean) return -5; if (!gvar) return -6;
    if (batch < 0) return -7; if (M < 0) return -8; if (n < 0) return -9; if (!Abar) return -10;
    if (batch == 0 || M == 0 || n == 0) return 0;
    Epi ep{};
    ep.kind = 2; ep.rv = m; ep.cs = gvar; ep.gc = gmean; ep.mat = A; ep.modeA = 0; ep.modeB = 1;
    return gemm_impl<T>(M, n, M, T(1), Lq, M, 1, M * M, 0, C, n, 1, M * n, 0, T(0), Abar, n, M * n, 0, batch, 1,
                        NSGP_GEMM_A_LOWER | NSGP_GEMM_NO_SPLITK, nullptr, 0, stream, &ep);
}
template <typename T>
int lqbar_impl(const T* A, const T* C, const T* gvar, int64_t batch, int64_t M, int64_t n, T* Lqbar, void* ws,
               size_t wsb, void* stream) {
    if (!A) return -1; if (!C) return -2; if (!gvar) return -3;
    if (batch < 0) return -4; if (M < 0) return -5; if (n < 0) return -6; if (!Lqbar) return -7;
    if (batch == 0 || M == 0) return 0;
    Epi ep{};
    ep.kind = 0; ep.ks = gvar; ep.modeA = 0; ep.modeB = 0;
    // Lqbar = tril(A diag(2 gvar) C^T):  B(k, j) = C[j][k], scaled along k by gvar, alpha = 2
    return gemm_impl<T>(M, M, n, T(2), A, n, 1, M * n, 0, C, 1, n, M * n, 0, T(0), Lqbar, M, M * M, 0, batch, 1,
                        NSGP_GEMM_C_LOWER, ws, wsb, stream, &ep);
}
}  // namespace

extern "C" {

size_t nsgp_gemm_workspace(int64_t M, int64_t N, int64_t K, int64_t nb1, int64_t nb2, int elem_size, int flags) {
    if (M <= 0 || N <= 0 || K <= 0 || nb1 < 1 || nb2 < 1) return 0;
    const int64_t nb = nb1 * nb2;
    const Plan p = elem_size == 4 ? make_plan<float>(M, N, K, nb, flags) : make_plan<double>(M, N, K, nb, flags);
    return p.ksplit > 1 ? (size_t)p.ksplit * nb * M * N * elem_size : 0;
}

int nsgp_gemm_f32(int64_t M, int64_t N, int64_t K, float alpha, const float* A, int64_t sam, int64_t sak,
                  int64_t sa1, int64_t sa2, const float* B, int64_t sbk, int64_t sbn, int64_t sb1, int64_t sb2,
                  float beta, float* C, int64_t ldc, int64_t sc1, int64_t sc2, int64_t nb1, int64_t nb2, int flags,
                  void* ws, size_t wsb, void* stream) {
    return gemm_impl<float>(M, N, K, alpha, A, sam, sak, sa1, sa2, B, sbk, sbn, sb1, sb2, beta, C, ldc, sc1, sc2,
                            nb1, nb2, flags, ws, wsb, stream);
}
int nsgp_gemm_f64(int64_t M, int64_t N, int64_t K, double alpha, const double* A, int64_t sam, int64_t sak,
                  int64_t sa1, int64_t sa2, const double* B, int64_t sbk, int64_t sbn, int64_t sb1, int64_t sb2,
                  double beta, double* C, int64_t ldc, int64_t sc1, int64_t sc2, int64_t nb1, int64_t nb2,
                  int flags, void* ws, size_t wsb, void* stream) {
    return gemm_impl<double>(M, N, K, alpha, A, sam, sak, sa1, sa2, B, sbk, sbn, sb1, sb2, beta, C, ldc, sc1, sc2,
                             nb1, nb2, flags, ws, wsb, stream);
}


size_t nsgp_svgp_colstats_tiles(int64_t M, int64_t n, int64_t batch, int elem_size) {
    if (M <= 0 || n <= 0 || batch <= 0) return 0;
    const int flags = NSGP_GEMM_A_LOWER | NSGP_GEMM_NO_SPLITK;
    const Plan p = elem_size == 4 ? make_plan<float>(M, n, M, batch, flags)
                                  : make_plan<double>(M, n, M, batch, flags, false);
    return (size_t)cdiv64(M, p.big ? 128 : 64);
}
int nsgp_svgp_tri_gemm_colstats_f32(const float* L, int trans, const float* X, const float* rowvec, int64_t batch,
                                    int64_t M, int64_t n, float* Y, float* part_dot, float* part_sq, void* stream) {
    return tri_gemm_colstats_impl<float>(L, trans, X, rowvec, batch, M, n, Y, part_dot, part_sq, stream);
}
int nsgp_svgp_tri_gemm_colstats_f64(const double* L, int trans, const double* X, const double* rowvec, int64_t batch,
                                    int64_t M, int64_t n, double* Y, double* part_dot, double* part_sq,
                                    void* stream) {
    return tri_gemm_colstats_impl<double>(L, trans, X, rowvec, batch, M, n, Y, part_dot, part_sq, stream);
}
int nsgp_svgp_abar_f32(const float* Lq, const float* C, const float* A, const float* m, const float* gmean,
                       const float* gvar, int64_t batch, int64_t M, int64_t n, float* Abar, void* stream) {
    return abar_impl<float>(Lq, C, A, m, gmean, gvar, batch, M, n, Abar, stream);
}
int nsgp_svgp_abar_f64(const double* Lq, const double* C, const double* A, const double* m, const double* gmean,
                       const double* gvar, int64_t batch, int64_t M, int64_t n, double* Abar, void* stream) {
    return abar_impl<double>(Lq, C, A, m, gmean, gvar, batch, M, n, Abar, stream);
}
size_t nsgp_svgp_lqbar_workspace(int64_t batch, int64_t M, int64_t n, int elem_size) {
    if (M <= 0 || n <= 0 || batch <= 0) return 0;
    const Plan p = elem_size == 4 ? make_plan<float>(M, M, n, batch, NSGP_GEMM_C_LOWER)
                                  : make_plan<double>(M, M, n, batch, NSGP_GEMM_C_LOWER, false);
    return p.ksplit > 1 ? (size_t)p.ksplit * batch * M * M * elem_size : 0;
}
int nsgp_svgp_lqbar_f32(const float* A, const float* C, const float* gvar, int64_t batch, int64_t M, int64_t n,
                        float* Lqbar, void* ws, size_t ws_bytes, void* stream) {
    return lqbar_impl<float>(A, C, gvar, batch, M, n, Lqbar, ws, ws_bytes, stream);
}
int nsgp_svgp_lqbar_f64(const double* A, const double* C, const double* gvar, int64_t batch, int64_t M, int64_t n,
                        double* Lqbar, void* ws, size_t ws_bytes, void* stream) {
    return lqbar_impl<double>(A, C, gvar, batch, M, n, Lqbar, ws, ws_bytes, stream);
}

}  // extern "C"
