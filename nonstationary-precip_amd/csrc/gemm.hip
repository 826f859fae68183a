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
#include <type_traits>
#include "common.h"

#ifndef NSGP_F64ACC_BN
#define NSGP_F64ACC_BN 64
#endif
#ifndef NSGP_F64ACC_BK
#define NSGP_F64ACC_BK 16
#endif

namespace {

#ifdef NSGP_GEMM_STAMPS
// Diagnostic build only (tools/probes/gemm_stamps.py): one 16-word record per workgroup -- shader-clock stamps at entry,
// after the prologue (first K-tile in LDS), after the K loop and after the epilogue, the 100 MHz wall clock at entry and
// exit, the tile and the hardware id.  Written to a buffer of their own; nothing else reads them.
__device__ unsigned long long* nsgp_stamp_buf = nullptr;
__device__ unsigned long long nsgp_stamp_cap = 0;
#define NSGP_STAMP(i) do { if (threadIdx.x == 0) stamp_v[i] = __builtin_amdgcn_s_memtime(); } while (0)
#define NSGP_STAMP_FLUSH() do { \
        if (threadIdx.x == 0 && nsgp_stamp_buf) { \
            const unsigned long long si = (unsigned long long)blockIdx.y * gridDim.x + blockIdx.x; \
            if (si < nsgp_stamp_cap) { \
                unsigned long long* sp = nsgp_stamp_buf + si * 16; \
                sp[0] = stamp_v[0]; sp[1] = stamp_v[1]; sp[2] = stamp_v[2]; sp[3] = __builtin_amdgcn_s_memtime(); \
                sp[4] = stamp_rt0; sp[5] = __builtin_amdgcn_s_memrealtime(); \
                sp[6] = ((unsigned long long)(unsigned)bm << 40) | ((unsigned long long)(unsigned)bn << 16) | (unsigned)nt; \
                sp[8] = stamp_bar; \
                sp[7] = ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | (unsigned)__builtin_amdgcn_s_getreg(63492); \
            } \
        } } while (0)
// barrier of the K loop with the cycles this wave waited at it added up (the s_memtime round trips cost ~2 % of a K-tile)
#define NSGP_LOOP_BARRIER() do { \
        const unsigned long long b0_ = __builtin_amdgcn_s_memtime(); \
        __syncthreads(); \
        stamp_bar += __builtin_amdgcn_s_memtime() - b0_; \
    } while (0)
#else
#define NSGP_STAMP(i) do { } while (0)
#define NSGP_STAMP_FLUSH() do { } while (0)
#define NSGP_LOOP_BARRIER() __syncthreads()
#endif

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
    static __device__ __forceinline__ int crow_lane(int lane) { return 4 * (lane >> 5); }          // crow = crow_lane + crow_reg
    static __device__ __forceinline__ constexpr int crow_reg(int r) { return (r & 3) + 8 * (r >> 2); }
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
    static __device__ __forceinline__ int crow_lane(int lane) { return lane >> 4; }
    static __device__ __forceinline__ constexpr int crow_reg(int r) { return 4 * r; }
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
    int nbk;                      // batch x K-slices (the logical extent of grid.y)
    int part_rows;                // EPI 1: tile rows per batch element in the partial buffers (>= tiles_m; 0 = tiles_m)
    int xcd_chunk;                // > 0: chunked XCD placement of a split-K launch, workgroups per XCD (see the kernel)
    int xcd_group;                // > 0: XCD-aware tile order with this many tile rows per group (see the kernel)
    int batch_perm;               // 1: single-round batched launch -- permute the tile order per batch element (see the kernel)
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
    int part_rows;          // kind 1: tile rows per batch element in the partial buffers (0 = the launch's own tile rows)
    int p64;                // kind 1, MIX != 0: p0 / p1 are float64 buffers (the float64 accumulators' sums, unrounded)
    // MIX = 2: the B operand is never read -- B(k, n) = os[b] exp(-1/2 |z_k / ls[b] - x_n / ls[b]|^2) is generated in the
    // loader from the inducing points z (batch, K, D), the inputs x (n, D) or (batch, n, D) (kx_sx = batch stride),
    // ls (batch, D), os (batch): the arithmetic of pairwise.hip's RbfOp, operation for operation
    const float *kz, *kx, *kls, *kos;
    int kD;
    int64_t kx_sx;
};
constexpr int KGEN_DMAX = 4;   // input dimensions the generated-operand loader keeps in registers

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

// Operand tile loader.  A thread owns P four-element pieces of the BR x BK operand tile (R = m for A, n for B):
//   MODE 0 (operand contiguous along k): piece p = row r0 + p * RS, columns k0 .. k0 + 3
//   MODE 1 (contiguous along r)        : piece p = rows r0 .. r0 + 3, column k0 + p * KS
// Three paths per K-tile: `fast` (whole tile in bounds, 16-byte aligned, off the diagonal: bare vector loads),
// `fast + mask` (diagonal blocks of a triangular operand: vector loads, then zeros by 32-bit compares on tile-local
// coordinates) and `edge` (ragged or unaligned tiles: scalar loads with bounds, rare).  zgt / zlt: zero where
// k > r / k < r (r the global row of A, or the global column of B).
template <typename T, int MODE, int BR, int BK, int P>
struct TileLoader {
    static constexpr int TPK = BK / 4, TPR = BR / 4;
    static constexpr int RS = MODE == 0 ? 256 / TPK : 0;      // row step between pieces
    static constexpr int KS = MODE == 0 ? 0 : 256 / TPR;      // k step between pieces
    int r0, k0;                                               // tile-local coordinates of piece 0
    const T* cur;                                             // fast-path pointer of piece 0 at the current K-tile
    int64_t pstep, kstep;                                     // element strides: between pieces, per k
    __device__ __forceinline__ void init(int tid, const T* base, int64_t row0, int64_t sr, int64_t sk, int64_t kbeg) {
        if (MODE == 0) { r0 = tid / TPK; k0 = (tid % TPK) * 4; }
        else { r0 = (tid % TPR) * 4; k0 = tid / TPR; }
        kstep = sk;
        pstep = MODE == 0 ? RS * sr : KS * sk;
        cur = base + (row0 + r0) * sr + (kbeg + k0) * sk;
    }
    __device__ __forceinline__ int prow(int p) const { return r0 + p * RS; }
    __device__ __forceinline__ int pk(int p) const { return k0 + p * KS; }
    // fast path: the whole tile is in bounds and every piece is a 16-byte aligned vector (koff = k offset from kbeg)
    __device__ __forceinline__ void load_fast(Frag4<T>* f, int64_t koff) const {
#pragma unroll
        for (int p = 0; p < P; ++p) f[p] = ldg4(cur + p * pstep + koff * kstep);
    }
    // edge path: ragged / unaligned tiles, element by element with bounds (rrem rows, krem columns are valid)
    __device__ __forceinline__ void load_edge(Frag4<T>* f, const T* base, int64_t row0, int64_t sr, int64_t sk, int64_t k0g,
                                              int rrem, int krem) const {
#pragma unroll
        for (int p = 0; p < P; ++p)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = prow(p) + (MODE == 0 ? 0 : e), k = pk(p) + (MODE == 0 ? e : 0);
                f[p].v[e] = (r < rrem && k < krem) ? base[(row0 + r) * sr + (k0g + k) * sk] : T(0);
            }
    }
    // diagonal block of a triangular operand: zero where k > r (zgt) / k < r (zlt), d = global k0 - global row0.
    // Applied per piece on its way INTO LDS (not right behind the load: that would wait for the load on the spot).
    __device__ __forceinline__ void mask_piece(Frag4<T>& f, int p, int d, bool zgt, bool zlt) const {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int r = prow(p) + (MODE == 0 ? 0 : e), k = pk(p) + (MODE == 0 ? e : 0) + d;
            const bool z = (zgt && k > r) || (zlt && k < r);
            f.v[e] = z ? T(0) : f.v[e];                          // select, not a branch
        }
    }
};

// MODE_A / MODE_B: 0 = operand contiguous along k, 1 = contiguous along m (n).
// EDGE = 0: every tile of the launch is whole and vector-loadable (M % BM == N % BN == K % BK == 0, 16-byte aligned
// operands: the host checks) -- no bounds code is compiled in; EDGE = 1: ragged / unaligned shapes, element-wise loads.
// MIX = 3: as MIX = 1 but the B operand is float64 in memory as well (Kzx BUILT in float64: layers whose output feeds another
// layer -- the rounding of a float32 Kzx, amplified by |W||Kzx| ~ 1e2, is what limits those layers' accuracy); C, rv and the
// partials stay float32.
// MIX = 1 (T = double only): float64 arithmetic on float32 data -- the B operand, the output, the row vector and the
// column-statistic partials are float32 in memory (pointers passed as T*, strides in elements); A and the MFMA
// accumulation are float64.  This is the whitened projection A = L^-1 Kzx of the SVGP layer: the reference solves it
// in float64 and rounds once; float32 accumulation of W Kzx loses 2e-4 at kappa(Kzz) ~ 1e6 (terms of size |W||K| ~ 80
// cancel to O(1); tools/probes/whiten_precision.py), float64 accumulation reproduces the float64 solve to 3e-8.
#ifndef NSGP_MIX_WG3
#define NSGP_MIX_WG3 0
#endif
#ifndef NSGP_MIX_DEEP
#define NSGP_MIX_DEEP 1
#endif
#ifndef NSGP_F32_DEEP
#define NSGP_F32_DEEP 0
#endif
#ifndef NSGP_F64_BK
#define NSGP_F64_BK 16
#endif
#ifndef NSGP_F64_DEPTH
#define NSGP_F64_DEPTH 2
#endif
#ifndef NSGP_GEMM_ROLL
#define NSGP_GEMM_ROLL 1
#endif
template <typename T, int BM, int BN, int BK, int MODE_A, int MODE_B, int EPI = 0, int KSC = 0, int PF = 0, int EDGE = 1,
          int MIX = 0>
__global__ __launch_bounds__(256, (BM == 128 && BN == 64 && (sizeof(T) == 4 || NSGP_MIX_WG3)) ? 3 : ((sizeof(T) == 8 && BM == 128 && BK == 32) ? 1 : 2)) void gemm_kernel(GemmArgs g, T alpha, const T* __restrict__ A,
                                                   const T* __restrict__ B, T beta, T* __restrict__ C,
                                                   T* __restrict__ slabs, Epi ep) {
    static_assert(MIX == 0 || (sizeof(T) == 8 && EPI == 1 && KSC == 0 && PF == 0), "MIX: float64 colstats projection only");
    static_assert(MIX != 2 || (EDGE == 0 && MODE_B == 1), "MIX = 2 (generated Kzx operand): whole tiles, n-contiguous pieces");
    using TB = std::conditional_t<MIX == 1 || MIX == 2, float, T>;   // element type of B in memory (MIX = 3: float64 B, float32 C)
    using TC = std::conditional_t<MIX != 0, float, T>;          // element type of C / rv / partials in memory
    using MF = Mfma<T>;
    constexpr int MT = MF::MT, KS = MF::KS, NKK = BK / KS;
    constexpr int WM = BM / 2, WN = BN / 2;
    constexpr int TM = WM / MT, TN = WN / MT;
    constexpr int PADA = (MODE_A == 0 && sizeof(T) == 4) ? 1 : MF::PAD;
    constexpr int PADB = (MODE_B == 0 && sizeof(T) == 4) ? 1 : MF::PAD;
    constexpr int LDA = BM + PADA, LDB = BN + PADB;
    constexpr int PA = BM * BK / 1024, PB = BN * BK / 1024;     // 4-element pieces per thread
    // all LDS in ONE dynamic region (the 128x128x32 f32 tile needs 66 KB > the 64 KB static limit)
    extern __shared__ __attribute__((aligned(32))) unsigned char gemm_smem[];
    T (*As)[BK * LDA] = reinterpret_cast<T (*)[BK * LDA]>(gemm_smem);
    T (*Bs)[BK * LDB] = reinterpret_cast<T (*)[BK * LDB]>(gemm_smem + 2 * BK * LDA * sizeof(T));

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;
#ifdef NSGP_GEMM_STAMPS
    unsigned long long stamp_v[4] = {0, 0, 0, 0};
    unsigned long long stamp_bar = 0;                    // cycles wave 0 spent at the K loop's barriers
    const unsigned long long stamp_rt0 = __builtin_amdgcn_s_memrealtime();
    NSGP_STAMP(0);
#endif
    // 32-bit tile arithmetic (the host checks tiles_m * tiles_n < 2^31): the 64-bit divisions of the first version
    // cost every workgroup ~1 us of scalar code before its first load.
    const int tiles_m = (int)g.tiles_m, tiles_n = (int)g.tiles_n;
    int tid_lin = (int)blockIdx.x, zz = (int)blockIdx.y;
    if (g.batch_perm == 3) {
        // one matrix, one round: workgroups i, i + 256, i + 512 share a CU (ids are dealt to the 256 CUs in turn) and the
        // tile order is longest-first, so CU 0 got the longest tile of every block of 256.  Every second block of 256 runs
        // backwards (snake): a CU's tiles have complementary lengths.  Speed only.
        const int ntile = (int)gridDim.x;
        const int blk = tid_lin >> 8;
        if (blk & 1) {
            const int base = blk << 8;
            const int top = base + 255 < ntile - 1 ? base + 255 : ntile - 1;
            tid_lin = base + (top - tid_lin);
        }
    } else if (g.batch_perm) {
        // A launch whose workgroups are all resident at once places workgroup i of EVERY batch element on the same CU
        // (ids i, i + n, i + 2n are dealt to the same XCD and CU in turn), so with a triangular operand a CU gets the
        // longest tile of each matrix and another the shortest of each: the launch takes as long as that one CU (Cholesky
        // adjoint, 3 x 1024^2: 3 x 64 K-tiles on one CU against a mean of 70).  Odd batch elements walk the tiles backwards
        // and every second pair starts half way round, so a CU's tiles have complementary lengths.  Speed only.
        const int ntile = (int)gridDim.x;
        int t = tid_lin + ((zz >> 1) & 1) * (ntile >> 1);
        if (t >= ntile) t -= ntile;
        tid_lin = (zz & 1) ? ntile - 1 - t : t;
        if (g.batch_perm == 2) {
            // more than one round: batch element fastest, so that the longest-first tile order holds across the whole launch
            // (with the batch on grid.y the second matrix's long tiles queued behind the first one's short ones)
            const int lin = zz * ntile + (int)blockIdx.x;
            tid_lin = lin / g.nbk;
            zz = lin % g.nbk;
        }
    }
    if (g.xcd_chunk > 0) {
        // Long-K products with a small output (Lqbar = A diag(v) C^T, Wbar = Abar Kzx^T; split along K into slabs): every
        // (tile, slice) workgroup streams a 128-row panel of both operands once, and the tiles of one slice read the SAME
        // k range.  Dealt round-robin to the XCDs they shared nothing below the Infinity Cache (4.5x the operand bytes
        // through the fabric, ~5 TB/s: these launches were bandwidth-bound).  Here XCD x takes the contiguous chunk
        // [x * per, (x + 1) * per) of the slice-major order, so the tiles of a slice sit on one XCD (two at a chunk
        // boundary), walk k in step and share each operand K-tile through that L2.  Speed only.
        const int L = zz * (int)gridDim.x + tid_lin;
        const int logical = (L & 7) * g.xcd_chunk + (L >> 3);
        const int ntile = (int)gridDim.x;
        tid_lin = logical % ntile;
        zz = logical / ntile;
        if (zz >= g.nbk) return;                          // padding of the last chunk
    }
    int bm, bn;
    if (g.flags & NSGP_GEMM_C_LOWER) {
        // compact enumeration of the ACTIVE tiles only (n0 <= m0 + BM - 1): launching the strictly-upper
        // tiles as no-op workgroups perturbs the dispatcher's CU placement (measured: lower-only product
        // no faster than the full one, CUs ~58 % busy).  Rows 0..T-1 form a triangle, the rest are full.
        const int TT = tiles_m < tiles_n ? tiles_m : tiles_n;
        const int tri = TT * (TT + 1) / 2;
        // triangular operands on top (Cholesky adjoint: tril(Wbar W^T), Phi W): the K range grows with the row, issue the
        // large rows first (the mirror flips below would leave the lower triangle)
        if (g.flags & (NSGP_GEMM_A_LOWER | NSGP_GEMM_B_UPPER)) tid_lin = (int)gridDim.x - 1 - tid_lin;
        if (tid_lin < tri) {
            int r = (int)((sqrtf(8.0f * (float)tid_lin + 1.0f) - 1.0f) * 0.5f);
            while (r * (r + 1) / 2 > tid_lin) --r;
            while ((r + 1) * (r + 2) / 2 <= tid_lin) ++r;
            bm = r; bn = tid_lin - r * (r + 1) / 2;
        } else {
            const int q = tid_lin - tri;
            bm = TT + q / tiles_n; bn = q % tiles_n;
        }
    } else if (g.xcd_group > 0) {
        // XCD-aware order for a triangular A operand (SVGP projections, n >> M): workgroups are dealt round-robin to
        // the 8 XCDs (id % 8), each with its own 4 MiB L2.  Logical order = column panel major, tile rows fastest
        // within a row group, and XCD x owns the column panels  bn = 8 * (j / R) + x : the R row tiles of a column
        // panel run on ONE XCD at the same time and walk k in step, so a K-tile of the B panel is fetched once from
        // the fabric and hit R - 1 times in that L2 (the bn-fastest order re-fetched every B panel once per tile row:
        // 4.5x the operand bytes, profiles/r01/gemm_hbm_pmc_v3.txt).  Row groups run longest K range first (the
        // tail of the launch is made of short tiles).  Speed only: correctness does not depend on the placement.
        const int R = g.xcd_group;                       // rows per group (divides tiles_m)
        const int x = tid_lin & 7, j = tid_lin >> 3;
        const int panels8 = (tiles_n + 7) >> 3;          // column panels per XCD
        const int per_group = panels8 * R;               // tiles per XCD and row group
        const int grp = j / per_group, jj = j - grp * per_group;
        bn = 8 * (jj / R) + x;
        bm = grp * R + (jj % R);
        if (bn >= tiles_n) return;                       // padding of the last 8-panel block
    } else {
        const int brow = tid_lin / tiles_n;
        // (j - row) mod tiles_n: spreads a triangular B operand's heavy columns over the XCDs (id % 8)
        bm = brow; bn = (tid_lin % tiles_n + tiles_n - brow % tiles_n) % tiles_n;
    }
    if (!(g.flags & NSGP_GEMM_C_LOWER)) {
        if (g.flags & NSGP_GEMM_A_LOWER) bm = tiles_m - 1 - bm;        // large m = long K range
        if (g.flags & NSGP_GEMM_B_UPPER) bn = tiles_n - 1 - bn;        // large n = long K range
    }
    const int ksp = (int)g.ksplit, nb2i = (int)g.nb2;
    const int64_t slice = zz % ksp, bb = zz / ksp;
    const int64_t b1 = (int)bb / nb2i, b2 = (int)bb % nb2i;
    const int64_t m0 = (int64_t)bm * BM, n0 = (int64_t)bn * BN;

    const T* Ab = A + b1 * g.sa1 + b2 * g.sa2;
    const TB* Bb = reinterpret_cast<const TB*>(B) + b1 * g.sb1 + b2 * g.sb2;
    TC* Cb = reinterpret_cast<TC*>(C) + b1 * g.sc1 + b2 * g.sc2;

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
    const int nt = kend > kbeg ? (int)((kend - kbeg + BK - 1) / BK) : 0;

    typename MF::acc_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < MF::NREG; ++r) acc[i][j][r] = T(0);

    TileLoader<T, MODE_A, BM, BK, PA> la;
    TileLoader<TB, MODE_B, BN, BK, PB> lb;
    la.init(tid, Ab, m0, g.sam, g.sak, kbeg);
    lb.init(tid, Bb, n0, g.sbn, g.sbk, kbeg);
    // MIX = 2: this thread's four columns x_n / ls (fixed for the whole K loop), the GP's lengthscales and output scale
    [[maybe_unused]] float kgxs[4][KGEN_DMAX], kgls[KGEN_DMAX], kgos = 0.f;
    [[maybe_unused]] const float* kgz = nullptr;
    [[maybe_unused]] int kgD = 0;
    if constexpr (MIX == 2) {
        kgD = ep.kD;
        kgz = ep.kz + bb * g.K * kgD;
        kgos = ep.kos[bb];
#pragma unroll
        for (int d = 0; d < KGEN_DMAX; ++d) kgls[d] = d < kgD ? ep.kls[bb * kgD + d] : 1.f;
        const float* xb = ep.kx + bb * ep.kx_sx;
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int d = 0; d < KGEN_DMAX; ++d)
                kgxs[e][d] = d < kgD ? xb[(n0 + lb.prow(0) + e) * kgD + d] / kgls[d] : 0.f;
    }
    const int mrem = (int)(g.M - m0 < BM ? g.M - m0 : BM), nrem = (int)(g.N - n0 < BN ? g.N - n0 : BN);

    Frag4<T> ra0[PA];
    Frag4<TB> rb0[PB];
    Frag4<T> rks[KSC ? PB : 1];
    const T* ksb = KSC ? reinterpret_cast<const T*>(ep.ks) + bb * g.K : nullptr;
    const bool ks_vec = KSC && ((uintptr_t)ksb % (4 * sizeof(T)) == 0);

    // per staging register set (the float64 loop keeps two tiles in flight): is the held tile a diagonal block of A / B, and its
    // k0 - row0 offsets
    // DEEP > 0 (plain float64 products on whole tiles): the K loop keeps DEEP register sets of loads in flight, see below
    constexpr int DEEP = (sizeof(T) == 8 && KSC == 0 && EDGE == 0 && PF == 0 && ((MIX == 0 && EPI == 0) || (MIX != 0 && NSGP_MIX_DEEP))) ? NSGP_F64_DEPTH
                         : ((sizeof(T) == 4 && KSC == 0 && EDGE == 0 && NSGP_F32_DEEP > 0) ? NSGP_F32_DEEP : 0);
    constexpr int NSET = DEEP > 2 ? DEEP : 2;
    bool st_adiag[NSET] = {}, st_bdiag[NSET] = {};
    int st_ad[NSET] = {}, st_bd[NSET] = {};
    auto load_ks = [&](int64_t k0) __attribute__((always_inline)) {
        if (MODE_B == 0) {
            // every piece of this thread covers the same 4 k's: ONE load
            const int64_t k = k0 + lb.k0;
            if (ks_vec && k + 3 < kend) {
                rks[0] = ldg4(ksb + k);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) rks[0].v[e] = k + e < kend ? ksb[k + e] : T(0);
            }
        } else {
#pragma unroll
            for (int p = 0; p < PB; ++p) {
                const int64_t k = k0 + lb.pk(p);
                rks[p].v[0] = k < kend ? ksb[k] : T(0);
            }
        }
    };
    auto gload = [&](int t, Frag4<T>* ra, Frag4<TB>* rb, int set = 0) __attribute__((always_inline)) {
        const int64_t k0 = kbeg + (int64_t)t * BK;
        if constexpr (EDGE == 0) {
            la.load_fast(ra, (int64_t)t * BK);
            lb.load_fast(rb, (int64_t)t * BK);
        } else {
            const int64_t left = kend - k0;
            const int krem = (int)(left < BK ? left : BK);
            la.load_edge(ra, Ab, m0, g.sam, g.sak, k0, mrem, krem);
            lb.load_edge(rb, Bb, n0, g.sbn, g.sbk, k0, nrem, krem);
        }
        // diagonal blocks of triangular operands are zeroed when the pieces are stored (store_piece), so that nothing
        // waits for these loads here
        st_adiag[set] = (aL || aU) && (k0 < m0 + BM) && (k0 + BK > m0);
        st_bdiag[set] = (bL || bU) && (k0 < n0 + BN) && (k0 + BK > n0);
        st_ad[set] = (int)(k0 - m0);
        st_bd[set] = (int)(k0 - n0);
        if constexpr (KSC != 0) load_ks(k0);
    };
    // MIX = 2: turn a staged piece of z rows into the four operand values B(k, n .. n + 3), in two halves so that the
    // arithmetic (a division per coordinate, two exp) rides under two different MFMA groups of the K-tile
    [[maybe_unused]] float kzs[PB][KGEN_DMAX];
    [[maybe_unused]] Frag4<TB> kval[PB];
    auto kgen_half = [&](Frag4<TB>* rb, int half) __attribute__((always_inline)) {
        if constexpr (MIX == 2) {
#pragma unroll
            for (int p = 0; p < PB; ++p) {
                if (half == 0) {
#pragma unroll
                    for (int d = 0; d < KGEN_DMAX; ++d) kzs[p][d] = d < kgD ? rb[p].v[d] / kgls[d] : 0.f;
                }
#pragma unroll
                for (int e = 2 * half; e < 2 * half + 2; ++e) {
                    float ex = 0.f;
#pragma unroll
                    for (int d = 0; d < KGEN_DMAX; ++d) {
                        const float df = kzs[p][d] - kgxs[e][d];
                        ex = __builtin_fmaf(df, df, ex);
                    }
                    kval[p].v[e] = kgos * t_fexp(-0.5f * ex);
                }
                if (half == 1) rb[p] = kval[p];
            }
        }
    };
    // one four-element piece of the staged tile -> LDS (k-major images As[k][m], Bs[k][n])
    constexpr int NPIECE = PA + PB;
    auto store_piece = [&](int buf, int q, Frag4<T>* ra, Frag4<TB>* rb, auto masked_c, int set = 0) __attribute__((always_inline)) {
        constexpr bool MASKED = decltype(masked_c)::value;
        if (q < PA) {
            const int p = q;
            if constexpr (MASKED) {
                if (st_adiag[set]) la.mask_piece(ra[p], p, st_ad[set], aL, aU);
            }
            T* dst = &As[buf][la.pk(p) * LDA + la.prow(p)];
#pragma unroll
            for (int e = 0; e < 4; ++e) dst[MODE_A == 0 ? e * LDA : e] = ra[p].v[e];
        } else {
            const int p = q - PA;
            if constexpr (MASKED) {          // B(k, n): "lower" zero where n > k <=> k < r ; "upper" zero where k > r
                if (st_bdiag[set]) lb.mask_piece(rb[p], p, st_bd[set], bU, bL);
            }
            if constexpr (KSC != 0) {
#pragma unroll
                for (int e = 0; e < 4; ++e) rb[p].v[e] *= (TB)(MODE_B == 0 ? rks[0].v[e] : rks[p].v[0]);
            }
            T* dst = &Bs[buf][lb.pk(p) * LDB + lb.prow(p)];
#pragma unroll
            for (int e = 0; e < 4; ++e) dst[MODE_B == 0 ? e * LDB : e] = (T)rb[p].v[e];
        }
    };
    auto sstore = [&](int buf, Frag4<T>* ra, Frag4<TB>* rb, int set = 0) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < NPIECE; ++q) store_piece(buf, q, ra, rb, std::true_type{}, set);
    };

    const int kr = MF::krow(lane), mc = MF::mcol(lane);
    // ---- software-pipelined K-tile (the main loop of every variant) ------------------------------------------------
    // One wave must keep its SIMD's matrix pipe busy on its own: the two workgroups of a CU run the same program and
    // fall into step, so whatever one wave exposes (LDS read latency in front of every k-step, the staging stores
    // and the barrier at the end of a K-tile) the other exposes at the same moment (round 1: pipe 66-73 % busy,
    // hipcc had sunk every fragment read to just before its MFMAs behind an lgkmcnt(0)).  Here, per k-step:
    //   fragments of k-step ks+1 are read into the OTHER register set  ->  one piece of the next K-tile's staging
    //   stores (second half of the tile only, so the global loads issued at its start have landed)  ->  the MFMAs of
    //   k-step ks.  sched_barrier(0) pins that order; the waits hipcc inserts are then counted (the reads a k-step
    //   consumes are a whole k-step old).
    // `tmask` (bit i * TN + j): which MFMA tiles of the wave's TM x TN this K-tile touches -- diagonal blocks of
    // triangular operands / lower-only outputs skip the MFMA tiles that lie entirely in the zero part.
    constexpr int KS0 = NKK / 2;                                     // first k-step that carries staging stores
    constexpr int PPS = (NPIECE + (NKK - KS0) - 1) / (NKK - KS0);    // pieces per k-step
    constexpr int FULLMASK = (1 << (TM * TN)) - 1;
    auto ktile = [&](int buf, bool stage, auto tmask_c, auto masked_c, Frag4<T>* ra, Frag4<TB>* rb, int set = 0) __attribute__((always_inline)) {
        constexpr int tmask = decltype(tmask_c)::value;          // compile-time: a runtime mask makes every MFMA conditional
        const T* as = &As[buf][kr * LDA + wm0 + mc];             // and hipcc then keeps two copies of the accumulators
        const T* bs = &Bs[buf][kr * LDB + wn0 + mc];
        T af[2][TM], bf[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[0][i] = as[i * MT];
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[0][j] = bs[j * MT];
#pragma unroll
        for (int ks = 0; ks < NKK; ++ks) {
            const int cur = ks & 1, nxt = cur ^ 1;
            if (ks + 1 < NKK) {
#pragma unroll
                for (int i = 0; i < TM; ++i) af[nxt][i] = as[(ks + 1) * KS * LDA + i * MT];
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[nxt][j] = bs[(ks + 1) * KS * LDB + j * MT];
            }
            if constexpr (MIX == 2) {
                static_assert(MIX != 2 || KS0 >= 2, "generated operand: two k-steps ahead of the staging stores");
                if (ks < 2 && stage) kgen_half(rb, ks);
            }
            if (ks >= KS0 && stage) {
#pragma unroll
                for (int q = (ks - KS0) * PPS; q < (ks - KS0 + 1) * PPS && q < NPIECE; ++q) store_piece(buf ^ 1, q, ra, rb, masked_c, set);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    if constexpr ((tmask >> 0) != 0) {
                        if ((tmask >> (i * TN + j)) & 1) acc[i][j] = MF::mma(af[cur][i], bf[cur][j], acc[i][j]);
                    }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // MFMA-tile masks.  Block row q = wm0 / MT + i covers rows [m0 + q MT, m0 + (q + 1) MT); block column likewise.
    int cmask_tile = FULLMASK;                       // lower-only output: tiles strictly above the diagonal are skipped
    if (cL) {
        cmask_tile = 0;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                if (n0 + wn0 + j * MT <= m0 + wm0 + (i + 1) * MT - 1) cmask_tile |= 1 << (i * TN + j);
    }
    auto tile_mask = [&](int64_t k0) __attribute__((always_inline)) {
        int mk = cmask_tile;
        if (aL || aU) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int64_t r0 = m0 + wm0 + i * MT;
                const bool on = aL ? (k0 < r0 + MT) : (k0 + BK > r0);        // k <= row  /  k >= row
                if (!on) mk &= ~(((1 << TN) - 1) << (i * TN));
            }
        }
        if (bL || bU) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int64_t c0 = n0 + wn0 + j * MT;
                const bool on = bU ? (k0 < c0 + MT) : (k0 + BK > c0);        // B(k, n): upper k <= n, lower k >= n
                if (!on) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) mk &= ~(1 << (i * TN + j));
                }
            }
        }
        return mk;
    };
    if constexpr (DEEP > 0) {
        // Latency-bound float64 products (the whitening chain: trtri levels, the Cholesky adjoint -- 3 x 1024^2, one
        // round of 64 x 64 tiles, K ranges cut short by TWO triangular operands): a K-tile is 16 MFMAs per wave (0.4 us)
        // while a load from the Infinity Cache takes 1-2 us, and the diagonal blocks at either end of a K range used to go
        // through the one-tile-ahead generic loop.  Here EVERY K-tile (all are whole: EDGE == 0) runs in one loop with
        // DEEP register sets: tile t issues the loads of tile t + DEEP into the set tile t came from and stores tile t + 1
        // (loaded DEEP - 1 tiles ago) to LDS, diagonal blocks masked as they are stored.  All MFMA tiles run (the masked
        // parts multiply zeros).  Unrolled by DEEP so the set indices are static.
        static_assert(DEEP % 2 == 0, "the LDS buffer index is static only for an even depth");
        if (nt > 0) {
            Frag4<T> rsa[DEEP][PA];
            Frag4<TB> rsb[DEEP][PB];
            auto issue = [&](int t, int set) __attribute__((always_inline)) {
                la.load_fast(rsa[set], (int64_t)t * BK);
                if constexpr (MIX == 2) {
                    // rows of z for this thread's pieces (one k each), parked in the staging registers of B
#pragma unroll
                    for (int p = 0; p < PB; ++p) {
                        const int64_t k = kbeg + (int64_t)t * BK + lb.pk(p);
#pragma unroll
                        for (int d = 0; d < KGEN_DMAX; ++d) rsb[set][p].v[d] = d < kgD ? kgz[k * kgD + d] : 0.f;
                    }
                } else {
                    lb.load_fast(rsb[set], (int64_t)t * BK);
                }
                const int64_t k0 = kbeg + (int64_t)t * BK;
                st_adiag[set] = (aL || aU) && (k0 < m0 + BM) && (k0 + BK > m0);
                st_bdiag[set] = (bL || bU) && (k0 < n0 + BN) && (k0 + BK > n0);
                st_ad[set] = (int)(k0 - m0);
                st_bd[set] = (int)(k0 - n0);
            };
            issue(0, 0);
#pragma unroll
            for (int d = 1; d < DEEP; ++d)
                if (d < nt) issue(d, d);
            kgen_half(rsb[0], 0);
            kgen_half(rsb[0], 1);
            sstore(0, rsa[0], rsb[0], 0);
            __syncthreads();
            NSGP_STAMP(1);
            for (int tb = 0; tb < nt; tb += DEEP) {
#pragma unroll
                for (int j = 0; j < DEEP; ++j) {
                    const int t = tb + j;
                    if (t >= nt) break;
                    if (t + DEEP < nt) issue(t + DEEP, j);
                    ktile(j & 1, t + 1 < nt, std::integral_constant<int, FULLMASK>{}, std::true_type{}, rsa[(j + 1) % DEEP],
                          rsb[(j + 1) % DEEP], (j + 1) % DEEP);
                    NSGP_LOOP_BARRIER();
                }
            }
        }
    } else {
        if (nt > 0) {
            // Regular K-tiles: full BK columns, no diagonal block of a triangular operand, vector-loadable.  They form one
            // contiguous range [r0, r1) (diagonal blocks sit at one end of the K range, a ragged tile at the end).  The hot
            // loop runs the tiles t of [r0, r1 - 1): tile t itself AND the tile t + 1 it stages are regular, so its body has
            // no branch at all -- bare vector loads, every MFMA, every staging store.  With the choice between loader paths /
            // MFMA masks inside ONE loop hipcc carried the accumulators and the staging registers through phi copies (two
            // register sets, 64 v_mov per K-tile, spills inside the loop).
            // ROLL (float32 kernels, NSGP_GEMM_ROLL): the hot loop also takes the diagonal blocks of triangular operands -- they
            // are masked as they are stored, and a wave whose rows / columns lie entirely in an operand's zero part for a
            // stretch of K-tiles runs a copy of the loop without the MFMAs for that stretch (below) -- so "regular" only means
            // "whole".  (A deeper prefetch was tried on top: the load of the K-tile after next issued INTO the registers whose
            // contents the ds_write in front of it had just sent to LDS -- in-place inline-asm loads with hand-counted vmcnt,
            // because the compiler hoists such a load into fresh registers and copies them back behind a wait.  One wave alone
            // on its SIMD went from 5400 to 5280 cycles per 4096-cycle K-tile: the stall is not in the loads; and hipcc may
            // copy an asm-loaded register before the hand-placed wait -- it did in the 64 x 64 kernel's last tile, wrong
            // results.  Dropped.)
            constexpr bool ROLL = NSGP_GEMM_ROLL && sizeof(T) == 4 && KSC == 0;
            auto regular = [&](int t) __attribute__((always_inline)) {
                const int64_t k0 = kbeg + (int64_t)t * BK;
                if (k0 + BK > kend) return false;
                if constexpr (!ROLL) {
                    if ((aL || aU) && (k0 < m0 + BM) && (k0 + BK > m0)) return false;
                    if ((bL || bU) && (k0 < n0 + BN) && (k0 + BK > n0)) return false;
                }
                return true;
            };
            int r0 = 0, r1 = 0;
            // (waves of a diagonal output tile of a lower-only product take the hot loop too and compute their whole 64 x 64
            // block: sending them through the one-tile-ahead generic loop to skip the MFMA tiles above the diagonal held
            // back the whole workgroup -- Wbar = tril(Abar Kzx^T): 482 -> 456 us)
            // EDGE = 1 (ragged or unaligned launch): a workgroup whose tile lies inside M and N, on 16-byte aligned
            // operands, still runs its whole K-tiles through the hot loop; only the ragged last K-tile and the boundary
            // workgroups pay for the bounds code (n = 40000 instead of 40960 used to halve the rate of every launch)
            // (not for the k-contiguous x k-contiguous variants -- the long-K split products: with the bounds code next to it
            // their hot loop spills, 1365 -> 1485 us at M = 1000, n = 40000)
            const bool interior = EDGE == 0 || (!(MODE_A == 0 && MODE_B == 0) && g.vecA && g.vecB && m0 + BM <= g.M &&
                                                n0 + BN <= g.N);
            if (interior) {
                while (r0 < nt && !regular(r0)) ++r0;
                r1 = r0;
                while (r1 < nt && regular(r1)) ++r1;
            }
            const int h0 = r0, h1 = r1 - 1 > r0 ? r1 - 1 : r0;             // hot range [h0, h1)
            auto generic_tiles = [&](int tb, int te) __attribute__((always_inline)) {
                for (int t = tb; t < te; ++t) {
                    const int buf = t & 1;
                    const bool more = t + 1 < nt;
                    if (more) gload(t + 1, ra0, rb0);                      // next tile -> registers (in flight)
                    const int mk = __builtin_amdgcn_readfirstlane(tile_mask(kbeg + (int64_t)t * BK));
                    // Two specialisations only -- all MFMA tiles, or none: a partly needed K-tile runs the full set (the
                    // skipped MFMA tiles only ever multiply the zeros the loader wrote).
                    if (mk == 0) ktile(buf, more, std::integral_constant<int, 0>{}, std::true_type{}, ra0, rb0);
                    else ktile(buf, more, std::integral_constant<int, FULLMASK>{}, std::true_type{}, ra0, rb0);
                    NSGP_LOOP_BARRIER();
                }
            };
            gload(0, ra0, rb0);
            sstore(0, ra0, rb0);
            __syncthreads();
            NSGP_STAMP(1);
            generic_tiles(0, h0);
            // float64 tiles are short (BK = 16: 2048 MFMA cycles per K-tile, the staging stores start after 1024): a load
            // issued at the top of a tile has not landed when its first store comes up (wait_any 20 % of wave time).
            // DEPTH2 keeps TWO K-tiles of loads in flight in two register sets: tile t stores the set loaded during tile
            // t - 1 and issues the loads of tile t + 2.
            constexpr bool DEPTH2 = sizeof(T) == 8 && KSC == 0;
            int hend = h0;                                                   // first tile the hot loop did NOT run
            if constexpr (ROLL) {
                // tiles t of [h0, h1): tile t + 1 is whole.  Tile t loads tile t + 1 at its top and stores it piece by piece in
                // its second half, diagonal blocks of triangular operands masked on the way (a uniform branch around the
                // select code; the MFMA part of the body is the same for every tile).
                if (h1 > h0) {
                    // K-tiles [lo, hi) in which THIS WAVE has work: its 64 rows against a triangular A, its columns against a
                    // triangular B, nothing at all for a wave above the diagonal of a lower-only output.  Outside that range
                    // the wave stages operands and keeps the barriers but issues no MFMA (a second copy of the loop): its SIMD
                    // goes to the co-resident workgroup.  (The one-tile-ahead generic loop that used to take the diagonal K-tiles
                    // ran them 40 % slower than the hot loop: 4 of a tile's 4..32 K-tiles, 8 us of a 92 us workgroup --
                    // tools/probes/gemm_stamps.py.)
                    int lo = h0, hi = h1;
                    {
                        const int64_t R0 = m0 + wm0, C0 = n0 + wn0;
                        if (aL) { const int64_t e = (R0 + WM - kbeg + BK - 1) / BK; if (e < hi) hi = (int)(e > 0 ? e : 0); }
                        if (aU) { const int64_t b = R0 > kbeg ? (R0 - kbeg) / BK : 0; if (b > lo) lo = (int)b; }
                        if (bU) { const int64_t e = (C0 + WN - kbeg + BK - 1) / BK; if (e < hi) hi = (int)(e > 0 ? e : 0); }
                        if (bL) { const int64_t b = C0 > kbeg ? (C0 - kbeg) / BK : 0; if (b > lo) lo = (int)b; }
                        if (cmask_tile == 0) { lo = h0; hi = h0; }
                        if (lo > h1) lo = h1;
                        if (hi < lo) hi = lo;
                        lo = __builtin_amdgcn_readfirstlane(lo);
                        hi = __builtin_amdgcn_readfirstlane(hi);
                    }
                    const T* pa = la.cur + (int64_t)(h0 + 1) * BK * la.kstep;
                    const TB* pb = lb.cur + (int64_t)(h0 + 1) * BK * lb.kstep;
                    const int64_t da = (int64_t)BK * la.kstep, db = (int64_t)BK * lb.kstep;
                    auto hot_seg = [&](int tb, int te, auto tmask_c, auto masked_c) __attribute__((always_inline)) {
                        for (int t = tb; t < te; ++t) {
#pragma unroll
                            for (int p = 0; p < PA; ++p) ra0[p] = ldg4(pa + p * la.pstep);
#pragma unroll
                            for (int p = 0; p < PB; ++p) rb0[p] = ldg4(pb + p * lb.pstep);
                            pa += da; pb += db;
                            if constexpr (decltype(masked_c)::value) {     // is the tile being staged a diagonal block?
                                const int64_t k0 = kbeg + (int64_t)(t + 1) * BK;
                                st_adiag[0] = (aL || aU) && (k0 < m0 + BM) && (k0 + BK > m0);
                                st_bdiag[0] = (bL || bU) && (k0 < n0 + BN) && (k0 + BK > n0);
                                st_ad[0] = (int)(k0 - m0);
                                st_bd[0] = (int)(k0 - n0);
                            }
                            ktile(t & 1, true, tmask_c, masked_c, ra0, rb0);
                            NSGP_LOOP_BARRIER();
                        }
                    };
                    // tiles [p0, p1): the tile they STAGE (t + 1) is no diagonal block -- the bare body, as before; the few
                    // tiles at either end stage through the masking copy.  Cut [h0, h1) at p0, p1, lo, hi: four copies of
                    // the loop (MFMAs on / off x masking on / off), picked per stretch.
                    int p0 = h0, p1 = h1;
                    {
                        auto first_k = [&](int t) { return kbeg + (int64_t)(t + 1) * BK; };
                        // a diagonal block of A covers k in [m0, m0 + BM), of B k in [n0, n0 + BN): staged tiles whose k range meets
                        // them lie at the ends of the K range (or nowhere)
                        int64_t dlo = INT64_MAX, dhi = INT64_MIN;                    // k range covered by diagonal blocks
                        if (aL || aU) { dlo = m0; dhi = m0 + BM; }
                        if (bL || bU) { if (n0 < dlo) dlo = n0; if (n0 + BN > dhi) dhi = n0 + BN; }
                        if (dhi > dlo) {
                            // staged tile t + 1 is plain iff first_k + BK <= dlo or first_k >= dhi.  With both ends possible the plain
                            // stretch is taken as the longer of the two sides; the rest goes through the masking copy (which is
                            // correct for every tile).
                            int below = h0, above = h1;                              // tiles [h0, below) are plain below; [above, h1) plain above
                            while (below < h1 && first_k(below) + BK <= dlo) ++below;
                            while (above > h0 && first_k(above - 1) >= dhi) --above;
                            if (below - h0 >= h1 - above) { p0 = h0; p1 = below; } else { p0 = above; p1 = h1; }
                        }
                        p0 = __builtin_amdgcn_readfirstlane(p0);
                        p1 = __builtin_amdgcn_readfirstlane(p1);
                    }
                    int cut[6] = {h0, p0, p1, lo, hi, h1};
#pragma unroll
                    for (int a = 1; a < 6; ++a)                                       // insertion sort of six scalars
#pragma unroll
                        for (int b = a; b > 0; --b)
                            if (cut[b] < cut[b - 1]) { const int tmp = cut[b]; cut[b] = cut[b - 1]; cut[b - 1] = tmp; }
#pragma unroll 1
                    for (int a = 0; a < 5; ++a) {
                        const int tb = cut[a], te = cut[a + 1];
                        if (tb >= te) continue;
                        const bool active = tb >= lo && te <= hi, plain = tb >= p0 && te <= p1;
                        if (active && plain) hot_seg(tb, te, std::integral_constant<int, FULLMASK>{}, std::false_type{});
                        else if (active) hot_seg(tb, te, std::integral_constant<int, FULLMASK>{}, std::true_type{});
                        else if (plain) hot_seg(tb, te, std::integral_constant<int, 0>{}, std::false_type{});
                        else hot_seg(tb, te, std::integral_constant<int, 0>{}, std::true_type{});
                    }
                    hend = h1;
                }
            }
            if (!ROLL && !DEPTH2 && h1 > h0) {
                const T* pa = la.cur + (int64_t)(h0 + 1) * BK * la.kstep;
                const TB* pb = lb.cur + (int64_t)(h0 + 1) * BK * lb.kstep;
                const int64_t da = (int64_t)BK * la.kstep, db = (int64_t)BK * lb.kstep;
                for (int t = h0; t < h1; ++t) {
#pragma unroll
                    for (int p = 0; p < PA; ++p) ra0[p] = ldg4(pa + p * la.pstep);
#pragma unroll
                    for (int p = 0; p < PB; ++p) rb0[p] = ldg4(pb + p * lb.pstep);
                    if constexpr (KSC != 0) load_ks(kbeg + (int64_t)(t + 1) * BK);
                    pa += da; pb += db;
                    ktile(t & 1, true, std::integral_constant<int, FULLMASK>{}, std::false_type{}, ra0, rb0);
                    NSGP_LOOP_BARRIER();
                }
                hend = h1;
            }
            if constexpr (DEPTH2) {
                // tiles t of [h0, h2): t + 1 and t + 2 regular.  Pairs of tiles, so the register sets alternate statically.
                const int h2 = h1 - 1;
                if (h2 - h0 >= 2) {
                    Frag4<T> rax[PA];
                    Frag4<TB> rbx[PB];
                    const T* pa = la.cur + (int64_t)(h0 + 1) * BK * la.kstep;
                    const TB* pb = lb.cur + (int64_t)(h0 + 1) * BK * lb.kstep;
                    const int64_t da = (int64_t)BK * la.kstep, db = (int64_t)BK * lb.kstep;
#pragma unroll
                    for (int p = 0; p < PA; ++p) ra0[p] = ldg4(pa + p * la.pstep);                  // tile h0 + 1 -> set 0
#pragma unroll
                    for (int p = 0; p < PB; ++p) rb0[p] = ldg4(pb + p * lb.pstep);
                    pa += da; pb += db;
                    const int npairs = (h2 - h0) / 2;
                    int t = h0;
                    for (int q = 0; q < npairs; ++q, t += 2) {
#pragma unroll
                        for (int p = 0; p < PA; ++p) rax[p] = ldg4(pa + p * la.pstep);              // tile t + 2 -> set 1
#pragma unroll
                        for (int p = 0; p < PB; ++p) rbx[p] = ldg4(pb + p * lb.pstep);
                        pa += da; pb += db;
                        ktile(t & 1, true, std::integral_constant<int, FULLMASK>{}, std::false_type{}, ra0, rb0);   // stores t + 1
                        NSGP_LOOP_BARRIER();
#pragma unroll
                        for (int p = 0; p < PA; ++p) ra0[p] = ldg4(pa + p * la.pstep);              // tile t + 3 -> set 0
#pragma unroll
                        for (int p = 0; p < PB; ++p) rb0[p] = ldg4(pb + p * lb.pstep);
                        pa += da; pb += db;
                        ktile((t + 1) & 1, true, std::integral_constant<int, FULLMASK>{}, std::false_type{}, rax, rbx);   // stores t + 2
                        NSGP_LOOP_BARRIER();
                    }
                    hend = t;           // LDS holds tile `hend`; set 0's prefetch of tile hend + 1 is dropped (reloaded below)
                }
            }
            generic_tiles(hend, nt);
        }
    }

    NSGP_STAMP(2);
    // epilogue
    if constexpr (EPI == 1) {
        // plain store + per-column partial sums over this tile's rows (rows >= M carry acc == 0)
        const TC* rv = ep.rv ? reinterpret_cast<const TC*>(ep.rv) + bb * g.M : nullptr;
        T sdot[TN], ssq[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) { sdot[j] = T(0); ssq[j] = T(0); }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < MF::NREG; ++r) {
                const int64_t row = m0 + wm0 + i * MT + MF::crow(r, lane);
                const T rvv = (rv && row < g.M) ? (T)rv[row] : T(0);
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int64_t col = n0 + wn0 + j * MT + MF::ccol(lane);
                    const T v = alpha * acc[i][j][r];
                    if (row < g.M && col < g.N) Cb[row * g.ldc + col] = (TC)v;
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
            const int64_t o = (bb * (g.part_rows > 0 ? g.part_rows : g.tiles_m) + bm) * g.N + n0 + tid;
            if constexpr (MIX != 0) {
                if (ep.p64) {               // var = os + colsum(C^2) - colsum(A^2) cancels to << os once q(u) has trained
                    if (ep.p0) reinterpret_cast<double*>(ep.p0)[o] = red[tid] + red[BN + tid];
                    reinterpret_cast<double*>(ep.p1)[o] = red[2 * BN + tid] + red[3 * BN + tid];
                    NSGP_STAMP_FLUSH();
                    return;
                }
            }
            if (ep.p0) reinterpret_cast<TC*>(ep.p0)[o] = (TC)(red[tid] + red[BN + tid]);
            reinterpret_cast<TC*>(ep.p1)[o] = (TC)(red[2 * BN + tid] + red[3 * BN + tid]);
        }
        NSGP_STAMP_FLUSH();
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
        NSGP_STAMP_FLUSH();
        return;
    }
    if constexpr (MIX == 0) {
    const bool to_slab = g.ksplit > 1;
    T* out = to_slab ? slabs + slice * g.slab + bb * g.M * g.N : Cb;
    const int64_t ldo = to_slab ? g.N : g.ldc;
    // Interior tile written as it stands (a slab, or alpha * acc with beta = 0, no triangle, no halved diagonal): bare stores
    // from one base pointer.  The general loop below decides bounds / triangle / beta per element; on the n-wide outputs
    // (Kzxbar = W^T Abar: every workgroup stores a whole 128 x 128 tile) it took 14 us of a 98 us workgroup, twice the
    // column-statistics epilogue that stores the same bytes (tools/probes/gemm_stamps.py).
    const bool plain_tile = (m0 + BM <= g.M) && (n0 + BN <= g.N) &&
                            (to_slab || (!cL && beta == T(0) && !(g.flags & NSGP_GEMM_C_HALFDIAG)));
    if (plain_tile) {
        const T sc = to_slab ? T(1) : alpha;
        T* o0 = out + (m0 + wm0 + MF::crow_lane(lane)) * ldo + n0 + wn0 + MF::ccol(lane);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < MF::NREG; ++r)       // row = the lane's part (in o0) + a compile-time offset
                    o0[(int64_t)(i * MT + MF::crow_reg(r)) * ldo + j * MT] = sc * acc[i][j][r];
        NSGP_STAMP_FLUSH();
        return;
    }
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
                            if ((g.flags & NSGP_GEMM_C_HALFDIAG) && row == col) v *= T(0.5);
                            if (beta != T(0)) v += beta * out[row * ldo + col];
                            out[row * ldo + col] = v;
                        }
                    }
                }
            }
    }   // MIX == 0
    NSGP_STAMP_FLUSH();
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
    if ((g.flags & NSGP_GEMM_C_HALFDIAG) && row == col) s *= T(0.5);
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
        // K-tiles of MFMA work per round + per-block fixed cost (prologue/epilogue ~ 6 K-tiles) + the split's slab
        // write/read.  One unit is ~1.3 us (measured: 5 rounds x 70 units = 455 us); ks slabs of M x N x nb elements
        // are written and read back once, at ~4 TB/s: 2 ks M N nb sizeof(T) / 4e12 s.  (The old constant term priced a
        // 1024 x 1024 output; an n-wide output such as Kzxbar = W^T Abar at n = 5120 -- one rank of eight -- was split
        // three ways and paid 25 us of reduce for it: 138 us against 80 us un-split.)
        const double slab_units = ks > 1 ? (double)ks * (double)M * (double)N * (double)nb * sizeof(T) / 2.6e6 : 0.0;
        const double cost = (double)rounds * ((double)cdiv64(ktiles, ks) + 6.0) + slab_units;
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
    {   // experiment switch: 128 x 64 tiles (three workgroups per CU) for every single-pass triangular-A launch
        const char* fn = getenv("NSGP_GEMM_FORCE_NARROW");
        if (fn && fn[0] == '1' && p.big && p.ksplit == 1 && !(flags & NSGP_GEMM_C_LOWER) &&
            (flags & (NSGP_GEMM_A_LOWER | NSGP_GEMM_A_UPPER))) p.narrow = 1;
    }
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
    if ((flags & NSGP_GEMM_C_LOWER) && !(flags & NSGP_GEMM_C_NOFILL) && g.ksplit == 1 && beta == T(0)) {
        // the strictly-upper tiles are not launched: keep the "strict upper triangle is zero" contract
        // (the split-K reduce kernel writes those zeros itself)
        for (int64_t i1 = 0; i1 < nb1; ++i1)
            for (int64_t i2 = 0; i2 < nb2; ++i2) {
                hipError_t e = hipMemset2DAsync(C + i1 * sc1 + i2 * sc2, (size_t)ldc * sizeof(T), 0,
                                                (size_t)N * sizeof(T), (size_t)M, st);
                if (e != hipSuccess) return (int)e;
            }
    }
    int64_t ngrid_x = ngrid, ngrid_y = nb * g.ksplit;
    g.xcd_group = 0;
    g.xcd_chunk = 0;
    g.batch_perm = 0;
    g.part_rows = epi ? epi->part_rows : 0;
    g.nbk = (int)(nb * g.ksplit);
    const char* no_xcd = getenv("NSGP_GEMM_NO_XCD");              // A/B switches for tools/gemm_bench.py
    const bool xcd_ok = !(no_xcd && no_xcd[0] == '1');
    // The column-panel-grouped order is OFF by default: it cuts the fabric traffic of the projections ~3x, but the
    // hardware deals workgroups to the XCDs in strict rotation, so tiles of unequal length in flight stall the
    // dispatcher (measured: 611 us grouped vs 472 us in the row-major longest-first order, M=1024, n=40960).
    const char* grp = getenv("NSGP_GEMM_XCD_GROUP");
    const int grp_rows = grp ? atoi(grp) : 0;                       // 1: half the tile rows per group; r > 1: r rows per group
    const bool group_ok = grp_rows >= 1;
    if (xcd_ok && group_ok && !(flags & NSGP_GEMM_C_LOWER) && g.ksplit == 1 && g.tiles_n >= 16 && g.tiles_m <= 64 &&
        !(flags & (NSGP_GEMM_B_LOWER | NSGP_GEMM_B_UPPER))) {
        const bool triA = flags & (NSGP_GEMM_A_LOWER | NSGP_GEMM_A_UPPER);
        int R = (int)g.tiles_m;
        if (triA && g.tiles_m >= 4 && g.tiles_m % 2 == 0) R = (int)g.tiles_m / 2;     // long rows first, short rows last
        if (grp_rows > 1 && g.tiles_m % grp_rows == 0) R = grp_rows;
        g.xcd_group = R;
        ngrid_x = 8 * (g.tiles_m / R) * (cdiv64(g.tiles_n, 8) * R);
    } else if (xcd_ok && g.ksplit > 1 && ngrid * ngrid_y >= 64) {
        const int64_t total = ngrid * ngrid_y;
        g.xcd_chunk = (int)cdiv64(total, 8);
        // the launch keeps its (tiles, batch x slices) shape; workgroups past `total` after the remap exit at once
        ngrid_y = cdiv64(8 * (int64_t)g.xcd_chunk, ngrid);
    } else if (g.ksplit == 1 && nb > 1 && (flags & (NSGP_GEMM_A_LOWER | NSGP_GEMM_A_UPPER | NSGP_GEMM_B_LOWER |
                                                     NSGP_GEMM_B_UPPER | NSGP_GEMM_C_LOWER))) {
        // batched triangular launch that fits one round of resident workgroups: complementary tile orders per batch element
        const int64_t slots = 256 * (p.big ? (p.narrow ? 3 : 2) : (sizeof(T) == 4 ? 6 : 3));
        const char* bpe = getenv("NSGP_GEMM_BATCH_PERM");            // A/B switch (default on)
        // (more rounds: batch element fastest -- hidden layer, n = 4096 x 2 GPs: 234 -> 225 us for the forward pair)
        if (!(bpe && bpe[0] == '0')) g.batch_perm = (ngrid * ngrid_y <= slots) ? 1 : ((bpe && bpe[0] == '1') ? 0 : 2);
    } else if (g.ksplit == 1 && nb == 1 && (flags & (NSGP_GEMM_A_LOWER | NSGP_GEMM_A_UPPER | NSGP_GEMM_B_LOWER |
                                                      NSGP_GEMM_B_UPPER | NSGP_GEMM_C_LOWER))) {
        const int64_t slots = 256 * (p.big ? (p.narrow ? 3 : 2) : (sizeof(T) == 4 ? 6 : 3));
        const char* bpe = getenv("NSGP_GEMM_BATCH_PERM");
        if (ngrid > 256 && ngrid <= slots && !(bpe && bpe[0] == '0')) g.batch_perm = 3;       // snake (see the kernel)
    }
    dim3 grid((unsigned)ngrid_x, (unsigned)ngrid_y, 1);
    Epi ep{};
    if (epi) ep = *epi;
    if (tiles_m_out) *tiles_m_out = g.tiles_m;
    const int ekind = ep.kind, eks = ep.ks != nullptr;
    if (ekind != 0 && (g.ksplit != 1 || beta != T(0) || (flags & (NSGP_GEMM_C_LOWER | NSGP_GEMM_C_HALFDIAG)))) return -30;
    const int64_t bnn = p.narrow ? 64 : bmn, bkk = (bmn == 128 ? 32 : ((sizeof(T) == 8 && ekind == 0 && !eks) ? NSGP_F64_BK : 16));
    // whole, vector-loadable tiles everywhere -> the variant without bounds code (EDGE = 0)
    const bool whole = g.vecA && g.vecB && M % bmn == 0 && N % bnn == 0 && K % bkk == 0 && g.kper % bkk == 0;
    // launch<BM, BN, MA, MB, EP, KS>()  (the PF template slot, round 1's separate two-tiles-in-flight loop, is always 0 now)
    auto launch = [&](auto bm_c, auto bn_c, auto ma_c, auto mb_c, auto ep_c, auto ks_c) {
        constexpr int BM_ = decltype(bm_c)::value, BN_ = decltype(bn_c)::value, MA = decltype(ma_c)::value,
                      MB = decltype(mb_c)::value, EP = decltype(ep_c)::value, KSv = decltype(ks_c)::value;
        constexpr int BKc = (BM_ == 128 ? 32 : ((sizeof(T) == 8 && EP == 0 && KSv == 0) ? NSGP_F64_BK : 16));
        constexpr int pa = (MA == 0 && sizeof(T) == 4) ? 1 : Mfma<T>::PAD;
        constexpr int pb = (MB == 0 && sizeof(T) == 4) ? 1 : Mfma<T>::PAD;
        constexpr size_t lds = 2 * BKc * ((BM_ + pa) + (BN_ + pb)) * sizeof(T);
        auto go = [&](auto kern) {
            // NSGP_GEMM_LDS_EXTRA (bytes): occupancy experiment -- extra dynamic LDS, e.g. 40000 leaves one workgroup per CU
            static const char* xe = getenv("NSGP_GEMM_LDS_EXTRA");
            const size_t ldsx = lds + (xe ? (size_t)atoi(xe) : 0);
            nsgp_opt_in_lds((const void*)kern, ldsx);
            hipLaunchKernelGGL(kern, grid, dim3(256), ldsx, st, g, alpha, A, B, beta, C, slabs, ep);
        };
        if (whole) go(gemm_kernel<T, BM_, BN_, BKc, MA, MB, EP, KSv, 0, 0>);
        else go(gemm_kernel<T, BM_, BN_, BKc, MA, MB, EP, KSv, 0, 1>);
    };
#define IC(v) std::integral_constant<int, v>{}
    // fused variants exist for the operand layouts the SVGP layer uses
    auto launch_epi = [&](auto bm_c, auto bn_c) -> int {
        if (ekind == 1 && !eks && g.modeA == 0 && g.modeB == 1) launch(bm_c, bn_c, IC(0), IC(1), IC(1), IC(0));
        else if (ekind == 1 && !eks && g.modeA == 1 && g.modeB == 1) launch(bm_c, bn_c, IC(1), IC(1), IC(1), IC(0));
        else if (ekind == 2 && !eks && g.modeA == 0 && g.modeB == 1) launch(bm_c, bn_c, IC(0), IC(1), IC(2), IC(0));
        else if (ekind == 0 && eks && g.modeA == 0 && g.modeB == 0) launch(bm_c, bn_c, IC(0), IC(0), IC(0), IC(1));
        else return -31;
        return 0;
    };
    auto launch_plain = [&](auto bm_c, auto bn_c) {
        if (g.modeA == 0 && g.modeB == 0) launch(bm_c, bn_c, IC(0), IC(0), IC(0), IC(0));
        else if (g.modeA == 0) launch(bm_c, bn_c, IC(0), IC(1), IC(0), IC(0));
        else if (g.modeB == 0) launch(bm_c, bn_c, IC(1), IC(0), IC(0), IC(0));
        else launch(bm_c, bn_c, IC(1), IC(1), IC(0), IC(0));
    };
    int lrc = 0;
    if (p.narrow) {
        if constexpr (sizeof(T) == 4) {
            if (eks) return -31;
            if (ekind != 0) lrc = launch_epi(IC(128), IC(64));
            else launch_plain(IC(128), IC(64));
        }
    } else if (ekind != 0 || eks) {
        if (p.big) {
            if constexpr (sizeof(T) == 4) lrc = launch_epi(IC(128), IC(128));
        } else {
            lrc = launch_epi(IC(64), IC(64));
        }
    } else {
        if (p.big) {
            if constexpr (sizeof(T) == 4) launch_plain(IC(128), IC(128));
        } else {
            launch_plain(IC(64), IC(64));
        }
    }
#undef IC
    if (lrc != 0) return lrc;
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
// tile rows of the float64-accumulating projection for a given shape (see tri_gemm_colstats_f64acc_impl)
static inline int f64acc_tile_rows(int64_t M, int64_t n, int64_t batch) {
    return cdiv64(M, 128) * cdiv64(n, NSGP_F64ACC_BN) * batch < 512 ? 64 : 128;
}

template <typename T>
int tri_gemm_colstats_impl(const T* L, int trans, const T* X, const T* rowvec, int64_t batch, int64_t M, int64_t n,
                           T* Y, T* part_dot, T* part_sq, void* stream, int64_t part_rows = 0) {
    if (!L) return -1; if (trans != 0 && trans != 1) return -2; if (!X) return -3;
    if (batch < 0) return -5; if (M < 0) return -6; if (n < 0) return -7; if (!Y) return -8; if (!part_sq) return -10;
    if (batch == 0 || M == 0 || n == 0) return 0;
    Epi ep{};
    ep.kind = 1; ep.rv = rowvec; ep.p0 = part_dot; ep.p1 = part_sq; ep.modeA = trans ? 1 : 0; ep.modeB = 1;
    ep.part_rows = (int)part_rows;
    const int flags = (trans ? NSGP_GEMM_A_UPPER : NSGP_GEMM_A_LOWER) | NSGP_GEMM_NO_SPLITK;
    int64_t tiles_m = 0;
    if (part_rows > 0) {                                  // the caller's buffers must hold this launch's tile rows
        Plan p = make_plan<T>(M, n, M, batch, flags, !(sizeof(T) == 8));
        if (part_rows < cdiv64(M, p.big ? 128 : 64)) return -12;
    }
    (void)tiles_m;
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
               size_t wsb, void* stream, T beta = T(0)) {
    if (!A) return -1; if (!C) return -2; if (!gvar) return -3;
    if (batch < 0) return -4; if (M < 0) return -5; if (n < 0) return -6; if (!Lqbar) return -7;
    if (batch == 0 || M == 0) return 0;
    Epi ep{};
    ep.kind = 0; ep.ks = gvar; ep.modeA = 0; ep.modeB = 0;
    // Lqbar = tril(A diag(2 gvar) C^T):  B(k, j) = C[j][k], scaled along k by gvar, alpha = 2
    return gemm_impl<T>(M, M, n, T(2), A, n, 1, M * n, 0, C, 1, n, M * n, 0, beta, Lqbar, M, M * M, 0, batch, 1,
                        NSGP_GEMM_C_LOWER, ws, wsb, stream, &ep);
}
}  // namespace

namespace {
// A = L^-1-style projection with float64 arithmetic on float32 data (gemm_kernel<double, 128, 64, 16, ..., MIX = 1>):
// Y[b] = W[b] X[b] (W: float64 lower triangular (M x M); X, Y: float32 (M x n)), plus the column-statistic partials of
// nsgp_svgp_tri_gemm_colstats (float32; ceil(M / 128) tile rows).
struct KzxGen { const float *z, *x, *ls, *os; int D; int64_t sx; };   // generated B operand (Epi::kz ...), or null

static inline bool f64acc_whole(const double* W, int64_t M, int64_t n, int bm) {
    return M % 4 == 0 && (uintptr_t)W % 32 == 0 && M % bm == 0 && n % NSGP_F64ACC_BN == 0 && M % NSGP_F64ACC_BK == 0;
}

int tri_gemm_colstats_f64acc_impl(const double* W, const float* X, const float* rowvec, int64_t batch, int64_t M, int64_t n,
                                  float* Y, float* part_dot, float* part_sq, int64_t part_rows, void* stream,
                                  const KzxGen* kg = nullptr, const double* X64 = nullptr, int trans = 0, int p64 = 0) {
    // X64 != null: the B operand is float64 in memory (MIX = 3); X is then ignored
    // trans = 1: Y = W^T X with the stored lower-triangular W (the second projection C = Lq^T A, MIX = 1 only)
    if (trans && (kg || X64)) return -3;
    if (X64) X = reinterpret_cast<const float*>(X64);
    if (!W) return -1; if (!X && !kg) return -2; if (batch < 0) return -4; if (M < 0) return -5; if (n < 0) return -6;
    if (!Y) return -7; if (!part_sq) return -9;
    if (batch == 0 || M == 0 || n == 0) return 0;
    constexpr int BN_ = NSGP_F64ACC_BN, BK_ = NSGP_F64ACC_BK;
    // 128-row tiles, or 64-row tiles when the 128-row grid would not fill one round of the chip (a rank's share of the
    // hidden layer at 4-8 GPUs, n = 512..1024: 128 workgroups of up to 64 K-tiles each were latency-bound, 96 us)
    const int BM_ = f64acc_tile_rows(M, n, batch);
    if (part_rows < cdiv64(M, BM_)) return -10;
    GemmArgs g{};
    g.M = M; g.N = n; g.K = M;
    g.sam = trans ? 1 : M; g.sak = trans ? M : 1; g.sa1 = M * M; g.sa2 = 0;
    g.sbk = n; g.sbn = 1; g.sb1 = M * n; g.sb2 = 0;
    g.ldc = n; g.sc1 = M * n; g.sc2 = 0; g.nb2 = 1;
    g.tiles_m = cdiv64(M, BM_); g.tiles_n = cdiv64(n, BN_);
    g.ksplit = 1; g.kper = cdiv64(M, 32) * 32; g.slab = 0;
    g.flags = (trans ? NSGP_GEMM_A_UPPER : NSGP_GEMM_A_LOWER) | NSGP_GEMM_NO_SPLITK;
    g.nbk = (int)batch; g.xcd_chunk = 0; g.xcd_group = 0; g.part_rows = (int)part_rows;
    g.modeA = trans ? 1 : 0; g.modeB = 1;
    g.vecA = (M % 4 == 0) && ((uintptr_t)W % 32 == 0);
    g.vecB = kg ? 1 : ((n % 4 == 0) && ((uintptr_t)X % (X64 ? 32 : 16) == 0));
    if (g.tiles_m * g.tiles_n > 2147483647LL || batch > 65535) return -24;
    if (batch > 1) {                                     // tile order across the batch, as in gemm_impl
        const char* bpe = getenv("NSGP_GEMM_BATCH_PERM");
        if (!(bpe && bpe[0] == '0'))
            g.batch_perm = (g.tiles_m * g.tiles_n * batch <= 512) ? 1 : 0;   // (batch-fastest measured slower here: 314 -> 329 us)
    }
    Epi ep{};
    ep.kind = 1; ep.rv = rowvec; ep.p0 = part_dot; ep.p1 = part_sq; ep.modeA = trans ? 1 : 0; ep.modeB = 1; ep.p64 = p64;
    if (kg) {
        if (!kg->z || !kg->x || !kg->ls || !kg->os) return -2;
        if (kg->D < 1 || kg->D > KGEN_DMAX) return -3;
        if (!f64acc_whole(W, M, n, BM_)) return -11;          // the generated operand exists for whole tiles only
        ep.kz = kg->z; ep.kx = kg->x; ep.kls = kg->ls; ep.kos = kg->os; ep.kD = kg->D; ep.kx_sx = kg->sx;
    }
    const bool whole = g.vecA && g.vecB && M % BM_ == 0 && n % BN_ == 0 && M % BK_ == 0;
    dim3 grid((unsigned)(g.tiles_m * g.tiles_n), (unsigned)batch, 1);
    hipStream_t st = (hipStream_t)stream;
    // the kernel takes its B / C pointers as double* and reinterprets them (MIX = 1)
    const double* Bp = reinterpret_cast<const double*>(X);
    double* Cp = reinterpret_cast<double*>(Y);
    auto go = [&](auto bm_c, auto edge_c, auto mix_c) {
        constexpr int BMc = decltype(bm_c)::value, EDGEc = decltype(edge_c)::value, MIXc = decltype(mix_c)::value;
        constexpr size_t lds = 2 * BK_ * ((BMc + Mfma<double>::PAD) + (BN_ + Mfma<double>::PAD)) * sizeof(double);
        nsgp_opt_in_lds((const void*)gemm_kernel<double, BMc, BN_, BK_, 0, 1, 1, 0, 0, EDGEc, MIXc>, lds);
        hipLaunchKernelGGL((gemm_kernel<double, BMc, BN_, BK_, 0, 1, 1, 0, 0, EDGEc, MIXc>), grid, dim3(256), lds, st, g, 1.0,
                           W, Bp, 0.0, Cp, (double*)nullptr, ep);
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
    using B64 = std::integral_constant<int, 64>; using B128 = std::integral_constant<int, 128>;
    using I3 = std::integral_constant<int, 3>;
    auto go_t = [&](auto bm_c, auto edge_c) {                 // transposed-A variant (operand contiguous along m)
        constexpr int BMc = decltype(bm_c)::value, EDGEc = decltype(edge_c)::value;
        constexpr size_t lds = 2 * BK_ * ((BMc + Mfma<double>::PAD) + (BN_ + Mfma<double>::PAD)) * sizeof(double);
        nsgp_opt_in_lds((const void*)gemm_kernel<double, BMc, BN_, BK_, 1, 1, 1, 0, 0, EDGEc, 1>, lds);
        hipLaunchKernelGGL((gemm_kernel<double, BMc, BN_, BK_, 1, 1, 1, 0, 0, EDGEc, 1>), grid, dim3(256), lds, st, g, 1.0,
                           W, Bp, 0.0, Cp, (double*)nullptr, ep);
    };
    if (trans) {
        if (BM_ == 128) { if (whole) go_t(B128{}, I0{}); else go_t(B128{}, I1{}); }
        else { if (whole) go_t(B64{}, I0{}); else go_t(B64{}, I1{}); }
    }
    else if (X64) {
        if (BM_ == 128) { if (whole) go(B128{}, I0{}, I3{}); else go(B128{}, I1{}, I3{}); }
        else { if (whole) go(B64{}, I0{}, I3{}); else go(B64{}, I1{}, I3{}); }
    }
    else if (kg) { if (BM_ == 128) go(B128{}, I0{}, I2{}); else go(B64{}, I0{}, I2{}); }
    else if (BM_ == 128) { if (whole) go(B128{}, I0{}, I1{}); else go(B128{}, I1{}, I1{}); }
    else { if (whole) go(B64{}, I0{}, I1{}); else go(B64{}, I1{}, I1{}); }
    return nsgp_launch_status();
}
}  // namespace

extern "C" {

#ifdef NSGP_GEMM_STAMPS
// diagnostic build: where the workgroups of the following GEMM launches write their stamp records (8 x uint64 each)
int nsgp_debug_gemm_stamps(void* buf, uint64_t cap_records) {
    unsigned long long* b = (unsigned long long*)buf;
    unsigned long long c = cap_records;
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(nsgp_stamp_buf), &b, sizeof(b));
    if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(nsgp_stamp_cap), &c, sizeof(c));
    return (int)e;
}
#endif

size_t nsgp_svgp_f64acc_tiles(int64_t M) { return M > 0 ? (size_t)cdiv64(M, 128) : 0; }
int nsgp_svgp_kzx_gemm_colstats_f64acc(const double* W, const float* z, const float* x, int64_t sx, const float* ls,
                                       const float* os, int D, const float* rowvec, int64_t batch, int64_t M, int64_t n,
                                       float* Y, float* part_dot, float* part_sq, int64_t part_rows, void* stream) {
    KzxGen kg{z, x, ls, os, D, sx};
    return tri_gemm_colstats_f64acc_impl(W, nullptr, rowvec, batch, M, n, Y, part_dot, part_sq, part_rows, stream, &kg);
}
int nsgp_svgp_kzx_gemm_supported(const double* W, int64_t M, int64_t n, int64_t batch, int D) {
    return (D >= 1 && D <= KGEN_DMAX && M > 0 && n > 0 && batch > 0 &&
            f64acc_whole(W, M, n, f64acc_tile_rows(M, n, batch))) ? 1 : 0;
}
size_t nsgp_svgp_f64acc_tiles_for(int64_t M, int64_t n, int64_t batch) {
    return (M > 0 && n > 0 && batch > 0) ? (size_t)cdiv64(M, f64acc_tile_rows(M, n, batch)) : 0;
}
int nsgp_svgp_tri_gemm_colstats_f64acc(const double* W, const float* X, const float* rowvec, int64_t batch, int64_t M,
                                       int64_t n, float* Y, float* part_dot, float* part_sq, int64_t part_rows,
                                       void* stream) {
    return tri_gemm_colstats_f64acc_impl(W, X, rowvec, batch, M, n, Y, part_dot, part_sq, part_rows, stream);
}

int nsgp_svgp_tri_gemm_colstats_f64acc_b64(const double* W, const double* X64, const float* rowvec, int64_t batch, int64_t M,
                                           int64_t n, float* Y, double* part_dot, double* part_sq, int64_t part_rows,
                                           void* stream) {
    if (!X64) return -2;
    return tri_gemm_colstats_f64acc_impl(W, nullptr, rowvec, batch, M, n, Y, reinterpret_cast<float*>(part_dot),
                                         reinterpret_cast<float*>(part_sq), part_rows, stream, nullptr, X64, 0, 1);
}

int nsgp_svgp_tri_gemm_colstats_f64acc_b64p32(const double* W, const double* X64, const float* rowvec, int64_t batch, int64_t M,
                                              int64_t n, float* Y, float* part_dot, float* part_sq, int64_t part_rows,
                                              void* stream) {
    if (!X64) return -2;
    return tri_gemm_colstats_f64acc_impl(W, nullptr, rowvec, batch, M, n, Y, part_dot, part_sq, part_rows, stream, nullptr, X64, 0, 0);
}
int nsgp_svgp_tri_gemm_colstats_f64acc_t(const double* L, const float* X, int64_t batch, int64_t M, int64_t n, float* Y,
                                         double* part_sq, int64_t part_rows, void* stream) {
    return tri_gemm_colstats_f64acc_impl(L, X, nullptr, batch, M, n, Y, nullptr, reinterpret_cast<float*>(part_sq), part_rows,
                                         stream, nullptr, nullptr, 1, 1);
}

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
int nsgp_svgp_tri_gemm_colstats_rows_f32(const float* L, int trans, const float* X, const float* rowvec, int64_t batch,
                                         int64_t M, int64_t n, float* Y, float* part_dot, float* part_sq,
                                         int64_t part_rows, void* stream) {
    return tri_gemm_colstats_impl<float>(L, trans, X, rowvec, batch, M, n, Y, part_dot, part_sq, stream, part_rows);
}
int nsgp_svgp_tri_gemm_colstats_rows_f64(const double* L, int trans, const double* X, const double* rowvec, int64_t batch,
                                         int64_t M, int64_t n, double* Y, double* part_dot, double* part_sq,
                                         int64_t part_rows, void* stream) {
    return tri_gemm_colstats_impl<double>(L, trans, X, rowvec, batch, M, n, Y, part_dot, part_sq, stream, part_rows);
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
int nsgp_svgp_lqbar_acc_f32(const float* A, const float* C, const float* gvar, int64_t batch, int64_t M, int64_t n,
                            float beta, float* Lqbar, void* ws, size_t ws_bytes, void* stream) {
    return lqbar_impl<float>(A, C, gvar, batch, M, n, Lqbar, ws, ws_bytes, stream, beta);
}
int nsgp_svgp_lqbar_acc_f64(const double* A, const double* C, const double* gvar, int64_t batch, int64_t M, int64_t n,
                            double beta, double* Lqbar, void* ws, size_t ws_bytes, void* stream) {
    return lqbar_impl<double>(A, C, gvar, batch, M, n, Lqbar, ws, ws_bytes, stream, beta);
}

}  // extern "C"
