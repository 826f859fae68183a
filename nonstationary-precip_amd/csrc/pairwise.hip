// pairwise.hip -- kernel-matrix builds K[i,j] = k(row_i, col_j) and their backward reductions.
//
//   K1 Gibbs (models/gibbs_kernels.py:154-162), K2 batched RBF-ARD (gpytorch ScaleKernel(RBF),
//   models/dgps.py:44-46), K3 Paciorek-Schervish D=2 (models/multivariate_gibbs_kernel.py:98-150).
//
// Forward: HBM-write bound.  One 256-thread workgroup owns a 64 x (64*CPT) tile; a lane owns CPT
// consecutive columns (16 B) so every wave store is one contiguous 1 KiB line group; the column
// operands live in registers for the whole tile and the row operands are computed once per tile and
// broadcast-read from LDS.
// Algorithmic bytes per build: s*(n1*n2 + 2*D*(n1+n2)).
// Measured at N = 16384 (tools/probes/): a torch fill_ writes 6.85 TB/s; this kernel with the arithmetic stubbed out
// reaches 5.8 TB/s (float32) / 6.2 TB/s (float64), and the real Gibbs build 5.7 / 5.0 TB/s -- float32 sits on the
// skeleton's ceiling, float64 still pays ~43 VALU instructions per entry (custom exp and rsqrt in common.h).
// Tried without gain: v_pk_* arithmetic on column pairs, non-temporal stores, 16 x (256*CPT) tiles with the four waves
// side by side (-12 %), an XCD-contiguous tile order.
//
// Backward: one pass over G (dLoss/dK).  A workgroup owns a 64 x 256 tile: wave w walks rows
// w, w+4, ..; a lane covers 4 columns 64 apart (coalesced G reads).  Row-side gradients are
// wave-reduced once per row (amortised over 256 columns), column-side gradients stay in
// registers and are combined across the 4 waves through LDS.  Tile partials go to a workspace and
// a second tiny kernel sums them: deterministic, no atomics.
#include "common.h"

namespace {

template <typename T> struct Cpt;
template <> struct Cpt<float> { static constexpr int v = 4; };
template <> struct Cpt<double> { static constexpr int v = 2; };

template <int D> struct DimMax { static constexpr int v = D ? D : NSGP_MAX_DIM; };

// ------------------------------------------------------------------------------------------
// functors
// ------------------------------------------------------------------------------------------
template <typename T, int D> struct GibbsOp {
    static constexpr int DM = DimMax<D>::v;
    static constexpr int NR = 2 * DM, NC = 2 * DM, NG = 1;
    const T *x1, *x2, *l1, *l2;
    int64_t n1, n2;
    int Drt;
    const T* osp;                          // device scalar or nullptr (= 1)
    __device__ __forceinline__ T os() const { return osp ? osp[0] : T(1); }
    // per-point operands: x, l, q = l^2 and h = sqrt(prod_d sqrt(2) l_d), so that
    //   k = prod_d sqrt(2 l1 l2 / s_d) * exp(-sum_d delta_d^2 / s_d),   s_d = q1_d + q2_d
    //     = h1 h2 rsqrt(S) * exp(-(sum_d delta_d^2 prod_{e != d} s_e) / S),   S = prod_d s_d
    // costs ONE rsqrt and ONE exp per matrix entry (the per-point sqrt is amortised over a tile).
    struct P { T x[DM]; T l[DM]; T q[DM]; T h; };
    __device__ __forceinline__ P point(const T* x, const T* l, int64_t n, int64_t i) const {
        P p;
        T hp = T(1);
#pragma unroll
        for (int d = 0; d < DM; ++d) {
            const bool on = D || d < Drt;
            p.x[d] = on ? x[i * Drt + d] : T(0);
            p.l[d] = on ? l[(int64_t)d * n + i] : T(1);
            p.q[d] = p.l[d] * p.l[d];
            if (on) hp *= T(1.4142135623730951) * p.l[d];
        }
        p.h = t_fsqrt(hp);
        return p;
    }
    __device__ __forceinline__ P row(int64_t, int64_t i) const { return point(x1, l1, n1, i); }
    __device__ __forceinline__ P col(int64_t, int64_t j) const { return point(x2, l2, n2, j); }
    // unscaled kernel value
    __device__ __forceinline__ T base(const P& r, const P& c) const {
        T S = r.q[0] + c.q[0];
        T num = (r.x[0] - c.x[0]) * (r.x[0] - c.x[0]);
#pragma unroll
        for (int d = 1; d < DM; ++d) {
            if (D || d < Drt) {
                const T s = r.q[d] + c.q[d];
                const T df = r.x[d] - c.x[d];
                num = num * s + df * df * S;               // sum_d delta_d^2 prod_{e != d} s_e, built incrementally
                S *= s;
            }
        }
        const T rs = t_rsqrt(S);
        return r.h * c.h * rs * t_fexp(-num * rs * rs);
    }
    __device__ __forceinline__ T eval(int64_t, const P& r, const P& c) const { return os() * base(r, c); }
    // forward-build variants: the output scale is folded into the column operand once per thread
    __device__ __forceinline__ P fcol(int64_t b, int64_t j) const { P p = col(b, j); p.h *= os(); return p; }
    __device__ __forceinline__ T feval(int64_t, const P& r, const P& c) const { return base(r, c); }
    // accumulate g * d k / d(param) into row / col / global accumulators
    __device__ __forceinline__ void grad(int64_t, const P& r, const P& c, T g, T* ra, T* ca, T* ga) const {
        const T kb = base(r, c);
        ga[0] += g * kb;
        const T w = g * kb * os();
#pragma unroll
        for (int d = 0; d < DM; ++d) {
            if (D || d < Drt) {
                const T a = r.l[d], b = c.l[d];
                const T s = a * a + b * b;
                const T inv = T(1) / s;
                const T df = r.x[d] - c.x[d];
                const T q = df * df * inv * inv;          // delta^2 / s^2
                ra[d] += w * (T(0.5) / a - a * inv + T(2) * a * q);
                ca[d] += w * (T(0.5) / b - b * inv + T(2) * b * q);
                const T gx = T(2) * w * df * inv;
                ra[DM + d] -= gx;
                ca[DM + d] += gx;
            }
        }
    }
};

template <typename T, int D> struct RbfOp {
    static constexpr int DM = DimMax<D>::v;
    static constexpr int NR = DM, NC = DM, NG = DM + 1;
    const T *x1, *x2, *ls, *os;           // ls:(batch,D) os:(batch)
    int64_t n1, n2, sx1, sx2;
    int Drt;
    struct P { T x[DM]; };                // pre-divided by the lengthscale
    __device__ __forceinline__ P row(int64_t b, int64_t i) const {
        P p;
#pragma unroll
        for (int d = 0; d < DM; ++d)
            p.x[d] = (D || d < Drt) ? x1[b * sx1 + i * Drt + d] / ls[b * Drt + d] : T(0);
        return p;
    }
    __device__ __forceinline__ P col(int64_t b, int64_t j) const {
        P p;
#pragma unroll
        for (int d = 0; d < DM; ++d)
            p.x[d] = (D || d < Drt) ? x2[b * sx2 + j * Drt + d] / ls[b * Drt + d] : T(0);
        return p;
    }
    __device__ __forceinline__ T base(const P& r, const P& c) const {
        T ex = T(0);
#pragma unroll
        for (int d = 0; d < DM; ++d) {
            const T df = r.x[d] - c.x[d];
            ex = t_fma(df, df, ex);                  // explicit: gemm.hip's generated-Kzx loader repeats this sum bit for bit
        }
        return t_fexp(T(-0.5) * ex);
    }
    __device__ __forceinline__ T eval(int64_t b, const P& r, const P& c) const { return os[b] * base(r, c); }
    __device__ __forceinline__ P fcol(int64_t b, int64_t j) const { return col(b, j); }
    __device__ __forceinline__ T feval(int64_t b, const P& r, const P& c) const { return eval(b, r, c); }
    // row/col accumulators are in units of d/d(x/ls) ("scaled x"); pass 2 divides by ls.
    __device__ __forceinline__ void grad(int64_t b, const P& r, const P& c, T g, T* ra, T* ca, T* ga) const {
        const T kb = base(r, c);
        ga[DM] += g * kb;
        const T w = g * kb * os[b];
#pragma unroll
        for (int d = 0; d < DM; ++d) {
            const T df = r.x[d] - c.x[d];
            ra[d] -= w * df;
            ca[d] += w * df;
            ga[d] += w * df * df;             // * 1/ls applied in pass 2
        }
    }
};

// K2' batched  os * RBF-ARD(x; ls_r) * Periodic(x; ls_p, period):
//   k = os * exp(-1/2 sum_d ((x_d - x'_d)/ls_r,d)^2) * exp(-2 sin^2(pi |x - x'| / period) / ls_p)
// (gpytorch PeriodicKernel < 1.9 as recalled in SURVEY A.2/A.7: Euclidean distance of x / period, division by
// the lengthscale, not its square).  One launch builds ScaleKernel(RBFKernel * PeriodicKernel) of
// models/spatio_temporal_models.py:22,42 and experiments/temporal_exp.py:39; ls_r == nullptr drops the RBF
// factor (plain PeriodicKernel), os == nullptr means 1.
template <typename T, int D> struct RbfPeriodicOp {
    static constexpr int DM = DimMax<D>::v;
    static constexpr int NR = DM, NC = DM, NG = DM + 3;        // globals: ls_r[D], ls_p, period, os
    const T *x1, *x2, *lsr, *lsp, *per, *os;
    int64_t n1, n2, sx1, sx2;
    int Drt;
    struct P { T x[DM]; };
    __device__ __forceinline__ P row(int64_t b, int64_t i) const {
        P p;
#pragma unroll
        for (int d = 0; d < DM; ++d) p.x[d] = (D || d < Drt) ? x1[b * sx1 + i * Drt + d] : T(0);
        return p;
    }
    __device__ __forceinline__ P col(int64_t b, int64_t j) const {
        P p;
#pragma unroll
        for (int d = 0; d < DM; ++d) p.x[d] = (D || d < Drt) ? x2[b * sx2 + j * Drt + d] : T(0);
        return p;
    }
    // exponent pieces: q = sum_d (delta_d / ls_r,d)^2, r = |delta|, (s, c) = sincos(pi r / period)
    __device__ __forceinline__ T base(int64_t b, const P& r, const P& c, T& rr, T& sn, T& cs) const {
        T q = T(0), r2 = T(0);
#pragma unroll
        for (int d = 0; d < DM; ++d) {
            if (D || d < Drt) {
                const T df = r.x[d] - c.x[d];
                r2 += df * df;
                if (lsr) { const T il = T(1) / lsr[b * Drt + d]; q += df * df * il * il; }
            }
        }
        rr = t_sqrt(r2);
        const T u = T(3.14159265358979323846) * rr / per[b];
        sn = sin(u); cs = cos(u);
        return t_exp(T(-0.5) * q - T(2) * sn * sn / lsp[b]);
    }
    __device__ __forceinline__ T eval(int64_t b, const P& r, const P& c) const {
        T rr, sn, cs;
        const T k = base(b, r, c, rr, sn, cs);
        return os ? os[b] * k : k;
    }
    __device__ __forceinline__ P fcol(int64_t b, int64_t j) const { return col(b, j); }
    __device__ __forceinline__ T feval(int64_t b, const P& r, const P& c) const { return eval(b, r, c); }
    __device__ __forceinline__ void grad(int64_t b, const P& r, const P& c, T g, T* ra, T* ca, T* ga) const {
        T rr, sn, cs;
        const T kb = base(b, r, c, rr, sn, cs);
        ga[DM + 2] += g * kb;                                              // d/d os
        const T w = g * kb * (os ? os[b] : T(1));
        const T ilp = T(1) / lsp[b], ip = T(1) / per[b];
        ga[DM] += w * T(2) * sn * sn * ilp * ilp;                          // d/d ls_p
        const T dsdu = T(4) * ilp * sn * cs * T(3.14159265358979323846);   // -(d exponent / d u), u = pi r / period
        ga[DM + 1] += w * dsdu * rr * ip * ip;                             // d/d period  (du/dp = -pi r / p^2)
        const T dr = rr > T(0) ? dsdu * ip / rr : T(0);                    // -(d exponent / d r) / r
#pragma unroll
        for (int d = 0; d < DM; ++d) {
            if (D || d < Drt) {
                const T df = r.x[d] - c.x[d];
                T gx = -dr * df;                                           // d exponent / d x1_d (periodic part)
                if (lsr) {
                    const T il = T(1) / lsr[b * Drt + d];
                    gx -= df * il * il;
                    ga[d] += w * df * df * il * il * il;                   // d/d ls_r,d
                }
                ra[d] += w * gx;
                ca[d] -= w * gx;
            }
        }
    }
};

template <typename T> struct PsOp {
    static constexpr int NR = 4, NC = 4, NG = 1;      // NG unused (kept 1 for array sizing)
    const T *x1, *x2, *s1, *s2;
    T jit;
    struct P { T x[2]; T s[4]; T q;  /* det^(1/4) */ };
    __device__ __forceinline__ P load(const T* x, const T* s, int64_t i) const {
        P p;
        p.x[0] = x[2 * i]; p.x[1] = x[2 * i + 1];
#pragma unroll
        for (int k = 0; k < 4; ++k) p.s[k] = s[4 * i + k];
        const T det = p.s[0] * p.s[3] - p.s[1] * p.s[2];
        p.q = t_sqrt(t_sqrt(det));
        return p;
    }
    __device__ __forceinline__ P row(int64_t, int64_t i) const { return load(x1, s1, i); }
    __device__ __forceinline__ P col(int64_t, int64_t j) const { return load(x2, s2, j); }
    __device__ __forceinline__ P fcol(int64_t b, int64_t j) const { return col(b, j); }
    __device__ __forceinline__ T feval(int64_t b, const P& r, const P& c) const { return eval(b, r, c); }
    __device__ __forceinline__ T eval(int64_t, const P& r, const P& c) const {
        const T a0 = T(0.5) * (r.s[0] + c.s[0]), a1 = T(0.5) * (r.s[1] + c.s[1]);
        const T a2 = T(0.5) * (r.s[2] + c.s[2]), a3 = T(0.5) * (r.s[3] + c.s[3]);
        const T detA = a0 * a3 - a1 * a2;
        const T b0 = a0 + jit, b3 = a3 + jit;
        const T detB = b0 * b3 - a1 * a2;
        const T d0 = r.x[0] - c.x[0], d1 = r.x[1] - c.x[1];
        const T quad = (b3 * d0 * d0 - (a1 + a2) * d0 * d1 + b0 * d1 * d1) / detB;
        return r.q * c.q / t_sqrt(detA) * t_exp(-quad);
    }
    __device__ __forceinline__ void grad(int64_t, const P& r, const P& c, T g, T* ra, T* ca, T*) const {
        const T a0 = T(0.5) * (r.s[0] + c.s[0]), a1 = T(0.5) * (r.s[1] + c.s[1]);
        const T a2 = T(0.5) * (r.s[2] + c.s[2]), a3 = T(0.5) * (r.s[3] + c.s[3]);
        const T detA = a0 * a3 - a1 * a2;
        const T b0 = a0 + jit, b3 = a3 + jit;
        const T detB = b0 * b3 - a1 * a2;
        const T d0 = r.x[0] - c.x[0], d1 = r.x[1] - c.x[1];
        const T quad = (b3 * d0 * d0 - (a1 + a2) * d0 * d1 + b0 * d1 * d1) / detB;
        const T k = r.q * c.q / t_sqrt(detA) * t_exp(-quad);
        const T w = g * k;
        // B^-1 d  and  B^-T d
        const T iB = T(1) / detB;
        const T v0 = (b3 * d0 - a1 * d1) * iB, v1 = (-a2 * d0 + b0 * d1) * iB;     // B^-1 d
        const T u0 = (b3 * d0 - a2 * d1) * iB, u1 = (-a1 * d0 + b0 * d1) * iB;     // B^-T d
        // shared term: -1/4 A^-T + 1/2 u v^T
        const T iA = T(0.25) / detA;
        const T sh0 = -iA * a3 + T(0.5) * u0 * v0;
        const T sh1 = iA * a2 + T(0.5) * u0 * v1;
        const T sh2 = iA * a1 + T(0.5) * u1 * v0;
        const T sh3 = -iA * a0 + T(0.5) * u1 * v1;
        const T dr = T(0.25) / (r.s[0] * r.s[3] - r.s[1] * r.s[2]);
        const T dc = T(0.25) / (c.s[0] * c.s[3] - c.s[1] * c.s[2]);
        ra[0] += w * (sh0 + dr * r.s[3]); ra[1] += w * (sh1 - dr * r.s[2]);
        ra[2] += w * (sh2 - dr * r.s[1]); ra[3] += w * (sh3 + dr * r.s[0]);
        ca[0] += w * (sh0 + dc * c.s[3]); ca[1] += w * (sh1 - dc * c.s[2]);
        ca[2] += w * (sh2 - dc * c.s[1]); ca[3] += w * (sh3 + dc * c.s[0]);
    }
};

// ------------------------------------------------------------------------------------------
// forward tile kernel
// ------------------------------------------------------------------------------------------
constexpr int FWD_TI = 64;

template <typename T, typename Op>
__global__ __launch_bounds__(256) void pairwise_fwd_kernel(Op op, int64_t n1, int64_t n2, T diag_add_v,
                                                           const T* __restrict__ diag_add_p,
                                                           T* __restrict__ K, int64_t ldk, int64_t sK, int vec_ok) {
    constexpr int CPT = Cpt<T>::v;
    __shared__ typename Op::P rows_s[FWD_TI];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t b = blockIdx.z;
    const int64_t jb = (int64_t)blockIdx.x * 64 * CPT;
    const int64_t j0 = jb + (int64_t)lane * CPT;
    const int64_t i0 = (int64_t)blockIdx.y * FWD_TI;
    // column operands first (clamped indices, so every thread may load): their global-load latency overlaps the row
    // operands' -- a tile is short (N = 4096: 1024 workgroups, all resident at once, ~20 us), two dependent
    // latencies in front of the first store were a tenth of the launch
    typename Op::P cols[CPT];
#pragma unroll
    for (int c = 0; c < CPT; ++c) cols[c] = op.fcol(b, j0 + c < n2 ? j0 + c : n2 - 1);
    const T diag_add = diag_add_p ? diag_add_p[0] : diag_add_v;
    // row operands of the tile: computed once (one thread per row), then broadcast-read from LDS
    if (threadIdx.x < FWD_TI && i0 + threadIdx.x < n1) rows_s[threadIdx.x] = op.row(b, i0 + threadIdx.x);
    __syncthreads();
    if (j0 >= n2) return;
    T* Kb = K + b * sK;
    const bool full = vec_ok && (j0 + CPT <= n2);
    const int rmax = (int)(n1 - i0 < FWD_TI ? n1 - i0 : FWD_TI);
    // only tiles that cross the diagonal test for it (block-uniform)
    const bool on_diag = i0 < jb + 64 * CPT && jb < i0 + FWD_TI;
    auto entry = [&](int r, T (&v)[CPT]) {
        const typename Op::P rp = rows_s[r];
#pragma unroll
        for (int c = 0; c < CPT; ++c) v[c] = op.feval(b, rp, cols[c]);
        if (on_diag) {
#pragma unroll
            for (int c = 0; c < CPT; ++c) if (i0 + r == j0 + c) v[c] += diag_add;
        }
    };
    if (full) {
#pragma unroll 2
        for (int r = w; r < rmax; r += 4) {
            T v[CPT];
            entry(r, v);
            T* dst = Kb + (i0 + r) * ldk + j0;
            if constexpr (CPT == 4) *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
            else *reinterpret_cast<double2*>(dst) = make_double2(v[0], v[1]);
        }
    } else {
        for (int r = w; r < rmax; r += 4) {
            T v[CPT];
            entry(r, v);
            T* dst = Kb + (i0 + r) * ldk + j0;
#pragma unroll
            for (int c = 0; c < CPT; ++c) if (j0 + c < n2) dst[c] = v[c];
        }
    }
}

template <typename T, typename Op>
int launch_fwd(const Op& op, int64_t batch, int64_t n1, int64_t n2, T diag_add, const T* diag_add_p, T* K,
               int64_t ldk, int64_t sK, void* stream) {
    if (n1 == 0 || n2 == 0 || batch == 0) return 0;
    constexpr int CPT = Cpt<T>::v;
    const int vec_ok = (ldk % CPT == 0) && (sK % CPT == 0) && ((uintptr_t)K % (CPT * sizeof(T)) == 0);
    dim3 grid((unsigned)cdiv64(n2, 64 * CPT), (unsigned)cdiv64(n1, FWD_TI), (unsigned)batch);
    hipLaunchKernelGGL((pairwise_fwd_kernel<T, Op>), grid, dim3(256), 0, (hipStream_t)stream, op, n1, n2,
                       diag_add, diag_add_p, K, ldk, sK, vec_ok);
    return nsgp_launch_status();
}

// ------------------------------------------------------------------------------------------
// backward tile kernel (pass 1) and partial reduction (pass 2)
// ------------------------------------------------------------------------------------------
constexpr int BWD_TI = 64, BWD_TJ = 256;
// Rows per backward workgroup: 64, or 16 when the 64-row grid would leave most of the 256 CUs idle (e.g. the float64
// Kzz adjoint of the DSVI step: 1024 x 1024 x 2 GPs = 128 workgroups of 64 x 256 entries took 28 us, latency-bound).
static inline int bwd_rows(int64_t batch, int64_t n1, int64_t n2) {
    return batch * cdiv64(n1, BWD_TI) * cdiv64(n2, BWD_TJ) < 256 ? 16 : BWD_TI;
}

template <typename T, typename Op, int TI>
__global__ __launch_bounds__(256) void pairwise_bwd_kernel(Op op, int64_t n1, int64_t n2,
                                                           const T* __restrict__ G, int64_t ldg, int64_t sG,
                                                           T* __restrict__ P1, T* __restrict__ P2,
                                                           T* __restrict__ PG) {
    constexpr int NR = Op::NR, NC = Op::NC, NG = Op::NG;
    __shared__ T lds[256 * NC];
    __shared__ T lds_g[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t b = blockIdx.z;
    const int64_t tj = blockIdx.x, ti = blockIdx.y;
    const int64_t ntj = gridDim.x, nti = gridDim.y;
    const int64_t j0 = tj * BWD_TJ, i0 = ti * TI;
    const T* Gb = G + b * sG;

    typename Op::P cols[4];
    bool cok[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int64_t j = j0 + c * 64 + lane;
        cok[c] = j < n2;
        cols[c] = op.col(b, cok[c] ? j : n2 - 1);
    }
    T ca[4][NC];
    T ga[NG];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int k = 0; k < NC; ++k) ca[c][k] = T(0);
#pragma unroll
    for (int k = 0; k < NG; ++k) ga[k] = T(0);

    for (int r = w; r < TI; r += 4) {
        const int64_t i = i0 + r;
        if (i >= n1) break;                         // wave-uniform
        const typename Op::P rp = op.row(b, i);
        T ra[NR];
#pragma unroll
        for (int k = 0; k < NR; ++k) ra[k] = T(0);
        T gv[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) gv[c] = cok[c] ? Gb[i * ldg + j0 + c * 64 + lane] : T(0);
#pragma unroll
        for (int c = 0; c < 4; ++c) op.grad(b, rp, cols[c], gv[c], ra, ca[c], ga);
#pragma unroll
        for (int k = 0; k < NR; ++k) ra[k] = wave_sum(ra[k]);
        if (lane == 0) {
            T* dst = P1 + ((b * ntj + tj) * n1 + i) * NR;
#pragma unroll
            for (int k = 0; k < NR; ++k) dst[k] = ra[k];
        }
    }
    // column side: combine the 4 waves through LDS, one chunk of 64 columns at a time
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        for (int ww = 0; ww < 4; ++ww) {
            if (w == ww) {
#pragma unroll
                for (int k = 0; k < NC; ++k) {
                    T* p = &lds[(c * 64 + lane) * NC + k];
                    *p = (ww == 0) ? ca[c][k] : (*p + ca[c][k]);
                }
            }
            __syncthreads();
        }
    }
    {
        const int64_t j = j0 + threadIdx.x;
        if (j < n2) {
            T* dst = P2 + ((b * nti + ti) * n2 + j) * NC;
#pragma unroll
            for (int k = 0; k < NC; ++k) dst[k] = lds[threadIdx.x * NC + k];
        }
    }
#pragma unroll
    for (int k = 0; k < NG; ++k) {
        const T s = block_sum_256(ga[k], lds_g);
        if (threadIdx.x == 0) PG[((b * nti + ti) * ntj + tj) * NG + k] = s;
    }
}

// out descriptor for pass 2: value k of item i goes to ptr[k][b*bstride[k] + i*stride[k]] * scale
template <typename T> struct OutDesc {
    T* ptr[2 * NSGP_MAX_DIM];
    int64_t stride[2 * NSGP_MAX_DIM];
    int64_t bstride[2 * NSGP_MAX_DIM];
    const T* div[2 * NSGP_MAX_DIM];       // optional per-batch divisor array (RBF: lengthscale), index b*divstride+divoff
    int64_t divstride[2 * NSGP_MAX_DIM];
    int64_t divoff[2 * NSGP_MAX_DIM];
    int nval;
};

// out[item] = sum_t P[t][item]: a workgroup is 32 consecutive items x 8 partial-sum chains (the tile partials of an
// item are `ntiles` strided loads; one serial chain of 160 dependent-latency loads took 40 us at n2 = 40960), combined
// through LDS in a fixed order (deterministic).
template <typename T>
__device__ __forceinline__ void reduce_items_body(const T* __restrict__ P, int64_t ntiles, int64_t n,
                                                  const OutDesc<T>& od, int64_t blk, int64_t b, T (*lds)[33],
                                                  const T* __restrict__ Pb = nullptr, int64_t ntb = 0) {
    const int ix = threadIdx.x & 31, cy = threadIdx.x >> 5;
    const int64_t idx = blk * 32 + ix;
    const int64_t tot = n * od.nval;
    T s = T(0);
    if (idx < tot) {
        const T* p = P + b * ntiles * tot + idx;
        for (int64_t t = cy; t < ntiles; t += 8) s += p[t * tot];
        if (Pb) {                                         // symmetric case: the other side's partials of the same item
            const T* q = Pb + b * ntb * tot + idx;
            for (int64_t t = cy; t < ntb; t += 8) s += q[t * tot];
        }
    }
    lds[cy][ix] = s;
    __syncthreads();
    if (cy != 0 || idx >= tot) return;
    const int64_t i = idx / od.nval;
    const int k = (int)(idx % od.nval);
    if (!od.ptr[k]) return;
#pragma unroll
    for (int c = 1; c < 8; ++c) s += lds[c][ix];
    if (od.div[k]) s /= od.div[k][b * od.divstride[k] + od.divoff[k]];
    od.ptr[k][b * od.bstride[k] + i * od.stride[k]] = s;
}

// ONE launch for the three reductions behind a backward tile pass (workgroups [0, nblk1): row items, [nblk1, nblk1 +
// nblk2): column items, the last one: the global values of this batch entry, summing `nparts` partials of `nval` values).
template <typename T>
__global__ __launch_bounds__(256) void reduce_all_kernel(const T* __restrict__ P1, int64_t ntj, int64_t n1,
                                                         OutDesc<T> rows, const T* __restrict__ P2, int64_t nti,
                                                         int64_t n2, OutDesc<T> cols, const T* __restrict__ PG,
                                                         int64_t nparts, OutDesc<T> globs, int64_t nblk1,
                                                         int64_t nblk2) {
    __shared__ T lds[8][33];
    const int64_t b = blockIdx.z, blk = blockIdx.x;
    if (blk < nblk1) { reduce_items_body<T>(P1, ntj, n1, rows, blk, b, lds, nblk2 == 0 ? P2 : nullptr, nti); return; }
    if (blk < nblk1 + nblk2) { reduce_items_body<T>(P2, nti, n2, cols, blk - nblk1, b, lds); return; }
    T* l4 = &lds[0][0];
    for (int k = 0; k < globs.nval; ++k) {
        T s = T(0);
        for (int64_t t = threadIdx.x; t < nparts; t += 256) s += PG[(b * nparts + t) * globs.nval + k];
        s = block_sum_256(s, l4);
        if (threadIdx.x == 0 && globs.ptr[k]) {
            if (globs.div[k]) s /= globs.div[k][b * globs.divstride[k] + globs.divoff[k]];
            globs.ptr[k][b * globs.bstride[k]] = s;
        }
        __syncthreads();
    }
}

template <typename Op> size_t bwd_ws_elems(int64_t batch, int64_t n1, int64_t n2) {
    const int64_t ntj = cdiv64(n2, BWD_TJ), nti = cdiv64(n1, bwd_rows(batch, n1, n2));
    return (size_t)(batch * (ntj * n1 * Op::NR + nti * n2 * Op::NC + nti * ntj * Op::NG));
}

template <typename T, typename Op>
int launch_bwd(const Op& op, int64_t batch, int64_t n1, int64_t n2, const T* G, int64_t ldg, int64_t sG,
               const OutDesc<T>& rows, const OutDesc<T>& cols, const OutDesc<T>& globs,
               void* ws, size_t ws_bytes, void* stream) {
    if (n1 == 0 || n2 == 0 || batch == 0) return 0;
    const int ti_rows = bwd_rows(batch, n1, n2);
    const int64_t ntj = cdiv64(n2, BWD_TJ), nti = cdiv64(n1, ti_rows);
    if (ws_bytes < bwd_ws_elems<Op>(batch, n1, n2) * sizeof(T) || !ws) return -100;
    T* P1 = (T*)ws;
    T* P2 = P1 + batch * ntj * n1 * Op::NR;
    T* PG = P2 + batch * nti * n2 * Op::NC;
    hipStream_t st = (hipStream_t)stream;
    if (ti_rows == 16)
        hipLaunchKernelGGL((pairwise_bwd_kernel<T, Op, 16>), dim3((unsigned)ntj, (unsigned)nti, (unsigned)batch),
                           dim3(256), 0, st, op, n1, n2, G, ldg, sG, P1, P2, PG);
    else
        hipLaunchKernelGGL((pairwise_bwd_kernel<T, Op, BWD_TI>), dim3((unsigned)ntj, (unsigned)nti, (unsigned)batch),
                           dim3(256), 0, st, op, n1, n2, G, ldg, sG, P1, P2, PG);
    // x1 == x2 with ONE gradient buffer for both sides (rows and cols describe the same outputs): the row-side blocks
    // add the column-side partials of their item and no column-side blocks run
    bool sym = n1 == n2 && Op::NR == Op::NC;
    bool any = false;
    for (int k = 0; sym && k < Op::NR; ++k) {
        sym = rows.ptr[k] == cols.ptr[k] && rows.stride[k] == cols.stride[k] && rows.bstride[k] == cols.bstride[k];
        any = any || rows.ptr[k] != nullptr;
    }
    sym = sym && any;
    const int64_t nblk1 = cdiv64(n1 * Op::NR, 32), nblk2 = sym ? 0 : cdiv64(n2 * Op::NC, 32);
    hipLaunchKernelGGL((reduce_all_kernel<T>), dim3((unsigned)(nblk1 + nblk2 + (globs.nval > 0 ? 1 : 0)), 1, (unsigned)batch),
                       dim3(256), 0, st, (const T*)P1, ntj, n1, rows, (const T*)(sym || nblk2 ? P2 : nullptr), nti, n2, cols,
                       (const T*)PG, nti * ntj, globs, nblk1, nblk2);
    return nsgp_launch_status();
}

template <typename T> OutDesc<T> empty_desc(int nval) {
    OutDesc<T> d;
    for (int k = 0; k < 2 * NSGP_MAX_DIM; ++k) {
        d.ptr[k] = nullptr; d.stride[k] = 0; d.bstride[k] = 0; d.div[k] = nullptr; d.divstride[k] = 0; d.divoff[k] = 0;
    }
    d.nval = nval;
    return d;
}

// ------------------------------------------------------------------------------------------
// typed entry points
// ------------------------------------------------------------------------------------------
template <typename T, int D>
int gibbs_fwd_d(const T* x1, const T* x2, const T* l1, const T* l2, int64_t n1, int64_t n2, int Drt, const T* os,
                const T* diag_add, T* K, int64_t ldk, void* stream) {
    GibbsOp<T, D> op{x1, x2, l1, l2, n1, n2, Drt, os};
    return launch_fwd<T>(op, 1, n1, n2, T(0), diag_add, K, ldk, 0, stream);
}

template <typename T>
int gibbs_fwd(const T* x1, const T* x2, const T* l1, const T* l2, int64_t n1, int64_t n2, int D, const T* os,
              const T* diag_add, T* K, int64_t ldk, void* stream) {
    if (!x1) return -1; if (!x2) return -2; if (!l1) return -3; if (!l2) return -4;
    if (n1 < 0) return -5; if (n2 < 0) return -6; if (D < 1 || D > NSGP_MAX_DIM) return -7;
    if (!K && n1 * n2 > 0) return -10; if (ldk < n2) return -11;
    switch (D) {
        case 1: return gibbs_fwd_d<T, 1>(x1, x2, l1, l2, n1, n2, D, os, diag_add, K, ldk, stream);
        case 2: return gibbs_fwd_d<T, 2>(x1, x2, l1, l2, n1, n2, D, os, diag_add, K, ldk, stream);
        case 3: return gibbs_fwd_d<T, 3>(x1, x2, l1, l2, n1, n2, D, os, diag_add, K, ldk, stream);
        default: return gibbs_fwd_d<T, 0>(x1, x2, l1, l2, n1, n2, D, os, diag_add, K, ldk, stream);
    }
}

template <typename T, int D>
int gibbs_bwd_d(const T* x1, const T* x2, const T* l1, const T* l2, int64_t n1, int64_t n2, int Drt, const T* os,
                const T* G, int64_t ldg, T* g_l1, T* g_l2, T* g_x1, T* g_x2, T* g_os, void* ws, size_t wsb,
                void* stream) {
    using Op = GibbsOp<T, D>;
    Op op{x1, x2, l1, l2, n1, n2, Drt, os};
    constexpr int DM = Op::DM;
    OutDesc<T> rows = empty_desc<T>(Op::NR), cols = empty_desc<T>(Op::NC), globs = empty_desc<T>(Op::NG);
    for (int d = 0; d < Drt; ++d) {
        if (g_l1) { rows.ptr[d] = g_l1 + (int64_t)d * n1; rows.stride[d] = 1; }
        if (g_x1) { rows.ptr[DM + d] = g_x1 + d; rows.stride[DM + d] = Drt; }
        if (g_l2) { cols.ptr[d] = g_l2 + (int64_t)d * n2; cols.stride[d] = 1; }
        if (g_x2) { cols.ptr[DM + d] = g_x2 + d; cols.stride[DM + d] = Drt; }
    }
    globs.ptr[0] = g_os;
    return launch_bwd<T>(op, 1, n1, n2, G, ldg, 0, rows, cols, globs, ws, wsb, stream);
}

template <typename T>
int gibbs_bwd(const T* x1, const T* x2, const T* l1, const T* l2, int64_t n1, int64_t n2, int D, const T* os, const T* G,
              int64_t ldg, T* g_l1, T* g_l2, T* g_x1, T* g_x2, T* g_os, void* ws, size_t wsb, void* stream) {
    if (!x1) return -1; if (!x2) return -2; if (!l1) return -3; if (!l2) return -4;
    if (n1 < 0) return -5; if (n2 < 0) return -6; if (D < 1 || D > NSGP_MAX_DIM) return -7;
    if (!G && n1 * n2 > 0) return -9; if (ldg < n2) return -10;
    switch (D) {
        case 1: return gibbs_bwd_d<T, 1>(x1, x2, l1, l2, n1, n2, D, os, G, ldg, g_l1, g_l2, g_x1, g_x2, g_os, ws, wsb, stream);
        case 2: return gibbs_bwd_d<T, 2>(x1, x2, l1, l2, n1, n2, D, os, G, ldg, g_l1, g_l2, g_x1, g_x2, g_os, ws, wsb, stream);
        case 3: return gibbs_bwd_d<T, 3>(x1, x2, l1, l2, n1, n2, D, os, G, ldg, g_l1, g_l2, g_x1, g_x2, g_os, ws, wsb, stream);
        default: return gibbs_bwd_d<T, 0>(x1, x2, l1, l2, n1, n2, D, os, G, ldg, g_l1, g_l2, g_x1, g_x2, g_os, ws, wsb, stream);
    }
}

template <typename T, int D>
int rbf_fwd_d(const T* x1, const T* x2, const T* ls, const T* os, int64_t batch, int64_t n1, int64_t n2, int Drt,
              int64_t sx1, int64_t sx2, T diag_add, T* K, int64_t ldk, int64_t sK, void* stream) {
    RbfOp<T, D> op{x1, x2, ls, os, n1, n2, sx1, sx2, Drt};
    return launch_fwd<T>(op, batch, n1, n2, diag_add, (const T*)nullptr, K, ldk, sK, stream);
}

template <typename T>
int rbf_fwd(const T* x1, const T* x2, const T* ls, const T* os, int64_t batch, int64_t n1, int64_t n2, int D,
            int64_t sx1, int64_t sx2, T diag_add, T* K, int64_t ldk, int64_t sK, void* stream) {
    if (!x1) return -1; if (!x2) return -2; if (!ls) return -3; if (!os) return -4;
    if (batch < 0) return -5; if (n1 < 0) return -6; if (n2 < 0) return -7; if (D < 1 || D > NSGP_MAX_DIM) return -8;
    if (!K && batch * n1 * n2 > 0) return -12; if (ldk < n2) return -13;
    switch (D) {
        case 1: return rbf_fwd_d<T, 1>(x1, x2, ls, os, batch, n1, n2, D, sx1, sx2, diag_add, K, ldk, sK, stream);
        case 2: return rbf_fwd_d<T, 2>(x1, x2, ls, os, batch, n1, n2, D, sx1, sx2, diag_add, K, ldk, sK, stream);
        case 3: return rbf_fwd_d<T, 3>(x1, x2, ls, os, batch, n1, n2, D, sx1, sx2, diag_add, K, ldk, sK, stream);
        default: return rbf_fwd_d<T, 0>(x1, x2, ls, os, batch, n1, n2, D, sx1, sx2, diag_add, K, ldk, sK, stream);
    }
}

template <typename T, int D>
int rbf_bwd_d(const T* x1, const T* x2, const T* ls, const T* os, int64_t batch, int64_t n1, int64_t n2, int Drt,
              int64_t sx1, int64_t sx2, const T* G, int64_t ldg, int64_t sG, T* g_x1, T* g_x2, T* g_ls, T* g_os,
              void* ws, size_t wsb, void* stream) {
    using Op = RbfOp<T, D>;
    Op op{x1, x2, ls, os, n1, n2, sx1, sx2, Drt};
    constexpr int DM = Op::DM;
    OutDesc<T> rows = empty_desc<T>(Op::NR), cols = empty_desc<T>(Op::NC), globs = empty_desc<T>(Op::NG);
    for (int d = 0; d < Drt; ++d) {
        if (g_x1) { rows.ptr[d] = g_x1 + d; rows.stride[d] = Drt; rows.bstride[d] = n1 * Drt;
                    rows.div[d] = ls; rows.divstride[d] = Drt; rows.divoff[d] = d; }
        if (g_x2) { cols.ptr[d] = g_x2 + d; cols.stride[d] = Drt; cols.bstride[d] = n2 * Drt;
                    cols.div[d] = ls; cols.divstride[d] = Drt; cols.divoff[d] = d; }
        if (g_ls) { globs.ptr[d] = g_ls + d; globs.bstride[d] = Drt;
                    globs.div[d] = ls; globs.divstride[d] = Drt; globs.divoff[d] = d; }
    }
    if (g_os) { globs.ptr[DM] = g_os; globs.bstride[DM] = 1; }
    return launch_bwd<T>(op, batch, n1, n2, G, ldg, sG, rows, cols, globs, ws, wsb, stream);
}

template <typename T>
int rbf_bwd(const T* x1, const T* x2, const T* ls, const T* os, int64_t batch, int64_t n1, int64_t n2, int D,
            int64_t sx1, int64_t sx2, const T* G, int64_t ldg, int64_t sG, T* g_x1, T* g_x2, T* g_ls, T* g_os,
            void* ws, size_t wsb, void* stream) {
    if (!x1) return -1; if (!x2) return -2; if (!ls) return -3; if (!os) return -4;
    if (batch < 0) return -5; if (n1 < 0) return -6; if (n2 < 0) return -7; if (D < 1 || D > NSGP_MAX_DIM) return -8;
    if (!G && batch * n1 * n2 > 0) return -11; if (ldg < n2) return -12;
    switch (D) {
        case 1: return rbf_bwd_d<T, 1>(x1, x2, ls, os, batch, n1, n2, D, sx1, sx2, G, ldg, sG, g_x1, g_x2, g_ls, g_os, ws, wsb, stream);
        case 2: return rbf_bwd_d<T, 2>(x1, x2, ls, os, batch, n1, n2, D, sx1, sx2, G, ldg, sG, g_x1, g_x2, g_ls, g_os, ws, wsb, stream);
        case 3: return rbf_bwd_d<T, 3>(x1, x2, ls, os, batch, n1, n2, D, sx1, sx2, G, ldg, sG, g_x1, g_x2, g_ls, g_os, ws, wsb, stream);
        default: return rbf_bwd_d<T, 0>(x1, x2, ls, os, batch, n1, n2, D, sx1, sx2, G, ldg, sG, g_x1, g_x2, g_ls, g_os, ws, wsb, stream);
    }
}

template <typename T>
int ps_fwd(const T* x1, const T* x2, const T* s1, const T* s2, int64_t n1, int64_t n2, T jit, T* K, int64_t ldk,
           void* stream) {
    if (!x1) return -1; if (!x2) return -2; if (!s1) return -3; if (!s2) return -4;
    if (n1 < 0) return -5; if (n2 < 0) return -6; if (!K && n1 * n2 > 0) return -8; if (ldk < n2) return -9;
    PsOp<T> op{x1, x2, s1, s2, jit};
    return launch_fwd<T>(op, 1, n1, n2, T(0), (const T*)nullptr, K, ldk, 0, stream);
}

template <typename T>
int ps_bwd(const T* x1, const T* x2, const T* s1, const T* s2, int64_t n1, int64_t n2, T jit, const T* G,
           int64_t ldg, T* g_s1, T* g_s2, void* ws, size_t wsb, void* stream) {
    if (!x1) return -1; if (!x2) return -2; if (!s1) return -3; if (!s2) return -4;
    if (n1 < 0) return -5; if (n2 < 0) return -6; if (!G && n1 * n2 > 0) return -8; if (ldg < n2) return -9;
    using Op = PsOp<T>;
    Op op{x1, x2, s1, s2, jit};
    OutDesc<T> rows = empty_desc<T>(4), cols = empty_desc<T>(4), globs = empty_desc<T>(0);
    for (int k = 0; k < 4; ++k) {
        if (g_s1) { rows.ptr[k] = g_s1 + k; rows.stride[k] = 4; }
        if (g_s2) { cols.ptr[k] = g_s2 + k; cols.stride[k] = 4; }
    }
    return launch_bwd<T>(op, 1, n1, n2, G, ldg, 0, rows, cols, globs, ws, wsb, stream);
}

template <typename T, int D>
int rbfper_fwd_d(const T* x1, const T* x2, const T* lsr, const T* lsp, const T* per, const T* os, int64_t batch,
                 int64_t n1, int64_t n2, int Drt, int64_t sx1, int64_t sx2, T diag_add, T* K, int64_t ldk, int64_t sK,
                 void* stream) {
    RbfPeriodicOp<T, D> op{x1, x2, lsr, lsp, per, os, n1, n2, sx1, sx2, Drt};
    return launch_fwd<T>(op, batch, n1, n2, diag_add, (const T*)nullptr, K, ldk, sK, stream);
}
template <typename T>
int rbfper_fwd(const T* x1, const T* x2, const T* lsr, const T* lsp, const T* per, const T* os, int64_t batch,
               int64_t n1, int64_t n2, int D, int64_t sx1, int64_t sx2, T diag_add, T* K, int64_t ldk, int64_t sK,
               void* stream) {
    if (!x1) return -1; if (!x2) return -2; if (!lsp) return -4; if (!per) return -5;
    if (batch < 0) return -7; if (n1 < 0) return -8; if (n2 < 0) return -9; if (D < 1 || D > NSGP_MAX_DIM) return -10;
    if (!K && batch * n1 * n2 > 0) return -14; if (ldk < n2) return -15;
    switch (D) {
        case 1: return rbfper_fwd_d<T, 1>(x1, x2, lsr, lsp, per, os, batch, n1, n2, D, sx1, sx2, diag_add, K, ldk, sK, stream);
        case 2: return rbfper_fwd_d<T, 2>(x1, x2, lsr, lsp, per, os, batch, n1, n2, D, sx1, sx2, diag_add, K, ldk, sK, stream);
        default: return rbfper_fwd_d<T, 0>(x1, x2, lsr, lsp, per, os, batch, n1, n2, D, sx1, sx2, diag_add, K, ldk, sK, stream);
    }
}
template <typename T, int D>
int rbfper_bwd_d(const T* x1, const T* x2, const T* lsr, const T* lsp, const T* per, const T* os, int64_t batch,
                 int64_t n1, int64_t n2, int Drt, int64_t sx1, int64_t sx2, const T* G, int64_t ldg, int64_t sG,
                 T* g_x1, T* g_x2, T* g_lsr, T* g_lsp, T* g_per, T* g_os, void* ws, size_t wsb, void* stream) {
    using Op = RbfPeriodicOp<T, D>;
    Op op{x1, x2, lsr, lsp, per, os, n1, n2, sx1, sx2, Drt};
    constexpr int DM = Op::DM;
    OutDesc<T> rows = empty_desc<T>(Op::NR), cols = empty_desc<T>(Op::NC), globs = empty_desc<T>(Op::NG);
    for (int d = 0; d < Drt; ++d) {
        if (g_x1) { rows.ptr[d] = g_x1 + d; rows.stride[d] = Drt; rows.bstride[d] = n1 * Drt; }
        if (g_x2) { cols.ptr[d] = g_x2 + d; cols.stride[d] = Drt; cols.bstride[d] = n2 * Drt; }
        if (g_lsr && lsr) { globs.ptr[d] = g_lsr + d; globs.bstride[d] = Drt; }
    }
    if (g_lsp) { globs.ptr[DM] = g_lsp; globs.bstride[DM] = 1; }
    if (g_per) { globs.ptr[DM + 1] = g_per; globs.bstride[DM + 1] = 1; }
    if (g_os) { globs.ptr[DM + 2] = g_os; globs.bstride[DM + 2] = 1; }
    return launch_bwd<T>(op, batch, n1, n2, G, ldg, sG, rows, cols, globs, ws, wsb, stream);
}
template <typename T>
int rbfper_bwd(const T* x1, const T* x2, const T* lsr, const T* lsp, const T* per, const T* os, int64_t batch,
               int64_t n1, int64_t n2, int D, int64_t sx1, int64_t sx2, const T* G, int64_t ldg, int64_t sG, T* g_x1,
               T* g_x2, T* g_lsr, T* g_lsp, T* g_per, T* g_os, void* ws, size_t wsb, void* stream) {
    if (!x1) return -1; if (!x2) return -2; if (!lsp) return -4; if (!per) return -5;
    if (batch < 0) return -7; if (n1 < 0) return -8; if (n2 < 0) return -9; if (D < 1 || D > NSGP_MAX_DIM) return -10;
    if (!G && batch * n1 * n2 > 0) return -13; if (ldg < n2) return -14;
    switch (D) {
        case 1: return rbfper_bwd_d<T, 1>(x1, x2, lsr, lsp, per, os, batch, n1, n2, D, sx1, sx2, G, ldg, sG, g_x1, g_x2, g_lsr, g_lsp, g_per, g_os, ws, wsb, stream);
        case 2: return rbfper_bwd_d<T, 2>(x1, x2, lsr, lsp, per, os, batch, n1, n2, D, sx1, sx2, G, ldg, sG, g_x1, g_x2, g_lsr, g_lsp, g_per, g_os, ws, wsb, stream);
        default: return rbfper_bwd_d<T, 0>(x1, x2, lsr, lsp, per, os, batch, n1, n2, D, sx1, sx2, G, ldg, sG, g_x1, g_x2, g_lsr, g_lsp, g_per, g_os, ws, wsb, stream);
    }
}

}  // namespace

extern "C" {

int nsgp_gibbs_build_fwd_f32(const float* x1, const float* x2, const float* l1, const float* l2, int64_t n1,
                             int64_t n2, int D, const float* os, const float* diag_add, float* K, int64_t ldk,
                             void* stream) {
    return gibbs_fwd<float>(x1, x2, l1, l2, n1, n2, D, os, diag_add, K, ldk, stream);
}
int nsgp_gibbs_build_fwd_f64(const double* x1, const double* x2, const double* l1, const double* l2, int64_t n1,
                             int64_t n2, int D, const double* os, const double* diag_add, double* K, int64_t ldk,
                             void* stream) {
    return gibbs_fwd<double>(x1, x2, l1, l2, n1, n2, D, os, diag_add, K, ldk, stream);
}
size_t nsgp_gibbs_build_bwd_workspace(int64_t n1, int64_t n2, int D, int elem_size) {
    (void)D;                                    // sized for the generic (NSGP_MAX_DIM) functor
    return bwd_ws_elems<GibbsOp<double, 0>>(1, n1, n2) * (size_t)elem_size + 256;
}
int nsgp_gibbs_build_bwd_f32(const float* x1, const float* x2, const float* l1, const float* l2, int64_t n1,
                             int64_t n2, int D, const float* os, const float* G, int64_t ldg, float* g_l1, float* g_l2,
                             float* g_x1, float* g_x2, float* g_os, void* ws, size_t wsb, void* stream) {
    return gibbs_bwd<float>(x1, x2, l1, l2, n1, n2, D, os, G, ldg, g_l1, g_l2, g_x1, g_x2, g_os, ws, wsb, stream);
}
int nsgp_gibbs_build_bwd_f64(const double* x1, const double* x2, const double* l1, const double* l2, int64_t n1,
                             int64_t n2, int D, const double* os, const double* G, int64_t ldg, double* g_l1,
                             double* g_l2, double* g_x1, double* g_x2, double* g_os, void* ws, size_t wsb,
                             void* stream) {
    return gibbs_bwd<double>(x1, x2, l1, l2, n1, n2, D, os, G, ldg, g_l1, g_l2, g_x1, g_x2, g_os, ws, wsb, stream);
}

int nsgp_rbf_build_fwd_f32(const float* x1, const float* x2, const float* ls, const float* os, int64_t batch,
                           int64_t n1, int64_t n2, int D, int64_t sx1, int64_t sx2, float diag_add, float* K,
                           int64_t ldk, int64_t sK, void* stream) {
    return rbf_fwd<float>(x1, x2, ls, os, batch, n1, n2, D, sx1, sx2, diag_add, K, ldk, sK, stream);
}
int nsgp_rbf_build_fwd_f64(const double* x1, const double* x2, const double* ls, const double* os, int64_t batch,
                           int64_t n1, int64_t n2, int D, int64_t sx1, int64_t sx2, double diag_add, double* K,
                           int64_t ldk, int64_t sK, void* stream) {
    return rbf_fwd<double>(x1, x2, ls, os, batch, n1, n2, D, sx1, sx2, diag_add, K, ldk, sK, stream);
}
size_t nsgp_rbf_build_bwd_workspace(int64_t batch, int64_t n1, int64_t n2, int D, int elem_size) {
    (void)D;
    return bwd_ws_elems<RbfOp<double, 0>>(batch, n1, n2) * (size_t)elem_size + 256;
}
int nsgp_rbf_build_bwd_f32(const float* x1, const float* x2, const float* ls, const float* os, int64_t batch,
                           int64_t n1, int64_t n2, int D, int64_t sx1, int64_t sx2, const float* G, int64_t ldg,
                           int64_t sG, float* g_x1, float* g_x2, float* g_ls, float* g_os, void* ws, size_t wsb,
                           void* stream) {
    return rbf_bwd<float>(x1, x2, ls, os, batch, n1, n2, D, sx1, sx2, G, ldg, sG, g_x1, g_x2, g_ls, g_os, ws, wsb, stream);
}
int nsgp_rbf_build_bwd_f64(const double* x1, const double* x2, const double* ls, const double* os, int64_t batch,
                           int64_t n1, int64_t n2, int D, int64_t sx1, int64_t sx2, const double* G, int64_t ldg,
                           int64_t sG, double* g_x1, double* g_x2, double* g_ls, double* g_os, void* ws,
                           size_t wsb, void* stream) {
    return rbf_bwd<double>(x1, x2, ls, os, batch, n1, n2, D, sx1, sx2, G, ldg, sG, g_x1, g_x2, g_ls, g_os, ws, wsb, stream);
}

int nsgp_ps2d_build_fwd_f32(const float* x1, const float* x2, const float* s1, const float* s2, int64_t n1,
                            int64_t n2, float jit, float* K, int64_t ldk, void* stream) {
    return ps_fwd<float>(x1, x2, s1, s2, n1, n2, jit, K, ldk, stream);
}
int nsgp_ps2d_build_fwd_f64(const double* x1, const double* x2, const double* s1, const double* s2, int64_t n1,
                            int64_t n2, double jit, double* K, int64_t ldk, void* stream) {
    return ps_fwd<double>(x1, x2, s1, s2, n1, n2, jit, K, ldk, stream);
}
size_t nsgp_ps2d_build_bwd_workspace(int64_t n1, int64_t n2, int elem_size) {
    return bwd_ws_elems<PsOp<double>>(1, n1, n2) * (size_t)elem_size + 256;
}
int nsgp_ps2d_build_bwd_f32(const float* x1, const float* x2, const float* s1, const float* s2, int64_t n1,
                            int64_t n2, float jit, const float* G, int64_t ldg, float* g_s1, float* g_s2, void* ws,
                            size_t wsb, void* stream) {
    return ps_bwd<float>(x1, x2, s1, s2, n1, n2, jit, G, ldg, g_s1, g_s2, ws, wsb, stream);
}
int nsgp_ps2d_build_bwd_f64(const double* x1, const double* x2, const double* s1, const double* s2, int64_t n1,
                            int64_t n2, double jit, const double* G, int64_t ldg, double* g_s1, double* g_s2,
                            void* ws, size_t wsb, void* stream) {
    return ps_bwd<double>(x1, x2, s1, s2, n1, n2, jit, G, ldg, g_s1, g_s2, ws, wsb, stream);
}


int nsgp_rbf_periodic_build_fwd_f32(const float* x1, const float* x2, const float* ls_rbf, const float* ls_per,
                                    const float* period, const float* os, int64_t batch, int64_t n1, int64_t n2, int D,
                                    int64_t sx1, int64_t sx2, float diag_add, float* K, int64_t ldk, int64_t sK,
                                    void* stream) {
    return rbfper_fwd<float>(x1, x2, ls_rbf, ls_per, period, os, batch, n1, n2, D, sx1, sx2, diag_add, K, ldk, sK, stream);
}
int nsgp_rbf_periodic_build_fwd_f64(const double* x1, const double* x2, const double* ls_rbf, const double* ls_per,
                                    const double* period, const double* os, int64_t batch, int64_t n1, int64_t n2,
                                    int D, int64_t sx1, int64_t sx2, double diag_add, double* K, int64_t ldk,
                                    int64_t sK, void* stream) {
    return rbfper_fwd<double>(x1, x2, ls_rbf, ls_per, period, os, batch, n1, n2, D, sx1, sx2, diag_add, K, ldk, sK, stream);
}
size_t nsgp_rbf_periodic_build_bwd_workspace(int64_t batch, int64_t n1, int64_t n2, int D, int elem_size) {
    (void)D;
    return bwd_ws_elems<RbfPeriodicOp<double, 0>>(batch, n1, n2) * (size_t)elem_size + 256;
}
int nsgp_rbf_periodic_build_bwd_f32(const float* x1, const float* x2, const float* ls_rbf, const float* ls_per,
                                    const float* period, const float* os, int64_t batch, int64_t n1, int64_t n2, int D,
                                    int64_t sx1, int64_t sx2, const float* G, int64_t ldg, int64_t sG, float* g_x1,
                                    float* g_x2, float* g_ls_rbf, float* g_ls_per, float* g_period, float* g_os,
                                    void* ws, size_t wsb, void* stream) {
    return rbfper_bwd<float>(x1, x2, ls_rbf, ls_per, period, os, batch, n1, n2, D, sx1, sx2, G, ldg, sG, g_x1, g_x2,
                             g_ls_rbf, g_ls_per, g_period, g_os, ws, wsb, stream);
}
int nsgp_rbf_periodic_build_bwd_f64(const double* x1, const double* x2, const double* ls_rbf, const double* ls_per,
                                    const double* period, const double* os, int64_t batch, int64_t n1, int64_t n2,
                                    int D, int64_t sx1, int64_t sx2, const double* G, int64_t ldg, int64_t sG,
                                    double* g_x1, double* g_x2, double* g_ls_rbf, double* g_ls_per, double* g_period,
                                    double* g_os, void* ws, size_t wsb, void* stream) {
    return rbfper_bwd<double>(x1, x2, ls_rbf, ls_per, period, os, batch, n1, n2, D, sx1, sx2, G, ldg, sG, g_x1, g_x2,
                              g_ls_rbf, g_ls_per, g_period, g_os, ws, wsb, stream);
}

}  // extern "C"
