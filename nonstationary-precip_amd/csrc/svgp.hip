// svgp.hip -- K6 SVGP epilogue / DeepGPLayer sampling and K7 ELBO reductions.
//
// gpytorch semantics restated (gpytorch itself is absent):
//   VariationalStrategy.forward : mean = A^T m + mu(x),  var = kxx + 1e-4 + colsum(A o ((S-I)A))
//                                 with S = Lq Lq^T  ->  colsum(C o C) - colsum(A o A), C = Lq^T A
//   DeepGPLayer.__call__        : h = mean + sqrt(var) * eps  (diagonal sampling)
//   GaussianLikelihood.expected_log_prob, VariationalELBO, DeepApproximateMLL, KL(q(u) || N(0,I))
// driven by models/dgps.py:48-51,92-98 and experiments/deepgp_spatial_bench.py:61,84-88.
// All of these are HBM-bound streaming passes over (M x n) or (S x n) data.
#include "common.h"

namespace {

// ---- colstats: mean[b,j] = sum_k A m ; var[b,j] = base[b] + sum_k (C^2 - A^2) ------------------
template <typename T>
__global__ __launch_bounds__(256) void colstats_kernel(const T* __restrict__ A, const T* __restrict__ C,
                                                       const T* __restrict__ m, const T* __restrict__ base,
                                                       int64_t M, int64_t n, T* __restrict__ mean,
                                                       T* __restrict__ var) {
    __shared__ T sm[4][64], sv[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t b = blockIdx.y, j = (int64_t)blockIdx.x * 64 + lane;
    const T* Ab = A + b * M * n;
    const T* Cb = C + b * M * n;
    const T* mb = m + b * M;
    T am = T(0), av = T(0);
    if (j < n) {
        const int64_t kq = (M + 3) / 4;
        const int64_t k0 = w * kq, k1 = (k0 + kq) < M ? (k0 + kq) : M;
        for (int64_t k = k0; k < k1; ++k) {
            const T a = Ab[k * n + j], c = Cb[k * n + j];
            am += a * mb[k];
            av += c * c - a * a;
        }
    }
    sm[w][lane] = am; sv[w][lane] = av;
    __syncthreads();
    if (w == 0 && j < n) {
        mean[b * n + j] = sm[0][lane] + sm[1][lane] + sm[2][lane] + sm[3][lane];
        var[b * n + j] = base[b] + (sv[0][lane] + sv[1][lane] + sv[2][lane] + sv[3][lane]);
    }
}

// ---- colstats from the GEMM epilogues' per-tile-row partial sums ----------------------------------
// Optional affine prior mean added on the way out (gpytorch ConstantMean / LinearMean of models/dgps.py:40-43):
//   mean[b,j] += sum_d x[b,j,d] w[b,d] + c[b]   (x / w / c batch strides may be 0 = shared; w or c may be null).
// TP: element type of the partial buffers (float64 partials of the float64-accumulating projections with a float32 layer)
template <typename T, typename TP = T>
__global__ void colstats_finalize_kernel(const TP* __restrict__ pdot, const TP* __restrict__ psqA,
                                         const TP* __restrict__ psqC, const T* __restrict__ base, T base_add,
                                         int64_t batch, int64_t tiles, int64_t n, const T* __restrict__ x,
                                         int64_t sxb, int D, const T* __restrict__ w, int64_t swb,
                                         const T* __restrict__ c, int64_t scb, T* __restrict__ mean,
                                         T* __restrict__ var) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= batch * n) return;
    const int64_t b = idx / n, j = idx % n;
    TP sm = TP(0), sa = TP(0), sc = TP(0);
    for (int64_t t = 0; t < tiles; ++t) {
        const int64_t o = (b * tiles + t) * n + j;
        sm += pdot[o]; sa += psqA[o]; sc += psqC[o];
    }
    if (w) {
        const T* xr = x + b * sxb + j * D;
        const T* wr = w + b * swb;
        TP mu = TP(0);
        for (int d = 0; d < D; ++d) mu += (TP)xr[d] * (TP)wr[d];
        sm += mu;
    }
    if (c) sm += (TP)c[b * scb];
    mean[idx] = (T)sm;
    var[idx] = (T)(((TP)base[b] + (TP)base_add) + (sc - sa));
}

// ---- out[b][i] = sum_j A[b][i][j] * g[b][j]  (one workgroup per row; the m-gradient  A gmean) ------
template <typename T>
__global__ __launch_bounds__(256) void rowdot_kernel(const T* __restrict__ A, const T* __restrict__ g, int64_t M,
                                                     int64_t n, T* __restrict__ out) {
    __shared__ T lds[4];
    const int64_t b = blockIdx.y, i = blockIdx.x;
    const T* row = A + (b * M + i) * n;
    const T* gb = g + b * n;
    T acc = T(0);
    constexpr int V = 16 / sizeof(T);
    const bool vec = (n % V == 0) && ((uintptr_t)row % 16 == 0) && ((uintptr_t)gb % 16 == 0);
    if (vec) {
        for (int64_t j = (int64_t)threadIdx.x * V; j < n; j += 256 * V) {
            if constexpr (sizeof(T) == 4) {
                const float4 a = *reinterpret_cast<const float4*>(row + j);
                const float4 q = *reinterpret_cast<const float4*>(gb + j);
                acc += a.x * q.x + a.y * q.y + a.z * q.z + a.w * q.w;
            } else {
                const double2 a = *reinterpret_cast<const double2*>(row + j);
                const double2 q = *reinterpret_cast<const double2*>(gb + j);
                acc += a.x * q.x + a.y * q.y;
            }
        }
    } else {
        for (int64_t j = threadIdx.x; j < n; j += 256) acc += row[j] * gb[j];
    }
    acc = block_sum_256(acc, lds);
    if (threadIdx.x == 0) out[b * M + i] = acc;
}

// ---- rowdot plus the reductions over the same columns that the layer's adjoint needs anyway:
//   rows i < M        : out[b,i]  = sum_j A[b,i,j] g[b,j]                      (mbar = A gmean)
//   row  M            : out_gv[b] = sum_j gv[b,j]                              (d/d outputscale through `base`)
//   row  M+1          : out_1     = sum_j g[b,j]                               (constant mean / bias gradient)
//   rows M+2 .. M+1+D : out_x[d]  = sum_j x[b,j,d] g[b,j]                      (linear-mean weight gradient)
// out_1 / out_x are per batch, or summed over the batch when the mean parameters are shared (`shared`).
// sum of q[0..n) by one 256-thread workgroup: 16-byte loads, four of them in flight per lane per trip (block-reduced; lds: 4
// elements)
template <typename T> __device__ __forceinline__ T strided_sum4(const T* __restrict__ q, int64_t n, T* lds) {
    constexpr int V = 16 / sizeof(T);
    T a0 = T(0), a1 = T(0), a2 = T(0), a3 = T(0);
    int64_t j = 0;
    if ((uintptr_t)q % 16 == 0) {
        auto ld = [&](int64_t at) __attribute__((always_inline)) {
            T v = T(0);
            if constexpr (sizeof(T) == 4) { const float4 t = *reinterpret_cast<const float4*>(q + at); v = (t.x + t.y) + (t.z + t.w); }
            else { const double2 t = *reinterpret_cast<const double2*>(q + at); v = t.x + t.y; }
            return v;
        };
        const int64_t step = 256 * V;
        for (j = (int64_t)threadIdx.x * V; j + 3 * step + V <= n; j += 4 * step) {
            a0 += ld(j); a1 += ld(j + step); a2 += ld(j + 2 * step); a3 += ld(j + 3 * step);
        }
        for (; j + V <= n; j += step) a0 += ld(j);
        // ragged end: elements [n - n % V, n) by the first lanes
        const int64_t tail0 = n - n % V;
        if ((int64_t)threadIdx.x < n - tail0) a1 += q[tail0 + threadIdx.x];
    } else {
        for (j = threadIdx.x; j + 768 < n; j += 1024) { a0 += q[j]; a1 += q[j + 256]; a2 += q[j + 512]; a3 += q[j + 768]; }
        for (; j < n; j += 256) a0 += q[j];
    }
    return block_sum_256((a0 + a1) + (a2 + a3), lds);
}

template <typename T>
__global__ __launch_bounds__(256) void rowdot_affine_kernel(const T* __restrict__ A, const T* __restrict__ g,
                                                            const T* __restrict__ gv, const T* __restrict__ x,
                                                            int64_t sxb, int D, int shared, int64_t batch, int64_t M,
                                                            int64_t n, T* __restrict__ out, T* __restrict__ out_gv,
                                                            T* __restrict__ out_x, T* __restrict__ out_1) {
    __shared__ T lds[4];
    const int64_t b = blockIdx.y, i = blockIdx.x;
    T acc = T(0);
    if (i < M) {
        const T* row = A + (b * M + i) * n;
        const T* gb = g + b * n;
        constexpr int V = 16 / sizeof(T);
        const bool vec = (n % V == 0) && ((uintptr_t)row % 16 == 0) && ((uintptr_t)gb % 16 == 0);
        if (vec) {
            // Chunks of 256 lanes x 16 bytes, walked from a start that differs from row to row: rows are n elements apart
            // (160 KB at the headline's last layer, a multiple of the memory channels' interleave), so workgroups that all
            // begin at column 0 march through the SAME channel together.  Eight chunks in flight per lane.
            const int64_t nch = (n + 256 * V - 1) / (256 * V);
            const int64_t c0 = i % nch;
            T a1 = T(0), a2 = T(0), a3 = T(0);
            auto chunk = [&](int64_t t, T& dst) __attribute__((always_inline)) {
                int64_t c = c0 + t;
                if (c >= nch) c -= nch;
                const int64_t j = (c * 256 + threadIdx.x) * V;
                if (j < n) {
                    if constexpr (sizeof(T) == 4) {
                        const float4 a = *reinterpret_cast<const float4*>(row + j);
                        const float4 q = *reinterpret_cast<const float4*>(gb + j);
                        dst += a.x * q.x + a.y * q.y + a.z * q.z + a.w * q.w;
                    } else {
                        const double2 a = *reinterpret_cast<const double2*>(row + j);
                        const double2 q = *reinterpret_cast<const double2*>(gb + j);
                        dst += a.x * q.x + a.y * q.y;
                    }
                }
            };
            int64_t t = 0;
            for (; t + 7 < nch; t += 8) {
                chunk(t, acc); chunk(t + 1, a1); chunk(t + 2, a2); chunk(t + 3, a3);
                chunk(t + 4, acc); chunk(t + 5, a1); chunk(t + 6, a2); chunk(t + 7, a3);
            }
            for (; t < nch; ++t) chunk(t, acc);
            acc = (acc + a1) + (a2 + a3);
        } else {
            for (int64_t j = threadIdx.x; j < n; j += 256) acc += row[j] * gb[j];
        }
        acc = block_sum_256(acc, lds);
        if (threadIdx.x == 0) out[b * M + i] = acc;
        return;
    }
    const int e = (int)(i - M);
    if (e == 0) {
        if (!gv) return;
        // (the virtual rows are ONE workgroup each: with a dependent-latency loop of n / 256 scalar loads they, not the M
        // rows of A, set the launch's duration -- 87 of 100 us at n = 40960.  Four independent loads per lane per trip.)
        const T* q = gv + b * n;
        acc = strided_sum4(q, n, lds);
        if (threadIdx.x == 0) out_gv[b] = acc;
        return;
    }
    if (e == 1 ? !out_1 : !out_x) return;
    if (shared && b != 0) return;
    const int64_t b1 = shared ? batch : b + 1;
    for (int64_t bb = b; bb < b1; ++bb) {
        const T* gb = g + bb * n;
        T a0 = T(0), a1 = T(0), a2 = T(0), a3 = T(0);
        int64_t j = threadIdx.x;
        if (e == 1) {
            acc += strided_sum4(gb, n, lds);             // (block-reduced already: every lane holds the sum)
            __syncthreads();
            continue;
        }
        const T* xc = x + bb * sxb + (e - 2);
        for (; j + 1792 < n; j += 2048) {                // eight (g, x) pairs in flight per lane
            T gq[8], xq[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { gq[u] = gb[j + 256 * u]; xq[u] = xc[(j + 256 * u) * D]; }
            a0 += xq[0] * gq[0] + xq[4] * gq[4]; a1 += xq[1] * gq[1] + xq[5] * gq[5];
            a2 += xq[2] * gq[2] + xq[6] * gq[6]; a3 += xq[3] * gq[3] + xq[7] * gq[7];
        }
        for (; j < n; j += 256) a0 += xc[j * D] * gb[j];
        acc += (a0 + a1) + (a2 + a3);
    }
    if (e != 1) acc = block_sum_256(acc, lds);
    if (threadIdx.x == 0) {
        if (e == 1) out_1[shared ? 0 : b] = acc;
        else out_x[(shared ? 0 : b) * D + (e - 2)] = acc;
    }
}

// ---- colstats backward: one block per (row k, batch b) ------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void colstats_bwd_kernel(const T* __restrict__ A, const T* __restrict__ C,
                                                           const T* __restrict__ m, const T* __restrict__ gmean,
                                                           const T* __restrict__ gvar, int64_t M, int64_t n,
                                                           T* __restrict__ Abar, T* __restrict__ C2,
                                                           T* __restrict__ mbar) {
    __shared__ T lds[4];
    const int64_t b = blockIdx.y, k = blockIdx.x;
    const int64_t off = (b * M + k) * n;
    const T mk = m[b * M + k];
    const T* gm = gmean + b * n;
    const T* gv = gvar + b * n;
    T acc = T(0);
    for (int64_t j = threadIdx.x; j < n; j += 256) {
        const T a = A[off + j], g1 = gm[j], g2 = T(2) * gv[j];
        acc += a * g1;
        Abar[off + j] = mk * g1 - g2 * a;
        C2[off + j] = g2 * C[off + j];
    }
    acc = block_sum_256(acc, lds);
    if (threadIdx.x == 0) mbar[b * M + k] = acc;
}

// ---- sampling ------------------------------------------------------------------------------------
template <typename T>
__global__ void sample_fwd_kernel(const T* __restrict__ mean, const T* __restrict__ var, const T* __restrict__ eps,
                                  int64_t S, int64_t ns, int64_t n, int64_t b, T* __restrict__ h) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= S * n * b) return;
    const int64_t c = idx % b, i = (idx / b) % n, s = idx / (b * n);
    const int64_t sp = ns == 1 ? 0 : s;
    const int64_t q = (c * ns + sp) * n + i;
    h[idx] = mean[q] + t_sqrt(var[q]) * eps[idx];
}

template <typename T>
__global__ void sample_bwd_kernel(const T* __restrict__ var, const T* __restrict__ eps, const T* __restrict__ gh,
                                  int64_t S, int64_t ns, int64_t n, int64_t b, T* __restrict__ gmean,
                                  T* __restrict__ gvar) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= b * ns * n) return;
    const int64_t i = q % n, sp = (q / n) % ns, c = q / (n * ns);
    const T hs = T(0.5) / t_sqrt(var[q]);
    T gm = T(0), gv = T(0);
    const int64_t s0 = ns == 1 ? 0 : sp, s1 = ns == 1 ? S : sp + 1;
    for (int64_t s = s0; s < s1; ++s) {
        const int64_t idx = (s * n + i) * b + c;
        const T g = gh[idx];
        gm += g;
        gv += g * eps[idx] * hs;
    }
    gmean[q] = gm;
    gvar[q] = gv;
}

// ---- generic two-stage sum ---------------------------------------------------------------------

template <typename T>
__global__ __launch_bounds__(256) void reduce_final_kernel(const T* __restrict__ part, int64_t nparts, int64_t nout,
                                                           T scale, T add, T* __restrict__ out,
                                                           const T* __restrict__ add_dev = nullptr) {
    __shared__ T lds[4];
    const int64_t o = blockIdx.x;
    if (o >= nout) return;
    T s = T(0);
    for (int64_t t = threadIdx.x; t < nparts; t += 256) s += part[o * nparts + t];
    s = block_sum_256(s, lds);
    if (threadIdx.x == 0) out[o] = scale * s + add + (add_dev ? add_dev[o] : T(0));
}

// ---- gaussian expected log-likelihood (per sample row s) -------------------------------------------
// part[s * gridDim.x + blk] = (gout ? gout[s * gs] : 1) * sum over a chunk of row s of      (gs = 0: one shared gout)
//     want_gnoise ? 1/2 (e/s2^2 - 1/s2) : -1/2 (e/s2 + log s2 + log 2pi),  e = (y - mu)^2 + v
template <typename T>
__global__ __launch_bounds__(256) void gauss_ell_part_kernel(const T* __restrict__ y, const T* __restrict__ mu,
                                                             const T* __restrict__ v, const T* __restrict__ noise,
                                                             const T* __restrict__ gout, int64_t gs, int64_t n,
                                                             int want_gnoise, T* __restrict__ part) {
    __shared__ T lds[4];
    const int64_t s = blockIdx.y;
    const T s2 = noise[0];
    const T is2 = T(1) / s2, ls2 = t_log(s2);
    const T l2pi = T(1.8378770664093454835606594728112);
    T acc = T(0);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const T d = y[i] - mu[s * n + i];
        const T e = d * d + v[s * n + i];
        acc += want_gnoise ? T(0.5) * (e * is2 * is2 - is2) : T(-0.5) * (e * is2 + ls2 + l2pi);
    }
    acc = block_sum_256(acc, lds);
    if (threadIdx.x == 0) part[s * gridDim.x + blockIdx.x] = gout ? gout[s * gs] * acc : acc;
}

template <typename T>
__global__ void gauss_ell_bwd_kernel(const T* __restrict__ y, const T* __restrict__ mu, const T* __restrict__ noise,
                                     const T* __restrict__ gout, int64_t gs, int64_t S, int64_t n, T scale,
                                     T* __restrict__ gmu, T* __restrict__ gv) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= S * n) return;
    const T is2 = T(1) / noise[0];
    const T coef = gout[(idx / n) * gs] * scale;
    gmu[idx] = coef * (y[idx % n] - mu[idx]) * is2;
    gv[idx] = T(-0.5) * coef * is2;
}

// ---- KL(q(u) || N(0, I)) --------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void kl_part_kernel(const T* __restrict__ m, const T* __restrict__ Lq, int64_t M,
                                                      T* __restrict__ part) {
    __shared__ T lds[4];
    const int64_t b = blockIdx.y;
    const T* L = Lq + b * M * M;
    T acc = T(0);
    // rows blockIdx.x, blockIdx.x + gridDim.x, ...; only the lower triangle is read (no 64-bit div/mod per element)
    for (int64_t i = blockIdx.x; i < M; i += gridDim.x) {
        const T* row = L + i * M;
        for (int64_t j = threadIdx.x; j <= i; j += 256) {
            const T l = row[j];
            acc += l * l;
            if (j == i) acc += m[b * M + i] * m[b * M + i] - T(2) * t_log(l < T(0) ? -l : l);
        }
    }
    acc = block_sum_256(acc, lds);
    if (threadIdx.x == 0) part[b * gridDim.x + blockIdx.x] = acc;
}

template <typename T>
__global__ void kl_bwd_kernel(const T* __restrict__ m, const T* __restrict__ Lq, int64_t batch, int64_t M, T gout,
                              const T* __restrict__ gdev, T* __restrict__ gm, T* __restrict__ gLq) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= batch * M * M) return;
    if (gdev) gout *= gdev[0];                          // upstream gradient read on the device (no host scalar)
    const int64_t e = idx % (M * M), b = idx / (M * M);
    const int64_t i = e / M, j = e % M;
    T g = T(0);
    if (j <= i) {
        const T l = Lq[idx];
        g = gout * (i == j ? l - T(1) / l : l);
        if (i == j) gm[b * M + i] = gout * m[b * M + i];
    }
    gLq[idx] = g;
}

template <typename T>
int colstats_impl(const T* A, const T* C, const T* m, const T* base, int64_t batch, int64_t M, int64_t n, T* mean,
                  T* var, void* stream) {
    if (!A) return -1; if (!C) return -2; if (!m) return -3; if (!base) return -4;
    if (batch < 0) return -5; if (M < 0) return -6; if (n < 0) return -7; if (!mean) return -8; if (!var) return -9;
    if (batch == 0 || n == 0) return 0;
    hipLaunchKernelGGL((colstats_kernel<T>), dim3((unsigned)cdiv64(n, 64), (unsigned)batch), dim3(256), 0,
                       (hipStream_t)stream, A, C, m, base, M, n, mean, var);
    return nsgp_launch_status();
}

template <typename T>
int colstats_bwd_impl(const T* A, const T* C, const T* m, const T* gmean, const T* gvar, int64_t batch, int64_t M,
                      int64_t n, T* Abar, T* C2, T* mbar, void* stream) {
    if (!A) return -1; if (!C) return -2; if (!m) return -3; if (!gmean) return -4; if (!gvar) return -5;
    if (batch < 0) return -6; if (M < 0) return -7; if (n < 0) return -8;
    if (!Abar) return -9; if (!C2) return -10; if (!mbar) return -11;
    if (batch == 0 || M == 0) return 0;
    hipLaunchKernelGGL((colstats_bwd_kernel<T>), dim3((unsigned)M, (unsigned)batch), dim3(256), 0, (hipStream_t)stream,
                       A, C, m, gmean, gvar, M, n, Abar, C2, mbar);
    return nsgp_launch_status();
}

template <typename T>
int sample_fwd_impl(const T* mean, const T* var, const T* eps, int64_t S, int64_t ns, int64_t n, int64_t b, T* h,
                    void* stream) {
    if (!mean) return -1; if (!var) return -2; if (!eps) return -3;
    if (S < 0) return -4; if (ns != 1 && ns != S) return -5; if (n < 0) return -6; if (b < 0) return -7; if (!h) return -8;
    const int64_t tot = S * n * b;
    if (tot == 0) return 0;
    hipLaunchKernelGGL((sample_fwd_kernel<T>), dim3((unsigned)cdiv64(tot, 256)), dim3(256), 0, (hipStream_t)stream,
                       mean, var, eps, S, ns, n, b, h);
    return nsgp_launch_status();
}

template <typename T>
int sample_bwd_impl(const T* var, const T* eps, const T* gh, int64_t S, int64_t ns, int64_t n, int64_t b, T* gmean,
                    T* gvar, void* stream) {
    if (!var) return -1; if (!eps) return -2; if (!gh) return -3;
    if (S < 0) return -4; if (ns != 1 && ns != S) return -5; if (n < 0) return -6; if (b < 0) return -7;
    if (!gmean) return -8; if (!gvar) return -9;
    const int64_t tot = b * ns * n;
    if (tot == 0) return 0;
    hipLaunchKernelGGL((sample_bwd_kernel<T>), dim3((unsigned)cdiv64(tot, 256)), dim3(256), 0, (hipStream_t)stream,
                       var, eps, gh, S, ns, n, b, gmean, gvar);
    return nsgp_launch_status();
}

static inline int64_t gauss_blocks(int64_t n) {
    int64_t nblk = cdiv64(n, 1024);
    if (nblk > 64) nblk = 64;
    if (nblk < 1) nblk = 1;
    return nblk;
}

// `total`: out[0] = scale * sum over ALL samples and points (the caller folds 1/S into scale), one launch less
// downstream than an (S,) vector followed by a mean; the backward then takes ONE upstream gradient gout[0].
template <typename T>
int gauss_fwd_impl(const T* y, const T* mu, const T* v, const T* noise, int64_t S, int64_t n, T scale, T* out,
                   void* ws, size_t wsb, void* stream, bool total = false) {
    if (!y) return -1; if (!mu) return -2; if (!v) return -3; if (!noise) return -4;
    if (S < 0 || S > 65535) return -5; if (n < 0) return -6; if (!out) return -8;
    if (S == 0) return 0;
    const int64_t nblk = gauss_blocks(n);
    if (!ws || wsb < (size_t)(S * nblk) * sizeof(T)) return -9;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL((gauss_ell_part_kernel<T>), dim3((unsigned)nblk, (unsigned)S), dim3(256), 0, st, y, mu, v, noise,
                       (const T*)nullptr, (int64_t)0, n, 0, (T*)ws);
    if (total)
        hipLaunchKernelGGL((reduce_final_kernel<T>), dim3(1), dim3(256), 0, st, (const T*)ws, S * nblk, (int64_t)1, scale,
                           T(0), out);
    else
        hipLaunchKernelGGL((reduce_final_kernel<T>), dim3((unsigned)S), dim3(256), 0, st, (const T*)ws, nblk, S, scale,
                           T(0), out);
    return nsgp_launch_status();
}

template <typename T>
int gauss_bwd_impl(const T* y, const T* mu, const T* v, const T* noise, int64_t S, int64_t n, T scale, const T* gout,
                   T* gmu, T* gv, T* gnoise, void* ws, size_t wsb, void* stream, bool total = false) {
    const int64_t gs = total ? 0 : 1;
    if (!y) return -1; if (!mu) return -2; if (!v) return -3; if (!noise) return -4;
    if (S < 0 || S > 65535) return -5; if (n < 0) return -6; if (!gout) return -8; if (!gmu) return -9;
    if (!gv) return -10;
    hipStream_t st = (hipStream_t)stream;
    const int64_t tot = S * n;
    if (tot > 0)
        hipLaunchKernelGGL((gauss_ell_bwd_kernel<T>), dim3((unsigned)cdiv64(tot, 256)), dim3(256), 0, st, y, mu, noise,
                           gout, gs, S, n, scale, gmu, gv);
    if (gnoise) {
        const int64_t nblk = gauss_blocks(n);
        if (!ws || wsb < (size_t)(S * nblk + 1) * sizeof(T)) return -12;
        if (S > 0)
            hipLaunchKernelGGL((gauss_ell_part_kernel<T>), dim3((unsigned)nblk, (unsigned)S), dim3(256), 0, st, y, mu,
                               v, noise, gout, gs, n, 1, (T*)ws);
        hipLaunchKernelGGL((reduce_final_kernel<T>), dim3(1), dim3(256), 0, st, (const T*)ws, S * nblk, (int64_t)1,
                           scale, T(0), gnoise);
    }
    return nsgp_launch_status();
}

template <typename T>
int kl_fwd_impl(const T* m, const T* Lq, int64_t batch, int64_t M, T* out, void* ws, size_t wsb, void* stream,
                bool total = false, T scale = T(1), const T* addin = nullptr) {
    if (!m) return -1; if (!Lq) return -2; if (batch < 0) return -3; if (M < 0) return -4; if (!out) return -5;
    if (batch == 0) return 0;
    int64_t nblk = cdiv64(M * M, 1024); if (nblk > 256) nblk = 256; if (nblk < 1) nblk = 1;
    if (!ws || wsb < (size_t)(batch * nblk) * sizeof(T)) return -6;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL((kl_part_kernel<T>), dim3((unsigned)nblk, (unsigned)batch), dim3(256), 0, st, m, Lq, M, (T*)ws);
    if (total)                                           // out[0] = scale * sum_b KL_b
        hipLaunchKernelGGL((reduce_final_kernel<T>), dim3(1), dim3(256), 0, st, (const T*)ws, batch * nblk, (int64_t)1,
                           T(0.5) * scale, T(-0.5) * T(M) * T(batch) * scale, out, addin);
    else
        hipLaunchKernelGGL((reduce_final_kernel<T>), dim3((unsigned)batch), dim3(256), 0, st, (const T*)ws, nblk, batch,
                           T(0.5), T(-0.5) * T(M), out);
    return nsgp_launch_status();
}

template <typename T>
int kl_bwd_impl(const T* m, const T* Lq, int64_t batch, int64_t M, T gout, T* gm, T* gLq, void* stream,
                const T* gdev = nullptr) {
    if (!m) return -1; if (!Lq) return -2; if (batch < 0) return -3; if (M < 0) return -4; if (!gm) return -6;
    if (!gLq) return -7;
    const int64_t tot = batch * M * M;
    if (tot == 0) return 0;
    hipLaunchKernelGGL((kl_bwd_kernel<T>), dim3((unsigned)cdiv64(tot, 256)), dim3(256), 0, (hipStream_t)stream, m, Lq,
                       batch, M, gout, gdev, gm, gLq);
    return nsgp_launch_status();
}

// ---- the whole DSVI objective as ONE scalar: two launches forward, two backward -------------------------------------
// out = ell_scale * sum_s sum_i E_q log N(y_i | f_si, noise) + kl_scale * sum_groups sum_b KL(N(m_b, Lq_b Lq_b^T) || N(0, I))
// (the caller folds 1/(B S), 1/num_data and the sign into the two scales).  The chain of per-term reductions this replaces
// cost 6 launches forward and 5 backward for the 2-layer model, ~5 us each under graph replay.
constexpr int OBJ_MAX_GROUPS = 8;
template <typename T> struct ObjGroups {
    const T* m[OBJ_MAX_GROUPS];
    const T* Lq[OBJ_MAX_GROUPS];
    T* gm[OBJ_MAX_GROUPS];
    T* gLq[OBJ_MAX_GROUPS];
    int batch[OBJ_MAX_GROUPS];
    int blk0[OBJ_MAX_GROUPS + 1];        // first block of each group in the KL region of a grid
    int ng;
};

static inline int64_t kl_blocks(int64_t M) {
    int64_t nblk = cdiv64(M * M, 1024); if (nblk > 256) nblk = 256; if (nblk < 1) nblk = 1;
    return nblk;
}

// blocks [0, S nblk_e): likelihood partials (block (s, x) strides over row s); then, per group and batch element, nblk_k
// blocks of KL partials (rows x, x + nblk_k, ... of the lower triangle): the arithmetic of gauss_ell_part / kl_part
template <typename T>
__global__ __launch_bounds__(256) void dsvi_obj_part_kernel(const T* __restrict__ y, const T* __restrict__ mu,
                                                            const T* __restrict__ v, const T* __restrict__ noise, int64_t n,
                                                            int nblk_e, int n_ell, ObjGroups<T> G, int64_t M, int nblk_k,
                                                            T* __restrict__ part) {
    __shared__ T lds[4];
    const int b = (int)blockIdx.x;
    T acc = T(0);
    if (b < n_ell) {
        const int64_t s = b / nblk_e, xb = b % nblk_e;
        const T s2 = noise[0];
        const T is2 = T(1) / s2, ls2 = t_log(s2);
        const T l2pi = T(1.8378770664093454835606594728112);
        for (int64_t i = xb * 256 + threadIdx.x; i < n; i += (int64_t)nblk_e * 256) {
            const T d = y[i] - mu[s * n + i];
            acc += T(-0.5) * ((d * d + v[s * n + i]) * is2 + ls2 + l2pi);
        }
    } else {
        const int kb = b - n_ell;
        int g = 0;
        while (g + 1 < G.ng && kb >= G.blk0[g + 1]) ++g;
        const int rel = kb - G.blk0[g];
        const int64_t bi = rel / nblk_k, xb = rel % nblk_k;
        const T* L = G.Lq[g] + bi * M * M;
        const T* mm = G.m[g] + bi * M;
        for (int64_t i = xb; i < M; i += nblk_k) {
            const T* row = L + i * M;
            for (int64_t j = threadIdx.x; j <= i; j += 256) {
                const T l = row[j];
                acc += l * l;
                if (j == i) acc += mm[i] * mm[i] - T(2) * t_log(l < T(0) ? -l : l);
            }
        }
    }
    acc = block_sum_256(acc, lds);
    if (threadIdx.x == 0) part[b] = acc;
}

// out[0] = ell_scale * sum(part[0 .. n_ell)) + kl_scale * (1/2 sum(part[n_ell .. n_all)) - kl_half_const)
template <typename T>
__global__ __launch_bounds__(256) void dsvi_obj_final_kernel(const T* __restrict__ part, int n_ell, int n_all, T ell_scale,
                                                             T kl_scale, T kl_half_const, T* __restrict__ out) {
    __shared__ T lds[4];
    T a = T(0), k = T(0);
    for (int t = threadIdx.x; t < n_ell; t += 256) a += part[t];
    for (int t = n_ell + threadIdx.x; t < n_all; t += 256) k += part[t];
    a = block_sum_256(a, lds);
    __syncthreads();
    k = block_sum_256(k, lds);
    if (threadIdx.x == 0) out[0] = ell_scale * a + kl_scale * (T(0.5) * k - kl_half_const);
}

// backward: blocks [0, nb_e): gmu / gv; then per group its gLq / gm elements; then (want_gn) S nblk_e blocks of
// noise-gradient partials.  gout[0] is the upstream gradient of the scalar, read on the device.
template <typename T>
__global__ __launch_bounds__(256) void dsvi_obj_bwd_kernel(const T* __restrict__ y, const T* __restrict__ mu,
                                                           const T* __restrict__ v, const T* __restrict__ noise, int64_t S,
                                                           int64_t n, T ell_scale, T kl_scale, const T* __restrict__ gout,
                                                           int nb_e, ObjGroups<T> G, int64_t M, int nb_kl, int nblk_e,
                                                           T* __restrict__ gmu, T* __restrict__ gv, T* __restrict__ part_gn) {
    __shared__ T lds[4];
    const int b = (int)blockIdx.x;
    const T up = gout[0];
    if (b < nb_e) {
        const int64_t idx = (int64_t)b * 256 + threadIdx.x;
        if (idx >= S * n) return;
        const T is2 = T(1) / noise[0];
        const T coef = up * ell_scale;
        gmu[idx] = coef * (y[idx % n] - mu[idx]) * is2;
        gv[idx] = T(-0.5) * coef * is2;
        return;
    }
    if (b < nb_e + nb_kl) {
        const int kb = b - nb_e;
        int g = 0;
        while (g + 1 < G.ng && kb >= G.blk0[g + 1]) ++g;
        const int64_t idx = (int64_t)(kb - G.blk0[g]) * 256 + threadIdx.x;
        if (idx >= (int64_t)G.batch[g] * M * M) return;
        const int64_t e = idx % (M * M), bi = idx / (M * M);
        const int64_t i = e / M, j = e % M;
        const T go = up * kl_scale;
        T gg = T(0);
        if (j <= i) {
            const T l = G.Lq[g][idx];
            gg = go * (i == j ? l - T(1) / l : l);
            if (i == j) G.gm[g][bi * M + i] = go * G.m[g][bi * M + i];
        }
        G.gLq[g][idx] = gg;
        return;
    }
    // noise-gradient partials: d/d noise of -1/2 (e / s2 + log s2) = 1/2 (e / s2^2 - 1 / s2)
    const int rb = b - nb_e - nb_kl;
    const int64_t s = rb / nblk_e, xb = rb % nblk_e;
    const T is2 = T(1) / noise[0];
    T acc = T(0);
    for (int64_t i = xb * 256 + threadIdx.x; i < n; i += (int64_t)nblk_e * 256) {
        const T d = y[i] - mu[s * n + i];
        acc += T(0.5) * ((d * d + v[s * n + i]) * is2 * is2 - is2);
    }
    acc = block_sum_256(acc, lds);
    if (threadIdx.x == 0) part_gn[rb] = acc;
}

template <typename T>
__global__ __launch_bounds__(256) void dsvi_obj_gnoise_kernel(const T* __restrict__ part, int nparts, T ell_scale,
                                                              const T* __restrict__ gout, T* __restrict__ gnoise) {
    __shared__ T lds[4];
    T a = T(0);
    for (int t = threadIdx.x; t < nparts; t += 256) a += part[t];
    a = block_sum_256(a, lds);
    if (threadIdx.x == 0) gnoise[0] = gout[0] * ell_scale * a;
}

template <typename T>
static int obj_groups(ObjGroups<T>& G, int ngroups, const void* const* m, const void* const* Lq, void* const* gm, void* const* gLq,
                      const int64_t* batch, int64_t per_batch_blocks_or_elems, bool elems, int64_t M) {
    G.ng = ngroups;
    int64_t off = 0;
    for (int g = 0; g < ngroups; ++g) {
        if (!m[g] || !Lq[g] || batch[g] < 1) return -1;
        G.m[g] = (const T*)m[g]; G.Lq[g] = (const T*)Lq[g];
        G.gm[g] = gm ? (T*)gm[g] : nullptr; G.gLq[g] = gLq ? (T*)gLq[g] : nullptr;
        if (gm && (!gm[g] || !gLq[g])) return -1;
        G.batch[g] = (int)batch[g];
        G.blk0[g] = (int)off;
        off += elems ? cdiv64(batch[g] * M * M, 256) : batch[g] * per_batch_blocks_or_elems;
        if (off > 2000000000LL) return -1;
    }
    G.blk0[ngroups] = (int)off;
    return 0;
}

template <typename T>
static int dsvi_obj_fwd_impl(const T* y, const T* mu, const T* v, const T* noise, int64_t S, int64_t n, T ell_scale,
                             int ngroups, const void* const* m, const void* const* Lq, const int64_t* batch, int64_t M,
                             T kl_scale, T* out, void* ws, size_t wsb, void* stream) {
    if (!y) return -1; if (!mu) return -2; if (!v) return -3; if (!noise) return -4;
    if (S < 1 || S > 65535) return -5; if (n < 1) return -6;
    if (ngroups < 0 || ngroups > OBJ_MAX_GROUPS) return -8; if (ngroups > 0 && (!m || !Lq || !batch)) return -9;
    if (M < 0) return -12; if (!out) return -14;
    const int64_t nblk_e = gauss_blocks(n), nblk_k = kl_blocks(M);
    ObjGroups<T> G{};
    if (obj_groups<T>(G, ngroups, m, Lq, nullptr, nullptr, batch, nblk_k, false, M)) return -9;
    const int64_t n_ell = S * nblk_e, n_all = n_ell + G.blk0[ngroups];
    if (!ws || wsb < (size_t)n_all * sizeof(T)) return -15;
    int64_t tot_b = 0;
    for (int g = 0; g < ngroups; ++g) tot_b += batch[g];
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL((dsvi_obj_part_kernel<T>), dim3((unsigned)n_all), dim3(256), 0, st, y, mu, v, noise, n, (int)nblk_e,
                       (int)n_ell, G, M, (int)nblk_k, (T*)ws);
    hipLaunchKernelGGL((dsvi_obj_final_kernel<T>), dim3(1), dim3(256), 0, st, (const T*)ws, (int)n_ell, (int)n_all, ell_scale,
                       kl_scale, T(0.5) * T(M) * T(tot_b), out);
    return nsgp_launch_status();
}

template <typename T>
static int dsvi_obj_bwd_impl(const T* y, const T* mu, const T* v, const T* noise, int64_t S, int64_t n, T ell_scale,
                             int ngroups, const void* const* m, const void* const* Lq, const int64_t* batch, int64_t M,
                             T kl_scale, const T* gout, T* gmu, T* gv, T* gnoise, void* const* gm, void* const* gLq, void* ws,
                             size_t wsb, void* stream) {
    if (!y) return -1; if (!mu) return -2; if (!v) return -3; if (!noise) return -4;
    if (S < 1 || S > 65535) return -5; if (n < 1) return -6;
    if (ngroups < 0 || ngroups > OBJ_MAX_GROUPS) return -8; if (ngroups > 0 && (!m || !Lq || !batch)) return -9;
    if (M < 0) return -12; if (!gout) return -14; if (!gmu) return -15; if (!gv) return -16;
    if (ngroups > 0 && (!gm || !gLq)) return -18;
    const int64_t nblk_e = gauss_blocks(n);
    ObjGroups<T> G{};
    if (obj_groups<T>(G, ngroups, m, Lq, gm, gLq, batch, 0, true, M)) return -9;
    const int64_t nb_e = cdiv64(S * n, 256), nb_kl = G.blk0[ngroups], nb_gn = gnoise ? S * nblk_e : 0;
    if (gnoise && (!ws || wsb < (size_t)nb_gn * sizeof(T))) return -20;
    if (nb_e + nb_kl + nb_gn > 2000000000LL) return -6;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL((dsvi_obj_bwd_kernel<T>), dim3((unsigned)(nb_e + nb_kl + nb_gn)), dim3(256), 0, st, y, mu, v, noise, S, n,
                       ell_scale, kl_scale, gout, (int)nb_e, G, M, (int)nb_kl, (int)nblk_e, gmu, gv, (T*)ws);
    if (gnoise)
        hipLaunchKernelGGL((dsvi_obj_gnoise_kernel<T>), dim3(1), dim3(256), 0, st, (const T*)ws, (int)nb_gn, ell_scale, gout,
                           gnoise);
    return nsgp_launch_status();
}

}  // namespace

template <typename T, typename TP = T>
static int finalize_affine_impl(const TP* part_dot, const TP* part_sq_a, const TP* part_sq_c, const T* base, T base_add,
                                int64_t batch, int64_t tiles, int64_t n, const T* x, int64_t sxb, int64_t D, const T* w,
                                int64_t swb, const T* c, int64_t scb, T* mean, T* var, void* stream) {
    if (!part_dot) return -1; if (!part_sq_a) return -2; if (!part_sq_c) return -3; if (!base) return -4;
    if (batch < 0) return -6; if (tiles < 0) return -7; if (n < 0) return -8;
    if (w && !x) return -9; if (sxb < 0) return -10; if (D < 0 || D > NSGP_MAX_DIM || (w && D == 0)) return -11;
    if (swb < 0) return -13; if (scb < 0) return -15; if (!mean) return -16; if (!var) return -17;
    if (batch * n == 0) return 0;
    hipLaunchKernelGGL((colstats_finalize_kernel<T, TP>), dim3((unsigned)cdiv64(batch * n, 256)), dim3(256), 0,
                       (hipStream_t)stream, part_dot, part_sq_a, part_sq_c, base, base_add, batch, tiles, n, x, sxb,
                       (int)D, w, swb, c, scb, mean, var);
    return nsgp_launch_status();
}

template <typename T>
static int rowdot_affine_impl(const T* A, const T* g, const T* gv, const T* x, int64_t sxb, int64_t D, int shared,
                              int64_t batch, int64_t M, int64_t n, T* out, T* out_gv, T* out_x, T* out_1,
                              void* stream) {
    if (!A) return -1; if (!g) return -2; if (sxb < 0) return -5; if (D < 0 || D > NSGP_MAX_DIM) return -6;
    if (batch < 0) return -8; if (M < 0) return -9; if (n < 0) return -10; if (!out) return -11;
    if (gv && !out_gv) return -12; if (out_x && (!x || D == 0)) return -13;
    if (batch == 0) return 0;
    const int64_t rows = M + 2 + (out_x ? D : 0);
    if (rows > 2147483647LL || batch > 65535) return -9;
    hipLaunchKernelGGL((rowdot_affine_kernel<T>), dim3((unsigned)rows, (unsigned)batch), dim3(256), 0,
                       (hipStream_t)stream, A, g, gv, x, sxb, (int)D, shared, batch, M, n, out, out_gv, out_x, out_1);
    return nsgp_launch_status();
}

extern "C" {

size_t nsgp_reduce_workspace(int64_t n_elems, int elem_size) {
    (void)n_elems;
    return (size_t)65536 * elem_size;       // covers RED_BLOCKS partials and 256 x batch(<=256) KL partials
}

int nsgp_svgp_colstats_f32(const float* A, const float* C, const float* m, const float* base, int64_t batch,
                           int64_t M, int64_t n, float* mean, float* var, void* stream) {
    return colstats_impl<float>(A, C, m, base, batch, M, n, mean, var, stream);
}
int nsgp_svgp_colstats_f64(const double* A, const double* C, const double* m, const double* base, int64_t batch,
                           int64_t M, int64_t n, double* mean, double* var, void* stream) {
    return colstats_impl<double>(A, C, m, base, batch, M, n, mean, var, stream);
}
int nsgp_svgp_colstats_bwd_f32(const float* A, const float* C, const float* m, const float* gmean,
                               const float* gvar, int64_t batch, int64_t M, int64_t n, float* Abar, float* C2,
                               float* mbar, void* stream) {
    return colstats_bwd_impl<float>(A, C, m, gmean, gvar, batch, M, n, Abar, C2, mbar, stream);
}
int nsgp_svgp_colstats_bwd_f64(const double* A, const double* C, const double* m, const double* gmean,
                               const double* gvar, int64_t batch, int64_t M, int64_t n, double* Abar, double* C2,
                               double* mbar, void* stream) {
    return colstats_bwd_impl<double>(A, C, m, gmean, gvar, batch, M, n, Abar, C2, mbar, stream);
}
int nsgp_dgp_sample_fwd_f32(const float* mean, const float* var, const float* eps, int64_t S, int64_t ns, int64_t n,
                            int64_t b, float* h, void* stream) {
    return sample_fwd_impl<float>(mean, var, eps, S, ns, n, b, h, stream);
}
int nsgp_dgp_sample_bwd_f32(const float* var, const float* eps, const float* gh, int64_t S, int64_t ns, int64_t n,
                            int64_t b, float* gmean, float* gvar, void* stream) {
    return sample_bwd_impl<float>(var, eps, gh, S, ns, n, b, gmean, gvar, stream);
}
int nsgp_dgp_sample_fwd_f64(const double* mean, const double* var, const double* eps, int64_t S, int64_t ns,
                            int64_t n, int64_t b, double* h, void* stream) {
    return sample_fwd_impl<double>(mean, var, eps, S, ns, n, b, h, stream);
}
int nsgp_dgp_sample_bwd_f64(const double* var, const double* eps, const double* gh, int64_t S, int64_t ns, int64_t n,
                            int64_t b, double* gmean, double* gvar, void* stream) {
    return sample_bwd_impl<double>(var, eps, gh, S, ns, n, b, gmean, gvar, stream);
}
int nsgp_gauss_ell_fwd_f32(const float* y, const float* mu, const float* v, const float* noise, int64_t S, int64_t n,
                           float scale, float* out, void* ws, size_t wsb, void* stream) {
    return gauss_fwd_impl<float>(y, mu, v, noise, S, n, scale, out, ws, wsb, stream);
}
int nsgp_gauss_ell_bwd_f32(const float* y, const float* mu, const float* v, const float* noise, int64_t S, int64_t n,
                           float scale, const float* gout, float* gmu, float* gv, float* gnoise, void* ws,
                           size_t wsb, void* stream) {
    return gauss_bwd_impl<float>(y, mu, v, noise, S, n, scale, gout, gmu, gv, gnoise, ws, wsb, stream);
}
int nsgp_gauss_ell_fwd_f64(const double* y, const double* mu, const double* v, const double* noise, int64_t S,
                           int64_t n, double scale, double* out, void* ws, size_t wsb, void* stream) {
    return gauss_fwd_impl<double>(y, mu, v, noise, S, n, scale, out, ws, wsb, stream);
}
int nsgp_gauss_ell_bwd_f64(const double* y, const double* mu, const double* v, const double* noise, int64_t S,
                           int64_t n, double scale, const double* gout, double* gmu, double* gv, double* gnoise,
                           void* ws, size_t wsb, void* stream) {
    return gauss_bwd_impl<double>(y, mu, v, noise, S, n, scale, gout, gmu, gv, gnoise, ws, wsb, stream);
}
int nsgp_kl_whitened_fwd_f32(const float* m, const float* Lq, int64_t batch, int64_t M, float* out, void* ws,
                             size_t wsb, void* stream) {
    return kl_fwd_impl<float>(m, Lq, batch, M, out, ws, wsb, stream);
}
int nsgp_kl_whitened_bwd_f32(const float* m, const float* Lq, int64_t batch, int64_t M, float gout, float* gm,
                             float* gLq, void* stream) {
    return kl_bwd_impl<float>(m, Lq, batch, M, gout, gm, gLq, stream);
}
int nsgp_kl_whitened_fwd_f64(const double* m, const double* Lq, int64_t batch, int64_t M, double* out, void* ws,
                             size_t wsb, void* stream) {
    return kl_fwd_impl<double>(m, Lq, batch, M, out, ws, wsb, stream);
}
int nsgp_kl_whitened_bwd_f64(const double* m, const double* Lq, int64_t batch, int64_t M, double gout, double* gm,
                             double* gLq, void* stream) {
    return kl_bwd_impl<double>(m, Lq, batch, M, gout, gm, gLq, stream);
}


int nsgp_svgp_colstats_finalize_f32(const float* part_dot, const float* part_sq_a, const float* part_sq_c,
                                    const float* base, int64_t batch, int64_t tiles, int64_t n, float* mean,
                                    float* var, void* stream) {
    if (!part_dot) return -1; if (!part_sq_a) return -2; if (!part_sq_c) return -3; if (!base) return -4;
    if (batch < 0) return -5; if (tiles < 0) return -6; if (n < 0) return -7; if (!mean) return -8; if (!var) return -9;
    if (batch * n == 0) return 0;
    hipLaunchKernelGGL((colstats_finalize_kernel<float>), dim3((unsigned)cdiv64(batch * n, 256)), dim3(256), 0,
                       (hipStream_t)stream, part_dot, part_sq_a, part_sq_c, base, (float)0, batch, tiles, n,
                       (const float*)nullptr, (int64_t)0, 0, (const float*)nullptr, (int64_t)0, (const float*)nullptr,
                       (int64_t)0, mean, var);
    return nsgp_launch_status();
}
int nsgp_svgp_colstats_finalize_f64(const double* part_dot, const double* part_sq_a, const double* part_sq_c,
                                    const double* base, int64_t batch, int64_t tiles, int64_t n, double* mean,
                                    double* var, void* stream) {
    if (!part_dot) return -1; if (!part_sq_a) return -2; if (!part_sq_c) return -3; if (!base) return -4;
    if (batch < 0) return -5; if (tiles < 0) return -6; if (n < 0) return -7; if (!mean) return -8; if (!var) return -9;
    if (batch * n == 0) return 0;
    hipLaunchKernelGGL((colstats_finalize_kernel<double>), dim3((unsigned)cdiv64(batch * n, 256)), dim3(256), 0,
                       (hipStream_t)stream, part_dot, part_sq_a, part_sq_c, base, (double)0, batch, tiles, n,
                       (const double*)nullptr, (int64_t)0, 0, (const double*)nullptr, (int64_t)0, (const double*)nullptr,
                       (int64_t)0, mean, var);
    return nsgp_launch_status();
}
int nsgp_svgp_colstats_finalize_affine_f32(const float* part_dot, const float* part_sq_a, const float* part_sq_c,
                                           const float* base, float base_add, int64_t batch, int64_t tiles, int64_t n,
                                           const float* x, int64_t x_batch_stride, int64_t D, const float* w,
                                           int64_t w_batch_stride, const float* c, int64_t c_batch_stride, float* mean,
                                           float* var, void* stream) {
    return finalize_affine_impl<float>(part_dot, part_sq_a, part_sq_c, base, base_add, batch, tiles, n, x, x_batch_stride,
                                       D, w, w_batch_stride, c, c_batch_stride, mean, var, stream);
}
int nsgp_svgp_colstats_finalize_affine_p64_f32(const double* part_dot, const double* part_sq_a, const double* part_sq_c,
                                               const float* base, float base_add, int64_t batch, int64_t tiles, int64_t n,
                                               const float* x, int64_t x_batch_stride, int64_t D, const float* w,
                                               int64_t w_batch_stride, const float* c, int64_t c_batch_stride, float* mean,
                                               float* var, void* stream) {
    return finalize_affine_impl<float, double>(part_dot, part_sq_a, part_sq_c, base, base_add, batch, tiles, n, x,
                                               x_batch_stride, D, w, w_batch_stride, c, c_batch_stride, mean, var, stream);
}
int nsgp_svgp_colstats_finalize_affine_f64(const double* part_dot, const double* part_sq_a, const double* part_sq_c,
                                           const double* base, double base_add, int64_t batch, int64_t tiles,
                                           int64_t n, const double* x, int64_t x_batch_stride, int64_t D,
                                           const double* w, int64_t w_batch_stride, const double* c,
                                           int64_t c_batch_stride, double* mean, double* var, void* stream) {
    return finalize_affine_impl<double>(part_dot, part_sq_a, part_sq_c, base, base_add, batch, tiles, n, x,
                                        x_batch_stride, D, w, w_batch_stride, c, c_batch_stride, mean, var, stream);
}
int nsgp_rowdot_affine_f32(const float* A, const float* g, const float* gv, const float* x, int64_t x_batch_stride,
                           int64_t D, int shared, int64_t batch, int64_t M, int64_t n, float* out, float* out_gv,
                           float* out_x, float* out_1, void* stream) {
    return rowdot_affine_impl<float>(A, g, gv, x, x_batch_stride, D, shared, batch, M, n, out, out_gv, out_x, out_1,
                                     stream);
}
int nsgp_rowdot_affine_f64(const double* A, const double* g, const double* gv, const double* x, int64_t x_batch_stride,
                           int64_t D, int shared, int64_t batch, int64_t M, int64_t n, double* out, double* out_gv,
                           double* out_x, double* out_1, void* stream) {
    return rowdot_affine_impl<double>(A, g, gv, x, x_batch_stride, D, shared, batch, M, n, out, out_gv, out_x, out_1,
                                      stream);
}
int nsgp_rowdot_f32(const float* A, const float* g, int64_t batch, int64_t M, int64_t n, float* out, void* stream) {
    if (!A) return -1; if (!g) return -2; if (batch < 0) return -3; if (M < 0) return -4; if (n < 0) return -5;
    if (!out) return -6;
    if (batch * M == 0) return 0;
    hipLaunchKernelGGL((rowdot_kernel<float>), dim3((unsigned)M, (unsigned)batch), dim3(256), 0, (hipStream_t)stream,
                       A, g, M, n, out);
    return nsgp_launch_status();
}
int nsgp_rowdot_f64(const double* A, const double* g, int64_t batch, int64_t M, int64_t n, double* out,
                    void* stream) {
    if (!A) return -1; if (!g) return -2; if (batch < 0) return -3; if (M < 0) return -4; if (n < 0) return -5;
    if (!out) return -6;
    if (batch * M == 0) return 0;
    hipLaunchKernelGGL((rowdot_kernel<double>), dim3((unsigned)M, (unsigned)batch), dim3(256), 0, (hipStream_t)stream,
                       A, g, M, n, out);
    return nsgp_launch_status();
}


/* scalar forms used by the fused ELBO tail (nsgp.ops.GaussEllTotalFn / KlWhitenedTotalFn) */
int nsgp_gauss_ell_total_fwd_f32(const float* y, const float* mu, const float* v, const float* noise, int64_t S,
                                 int64_t n, float scale, float* out, void* ws, size_t wsb, void* stream) {
    return gauss_fwd_impl<float>(y, mu, v, noise, S, n, scale, out, ws, wsb, stream, true);
}
int nsgp_gauss_ell_total_fwd_f64(const double* y, const double* mu, const double* v, const double* noise, int64_t S,
                                 int64_t n, double scale, double* out, void* ws, size_t wsb, void* stream) {
    return gauss_fwd_impl<double>(y, mu, v, noise, S, n, scale, out, ws, wsb, stream, true);
}
int nsgp_gauss_ell_total_bwd_f32(const float* y, const float* mu, const float* v, const float* noise, int64_t S,
                                 int64_t n, float scale, const float* gout, float* gmu, float* gv, float* gnoise,
                                 void* ws, size_t wsb, void* stream) {
    return gauss_bwd_impl<float>(y, mu, v, noise, S, n, scale, gout, gmu, gv, gnoise, ws, wsb, stream, true);
}
int nsgp_gauss_ell_total_bwd_f64(const double* y, const double* mu, const double* v, const double* noise, int64_t S,
                                 int64_t n, double scale, const double* gout, double* gmu, double* gv, double* gnoise,
                                 void* ws, size_t wsb, void* stream) {
    return gauss_bwd_impl<double>(y, mu, v, noise, S, n, scale, gout, gmu, gv, gnoise, ws, wsb, stream, true);
}
int nsgp_kl_whitened_total_fwd_f32(const float* m, const float* Lq, int64_t batch, int64_t M, float scale, float* out,
                                   void* ws, size_t wsb, void* stream) {
    return kl_fwd_impl<float>(m, Lq, batch, M, out, ws, wsb, stream, true, scale);
}
int nsgp_kl_whitened_total_fwd_f64(const double* m, const double* Lq, int64_t batch, int64_t M, double scale,
                                   double* out, void* ws, size_t wsb, void* stream) {
    return kl_fwd_impl<double>(m, Lq, batch, M, out, ws, wsb, stream, true, scale);
}
int nsgp_kl_whitened_total_acc_fwd_f32(const float* m, const float* Lq, int64_t batch, int64_t M, float scale,
                                       const float* addin, float* out, void* ws, size_t wsb, void* stream) {
    if (batch == 0) return addin ? -3 : 0;               // nothing would write out = addin
    return kl_fwd_impl<float>(m, Lq, batch, M, out, ws, wsb, stream, true, scale, addin);
}
int nsgp_kl_whitened_total_acc_fwd_f64(const double* m, const double* Lq, int64_t batch, int64_t M, double scale,
                                       const double* addin, double* out, void* ws, size_t wsb, void* stream) {
    if (batch == 0) return addin ? -3 : 0;
    return kl_fwd_impl<double>(m, Lq, batch, M, out, ws, wsb, stream, true, scale, addin);
}
int nsgp_kl_whitened_total_bwd_f32(const float* m, const float* Lq, int64_t batch, int64_t M, float scale,
                                   const float* gout, float* gm, float* gLq, void* stream) {
    if (!gout) return -6;
    return kl_bwd_impl<float>(m, Lq, batch, M, scale, gm, gLq, stream, gout);
}
int nsgp_kl_whitened_total_bwd_f64(const double* m, const double* Lq, int64_t batch, int64_t M, double scale,
                                   const double* gout, double* gm, double* gLq, void* stream) {
    if (!gout) return -6;
    return kl_bwd_impl<double>(m, Lq, batch, M, scale, gm, gLq, stream, gout);
}

size_t nsgp_dsvi_objective_workspace(int64_t S, int64_t n, int64_t M, int64_t total_batch, int elem_size) {
    if (S < 1 || n < 1) return 0;
    return (size_t)(S * gauss_blocks(n) + (M > 0 ? total_batch * kl_blocks(M) : 0)) * (size_t)elem_size;
}
int nsgp_dsvi_objective_fwd_f32(const float* y, const float* mu, const float* v, const float* noise, int64_t S, int64_t n,
                                float ell_scale, int ngroups, const void* const* m, const void* const* Lq,
                                const int64_t* batch, int64_t M, float kl_scale, float* out, void* ws, size_t wsb,
                                void* stream) {
    return dsvi_obj_fwd_impl<float>(y, mu, v, noise, S, n, ell_scale, ngroups, m, Lq, batch, M, kl_scale, out, ws, wsb, stream);
}
int nsgp_dsvi_objective_fwd_f64(const double* y, const double* mu, const double* v, const double* noise, int64_t S, int64_t n,
                                double ell_scale, int ngroups, const void* const* m, const void* const* Lq,
                                const int64_t* batch, int64_t M, double kl_scale, double* out, void* ws, size_t wsb,
                                void* stream) {
    return dsvi_obj_fwd_impl<double>(y, mu, v, noise, S, n, ell_scale, ngroups, m, Lq, batch, M, kl_scale, out, ws, wsb, stream);
}
int nsgp_dsvi_objective_bwd_f32(const float* y, const float* mu, const float* v, const float* noise, int64_t S, int64_t n,
                                float ell_scale, int ngroups, const void* const* m, const void* const* Lq,
                                const int64_t* batch, int64_t M, float kl_scale, const float* gout, float* gmu, float* gv,
                                float* gnoise, void* const* gm, void* const* gLq, void* ws, size_t wsb, void* stream) {
    return dsvi_obj_bwd_impl<float>(y, mu, v, noise, S, n, ell_scale, ngroups, m, Lq, batch, M, kl_scale, gout, gmu, gv, gnoise,
                                    gm, gLq, ws, wsb, stream);
}
int nsgp_dsvi_objective_bwd_f64(const double* y, const double* mu, const double* v, const double* noise, int64_t S, int64_t n,
                                double ell_scale, int ngroups, const void* const* m, const void* const* Lq,
                                const int64_t* batch, int64_t M, double kl_scale, const double* gout, double* gmu, double* gv,
                                double* gnoise, void* const* gm, void* const* gLq, void* ws, size_t wsb, void* stream) {
    return dsvi_obj_bwd_impl<double>(y, mu, v, noise, S, n, ell_scale, ngroups, m, Lq, batch, M, kl_scale, gout, gmu, gv, gnoise,
                                     gm, gLq, ws, wsb, stream);
}

}  // extern "C"
