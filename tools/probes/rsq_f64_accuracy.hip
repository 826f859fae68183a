// Accuracy probe for the gfx950 v_rsq_f64 seed and its Newton refinements (decides the step count in common.h t_rsqrt).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/rsq_f64_accuracy.hip -o tools/probes/_bin/rsq_probe && tools/probes/_bin/rsq_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

__global__ void k(const double* x, double* y0, double* y1, double* y1b, double* y2, double* y3, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = x[i];
    double r = __builtin_amdgcn_rsq(v);
    y0[i] = r;
    const double r1 = r * (1.5 - 0.5 * v * r * r);
    y1[i] = r1;
    // residual form: e = 1 - v r^2 (one fma after one product), r' = r + (r/2) e
    const double e = __builtin_fma(-(v * r), r, 1.0);
    y1b[i] = __builtin_fma(0.5 * r, e, r);
    y2[i] = r1 * (1.5 - 0.5 * v * r1 * r1);
    const double pc = __builtin_fma(e, 0.375, 0.5);
    y3[i] = __builtin_fma(r * e, pc, r);          // one cubic step
}

int main() {
    const int n = 1 << 22;
    std::vector<double> h(n);
    std::mt19937_64 g(1);
    std::uniform_real_distribution<double> u(-40.0, 40.0);
    for (auto& v : h) v = std::exp2(u(g));
    double *x, *y[5];
    hipMalloc(&x, n * 8);
    for (auto& p : y) hipMalloc(&p, n * 8);
    hipMemcpy(x, h.data(), n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(x, y[0], y[1], y[2], y[3], y[4], n);
    const char* names[5] = {"v_rsq_f64 seed", "1 Newton (classic)", "1 Newton (residual fma)", "2 Newton (classic)",
                            "1 cubic step"};
    std::vector<double> o(n);
    for (int t = 0; t < 5; ++t) {
        hipMemcpy(o.data(), y[t], n * 8, hipMemcpyDeviceToHost);
        long double worst = 0;
        for (int i = 0; i < n; ++i) {
            const long double ref = 1.0L / sqrtl((long double)h[i]);
            const long double err = fabsl(((long double)o[i] - ref) / ref);
            if (err > worst) worst = err;
        }
        printf("%-26s max rel err %.3Le  (%.2Lf ulp of 2^-53)\n", names[t], worst, worst / 1.1102230246251565e-16L);
    }
    return 0;
}
