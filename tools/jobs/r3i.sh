#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3i
mkdir -p $O
cd $R
true
grep "measured\|passed\|failed\|kappa-free" $O/tests.log
python - <<'PY' > $O/i8_timing.log 2>&1
import sys, torch
sys.path.insert(0, 'nonstationary-precip_amd')
from nsgp import ops, _lib
lib = _lib.load()
for (b, M, n, D) in ((1, 1024, 40960, 2), (2, 1024, 4096, 3), (3, 2048, 40960, 3)):
    g = torch.Generator().manual_seed(0)
    Z = torch.randn(b, M, D, generator=g).cuda(); x = torch.randn(n, D, generator=g).cuda()
    ls = (torch.rand(b, D, generator=g) + 0.7).cuda(); os_ = (torch.rand(b, generator=g) + 0.5).cuda()
    W64 = torch.tril(torch.randn(b, M, M, generator=g, dtype=torch.float64)).cuda()
    m = torch.randn(b, M, generator=g).cuda()
    st = ops._stream; p = ops._p
    Wd = torch.empty(int(lib.nsgp_i8_w_planes_bytes(b, M)), dtype=torch.uint8, device='cuda')
    Kd = torch.empty(int(lib.nsgp_i8_k_planes_bytes(b, M, n, 4)), dtype=torch.uint8, device='cuda')
    wsc = torch.empty((b, M), dtype=torch.float64, device='cuda'); ksc = torch.empty(b, dtype=torch.float64, device='cuda')
    A = torch.empty((b, M, n), device='cuda'); part = torch.empty((2, b, 16, n), device='cuda')
    Kzx = ops.rbf_build(Z, x, ls, os_)
    def t(fn, reps=10):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    t_sl = t(lambda: _lib.call('nsgp_i8_slice_w_f64', p(W64), b, M, p(Wd), p(wsc), st()))
    t_bd = t(lambda: _lib.call('nsgp_i8_rbf_build_f32', p(Z), p(x), 0, p(ls), p(os_), b, M, n, D, 4, p(Kd), p(ksc), st()))
    t_mm = t(lambda: _lib.call('nsgp_svgp_tri_gemm_colstats_i8', p(Wd), p(wsc), p(Kd), p(ksc), 4, p(m), b, M, n, p(A), p(part[0]), p(part[1]), 16, 0, st()))
    t_f64 = t(lambda: _lib.call('nsgp_svgp_tri_gemm_colstats_f64acc', p(W64), p(Kzx), p(m), b, M, n, p(A), p(part[0]), p(part[1]), 16, st()))
    t_b32 = t(lambda: ops.rbf_build(Z, x, ls, os_))
    fl = 1.0 * M * M * n * b
    print(f'b={b} M={M} n={n} D={D}: slice W {t_sl:7.1f} us | build Kd {t_bd:7.1f} us (f32 Kzx build {t_b32:6.1f}) | i8 product {t_mm:7.1f} us = '
          f'{fl / t_mm / 1e6:6.1f} TFLOP/s f64-equivalent, {14 * fl / t_mm / 1e6 / 1e3:5.2f} POP/s int8 | f64acc product {t_f64:7.1f} us', flush=True)
PY
cat $O/i8_timing.log | grep -v amdgpu
