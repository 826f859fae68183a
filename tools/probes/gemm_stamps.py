#!/usr/bin/env python3
"""Where a GEMM launch's time goes, per workgroup: runs the SVGP projection shapes on the DIAGNOSTIC build of the library
(csrc/gemm.hip compiled with -DNSGP_GEMM_STAMPS -> tools/probes/_bin/libnsgp_stamps.so; every workgroup records shader-clock
stamps at entry / after its prologue / after its K loop / after its epilogue, the 100 MHz wall clock at entry and exit, its
tile and hardware id) and prints, per launch: the clock the chip held, the share of workgroup time in prologue / K loop /
epilogue, cycles per K-tile, per-CU occupancy over the launch and the length of the tail.

    make -C nonstationary-precip_amd/csrc stamps          # builds tools/probes/_bin/libnsgp_stamps.so
    python tools/probes/gemm_stamps.py
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('NSGP_LIB', os.path.join(ROOT, 'tools', 'probes', '_bin', 'libnsgp_stamps.so'))
sys.path.insert(0, os.path.join(ROOT, 'nonstationary-precip_amd'))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from nsgp import _lib, ops  # noqa: E402

CAP = 1 << 16
lib = _lib.load()
raw = ctypes.CDLL(_lib.LIB_PATH)
raw.nsgp_debug_gemm_stamps.argtypes = [ctypes.c_void_p, ctypes.c_uint64]
buf = torch.zeros(CAP * 16, dtype=torch.int64, device='cuda')
assert raw.nsgp_debug_gemm_stamps(ctypes.c_void_p(buf.data_ptr()), CAP) == 0


def analyse(name, flops):
    torch.cuda.synchronize()
    a = buf.cpu().numpy().view(np.uint64).reshape(CAP, 16)
    a = a[a[:, 3] != 0]
    if len(a) == 0:
        print(name, ': no records')
        return
    t0, t1, t2, t3 = [a[:, i].astype(np.float64) for i in range(4)]
    rt0, rt1 = a[:, 4].astype(np.float64), a[:, 5].astype(np.float64)
    nt = (a[:, 6] & np.uint64(0xFFFF)).astype(np.int64)
    hw = a[:, 7]
    xcc = (hw >> np.uint64(32)).astype(np.int64) & 0xF
    hid = (hw & np.uint64(0xFFFFFFFF)).astype(np.int64)
    cu = ((hid >> 8) & 0xF) | (((hid >> 12) & 0x1) << 4) | (((hid >> 13) & 0x7) << 5) | (xcc << 8)
    has_loop = t1 > 0
    wall = (rt1 - rt0) * 10e-9                                   # seconds per workgroup
    ok = wall > 2e-6
    ghz = np.median((t3 - t0)[ok] / wall[ok]) / 1e9 if ok.any() else float('nan')
    span_us = (rt1.max() - rt0.min()) * 1e-2
    pro = np.where(has_loop, t1 - t0, 0.0)
    loop = np.where(has_loop, t2 - t1, 0.0)
    epi = np.where(has_loop, t3 - t2, t3 - t0)
    tot = t3 - t0
    print(f'{name}: {len(a)} workgroups, launch span {span_us:7.1f} us = {flops / span_us / 1e6:6.1f} TFLOP/s (algorithmic), '
          f'shader clock {ghz:4.2f} GHz')
    print(f'    workgroup time: prologue {pro.sum() / tot.sum():5.1%}  K loop {loop.sum() / tot.sum():5.1%}  epilogue '
          f'{epi.sum() / tot.sum():5.1%};  mean per workgroup {tot.mean() / ghz / 1e3:6.1f} us (prologue {pro.mean() / ghz / 1e3:5.2f}, '
          f'epilogue {epi.mean() / ghz / 1e3:5.2f} us)')
    by = {}
    for k in np.unique(nt):
        sel = (nt == k) & has_loop
        if sel.any() and k > 0:
            by[int(k)] = (int(sel.sum()), float(np.median(loop[sel] / k)))
    bar = a[:, 8].astype(np.float64)
    print(f'    wave 0 waited at the K loop barriers for {bar.sum() / max(loop.sum(), 1):5.1%} of the loop time '
          f'({np.median(bar[has_loop] / np.maximum(nt[has_loop], 1)):.0f} cycles per K-tile, median)')
    print('    cycles per K-tile in the loop, by K-tiles per workgroup (n workgroups): ' +
          ', '.join(f'{k}: {v[1]:.0f} ({v[0]})' for k, v in sorted(by.items())))
    # per-CU occupancy: union of the workgroups' wall intervals on each CU / launch span; tail = launch end - the CU's last end
    start, end = rt0.min(), rt1.max()
    occ, tails, nconc = [], [], []
    for c in np.unique(cu):
        sel = cu == c
        iv = sorted(zip(rt0[sel], rt1[sel]))
        busy, cur_s, cur_e = 0.0, iv[0][0], iv[0][1]
        for s_, e_ in iv[1:]:
            if s_ > cur_e:
                busy += cur_e - cur_s
                cur_s, cur_e = s_, e_
            else:
                cur_e = max(cur_e, e_)
        busy += cur_e - cur_s
        occ.append(busy / (end - start))
        tails.append((end - max(e_ for _, e_ in iv)) * 1e-2)
        ev = sorted([(s_, 1) for s_, _ in iv] + [(e_, -1) for _, e_ in iv])          # time with >= 2 workgroups resident
        two, lvl, last = 0.0, 0, start
        for t_, d in ev:
            if lvl >= 2:
                two += t_ - last
            lvl += d
            last = t_
        nconc.append(two / (end - start))
    occ, tails, nconc = np.array(occ), np.array(tails), np.array(nconc)
    print(f'    {len(occ)} CUs seen; a CU holds >= 1 workgroup {occ.mean():5.1%} of the launch (min {occ.min():5.1%}), >= 2 workgroups '
          f'{nconc.mean():5.1%}; idle tail per CU: mean {tails.mean():5.1f} us, max {tails.max():5.1f} us '
          f'({tails.mean() / span_us:5.1%} of the launch)')
    first = (rt0 - start) * 1e-2
    print(f'    workgroup starts within [{first.min():.1f}, {first.max():.1f}] us of the launch; first {2 * len(occ)} within '
          f'{np.sort(first)[min(len(first), 2 * len(occ)) - 1]:.1f} us')


def run(name, fn, flops):
    only = os.environ.get('ONLY')
    if only and not any(o in name for o in only.split(',')):
        return
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    buf.zero_()
    torch.cuda.synchronize()
    fn()
    analyse(name, flops)


def main():
    M = int(os.environ.get('MROWS', 1024))
    g = torch.Generator().manual_seed(0)
    st = ops._stream
    p = ops._p
    for n, b in ((40960, 1), (4096, 2)):
        W = torch.tril(torch.randn(b, M, M, generator=g)).cuda()
        Lq = torch.tril(torch.randn(b, M, M, generator=g)).cuda()
        K = torch.randn(b, M, n, generator=g).cuda()
        A = torch.randn(b, M, n, generator=g).cuda()
        Cc = torch.randn(b, M, n, generator=g).cuda()
        m = torch.randn(b, M, generator=g).cuda()
        gm, gv = torch.randn(b, n, generator=g).cuda(), torch.randn(b, n, generator=g).cuda()
        W64 = W.double()
        tri = 1.0 * M * M * n * b
        print(f'==== M = {M}, n = {n}, batch = {b}')
        part = torch.empty((3, b, 16, n), device='cuda')
        Y = torch.empty_like(K)
        run('C = Lq^T A  <1,1,EPI1> single pass', lambda: _lib.call(
            'nsgp_svgp_tri_gemm_colstats_rows_f32', p(Lq), 1, p(A), None, b, M, n, p(Y), None, p(part[2]), 16, st()), tri)
        run('A = W K     <0,1,EPI1> single pass', lambda: _lib.call(
            'nsgp_svgp_tri_gemm_colstats_rows_f32', p(W), 0, p(K), p(m), b, M, n, p(Y), p(part[0]), p(part[1]), 16, st()), tri)
        run('Abar        <0,1,EPI2> single pass', lambda: _lib.call(
            'nsgp_svgp_abar_f32', p(Lq), p(Cc), p(A), p(m), p(gm), p(gv), b, M, n, p(Y), st()), tri)
        run('Kzxbar = W^T Abar <1,1,0> single pass', lambda: ops.gemm(W, A, ta=True, flags=ops.GEMM_A_UPPER), tri)
        run('Wbar = tril(Abar K^T) split-K', lambda: ops.gemm(A, K, tb=True, flags=ops.GEMM_C_LOWER), tri)
        run('A = W K f64-accumulating <double,128,64,16,MIX>', lambda: _lib.call(
            'nsgp_svgp_tri_gemm_colstats_f64acc', p(W64), p(K), p(m), b, M, n, p(Y), p(part[0]), p(part[1]), 16, st()), tri)
        run('dense W K (no triangle)', lambda: ops.gemm(W, K), 2 * tri)


if __name__ == '__main__':
    main()
