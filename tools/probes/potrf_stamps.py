#!/usr/bin/env python3
"""Where a panel launch of the blocked Cholesky spends its time: runs nsgp_potrf on the DIAGNOSTIC build of the library
(csrc/potrf.hip compiled with -DNSGP_POTRF_STAMPS; workgroup 0 of matrix 0 records the shader clock at the phase boundaries
of panel_body2) and prints the median cycles per phase over the panels of one factorisation.

    make -C nonstationary-precip_amd/csrc stamps
    python tools/probes/potrf_stamps.py [N] [float32|float64] [batch]      (batch > 1 also prints a per-panel table)
    NSGP_LIB=tools/probes/_bin/libnsgp_stamps_blk0.so python tools/probes/potrf_stamps.py      # one pivot at a time
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

PHASES = ['load+stage', 'U0 (prev rank-64 on cols 0..15)', 'factor 0', 'barrier', 'trail 0', 'factor 1', 'barrier', 'trail 1',
          'factor 2', 'barrier', 'trail 2', 'factor 3', 'barrier', 'after last sub-panel', 'last subst + store']


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    dt = getattr(torch, sys.argv[2]) if len(sys.argv) > 2 else torch.float64
    batch = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    _lib.load()
    raw = ctypes.CDLL(_lib.LIB_PATH)
    raw.nsgp_debug_potrf_stamps.argtypes = [ctypes.c_void_p, ctypes.c_uint64]
    cap = 4096
    buf = torch.zeros(cap * 64, dtype=torch.int64, device='cuda')
    assert raw.nsgp_debug_potrf_stamps(ctypes.c_void_p(buf.data_ptr()), cap) == 0
    g = torch.Generator().manual_seed(0)
    X = torch.randn(n, n + 8, generator=g, dtype=torch.float64)
    K = (X @ X.T / n + torch.eye(n, dtype=torch.float64)).to(dt).cuda()
    if batch > 1:
        K = K.unsqueeze(0).repeat(batch, 1, 1).contiguous()
    for _ in range(3):
        L, info = ops.potrf(K.clone())
    torch.cuda.synchronize()
    buf.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    A = K.clone()
    e0.record()
    L, info = ops.potrf(A)
    e1.record()
    torch.cuda.synchronize()
    assert int(info.max()) == 0
    full = buf.cpu().numpy().view(np.uint64).reshape(cap, 64)
    full = full[full[:, 15] != 0].astype(np.float64)
    if batch > 1:
        print('per panel (workgroup 0 of matrix 0): whole-workgroup cycles, load+stage cycles')
        for i, r in enumerate(full):
            print(f'    panel {i:3d}: {r[15] - r[0]:8.0f} {r[1] - r[0]:8.0f}')
    full = full[1:-1] if len(full) > 4 else full
    a = full[:, :16]
    d = np.diff(a, axis=1)
    print(f'{os.path.basename(_lib.LIB_PATH)}: potrf n = {n} {dt}: {e0.elapsed_time(e1) * 1e3:.1f} us for {len(a) + 2} panel launches; '
          f'cycles per phase of workgroup 0, wave 0 (median over panels):')
    tot = np.median(a[:, 15] - a[:, 0])
    for i, name in enumerate(PHASES):
        print(f'    {name:36s} {np.median(d[:, i]):8.0f}  ({np.median(d[:, i]) / tot:5.1%})')
    print(f'    {"whole workgroup":36s} {tot:8.0f}')
    # arrival of each wave at the barrier ending a phase, relative to the phase's start (the previous barrier's release ~ wave 0's stamp)
    names = ['F0', 'trail 0', 'F1', 'trail 1', 'F2', 'trail 2', 'F3', 'after-last']
    starts = [2, 4, 5, 7, 8, 10, 11, 13]                   # wave 0 stamp index at which the phase starts
    print('    cycles from phase start to each wave reaching the phase-ending barrier (median):   wave0   wave1   wave2   wave3')
    for i, nm in enumerate(names):
        w = [np.median(full[:, 16 + 8 * k + i] - full[:, starts[i]]) for k in range(4)]
        print(f'        {nm:12s} ' + ' '.join(f'{v:8.0f}' for v in w))


if __name__ == '__main__':
    main()
