import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'nonstationary-precip_amd'))
import torch
from nsgp import ops
for (M, N, K, ta, tb) in [(300, 1100, 260, True, False), (300, 1100, 260, False, False), (256, 1024, 256, True, False), (256, 1024, 260, True, False), (300,1100,256,True,False)]:
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn((K, M) if ta else (M, K), generator=g)
    B = torch.randn((N, K) if tb else (K, N), generator=g)
    ref = ((A.double().T if ta else A.double()) @ (B.double().T if tb else B.double()))
    for rep in range(3):
        got = ops.gemm(A.cuda(), B.cuda(), ta, tb).cpu().double()
        bad = (got - ref).abs() > 1e-3 * (1 + ref.abs())
        idx = bad.nonzero()
        print((M, N, K, ta, tb), 'rep', rep, 'bad', int(bad.sum()), 'rows', sorted(set((idx[:, 0] // 64).tolist()))[:10], 'cols', sorted(set((idx[:, 1] // 64).tolist()))[:20],
              'maxerr', float((got - ref).abs().max()))
        if len(idx):
            r, c = idx[0].tolist()
            print('   first bad', r, c, float(got[r, c]), float(ref[r, c]), 'rowset', sorted(set((idx[:, 0] % 64).tolist()))[:16], 'colset', sorted(set((idx[:, 1] % 64).tolist()))[:16])
