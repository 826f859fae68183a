import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'nonstationary-precip_amd'))
import torch
from nsgp import ops
for (M, n, batch, D) in [(256, 512, 2, 3), (128, 128, 1, 2), (128, 128, 1, 1)]:
    g = torch.Generator().manual_seed(1)
    Z = torch.randn(batch, M, D, generator=g).cuda(); x = (1.3 * torch.randn(n, D, generator=g)).cuda()
    ls = (0.6 + torch.rand(batch, D, generator=g)).cuda(); os_ = (0.5 + torch.rand(batch, generator=g)).cuda()
    W64 = torch.eye(M, dtype=torch.float64).repeat(batch, 1, 1).cuda()
    Lq = torch.eye(M).repeat(batch, 1, 1).cuda(); m = torch.zeros(batch, M).cuda()
    Kzx = ops.rbf_build(Z, x, ls, os_)
    A = ops.svgp_project(W64.float(), None, Lq, m, os_, W64f=W64, kernel_inputs=(Z, x, ls, os_))[0]
    A2 = ops.svgp_project(W64.float(), Kzx, Lq, m, os_, W64f=W64)[0]
    d = (A - Kzx).abs(); bad = (A != Kzx)
    print(M, n, batch, D, 'mismatch', int(bad.sum()), 'of', A.numel(), 'max rel', float((d / Kzx.abs().clamp_min(1e-30)).max()), 'materialised==Kzx', bool(torch.equal(A2, Kzx)))
    idx = bad.nonzero()[:5]
    for i in idx.tolist():
        print('   ', i, float(A[tuple(i)]), float(Kzx[tuple(i)]))
    # os scaling / exp argument check on CPU in float32
    Zc, xc, lc = Z.cpu(), x.cpu(), ls.cpu()
    ex = (((Zc[:, :, None, :] / lc[:, None, None, :]) - (xc[None, None, :, :] / lc[:, None, None, :])) ** 2).sum(-1)
    print('   cpu-ish check max rel vs Kzx', float(((os_.cpu()[:, None, None] * torch.exp(-0.5 * ex)) / Kzx.cpu() - 1).abs().max()))
