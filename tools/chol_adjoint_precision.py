import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/nonstationary-precip_amd')
import torch
import models.dgps as m
from nsgp.gp import settings
from nsgp.gp.mlls import DeepApproximateMLL, VariationalELBO
torch.manual_seed(0)
N, D, M, S, B = 100000, 3, 1024, 10, 4096
model = m.DeepGP(1, (N, D), num_inducing=M).cuda()
mll = DeepApproximateMLL(VariationalELBO(model.likelihood, model, N))
g = torch.Generator().manual_seed(1)
x = torch.randn(B, D, generator=g).cuda(); y = torch.randn(B, generator=g).cuda()
eps = torch.randn(S, B, 2, generator=g).cuda()
class E:
    def __call__(self, shape, dtype, device): return eps
model.train()
# a few Adam steps so that parameters are not at the trivial init
opt = torch.optim.Adam(model.parameters(), lr=0.01)
for _ in range(5):
    with settings.num_likelihood_samples(S), settings.eps_provider(E()):
        opt.zero_grad(); loss = -mll(model(x), y); loss.backward(); opt.step()
res = {}
for flag in (True, False):
    with settings.num_likelihood_samples(S), settings.eps_provider(E()), settings.chol_bwd_f64(flag):
        model.zero_grad(); loss = -mll(model(x), y); loss.backward()
    res[flag] = {n: p.grad.detach().double().clone() for n, p in model.named_parameters()}
for n in res[True]:
    a, b = res[True][n], res[False][n]
    print(f'{n:70s} |g| {float(a.abs().max()):.3e}  max|diff|/max|g| {float((a-b).abs().max())/ (float(a.abs().max())+1e-30):.3e}')
