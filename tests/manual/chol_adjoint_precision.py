#!/usr/bin/env python3
"""How much does the float32 Cholesky adjoint (settings.chol_bwd_f64(False)) cost in gradient accuracy at the headline
inducing size?  M = 1024 inducing points, 2-layer DeepGP; the float64 CPU oracle (torch autograd of oracle.svgp) is the
reference, the GPU model runs with the float64 and with the float32 adjoint on the same parameters and noise.
Prints the max-norm relative error of every parameter gradient for both variants.

    python tests/manual/chol_adjoint_precision.py [B] [S]        (defaults 512, 2: keeps the CPU oracle at a few seconds)
"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'nonstationary-precip_amd'))
import torch  # noqa: E402
import models.dgps as m  # noqa: E402
from nsgp.gp import settings  # noqa: E402
from nsgp.gp.mlls import DeepApproximateMLL, VariationalELBO  # noqa: E402
from oracle import svgp  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
S = int(sys.argv[2]) if len(sys.argv) > 2 else 2
N, D, M = 100000, 3, 1024
torch.manual_seed(0)
model = m.DeepGP(1, (N, D), num_inducing=M).cuda()
mll = DeepApproximateMLL(VariationalELBO(model.likelihood, model, N))
g = torch.Generator().manual_seed(1)
x = torch.randn(B, D, generator=g).cuda()
y = torch.randn(B, generator=g).cuda()
eps = torch.randn(S, B, 2, generator=g).cuda()
prov = lambda shape, dtype, device: eps                       # noqa: E731
model.train()
opt = torch.optim.Adam(model.parameters(), lr=0.01)
for _ in range(20):                                            # leave the trivial initialisation
    with settings.num_likelihood_samples(S), settings.eps_provider(prov):
        opt.zero_grad()
        (-mll(model(x), y)).backward()
        opt.step()

sp = torch.nn.functional.softplus
leaves = {}


def leaf(name, t):
    leaves[name] = t.detach().cpu().double().clone().requires_grad_()
    return leaves[name]


def layer(prefix, mod, linear):
    vs = mod.variational_strategy
    p = dict(Z=leaf(prefix + 'Z', vs.inducing_points),
             lengthscale=sp(leaf(prefix + 'raw_ls', mod.covar_module.base_kernel.raw_lengthscale)),
             outputscale=sp(leaf(prefix + 'raw_os', mod.covar_module.raw_outputscale)),
             m=leaf(prefix + 'm', vs._variational_distribution.variational_mean),
             Lq=leaf(prefix + 'Lq', vs._variational_distribution.chol_variational_covar))
    p['mean'] = ('linear', leaf(prefix + 'w', mod.mean_module.weights), leaf(prefix + 'b', mod.mean_module.bias)) \
        if linear else ('constant', leaf(prefix + 'c', mod.mean_module.constant))
    return p


hidden, last = layer('h.', model.layers[0], True), layer('l.', model.last_layer, False)
noise = sp(leaf('raw_noise', model.likelihood.noise_covar.raw_noise)) + 1e-4
ref = svgp.dsvi_elbo(x.cpu().double(), y.cpu().double(), hidden, last, 1, [eps.cpu().double()], S, noise, N)
(-ref).backward()
params = {'h.Z': model.layers[0].variational_strategy.inducing_points,
          'h.raw_ls': model.layers[0].covar_module.base_kernel.raw_lengthscale,
          'h.raw_os': model.layers[0].covar_module.raw_outputscale,
          'l.Z': model.last_layer.variational_strategy.inducing_points,
          'l.raw_ls': model.last_layer.covar_module.base_kernel.raw_lengthscale,
          'l.raw_os': model.last_layer.covar_module.raw_outputscale,
          'h.m': model.layers[0].variational_strategy._variational_distribution.variational_mean,
          'l.Lq': model.last_layer.variational_strategy._variational_distribution.chol_variational_covar}
print(f'M={M} B={B} S={S}; ELBO oracle {float(ref.detach()):.6f}')
print(f'{"parameter":10s} {"|grad|max":>10s}   rel.err f64 adjoint   rel.err f32 adjoint')
res = {}
for flag in (True, False):
    with settings.num_likelihood_samples(S), settings.eps_provider(prov), settings.chol_bwd_f64(flag):
        model.zero_grad()
        loss = -mll(model(x), y)
        loss.backward()
    res[flag] = {k: p.grad.detach().cpu().double().clone() for k, p in params.items()}
for k in params:
    want = leaves[k].grad
    if k.endswith('Lq'):
        want = torch.tril(want)
    sc = float(want.abs().max()) + 1e-30
    e64 = float((res[True][k] - want).abs().max()) / sc
    e32 = float((res[False][k] - want).abs().max()) / sc
    print(f'{k:10s} {sc:10.3e}   {e64:18.3e}   {e32:18.3e}')
