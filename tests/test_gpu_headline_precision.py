"""the north-star tolerance (posterior mean within 1e-4 relative error) asserted AT the headline
shape -- BASELINE configs[3]: 2-layer DSVI DeepGP, M = 1024 inducing points, minibatch B = 4096, S = 10 likelihood
samples, D = 3 -- after a few Adam steps from bench.py's own initialisation (so kappa(Kzz) is that of a model in
training, not of the random init; 25 and 1000 steps), float32 HIP path against the float64 CPU oracle (models/dgps.py:44-51,92-98
through gpytorch's whitened VariationalStrategy.forward).

Error measure: max-norm relative error  max_i |got_i - ref_i| / max_i |ref_i|  per layer output (means AND
variances) and relative error of the ELBO.  kappa(Kzz + jitter I) is printed for each GP."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
F64 = torch.float64
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')


def _maxrel(got, ref):
    return float((got.double().cpu() - ref).abs().max() / ref.abs().max())


@pytest.mark.parametrize('train_steps', [25, 1000])
def test_posterior_mean_within_1e4_at_the_headline_shape(train_steps):
    """train_steps = 25: a model early in training; 1000: kappa(Kzz) has moved with the inducing points / lengthscales
    (the bound must hold along the training trajectory, not only near the initialisation)."""
    _need_gpu()
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import bench
    from nsgp.gp import settings
    from nsgp.gp.mlls import DeepApproximateMLL, VariationalELBO
    from oracle import kernels, svgp
    from test_gpu_dgp import _FixedEps, _oracle_layers
    M, B, S, N = bench.M_INDUCING, bench.BATCH, bench.S_SAMPLES, bench.N_DATA
    assert (M, B, S) == (1024, 4096, 10)
    dev = torch.device('cuda', torch.cuda.current_device())
    x_all, y_all = bench.synthetic_grid()
    model, mll, opt = bench.build(dev, 1)
    perm = torch.randperm(N, generator=torch.Generator().manual_seed(bench.SEED))
    model.train()
    g = torch.Generator().manual_seed(5)
    with settings.num_likelihood_samples(S):
        for k in range(train_steps):                                 # Adam steps (lr 0.01) on fresh minibatches
            rows = perm[(k % (N // B)) * B:(k % (N // B) + 1) * B]
            opt.zero_grad()
            loss = -mll(model(x_all[rows].to(dev)), y_all[rows].to(dev))
            loss.backward()
            opt.step()
    assert bool(torch.isfinite(loss))
    rows = torch.randperm(N, generator=g)[:B]                   # a fresh minibatch
    xb, yb = x_all[rows], y_all[rows]
    eps = [torch.randn(S, B, 2, generator=g)]
    model.eval()                       # eval: no variational-mean init noise; same marginals as train mode
    with torch.no_grad(), settings.num_likelihood_samples(S), settings.eps_provider(_FixedEps(eps)):
        hid = model.layers[0](xb.to(dev))                           # MultitaskMVN over (n, 2)
        h_mean, h_var = hid.mean, hid.variance
        out = model(xb.to(dev))
        o_mean, o_var = out.mean, out.variance                      # (S, B)
        model.train()
        elbo = mll(model(xb.to(dev)), yb.to(dev))
    with torch.no_grad():
        hidden, last, noise, _ = _oracle_layers(model)
        det = lambda p: {k: (v.detach() if torch.is_tensor(v) else tuple(t.detach() if torch.is_tensor(t) else t for t in v))
                         for k, v in p.items()}
        hidden, last, noise = det(hidden), det(last), noise.detach()
        xd = xb.double()
        xin = xd.unsqueeze(-3).expand(2, B, 3)
        hm_ref, hv_ref = svgp.svgp_marginal(xin, hidden)            # (2, B)
        om_ref, ov_ref = svgp.dgp_forward(xd, hidden, last, 1, [e.double() for e in eps], S)
        elbo_ref = svgp.dsvi_elbo(xd, yb.double(), hidden, last, 1, [e.double() for e in eps], S, noise, N)
        kap = []
        for p in (hidden, last):
            Kzz = kernels.rbf_ard(p['Z'], p['Z'], p['lengthscale'], p['outputscale'])
            Kzz = Kzz + 1e-4 * torch.eye(M, dtype=F64)
            ev = torch.linalg.eigvalsh(Kzz)
            kap += [float(k_) for k_ in (ev[..., -1] / ev[..., 0]).reshape(-1)]
    errs = dict(hidden_mean=_maxrel(h_mean.transpose(-1, -2), hm_ref), hidden_var=_maxrel(h_var.transpose(-1, -2), hv_ref),
                out_mean=_maxrel(o_mean, om_ref), out_var=_maxrel(o_var, ov_ref),
                elbo=abs(float(elbo) - float(elbo_ref)) / abs(float(elbo_ref)))
    print(f'after {train_steps} Adam steps: kappa(Kzz + 1e-4 I) [hidden 0, hidden 1, last]:', ['%.3g' % k_ for k_ in kap])
    print('max-norm relative errors (f32 HIP vs f64 oracle):', {k: '%.3g' % v for k, v in errs.items()})
    assert errs['hidden_mean'] <= 1e-4, errs                        # the north-star bound, per layer
    assert errs['out_mean'] <= 1e-4, errs
    assert errs['hidden_var'] <= 1e-4 and errs['out_var'] <= 1e-4, errs
    assert errs['elbo'] <= 1e-4, errs
