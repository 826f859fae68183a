#!/usr/bin/env python3
"""Where the posterior-mean error of the float32 DSVI path comes from once the model has trained.

Headline model (2-layer DGP, M = 1024, B = 4096, S = 10) after TRAIN_STEPS Adam steps; max-norm relative errors of the layer
outputs against the float64 oracle for
  * this path (float32, float64 Kzz chain, float64-accumulated A = W Kzx),
  * the REFERENCE ARITHMETIC itself: the oracle run in float32 (float32 Kzx, float64 Cholesky + solve, cast back --
    what gpytorch executes), i.e. the error floor of "the reference CPU path",
  * this path with the hidden layer's marginals replaced by exact (float64 oracle) ones: the last layer's own error,
  * this path with the hidden layer evaluated in float64 on the GPU.

    python tools/probes/precision_after_training.py [TRAIN_STEPS ...]      (default 25 1000)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'nonstationary-precip_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch  # noqa: E402
import bench  # noqa: E402
from nsgp import ops  # noqa: E402
from nsgp.gp import settings  # noqa: E402
from nsgp.svgp import svgp_marginal  # noqa: E402
from oracle import svgp as OS  # noqa: E402
from test_gpu_dgp import _FixedEps, _oracle_layers  # noqa: E402

F64 = torch.float64


def maxrel(got, ref):
    return float((got.double().cpu() - ref.double()).abs().max() / ref.double().abs().max())


def to_dtype(p, dt):
    out = {}
    for k, v in p.items():
        if torch.is_tensor(v):
            out[k] = v.detach().to(dt)
        elif isinstance(v, tuple):
            out[k] = tuple(t.detach().to(dt) if torch.is_tensor(t) else t for t in v)
        else:
            out[k] = v
    return out


def main():
    steps_list = [int(a) for a in sys.argv[1:]] or [25, 1000]
    M, B, S, N = bench.M_INDUCING, bench.BATCH, bench.S_SAMPLES, bench.N_DATA
    dev = torch.device('cuda', 0)
    x_all, y_all = bench.synthetic_grid()
    model, mll, opt = bench.build(dev, 1)
    perm = torch.randperm(N, generator=torch.Generator().manual_seed(bench.SEED))
    done = 0
    for target in steps_list:
        model.train()
        with settings.num_likelihood_samples(S):
            for k in range(done, target):
                rows = perm[(k % (N // B)) * B:(k % (N // B) + 1) * B]
                opt.zero_grad()
                loss = -mll(model(x_all[rows].to(dev)), y_all[rows].to(dev))
                loss.backward()
                opt.step()
        done = target
        g = torch.Generator().manual_seed(5)
        rows = torch.randperm(N, generator=g)[:B]
        xb = x_all[rows]
        eps = [torch.randn(S, B, 2, generator=g)]
        model.eval()
        with torch.no_grad(), settings.num_likelihood_samples(S), settings.eps_provider(_FixedEps(eps)):
            hid = model.layers[0](xb.to(dev))
            h_mean, h_var = hid.mean.transpose(-1, -2), hid.variance.transpose(-1, -2)          # (2, B)
            out = model(xb.to(dev))
            o_mean = out.mean
        with torch.no_grad():
            hidden, last, noise, _ = _oracle_layers(model)
            hidden, last = to_dtype(hidden, F64), to_dtype(last, F64)
            xd = xb.double()
            xin = xd.unsqueeze(-3).expand(2, B, 3)
            hm64, hv64 = OS.svgp_marginal(xin, hidden)
            om64, ov64 = OS.dgp_forward(xd, hidden, last, 1, [e.double() for e in eps], S)
            # the reference arithmetic: float32 everywhere except the Cholesky / solve
            h32, l32 = to_dtype(hidden, torch.float32), to_dtype(last, torch.float32)
            hm32, hv32 = OS.svgp_marginal(xin.float(), h32)
            om32, ov32 = OS.dgp_forward(xb.float(), h32, l32, 1, eps, S)
            # last layer of THIS path fed with the exact hidden sample
            hs = (hm64.transpose(-1, -2) + hv64.transpose(-1, -2).sqrt() * eps[0].double()).float()     # (S, B, 2)
            lp = model.last_layer.variational_strategy
            Z, ls, os_, m, Lq = lp._flat_params()
            mm = model.last_layer.mean_module
            mean_l, var_l, _ = svgp_marginal(hs.reshape(S * B, 2).to(dev), Z, ls.contiguous(), os_.contiguous(), m, Lq,
                                             mean_c=mm.constant.reshape(-1))
            # hidden layer of this path in float64 on the GPU
            hp = model.layers[0].variational_strategy
            Zh, lsh, osh, mh, Lqh = hp._flat_params()
            hmod = model.layers[0].mean_module
            d = lambda t: t.double()  # noqa: E731
            mean_h64, var_h64, _ = svgp_marginal(xb.double().to(dev), d(Zh), d(lsh).contiguous(), d(osh).contiguous(), d(mh), d(Lqh),
                                                 mean_w=d(hmod.weights.reshape(-1, 3)), mean_c=d(hmod.bias.reshape(-1)))
            hs2 = (mean_h64.transpose(-1, -2) + var_h64.transpose(-1, -2).sqrt() * eps[0].double().to(dev)).float()
            mean_l2, _, _ = svgp_marginal(hs2.reshape(S * B, 2), Z, ls.contiguous(), os_.contiguous(), m, Lq,
                                          mean_c=mm.constant.reshape(-1))
        print(f'==== after {target} Adam steps; |m| max hidden {float(mh.abs().max()):.3g} last {float(m.abs().max()):.3g}; '
              f'max|h mean| {float(hm64.abs().max()):.3g} max|out mean| {float(om64.abs().max()):.3g}')
        print(f'  this path            : hidden mean {maxrel(h_mean, hm64):.3g}  hidden var {maxrel(h_var, hv64):.3g}  '
              f'out mean {maxrel(o_mean, om64):.3g}')
        print(f'  reference arithmetic : hidden mean {maxrel(hm32, hm64):.3g}  hidden var {maxrel(hv32, hv64):.3g}  '
              f'out mean {maxrel(om32, om64):.3g}   (oracle in float32 vs oracle in float64)')
        print(f'  this path vs reference arithmetic: hidden mean {maxrel(h_mean, hm32):.3g}  out mean {maxrel(o_mean, om32):.3g}')
        print(f'  last layer of this path on the EXACT hidden sample: out mean {maxrel(mean_l.reshape(S, B), om64):.3g}')
        print(f'  hidden layer in float64 on the GPU (mean err {maxrel(mean_h64, hm64):.3g}), last layer float32: out mean '
              f'{maxrel(mean_l2.reshape(S, B), om64):.3g}', flush=True)


if __name__ == '__main__':
    main()
