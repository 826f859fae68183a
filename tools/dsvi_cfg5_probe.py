#!/usr/bin/env python3
"""DSVI step time of BASELINE configs[4]'s shape on ONE MI355X (float32, or the config's "bf16 forward" with --forward):
3-layer DeepGP = DeepGP(num_layers=2) (tied hidden layer 3->3 applied twice + last 3->1, so num_output_dims = 3),
M=2048 inducing points, synthetic N=1e6 rows on a 100^3 grid, minibatch 4096, S=10; forward + ELBO + backward + Adam
captured as one hipGraph, like bench.py.  Not a bench.py line (that is configs[3]); prints one JSON line.

    python tools/dsvi_cfg5_probe.py [--steps 10] [--warmup 3] [--M 2048] [--forward f32|bf16|bf16_all] [--no-f64acc]
    python bench.py --config cfg5 [--forward bf16]          # the same run as a bench.py JSON line
"""
import argparse
import json
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'nonstationary-precip_amd'))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from bench import GemmTimer, MFMA_F32_PEAK_TFLOPS  # noqa: E402


def run(args):
    dev = torch.device('cuda', torch.cuda.current_device())
    import models.dgps as dgps
    from nsgp import ops
    from nsgp.dist import PhiloxEps
    from nsgp.gp import settings
    from nsgp.gp.mlls import DeepApproximateMLL, VariationalELBO
    from nsgp.graph import GraphedCallable
    from nsgp.optim import FusedAdam

    n_side, seed = 100, 173
    g = torch.Generator().manual_seed(seed)
    ax = torch.linspace(-1.7, 1.7, n_side)
    x_all = torch.cartesian_prod(ax, ax, ax)                                   # (1e6, 3), z-scored lattice
    y_all = torch.sin(2.0 * x_all[:, 0]) * torch.cos(1.5 * x_all[:, 1]) * torch.exp(-0.3 * x_all[:, 2] ** 2) \
        + 0.1 * torch.randn(x_all.shape[0], generator=g)
    y_all = (y_all - y_all.mean()) / y_all.std()
    N, B, S = x_all.shape[0], args.batch, args.samples
    perm = torch.randperm(N, generator=g)
    nb = 8
    xs = [x_all[perm[i * B:(i + 1) * B]].to(dev) for i in range(nb)]
    ys = [y_all[perm[i * B:(i + 1) * B]].to(dev) for i in range(nb)]

    dgps.num_output_dims = 3
    torch.manual_seed(seed)
    model = dgps.DeepGP(2, (N, 3), num_inducing=args.M).to(dev)
    mll = DeepApproximateMLL(VariationalELBO(model.likelihood, model, N))
    opt = FusedAdam(model.parameters(), lr=0.01, capturable=True, grads_as_views=False)
    eps = PhiloxEps(seed, row0=0, step_dev=opt.step_dev)
    x_in, y_in = torch.empty_like(xs[0]), torch.empty_like(ys[0])
    model.train()

    def whole_step():
        eps.start_step(0, row0=0)
        opt.zero_grad()
        loss = -mll(model(x_in), y_in)
        loss.backward()
        opt.bucket.gather_grads()
        opt.step(gather=False)
        return loss.detach()

    with settings.num_likelihood_samples(S), settings.eps_provider(eps), settings.forward_precision(args.forward), \
            settings.whiten_matmul_f64(not args.no_f64acc):
        x_in.copy_(xs[0]); y_in.copy_(ys[0])
        with torch.no_grad():
            model(x_in)
        g_step = GraphedCallable(whole_step)
        losses = []
        for k in range(args.warmup):
            x_in.copy_(xs[k % nb]); y_in.copy_(ys[k % nb]); g_step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(args.steps):
            x_in.copy_(xs[k % nb]); y_in.copy_(ys[k % nb]); losses.append(g_step().clone())
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        timer = GemmTimer()
        ops.set_gemm_timer(timer)
        whole_step()
        ms32, fl32, n32 = timer.summary(torch.float32)
        ms64, fl64, n64 = timer.summary(torch.float64)
        msacc, flacc, nacc = timer.summary('f64acc')
        msbf, flbf, nbf = timer.summary('bf16')
        msi8, fli8, ni8 = timer.summary('i8')
        ops.set_gemm_timer(None)
    return {
        'workload': f'3-layer DSVI DeepGP (tied 3->3 hidden x2 + last 3->1), M={args.M}, S={S}, minibatch {B}, '
                    f'synthetic N={N} 3-D grid; fwd+ELBO+bwd+Adam, float32 with float64 Kzz Cholesky; forward '
                    f'projections: {args.forward}' + ('' if args.no_f64acc else ' (A = W Kzx as an exact int8 digit-plane product)'),
        'forward': args.forward, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(el / args.steps * 1e3, 2), 'steps_per_sec': round(args.steps / el, 2),
        'loss_first': round(float(losses[0]), 4), 'loss_last': round(float(losses[-1]), 4),
        'f32_gemm_ms_per_step': round(ms32, 2), 'f32_gemm_TFLOPs': round(fl32 / (ms32 * 1e-3) / 1e12, 1),
        'f32_gemm_frac_of_peak': round(fl32 / (ms32 * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, 3),
        'f32_gemm_launches': n32, 'f64_gemm_ms_per_step': round(ms64, 2), 'f64_gemm_launches': n64,
        'f64acc_gemm_ms_per_step': round(msacc, 2), 'f64acc_gemm_launches': nacc,
        'f64acc_gemm_TFLOPs': round(flacc / (msacc * 1e-3) / 1e12, 1) if nacc else None,
        'i8_gemm_ms_per_step': round(msi8, 2), 'i8_gemm_launches': ni8,
        'i8_gemm_TFLOPs_f64eq': round(fli8 / (msi8 * 1e-3) / 1e12, 1) if ni8 else None,
        'bf16_gemm_ms_per_step': round(msbf, 2), 'bf16_gemm_launches': nbf,
        'bf16_gemm_TFLOPs': round(flbf / (msbf * 1e-3) / 1e12, 1) if nbf else None,
        'hbm_peak_allocated_GB': round(torch.cuda.max_memory_allocated() / 1e9, 2)}


def parser():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--M', type=int, default=2048)
    ap.add_argument('--batch', type=int, default=4096)
    ap.add_argument('--samples', type=int, default=10)
    ap.add_argument('--forward', choices=('f32', 'bf16', 'bf16_all'), default='f32')
    ap.add_argument('--no-f64acc', action='store_true', help='float32 accumulation of A = W Kzx (round-1 arithmetic)')
    return ap


def main():
    print(json.dumps(run(parser().parse_args())))


if __name__ == '__main__':
    main()
